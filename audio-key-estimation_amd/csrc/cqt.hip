// CQT front end for gfx950: waveform -> log(1 + |constant-Q transform|).
//
// Replaces librosa.cqt + abs + log1p (KeyDataset.py:485,490-499).  Specification: the
// direct-form transform of oracle/cqt_oracle.py.  Evaluation (MI355X-first):
//
//   1. cqt_decimate_kernel   y_{o+1}[m] = sum_j h[j] y_o[2m+j]   (Kaiser half-band; one launch per octave
//                            step).  HBM-bound streaming: 16-byte loads, the tile is de-interleaved into
//                            even / odd samples in LDS (a half-band filter touches the centre sample and
//                            odd offsets only), each thread produces 4 outputs from 8 ds_read_b128 and
//                            stores them with one 16-byte store.
//   2. cqt_bank_kernel       for every (octave o, frame t): the 36 bins of the octave are
//                            <=277-tap complex FIRs on the 2^o-decimated signal.  A frame centre
//                            t*hop is not a multiple of 2^o in general, so the plan holds one
//                            filter bank per fractional phase (t*hop mod 2^o).  The bank apply is a
//                            GEMM  out[clip][72 re/im columns] = X[clip][taps] * W[taps][72]  on
//                            v_mfma_f32_16x16x4_f32 (exact fp32): M = 16 clips per wave, 5 N-tiles
//                            of 8 bins each with its own tap window (shorter filters skip their zero
//                            taps), W fragments staged through LDS per 64-tap chunk, each lane
//                            streams its clip's window with one 16-byte load per 4 k-steps.
//                            |.| (re/im pair via one lane shuffle) and log1p are the epilogue.
//
// HBM layout: audio [B][stride] f32 (caller's), y_o [B][len_o] f32 in the workspace
// (sample m stored at m + pad, pad = Hh rounded up to 4; len_o = ceil(n/2^o) + 2*pad rounded up to 4, so every
// row and every 512-output tile starts 16-byte aligned), out [B][n_bins][out_frames] f32.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <numeric>
#include <thread>
#include <vector>

#include "common.h"

namespace {

constexpr int kMaxOct = 12;
constexpr int kTileBins = 8;    // CQT bins per MFMA N-tile (16 columns = 8 bins x re/im)
constexpr int kMaxTiles = 5;    // N-tiles per octave (36 bins -> 4.5 tiles)
constexpr int kChunkBlocks = 4; // 16-tap blocks staged in LDS at a time
constexpr int kMaxOddTaps = 32; // decimator: odd taps 1,3,..,2*32-1
constexpr double kC1 = 32.70319566257483;
constexpr int kDecimOutPerBlock = 512;
constexpr int kDecimThreads = 128;   // x 4 outputs per thread

struct OctDesc {         // one octave of the filter bank in MFMA fragment order
    int k0;              // first CQT bin of the octave
    int n_bins;          // bins in the octave (36)
    int n_tiles;         // N-tiles (ceil(n_bins / 8))
    int uh;              // taps u = -uh .. +uh on the decimated grid (window of the longest filter)
    int n_blocks;        // 16-tap blocks covering the window
    int blk_lo[kMaxTiles], blk_hi[kMaxTiles];   // active tap blocks of each N-tile (shorter filters skip the rest)
    int table_off;       // float offset of phase 0: [phase][block][e][tile][64 lanes]
    int phase_stride;    // floats between consecutive phases
};

struct DecimTaps {
    float h0;
    float hodd[kMaxOddTaps];
    int n_odd;
    int half_len;
};

// engine 3 (bf16x3 filter bank), see cqt_bank_bf16_kernel
struct BankCall2 {
    const unsigned int* xw[kMaxOct];     // per level: one word per sample, (bf16 hi << 16) | bf16 lo, sample m at xw[m + pad]
    long long stride[kMaxOct];
    int pad;
};

struct OctDesc2 {
    int k0, n_bins, n_tiles, uh, n_blk;              // n_blk: 32-tap blocks
    int blk_lo[kMaxTiles], blk_hi[kMaxTiles];
    long long table_off, phase_stride;               // in 16-byte units: [phase][blk][tile][hi|lo][64 lanes] x 8 bf16
};

struct BankCall {
    const float* x[kMaxOct];   // per octave: sample m at x[m - lo]
    long long stride[kMaxOct];
    int lo[kMaxOct];
    int count[kMaxOct];
};

}  // namespace

#include "cqt_stream.h"

struct ake_cqt_plan {
    ake_cqt_config cfg;
    int n_oct;

    int hop_twos;       // trailing zero bits of hop
    int half_len;
    DecimTaps taps;
    std::vector<OctDesc> octs;
    OctDesc* octs_dev = nullptr;
    float* table_dev = nullptr;
    size_t table_floats = 0;
    int engine = 1;                      // resolved ake_cqt_config::engine (1, 2 or 3)
    std::vector<OctDesc2> octs2;         // engine 3: split-bf16 phase tables
    OctDesc2* octs2_dev = nullptr;
    uint4* table2_dev = nullptr;
    int ppad = 0;                        // pad of the split planes (covers every tap window)
    size_t bank2_lds = 0;
    uint4* toep5_dev = nullptr;          // engine 5: the four Toeplitz matrices of the streaming MFMA cascade (sm::build_toeplitz)
    int n_cu = 256;
};

// ------------------------------------------------------------------------------------------
// device code
// ------------------------------------------------------------------------------------------

typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));

typedef float f32x2v __attribute__((ext_vector_type(2)));

// 4 consecutive outputs of the half-band decimator from the de-interleaved input: P[i] = (odd sample 2i, 2i+1 of the window),
// i.e. O[j] = P[j/2][j%2] is the odd-sample stream starting NODD samples before output 0's centre; Cc = the 4 centre samples.
//   r[u] = h0*Cc[u] + sum_q hodd[q] * (O[u+NODD-1-q] + O[u+NODD+q])
// written for v_pk_add_f32 / v_pk_fma_f32: two taps (q, q+1) per instruction.  For even u the pairs (q even) are
// (swap(P[(u+NODD-2-q)/2]) + P[(u+NODD+q)/2]) * (hodd[q], hodd[q+1]); for odd u the same with q odd, plus the taps q = 0 and
// q = NODD-1 on their own.  Both decimator kernels use this one function, so their outputs are bit-identical.
template <int NODD>
__device__ __forceinline__ void halfband4(const f32x2v (&P)[NODD + 2], const float (&Cc)[4], const DecimTaps& taps, float (&r)[4]) {
    static_assert(NODD % 2 == 0, "tap pairs");
#pragma unroll
    for (int u = 0; u < 4; u += 2) {                                  // even outputs
        f32x2v acc = {0.f, 0.f};
#pragma unroll
        for (int q = 0; q < NODD; q += 2) {
            const f32x2v lo = P[(u + NODD - 2 - q) / 2], hi = P[(u + NODD + q) / 2];
            const f32x2v sum = __builtin_shufflevector(lo, lo, 1, 0) + hi;
            const f32x2v h = {taps.hodd[q], taps.hodd[q + 1]};
            acc = __builtin_elementwise_fma(h, sum, acc);
        }
        r[u] = fmaf(taps.h0, Cc[u], acc[0] + acc[1]);
    }
#pragma unroll
    for (int u = 1; u < 4; u += 2) {                                  // odd outputs
        f32x2v acc = {0.f, 0.f};
#pragma unroll
        for (int q = 1; q < NODD - 1; q += 2) {
            const f32x2v lo = P[(u + NODD - 2 - q) / 2], hi = P[(u + NODD + q) / 2];
            const f32x2v sum = __builtin_shufflevector(lo, lo, 1, 0) + hi;
            const f32x2v h = {taps.hodd[q], taps.hodd[q + 1]};
            acc = __builtin_elementwise_fma(h, sum, acc);
        }
        float tail = taps.hodd[0] * (P[(u + NODD - 1) / 2][(u + NODD - 1) % 2] + P[(u + NODD) / 2][(u + NODD) % 2]);
        tail = fmaf(taps.hodd[NODD - 1], P[u / 2][u % 2] + P[(u + 2 * NODD - 1) / 2][(u + 2 * NODD - 1) % 2], tail);
        r[u] = fmaf(taps.h0, Cc[u], (acc[0] + acc[1]) + tail);
    }
}

// y_out[m] = h0*y[2m] + sum_{j odd} h[j]*(y[2m-j] + y[2m+j]);  input sample s lives at in[s + in_pad], output sample m
// at out[m + out_pad].  One workgroup = 512 consecutive output indices of one clip.  Requires Hh % 4 == 3 and
// out_pad = Hh + 1 (checked by the host): then the first input the tile needs sits at local position 1 of a 16-byte
// aligned window, centres are the even local positions and all filter taps the odd ones.
template <int NODD>
__global__ __launch_bounds__(kDecimThreads) void cqt_decimate_kernel(
    const float* __restrict__ in, long long in_stride, int in_pad, int in_count,
    float* __restrict__ out, long long out_stride, int out_pad, int out_count, DecimTaps taps) {
    constexpr int H = 2 * NODD - 1;
    constexpr int NLOC = 2 * kDecimOutPerBlock + 2 * H + 2;         // local positions 0 .. NLOC-1 (position 1 = first needed)
    constexpr int NV4 = (NLOC + 3) / 4;
    __shared__ __attribute__((aligned(16))) float ev[NV4 * 2 + 4];   // even local positions (centres)
    __shared__ __attribute__((aligned(16))) float od[NV4 * 2 + 4];   // odd local positions (taps)
    const int clip = blockIdx.y;
    const int o0 = blockIdx.x * kDecimOutPerBlock;                   // first output index of the tile
    const int m0 = o0 - out_pad;                                     // ... and its sample number
    const int i_al = 2 * m0 - H + in_pad - 1;                        // input index of local position 0 (multiple of 4)
    const float* src = in + clip * in_stride;
    for (int v = threadIdx.x; v < NV4; v += kDecimThreads) {
        const int i0 = i_al + 4 * v;
        float x0, x1, x2, x3;
        if (i0 >= 0 && i0 + 4 <= in_count) {
            const f4u t = *reinterpret_cast<const f4u*>(src + i0);
            x0 = t[0]; x1 = t[1]; x2 = t[2]; x3 = t[3];
        } else {
            x0 = (i0 >= 0 && i0 < in_count) ? src[i0] : 0.f;
            x1 = (i0 + 1 >= 0 && i0 + 1 < in_count) ? src[i0 + 1] : 0.f;
            x2 = (i0 + 2 >= 0 && i0 + 2 < in_count) ? src[i0 + 2] : 0.f;
            x3 = (i0 + 3 >= 0 && i0 + 3 < in_count) ? src[i0 + 3] : 0.f;
        }
        *reinterpret_cast<float2*>(ev + 2 * v) = make_float2(x0, x2);
        *reinterpret_cast<float2*>(od + 2 * v) = make_float2(x1, x3);
    }
    __syncthreads();
    // outputs 4*tid .. 4*tid+3 of the tile: centre of output r at even slot 4*tid + r + NODD, taps at odd slots 4*tid + r + i
    const int tid = threadIdx.x;
    if (o0 + 4 * tid >= out_count) return;
    f32x2v P[NODD + 2];
    float C[4];
    {
        const float4* po = reinterpret_cast<const float4*>(od + 4 * tid);
#pragma unroll
        for (int i = 0; i < (2 * NODD + 4) / 4; ++i) {
            const float4 t = po[i];
            P[2 * i] = f32x2v{t.x, t.y}; P[2 * i + 1] = f32x2v{t.z, t.w};
        }
        const float4 c = *reinterpret_cast<const float4*>(ev + 4 * tid + NODD);
        C[0] = c.x; C[1] = c.y; C[2] = c.z; C[3] = c.w;
    }
    float r[4];
    halfband4<NODD>(P, C, taps, r);
    *reinterpret_cast<float4*>(out + clip * out_stride + o0 + 4 * tid) = make_float4(r[0], r[1], r[2], r[3]);
}


// ---- fused decimator cascade ---------------------------------------------------------------------------------------
// All half-band stages in ONE pass over the audio: a workgroup streams a segment of one clip in chunks of C samples
// ("ticks"), keeps every level's recent samples in LDS rings (de-interleaved into even / odd samples like the per-stage
// kernel) and emits, per tick, C/2 samples of level 1, C/4 of level 2, ... -- each level lags the one above by the filter
// half length, rounded so that every LDS access stays 16-byte aligned (kLagHost).  The audio is read from HBM once; the
// decimated signals are written once, and for the top levels only where a CQT frame's tap window will read them
// (a frame every `hop` samples touches ~6 % of the full-rate signal, 12 % of level 1, 25 % of level 2, ...).
// Arithmetic per output is identical to cqt_decimate_kernel (same operation order -> bit-identical signals).
constexpr int kCascMax = 7;                                            // stages the fused kernel handles
constexpr int kCascHist = 32;                                          // ring history in pair units (64 samples)
constexpr int kLagHost[kCascMax + 1] = {0, 16, 24, 24, 24, 24, 24, 24};

struct CascArgs {
    const float* x;            // [batch][x_stride] audio
    long long x_stride;
    int n;                     // samples per clip (ragged batches: of the longest clip)
    const long long* n_clip;   // ragged batches: samples of each clip (<= n), or null; the rest of a row reads as zero
    float* y[kCascMax + 1];    // level l = 1..n_stage: sample m at y[l][clip * y_stride[l] + m + pad]
    long long y_stride[kCascMax + 1];
    int y_count[kCascMax + 1]; // floats stored per clip (sample range [-pad, y_count - pad))
    int need[kCascMax + 1];    // level l is stored only within `need[l]` full-rate samples of a frame centre (< 0: everywhere)
    // split-bf16 copies for the bf16x3 filter bank (engine 3): value = hi + lo, both bf16, level 0 (the audio) included;
    // sample m at ph[l][clip * p_stride[l] + m + ppad] (hi and lo interleaved in one word, so that any tap window is a
    // dword-aligned vector load: 2-byte aligned 16-byte loads are split into 2-byte requests by the texture addresser,
    // measured 125 cache accesses per wave load).  null = not wanted.  The pads [-ppad, -pad) and
    // [p_count - ppad - ..., ) around the produced range are zero-filled so that the bank never checks bounds.
    unsigned int* ph[kCascMax + 1];      // one 32-bit word per sample: (bf16 hi << 16) | bf16 lo
    long long p_stride[kCascMax + 1];
    int p_count[kCascMax + 1];
    int ppad;
    int pad, hop, n_stage;
    int g0;                    // full-rate frontier before tick 0 (multiple of 512, <= -512)
    int ticks_total, ticks_per_seg, warm;
    DecimTaps taps;
    unsigned long long* stamps;   // diagnostic build: [4 waves][16] cycle sums per phase of a tick
};

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// 4 consecutive samples -> 4 words (bf16 hi << 16 | bf16 lo): hi = bf16(v) (RNE, v_cvt_pk_bf16_f32), lo = bf16(v - hi);
// v - (hi + lo) <= 2^-17 |v|
__device__ __forceinline__ void store_split4(unsigned int* pw, long long idx, float r0, float r1, float r2, float r3) {
    const f32x2 a = {r0, r1}, b = {r2, r3};
    const bf16x2 ha = __builtin_convertvector(a, bf16x2), hb = __builtin_convertvector(b, bf16x2);
    const f32x2 la = a - __builtin_convertvector(ha, f32x2), lb = b - __builtin_convertvector(hb, f32x2);
    const bf16x2 qa = __builtin_convertvector(la, bf16x2), qb = __builtin_convertvector(lb, bf16x2);
    const unsigned int h01 = __builtin_bit_cast(unsigned int, ha), h23 = __builtin_bit_cast(unsigned int, hb);
    const unsigned int l01 = __builtin_bit_cast(unsigned int, qa), l23 = __builtin_bit_cast(unsigned int, qb);
    // perm(a, b, sel): bytes 0-3 index b, 4-7 index a
    const uint4 w = make_uint4(__builtin_amdgcn_perm(h01, l01, 0x05040100), __builtin_amdgcn_perm(h01, l01, 0x07060302),
                               __builtin_amdgcn_perm(h23, l23, 0x05040100), __builtin_amdgcn_perm(h23, l23, 0x07060302));
    *reinterpret_cast<uint4*>(pw + idx) = w;
}

// LDS image of level l (an INPUT level, l = 0 .. S-1), in "pairs" (even sample 2h -> ev[h], odd sample 2h+1 -> od[h]):
//   [0, kCascHist) history carried over from the previous tick, [kCascHist, kCascHist + C/2^(l+1)) this tick's samples.
template <int C>
struct CascLayout {
    static constexpr int pairs(int l) { return (C >> (l + 1)) + kCascHist; }
    static constexpr int ev(int l) { int o = 0; for (int i = 0; i < l; ++i) o += 2 * pairs(i); return o; }
    static constexpr int od(int l) { return ev(l) + pairs(l); }
    static constexpr int total = 2 * (C - (C >> kCascMax)) + 2 * kCascMax * kCascHist;
};

template <int NODD, int C, int NT, int L, bool SPLIT>
__device__ __forceinline__ void cascade_level(const CascArgs& a, float* lds, int tid, int clip, int k, bool owned) {
    using Lay = CascLayout<C>;
    constexpr int chunk_out = C >> (L + 1);                          // outputs of level L+1 per tick
    constexpr int lag_in = L == 0 ? 0 : (L == 1 ? 16 : 24), lag_out = L == 0 ? 16 : 24;   // kLag[L], kLag[L+1]
    constexpr int E = (2 * lag_out - lag_in) / 2;                    // pair offset of the first new output's centre
    const float* ev = lds + Lay::ev(L);
    const float* od = lds + Lay::od(L);
    const int f_prev = ((a.g0 + k * C) >> (L + 1)) - lag_out;        // first sample of level L+1 produced this tick
    float* yrow = a.y[L + 1] ? a.y[L + 1] + clip * a.y_stride[L + 1] + a.pad : nullptr;
    const int need = a.need[L + 1];
    const float hopf = static_cast<float>(a.hop), inv_hop = 1.f / hopf;
#pragma unroll
    for (int j0 = 0; j0 < chunk_out / 4; j0 += NT) {
        const int j = j0 + tid;
        if (chunk_out / 4 - j0 < NT && j >= chunk_out / 4) break;
        f32x2v P[NODD + 2];
        float Cc[4];
        const float* po = od + kCascHist + 4 * j - E - NODD;
#pragma unroll
        for (int i = 0; i < (2 * NODD + 4) / 4; ++i) {
            const float4 t = *reinterpret_cast<const float4*>(po + 4 * i);
            P[2 * i] = f32x2v{t.x, t.y}; P[2 * i + 1] = f32x2v{t.z, t.w};
        }
        {
            const float4 c = *reinterpret_cast<const float4*>(ev + kCascHist + 4 * j - E);
            Cc[0] = c.x; Cc[1] = c.y; Cc[2] = c.z; Cc[3] = c.w;
        }
        float r[4];
        halfband4<NODD>(P, Cc, a.taps, r);
        if (L + 1 < kCascMax && L + 1 < a.n_stage) {
            float* evn = lds + Lay::ev(L + 1 < kCascMax ? L + 1 : 0) + kCascHist;
            float* odn = lds + Lay::od(L + 1 < kCascMax ? L + 1 : 0) + kCascHist;
            *reinterpret_cast<float2*>(evn + 2 * j) = make_float2(r[0], r[2]);
            *reinterpret_cast<float2*>(odn + 2 * j) = make_float2(r[1], r[3]);
        }
        const int m0 = f_prev + 4 * j;
        if (owned && m0 >= -a.pad && m0 + a.pad < a.y_count[L + 1]) {
            bool store = true;
            if (need >= 0) {                                         // distance to the nearest frame centre, full-rate samples
                const float pos = static_cast<float>(m0 * (1 << (L + 1)));
                store = fabsf(pos - rintf(pos * inv_hop) * hopf) <= static_cast<float>(need);
            }
            if (store) {
                if (!SPLIT) *reinterpret_cast<float4*>(yrow + m0) = make_float4(r[0], r[1], r[2], r[3]);
                else store_split4(a.ph[L + 1], clip * a.p_stride[L + 1] + m0 + a.ppad, r[0], r[1], r[2], r[3]);
            }
        }
    }
    // carry this level's last kCascHist pairs over to the history slots -- by threads from the top of the block, which have
    // no task in the deeper (smaller) levels; level L was last read in the previous phase, so there is no reader left
    if (L >= 1) {
        constexpr int LP = L - 1;                                    // the level whose consumer ran in the previous phase
        const int t2 = NT - 1 - tid;
        if (t2 < 2 * kCascHist / 4) {
            float* base = lds + (t2 < kCascHist / 4 ? Lay::ev(LP) : Lay::od(LP));
            const int v = t2 < kCascHist / 4 ? t2 : t2 - kCascHist / 4;
            const float4 h = *reinterpret_cast<const float4*>(base + (C >> (LP + 1)) + 4 * v);
            *reinterpret_cast<float4*>(base + 4 * v) = h;
        }
    }
}

__device__ __forceinline__ unsigned long long casc_stamp() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}

// STAMP: diagnostic build (AKE_CQT_CASC_STAMP): s_memtime stamps around the phases of a tick, workgroup (0, 0); never timed
template <int NODD, int C, int NT, bool SPLIT, bool STAMP = false>
__global__ __launch_bounds__(NT) void cqt_cascade_kernel(CascArgs a) {
    static_assert(C % 512 == 0 && C >= 1024 && (C / 4) % NT == 0, "chunk");
    using Lay = CascLayout<C>;
    __shared__ __attribute__((aligned(16))) float lds[Lay::total];
    const int tid = threadIdx.x;
    const int clip = blockIdx.y;
    const int S = a.n_stage;
    const int k_own = blockIdx.x * a.ticks_per_seg;
    const int k_end = k_own + a.ticks_per_seg < a.ticks_total ? k_own + a.ticks_per_seg : a.ticks_total;
    const int k_start = k_own - a.warm > 0 ? k_own - a.warm : 0;
    if (k_own >= a.ticks_total) return;
    for (int i = tid; i < Lay::total; i += NT) lds[i] = 0.f;
    const float* xs = a.x + clip * a.x_stride;

    constexpr int G = C / 4 / NT;                                     // float4 groups of the audio chunk per thread
    // Audio through a buffer resource: hardware range checking returns 0 for every dword outside [0, n) -- the zero padding
    // of the transform's definition -- so the prefetch is branch-free and all G loads are in flight together.
    int n_here = a.n;
    if (a.n_clip) { const long long nc = a.n_clip[clip]; n_here = nc < 0 ? 0 : (nc < a.n ? static_cast<int>(nc) : a.n); }
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xs), 0, n_here * 4, 0x00020000);
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    f4u pre[G];
    auto fetch = [&](int k) {                                          // audio samples [F0(k-1), F0(k))
        const int s0 = a.g0 + k * C;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int i0 = s0 + 4 * (tid + NT * g);                   // negative -> huge unsigned offset -> out of range -> 0
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, i0 * 4, 0, 0);
            pre[g] = __builtin_bit_cast(f4u, v);
        }
    };
    fetch(k_start);
    if (SPLIT) {   // zero the pads of the split signals outside the produced range [-pad, y_count - pad) (first / last segment)
        const bool first = blockIdx.x == 0, last = k_end == a.ticks_total;
        for (int l = 0; l <= S; ++l) {
            if (!a.ph[l]) continue;
            const int lo_end = l == 0 ? 0 : -a.pad;                                   // produced range starts here
            const int hi_beg = l == 0 ? (a.n + 3) / 4 * 4 : a.y_count[l] - a.pad;     // ... and ends here (level 0: audio, rounded up)
            unsigned int* h = a.ph[l] + clip * a.p_stride[l];
            if (first) for (int i = 4 * tid; i < lo_end + a.ppad; i += 4 * NT) *reinterpret_cast<uint4*>(h + i) = make_uint4(0, 0, 0, 0);
            if (last) for (int i = hi_beg + a.ppad + 4 * tid; i < a.p_count[l]; i += 4 * NT) *reinterpret_cast<uint4*>(h + i) = make_uint4(0, 0, 0, 0);
        }
    }
    const float hopf0 = static_cast<float>(a.hop), inv_hop0 = 1.f / hopf0;
    __syncthreads();
    unsigned long long sm[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, ts[10];
    for (int k = k_start; k < k_end; ++k) {
        if (STAMP) {
            ts[0] = casc_stamp();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // how long the prefetched chunk is still outstanding
            sm[9] += casc_stamp() - ts[0];
        }
        const bool owned = k >= k_own;
        if (S == 1 && k > k_start) {                                  // single stage: level 0 is also the deepest level
            float4 h = {0.f, 0.f, 0.f, 0.f};
            if (tid < 2 * kCascHist / 4) h = *reinterpret_cast<const float4*>(lds + (tid < kCascHist / 4 ? Lay::ev(0) : Lay::od(0)) + (C >> 1) + 4 * (tid % (kCascHist / 4)));
            __syncthreads();
            if (tid < 2 * kCascHist / 4) *reinterpret_cast<float4*>(lds + (tid < kCascHist / 4 ? Lay::ev(0) : Lay::od(0)) + 4 * (tid % (kCascHist / 4))) = h;
        }
        {   // level 0: the prefetched chunk (pairs: even sample -> ev, odd -> od)
            float* ev = lds + Lay::ev(0) + kCascHist;
            float* od = lds + Lay::od(0) + kCascHist;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int p = 2 * (tid + NT * g);
                *reinterpret_cast<float2*>(ev + p) = make_float2(pre[g][0], pre[g][2]);
                *reinterpret_cast<float2*>(od + p) = make_float2(pre[g][1], pre[g][3]);
                if (SPLIT && owned) {                                 // split copy of the audio where the top octave's windows read it
                    const int i0 = a.g0 + k * C + 4 * (tid + NT * g);
                    if (i0 >= 0 && i0 < a.n) {
                        bool store = true;
                        if (a.need[0] >= 0) {
                            const float pos = static_cast<float>(i0);
                            store = fabsf(pos - rintf(pos * inv_hop0) * hopf0) <= static_cast<float>(a.need[0]);
                        }
                        if (store) store_split4(a.ph[0], clip * a.p_stride[0] + i0 + a.ppad, pre[g][0], pre[g][1], pre[g][2], pre[g][3]);
                    }
                }
            }
            // ... and the deepest input level's history (its consumer ran in the last phase of the previous tick)
            if (S > 1 && k > k_start) {
                const int LP = S - 1;
                const int t2 = NT - 1 - tid;
                if (t2 < 2 * kCascHist / 4) {
                    int evo = 0;
                    for (int i = 0; i < LP; ++i) evo += 2 * ((C >> (i + 1)) + kCascHist);
                    const int pairs = (C >> (LP + 1)) + kCascHist;
                    float* base = lds + evo + (t2 < kCascHist / 4 ? 0 : pairs);
                    const int v = t2 < kCascHist / 4 ? t2 : t2 - kCascHist / 4;
                    const float4 h = *reinterpret_cast<const float4*>(base + (C >> (LP + 1)) + 4 * v);
                    *reinterpret_cast<float4*>(base + 4 * v) = h;
                }
            }
        }
        if (k + 1 < k_end) fetch(k + 1);                              // next chunk in flight during the whole tick
        __syncthreads();
        if (STAMP) ts[1] = casc_stamp();
        cascade_level<NODD, C, NT, 0, SPLIT>(a, lds, tid, clip, k, owned);
        __syncthreads();
        if (STAMP) ts[2] = casc_stamp();
        if (S > 1) { cascade_level<NODD, C, NT, 1, SPLIT>(a, lds, tid, clip, k, owned); __syncthreads(); }
        if (STAMP) ts[3] = casc_stamp();
        if (S > 2) { cascade_level<NODD, C, NT, 2, SPLIT>(a, lds, tid, clip, k, owned); __syncthreads(); }
        if (STAMP) ts[4] = casc_stamp();
        if (S > 3) { cascade_level<NODD, C, NT, 3, SPLIT>(a, lds, tid, clip, k, owned); __syncthreads(); }
        if (STAMP) ts[5] = casc_stamp();
        if (S > 4) { cascade_level<NODD, C, NT, 4, SPLIT>(a, lds, tid, clip, k, owned); __syncthreads(); }
        if (STAMP) ts[6] = casc_stamp();
        if (S > 5) { cascade_level<NODD, C, NT, 5, SPLIT>(a, lds, tid, clip, k, owned); __syncthreads(); }
        if (STAMP) ts[7] = casc_stamp();
        if (S > 6) { cascade_level<NODD, C, NT, 6, SPLIT>(a, lds, tid, clip, k, owned); __syncthreads(); }
        if (STAMP) {
            ts[8] = casc_stamp();
#pragma unroll
            for (int i = 0; i < 8; ++i) sm[i] += ts[i + 1] - ts[i];
            sm[8] += 1;
        }
    }
    if (STAMP && blockIdx.x == 1 && blockIdx.y == 0 && (tid & 63) == 0 && a.stamps) {
#pragma unroll
        for (int i = 0; i < 10; ++i) a.stamps[(tid >> 6) * 16 + i] = sm[i];
    }
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Workgroup = (frame t, octave o, 64 clips): 4 waves x one 16-clip M-tile, all N-tiles of the octave.
// A[m = clip][k]: lane (r = lane&15, q = lane>>4) loads X[clip r][16*blk + 4q .. +3] (one dwordx4 per 16 taps);
// component e of that vector is the A operand of k-step e, i.e. MFMA k index q <-> tap 16*blk + 4q + e, and the
// W fragments are packed on the host with the same tap permutation.
__global__ __launch_bounds__(256) void cqt_bank_kernel(
    BankCall call, const OctDesc* __restrict__ octs, const float* __restrict__ table,
    int batch, int hop, int hop_twos, float* __restrict__ out, long long out_clip_stride, int n_bins_total) {
    __shared__ __attribute__((aligned(16))) float ldsB[kChunkBlocks * 4 * kMaxTiles * 64];
    const int t = blockIdx.x;
    const int o = blockIdx.y;
    const OctDesc g = octs[o];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r16 = lane & 15, q = lane >> 4;
    const int clip_raw = blockIdx.z * 64 + wave * 16 + r16;
    const int clip = clip_raw < batch ? clip_raw : batch - 1;       // idle rows redo the last clip (never stored)
    const long long c = static_cast<long long>(t) * hop;
    const int c_int = static_cast<int>(c >> o);
    const int ph = static_cast<int>(c & ((1ll << o) - 1));
    const int sh = hop_twos < o ? hop_twos : o;
    const float* __restrict__ w = table + g.table_off + static_cast<long long>(ph >> sh) * g.phase_stride;
    const int lo = call.lo[o], cnt = call.count[o];
    const float* __restrict__ x = call.x[o] + clip * call.stride[o];
    const int s_first = c_int - g.uh - lo;                           // array index of tap 0 of the window

    f32x4 acc[kMaxTiles];
#pragma unroll
    for (int j = 0; j < kMaxTiles; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto load_x = [&](int blk) -> f32x4 {
        const int i0 = s_first + 16 * blk + 4 * q;
        if (i0 >= 0 && i0 + 4 <= cnt) {
            const f4u v = *reinterpret_cast<const f4u*>(x + i0);
            return f32x4{v[0], v[1], v[2], v[3]};
        }
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (i0 + e >= 0 && i0 + e < cnt) ? x[i0 + e] : 0.f;   // zero padding outside the clip
        return v;
    };

    constexpr int BF = 4 * kMaxTiles * 64;                           // floats per tap block
    f32x4 xv = load_x(0);
    for (int b0 = 0; b0 < g.n_blocks; b0 += kChunkBlocks) {
        const int nb = g.n_blocks - b0 < kChunkBlocks ? g.n_blocks - b0 : kChunkBlocks;
        if (b0) __syncthreads();
        {   // stage the W fragments of this chunk (contiguous in global memory)
            const float4* src = reinterpret_cast<const float4*>(w + static_cast<long long>(b0) * BF);
            float4* dst = reinterpret_cast<float4*>(ldsB);
            for (int i = threadIdx.x; i < nb * BF / 4; i += 256) dst[i] = src[i];
        }
        __syncthreads();
        for (int bi = 0; bi < nb; ++bi) {
            const int blk = b0 + bi;
            const f32x4 xc = xv;
            xv = load_x(blk + 1 < g.n_blocks ? blk + 1 : blk);       // next block's window, in flight during the MFMAs
#pragma unroll
            for (int j = 0; j < kMaxTiles; ++j) {
                if (j < g.n_tiles && blk >= g.blk_lo[j] && blk <= g.blk_hi[j]) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(xc[e], ldsB[((bi * 4 + e) * kMaxTiles + j) * 64 + lane], acc[j], 0, 0, 0);
                }
            }
        }
    }
    // epilogue: D[row = clip 4q+reg][col r16]: col = 2*(bin within tile) + (re|im); |C| needs the neighbouring lane
#pragma unroll
    for (int j = 0; j < kMaxTiles; ++j) {
        const int b = kTileBins * j + (r16 >> 1);
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const float v2 = acc[j][reg] * acc[j][reg];
            const float m2 = v2 + __shfl_xor(v2, 1);
            const int cl = blockIdx.z * 64 + wave * 16 + 4 * q + reg;
            // scratch layout [clip][frame][bin]: the 8 bins of a tile are 32 contiguous bytes, the octave 144
            if ((r16 & 1) == 0 && j < g.n_tiles && b < g.n_bins && cl < batch)
                out[cl * out_clip_stride + static_cast<long long>(t) * n_bins_total + g.k0 + b] = log1pf(sqrtf(m2));
        }
    }
}

// ---- filter bank on bf16 MFMA with split operands ("bf16x3") -----------------------------------------------------------
// x = xh + xl and w = wh + wl (each bf16, residual <= 2^-17 relative); acc += xh*wh + xl*wh + xh*wl on
// v_mfma_f32_16x16x32_bf16 (f32 accumulation): 3 MFMAs of 16 cycles per 32 taps and N-tile instead of 8 f32 MFMAs of 32
// cycles, at ~1e-5 relative accuracy (the multirate design itself is specified to 1.6e-4).
// Workgroup = (frame t, octave o, kBankClips clips): the phase table of (o, t) is staged in LDS in chunks (51 KB) and shared
// by the workgroup's waves, one 16-clip M-tile each.  A operand: lane (row r = clip, q) holds taps 32*blk + 8q .. +7 of its clip =
// two dword-aligned 16-byte loads of interleaved (hi, lo) words, de-interleaved with 8 v_perm_b32 (the signals are padded,
// no bounds checks).
constexpr int kBankWaves = 8;           // waves (M-tiles of 16 clips) per workgroup: 128 clips.  (16 waves = 256 clips read each phase table once, but at 80
                                        // registers only ONE such workgroup fits a CU; three of these do, and a workgroup's two table chunks are latency chains)
constexpr int kBankClips = 16 * kBankWaves, kBankThreads = 64 * kBankWaves;
constexpr int kBank2Chunk = 5;          // 32-tap blocks of the phase table resident in LDS at a time (51 KB: two workgroups per CU)
typedef unsigned int u32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4b __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(kBankThreads) void cqt_bank_bf16_kernel(
    BankCall2 call, const OctDesc2* __restrict__ octs, const uint4* __restrict__ table,
    int batch, int hop, int hop_twos, float* __restrict__ out, long long out_clip_stride, int n_bins_total, int n_frames, int o_first) {
    extern __shared__ __attribute__((aligned(16))) uint4 ldsW[];
    // gridDim.x is a multiple of 8, so blockIdx.x % 8 is the XCD: consecutive frames go to ONE XCD, whose L2 then serves the parts of
    // their tap windows that overlap (the windows of octaves 5..7 overlap 2-8 fold)
    const int per_xcd = gridDim.x >> 3;
    const int t = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (t >= n_frames) return;
    const int o = blockIdx.y + o_first;
    const OctDesc2 g = octs[o];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r16 = lane & 15, q = lane >> 4;
    const int clip_raw = blockIdx.z * kBankClips + wave * 16 + r16;
    const int clip = clip_raw < batch ? clip_raw : batch - 1;       // idle rows redo the last clip (never stored)
    const long long c = static_cast<long long>(t) * hop;
    const int c_int = static_cast<int>(c >> o);
    const int ph = static_cast<int>(c & ((1ll << o) - 1));
    const int sh = hop_twos < o ? hop_twos : o;
    const uint4* __restrict__ w = table + g.table_off + static_cast<long long>(ph >> sh) * g.phase_stride;
    const long long base = clip * call.stride[o] + (c_int - g.uh + call.pad) + 8 * q;
    const unsigned int* __restrict__ xw = call.xw[o] + base;
    f32x4b acc[kMaxTiles];
#pragma unroll
    for (int j = 0; j < kMaxTiles; ++j) acc[j] = f32x4b{0.f, 0.f, 0.f, 0.f};
    const bool idle = blockIdx.z * kBankClips + wave * 16 >= batch;        // whole M-tile beyond the batch: only helps staging
#pragma unroll 1                                                      // (unrolled by two the chunks' A fragments are live together: 485 spills)
    for (int b0 = 0; b0 < g.n_blk; b0 += kBank2Chunk) {             // the phase table in chunks of kBank2Chunk blocks
        const int nbc = g.n_blk - b0 < kBank2Chunk ? g.n_blk - b0 : kBank2Chunk;
        // every A fragment of the chunk is requested before anything else: 2 x 16 B per lane and block, all in flight together
        // (a wave that prefetches one block ahead keeps 2 KB in flight -- the kernel then runs at the HBM latency, not bandwidth)
        u32x4u xa[kBank2Chunk], xb[kBank2Chunk];
#pragma unroll
        for (int bi = 0; bi < kBank2Chunk; ++bi) {
            const int blk = b0 + (bi < nbc ? bi : nbc - 1);
            xa[bi] = *reinterpret_cast<const u32x4u*>(xw + 32 * blk);
            xb[bi] = *reinterpret_cast<const u32x4u*>(xw + 32 * blk + 4);
        }
        if (b0) __syncthreads();
        {
            // LDS-DMA (global_load_lds_dwordx4): the chunk's pieces go straight to the LDS, all in flight at once and without registers.  As a loop
            // of load -> LDS store per iteration this was a chain of up to four dependent memory round trips per chunk behind the A fragments' one
            // (0.070 -> 0.066 ms per 256 clips).
            const uint4* src = w + static_cast<long long>(b0) * (kMaxTiles * 2 * 64);
            const int total = nbc * kMaxTiles * 2 * 64;                      // a multiple of 64: a wave's piece is whole or absent
            constexpr int kU = (kBank2Chunk * kMaxTiles * 2 * 64 + kBankThreads - 1) / kBankThreads;
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                const int i0 = wave * 64 + kBankThreads * u;
                if (i0 < total) {
                    const uint4* sp = src + i0 + lane;
                    const unsigned int lds_dst = static_cast<unsigned int>(reinterpret_cast<unsigned long long>(ldsW + i0));   // LDS aperture: low 32 bits = byte address
                    unsigned int keep;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "v"(sp), "s"(lds_dst) : "memory");
                }
            }
            __builtin_amdgcn_s_waitcnt(0x0F70);                              // vmcnt(0): this wave's pieces (and its A fragments) have landed
        }
        __syncthreads();
        if (idle) continue;
#pragma unroll
        for (int bi = 0; bi < kBank2Chunk; ++bi) {
            if (bi < nbc) {
                const int blk = b0 + bi;
                // de-interleave 8 words into the hi and the lo operand (v_perm_b32: bytes 0-3 index the 2nd source, 4-7 the 1st)
                const u32x4u va = xa[bi], vb = xb[bi];
                const u32x4u hv = {__builtin_amdgcn_perm(va[1], va[0], 0x07060302), __builtin_amdgcn_perm(va[3], va[2], 0x07060302),
                                   __builtin_amdgcn_perm(vb[1], vb[0], 0x07060302), __builtin_amdgcn_perm(vb[3], vb[2], 0x07060302)};
                const u32x4u lv = {__builtin_amdgcn_perm(va[1], va[0], 0x05040100), __builtin_amdgcn_perm(va[3], va[2], 0x05040100),
                                   __builtin_amdgcn_perm(vb[1], vb[0], 0x05040100), __builtin_amdgcn_perm(vb[3], vb[2], 0x05040100)};
                const bf16x8 ch = __builtin_bit_cast(bf16x8, hv), cl = __builtin_bit_cast(bf16x8, lv);
#pragma unroll
                for (int j = 0; j < kMaxTiles; ++j) {
                    if (j < g.n_tiles && blk >= g.blk_lo[j] && blk <= g.blk_hi[j]) {
                        const bf16x8 bh = __builtin_bit_cast(bf16x8, ldsW[((bi * kMaxTiles + j) * 2 + 0) * 64 + lane]);
                        const bf16x8 bl = __builtin_bit_cast(bf16x8, ldsW[((bi * kMaxTiles + j) * 2 + 1) * 64 + lane]);
                        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ch, bh, acc[j], 0, 0, 0);
                        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cl, bh, acc[j], 0, 0, 0);
                        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ch, bl, acc[j], 0, 0, 0);
                    }
                }
            }
        }
    }
    if (idle) return;
#pragma unroll
    for (int j = 0; j < kMaxTiles; ++j) {
        const int b = kTileBins * j + (r16 >> 1);
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const float v2 = acc[j][reg] * acc[j][reg];
            const float m2 = v2 + __shfl_xor(v2, 1);
            const int cl2 = blockIdx.z * kBankClips + wave * 16 + 4 * q + reg;
            // KeyDataset.py:497-499 is literally log(1 + |C|); v_sqrt_f32 / v_log_f32 (1 ulp class) instead of libm's log1pf
            if ((r16 & 1) == 0 && j < g.n_tiles && b < g.n_bins && cl2 < batch)
                out[cl2 * out_clip_stride + static_cast<long long>(t) * n_bins_total + g.k0 + b] = __logf(1.f + __builtin_amdgcn_sqrtf(m2));
        }
    }
}

// [clip][frame][bin] scratch -> [clip][bin][out_frames] (the reference layout), zero-filling frames >= T
// (KeyDataset.py:245 padding).  32x32 LDS tile transpose: coalesced on both sides.
__global__ __launch_bounds__(256) void cqt_transpose_kernel(const float* __restrict__ src, float* __restrict__ dst, int T,
                                                            int n_bins, int out_frames, const long long* __restrict__ n_clip, int hop) {
    __shared__ float tile[32][33];
    const int clip = blockIdx.z;
    int Tc = T;                                                        // ragged batches: frames of THIS clip; the rest is zero padding
    if (n_clip) { const long long nc = n_clip[clip]; const long long tc = nc < 0 ? 0 : 1 + nc / hop; Tc = tc < T ? static_cast<int>(tc) : T; }
    const int k0 = blockIdx.x * 32, t0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;           // 32 x 8
    const float* s = src + static_cast<long long>(clip) * T * n_bins;
    float* d = dst + static_cast<long long>(clip) * n_bins * out_frames;
#pragma unroll
    for (int r = 0; r < 32; r += 8) {
        const int t = t0 + ty + r, k = k0 + tx;
        tile[ty + r][tx] = (t < Tc && k < n_bins) ? s[static_cast<long long>(t) * n_bins + k] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 32; r += 8) {
        const int k = k0 + ty + r, t = t0 + tx;
        if (k < n_bins && t < out_frames) d[static_cast<long long>(k) * out_frames + t] = tile[tx][ty + r];
    }
}

// ------------------------------------------------------------------------------------------
// host: plan
// ------------------------------------------------------------------------------------------
namespace {

double bessel_i0(double x) {
    double sum = 1.0, term = 1.0;
    const double q = x * x / 4.0;
    for (int k = 1; k < 200; ++k) {
        term *= q / (static_cast<double>(k) * k);
        sum += term;
        if (term < 1e-18 * sum) break;
    }
    return sum;
}

std::vector<double> kaiser_halfband(int half_len, double beta) {
    std::vector<double> h(2 * half_len + 1);
    double s = 0;
    for (int j = -half_len; j <= half_len; ++j) {
        const double x = j / 2.0;
        const double sinc = (j == 0) ? 1.0 : std::sin(M_PI * x) / (M_PI * x);
        const double r = static_cast<double>(j) / half_len;
        const double win = bessel_i0(beta * std::sqrt(std::max(0.0, 1.0 - r * r))) / bessel_i0(beta);
        h[j + half_len] = 0.5 * sinc * win;
        s += h[j + half_len];
    }
    for (auto& v : h) v /= s;
    return h;
}

int pad_of(const ake_cqt_plan* p) { return p->half_len + 1; }   // Hh % 4 == 3  ->  multiple of 4

int len_store(const ake_cqt_plan* p, int o, int64_t n) {   // floats stored per clip for octave o >= 1 (multiple of 4)
    const int64_t l = (n + (1ll << o) - 1) >> o;
    return (static_cast<int>(l) + 2 * pad_of(p) + 3) / 4 * 4;
}

// level 4 handed from a first launch to the deep levels' cascade launch: its sample m sits at input index m + kNextPad (engine 3's two-launch
// cascade reads level 4's split plane that way, engine 5 an f32 copy)
constexpr int kNextPad = 24;
int next_len(int64_t n) { return (static_cast<int>((n + 15) / 16) + 2 * kNextPad + 3) / 4 * 4; }

int plane_len(const ake_cqt_plan* p, int l, int64_t n) {   // split words per clip of level l (multiple of 8)
    const int64_t ln = (n + (1ll << l) - 1) >> l;
    return (static_cast<int>(ln) + 2 * pad_of(p) + 2 * p->ppad + 8 + 7) / 8 * 8;
}

}  // namespace

extern "C" {

int ake_cqt_default_config(ake_cqt_config* cfg, int sample_rate, int frames_per_second, int octaves) {
    AKE_REQUIRE(cfg && sample_rate > 0 && octaves > 0, AKE_ERR_INVALID, "ake_cqt_default_config: bad argument");
    std::memset(cfg, 0, sizeof(*cfg));
    cfg->sample_rate = sample_rate;
    // KeyDataset.py:485  hop_length = round(rate / (opt.frames if opt.frames > 0 else 1)); Python's
    // round() is round-half-even
    const double q = static_cast<double>(sample_rate) / (frames_per_second > 0 ? frames_per_second : 1);
    cfg->hop_length = static_cast<int>(std::nearbyint(q));
    cfg->n_bins = 36 * octaves;
    cfg->bins_per_octave = 36;
    cfg->fmin = kC1;
    cfg->q_mode = 0;
    cfg->decim_half_len = 23;
    cfg->decim_beta = 8.0;
    return AKE_OK;
}

int ake_cqt_plan_create(const ake_cqt_config* cfg_in, ake_cqt_plan** out) {
    AKE_REQUIRE(cfg_in && out, AKE_ERR_INVALID, "ake_cqt_plan_create: null argument");
    ake_cqt_config cfg = *cfg_in;
    if (cfg.fmin <= 0) cfg.fmin = kC1;
    if (cfg.decim_half_len <= 0) cfg.decim_half_len = 23;
    if (cfg.decim_beta <= 0) cfg.decim_beta = 8.0;
    AKE_REQUIRE(cfg.sample_rate > 0 && cfg.hop_length > 0 && cfg.n_bins > 0, AKE_ERR_INVALID, "cqt: bad rate/hop/bins");
    AKE_REQUIRE(cfg.n_bins % cfg.bins_per_octave == 0 && cfg.bins_per_octave <= kTileBins * kMaxTiles, AKE_ERR_UNSUPPORTED,
                "cqt: bins_per_octave must divide n_bins and be <= %d", kTileBins * kMaxTiles);
    AKE_REQUIRE(cfg.decim_half_len == 15 || cfg.decim_half_len == 23 || cfg.decim_half_len == 31, AKE_ERR_INVALID,
                "cqt: decim_half_len must be 15, 23 or 31 (half-band taps, 4k+3 keeps every tile 16-byte aligned)");
    const int bpo = cfg.bins_per_octave;
    const int n_oct = cfg.n_bins / bpo;
    AKE_REQUIRE(n_oct <= kMaxOct, AKE_ERR_INVALID, "cqt: too many octaves (%d)", n_oct);
    const double sr = cfg.sample_rate;
    const double r = std::pow(2.0, 1.0 / bpo);
    const double Q = cfg.q_mode == 0 ? (r * r + 1.0) / (r * r - 1.0) : 1.0 / (r - 1.0);
    std::vector<double> freq(cfg.n_bins), len(cfg.n_bins);
    for (int k = 0; k < cfg.n_bins; ++k) {
        freq[k] = cfg.fmin * std::pow(2.0, static_cast<double>(k) / bpo);
        len[k] = Q * sr / freq[k];
    }
    // librosa refuses a filter bank whose top filter passes Nyquist (hann bandwidth 1.50018)
    const double cutoff = freq.back() * (1.0 + 0.5 * 1.50018310546875 / Q);
    AKE_REQUIRE(cutoff <= sr / 2.0, AKE_ERR_INVALID, "cqt: top bin cutoff %.1f Hz exceeds Nyquist %.1f Hz", cutoff, sr / 2.0);

    auto* p = new ake_cqt_plan();
    p->cfg = cfg;
    p->n_oct = n_oct;
    p->half_len = cfg.decim_half_len;
    p->hop_twos = 0;
    while (((cfg.hop_length >> p->hop_twos) & 1) == 0) ++p->hop_twos;

    const std::vector<double> h = kaiser_halfband(p->half_len, cfg.decim_beta);
    std::memset(&p->taps, 0, sizeof(p->taps));
    p->taps.half_len = p->half_len;
    p->taps.h0 = static_cast<float>(h[p->half_len]);
    p->taps.n_odd = (p->half_len + 1) / 2;
    for (int q = 0; q < p->taps.n_odd; ++q) p->taps.hodd[q] = static_cast<float>(h[p->half_len + 2 * q + 1]);
    // the even taps of a windowed half-band sinc are exactly zero apart from the centre; what the
    // float kernel drops is below 1e-17, and the per-bin gain below is computed from the taps it keeps
    auto cascade_gain = [&](double f_hz, int o) {
        double gain = 1.0;
        for (int s = 0; s < o; ++s) {
            const double wn = 2.0 * M_PI * f_hz / (sr / std::pow(2.0, s));
            double acc = p->taps.h0;
            for (int q = 0; q < p->taps.n_odd; ++q) acc += 2.0 * p->taps.hodd[q] * std::cos(wn * (2 * q + 1));
            gain *= std::fabs(acc);
        }
        return gain;
    };

    std::vector<float> table;
    for (int o = 0; o < n_oct; ++o) {
        const int dec = 1 << o;
        const int sh = std::min(p->hop_twos, o);
        const int nph = dec >> sh;
        OctDesc g;
        std::memset(&g, 0, sizeof(g));
        g.k0 = cfg.n_bins - bpo * (o + 1);
        g.n_bins = bpo;
        g.n_tiles = (bpo + kTileBins - 1) / kTileBins;
        auto uh_of = [&](int k) { return static_cast<int>(std::ceil(-std::floor(-len[k] / 2.0) / dec)) + 1; };
        g.uh = uh_of(g.k0);                                    // lowest bin of the octave = longest filter
        g.n_blocks = (2 * g.uh + 1 + 15) / 16;
        for (int j = 0; j < g.n_tiles; ++j) {
            const int uh_j = uh_of(g.k0 + kTileBins * j);       // longest filter of the tile
            g.blk_lo[j] = (g.uh - uh_j) / 16;
            g.blk_hi[j] = std::min(g.n_blocks - 1, (g.uh + uh_j) / 16);
        }
        constexpr int BF = 4 * kMaxTiles * 64;
        g.phase_stride = g.n_blocks * BF;
        g.table_off = static_cast<int>(table.size());
        table.resize(table.size() + static_cast<size_t>(nph) * g.phase_stride, 0.f);
        for (int pi = 0; pi < nph; ++pi) {
            const double ph = static_cast<double>(pi << sh);
            float* dst = table.data() + g.table_off + static_cast<size_t>(pi) * g.phase_stride;
            for (int b = 0; b < bpo; ++b) {
                const int k = g.k0 + b;
                const int j = b / kTileBins;
                const double lo = std::floor(-len[k] / 2.0);
                const double L = std::floor(len[k] / 2.0) - lo;
                const double scale = dec * std::sqrt(len[k]) / (L / 2.0) / cascade_gain(freq[k], o);
                for (int tap = 0; tap < 16 * g.n_blocks; ++tap) {
                    const int u = tap - g.uh;
                    const double pos = static_cast<double>(dec) * u - ph;   // full-rate offset from the frame centre
                    if (!(pos >= lo && pos <= lo + L)) continue;
                    const double win = 0.5 - 0.5 * std::cos(2.0 * M_PI * (pos - lo) / L);
                    const double arg = 2.0 * M_PI * freq[k] * pos / sr;
                    const int blk = tap / 16, qq = (tap % 16) / 4, e = tap % 4;   // MFMA k index qq <-> tap 16*blk + 4*qq + e
                    const int col = 2 * (b % kTileBins);
                    float* f = dst + ((static_cast<size_t>(blk) * 4 + e) * kMaxTiles + j) * 64 + qq * 16;
                    f[col] = static_cast<float>(scale * win * std::cos(arg));
                    f[col + 1] = static_cast<float>(-scale * win * std::sin(arg));
                }
            }
        }
        p->octs.push_back(g);
    }
    p->table_floats = table.size();
    hipError_t e = hipMalloc(&p->table_dev, table.size() * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&p->octs_dev, p->octs.size() * sizeof(OctDesc));
    if (e == hipSuccess) e = hipMemcpy(p->table_dev, table.data(), table.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(p->octs_dev, p->octs.data(), p->octs.size() * sizeof(OctDesc), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        ake::set_error("cqt plan upload failed: %s", hipGetErrorString(e));
        ake_cqt_plan_destroy(p);
        return AKE_ERR_HIP;
    }
    // ---- engine ----
    {
        int want = cfg.engine;
        if (const char* e = ake::diag_env("AKE_CQT_ENGINE")) want = std::atoi(e);
        const bool can_fuse = p->half_len <= 23;
        const bool can_bf16 = can_fuse && n_oct >= 2 && n_oct - 1 <= kCascMax;
        if (want == 3 && !can_bf16) { ake::set_error("cqt: engine 3 needs 2..%d octaves and decim_half_len <= 23", kCascMax + 1); ake_cqt_plan_destroy(p); return AKE_ERR_UNSUPPORTED; }
        if (want == 2 && !can_fuse) { ake::set_error("cqt: engine 2 needs decim_half_len <= 23"); ake_cqt_plan_destroy(p); return AKE_ERR_UNSUPPORTED; }
        // engine 5 (streaming MFMA cascade of levels 0..4 + engine 3's cascade for the deeper levels + engine 3's bank)
        const bool can_sm = p->half_len == 23 && n_oct >= sm::kStages + 2 && n_oct <= sm::kStages + kCascMax + 1;
        if (want == 5 && !can_sm) {
            ake::set_error("cqt: engine 5 needs decim_half_len 23 and %d-%d octaves", sm::kStages + 2, sm::kStages + kCascMax + 1);
            ake_cqt_plan_destroy(p);
            return AKE_ERR_UNSUPPORTED;
        }
        if (want == 4) { ake::set_error("cqt: engine 4 (round 2's fused LDS-ring kernel) was removed in round 3: it never beat engine 3; see engine 5"); ake_cqt_plan_destroy(p); return AKE_ERR_UNSUPPORTED; }
        p->engine = (want >= 1 && want <= 5) ? want : (can_bf16 ? 3 : (can_fuse ? 2 : 1));
        p->cfg.engine = p->engine;
    }
    if (p->engine == 5) {
        std::vector<uint16_t> t5;
        sm::build_toeplitz(p->taps.hodd, p->taps.n_odd, t5);
        hipError_t e5 = hipMalloc(&p->toep5_dev, t5.size() * sizeof(uint16_t));
        if (e5 == hipSuccess) e5 = hipMemcpy(p->toep5_dev, t5.data(), t5.size() * sizeof(uint16_t), hipMemcpyHostToDevice);
        if (e5 != hipSuccess) {
            ake::set_error("cqt plan (engine 5): Toeplitz upload failed: %s", hipGetErrorString(e5));
            ake_cqt_plan_destroy(p);
            return AKE_ERR_HIP;
        }
    }
    if (p->engine == 3 || p->engine == 5) {
        auto bf16_rne = [](float v) -> uint16_t {
            uint32_t u;
            std::memcpy(&u, &v, 4);
            u += 0x7FFFu + ((u >> 16) & 1u);
            return static_cast<uint16_t>(u >> 16);
        };
        auto bf16_f32 = [](uint16_t h) { uint32_t u = static_cast<uint32_t>(h) << 16; float f; std::memcpy(&f, &u, 4); return f; };
        std::vector<uint16_t> t2;
        int ppad = 0;
        for (int o = 0; o < n_oct; ++o) {
            const OctDesc& g1 = p->octs[o];
            const int dec = 1 << o;
            const int sh = std::min(p->hop_twos, o);
            const int nph = dec >> sh;
            OctDesc2 g;
            std::memset(&g, 0, sizeof(g));
            g.k0 = g1.k0; g.n_bins = g1.n_bins; g.n_tiles = g1.n_tiles; g.uh = g1.uh;
            g.n_blk = (2 * g.uh + 1 + 31) / 32;
            auto uh_of = [&](int k) { return static_cast<int>(std::ceil(-std::floor(-len[k] / 2.0) / dec)) + 1; };
            for (int j = 0; j < g.n_tiles; ++j) {
                const int uh_j = uh_of(g.k0 + kTileBins * j);
                g.blk_lo[j] = (g.uh - uh_j) / 32;
                g.blk_hi[j] = std::min(g.n_blk - 1, (g.uh + uh_j) / 32);
            }
            ppad = std::max(ppad, std::max(g.uh, 32 * g.n_blk - g.uh));
            const size_t per_phase = static_cast<size_t>(g.n_blk) * kMaxTiles * 2 * 64 * 8;      // uint16 entries
            g.phase_stride = static_cast<long long>(per_phase / 8);
            g.table_off = static_cast<long long>(t2.size() / 8);
            t2.resize(t2.size() + static_cast<size_t>(nph) * per_phase, 0);
            for (int pi = 0; pi < nph; ++pi) {
                const double ph = static_cast<double>(pi << sh);
                uint16_t* dst = t2.data() + static_cast<size_t>(g.table_off) * 8 + static_cast<size_t>(pi) * per_phase;
                for (int b = 0; b < bpo; ++b) {
                    const int k = g.k0 + b;
                    const int j = b / kTileBins;
                    const double lo = std::floor(-len[k] / 2.0);
                    const double L = std::floor(len[k] / 2.0) - lo;
                    const double scale = dec * std::sqrt(len[k]) / (L / 2.0) / cascade_gain(freq[k], o);
                    for (int tap = 0; tap < 32 * g.n_blk; ++tap) {
                        const int u = tap - g.uh;
                        const double pos = static_cast<double>(dec) * u - ph;
                        if (!(pos >= lo && pos <= lo + L)) continue;
                        const double win = 0.5 - 0.5 * std::cos(2.0 * M_PI * (pos - lo) / L);
                        const double arg = 2.0 * M_PI * freq[k] * pos / sr;
                        const int blk = tap / 32, qq = (tap % 32) / 8, e = tap % 8;      // MFMA k index 8*qq + e
                        const float vals[2] = {static_cast<float>(scale * win * std::cos(arg)), static_cast<float>(-scale * win * std::sin(arg))};
                        for (int ri = 0; ri < 2; ++ri) {
                            const int lane_i = qq * 16 + 2 * (b % kTileBins) + ri;
                            const uint16_t hi = bf16_rne(vals[ri]);
                            const uint16_t lo16 = bf16_rne(vals[ri] - bf16_f32(hi));
                            const size_t f = ((static_cast<size_t>(blk) * kMaxTiles + j) * 2) * 64;
                            dst[(f + lane_i) * 8 + e] = hi;
                            dst[(f + 64 + lane_i) * 8 + e] = lo16;
                        }
                    }
                }
            }
            p->octs2.push_back(g);
            p->bank2_lds = std::max(p->bank2_lds, static_cast<size_t>(std::min(g.n_blk, kBank2Chunk)) * kMaxTiles * 2 * 64 * 16);
        }
        p->ppad = (ppad + 8 + 7) / 8 * 8;
        const char* what = "hipMalloc(table)";
        hipError_t e2 = hipMalloc(&p->table2_dev, t2.size() * sizeof(uint16_t));
        if (e2 == hipSuccess) { what = "hipMalloc(octs)"; e2 = hipMalloc(&p->octs2_dev, p->octs2.size() * sizeof(OctDesc2)); }
        if (e2 == hipSuccess) { what = "hipMemcpy(table)"; e2 = hipMemcpy(p->table2_dev, t2.data(), t2.size() * sizeof(uint16_t), hipMemcpyHostToDevice); }
        if (e2 == hipSuccess) { what = "hipMemcpy(octs)"; e2 = hipMemcpy(p->octs2_dev, p->octs2.data(), p->octs2.size() * sizeof(OctDesc2), hipMemcpyHostToDevice); }
        if (e2 == hipSuccess) {
            what = "hipFuncSetAttribute(max dynamic LDS)";
            e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(cqt_bank_bf16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     static_cast<int>(p->bank2_lds));
        }
        if (e2 != hipSuccess) {
            ake::set_error("cqt plan (bf16 tables, %zu B of LDS per workgroup): %s failed: %s", p->bank2_lds, what, hipGetErrorString(e2));
            ake_cqt_plan_destroy(p);
            return AKE_ERR_HIP;
        }
    }
    *out = p;
    return AKE_OK;
}

void ake_cqt_plan_destroy(ake_cqt_plan* p) {
    if (!p) return;
    if (p->table_dev) (void)hipFree(p->table_dev);
    if (p->octs_dev) (void)hipFree(p->octs_dev);
    if (p->table2_dev) (void)hipFree(p->table2_dev);
    if (p->octs2_dev) (void)hipFree(p->octs2_dev);
    if (p->toep5_dev) (void)hipFree(p->toep5_dev);
    delete p;
}

int ake_cqt_plan_n_bins(const ake_cqt_plan* p) { return p ? p->cfg.n_bins : 0; }

int ake_cqt_plan_hop(const ake_cqt_plan* p) { return p ? p->cfg.hop_length : 0; }

int64_t ake_cqt_num_frames(const ake_cqt_plan* p, int64_t n_samples) {
    if (!p || n_samples < 0) return -1;
    return 1 + n_samples / p->cfg.hop_length;
}

size_t ake_cqt_workspace_bytes(const ake_cqt_plan* p, int batch, int64_t n_samples) {
    if (!p || batch <= 0 || n_samples <= 0) return 0;
    ake::Carver c(nullptr, 0);
    if (p->engine == 5) {
        for (int l = 0; l < sm::kStages; ++l) c.take<unsigned int>(static_cast<size_t>(batch) * plane_len(p, l, n_samples));   // split planes of levels 0..3
        c.take<float>(static_cast<size_t>(batch) * next_len(n_samples));                                                        // level 4, f32
        for (int l = 0; l + sm::kStages < p->n_oct; ++l)                                                                        // split planes of levels 4..
            c.take<unsigned int>(static_cast<size_t>(batch) * plane_len(p, l, next_len(n_samples)));
    } else if (p->engine == 3) {
        for (int l = 0; l < p->n_oct; ++l) c.take<unsigned int>(static_cast<size_t>(batch) * plane_len(p, l, n_samples));   // split-bf16 level signals
    } else {
        for (int o = 1; o < p->n_oct; ++o) c.take<float>(static_cast<size_t>(batch) * len_store(p, o, n_samples));
    }
    c.take<float>(static_cast<size_t>(batch) * (1 + n_samples / p->cfg.hop_length) * p->cfg.n_bins);   // [clip][frame][bin] scratch
    return ake::align_up(c.off, 256);
}

}  // extern "C"

namespace {
int cqt_logmag_impl(const ake_cqt_plan* p, const float* audio, int batch, int64_t n, int64_t audio_stride, const int64_t* n_clip,
                    float* out, int64_t out_frames, void* workspace, size_t ws_bytes, ake_stream_t stream_, bool frames_major = false);
}

extern "C" {

int ake_cqt_logmag_f32(const ake_cqt_plan* p, const float* audio, int batch, int64_t n, int64_t audio_stride,
                       float* out, int64_t out_frames, void* workspace, size_t ws_bytes, ake_stream_t stream_) {
    return cqt_logmag_impl(p, audio, batch, n, audio_stride, nullptr, out, out_frames, workspace, ws_bytes, stream_);
}

int ake_cqt_frames_major_supported(const ake_cqt_plan* p) { return p && (p->engine == 3 || p->engine == 5) ? 1 : 0; }

int ake_cqt_logmag_frames_major_f32(const ake_cqt_plan* p, const float* audio, int batch, int64_t n, int64_t audio_stride, float* out,
                                    void* workspace, size_t ws_bytes, ake_stream_t stream_) {
    AKE_REQUIRE(p, AKE_ERR_INVALID, "ake_cqt_logmag_frames_major_f32: null plan");
    return cqt_logmag_impl(p, audio, batch, n, audio_stride, nullptr, out, ake_cqt_num_frames(p, n), workspace, ws_bytes, stream_, true);
}

int ake_cqt_logmag_ragged_f32(const ake_cqt_plan* p, const float* audio, int batch, int64_t n_max, int64_t audio_stride,
                              const int64_t* n_samples_dev, float* out, int64_t out_frames, void* workspace, size_t ws_bytes,
                              ake_stream_t stream_) {
    AKE_REQUIRE(n_samples_dev, AKE_ERR_INVALID, "ake_cqt_logmag_ragged_f32: null n_samples_dev");
    AKE_REQUIRE(p && (p->engine == 3 || p->engine == 5), AKE_ERR_UNSUPPORTED, "cqt: ragged batches need engine 3 or 5 (the defaults up to 8 octaves)");
    return cqt_logmag_impl(p, audio, batch, n_max, audio_stride, n_samples_dev, out, out_frames, workspace, ws_bytes, stream_);
}

}  // extern "C"

namespace {

int cqt_logmag_impl(const ake_cqt_plan* p, const float* audio, int batch, int64_t n, int64_t audio_stride, const int64_t* n_clip,
                    float* out, int64_t out_frames, void* workspace, size_t ws_bytes, ake_stream_t stream_, bool frames_major) {
    AKE_REQUIRE(p && audio && out, AKE_ERR_INVALID, "ake_cqt_logmag_f32: null argument");
    // frames_major: leave the result as the filter bank writes it, [clip][frame][bin] (no transpose pass); engine 3, equal-length clips
    AKE_REQUIRE(!frames_major || ((p->engine == 3 || p->engine == 5) && !n_clip), AKE_ERR_UNSUPPORTED, "cqt: the frames-major output needs engine 3 or 5 and equal-length clips");
    AKE_REQUIRE(batch > 0 && n > 0 && audio_stride >= n, AKE_ERR_INVALID, "cqt: bad batch/n_samples/stride");
    AKE_REQUIRE(n < (1ll << 30), AKE_ERR_INVALID, "cqt: clip too long (%lld samples)", static_cast<long long>(n));
    const int64_t T = ake_cqt_num_frames(p, n);
    AKE_REQUIRE(out_frames >= T, AKE_ERR_INVALID, "cqt: out_frames %lld < %lld frames", static_cast<long long>(out_frames), static_cast<long long>(T));
    AKE_REQUIRE(ws_bytes >= ake_cqt_workspace_bytes(p, batch, n) && (workspace || p->n_oct == 1), AKE_ERR_WORKSPACE, "cqt: workspace too small");
    hipStream_t stream = static_cast<hipStream_t>(stream_);

    ake::Carver c(workspace, ws_bytes);
    float* scratch = nullptr;
    constexpr int C = 4096, NT = 256;
    auto fill_cascade_on = [&](CascArgs& a, int fused, const float* x_in, int64_t x_stride, int64_t n_in, const int64_t* n_clip_in) {
        std::memset(&a, 0, sizeof(a));
        a.x = x_in; a.x_stride = x_stride; a.n = static_cast<int>(n_in);
        a.n_clip = reinterpret_cast<const long long*>(n_clip_in);
        a.pad = pad_of(p); a.hop = p->cfg.hop_length; a.n_stage = fused; a.taps = p->taps;
        for (int l = 1; l <= fused; ++l) a.y_count[l] = len_store(p, l, n_in);
        a.g0 = -512;
        const long long g1 = n_in + static_cast<long long>(25 + kLagHost[fused]) * (1ll << fused) + 512;
        a.ticks_total = static_cast<int>((g1 - a.g0 + C - 1) / C);
        static const int segs_env = ake::diag_env("AKE_CQT_SEGS") ? std::atoi(ake::diag_env("AKE_CQT_SEGS")) : 0;
        const int segs = std::max(1, std::min(a.ticks_total, segs_env > 0 ? segs_env : 4));
        a.ticks_per_seg = (a.ticks_total + segs - 1) / segs;
        a.warm = 3;                                         // >= 64 * 2^7 / C ticks of history before the first owned tick
    };
    auto fill_cascade = [&](CascArgs& a, int fused) { fill_cascade_on(a, fused, audio, audio_stride, n, n_clip); };
    auto launch_cascade = [&](const CascArgs& a) {
        dim3 grid((a.ticks_total + a.ticks_per_seg - 1) / a.ticks_per_seg, batch);
        ake::ProfScope ps("cqt_cascade_kernel", stream);
        const bool split = a.ph[0] != nullptr;
        if (p->half_len == 15 && split) hipLaunchKernelGGL((cqt_cascade_kernel<8, C, NT, true>), grid, dim3(NT), 0, stream, a);
        else if (p->half_len == 15) hipLaunchKernelGGL((cqt_cascade_kernel<8, C, NT, false>), grid, dim3(NT), 0, stream, a);
        else if (split) {
            static const bool stamp_env = ake::diag_env("AKE_CQT_CASC_STAMP") != nullptr;
            unsigned long long* sb = nullptr;
            if (stamp_env && a.n_stage == 7 && hipMalloc(&sb, 64 * sizeof(unsigned long long)) == hipSuccess) {
                CascArgs a2 = a;
                a2.stamps = sb;
                (void)hipMemsetAsync(sb, 0, 64 * sizeof(unsigned long long), stream);
                hipLaunchKernelGGL((cqt_cascade_kernel<12, C, NT, true, true>), grid, dim3(NT), 0, stream, a2);
                unsigned long long hb[64];
                (void)hipMemcpyAsync(hb, sb, sizeof(hb), hipMemcpyDeviceToHost, stream);
                (void)hipStreamSynchronize(stream);
                (void)hipFree(sb);
                for (int wv = 0; wv < 4; ++wv) {
                    fprintf(stderr, "cascade stamps wave %d: ticks %llu  cycles/tick: level-0 write+barrier %.0f", wv, hb[wv * 16 + 8], hb[wv * 16] / double(hb[wv * 16 + 8]));
                    for (int i = 1; i < 8; ++i) fprintf(stderr, "  stage %d: %.0f", i - 1, hb[wv * 16 + i] / double(hb[wv * 16 + 8]));
                    fprintf(stderr, "  (audio prefetch still outstanding at tick start: %.0f)\n", hb[wv * 16 + 9] / double(hb[wv * 16 + 8]));
                }
            } else hipLaunchKernelGGL((cqt_cascade_kernel<12, C, NT, true>), grid, dim3(NT), 0, stream, a);
        }
        else hipLaunchKernelGGL((cqt_cascade_kernel<12, C, NT, false>), grid, dim3(NT), 0, stream, a);
    };
    if (p->engine == 5) {
        // levels 0..4: cqt_stream_kernel (MFMA half-band stages in registers; split planes of levels 0..3 near the frame centres + level 4 as f32)
        sm::Args sa;
        std::memset(&sa, 0, sizeof(sa));
        BankCall2 call2;
        std::memset(&call2, 0, sizeof(call2));
        call2.pad = p->ppad;
        sa.x = audio; sa.x_stride = audio_stride; sa.n = static_cast<int>(n); sa.n_clip = reinterpret_cast<const long long*>(n_clip); sa.batch = batch;
        sa.ppad = p->ppad; sa.hop = p->cfg.hop_length; sa.toep = p->toep5_dev; sa.h0 = p->taps.h0;
        if (const char* e = ake::diag_env("AKE_SM_ABLATE")) sa.dbg = std::atoi(e);
        for (int l = 0; l < sm::kStages; ++l) {
            const int pl = plane_len(p, l, n);
            sa.ph[l] = c.take<unsigned int>(static_cast<size_t>(batch) * pl);
            sa.p_stride[l] = pl; sa.p_count[l] = pl;
            call2.xw[l] = sa.ph[l]; call2.stride[l] = pl;
            const OctDesc2& g = p->octs2[l];
            const long long need = static_cast<long long>(std::max(g.uh, 32 * g.n_blk - g.uh) + 8) << l;      // taps [c - uh, c - uh + 32 n_blk) around a centre c
            sa.need[l] = 2 * need >= sa.hop ? -1 : static_cast<int>(need);
        }
        const int n_next = next_len(n);
        float* next = c.take<float>(static_cast<size_t>(batch) * n_next);
        sa.next = next; sa.next_stride = n_next; sa.next_count = n_next; sa.pad_next = kNextPad;
        // (everything in periods of 8 chunks: the kernel's stage parities are compile-time)
        sa.c_begin = -static_cast<int>((((static_cast<long long>(p->ppad) + 64) * 8 + 63) / 64 + 7) / 8 * 8);  // level 3's left pad, as silence
        sa.c_end = static_cast<int>(((n + 2048) / 64 + 9 + 7) / 8 * 8);                                        // ... and every level's tail + the pipeline's delay
        static const int segs_env = ake::diag_env("AKE_CQT_SEGS") ? std::atoi(ake::diag_env("AKE_CQT_SEGS")) : 0;
        const int n_groups = (batch + 15) / 16;
        int segs = segs_env > 0 ? segs_env : std::max(1, (4 * p->n_cu + n_groups - 1) / n_groups);            // one wave per SIMD
        sa.warm = 24;
        const int total = sa.c_end - sa.c_begin;
        segs = std::max(1, std::min(segs, total / (2 * sa.warm)));                                             // (short clips: warm-up must not dominate)
        sa.chunks_per_seg = ((total + segs - 1) / segs + 7) / 8 * 8;
        {
            dim3 grid((total + sa.chunks_per_seg - 1) / sa.chunks_per_seg, n_groups);
            ake::ProfScope ps("cqt_stream_kernel", stream);
            hipLaunchKernelGGL(sm::cqt_stream_kernel, grid, dim3(64), 0, stream, sa);
        }
        // levels 5..: engine 3's cascade on level 4 (as its "audio": index i <-> level-4 sample i - kNextPad, so its level l holds true
        // level 4 + l shifted by kNextPad >> l samples), which also writes level 4's own split plane
        const int S = p->n_oct - sm::kStages - 1;
        CascArgs ca;
        fill_cascade_on(ca, S, next, n_next, n_next, nullptr);
        ca.ppad = p->ppad;
        for (int l = 0; l <= S; ++l) {
            const int pl = plane_len(p, l, n_next);
            ca.ph[l] = c.take<unsigned int>(static_cast<size_t>(batch) * pl);
            ca.p_stride[l] = pl; ca.p_count[l] = pl;
            ca.need[l] = -1;                                               // every sample: the tap windows of these octaves overlap
            call2.xw[sm::kStages + l] = ca.ph[l] + (kNextPad >> l);
            call2.stride[sm::kStages + l] = pl;
        }
        launch_cascade(ca);
        scratch = frames_major ? out : c.take<float>(static_cast<size_t>(batch) * T * p->cfg.n_bins);
        dim3 grid(static_cast<unsigned>((T + 7) / 8 * 8), p->n_oct, (batch + kBankClips - 1) / kBankClips);
        ake::ProfScope ps("cqt_bank_bf16_kernel", stream);
        hipLaunchKernelGGL(cqt_bank_bf16_kernel, grid, dim3(kBankThreads), p->bank2_lds, stream, call2, p->octs2_dev, p->table2_dev, batch,
                           p->cfg.hop_length, p->hop_twos, scratch, static_cast<long long>(T) * p->cfg.n_bins, p->cfg.n_bins, static_cast<int>(T), 0);
    } else if (p->engine == 3) {
        BankCall2 call2;
        std::memset(&call2, 0, sizeof(call2));
        CascArgs a;
        // (Round 3 tried the cascade as two launches -- stages 0..3 streaming, stages 4..6 in a small second launch reading level 4's split
        // plane, because in-kernel stamps put half of every tick in the deep stages: the first launch went 0.155 -> 0.128 ms per 256 clips, the
        // second took 0.025 ms whatever its segment count -- a latency chain over 6 ticks -- so the pair gained nothing and was removed.)
        fill_cascade(a, p->n_oct - 1);
        a.ppad = p->ppad;
        call2.pad = p->ppad;
        for (int l = 0; l < p->n_oct; ++l) {
            const int pl = plane_len(p, l, n);
            a.ph[l] = c.take<unsigned int>(static_cast<size_t>(batch) * pl);
            a.p_stride[l] = pl; a.p_count[l] = pl;
            call2.xw[l] = a.ph[l]; call2.stride[l] = pl;
            // taps [c - uh, c - uh + 32 * n_blk) of level l around every frame centre c
            const OctDesc2& g = p->octs2[l];
            const long long need = static_cast<long long>(std::max(g.uh, 32 * g.n_blk - g.uh) + 8) << l;
            a.need[l] = 2 * need >= a.hop ? -1 : static_cast<int>(need);
        }
        launch_cascade(a);
        scratch = frames_major ? out : c.take<float>(static_cast<size_t>(batch) * T * p->cfg.n_bins);
        dim3 grid(static_cast<unsigned>((T + 7) / 8 * 8), p->n_oct, (batch + kBankClips - 1) / kBankClips);
        ake::ProfScope ps("cqt_bank_bf16_kernel", stream);
        hipLaunchKernelGGL(cqt_bank_bf16_kernel, grid, dim3(kBankThreads), p->bank2_lds, stream, call2, p->octs2_dev, p->table2_dev, batch,
                           p->cfg.hop_length, p->hop_twos, scratch, static_cast<long long>(T) * p->cfg.n_bins, p->cfg.n_bins, static_cast<int>(T), 0);
    } else {
        BankCall call;
        std::memset(&call, 0, sizeof(call));
        call.x[0] = audio;
        call.stride[0] = audio_stride;
        call.lo[0] = 0;
        call.count[0] = static_cast<int>(n);
        for (int o = 1; o < p->n_oct; ++o) {
            const int ls = len_store(p, o, n);
            call.x[o] = c.take<float>(static_cast<size_t>(batch) * ls);
            call.stride[o] = ls;
            call.lo[o] = -pad_of(p);
            call.count[o] = ls;
        }
        const int fused = p->engine == 2 ? std::min(p->n_oct - 1, kCascMax) : 0;
        if (fused > 0) {
            CascArgs a;
            fill_cascade(a, fused);
            for (int l = 1; l <= fused; ++l) {
                a.y[l] = const_cast<float*>(call.x[l]);
                a.y_stride[l] = call.stride[l];
                // the bank reads, per frame, taps [c - uh, c - uh + 16 * n_blocks) of level l around the frame centre c; a later
                // per-stage kernel (more than kCascMax stages) needs its input level everywhere
                const long long need = static_cast<long long>(p->octs[l].uh + 24) << l;
                a.need[l] = (2 * need >= a.hop || (l == fused && fused < p->n_oct - 1)) ? -1 : static_cast<int>(need);
            }
            launch_cascade(a);
        }
        for (int o = fused + 1; o < p->n_oct; ++o) {
            const int ls = call.count[o];
            float* y = const_cast<float*>(call.x[o]);
            dim3 grid((ls + kDecimOutPerBlock - 1) / kDecimOutPerBlock, batch);
            ake::ProfScope ps("cqt_decimate_kernel", stream);
            const int in_pad = -call.lo[o - 1];
#define AKE_DECIM(N_) hipLaunchKernelGGL((cqt_decimate_kernel<N_>), grid, dim3(kDecimThreads), 0, stream, call.x[o - 1], \
                                         call.stride[o - 1], in_pad, call.count[o - 1], y, static_cast<long long>(ls), pad_of(p), ls, p->taps)
            if (p->half_len == 15) AKE_DECIM(8);
            else if (p->half_len == 23) AKE_DECIM(12);
            else AKE_DECIM(16);
#undef AKE_DECIM
        }
        scratch = c.take<float>(static_cast<size_t>(batch) * T * p->cfg.n_bins);
        dim3 grid(static_cast<unsigned>(T), p->n_oct, (batch + 63) / 64);
        ake::ProfScope ps("cqt_bank_kernel", stream);
        hipLaunchKernelGGL(cqt_bank_kernel, grid, dim3(256), 0, stream, call, p->octs_dev, p->table_dev, batch,
                           p->cfg.hop_length, p->hop_twos, scratch, static_cast<long long>(T) * p->cfg.n_bins, p->cfg.n_bins);
    }
    if (frames_major) {
        AKE_HIP_CHECK(hipGetLastError());
        return AKE_OK;
    }
    {
        dim3 grid((p->cfg.n_bins + 31) / 32, static_cast<unsigned>((out_frames + 31) / 32), batch);
        ake::ProfScope ps("cqt_transpose_kernel", stream);
        hipLaunchKernelGGL(cqt_transpose_kernel, grid, dim3(256), 0, stream, scratch, out, static_cast<int>(T), p->cfg.n_bins,
                           static_cast<int>(out_frames), reinterpret_cast<const long long*>(n_clip), p->cfg.hop_length);
    }
    AKE_HIP_CHECK(hipGetLastError());
    return AKE_OK;
}

}  // namespace

// CQT engine 4: ONE streaming pass per four octaves -- the half-band decimator cascade AND the per-phase filter bank both run on
// v_mfma_f32_16x16x32_bf16 with split operands, and the level signals never leave the LDS.
//
// Why: engines 2/3 write every level signal to HBM where a frame's tap window reads it and read it back in a second
// kernel (132 MB written + 202 MB read per 256 clips next to the 339 MB of audio), and their cascade is a VALU FIR with 8
// workgroup barriers per 4096 samples.  Here a workgroup owns 16 clips x one time segment and walks it in steps of 128 samples:
//
//   phase A  every thread takes 4 audio samples from the LDS staging ring (filled kStage steps ahead by LDS-DMA: no registers, no
//            unrolling, ~40 KB in flight per CU), splits them into (bf16 hi, bf16 lo) and writes them to the level-0 ring
//   barrier  (the only one per step)
//   phase B  wave w computes ONE 16-outputs x 16-clips tile of the cascade -- waves 0-3 level 0->1, 4-5 level 1->2, 6 level 2->3,
//            7 level 3->4 (every other step) -- as D[output n][clip] = Toeplitz(h)[n][k] . X[k][clip]: the filter taps are 6 constant
//            A fragments (3 k-steps x hi/lo), the B operand is three aligned 16-byte LDS reads of the input ring, 9 MFMAs per tile;
//            each stage reads what the stage above wrote in the PREVIOUS step, so one barrier per step is enough.
//            When a frame's tap window of some level is complete, the waves also run that (frame, octave)'s filter bank straight
//            from the ring: D[bin re/im][clip] = W[phase][col][tap] . X[tap][clip], 5 N-tiles dealt to the waves, table fragments
//            read from L2 (the 16 clip groups of a segment run on one XCD and read the same tables at the same time).
//
// Rings are laid out [sample / 8][clip][8 samples] (hi plane, lo plane): a B operand (lane = clip + 16 q reads samples 8q..8q+7) is
// 1 KiB of consecutive bytes per wave -- conflict-free without padding.  Windows are anchored on multiples of 8 samples so that
// those reads stay 16-byte aligned; the anchor's offset from the frame centre joins the fractional phase: one filter bank per
// (t * hop - uh * 2^o) mod (8 * 2^o).
//
// Octaves beyond the fourth: the kernel emits level 4 (f32, 1/16 of the audio) and is launched a second time on it.
// HBM traffic: audio once (+ halo re-reads at segment seams), level 4 once out and in, the log-magnitudes out.
#pragma once

namespace fz {

constexpr int kClips = 16;     // clips per workgroup = MFMA N
constexpr int kStep = 128;     // input samples per step
constexpr int kNT = 512;       // threads (8 waves)
constexpr int kMaxLv = 4;      // octaves per launch
constexpr int kStage = 5;      // audio tiles (8 KB each) in flight per workgroup, LDS-DMA into a staging ring
constexpr int kMaxBlk = 10;    // 32-tap blocks of a window (LDS bounds it: see rings())

struct Level {
    int uh, n_blk, n_tiles, k0, n_bins, period;
    int blk_lo[kMaxTiles], blk_hi[kMaxTiles];
    long long table_off, phase_stride;    // 16-byte units
    int ring_units;                        // ring length in units of 8 samples
    int lds_off;                           // uint4 index of the hi plane; the lo plane follows at + ring_units * 16
};

struct Args {
    const float* x;            // [batch][x_stride]: index i <-> level-L0 sample number i - pad_in
    long long x_stride;
    unsigned x_bytes;          // whole tensor as one range-checked buffer
    long long n_valid;         // indices [0, n_valid) hold samples, everything else reads as zero
    const long long* n_clip;   // ragged batches: full-rate samples of each clip (or null)
    int pad_in, L0, hop, batch, T;
    float* next;               // emitted level L0 + 4 (f32): sample m at next[clip * next_stride + m + pad_next]
    long long next_stride;
    int next_count, pad_next;
    float* out;                // [clip][T][n_bins_total] log-magnitudes
    long long out_clip_stride;
    int n_bins_total;
    long long M_begin;         // level-L0 sample number where segment 0's owned range starts (multiple of 128)
    int seg_len, n_seg, h_pre, h_post;
    Level lv[kMaxLv];
    const uint4* table;
    const uint4* toep;         // half-band Toeplitz fragments [3 k-steps][hi | lo][64 lanes]
    int dbg;                   // timing experiments only (AKE_CQT_FZ_DBG): 1 no filter banks, 2 no cascade tiles, 4 no ring-0 writes, 8 no table touch
};

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// 4 consecutive samples -> (4 x bf16 hi, 4 x bf16 lo): hi = RNE(v), lo = RNE(v - hi); v - (hi + lo) <= 2^-17 |v|
__device__ __forceinline__ void split4(float r0, float r1, float r2, float r3, uint2& hi, uint2& lo) {
    const f32x2 a = {r0, r1}, b = {r2, r3};
    const bf16x2 ha = __builtin_convertvector(a, bf16x2), hb = __builtin_convertvector(b, bf16x2);
    const f32x2 la = a - __builtin_convertvector(ha, f32x2), lb = b - __builtin_convertvector(hb, f32x2);
    const bf16x2 qa = __builtin_convertvector(la, bf16x2), qb = __builtin_convertvector(lb, bf16x2);
    hi = make_uint2(__builtin_bit_cast(unsigned int, ha), __builtin_bit_cast(unsigned int, hb));
    lo = make_uint2(__builtin_bit_cast(unsigned int, qa), __builtin_bit_cast(unsigned int, qb));
}

__device__ __forceinline__ f32x4 mfma3(const uint4& ah, const uint4& al, const uint4& bh, const uint4& bl, f32x4 acc) {
    const bf16x8 xah = __builtin_bit_cast(bf16x8, ah), xal = __builtin_bit_cast(bf16x8, al);
    const bf16x8 xbh = __builtin_bit_cast(bf16x8, bh), xbl = __builtin_bit_cast(bf16x8, bl);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xah, xbh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xah, xbl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xal, xbh, acc, 0, 0, 0);
    return acc;
}

// Stores the compiler must not count: hipcc waits (s_waitcnt vmcnt) before it reuses the registers of a pending store, and on the
// step loop's common path such a wait drains the LDS-DMA queue it knows nothing about.  (s_nop 1: the store reads its data registers
// after issue, cdna_hip_programming.md section 5.7.)
__device__ __forceinline__ void store_f32x4(float* p, f32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void store_f32(float* p, float v) {
    asm volatile("global_store_dword %0, %1, off\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
}

__device__ __forceinline__ int wrap_once(int u, int n) { return u >= n ? u - n : u; }
__device__ __forceinline__ int wrap_any(long long u, int n) { int r = static_cast<int>(u % n); return r < 0 ? r + n : r; }

template <int NL, bool EMIT>
__global__ __launch_bounds__(kNT) void cqt_fused_kernel(Args a) {
    extern __shared__ __attribute__((aligned(16))) uint4 lds[];
    constexpr int NS = NL - 1 + (EMIT ? 1 : 0);                       // cascade stages in this launch
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c16 = lane & 15, q = lane >> 4;
    // workgroup -> (segment, clip group): the clip groups of one segment are neighbours within an XCD (blockIdx % 8): they stream
    // the same filter tables at the same time, so all but one of them hit that XCD's L2
    const int n_groups = (a.batch + kClips - 1) / kClips;
    const int n_wg = n_groups * a.n_seg;
    const int per_xcd = gridDim.x >> 3;
    const int w = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (w >= n_wg) return;
    const int seg = w / n_groups, grp = w - seg * n_groups;
    const long long Ms = a.M_begin + static_cast<long long>(seg) * a.seg_len, Me = Ms + a.seg_len;
    const long long M0 = Ms - static_cast<long long>(a.h_pre) * kStep;           // sample number of relative index 0
    const int n_steps = a.seg_len / kStep + a.h_pre + a.h_post;

    int lds_total = 0;
#pragma unroll
    for (int l = 0; l < NL; ++l) lds_total = a.lv[l].lds_off + 2 * a.lv[l].ring_units * 16;
    for (int i = tid; i < lds_total; i += kNT) lds[i] = make_uint4(0, 0, 0, 0);

    uint4 th[3], tl[3];                                               // half-band Toeplitz A fragments
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) { th[ks] = a.toep[(ks * 2 + 0) * 64 + lane]; tl[ks] = a.toep[(ks * 2 + 1) * 64 + lane]; }

    // ---- audio loader.  Thread -> (clip (wave & 1) * 8 + (lane & 7), samples 4 * ((wave >> 1) * 8 + (lane >> 3)) .. + 3 of the step's
    // 128): a wave instruction fetches 8 clips x 128 contiguous bytes, and the ring-0 stores of a 16-lane group (8 clips x 2 halves
    // of one 8-sample unit) fall on 32 different banks.  The tile goes global -> LDS staging slot by LDS-DMA (asm: hipcc would order
    // every later LDS access behind a builtin DMA with vmcnt(0)); each thread later reads back exactly the 16 bytes its own lane
    // fetched, so the only ordering needed is this wave's own counted vmcnt. ----
    const int ld_c = (wave & 1) * 8 + (lane & 7), ld_s = (wave >> 1) * 8 + (lane >> 3);
    const int ld_clip = grp * kClips + ld_c < a.batch ? grp * kClips + ld_c : a.batch - 1;
    long long nv = a.n_valid;
    if (a.n_clip && a.L0 == 0) { const long long nc = a.n_clip[ld_clip]; nv = nc < 0 ? 0 : (nc < nv ? nc : nv); }
    const long long ld_row = static_cast<long long>(ld_clip) * a.x_stride;
    const long long ld_i0 = M0 + a.pad_in + 4 * ld_s;                 // array index of this thread's samples in step 0
    const long long ld_max = static_cast<long long>(a.x_bytes / 4) - 4;   // last float4 inside the tensor
    uint4* const stage = lds + lds_total;                             // [kStage][512] float4, lane-linear per wave
    auto dma = [&](int k) {                                           // tile k -> slot k % kStage (out-of-clip samples are masked at use)
        long long gi = ld_row + ld_i0 + static_cast<long long>(k) * kStep;
        if (a.dbg & 128) {                                            // timing experiment: 1 KiB of ONE clip per wave instruction
            const int cl = grp * kClips + 2 * wave + (k & 1);
            gi = static_cast<long long>(cl < a.batch ? cl : a.batch - 1) * a.x_stride + M0 + a.pad_in + static_cast<long long>(k >> 1) * 256 + 4 * lane;
        }
        gi = gi < 0 ? 0 : (gi > ld_max ? ld_max : gi);                // any in-tensor address will do where the samples are masked
        const float* src = a.x + gi;
        const unsigned int dst = static_cast<unsigned int>(reinterpret_cast<unsigned long long>(stage + (k % kStage) * kNT + wave * 64));
        unsigned int keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
    };
#pragma unroll
    for (int i = 0; i < kStage; ++i) dma(i);

    // ---- cascade role of this wave: stage s, tile tw; ring units of its input window and output tile at step 0 ----
    const int st = wave < 4 ? 0 : (wave < 6 ? 1 : (wave == 6 ? 2 : 3));
    const int tw = wave < 4 ? wave : (wave < 6 ? wave - 4 : 0);
    const bool has_stage = st < NS;
    // (kernel-argument arrays are only ever indexed with constants: a runtime index sends the whole struct through scratch memory)
    auto ring_units_of = [&](int l) { return l == 0 ? a.lv[0].ring_units : (l == 1 ? a.lv[1].ring_units : (l == 2 ? a.lv[2].ring_units : a.lv[3].ring_units)); };
    auto lds_off_of = [&](int l) { return l == 0 ? a.lv[0].lds_off : (l == 1 ? a.lv[1].lds_off : (l == 2 ? a.lv[2].lds_off : a.lv[3].lds_off)); };
    const int lds_off0 = a.lv[0].lds_off;
    int u_in = 0, u_out = 0, inc_in = 0, inc_out = 0, ru_in = 1, ru_out = 1, off_in = 0, off_out = 0;
    if (has_stage) {
        // input base (relative index / 8): 16k - 7 + 4tw | 8k - 19 + 4tw | 4k - 19 | 2k - 19;  output P / 8: 8k - 2 + 2tw | 4k - 8 + 2tw | 2k - 8
        const int in0 = st == 0 ? -7 + 4 * tw : -19 + (st == 1 ? 4 * tw : 0);
        const int out0 = st == 0 ? -2 + 2 * tw : (st == 1 ? -8 + 2 * tw : -8);
        inc_in = 16 >> st; inc_out = 8 >> st;
        ru_in = ring_units_of(st); off_in = lds_off_of(st);
        u_in = wrap_any(in0, ru_in);
        if (st + 1 < NL) { ru_out = ring_units_of(st + 1); off_out = lds_off_of(st + 1); u_out = wrap_any(out0, ru_out); }
    }
    int u_a = 0;                                                      // level-0 ring unit of relative index 128 k
    const int ru0 = a.lv[0].ring_units;

    // ---- frames this workgroup owns: centre sample number floor(t * hop / 2^L0) in [Ms, Me) ----
    auto tb = [&](long long X) -> long long { return X <= 0 ? 0 : ((X << a.L0) + a.hop - 1) / a.hop; };
    const int t_first = static_cast<int>(tb(Ms));
    const int t_end = static_cast<int>(tb(Me) < a.T ? tb(Me) : a.T);
    // Per level: next frame, its filter-bank phase, the step at which its tap window is complete and the step at which its table
    // lines are pulled into this XCD's L2.  The Level records are read with scalar loads through a laundered kernel-argument pointer
    // INSIDE the (rare) blocks that need them: as plain kernel arguments the compiler keeps all ~100 of their dwords live in SGPRs
    // across the step loop and spills them (measured: 1 700 SGPR spills, 70 VGPRs to scratch)
    typedef const __attribute__((address_space(4))) Level* LevelCP;
    typedef const __attribute__((address_space(4))) char* CharCP;
    auto level_ptr = [&](int l) -> LevelCP {
        LevelCP pL = (LevelCP)((CharCP)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(Args, lv)) + l;
        asm volatile("" : "+s"(pL));
        return pL;
    };
    // anchored window start of (t, level l): relative index / 8, and the first step whose frontier covers the window
    auto window_of = [&](LevelCP L, int l, int t, long long& a_unit) -> int {
        const int sh = a.L0 + l;
        const long long E = static_cast<long long>(t) * a.hop - (static_cast<long long>(L->uh) << sh);   // nominal start, full-rate position
        a_unit = (E >> (sh + 3)) - ((M0 >> l) >> 3);
        // frontier(l, k) = (128 >> l) * k - lag_l >= 8 * a_unit + W - 1
        const long long x = 8 * a_unit + 32 * L->n_blk - 1 + (l == 0 ? -127 : (l == 1 ? 17 : 65));
        return x <= 0 ? 0 : static_cast<int>((x + (127 >> l)) >> (7 - l));
    };
    int tn[NL], pi[NL], kfire[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        LevelCP L = level_ptr(l);
        tn[l] = t_first;
        pi[l] = t_first % L->period;
        long long au;
        kfire[l] = t_first < t_end ? window_of(L, l, t_first, au) : 0x7fffffff;
    }

    const int my_clip = grp * kClips + c16;                           // MFMA column of this lane
    int tc = a.T;                                                     // frames of this clip (the rest is zero padding)
    if (a.n_clip && my_clip < a.batch) { const long long nc = a.n_clip[my_clip]; const long long f = nc < 0 ? 0 : 1 + nc / a.hop; tc = f < tc ? static_cast<int>(f) : tc; }
    {   // Materialise every loop invariant that came from a (tracked) load BEFORE the step loop: hipcc waits for a pending load at its
        // first use, and a first use inside the loop becomes an s_waitcnt vmcnt(0) on EVERY step -- which drains the untracked DMA queue
        // (measured: the staging ring then holds one tile in flight instead of kStage)
        unsigned int nlo = static_cast<unsigned int>(nv), nhi = static_cast<unsigned int>(static_cast<unsigned long long>(nv) >> 32);
        asm volatile("" : "+v"(nlo), "+v"(nhi), "+v"(tc));
        nv = static_cast<long long>((static_cast<unsigned long long>(nhi) << 32) | nlo);
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            asm volatile("" : "+v"(th[ks].x), "+v"(th[ks].y), "+v"(th[ks].z), "+v"(th[ks].w));
            asm volatile("" : "+v"(tl[ks].x), "+v"(tl[ks].y), "+v"(tl[ks].z), "+v"(tl[ks].w));
        }
    }
    __syncthreads();

    for (int k = 0; k < n_steps; ++k) {
        // ================= phase A: audio tile k (staging slot k % kStage) -> level-0 ring =================
        {
            // the kStage - 1 younger DMA tiles may still be in flight; anything issued after them only makes this wait longer
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(kStage - 1) : "memory");
            const f32x4 v = __builtin_bit_cast(f32x4, stage[(k % kStage) * kNT + tid]);
            const long long gi = ld_i0 + static_cast<long long>(k) * kStep;
            const long long left = gi < 0 ? 0 : nv - gi;              // valid samples from gi on (gi is a multiple of 4)
            // the last vector of the whole tensor may have been fetched from up to 3 floats earlier (dma() keeps every address inside it)
            const long long over = ld_row + gi - ld_max;
            const int sh = over > 0 ? static_cast<int>(over) : 0;
            const float e0 = sh == 0 ? v[0] : (sh == 1 ? v[1] : (sh == 2 ? v[2] : v[3]));
            const float e1 = sh == 0 ? v[1] : (sh == 1 ? v[2] : v[3]);
            const float e2 = sh == 0 ? v[2] : v[3];
            const float r0 = left > 0 ? e0 : 0.f, r1 = left > 1 ? e1 : 0.f, r2 = left > 2 ? e2 : 0.f, r3 = left > 3 ? v[3] : 0.f;
            uint2 hi, lo;
            split4(r0, r1, r2, r3, hi, lo);
            const int u = (a.dbg & 32) ? (ld_s >> 1) : wrap_once(u_a + (ld_s >> 1), ru0);
            uint2* ph = reinterpret_cast<uint2*>(lds + lds_off0);
            uint2* pl = reinterpret_cast<uint2*>(lds + lds_off0 + ru0 * 16);
            if (!(a.dbg & 4) && (!(a.dbg & 64) || wave < 4)) {
                ph[(u * 16 + ld_c) * 2 + (ld_s & 1)] = hi;
                pl[(u * 16 + ld_c) * 2 + (ld_s & 1)] = lo;
            } else if (a.dbg & 16) {
                asm volatile("" :: "v"(hi.x), "v"(hi.y), "v"(lo.x), "v"(lo.y));     // timing: staging read + split stay, the ring stores go
            }
        }
        __syncthreads();
        // ================= phase B: one cascade tile per wave =================
        if (has_stage && (st < 3 || (k & 1) == 0) && !(a.dbg & 2)) {
            const uint4* ph = lds + off_in;
            const uint4* pl = ph + ru_in * 16;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) {
                const int u = wrap_once(u_in + 4 * ks + q, ru_in);
                acc = mfma3(th[ks], tl[ks], ph[u * 16 + c16], pl[u * 16 + c16], acc);
            }
            // D[row = output n][col = clip]: this lane holds outputs 4q .. 4q + 3 of the tile for clip c16
            if (st + 1 < NL) {
                uint2 hi, lo;
                split4(acc[0], acc[1], acc[2], acc[3], hi, lo);
                const int u = wrap_once(u_out + (q >> 1), ru_out);
                uint2* oh = reinterpret_cast<uint2*>(lds + off_out);
                uint2* ol = oh + ru_out * 32;
                oh[(u * 16 + c16) * 2 + (q & 1)] = hi;
                ol[(u * 16 + c16) * 2 + (q & 1)] = lo;
            } else if (EMIT) {
                // level NL of this launch -> memory: sample number m = relative index + (M0 >> NL)
                const long long m = (static_cast<long long>(kStep >> NL) * k - 64) + 4 * q + (M0 >> NL);
                const long long idx = m + a.pad_next;
                if (my_clip < a.batch && m >= (Ms >> NL) && m < (Me >> NL) && idx >= 0 && idx + 4 <= a.next_count)
                    store_f32x4(a.next + my_clip * a.next_stride + idx, acc);
            }
        }
        // ================= phase B: filter banks =================
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            while (k >= kfire[l]) {                                     // the tap window of frame tn[l] is complete: run its bank
                if (a.dbg & 1) { kfire[l] = 0x7fffffff; break; }
                LevelCP L = level_ptr(l);
                const int t = tn[l];
                long long a_unit;
                (void)window_of(L, l, t, a_unit);
                const int ru = L->ring_units;
                const int ub = wrap_any(a_unit, ru);
                const int rot = t + 3 * l;
                const uint4* ph = lds + L->lds_off;
                const uint4* pl = ph + ru * 16;
                const int n_tiles = L->n_tiles;
                const uint4* tbase = a.table + L->table_off + static_cast<long long>(pi[l]) * L->phase_stride + lane;
#pragma unroll
                for (int j = 0; j < kMaxTiles; ++j) {
                    if (j >= n_tiles || ((j + rot) & 7) != wave) continue;
                    const int b_lo = L->blk_lo[j], b_hi = L->blk_hi[j];
                    const uint4* tbp = tbase + j * 128;
                    uint4 fh[kMaxBlk], fl[kMaxBlk];
#pragma unroll
                    for (int b = 0; b < kMaxBlk; ++b) {                  // every fragment requested up front, branch-free (clamped index)
                        const int blk = b_lo + b <= b_hi ? b_lo + b : b_hi;
                        fh[b] = tbp[static_cast<long long>(blk) * (kMaxTiles * 128)];
                        fl[b] = tbp[static_cast<long long>(blk) * (kMaxTiles * 128) + 64];
                    }
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int b = 0; b < kMaxBlk; ++b) {
                        if (b_lo + b <= b_hi) {
                            const int u = wrap_once(ub + 4 * (b_lo + b) + q, ru);
                            acc = mfma3(fh[b], fl[b], ph[u * 16 + c16], pl[u * 16 + c16], acc);
                        }
                    }
                    // every fragment register is read here, also those of the clamped (unused) requests: a load left pending at the end
                    // of this rare block would make hipcc wait for it wherever its register is next written -- on every step
#pragma unroll
                    for (int b = 0; b < kMaxBlk; ++b) asm volatile("" :: "v"(fh[b].x), "v"(fl[b].x));
                    // D[row = 2 * bin + (re | im)][col = clip]: this lane holds bins 2q, 2q + 1 of the tile (re, im, re, im)
                    if (my_clip < a.batch) {
                        float* o = a.out + my_clip * a.out_clip_stride + static_cast<long long>(t) * a.n_bins_total + L->k0 + kTileBins * j + 2 * q;
                        const bool live = t < tc;
                        const float m0 = __logf(1.f + __builtin_amdgcn_sqrtf(acc[0] * acc[0] + acc[1] * acc[1]));
                        const float m1 = __logf(1.f + __builtin_amdgcn_sqrtf(acc[2] * acc[2] + acc[3] * acc[3]));
                        const int nb = L->n_bins;
                        if (kTileBins * j + 2 * q < nb) store_f32(o, live ? m0 : 0.f);
                        if (kTileBins * j + 2 * q + 1 < nb) store_f32(o + 1, live ? m1 : 0.f);
                    }
                }
                ++tn[l];
                pi[l] = pi[l] + 1 == L->period ? 0 : pi[l] + 1;
                long long au;
                kfire[l] = tn[l] < t_end ? window_of(L, l, tn[l], au) : 0x7fffffff;
            }
        }
        dma(k + kStage);                                                // refill the slot this step consumed (this wave's own 1 KiB of it)
        // ---- advance the ring positions ----
        u_a = wrap_once(u_a + 16, ru0);
        if (has_stage) {
            u_in = wrap_once(u_in + inc_in, ru_in);                     // (stage 3 runs on even k only; its position advances every step all the same)
            if (st + 1 < NL) u_out = wrap_once(u_out + inc_out, ru_out);
        }
    }
}

}  // namespace fz

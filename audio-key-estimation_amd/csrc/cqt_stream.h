// Engine 5: the half-band cascade of the top four levels as a register-resident streaming MFMA kernel.
// (implementation include of cqt.hip; lives in its translation unit)
//
// Why.  The VALU cascade (cqt_cascade_kernel) costs 25 vector operations per output sample -- 36 M wave-instructions per 256 clips,
// a floor of ~30 us at perfect issue and 135 us as measured -- and is the longest kernel of an inference step.  A half-band stage is
//     v[m] = h0 u[2m] + sum_{j=-12}^{11} g[j] o[m + j],      o[i] = u[2i + 1]  (the odd samples: a DENSE 24-tap FIR, no zero taps),
// i.e. a banded Toeplitz product, and with the CLIPS as the MFMA's N dimension nothing has to be transposed:
//
//     D[m = 16 outputs][n = 16 clips] = T[m][k = 32 odd samples] x O[k][n]          v_mfma_f32_16x16x32_bf16, split operands (3 products)
//
// The B operand of that instruction wants, in lane (n, q = lane >> 4), eight consecutive k of column n: eight samples of ONE clip, which is
// what a lane loads from its clip's row anyway.  The Toeplitz matrices (4 of them: two output tiles x two input blocks) are constants in
// registers.  The D fragment leaves 4 consecutive outputs of clip n in lane (n, q); two v_permlane swaps turn the two tiles of a step
// into 8 consecutive samples per lane -- exactly the piece of the NEXT level's chunk that lane must hold -- so the four stages chain
// through registers: no LDS, no barrier, one wave = 16 clips x a time segment.
//
// Index bookkeeping (all absolute).  Level l is cut into chunks of 64 samples, chunk T = [64 T - kOff[l], 64 T - kOff[l] + 64), and lane
// (n, q) holds its pieces q and 4 + q (samples 8q..8q+7 and 32+8q..32+8q+7 of the chunk) of clip n.  The chunk's odd samples form the
// k-block O_T (slot e < 4: piece q's odd sample e, slot e >= 4: piece 4+q's odd sample e - 4; the Toeplitz constants are built for that
// order), its even samples the centre taps.  Step T of a stage has O_{T-1} and O_T and produces
//     tile B_{T-1} = outputs 16..31 of block T-1   and   tile A_T = outputs 0..15 of block T          (block T = the 32 outputs centred on chunk T's even samples)
// = 32 consecutive samples of level l+1 starting at 32 T - kOff[l+1], kOff[l+1] = kOff[l] / 2 + 16.  Two such groups are the next level's
// chunk.  kOff = 0, 16, 24, 28, 30: even up to level 4, which is as far as this kernel goes (level 4 leaves as f32 for engine 3's VALU
// cascade, which makes levels 5..7 from it: cqt_cascade_kernel; 1/16 of the samples).
//
// Segments.  16 clips x 64 segments = one wave per SIMD of the chip.  A segment starts `warm` chunks early with zero state (24 chunks: the
// level-3 stage needs 1 + 2 (1 + 2 (1 + 2)) = 15 level-0 chunks of true history, the chunk assembly of each level one more) and stores
// nothing until its own first chunk; what a chunk's processing stores is the same whichever segment runs it, so every sample is stored once.
//
// Stores.  Levels 0..3 leave as split words ((bf16 hi << 16) | bf16 lo, the filter bank's operand format) and only within `need[l]` raw
// samples of a frame centre (6.8 % / 14 % / 27 % / 54 % of the levels).  Arithmetic: taps and samples as bf16 hi + lo (2^-17 each), products
// Th Oh + Tl Oh + Th Ol in f32 accumulators, the centre tap an exact f32 fma: level signals within ~1e-5 of the VALU cascade's.
#pragma once

namespace sm {

constexpr int kStages = 4;
constexpr int kOff[kStages + 1] = {0, 16, 24, 28, 30};

struct Args {
    const float* x;                 // [batch][x_stride] audio
    long long x_stride;
    int n;                          // samples per clip (ragged: of the longest)
    const long long* n_clip;        // ragged batches: samples of each clip, or null
    int batch;
    unsigned int* ph[kStages];      // split planes of levels 0..3: sample m at ph[l][clip * p_stride[l] + m + ppad]
    long long p_stride[kStages];
    int p_count[kStages];
    int need[kStages];              // store level l only within need[l] raw samples of a frame centre (< 0: everywhere)
    int ppad;
    float* next;                    // level 4, f32: sample m at next[clip * next_stride + m + pad_next], for 0 <= m + pad_next < next_count
    long long next_stride;
    int next_count, pad_next;
    int hop;
    int c_begin, c_end;             // level-0 chunks [c_begin, c_end) are processed; everything before c_begin is silence
    int chunks_per_seg, warm;
    int dbg;                        // diagnostic build (AKE_SM_ABLATE; timing only, wrong results): 1 no audio loads, 2 no stage work, 4 no stores
    const uint4* toep;              // [4 matrices: A_T x O_{T-1}, A_T x O_T, B_{T-1} x O_{T-1}, B_{T-1} x O_T][hi | lo][64 lanes] x 8 bf16
    float h0;
};

typedef unsigned int u32;
typedef u32 u32x4s __attribute__((ext_vector_type(4)));
typedef float f32x4s __attribute__((ext_vector_type(4)));
typedef float f32x2s __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8s __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2s __attribute__((ext_vector_type(2)));
typedef float f4un __attribute__((ext_vector_type(4), aligned(4)));

struct Frag {                       // 8 values of one lane as bf16 hi and bf16 lo (value = hi + lo to 2^-17)
    u32x4s h, l;
};

__device__ __forceinline__ void split_pair(float a, float b, u32& h, u32& l) {
    const f32x2s v = {a, b};
    const bf16x2s hv = __builtin_convertvector(v, bf16x2s);                      // v_cvt_pk_bf16_f32 (RNE)
    const f32x2s r = v - __builtin_convertvector(hv, f32x2s);
    const bf16x2s lv = __builtin_convertvector(r, bf16x2s);
    h = __builtin_bit_cast(u32, hv);
    l = __builtin_bit_cast(u32, lv);
}

__device__ __forceinline__ Frag split8(float v0, float v1, float v2, float v3, float v4, float v5, float v6, float v7) {
    Frag f;
    u32 h, l;
    split_pair(v0, v1, h, l); f.h[0] = h; f.l[0] = l;
    split_pair(v2, v3, h, l); f.h[1] = h; f.l[1] = l;
    split_pair(v4, v5, h, l); f.h[2] = h; f.l[2] = l;
    split_pair(v6, v7, h, l); f.h[3] = h; f.l[3] = l;
    return f;
}

__device__ __forceinline__ void mma3(f32x4s& acc, const Frag& t, const Frag& o) {
    const bf16x8s th = __builtin_bit_cast(bf16x8s, t.h), tl = __builtin_bit_cast(bf16x8s, t.l);
    const bf16x8s oh = __builtin_bit_cast(bf16x8s, o.h), ol = __builtin_bit_cast(bf16x8s, o.l);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(th, oh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tl, oh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(th, ol, acc, 0, 0, 0);
}

// 8 consecutive samples -> 8 split words, two 16-byte stores (idx: word index of the first, a multiple of 4)
__device__ __forceinline__ void store_words8(unsigned int* pw, long long idx, const float (&c)[8]) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        u32 h01, l01, h23, l23;
        split_pair(c[4 * h], c[4 * h + 1], h01, l01);
        split_pair(c[4 * h + 2], c[4 * h + 3], h23, l23);
        // perm(a, b, sel): bytes 0-3 index b, 4-7 index a  ->  word = (hi << 16) | lo
        const uint4 w = make_uint4(__builtin_amdgcn_perm(h01, l01, 0x05040100), __builtin_amdgcn_perm(h01, l01, 0x07060302),
                                   __builtin_amdgcn_perm(h23, l23, 0x05040100), __builtin_amdgcn_perm(h23, l23, 0x07060302));
        *reinterpret_cast<uint4*>(pw + idx + 4 * h) = w;
    }
}

struct Stage {
    Frag oprev;                     // O_{T-1}
    float eprev[4];                 // even samples 16+4q .. 16+4q+3 of chunk T-1 (centre taps of tile B_{T-1})
    float half[8];                  // first group of the next level's chunk being assembled (lives inside one 8-chunk period only)
};

struct Ctx {                        // wave-uniform state of the segment + this lane's row
    int q, clip, live;              // live: the lane's clip exists (idle lanes of the last clip group redo the last clip, never store)
    int owned;                      // the 8-chunk period being processed belongs to this segment (stores enabled)
    int fc[kStages];                // smallest frame centre (raw samples) not yet left behind by level l's store window
};

// does [lo, hi) (raw samples) come within `need` of a frame centre?  Advances the level's frame cursor (lo never decreases).  All scalar.
__device__ __forceinline__ bool near_frame(int& fc, int lo, int hi, int need, int hop) {
    if (need < 0) return true;
    while (fc < lo - need) fc += hop;
    return fc <= hi - 1 + need;
}

template <int LV>                   // LV = 1..4: the level the 8 samples c[] belong to; gs = index of the group's first sample (32 per group)
__device__ __forceinline__ void store_level(const Args& a, Ctx& cx, const float (&c)[8], int gs) {
    if (!cx.owned) return;
    constexpr int P = LV < kStages ? LV : 0;
    if (LV < kStages) {
        if (!near_frame(cx.fc[P], gs * (1 << LV), (gs + 32) * (1 << LV), a.need[P], a.hop)) return;
        const int g0 = gs + a.ppad;                                               // the whole group inside the row (its pads are wider than any window)
        if (g0 < 0 || g0 + 32 > a.p_count[P]) return;
        if (cx.live) store_words8(a.ph[P], static_cast<long long>(cx.clip) * a.p_stride[P] + g0 + 8 * cx.q, c);
    } else {
        if (!cx.live) return;
        float* row = a.next + static_cast<long long>(cx.clip) * a.next_stride;
        const int idx = gs + 8 * cx.q + a.pad_next;                               // == 2 (mod 4): 8-byte stores
        const int g0 = gs + a.pad_next;
        if (g0 >= 0 && g0 + 32 <= a.next_count) {
#pragma unroll
            for (int k = 0; k < 8; k += 2) *reinterpret_cast<float2*>(row + idx + k) = make_float2(c[k], c[k + 1]);
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (idx + k >= 0 && idx + k < a.next_count) row[idx + k] = c[k];
        }
    }
}

// Step K (of the level's 8 >> L steps per period) of stage L: raw = the lane's 16 samples of level-L chunk T = (c0 >> L) + K (pieces q, 4 + q).
// c0: the period's first level-0 chunk, a multiple of 8, so every parity below is a compile-time constant and the period is straight-line code.
template <int L, int K>
__device__ __forceinline__ void run_stage(const Args& a, Ctx& cx, Stage (&st)[kStages], const Frag (&toep)[4], const float (&raw)[16], int c0) {
    Stage& s = st[L];
    const int T = (c0 >> L) + K;
    const Frag oc = split8(raw[1], raw[3], raw[5], raw[7], raw[9], raw[11], raw[13], raw[15]);
    f32x4s accA = {0.f, 0.f, 0.f, 0.f}, accB = {0.f, 0.f, 0.f, 0.f};
    mma3(accA, toep[0], s.oprev);
    mma3(accB, toep[2], s.oprev);
    mma3(accA, toep[1], oc);
    mma3(accB, toep[3], oc);
    float c[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float av = fmaf(a.h0, s.eprev[i], accB[i]);                         // tile B_{T-1}, row 4q + i
        const float bv = fmaf(a.h0, raw[2 * i], accA[i]);                         // tile A_T, row 4q + i
        // rows (q) of the two tiles -> 8 consecutive samples per lane: lanes 32-63 of a <-> lanes 0-31 of b, then odd rows of a <-> even rows of b
        auto r1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(av), __float_as_uint(bv), false, false);
        auto r2 = __builtin_amdgcn_permlane16_swap(r1[0], r1[1], false, false);
        c[i] = __uint_as_float(r2[0]);
        c[4 + i] = __uint_as_float(r2[1]);
    }
    s.oprev = oc;
#pragma unroll
    for (int i = 0; i < 4; ++i) s.eprev[i] = raw[8 + 2 * i];
    store_level<L + 1>(a, cx, c, 32 * T - kOff[L + 1]);
    if constexpr (L + 1 < kStages) {
        if constexpr (K & 1) {
            float nr[16];
#pragma unroll
            for (int i = 0; i < 8; ++i) { nr[i] = s.half[i]; nr[8 + i] = c[i]; }
            run_stage<L + 1, (K >> 1)>(a, cx, st, toep, nr, c0);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) s.half[i] = c[i];
        }
    }
}

constexpr int kBufs = 4, kAhead = 3;                                              // audio chunks: 4 register buffers, loads 3 chunks ahead

// One wave = 16 clips x one time segment.  Measured at 256 clips (tests/tools/cqt_engine5_ablate.py, diagnostic build): 0.140 ms, of which vector /
// scalar issue of the lone wave 0.076 (loads and stores off), the stores +0.03..0.046, the loads +0.018 where the period is branch-free (a
// branch around a load made every join wait for vmcnt(0): 0.203 ms).  A producer / consumer split over two waves per SIMD (stage 0 | stages
// 1-3 through an LDS ring) halves the issue time on paper but the compiler needs 373 registers for the producer role and spills: 0.27 ms.
__global__ __launch_bounds__(64) void cqt_stream_kernel(Args a) {
    const int lane = threadIdx.x;
    Ctx cx;
    cx.q = lane >> 4;
    const int clip_raw = blockIdx.y * 16 + (lane & 15);
    cx.live = clip_raw < a.batch;
    cx.clip = cx.live ? clip_raw : a.batch - 1;
    const int seg = blockIdx.x;
    const int c_own = a.c_begin + seg * a.chunks_per_seg;                        // (c_begin, chunks_per_seg, warm: multiples of 8)
    if (c_own >= a.c_end) return;                                                 // (the whole workgroup: no barrier is left waiting)
    const int c_stop = c_own + a.chunks_per_seg < a.c_end ? c_own + a.chunks_per_seg : a.c_end;
    const int c_start = seg == 0 ? c_own : c_own - a.warm;
#pragma unroll
    for (int l = 0; l < kStages; ++l) cx.fc[l] = 0;

    Frag toep[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        toep[m].h = __builtin_bit_cast(u32x4s, a.toep[(2 * m) * 64 + lane]);
        toep[m].l = __builtin_bit_cast(u32x4s, a.toep[(2 * m + 1) * 64 + lane]);
    }
    Stage st[kStages];
#pragma unroll
    for (int l = 0; l < kStages; ++l) {
        st[l].oprev.h = u32x4s{0, 0, 0, 0}; st[l].oprev.l = u32x4s{0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 4; ++i) st[l].eprev[i] = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) st[l].half[i] = 0.f;
    }

    // this lane's clip: samples [0, n_lane); the wave's common range decides between the branch-free loads and the masked ones
    int n_lane = a.n;
    if (a.n_clip) { const long long nc = a.n_clip[cx.clip]; n_lane = nc < 0 ? 0 : (nc < a.n ? static_cast<int>(nc) : a.n); }
    int n_min = n_lane, n_max = n_lane;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) { n_min = min(n_min, __shfl_xor(n_min, o)); n_max = max(n_max, __shfl_xor(n_max, o)); }
    n_min = __builtin_amdgcn_readfirstlane(n_min);
    n_max = __builtin_amdgcn_readfirstlane(n_max);
    const int c_fast_end = n_min >> 6;                                            // chunks [0, c_fast_end) lie inside every clip of the wave
    const int c_zero = (n_max + 63) >> 6;                                         // chunks >= c_zero are silence

    // Loads.  Instruction j of a chunk gives lane (n, q) the 4 samples of group q + 4 j (16 groups of 4 per chunk): the four lanes of a clip read
    // 64 contiguous bytes, 16 requests per wave instruction (pieces of 16 B strided by 32 B -- the layout the stages want -- would be 64).  Two
    // permlane swaps per register pair then deal the groups out as pieces: (R0, R1) -> groups 2q, 2q+1 = piece q; (R2, R3) -> piece 4 + q.
    // FAST: every chunk of the period (and of its look-ahead) lies inside every clip of the wave: no branch anywhere near a load, so the
    // compiler's vmcnt bookkeeping keeps kAhead chunks in flight (a branch around a load makes the join wait for vmcnt(0): measured, the
    // branchy form ran at one chunk per memory latency: 117 us of a 203 us launch).
    auto load_groups = [&](f4un (&g)[4], int c) {                                  // branch-free, FAST periods
        const float* p = a.x + static_cast<long long>(cx.clip) * a.x_stride + 64ll * c + 4 * cx.q;
#pragma unroll
        for (int j = 0; j < 4; ++j) g[j] = *reinterpret_cast<const f4un*>(p + 16 * j);
    };
    auto load_groups_masked = [&](f4un (&g)[4], int c) {                           // edges: samples outside [0, n_lane) read as zero
        if (a.dbg & 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) g[j] = f4un{1.f, 1.f, 1.f, 1.f};
            return;
        }
        if (c < 0 || c >= c_zero) {
#pragma unroll
            for (int j = 0; j < 4; ++j) g[j] = f4un{0.f, 0.f, 0.f, 0.f};
            return;
        }
        const int s0 = 64 * c + 4 * cx.q;
        const float* p = a.x + static_cast<long long>(cx.clip) * a.x_stride + s0;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) g[j][i] = s0 + 16 * j + i < n_lane ? p[16 * j + i] : 0.f;
    };
    auto deal = [&](const f4un (&g)[4], float (&r)[16]) {                          // groups q + 4 j  ->  pieces q (r[0..7]) and 4 + q (r[8..15])
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                auto r1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(g[2 * h][i]), __float_as_uint(g[2 * h + 1][i]), false, false);
                auto r2 = __builtin_amdgcn_permlane16_swap(r1[0], r1[1], false, false);
                r[8 * h + i] = __uint_as_float(r2[0]);
                r[8 * h + 4 + i] = __uint_as_float(r2[1]);
            }
    };
    auto store_level0 = [&](const float (&r)[16], int c) {                        // the split copy of the audio near the frame centres
        if (!cx.owned) return;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int g = 64 * c + 32 * h;
            if (!near_frame(cx.fc[0], g, g + 32, a.need[0], a.hop)) continue;
            const int g0 = g + a.ppad;
            if (g0 < 0 || g0 + 32 > a.p_count[0] || !cx.live) continue;
            float cc[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) cc[i] = r[8 * h + i];
            store_words8(a.ph[0], static_cast<long long>(cx.clip) * a.p_stride[0] + g0 + 8 * cx.q, cc);
        }
    };

    f4un buf[kBufs][4];
#pragma unroll
    for (int k = 0; k < kAhead; ++k) load_groups_masked(buf[k], c_start + k);
    for (int c0 = c_start; c0 < c_stop; c0 += 8) {                                // one period: 8 level-0 chunks = 4 / 2 / 1 chunks of levels 1 / 2 / 3
        cx.owned = c0 >= c_own && !(a.dbg & 4);
        const bool fast = c0 >= 0 && c0 + 8 + kAhead <= c_fast_end && !(a.dbg & 1);
#define AKE_SM_STEP(K_, LOAD_)                                                                   \
        {                                                                                         \
            LOAD_(buf[(K_ + kAhead) % kBufs], c0 + K_ + kAhead);                                  \
            float raw[16];                                                                        \
            deal(buf[K_ % kBufs], raw);                                                           \
            store_level0(raw, c0 + K_);                                                           \
            if (!(a.dbg & 2)) run_stage<0, K_>(a, cx, st, toep, raw, c0);                         \
            else if (cx.owned && cx.live && raw[0] == 123.f) a.next[0] = 1.f;                     \
        }
        if (fast) {
            AKE_SM_STEP(0, load_groups) AKE_SM_STEP(1, load_groups) AKE_SM_STEP(2, load_groups) AKE_SM_STEP(3, load_groups)
            AKE_SM_STEP(4, load_groups) AKE_SM_STEP(5, load_groups) AKE_SM_STEP(6, load_groups) AKE_SM_STEP(7, load_groups)
        } else {
            AKE_SM_STEP(0, load_groups_masked) AKE_SM_STEP(1, load_groups_masked) AKE_SM_STEP(2, load_groups_masked) AKE_SM_STEP(3, load_groups_masked)
            AKE_SM_STEP(4, load_groups_masked) AKE_SM_STEP(5, load_groups_masked) AKE_SM_STEP(6, load_groups_masked) AKE_SM_STEP(7, load_groups_masked)
        }
#undef AKE_SM_STEP
    }
}

// The four Toeplitz matrices in MFMA A-fragment order.  Lane (m = lane & 15, kg = lane >> 4), element e: k-slot 8 kg + e holds the chunk's odd
// sample o_loc = (e < 4 ? 4 kg + e : 16 + 4 kg + e - 4); entry = g[j], j = (o index) - (output index), g[j] = hodd[j] (j >= 0), hodd[-j - 1] (j < 0).
inline void build_toeplitz(const float* hodd, int n_odd, std::vector<uint16_t>& out) {
    auto bf16_rne = [](float v) -> uint16_t {
        uint32_t u;
        std::memcpy(&u, &v, 4);
        u += 0x7FFFu + ((u >> 16) & 1u);
        return static_cast<uint16_t>(u >> 16);
    };
    auto bf16_f32 = [](uint16_t h) { uint32_t u = static_cast<uint32_t>(h) << 16; float f; std::memcpy(&f, &u, 4); return f; };
    out.assign(static_cast<size_t>(4) * 2 * 64 * 8, 0);
    const int shift[4] = {-32, 0, -16, 16};                                       // j = o_loc + shift - m   (A x O_{T-1}, A x O_T, B x O_{T-1}, B x O_T)
    for (int mat = 0; mat < 4; ++mat)
        for (int lane = 0; lane < 64; ++lane)
            for (int e = 0; e < 8; ++e) {
                const int m = lane & 15, kg = lane >> 4;
                const int oloc = e < 4 ? 4 * kg + e : 16 + 4 * kg + (e - 4);
                const int j = oloc + shift[mat] - m;
                float g = 0.f;
                if (j >= 0 && j < n_odd) g = hodd[j];
                else if (j < 0 && -j - 1 < n_odd) g = hodd[-j - 1];
                const uint16_t hi = bf16_rne(g);
                const uint16_t lo = bf16_rne(g - bf16_f32(hi));
                out[((static_cast<size_t>(2 * mat) * 64 + lane) * 8) + e] = hi;
                out[((static_cast<size_t>(2 * mat + 1) * 64 + lane) * 8) + e] = lo;
            }
}

}  // namespace sm

// CQT engine 4: ONE streaming pass over the audio for the top four octaves -- the half-band decimator cascade AND the per-phase
// filter bank both run on v_mfma_f32_16x16x32_bf16 with split operands, and the level signals never leave the LDS.
//
// Why: engines 2/3 write every level signal to HBM where a frame's tap window reads it and read it back in a second
// kernel (132 MB written + 202 MB read per 256 clips next to the 339 MB of audio), and their cascade is a VALU FIR with 8
// workgroup barriers per 4096 samples.  Here a workgroup owns 16 clips x one time segment and walks it in steps of 128 samples.
// Eight waves; wave 7 (whose cascade stage runs every other step only, and which takes no filter-bank tiles) is also the LOADER.
//
//   loader   keeps kStage - 1 audio tiles (16 clips x 128 samples, 8 KB) in flight by LDS-DMA into a staging ring; before the
//            step's barrier it waits (counted vmcnt) for the tile that is converted in that step.  ONE wave does all of it because
//            vmcnt retires in order PER WAVE: behind a deep queue of HBM loads every other load of the same wave takes the
//            queue's latency (measured: a filter-bank fragment load took 3 us behind 5 audio tiles), so the loader wave issues no
//            other vector-memory load (its level-4 stores are untracked asm stores).
//   barrier  (the only one per step)
//   compute  (a) converts the NEXT tile: 4 staged f32 samples per thread -> (bf16 hi, bf16 lo) -> level-0 ring;
//            (b) one 16-outputs x 16-clips tile of the cascade per wave -- waves 0-3 level 0->1, 4-5 level 1->2, 6 level 2->3,
//            7 level 3->4 (every other step) -- as D[output n][clip] = Toeplitz(h)[n][k] . X[k][clip]: the filter taps are 6 constant
//            A fragments (3 k-steps x hi/lo), the B operand is three aligned 16-byte LDS reads of the input ring, 9 MFMAs per tile
//            in three independent chains; each stage reads what the stage above wrote in the PREVIOUS step, so one barrier per step
//            is enough;
//            (c) the filter banks, in PARTS of kPart 32-tap blocks: as soon as a part of a frame's tap window is complete in a
//            level's ring, the wave that owns an N-tile of that (frame, octave) accumulates D[bin re/im][clip] += W[phase][col][tap] .
//            X[tap][clip] over the part; the table fragments were requested one step earlier (per-level registers), the running sums
//            stay in registers until the last part.  Parts keep the rings short (a step's production twice over + one part instead
//            of + one whole window: 55 KB instead of 104 KB), which is what leaves room for a deep staging ring.  Parts need the
//            windows of consecutive frames of a level to be disjoint enough (hop / 2^o >= window - part): the host checks it and
//            uses engine 3 otherwise; octaves below the fourth always overlap and go through engine 3's kernels, fed with level 4.
//
// Rings are laid out [sample / 8][clip][8 samples] (hi plane, lo plane): a B operand (lane = clip + 16 q reads samples 8q..8q+7) is
// 1 KiB of consecutive bytes per wave -- conflict-free without padding.  Windows are anchored on multiples of 8 samples so that
// those reads stay 16-byte aligned; the anchor's offset from the frame centre joins the fractional phase: one filter bank per
// (t * hop - uh * 2^o) mod (8 * 2^o).
//
// HBM traffic: audio once (+ halo re-reads at segment seams), level 4 (f32, 1/16 of the audio) out, the log-magnitudes out.
#pragma once

namespace fz {

constexpr int kClips = 16;     // clips per workgroup = MFMA N
constexpr int kStep = 128;     // input samples per step
constexpr int kNT = 512;       // threads: 8 waves; wave 7 is also the loader
constexpr int kMaxLv = 4;      // octaves per launch
constexpr int kStage = 9;      // audio tiles (8 KB each) of the staging ring: kStage - 1 in flight (8 DMA instructions each; vmcnt counts to 63)
constexpr int kPart = 3;       // 32-tap blocks per filter-bank part
constexpr int kMaxBlk = 12;    // 32-tap blocks of a window

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
    unsigned long long* stamps;   // diagnostic build (AKE_CQT_FZ_STAMP): [8 waves][8] cycle sums of the step loop's sections, workgroup 0
    int dbg;                   // timing experiments only (AKE_CQT_FZ_DBG): 1 no filter banks, 2 no cascade tiles, 4 no ring-0 writes, 256 no DMA, 512 no DMA wait
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

__device__ __forceinline__ unsigned long long stamp_now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}

template <int NL, bool EMIT, bool STAMP = false>
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
    uint4* const stage = lds + lds_total;                             // [kStage][512] float4: tile t in slot t % kStage, chunk i = wave i's lanes
    const long long ld_max = static_cast<long long>(a.x_bytes / 4) - 4;   // last float4 inside the tensor
    uint4 wh[3], wl[3];                                               // half-band Toeplitz A fragments
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) { wh[ks] = a.toep[(ks * 2 + 0) * 64 + lane]; wl[ks] = a.toep[(ks * 2 + 1) * 64 + lane]; }

    // Chunk (wave i, lane) of a tile = clip (i & 1) * 8 + (lane & 7), samples 4 * ((i >> 1) * 8 + (lane >> 3)) .. + 3 of the step's 128:
    // a DMA instruction fetches 8 clips x 128 contiguous bytes, and the ring-0 stores of a 16-lane group (8 clips x 2 halves of one
    // 8-sample unit) fall on 32 different banks.
    // =========================== loader (wave 7) ===========================
    const bool is_loader = wave == 7;
    long long row[8];                                                 // tensor index of the lane's 4 samples in tile 0, chunk i
    long long rmin = 0x7fffffffffffffffll, rmax = -0x7fffffffffffffffll;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = (i & 1) * 8 + (lane & 7);
        const int clip = grp * kClips + c < a.batch ? grp * kClips + c : a.batch - 1;
        row[i] = static_cast<long long>(clip) * a.x_stride + M0 + a.pad_in + 4 * ((i >> 1) * 8 + (lane >> 3));
        rmin = row[i] < rmin ? row[i] : rmin;
        rmax = row[i] > rmax ? row[i] : rmax;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {                                // wave-wide extremes: decide per tile, with two scalar compares,
        const long long o1 = __shfl_xor(rmin, d), o2 = __shfl_xor(rmax, d);   // whether any address needs clamping
        rmin = o1 < rmin ? o1 : rmin;
        rmax = o2 > rmax ? o2 : rmax;
    }
    rmin = (static_cast<long long>(__builtin_amdgcn_readfirstlane(static_cast<int>(rmin >> 32))) << 32) |
           static_cast<unsigned int>(__builtin_amdgcn_readfirstlane(static_cast<int>(rmin)));
    rmax = (static_cast<long long>(__builtin_amdgcn_readfirstlane(static_cast<int>(rmax >> 32))) << 32) |
           static_cast<unsigned int>(__builtin_amdgcn_readfirstlane(static_cast<int>(rmax)));
    auto dma = [&](int t) {                                           // tile t -> slot t % kStage (out-of-clip samples are masked at use)
        const long long adv = static_cast<long long>(t) * kStep;
        const float* src[8];
        if (rmin + adv >= 0 && rmax + adv <= ld_max) {                // (uniform) interior tile: no clamps
#pragma unroll
            for (int i = 0; i < 8; ++i) src[i] = a.x + (row[i] + adv);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                long long gi = row[i] + adv;
                gi = gi < 0 ? 0 : (gi > ld_max ? ld_max : gi);        // any in-tensor address will do where the samples are masked
                src[i] = a.x + gi;
            }
        }
        const unsigned int dst = static_cast<unsigned int>(reinterpret_cast<unsigned long long>(stage + (t % kStage) * kNT));
        unsigned int keep;
        // 8 x (1 KiB per wave instruction) into consecutive KiB of the slot; M0 = LDS destination, restored at the end.  asm: hipcc
        // would order every later LDS access behind a builtin LDS-DMA with vmcnt(0)
        asm volatile("s_mov_b32 %0, m0\n\t"
                     "s_mov_b32 m0, %9\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, off\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, off\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %5, off\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %6, off\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %7, off\n\t"
                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %8, off\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(src[0]), "v"(src[1]), "v"(src[2]), "v"(src[3]), "v"(src[4]), "v"(src[5]), "v"(src[6]), "v"(src[7]), "s"(dst)
                     : "memory", "scc");
    };
    if (is_loader && !(a.dbg & 256)) for (int t = 0; t < kStage - 1; ++t) dma(t);

    // ---- converter role: this thread's 4 samples of every tile ----
    const int ld_c = (wave & 1) * 8 + (lane & 7), ld_s = (wave >> 1) * 8 + (lane >> 3);
    const int ld_clip = grp * kClips + ld_c < a.batch ? grp * kClips + ld_c : a.batch - 1;
    long long nv = a.n_valid;
    if (a.n_clip && a.L0 == 0) { const long long nc = a.n_clip[ld_clip]; nv = nc < 0 ? 0 : (nc < nv ? nc : nv); }
    const long long ld_row = static_cast<long long>(ld_clip) * a.x_stride;
    const long long ld_i0 = M0 + a.pad_in + 4 * ld_s;                 // array index of this thread's samples in tile 0
    // tiles [t_plain_lo, t_plain_hi] need no masking in ANY lane of this wave (all four samples inside the clip, vector inside the tensor)
    int t_plain_lo, t_plain_hi;
    {
        const long long lo_l = ld_i0 >= 0 ? 0 : (-ld_i0 + kStep - 1) / kStep;                       // first tile with gi >= 0
        const long long lim = nv - 4 < ld_max - ld_row ? nv - 4 : ld_max - ld_row;                   // gi <= lim
        const long long hi_l = lim < ld_i0 ? -1 : (lim - ld_i0) / kStep;
        int lo_i = static_cast<int>(lo_l > 0x3fffffff ? 0x3fffffff : lo_l), hi_i = static_cast<int>(hi_l > 0x3fffffff ? 0x3fffffff : hi_l);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const int o1 = __shfl_xor(lo_i, d), o2 = __shfl_xor(hi_i, d);
            lo_i = o1 > lo_i ? o1 : lo_i;
            hi_i = o2 < hi_i ? o2 : hi_i;
        }
        t_plain_lo = __builtin_amdgcn_readfirstlane(lo_i);
        t_plain_hi = __builtin_amdgcn_readfirstlane(hi_i);
    }

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
    const int ru0 = a.lv[0].ring_units;
    int u_a = 0;                                                      // level-0 ring unit of the tile converted next (tile 0 -> unit 0)

    // ---- frames this workgroup owns: centre sample number floor(t * hop / 2^L0) in [Ms, Me) ----
    auto tb = [&](long long X) -> long long { return X <= 0 ? 0 : ((X << a.L0) + a.hop - 1) / a.hop; };
    const int t_first = static_cast<int>(tb(Ms));
    const int t_end = static_cast<int>(tb(Me) < a.T ? tb(Me) : a.T);
    // The Level records are read with scalar loads through a laundered kernel-argument pointer INSIDE the (rare) blocks that need them:
    // as plain kernel arguments the compiler keeps all ~100 of their dwords live in SGPRs across the step loop and spills them
    // (measured twice: 600 - 1 700 SGPR spills and a slower loop).
    typedef const __attribute__((address_space(4))) Level* LevelCP;
    typedef const __attribute__((address_space(4))) char* CharCP;
    auto level_ptr = [&](int l) -> LevelCP {
        LevelCP pL = (LevelCP)((CharCP)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(Args, lv)) + l;
        asm volatile("" : "+s"(pL));
        return pL;
    };
    // first step whose frontier covers relative indices up to e of level l: frontier(l, k) = (128 >> l) * k - lag_l >= e
    auto fire_step = [&](int l, int e) -> int {
        const int x = e + (l == 0 ? -127 : (l == 1 ? 17 : 65));
        return x <= 0 ? 0 : (x + (127 >> l)) >> (7 - l);
    };
    // Per level: frame tn, its phase pi, the part of its window that is next (blocks [bp, bp + kPart)), the step it fires at, and
    // this wave's share of it: N-tile jt (or -1), ring unit ub of the part's first block, running sums, and the part's table
    // fragments (requested one step before the part fires; pend says whether they are).
    int tn[NL], pi[NL], bp[NL], kfire[NL], jt[NL], ub[NL], blo[NL], bhi[NL], au8[NL], nblk[NL], rul[NL];
    bool pend[NL];
    f32x4 accp[NL];
    uint4 fh[NL][kPart], fl[NL][kPart];
    const uint4* tptr[NL];
    auto open_frame = [&](int l, LevelCP L) {                         // frame tn[l] becomes the level's current frame
        jt[l] = -1;
        if (tn[l] >= t_end) { kfire[l] = 0x7fffffff; return; }
        const int t = tn[l];
        const int sh = a.L0 + l;
        const long long E = static_cast<long long>(t) * a.hop - (static_cast<long long>(L->uh) << sh);   // nominal window start, full-rate position
        const int au = static_cast<int>((E >> (sh + 3)) - ((M0 >> l) >> 3));                             // anchored start, relative index / 8
        const int ru = L->ring_units;
        int u = au % ru;                                              // (32-bit; once per frame)
        ub[l] = u < 0 ? u + ru : u;
        au8[l] = 8 * au;
        rul[l] = ru;
        const int n_blk = L->n_blk;
        nblk[l] = n_blk;
        // N-tile of this wave: (j + rot) % 7 == wave for j < n_tiles; waves 0..6 only (the loader takes none)
        const int rot = t + 3 * l;
        int j = (wave - rot) % 7;
        j = j < 0 ? j + 7 : j;
        blo[l] = 0; bhi[l] = -1;
        if (wave < 7 && j < L->n_tiles) {
            jt[l] = j;
            const __attribute__((address_space(4))) int* bl = (const __attribute__((address_space(4))) int*)L + offsetof(Level, blk_lo) / 4;
            const __attribute__((address_space(4))) int* bh = (const __attribute__((address_space(4))) int*)L + offsetof(Level, blk_hi) / 4;
            blo[l] = bl[j]; bhi[l] = bh[j];
            tptr[l] = a.table + L->table_off + static_cast<long long>(pi[l]) * L->phase_stride + j * 128 + lane;
        }
        bp[l] = 0;
        kfire[l] = fire_step(l, au8[l] + 32 * (kPart < n_blk ? kPart : n_blk) - 1);
        accp[l] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        LevelCP L = level_ptr(l);
        tn[l] = t_first;
        pi[l] = t_first % L->period;
        pend[l] = false;
        tptr[l] = a.table;
        open_frame(l, L);
    }
    auto request = [&](int l) {                                       // the current part's fragments -> registers (branch-free: clamped block index)
#pragma unroll
        for (int b = 0; b < kPart; ++b) {
            int blk = bp[l] + b;
            blk = blk < blo[l] ? blo[l] : (blk > bhi[l] ? bhi[l] : blk);
            fh[l][b] = tptr[l][static_cast<long long>(blk) * (kMaxTiles * 128)];
            fl[l][b] = tptr[l][static_cast<long long>(blk) * (kMaxTiles * 128) + 64];
        }
    };
    int k_evt = 0;                                                    // first step at which any level has a part to run or to request

    const int my_clip = grp * kClips + c16;                           // MFMA column of this lane
    int tc = a.T;                                                     // frames of this clip (the rest is zero padding)
    if (a.n_clip && my_clip < a.batch) { const long long nc = a.n_clip[my_clip]; const long long f = nc < 0 ? 0 : 1 + nc / a.hop; tc = f < tc ? static_cast<int>(f) : tc; }
    {   // Materialise the loop invariants that came from (tracked) loads BEFORE the step loop: hipcc waits for a pending load at its
        // first use, and a first use inside the loop becomes an s_waitcnt vmcnt(0) on EVERY step -- which, in the loader wave, drains
        // the untracked DMA queue (measured: one tile in flight instead of kStage - 1)
        unsigned int nlo = static_cast<unsigned int>(nv), nhi = static_cast<unsigned int>(static_cast<unsigned long long>(nv) >> 32);
        asm volatile("" : "+v"(nlo), "+v"(nhi), "+v"(tc));
        nv = static_cast<long long>((static_cast<unsigned long long>(nhi) << 32) | nlo);
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            asm volatile("" : "+v"(wh[ks].x), "+v"(wh[ks].y), "+v"(wh[ks].z), "+v"(wh[ks].w));
            asm volatile("" : "+v"(wl[ks].x), "+v"(wl[ks].y), "+v"(wl[ks].z), "+v"(wl[ks].w));
        }
    }
    __syncthreads();                                                  // (the zero fill of the rings)

    unsigned long long sm[6] = {0, 0, 0, 0, 0, 0}, ts[6];
    for (int k = -1; k < n_steps; ++k) {
        if (STAMP) ts[0] = stamp_now();
        // the loader: tile k + 1 is converted in this step -- everything but the kStage - 2 youngest tiles must have landed
        if (is_loader && !(a.dbg & 512)) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(8 * (kStage - 2)) : "memory");
        __syncthreads();
        if (is_loader && !(a.dbg & 256)) dma(k + kStage);               // into the slot of tile k, converted in step k - 1
        if (STAMP) ts[1] = stamp_now();
        // ================= (a) tile k + 1: staging -> level-0 ring =================
        {
            const f32x4 v = __builtin_bit_cast(f32x4, stage[((k + 1) % kStage) * kNT + tid]);
            float r0 = v[0], r1 = v[1], r2 = v[2], r3 = v[3];
            if (k + 1 < t_plain_lo || k + 1 > t_plain_hi) {           // (uniform) a clip boundary somewhere in this wave's chunk
                const long long gi = ld_i0 + static_cast<long long>(k + 1) * kStep;
                const long long left = gi < 0 ? 0 : nv - gi;          // valid samples from gi on (gi is a multiple of 4)
                // the last vector of the whole tensor may have been fetched from up to 3 floats earlier (the loader keeps every address inside it)
                const long long over = ld_row + gi - ld_max;
                const int sh = over > 0 ? static_cast<int>(over) : 0;
                const float e0 = sh == 0 ? v[0] : (sh == 1 ? v[1] : (sh == 2 ? v[2] : v[3]));
                const float e1 = sh == 0 ? v[1] : (sh == 1 ? v[2] : v[3]);
                const float e2 = sh == 0 ? v[2] : v[3];
                r0 = left > 0 ? e0 : 0.f; r1 = left > 1 ? e1 : 0.f; r2 = left > 2 ? e2 : 0.f; r3 = left > 3 ? v[3] : 0.f;
            }
            uint2 hi, lo;
            split4(r0, r1, r2, r3, hi, lo);
            const int u = wrap_once(u_a + (ld_s >> 1), ru0);
            uint2* ph = reinterpret_cast<uint2*>(lds + lds_off0);
            uint2* pl = reinterpret_cast<uint2*>(lds + lds_off0 + ru0 * 16);
            if (!(a.dbg & 4)) {
                ph[(u * 16 + ld_c) * 2 + (ld_s & 1)] = hi;
                pl[(u * 16 + ld_c) * 2 + (ld_s & 1)] = lo;
            }
            u_a = wrap_once(u_a + 16, ru0);
        }
        if (k < 0) continue;
        if (STAMP) ts[2] = stamp_now();
        // ================= (b) one cascade tile per wave =================
        if (has_stage && (st < 3 || (k & 1) == 0) && !(a.dbg & 2)) {
            const uint4* ph = lds + off_in;
            const uint4* pl = ph + ru_in * 16;
            // all six operands requested first, then three independent accumulation chains (one per k-step): the nine MFMAs issue back
            // to back instead of waiting for one another and for an LDS read each
            uint4 xh[3], xl[3];
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) {
                const int u = wrap_once(u_in + 4 * ks + q, ru_in);
                xh[ks] = ph[u * 16 + c16]; xl[ks] = pl[u * 16 + c16];
            }
            f32x4 ac[3];
#pragma unroll
            for (int ks = 0; ks < 3; ++ks)
                ac[ks] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wh[ks]), __builtin_bit_cast(bf16x8, xh[ks]), f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
            for (int ks = 0; ks < 3; ++ks)
                ac[ks] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wh[ks]), __builtin_bit_cast(bf16x8, xl[ks]), ac[ks], 0, 0, 0);
#pragma unroll
            for (int ks = 0; ks < 3; ++ks)
                ac[ks] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wl[ks]), __builtin_bit_cast(bf16x8, xh[ks]), ac[ks], 0, 0, 0);
            const f32x4 acc = (ac[0] + ac[1]) + ac[2];
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
        if (STAMP) ts[3] = stamp_now();
        // ================= (c) filter-bank parts whose blocks are complete =================
        if (k >= k_evt) {                                               // one scalar test per step
#pragma unroll
            for (int l = 0; l < NL; ++l) {
                while (k >= kfire[l]) {
                    if (a.dbg & 1) { kfire[l] = 0x7fffffff; break; }
                    const bool last = bp[l] + kPart >= nblk[l];
                    if (jt[l] >= 0) {
                        if (!pend[l]) request(l);
                        LevelCP L = level_ptr(l);
                        const int ru = rul[l];
                        const uint4* ph = lds + L->lds_off;
                        const uint4* pl = ph + ru * 16;
#pragma unroll
                        for (int b = 0; b < kPart; ++b) {
                            const int blk = bp[l] + b;
                            if (blk >= blo[l] && blk <= bhi[l]) {
                                const int u = wrap_once(ub[l] + 4 * b + q, ru);          // < 2 ru: a ring holds at least 16 units, a part 12
                                accp[l] = mfma3(fh[l][b], fl[l][b], ph[u * 16 + c16], pl[u * 16 + c16], accp[l]);
                            }
                        }
                        // every fragment register is read here, also those of the clamped (unused) requests: a load left pending
                        // would make hipcc wait for it wherever its register is next written
#pragma unroll
                        for (int b = 0; b < kPart; ++b) asm volatile("" :: "v"(fh[l][b].x), "v"(fl[l][b].x));
                        if (last && my_clip < a.batch) {
                            // D[row = 2 * bin + (re | im)][col = clip]: this lane holds bins 2q, 2q + 1 of the tile (re, im, re, im)
                            const int t = tn[l], j = jt[l];
                            float* o = a.out + my_clip * a.out_clip_stride + static_cast<long long>(t) * a.n_bins_total + L->k0 + kTileBins * j + 2 * q;
                            const bool live = t < tc;
                            const f32x4 acc = accp[l];
                            const float m0 = __logf(1.f + __builtin_amdgcn_sqrtf(acc[0] * acc[0] + acc[1] * acc[1]));
                            const float m1 = __logf(1.f + __builtin_amdgcn_sqrtf(acc[2] * acc[2] + acc[3] * acc[3]));
                            const int nb = L->n_bins;
                            if (kTileBins * j + 2 * q < nb) store_f32(o, live ? m0 : 0.f);
                            if (kTileBins * j + 2 * q + 1 < nb) store_f32(o + 1, live ? m1 : 0.f);
                        }
                    }
                    pend[l] = false;
                    if (!last) {
                        bp[l] += kPart;
                        ub[l] = wrap_once(ub[l] + 4 * kPart, rul[l]);
                        const int b_end = bp[l] + kPart < nblk[l] ? bp[l] + kPart : nblk[l];
                        kfire[l] = fire_step(l, au8[l] + 32 * b_end - 1);
                    } else {
                        LevelCP L = level_ptr(l);
                        ++tn[l];
                        pi[l] = pi[l] + 1 == L->period ? 0 : pi[l] + 1;
                        open_frame(l, L);
                    }
                }
                // the part that fires in the next step: request its fragments now (they land during the rest of this step and (a), (b) of the next)
                if (!pend[l] && jt[l] >= 0 && kfire[l] == k + 1 && !(a.dbg & 1)) { request(l); pend[l] = true; }
            }
            k_evt = 0x7fffffff;
#pragma unroll
            for (int l = 0; l < NL; ++l) {
                const int e = (pend[l] || jt[l] < 0) ? kfire[l] : kfire[l] - 1;
                k_evt = e < k_evt ? e : k_evt;
            }
            k_evt = k_evt > k ? k_evt : k + 1;
        }
        if (STAMP) ts[4] = stamp_now();
        // ---- advance the ring positions ----
        if (has_stage) {
            u_in = wrap_once(u_in + inc_in, ru_in);                     // (stage 3 runs on even k only; its position advances every step all the same)
            if (st + 1 < NL) u_out = wrap_once(u_out + inc_out, ru_out);
        }
        if (STAMP) {
            ts[5] = stamp_now();
#pragma unroll
            for (int i = 0; i < 5; ++i) sm[i] += ts[i + 1] - ts[i];
            sm[5] += 1;
        }
    }
    if (STAMP && w == 0 && lane == 0 && a.stamps) {
#pragma unroll
        for (int i = 0; i < 6; ++i) a.stamps[wave * 8 + i] = sm[i];
    }
}

}  // namespace fz

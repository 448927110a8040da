// Device kernels of the PitchClassNet forward pass (gfx950).  Included by pcnet.hip only.
//
// Everything here is fp32.  All convolutions of the net are "rows circular, time local":
//
//     out[co][y][t] = b[co] + sum_{ci,dy,dx} w[co][ci][dy][dx] * in[ci][(y + dy - py) mod H][t + dx - pad_l]
//
//   pitch conv 7x7 (models.py:230-232): H = pitches, py = 3, time wraps too (circular)
//   equivariant pitch-class conv (models.py:45-47): H = 12, KH = 12, py = 0, time zero-pad / valid
//   genre convs (models.py:724,733): KH = 1 / 2, py = 0, valid; the row H_out..H-1 is not stored
//
// so one kernel template (conv_mfma_kernel, f32-MFMA implicit GEMM, further down) covers them.  Small ops
// (semitone conv + octave fold, up_sixth, time pooling, head pooling) are plain vector kernels whose convolution
// weights are wave-uniform and therefore come in through SGPRs (s_load), never through VGPRs or LDS.
//
// Inference: BatchNorm (running statistics) is folded into w/b on the host, LeakyReLU is the epilogue.
// Training-mode forward (batch statistics): a convolution writes its RAW output and accumulates per-channel
// sum / sum-of-squares in its epilogue (`stats`); BatchNorm + LeakyReLU are applied by whoever reads the tensor
// next, as a per-input-channel (scale, shift, negative slope) triple applied while staging (`in_affine`), so the
// normalisation never costs a pass of its own over the activations.
#pragma once

#include <hip/hip_runtime.h>

namespace ake_k {

constexpr int TW = 4;              // frames per thread
constexpr float kSlope = 0.01f;    // nn.LeakyReLU() default (models.py:197)

struct ConvArgs {
    const float* src0;        // [B][c0][H][T_src]
    const float* src1;        // [B][c1][h1][T_src] rows repeated: row y reads y % h1 (PitchClass2Pitch, models.py:140-143)
    long long src0_clip_stride, src1_clip_stride;
    int c0, c1, h1;
    int H;                    // circular row count
    int T_in;                 // frames of the input
    int T_out;                // frames of the output
    int H_out;                // rows stored
    int py;                   // row anchor
    int pad_l;                // time anchor: input frame = t + dx - pad_l
    int time_circ;            // 1: frames wrap (padding_mode="circular"), 0: zeros outside
    const float* w;           // [groups][cin][KH][KW][CO]
    const float* bias;        // [groups*CO]
    int cout;
    float* dst;               // [B][dst_ctot][H_out][T_out]
    long long dst_clip_stride;
    int dst_coff;
    int lrelu;
    int R;                    // output rows per workgroup (ignored when FULLROWS)
    int TT;                   // output frames per workgroup (multiple of TW)
    int n_row_tiles, n_time_tiles;
    int Tp;                   // LDS row pitch in floats (multiple of 4, >= TT + KW - 1)
    const float* in_affine;   // [c0+c1][3] (scale, shift, negative slope) applied to every staged input value, or null
    double* stats;            // [kStatSlots][..][2] += (sum, sum of squares) of the raw outputs, or null
    int stats_stride;         // doubles between two slots of `stats` (workgroups spread their atomics over the slots)
    int accumulate;           // 1: dst += result (several data-gradients landing on one tensor)
    const float* residual;    // --resblock: [B][cout][H_out][T_out] added before the LeakyReLU (may be dst itself), or null
    long long residual_clip_stride;
    int rows_zero;            // 1: rows outside [0, H) read as zero instead of wrapping (--denseblock's zero-padded pitch convs; TRAIN form only)
};

// Per-channel batch statistics are accumulated with double atomics; thousands of workgroups hitting the same 16 addresses
// serialise in one L2 channel, so every producer adds into one of kStatSlots copies and bn_finalize_kernel sums them.
constexpr int kStatSlots = 64;

// ---- order-independent reductions -------------------------------------------------------------------------------------
// Every cross-workgroup sum of the training path (batch statistics, BatchNorm-backward sums, weight and bias gradients) is
// accumulated in 64-bit FIXED POINT: integer addition is associative, so the result does not depend on the order in which
// workgroups (or the threads of one) arrive, and two runs of a step are bit-identical.  (Float or double atomics give sums
// that differ in the last bits from run to run; through Adam's normalisation that is a +-lr difference in single weights.)
// The cells keep their old size: a `double` cell holds a long long.
constexpr double kFxStat = 16777216.0;            // 2^24: sums of activations and their squares (|sum| < 5.5e11, i.e. an rms of 300 over
                                                  // 5.6 M positions; resolution 6e-8 per partial sum = < 1e-9 of a variance; overflow -> NaN, below)
constexpr double kFxGrad = 1099511627776.0;       // 2^40: gradient-side sums, resolution 9.1e-13.  Trusted range |sum| < 2^61 / 2^40 = 2.1e6 (fx_checked below: beyond it
                                                  // the value read back is NaN -- a NaN gradient, which Adam propagates into the weights, is this path's overflow signal;
                                                  // the reference's float64 sums stay finite there, so a caller with raw gradients of that size scales the loss)
typedef long long gfx_t;                          // a gradient slot cell
__device__ __forceinline__ void fx_add(long long* cell, double v, double scale) {
    atomicAdd(reinterpret_cast<unsigned long long*>(cell), static_cast<unsigned long long>(__double2ll_rn(v * scale)));
}
__device__ __forceinline__ void fx_add(double* cell, double v, double scale) { fx_add(reinterpret_cast<long long*>(cell), v, scale); }
__device__ __forceinline__ double fx_get(const double* cell, double scale) { return static_cast<double>(*reinterpret_cast<const long long*>(cell)) / scale; }
__device__ __forceinline__ double fx_checked(long long sum, double scale) {     // a sum that came near the range cannot be trusted: fail loudly
    const long long lim = 1ll << 61;
    return (sum > lim || sum < -lim) ? __longlong_as_double(0x7ff8000000000000ll) : static_cast<double>(sum) / scale;
}

__device__ __forceinline__ void fx_add_int(double* cell, long long v) {
    atomicAdd(reinterpret_cast<unsigned long long*>(cell), static_cast<unsigned long long>(v));
}
// BatchNorm batch statistics, one lane's share, in DOUBLE: sum v and sum v^2 (the square of an f32 is exact in double, the sums are good to
// 1e-16 of their size).  Two reasons.  (1) A raw convolution output whose channel mean is many standard deviations from zero (non-negative
// inputs) made the one-pass E[z^2] - mean^2 on f32 partial sums lose 3.6e-5 of the variance at 256 clips.  (2) With f32 partials the LAST BIT
// of a BatchNorm table depended on which elements a lane happened to see, i.e. on the tiling the batch size selects: of the 4e7 LeakyReLU /
// max decisions of a step a handful then went the other way between a 192- and a 256-clip batch of the same clips, and because a gradient
// tensor is a cancelling sum over ~1e6 positions ONE flipped position moves it by 1e-3 (tests/test_gpu_train_scale.py).  Now the partials are
// exact to rounding of the total, the fixed-point cells add integers, and the tables are the same bits whatever the tiling.
struct ShiftStat {
    double s1, s2;
    __device__ __forceinline__ void init() { s1 = 0.0; s2 = 0.0; }
    __device__ __forceinline__ void add(float v) {
        const double d = v;
        s1 += d;
        s2 = fma(d, d, s2);
    }
    __device__ __forceinline__ void fixed(long long& i1, long long& i2) const {
        i1 = __double2ll_rn(s1 * kFxStat);
        i2 = __double2ll_rn(s2 * kFxStat);
    }
};
__device__ __forceinline__ long long shfl_xor_ll(long long v, int o) {
    const int lo = __shfl_xor(static_cast<int>(v & 0xffffffffll), o), hi = __shfl_xor(static_cast<int>(v >> 32), o);
    return (static_cast<long long>(hi) << 32) | static_cast<unsigned int>(lo);
}

// float -> bf16 bits, round to nearest even (finite inputs)
__device__ __forceinline__ unsigned int bf16_bits(float v) {
    const unsigned int u = __float_as_uint(v);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}

__device__ __forceinline__ int wrap(int i, int n) {
    i %= n;
    return i < 0 ? i + n : i;
}

// ------------------------------------------------------------------------------------------
// Semitone conv (3x3, stride (3,1), time circular; models.py:313-315 / 337-339) + BN + LeakyReLU
// fused with the pitch -> pitch-class fold (dilated max over octaves, models.py:95-106).
// thread = (clip, CO channels, pitch class p, 4 frames); loops over the octaves.
// ------------------------------------------------------------------------------------------
struct SemiArgs {
    const float* src;   // [B][C][H][T]
    long long src_clip_stride;
    int C, H, T;
    const float* w;     // [groups][C][3][3][CO]
    const float* bias;
    float* dst;         // [B][dst_ctot][12][T]
    long long dst_clip_stride;
    int dst_coff;
    int n_strips;       // ceil(T / TW)
};

template <int CO>
__global__ void semi_fold_kernel(SemiArgs a) {
    const int item = blockIdx.x * blockDim.x + threadIdx.x;
    const int per_clip = 12 * a.n_strips;
    if (item >= per_clip) return;
    const int p = item / a.n_strips;
    const int s = item - p * a.n_strips;
    const int grp = blockIdx.y;
    const int clip = blockIdx.z;
    const int t0 = s * TW;
    int tix[TW + 2];
#pragma unroll
    for (int j = 0; j < TW + 2; ++j) tix[j] = wrap(t0 - 1 + j, a.T);
    const float* src = a.src + clip * a.src_clip_stride;
    const float* __restrict__ wg = a.w + static_cast<long long>(grp) * a.C * (9 * CO);
    float best[CO][TW];
#pragma unroll
    for (int co = 0; co < CO; ++co)
#pragma unroll
        for (int j = 0; j < TW; ++j) best[co][j] = -INFINITY;
    const int n_oct = a.H / 36;
    for (int o = 0; o < n_oct; ++o) {
        const int row0 = 3 * (p + 12 * o);
        float acc[CO][TW];
#pragma unroll
        for (int co = 0; co < CO; ++co)
#pragma unroll
            for (int j = 0; j < TW; ++j) acc[co][j] = 0.f;
        for (int ci = 0; ci < a.C; ++ci) {
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const float* rowp = src + (static_cast<long long>(ci) * a.H + row0 + dy) * a.T;
                float in[TW + 2];
#pragma unroll
                for (int j = 0; j < TW + 2; ++j) in[j] = rowp[tix[j]];
                const float* __restrict__ wp = wg + (ci * 3 + dy) * (3 * CO);
#pragma unroll
                for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                    for (int co = 0; co < CO; ++co) {
                        const float wv = wp[dx * CO + co];
#pragma unroll
                        for (int j = 0; j < TW; ++j) acc[co][j] = fmaf(in[j + dx], wv, acc[co][j]);
                    }
            }
        }
#pragma unroll
        for (int co = 0; co < CO; ++co) {
            const int c = grp * CO + co;
            const float b = c < a.C ? a.bias[c] : 0.f;
#pragma unroll
            for (int j = 0; j < TW; ++j) {
                float v = acc[co][j] + b;
                v = v > 0.f ? v : v * kSlope;
                best[co][j] = fmaxf(best[co][j], v);
            }
        }
    }
    float* d = a.dst + clip * a.dst_clip_stride;
#pragma unroll
    for (int co = 0; co < CO; ++co) {
        const int c = grp * CO + co;
        if (c >= a.C) break;
        float* drow = d + (static_cast<long long>(a.dst_coff + c) * 12 + p) * a.T;
#pragma unroll
        for (int j = 0; j < TW; ++j)
            if (t0 + j < a.T) drow[t0 + j] = best[co][j];
    }
}

// ------------------------------------------------------------------------------------------
// up_sixth: ConvTranspose2d(C, C, (3,1), stride (3,1)) + BN + LeakyReLU (models.py:325-327):
// out[co][3p+j][t] = lrelu(b[co] + sum_ci in[ci][p][t] * w[ci][co][j])
// ------------------------------------------------------------------------------------------
__global__ void up_sixth_kernel(const float* __restrict__ src, long long src_clip_stride, const float* __restrict__ w,
                                const float* __restrict__ bias, float* __restrict__ dst, int C, int T, long long total) {
    const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int t = static_cast<int>(i % T);
    long long q = i / T;
    const int row = static_cast<int>(q % 36);
    q /= 36;
    const int co = static_cast<int>(q % C);
    const long long clip = q / C;
    const int p = row / 3, j = row - 3 * p;
    const float* s = src + clip * src_clip_stride + static_cast<long long>(p) * T + t;   // src may be a channel slice of a concat buffer
    float acc = bias[co];
    for (int ci = 0; ci < C; ++ci) acc = fmaf(s[static_cast<long long>(ci) * 12 * T], w[(ci * C + co) * 3 + j], acc);
    dst[i] = acc > 0.f ? acc : acc * kSlope;
}

// nn.MaxPool2d((1, tp)) (models.py:349-350): [B][C][H][T] -> [B][dst_ctot][H][T/tp] at channel dst_coff
__global__ void time_pool_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, int H, int T, int tp,
                                 int dst_ctot, int dst_coff, long long total) {
    const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int To = T / tp;
    const int t = static_cast<int>(i % To);
    long long q = i / To;
    const int y = static_cast<int>(q % H);
    q /= H;
    const int c = static_cast<int>(q % C);
    const long long clip = q / C;
    const float* s = src + ((clip * C + c) * H + y) * T + static_cast<long long>(t) * tp;
    float m = s[0];
    for (int j = 1; j < tp; ++j) m = fmaxf(m, s[j]);
    dst[((clip * dst_ctot + dst_coff + c) * H + y) * To + t] = m;
}

// ------------------------------------------------------------------------------------------
// Masked temporal mean over the head maps + sigmoid on key (models.py:754-804).
// maps: [B][rows][Tm]; one thread per (clip, row).
// ------------------------------------------------------------------------------------------
struct PoolHeadArgs {
    const float* maps[3];     // key, tonic, genre (genre may be null)
    float* outs[3];
    int rows[3];              // 12, 12, 11
    int Tm;                   // frames of the maps
    const long long* seq;     // [B] or null
    int n_pool_layers;        // num_layers - 1
    int tp;                   // time_pool_size
    int shrink;               // (kernel_size - 1) * head_layers
    int max_pool;
    int batch;
    int clip0;                // global index of the first clip of this chunk (for the max_pool sample-0 quirk)
};

// frames pooled for `clip` (models.py:754-785) and whether the maximum is taken instead of the mean
__device__ __forceinline__ int pool_frames(const long long* seq, int clip, int Tm, int n_pool_layers, int tp, int shrink, int clip0, bool* use_max) {
    int L = Tm;
    if (seq) {
        long long l = seq[clip];
        for (int k = 0; k < n_pool_layers; ++k) l = l / tp;          // floor (models.py:759)
        L = static_cast<int>(l) - shrink;                             // models.py:760
        if (L > Tm) L = Tm;                                           // x[..., :L] clamps at the end ...
        if (L < 0) L = Tm + L > 0 ? Tm + L : 0;                       // ... and counts from the end when negative
        *use_max = *use_max && (clip0 + clip == 0);                   // models.py:764-785 quirk
    }
    return L;
}

__global__ void head_pool_kernel(PoolHeadArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int which = blockIdx.y;
    if (a.maps[which] == nullptr) return;
    const int rows = a.rows[which];
    if (i >= a.batch * rows) return;
    const int clip = i / rows;
    bool use_max = a.max_pool != 0;
    const int L = pool_frames(a.seq, clip, a.Tm, a.n_pool_layers, a.tp, a.shrink, a.clip0, &use_max);
    const float* m = a.maps[which] + static_cast<long long>(i) * a.Tm;
    float v;
    if (use_max) {
        v = -INFINITY;
        for (int t = 0; t < L; ++t) v = fmaxf(v, m[t]);
    } else {
        float sum = 0.f;
        for (int t = 0; t < L; ++t) sum += m[t];
        v = sum / static_cast<float>(L > 0 ? L : 0);                  // empty slice -> NaN, as torch.mean
    }
    if (which == 0) v = 1.f / (1.f + expf(-v));                       // self.sig(key_out), models.py:802
    a.outs[which][i] = v;
}


// --denseblock: the pitch stack's input (pitch stream | up_sixth map repeated over the octaves, models.py:378-383) materialised as
// channels [0, c0 + c1) of the block's feature buffer [B][ctot][H][T] -- every later dense layer re-reads it with its own BatchNorm
__global__ void concat_repeat_kernel(const float* __restrict__ src0, int c0, const float* __restrict__ src1, int c1, int h1, float* __restrict__ dst,
                                     int ctot, int H, int T, long long total, const float* __restrict__ aff1 = nullptr) {   // aff1 (training): src1 is raw, its [c1][3] table applied here
    const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int t = static_cast<int>(i % T);
    long long r = i / T;
    const int y = static_cast<int>(r % H); r /= H;
    const int c = static_cast<int>(r % (c0 + c1));
    const long long clip = r / (c0 + c1);
    float v = c < c0 ? src0[((clip * c0 + c) * H + y) * T + t] : src1[((clip * c1 + (c - c0)) * h1 + y % h1) * T + t];
    if (aff1 && c >= c0) { const float x = fmaf(v, aff1[3 * (c - c0)], aff1[3 * (c - c0) + 1]); v = x > 0.f ? x : x * aff1[3 * (c - c0) + 2]; }
    dst[((clip * ctot + c) * H + y) * T + t] = v;
}

// --denseblock: eval-mode BatchNorm in front of a convolution cannot be folded into weights shared by several consumers; it becomes a
// per-channel (scale, shift, negative slope) row that the convolution applies while loading.  idx = [n][5]: flat-parameter offsets of
// gamma, beta, running_mean, running_var, and the slope's bits (LeakyReLU 0.01 / ReLU 0)
__global__ void dense_affine_kernel(const float* __restrict__ params, const int* __restrict__ idx, float* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int* o = idx + 5 * i;
    const double sc = static_cast<double>(params[o[0]]) / sqrt(static_cast<double>(params[o[3]]) + 1e-5);
    out[3 * i] = static_cast<float>(sc);
    out[3 * i + 1] = static_cast<float>(static_cast<double>(params[o[1]]) - static_cast<double>(params[o[2]]) * sc);
    out[3 * i + 2] = __int_as_float(o[4]);
}

// ==========================================================================================
// f32-MFMA implicit-GEMM form of the same "rows circular, time local" convolution.
//
//   D[m][n] = sum_k A[m][k] * B[k][n]          v_mfma_f32_16x16x4_f32 (exact fp32: a k-ordered fmaf chain)
//
//   m = output position (row y, frame group j)         16 positions per MFMA tile
//   n = (output channel co, frame offset tau < TB)     N' = Cout*TB columns, 16 per N-tile
//   k = (ci, dy, u < KU)                               A[m][k] = in[ci][y+dy-py][TB*j + u - pad]
//   B[k][n] = w[co][ci][dy][u - tau]  if 0 <= u - tau < 7 else 0      (Toeplitz in time)
//
// The Toeplitz expansion lets narrow layers fill the 16-wide N dimension: the 7x7 pitch conv has only
// 8 output channels, so TB = 2 gives 8 x 2 = 16 columns and KU = 8 taps per (ci,dy) -- 7/8 of every MFMA is
// useful work and nothing is padded.  TB = 1 for 16/32 channels, 4 for 4, 16 for the 1-channel head convs.
//
// A comes from the same LDS patch as the VALU kernel (one f32 per lane per MFMA: lane (r,q) reads
// patch[row_m + dy][TB*j_m + 4*s + q], conflict-free), B is pre-packed on the host in fragment order
// [ci][dy][s][ntile][lane] and streamed from L2 with coalesced 256-byte loads, one k-step ahead.
// ==========================================================================================
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct MfmaArgs {
    ConvArgs c;           // geometry, sources, destination (c.w = packed B fragments, c.bias = per-channel bias)
    int TB;               // frames per column group
    int KH;               // kernel rows
    int ntiles_total;     // N-tiles of the whole layer (fragment stride)
    int cin_chunk;        // input channels staged in LDS at a time
    int h1_magic;         // ceil(2^16 / h1): (row * h1_magic) >> 16 == row / h1 for the row counts used here
    int dbg;              // ablation (AKE_ABLATE): 1 = skip the MFMA steps, 2 = skip the staging loads (results wrong, timing only)
    int row_k;            // 1: 1-wide kernels over KH = 4 * (steps per channel) circular rows (--denseblock's 12 x 1 bottleneck): the four k
                          //    of a step are four consecutive ROWS of one frame instead of four taps of one row; `KH` counts steps
    int ksplit;           // 1: the layer has <= MT M-tiles per workgroup (1-channel head convs): all waves share them and
                          //    split the (channel, dy) steps of every chunk (step = wave, wave + W, ...); partial sums are
                          //    reduced through LDS
};

template <int KU, int NT, int MT, bool TRAIN = false>   // TRAIN: in_affine on load, statistics epilogue, accumulate (training fwd + data gradients)
__global__ __launch_bounds__(512) void conv_mfma_kernel(MfmaArgs ma) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const ConvArgs& a = ma.c;
    const int TB = ma.TB, KH = ma.KH;
    const int cin = a.c0 + a.c1;
    const int tile = blockIdx.x;
    const int row_tile = tile / a.n_time_tiles;
    const int time_tile = tile - row_tile * a.n_time_tiles;
    const int ngrp = blockIdx.y;                 // group of NT N-tiles
    const int clip = blockIdx.z;
    const int y0 = row_tile * a.R;
    const int t0 = time_tile * a.TT;
    const int R_in = a.R + (ma.row_k ? 4 * KH : KH) - 1;   // patch rows: the row halo is always materialised (wrapped rows are copied)
    const int Tp = a.Tp;
    const int rstep = ma.row_k ? 4 * Tp : Tp;    // LDS floats from one step of a channel to the next

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    const int r16 = lane & 15, q = lane >> 4;
    const int rows_here = a.H - y0 < a.R ? a.H - y0 : a.R;
    const int tt_here = a.T_out - t0 < a.TT ? a.T_out - t0 : a.TT;   // frames of this time tile
    const int J = (tt_here + TB - 1) / TB;                             // frame groups per row
    const int Mblk = rows_here * J;
    const int mtiles = (Mblk + 15) / 16;
    const int grp = ma.ksplit ? 0 : wave;                              // M-tile group of this wave
    const bool active = grp * MT < mtiles;
    const int cstride = R_in * Tp;
    constexpr int KS = KU / 4;                                          // k-steps per (ci,dy)

    // per-lane LDS float index of A[m][u = q] at (ci = 0, dy = 0) for each of the wave's MT tiles
    int abase[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int m = (grp * MT + mt) * 16 + r16;
        if (m >= Mblk) m = Mblk - 1;
        const int r = m / J, j = m - r * J;
        abase[mt] = r * Tp + TB * j + (ma.row_k ? q * Tp : q);
    }
    // two independent accumulator chains per tile (even / odd k-steps): back-to-back MFMAs never depend on each other
    f32x4 acc[MT][NT], acc2[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) { acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f}; acc2[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    // B fragments in global memory: [ci][dy][s][ntile][64 lanes]; the block's NT tiles of a chunk are staged in LDS
    // behind the A patch as [cl][dy][s][nt][64] so that no wave ever waits on an L2 round trip inside the MFMA loop
    const float* __restrict__ bglob = a.w + (static_cast<long long>(ngrp) * NT) * 64;
    const int bstep = ma.ntiles_total * 64;                              // floats per k-step in global memory
    float* const ldsB = lds + ma.cin_chunk * cstride;
    const int sstride = ma.ksplit ? nw : 1;                              // this wave takes every sstride-th step of a chunk

    auto load_a = [&](float (&A)[KS][MT], int soff) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int s = 0; s < KS; ++s) A[s][mt] = lds[abase[mt] + soff + 4 * s];
    };
    auto load_b = [&](float (&B)[KS][NT], int boff) {
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) B[s][nt] = ldsB[boff + (s * NT + nt) * 64 + lane];
    };
    auto mma = [&](const float (&A)[KS][MT], const float (&B)[KS][NT]) {
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    if (s & 1) acc2[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[s][mt], B[s][nt], acc2[mt][nt], 0, 0, 0);
                    else acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[s][mt], B[s][nt], acc[mt][nt], 0, 0, 0);
                }
    };

    const float* s0 = a.src0 + clip * a.src0_clip_stride;
    const float* s1 = a.src1 ? a.src1 + clip * a.src1_clip_stride : nullptr;
    const int ncb = (Tp + 63) >> 6;                                     // 64-lane column blocks per patch row (<= 3)
    const bool fast_wrap = a.T_in >= Tp;                                // one conditional add/sub resolves the circular frame index
    for (int c_lo = 0; c_lo < cin; c_lo += ma.cin_chunk) {
        const int cc = cin - c_lo < ma.cin_chunk ? cin - c_lo : ma.cin_chunk;
        if (c_lo) __syncthreads();
        // ---- stage channels [c_lo, c_lo+cc): rows and frames with both halos resolved.  One wave per patch row (row
        //      arithmetic is wave-uniform), lanes walk consecutive frames (coalesced); 4 rows x <=3 column blocks of
        //      loads are in flight before the first LDS write.
        if (ma.dbg != 2) {
            const int nrows = cc * R_in;
            constexpr int UR = 4;
            // (cl, rj) of patch row `wave`, advanced by nw rows per step without any division
            int cl_it = 0, rj_it = wave;
            while (rj_it >= R_in) { rj_it -= R_in; ++cl_it; }
            for (int base = wave; base < nrows; base += UR * nw) {
                float v[UR][3];
#pragma unroll
                for (int u = 0; u < UR; ++u) {
                    const bool live = base + u * nw < nrows;
                    const int cl = live ? cl_it : 0, rj = live ? rj_it : 0;
                    rj_it += nw;
                    while (rj_it >= R_in) { rj_it -= R_in; ++cl_it; }
                    const int cs = c_lo + cl;
                    int row = y0 - a.py + rj;                            // in (-H, 2H): one conditional wrap each way
                    const bool row_ok = !(TRAIN && a.rows_zero) || (row >= 0 && row < a.H);   // zero padding: the row contributes nothing
                    row += row < 0 ? a.H : 0;
                    row -= row >= a.H ? a.H : 0;
                    float asc = 1.f, ash = 0.f, ang = 1.f;                // BatchNorm + LeakyReLU of the producer, applied on load
                    if (TRAIN && a.in_affine) { asc = a.in_affine[3 * cs]; ash = a.in_affine[3 * cs + 1]; ang = a.in_affine[3 * cs + 2]; }
                    const float* srow;
                    if (cs < a.c0) srow = s0 + (static_cast<long long>(cs) * a.H + row) * a.T_in;
                    else {
                        const int r1 = row - ((row * ma.h1_magic) >> 16) * a.h1;     // row % h1 (PitchClass2Pitch repeat)
                        srow = s1 + (static_cast<long long>(cs - a.c0) * a.h1 + r1) * a.T_in;
                    }
#pragma unroll
                    for (int h = 0; h < 3; ++h) {
                        v[u][h] = 0.f;
                        const int tj = lane + 64 * h;
                        if (h < ncb && tj < Tp) {
                            int ti = t0 - a.pad_l + tj;
                            bool ok = true;
                            if (a.time_circ) {
                                if (fast_wrap) { ti += ti < 0 ? a.T_in : 0; ti -= ti >= a.T_in ? a.T_in : 0; }
                                else ti = wrap(ti, a.T_in);
                            } else {
                                ok = ti >= 0 && ti < a.T_in;
                            }
                            if (ok && row_ok) {
                                if (TRAIN) {
                                    const float x = fmaf(srow[ti], asc, ash);
                                    v[u][h] = x > 0.f ? x : x * ang;      // zero padding stays zero: it pads the activation
                                } else {
                                    v[u][h] = srow[ti];
                                }
                            }
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < UR; ++u) {
                    const int rr = base + u * nw;
                    if (rr < nrows) {
#pragma unroll
                        for (int h = 0; h < 3; ++h) {
                            const int tj = lane + 64 * h;
                            if (h < ncb && tj < Tp) lds[rr * Tp + tj] = v[u][h];
                        }
                    }
                }
            }
            // B fragments of the chunk: cc*KH k-step groups of KS*NT*64 floats (contiguous in LDS, strided in global)
            constexpr int GF = KS * NT * 64;
            const int ngroups = cc * KH;
            const float* bsrc = bglob + static_cast<long long>(c_lo) * KH * KS * bstep;
            for (int g = wave; g < ngroups; g += nw) {
#pragma unroll
                for (int e = 0; e < KS * NT; ++e) {
                    const int sidx = e / NT, nt = e - sidx * NT;
                    ldsB[g * GF + e * 64 + lane] = bsrc[(static_cast<long long>(g) * KS + sidx) * bstep + nt * 64 + lane];
                }
            }
        }
        __syncthreads();
        if (!active || ma.dbg == 1) continue;
        // ---- this wave's steps of the chunk: step index -> (channel cl, row dy); fragments are fetched one step ahead.
        //      The cursor is wave-uniform and advanced with adds only: a_off / b_off are LDS float offsets. ----
        constexpr int GF = KS * NT * 64;
        const int steps_total = cc * KH;
        const int first = ma.ksplit ? wave : 0;
        if (first >= steps_total) continue;
        const int nsteps = (steps_total - first + sstride - 1) / sstride;
        int pf_dy = first, pf_cl = 0;
        while (pf_dy >= KH) { pf_dy -= KH; ++pf_cl; }
        int a_off = pf_cl * cstride + pf_dy * rstep, b_off = first * GF;
        float Ac[KS][MT], Bc[KS][NT];
        load_a(Ac, a_off);
        load_b(Bc, b_off);
        for (int st = 0; st < nsteps; ++st) {
            if (st + 1 < nsteps) {                                       // (the last iteration re-fetches its own step)
                pf_dy += sstride;
                a_off += sstride * rstep;
                b_off += sstride * GF;
                while (pf_dy >= KH) { pf_dy -= KH; a_off += cstride - KH * rstep; }
            }
            float An[KS][MT], Bn[KS][NT];
            load_a(An, a_off);
            load_b(Bn, b_off);
            mma(Ac, Bc);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) Ac[s][mt] = An[s][mt];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) Bc[s][nt] = Bn[s][nt];
            }
        }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] += acc2[mt][nt];

    if (ma.ksplit) {   // sum the waves' partial accumulators through LDS (the patch is dead now); wave 0 stores
        __syncthreads();
        f32x4* red = reinterpret_cast<f32x4*>(lds);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) red[((wave * MT + mt) * NT + nt) * 64 + lane] = acc[mt][nt];
        __syncthreads();
        if (wave != 0) return;
        for (int w = 1; w < nw; ++w)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[mt][nt] += red[((w * MT + mt) * NT + nt) * 64 + lane];
    }
    if (!active) return;

    // ---- epilogue: D[row = 4*(lane>>4) + reg][col = lane&15]; bias (BN folded), LeakyReLU.
    //      One division per M-tile: positions 4q..4q+3 of a tile are consecutive frame groups (j, j+1, ..) of row r,
    //      carried into the next row when j reaches J. ----
    float* d = a.dst + clip * a.dst_clip_stride + t0;
    const float* rs = a.residual ? a.residual + clip * a.residual_clip_stride + t0 : nullptr;
    const int row_elems = a.T_out;
    ShiftStat stt[TRAIN ? NT : 1];
#pragma unroll
    for (int nt = 0; nt < (TRAIN ? NT : 1); ++nt) stt[nt].init();
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m0 = (grp * MT + mt) * 16 + 4 * q;
        int r = m0 / J, j = m0 - r * J;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const bool ok_m = m0 + reg < Mblk && y0 + r < a.H_out;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int n = (ngrp * NT + nt) * 16 + r16;
                const int co = n / TB, tau = n - co * TB;        // TB is a power of two in practice; wave-invariant per lane
                const int tl = TB * j + tau;
                if (ok_m && co < a.cout && tl < tt_here) {
                    float v = acc[mt][nt][reg] + a.bias[co];
                    if (rs) v += rs[(co * a.H_out + (y0 + r)) * row_elems + tl];
                    if (TRAIN && a.stats) stt[TRAIN ? nt : 0].add(v);
                    if (a.lrelu) v = v > 0.f ? v : v * kSlope;
                    float* dp = d + ((a.dst_coff + co) * a.H_out + (y0 + r)) * row_elems + tl;
                    *dp = (TRAIN && a.accumulate) ? *dp + v : v;
                }
            }
            if (++j == J) { j = 0; ++r; }
        }
    }
    if (TRAIN && a.stats) {   // per-channel batch statistics: lanes of one channel = TB adjacent columns x 4 row groups
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            long long s1, s2;
            stt[TRAIN ? nt : 0].fixed(s1, s2);
            s1 += shfl_xor_ll(s1, 16); s2 += shfl_xor_ll(s2, 16);
            s1 += shfl_xor_ll(s1, 32); s2 += shfl_xor_ll(s2, 32);
            for (int o = 1; o < TB; o <<= 1) { s1 += shfl_xor_ll(s1, o); s2 += shfl_xor_ll(s2, o); }
            const int n = (ngrp * NT + nt) * 16 + r16;
            const int co = n / TB;
            if (q == 0 && n - co * TB == 0 && co < a.cout) {
                double* st = a.stats + static_cast<size_t>((blockIdx.x + 7 * blockIdx.z + wave) & (kStatSlots - 1)) * a.stats_stride;
                fx_add_int(st + 2 * co, s1);
                fx_add_int(st + 2 * co + 1, s2);
            }
        }
    }
}

// ==========================================================================================
// 7x7 circular pitch convolution, 8 -> 8 channels, on v_mfma_f32_16x16x32_f16: f16 activations x f16 weights, f32 accumulation.
//
// The f32 MFMA above runs at the vector rate; this form does the same contraction with 16x fewer matrix-pipe cycles.
//   * activations: ONE f16 value (11 significant bits, rounding 2^-12 relative, unbiased), channels-last;
//   * weights: ONE f16 value each (kP2pProducts == 1), after scaling every output channel by a power of two so that its largest weight
//     sits in [2^13, 2^14) -- whatever BatchNorm folded into it, nothing overflows or goes subnormal; the scale is undone by the
//     epilogue's fma.  kP2pProducts == 2 adds the weights' second half (wl = rn((w - wh) * 2^11), products in an accumulator of their
//     own): weights exact to 2^-22.
//   What the roundings cost (tests/tools/split_precision_proto.py: the oracle's float64 forward with the rounding applied to these
//   three convolutions, golden and sparse inputs; measured again end to end on the bench line): outputs move by 1.5e-6 of their range
//   with exact weights and by 1.0e-5 .. 1.3e-5 with f16 weights (bench: max_rel_err 9.3e-6 -> 1.8e-5, budget 1e-3) -- the
//   split-bf16 pair this replaces (hi + lo on BOTH operands, 3 products) cost 2.8e-6.  392 products with independent rounding errors
//   are summed per output, and behind these convolutions sit an octave max, 1344-term pitch-class convolutions and a temporal mean.
//   The same prototype shows that the layer-1 pitch-class stack and the heads are NOT that tolerant (3e-5 .. 5e-4 with f16
//   activations): they keep the three-product split-bf16 form.
//   Why it pays: on gfx950 a SIMD does not issue VALU work while its matrix pipe executes an MFMA, not even another wave's
//   (tools/micro/mfma_valu_overlap.hip: 116 + 95 cycles alone, 210 together), so a tile costs MFMA cycles + VALU cycles + LDS waits.
//   In-kernel stamps per tile (cycles): split-bf16 x3 ~7000 (A fragments of two planes: 5376 LDS cycles > 4032 MFMA cycles);
//   f16 x (hi + lo) 4650 = 3260 MFMA + 1040 epilogue + 350 barrier; f16 x f16 3700 (now the A-fragment reads bound the loop: 336
//   ds_read_b128 = 2688 LDS cycles against 1630 MFMA cycles).
//   Range: activations beyond +-65504 saturate (MODE.FP16_OVFL is set, so nothing turns into inf); the net's activations sit behind
//   BatchNorm + LeakyReLU and the log-magnitude CQT, orders of magnitude below that.
// What makes it cheap is the layout: activations are CHANNELS-LAST, [clip][row][frame][8 ch], so the 8
// consecutive k of one MFMA lane are the 8 input channels of one tap = one aligned 16-byte LDS read, with no Toeplitz
// gather on the A side:
//   m = (row r, frame pair j)            A[m][k = (tap q' of the k-step, ci)] = X[r + dy][2j + 4h + q'][ci]
//   n = (tau, co) = 8*tau + co           B[k][n] = w[co][ci][dy][4h + q' - tau]   (zero outside the 7 taps)
//   k-step = (dy, h):  7 x 2 = 14 steps of K = 32, two MFMAs each per M-tile.
// One workgroup = R rows x all frames of one clip (8 waves x 3 M-tiles), patch in LDS.
// Output: NCHW f32 (for the semitone pooling that follows the stack) or the channels-last f16 plane (next conv).
// ==========================================================================================
struct P2pBfArgs {
    const unsigned short* xh;     // [clip][H][T][8] f16   (IN_NCHW == false)
    // IN_NCHW == true (first convolution of a stack): the input is assembled while staging, channels [0, c0) from the pitch stream
    // p [clip][c0][H][T], channels [c0, c0 + c1) from the up_sixth output u [clip][c1][h1][T] repeated over the octaves (row % h1;
    // models.py:140-143, 378-383), the rest zero -- the concatenated tensor never exists in memory
    const float* p;
    const float* u;
    int c0, c1, h1;
    const uint4* bfrag;           // [14 k-steps][hi|lo * 2^11][64 lanes] x 8 f16
    const float* bias;            // [8] (BatchNorm folded)
    float* dst;                   // NCHW f32 [clip][dst_ctot][H][T] (OUT_CL == false)
    long long dst_clip_stride;
    int dst_coff;
    unsigned short* oh;           // channels-last f16 plane (OUT_CL == true)
    int H, T, R, J, Tp, n_row_tiles;
};

typedef __bf16 bf16x8c __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8c __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2c __attribute__((ext_vector_type(2)));
constexpr int kP2pProducts = 1;                                      // 1: f16 weights;  2: f16 hi + f16 lo weights (see above)
constexpr float kP2pLoScale = 2048.f, kP2pLoInv = 1.f / 2048.f;     // the lo weight plane is stored times 2^11 (kept normal in f16)
constexpr int kP2pFragScale = 14 * 2 * 64;                           // uint4 index of the 8 per-channel inverse scales behind a conv's fragments
// v_cvt_f16_f32 saturates to +-65504 instead of producing inf (MODE.FP16_OVFL, bit 23)
__device__ __forceinline__ void f16_saturate_mode() { asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1"); }
// power of two s with s * wmax in [2^13, 2^14) (1 for an all-zero channel): the channel's weights use f16's normal range whatever
// BatchNorm folded into them
__device__ __forceinline__ float f16_weight_scale(float wmax) {
    if (!(wmax > 0.f) || !(wmax < INFINITY)) return 1.f;
    int e;
    frexpf(wmax, &e);                                         // wmax = m * 2^e, m in [0.5, 1)
    return ldexpf(1.f, 14 - e < 100 ? 14 - e : 100);          // (a channel of f32 denormals: the scale itself must stay finite)
}
// max |dz| of a tensor: bn_bwd_apply_kernel leaves it as kAmaxSlots partial maxima (float bits); every lane of a wave gets the maximum
constexpr int kAmaxSlots = 64;
__device__ __forceinline__ float amax_load(const unsigned int* cells) {
    float m = __uint_as_float(cells[threadIdx.x & (kAmaxSlots - 1)]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    return m;
}
__device__ __forceinline__ unsigned int f16_bits(float v) { return __builtin_bit_cast(unsigned short, static_cast<_Float16>(v)); }
constexpr bool kP2pStreamB = true;
constexpr int kP2pMT = 3;            // M-tiles (16 positions) per wave
// a pitch conv with <= 5 input channels (the first of a stack: raw pitch stream | up_sixth channels) carries channel 0 as a split operand in
// its idle slots 5 and 6 (pack_p2p_f16_kernel): the raw log-CQT is the one activation the trained filters difference against itself
constexpr int kP2pSplit0 = 5;
// a wave-uniform pointer, told to the compiler (scalar registers): loads through it take the saddr + 32-bit lane offset form
template <typename T>
__device__ __forceinline__ const T* uniform_ptr(const T* p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned int lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned int>(v)), hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned int>(v >> 32));
    return reinterpret_cast<const T*>((static_cast<unsigned long long>(hi) << 32) | lo);
}

template <bool OUT_CL, bool IN_NCHW>
__global__ __launch_bounds__(512) void conv_p2p_f16_kernel(P2pBfArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint4 lds4[];
    const int clip = blockIdx.z;
    const int y0 = blockIdx.x * a.R;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r16 = lane & 15, q = lane >> 4;
    const int R_in = a.R + 6, Tp = a.Tp, J = a.J;
    const int rows_here = a.H - y0 < a.R ? a.H - y0 : a.R;
    const int Mblk = rows_here * J;
    f16_saturate_mode();
    uint4* const pH = lds4;                                  // [R_in][Tp] positions, 16 B each
    uint4* const pB = lds4 + R_in * Tp;                      // [14][2][64]
    // ---- stage: patch (both circular halos resolved) and the weight fragments ----
    {
        const long long cbase = static_cast<long long>(clip) * a.H * a.T;
        const uint4* gh = reinterpret_cast<const uint4*>(a.xh) + cbase;
        const int npos = R_in * Tp;
        for (int i = threadIdx.x; i < npos; i += blockDim.x) {
            const int rj = i / Tp, f = i - rj * Tp;
            int row = y0 - 3 + rj;
            row += row < 0 ? a.H : 0;
            row -= row >= a.H ? a.H : 0;
            const int t = wrap(f - 3, a.T);
            if (IN_NCHW) {
                unsigned int hi[4] = {0, 0, 0, 0};
                const int ru = row % a.h1;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    if (c < a.c0 + a.c1) {
                        const float v = c < a.c0 ? a.p[((static_cast<long long>(clip) * a.c0 + c) * a.H + row) * a.T + t]
                                                 : a.u[((static_cast<long long>(clip) * a.c1 + (c - a.c0)) * a.h1 + ru) * a.T + t];
                        hi[c >> 1] |= f16_bits(v) << (16 * (c & 1));
                        if (c == 0 && a.c0 + a.c1 <= kP2pSplit0) {               // channel 0 again: its low half (slot 5), its high half (slot 6)
                            const _Float16 h0 = static_cast<_Float16>(v);
                            hi[2] |= f16_bits(v - static_cast<float>(h0)) << 16;
                            hi[3] |= static_cast<unsigned int>(__builtin_bit_cast(unsigned short, h0));
                        }
                    }
                }
                pH[i] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
            } else {
                const long long g = static_cast<long long>(row) * a.T + t;
                pH[i] = gh[g];
            }
        }
        if (!kP2pStreamB)
            for (int i = threadIdx.x; i < 14 * 2 * 64; i += blockDim.x) pB[i] = a.bfrag[i];
    }
    __syncthreads();
    constexpr int MT = kP2pMT;
    if (wave * MT * 16 >= Mblk) return;
    int abase[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int m = (wave * MT + mt) * 16 + r16;
        if (m >= Mblk) m = Mblk - 1;
        const int r = m / J, j = m - r * J;
        abase[mt] = r * Tp + 2 * j + q;
    }
    typedef float f32x4c __attribute__((ext_vector_type(4)));
    f32x4c acc[MT], accl[MT];       // products with the hi / the (scaled) lo weight plane
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) { acc[mt] = f32x4c{0.f, 0.f, 0.f, 0.f}; accl[mt] = f32x4c{0.f, 0.f, 0.f, 0.f}; }
    const uint4* __restrict__ bg = a.bfrag + lane;
    uint4 nbh = bg[0], nbl = bg[64];
#pragma unroll 2
    for (int ks = 0; ks < 14; ++ks) {
        const int dy = ks >> 1, h = ks & 1;
        f16x8c bh, bl;
        if (kP2pStreamB) {   // weight fragments straight from L2, one k-step ahead: 28 KB less LDS per workgroup
            bh = __builtin_bit_cast(f16x8c, nbh); bl = __builtin_bit_cast(f16x8c, nbl);
            const int kn = ks + 1 < 14 ? ks + 1 : ks;
            nbh = bg[(2 * kn + 0) * 64]; nbl = bg[(2 * kn + 1) * 64];
        } else {
            bh = __builtin_bit_cast(f16x8c, pB[(2 * ks + 0) * 64 + lane]);
            bl = __builtin_bit_cast(f16x8c, pB[(2 * ks + 1) * 64 + lane]);
        }
        f16x8c ah[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) ah[mt] = __builtin_bit_cast(f16x8c, pH[abase[mt] + dy * Tp + 4 * h]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt], bh, acc[mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            if (kP2pProducts == 2) accl[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt], bl, accl[mt], 0, 0, 0);
    }
    // ---- epilogue: D[row m = 4q + i][col n = 8*tau + co]; 32-bit offsets inside the clip ----
    const int tau = r16 >> 3, co = r16 & 7;
    const float bias = a.bias[co];
    const float iscale = reinterpret_cast<const float*>(a.bfrag + kP2pFragScale)[co];
    unsigned short* const oh = OUT_CL ? a.oh + static_cast<long long>(clip) * a.H * a.T * 8 + co : nullptr;
    float* const od = OUT_CL ? nullptr : a.dst + clip * a.dst_clip_stride + static_cast<long long>(a.dst_coff + co) * a.H * a.T;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m0 = (wave * MT + mt) * 16 + 4 * q;
        int r = m0 / J, j = m0 - r * J;
        float v[4];
        int pos[4];                                              // position (row * T + frame) inside the clip, or -1
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int t = 2 * j + tau;
            const float x = fmaf(kP2pProducts == 2 ? fmaf(accl[mt][i], kP2pLoInv, acc[mt][i]) : acc[mt][i], iscale, bias);
            v[i] = x > 0.f ? x : x * kSlope;
            pos[i] = (m0 + i < Mblk && t < a.T) ? (y0 + r) * a.T + t : -1;
            if (++j == J) { j = 0; ++r; }
        }
        if (OUT_CL) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (pos[i] >= 0) oh[pos[i] * 8] = static_cast<unsigned short>(f16_bits(v[i]));
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (pos[i] >= 0) od[pos[i]] = v[i];
        }
    }
}

// ==========================================================================================
// Persistent form of conv_p2p_f16_kernel (channels-last f16 plane in; even frame counts).
//
// Measured on the kernel above (256 clips, by switching its phases off one at a time): skeleton 39 us + patch loads 40 us +
// MFMA loop 63 us + stores 19 us = the 161 us of a launch -- the phases ADD, nothing overlaps.  All three memory phases go
// through the CU's one vector-memory path: the weight fragments streamed from L2 (229 KB per workgroup, 5x the patch), the
// patch, and 192 two-byte store instructions per workgroup; and a workgroup lives for 16 us, a third of it latency.
// Here ONE workgroup per CU walks ~29 row tiles:
//   * the 28 weight fragments (hi | lo * 2^11, 14 k-steps) live in registers for the whole launch: no weight traffic in the loop;
//   * the patch of tile k+1 is fetched by LDS-DMA (global_load_lds_dwordx4, circular halos resolved on the per-lane source
//     address) into the other half of a double buffer while tile k is multiplied;
//   * the epilogue transposes each 16 x 16 accumulator tile through a wave-private 1 KB LDS slab, so that a tile leaves as
//     ONE 16-byte store per lane (32 positions x 8 f16 channels, two tiles per instruction; or 8 channels x 32 consecutive frames of NCHW f32);
//     the stores of tile k are issued after tile k+1's loads, under its MFMA loop;
//   * tiles are dealt so that the workgroups of one XCD (blockIdx % 8) hold neighbouring row tiles: halo rows hit that L2.
// One barrier per tile.
// ==========================================================================================
struct P2pPsArgs {
    const unsigned short* xh;     // [clip][H][T][8] f16   (IN_NCHW == false)
    const float* p;               // IN_NCHW == true: the stack's input is assembled by the loader, as in conv_p2p_f16_kernel
    const float* u;
    const uint2* uh;              // NIN == 3: u as four f16 channels per (row, frame), [clip][h1][T] x 8 bytes (layer0_mfma_kernel's psix_h); c0 == 1
    const unsigned int* ph;       // NIN == 3: p as f16 hi | f16 lo << 16 words, [clip][H][T] (layer0_mfma_kernel's melh)
    int c0, c1, h1;
    int p_fm;                     // p is ONE channel stored frames-major, [clip][T][H] (the CQT filter bank's own output order)
    const uint4* bfrag;           // [14 k-steps][hi|lo * 2^11][64 lanes] x 8 f16
    const float* bias;            // [8]
    float* dst;                   // OUT == 0: NCHW f32 [clip][dst_ctot][H][T];  OUT == 2: semitone maps [clip][8][H / 3][T]
    long long dst_clip_stride;
    unsigned short* oh;           // OUT == 1: channels-last f16 plane
    const uint4* sfrag;           // OUT == 2: B fragments of the semitone conv [3 dy][hi|lo * 2^11][64 lanes] x 8 f16, and its bias [8]
    const float* sbias;
    int H, T, R, J, Tp, n_row_tiles, n_tiles, plane_pos;   // plane_pos: (R + 6) * Tp rounded up to 64 positions
    int n_oct, n_units;           // OUT == 3: octaves (H / 36) and work units (clip, group of R rows within an octave) = batch * 36 / R
    unsigned long long* stamps;   // diagnostic build (AKE_P2P_STAMP): [8 waves][8] cycle sums of the tile loop's sections, workgroup 0
};

__device__ __forceinline__ unsigned long long p2p_stamp() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}

constexpr int kP2pPieces = 3;        // 1 KB pieces of the patch a wave requests per tile (8 waves: planes of up to 24 pieces)
constexpr int kP2pPsStage = 80;      // uint4 per M-tile of a wave's staging slab (NCHW form: 8 channels x 36 floats, padded)

// OUT: 0 = NCHW f32, 1 = channels-last f16 plane (next conv of the stack), 2 = the stack's last conv FUSED with the semitone conv
//   that follows it (3x3, stride (3,1), time circular + BN + LeakyReLU; models.py:337-339, 386-388): the activated tile (R = 3k rows)
//   stays in LDS as an f16 plane [m][tau][8 ch] = position-major, the semitone conv runs over it on the same MFMA form
//   (m = (semitone row, frame pair), n = (tau, co), k-step = one of its 3 rows: 4 positions x 8 channels, the 4th tap zero), one
//   M-tile per wave, and only the semitone maps [clip][8][H / 3][T] go to memory: the 8 x H x T pitch tensor is never written.
//   OUT 3 = OUT 2 + Pitch2PitchClassPool (models.py:95-106): a workgroup walks the SAME rows of all octaves one after the other (its work
//   unit = (clip, group of R rows within an octave), tiles 36 rows apart), keeps the running maximum of the semitone outputs in
//   registers and writes the folded maps [clip][dst channel][12][T] once per unit: neither the semitone maps (60 MB per 256 clips) nor
//   a fold launch exist.
// NIN: 0 = channels-last f16 plane in; else the number of f32 channels the loader assembles (5: default net, 8: any); 3 = the default net's
//   first conv fed by layer 0's launch: the log-CQT as (f16 hi | f16 lo) words (P2pPsArgs::ph) + the repeated up_sixth map as f16 x 4
//   (P2pPsArgs::uh): two loads, three registers and six bit operations per patch position instead of five loads, five registers and the
//   conversions, which lets this form run two workgroups per CU like its plane-fed siblings (<1, 5>: 133 VGPRs, one workgroup per CU, 83 us
//   against their 46; the f32 log-CQT handed over frames-major also cost it a cache line per LANE in the texture addresser)
// STAMP: diagnostic build with s_memtime stamps around the tile loop's sections (tools/p2p_stamp.py; shares, never timed)
template <int OUT, int NIN, bool STAMP = false>
__global__ __launch_bounds__(512, ((NIN > 0 && NIN != 3) ? 2 : 4)) void conv_p2p_f16_ps_kernel(P2pPsArgs a) {   // (the assembling loader's 15 input registers do not fit 128)
    constexpr bool OUT_CL = OUT == 1, OUT_SEMI = OUT == 2 || OUT == 3, OUT_FOLD = OUT == 3;
    constexpr bool IN_NCHW = NIN > 0;
    constexpr bool IN_UH = NIN == 3;
    constexpr int NV = IN_UH ? 1 : (IN_NCHW ? NIN : 1);
    extern __shared__ __attribute__((aligned(16))) uint4 lds4[];
    constexpr int MT = kP2pMT;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r16 = lane & 15, q = lane >> 4;
    const int Tp = a.Tp, J = a.J, T = a.T;
    f16_saturate_mode();
    const int nchunk = a.plane_pos >> 6;                        // 1 KB pieces per plane
    const int npos = (a.R + 6) * Tp;
    // tiles of this workgroup: first, first + gridDim.x, ...; workgroups of one XCD take neighbouring tiles
    const int nwg = gridDim.x, per_xcd = nwg >> 3;
    const int first = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    // ---- loader: the pieces c = wave, wave + 8, ... of the plane; lane -> patch position -> (row offset, frame) ----
    int pk[kP2pPieces];
#pragma unroll
    for (int k = 0; k < kP2pPieces; ++k) {
        const int c = wave + 8 * k;
        int i = c * 64 + lane;
        i = i < npos ? i : npos - 1;
        const int rj = i / Tp, f = i - rj * Tp;
        pk[k] = (rj << 16) | wrap(f - 3, T);
    }
    auto issue_loads = [&](int tile, int buf) {
        const int clip = tile / a.n_row_tiles;
        const int y0 = (tile - clip * a.n_row_tiles) * a.R;
        const long long cbase = static_cast<long long>(clip) * a.H * T;
#pragma unroll
        for (int k = 0; k < kP2pPieces; ++k) {
            const int c = wave + 8 * k;
            if (c < nchunk) {
                int row = y0 - 3 + (pk[k] >> 16);
                row += row < 0 ? a.H : 0;
                row -= row >= a.H ? a.H : 0;
                const uint4* src = reinterpret_cast<const uint4*>(a.xh) + cbase + static_cast<long long>(row) * T + (pk[k] & 0xffff);
                uint4* dstl = lds4 + buf * a.plane_pos + c * 64;
                // inline asm: hipcc orders every later LDS access behind a builtin LDS-DMA with vmcnt(0) (it cannot tell the two buffer
                // halves apart), which would serialise load and multiply; the wait is placed by hand before the barrier instead
                const unsigned int lds_dst = static_cast<unsigned int>(reinterpret_cast<unsigned long long>(dstl));   // LDS aperture: low 32 bits = LDS byte address
                unsigned int keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(src), "s"(lds_dst) : "memory");
            }
        }
    };
    // IN_NCHW: f32 planes -> registers (requested before the multiply loop) -> f16 channels-last patch (written after it);
    // thread -> the patch positions threadIdx.x, + 512, + 1024
    float vin[3][NV];
    uint2 vuh[3];
    unsigned int vw[3];
    int pn[3];
    if (IN_NCHW) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int i = threadIdx.x + 512 * k;
            const int ic = i < npos ? i : npos - 1;
            const int rj = ic / Tp, f = ic - rj * Tp;
            pn[k] = (rj << 16) | wrap(f - 3, T);
        }
    }
    auto load_regs = [&](int tile) {      // branch-free (a branch around a load makes hipcc wait for every load at the join)
        const int clip = tile / a.n_row_tiles;
        const int y0 = (tile - clip * a.n_row_tiles) * a.R;
        const int ctot = a.c0 + a.c1;
        // Addressing: per channel a wave-uniform base (scalar registers), per patch position ONE 32-bit lane offset shared by all channels of a
        // source -- the loads then take the `saddr + voffset` form and cost no vector instruction each (round 2 built a 64-bit address per load:
        // 15 loads x ~4 vector instructions per thread and tile made this launch issue twice the vector instructions of its plane-fed siblings,
        // profiles/r03_a_pmc_mfma.md: 19.95 M against 9.84 M).
        const float* const pb = a.p_fm ? a.p + static_cast<long long>(clip) * T * a.H : a.p + static_cast<long long>(clip) * a.c0 * a.H * T;
        const float* const ub = a.u + static_cast<long long>(clip) * a.c1 * a.h1 * T;
        const uint2* const uhb = IN_UH ? uniform_ptr(a.uh + static_cast<long long>(clip) * a.h1 * T) : nullptr;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            int row = y0 - 3 + (pn[k] >> 16);
            row += row < 0 ? a.H : 0;
            row -= row >= a.H ? a.H : 0;
            const int t = pn[k] & 0xffff;
            if (IN_UH) {
                vw[k] = uniform_ptr(a.ph + static_cast<long long>(clip) * a.H * T)[static_cast<unsigned int>(row * T + t)];
                vuh[k] = uhb[static_cast<unsigned int>((row % a.h1) * T + t)];
                continue;
            }
            // (frames-major p: one 64-byte line holds 16 bins of a frame = this tile's rows; the lanes of a request walk the frames, so
            // it costs a cache line per lane in the texture addresser, but every line is read 16 times from the L1)
            const int poff = a.p_fm ? t * a.H + row : row * T + t;
            const int uoff = (row % a.h1) * T + t;
#pragma unroll
            for (int c = 0; c < NV; ++c) {
                const int cc = c < ctot ? c : ctot - 1;               // (channels >= ctot re-read the last one; zeroed when the patch is written)
                const float* const base = uniform_ptr(cc < a.c0 ? pb + static_cast<long long>(cc) * a.H * T : ub + static_cast<long long>(cc - a.c0) * a.h1 * T);
                vin[k][c] = base[static_cast<unsigned int>(cc < a.c0 ? poff : uoff)];
            }
        }
    };
    auto write_lds = [&](int buf) {
        uint4* const wH = lds4 + buf * a.plane_pos;
        const int ctot = a.c0 + a.c1;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            unsigned int hi[4] = {0, 0, 0, 0};
            typedef float f32x2w __attribute__((ext_vector_type(2)));
            if (IN_UH) {        // slots: 0 the log-CQT, 1..4 the up_sixth channels, 5 / 6 the log-CQT's low half / high half again (as <1, 5>)
                const unsigned int hb = vw[k] & 0xffffu;
                hi[0] = hb | (vuh[k].x << 16);
                hi[1] = (vuh[k].x >> 16) | (vuh[k].y << 16);
                hi[2] = (vuh[k].y >> 16) | (vw[k] & 0xffff0000u);
                hi[3] = hb;
                const int i = threadIdx.x + 512 * k;
                if (i < npos) wH[i] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
                continue;
            }
#pragma unroll
            for (int c = 0; c < NV; c += 2) {   // round to nearest even, two channels at a time
                const f32x2w v = {c < ctot ? vin[k][c] : 0.f, (c + 1 < NV && c + 1 < ctot) ? vin[k][c + 1 < NV ? c + 1 : c] : 0.f};
                hi[c >> 1] = __builtin_bit_cast(unsigned int, __builtin_convertvector(v, f16x2c));
            }
            if (NV <= kP2pSplit0 && ctot <= kP2pSplit0) {   // channel 0 (the raw log-CQT) again in the idle slots: low half (5), high half (6)
                const _Float16 h0 = static_cast<_Float16>(vin[k][0]);
                hi[2] = (hi[2] & 0xffffu) | (f16_bits(vin[k][0] - static_cast<float>(h0)) << 16);
                hi[3] = static_cast<unsigned int>(__builtin_bit_cast(unsigned short, h0));
            }
            const int i = threadIdx.x + 512 * k;
            if (i < npos) wH[i] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
        }
    };
    // tile sequence: first, first + nwg, ...; OUT 3: units first, first + nwg, ..., each the n_oct tiles (clip, octave o, group g)
    const int G = OUT_FOLD ? a.n_row_tiles / a.n_oct : 1;
    auto fold_tile = [&](int u, int o) { const int clip = u / G; return clip * a.n_row_tiles + o * G + (u - clip * G); };
    const int tile0 = OUT_FOLD ? (first < a.n_units ? fold_tile(first, 0) : -1) : (first < a.n_tiles ? first : -1);
    if (tile0 >= 0) {
        if (IN_NCHW) { load_regs(tile0); write_lds(0); }
        else issue_loads(tile0, 0);
    }
    // ---- weight fragments: registers, for the whole launch ----
    uint4 breg[28];                  // (kP2pProducts == 1: the lo halves are never used and never loaded)
#pragma unroll
    for (int i = 0; i < 28; ++i) breg[i] = a.bfrag[i * 64 + lane];
    int abase[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int m = (wave * MT + mt) * 16 + r16;
        m = m < a.R * J ? m : a.R * J - 1;
        const int r = m / J, j = m - r * J;
        abase[mt] = r * Tp + 2 * j + q;
    }
    const int tau = r16 >> 3, co = r16 & 7;
    const float bias = a.bias[co];
    const float iscale = reinterpret_cast<const float*>(a.bfrag + kP2pFragScale)[co];
    uint4* const stage = lds4 + 2 * a.plane_pos + wave * (MT * kP2pPsStage);          // OUT 0 / 1: wave-private slabs
    uint4* const opatch = lds4 + 2 * a.plane_pos;                                      // OUT 2: [2 buffers][384 m][2 tau] positions
    constexpr int kOP = 8 * MT * 16 * 2;                                               // positions of one output patch
    typedef float f32x2e __attribute__((ext_vector_type(2)));
    long long prev_base = 0;          // element offset of the pending tile's first position (channels-last: position index)
    int prev_mblk = 0;
    bool has_prev = false;
    auto store_pending = [&]() {      // the finished tile waits in the staging slab (not in registers: the multiply loop needs them all)
        if (OUT_CL) {       // lane = (M-tile of a pair, m, tau): position 2 * (mbase + m) + tau of the tile, 8 channels = 16 bytes
            static_assert(MT == 3, "the M-tiles leave as one pair and a single");
            const uint4 o01 = stage[(lane >> 5) * kP2pPsStage + (lane & 31)], o2 = stage[2 * kP2pPsStage + (lane & 31)];
            const int mb01 = (wave * MT + (lane >> 5)) * 16, mb2 = (wave * MT + 2) * 16, ml = (lane >> 1) & 15;
            uint4* const o = reinterpret_cast<uint4*>(a.oh) + prev_base + (lane & 31);
            if (mb01 + ml < prev_mblk) o[2 * mb01] = o01;
            if (lane < 32 && mb2 + ml < prev_mblk) o[2 * mb2] = o2;
            return;
        }
        uint4 outv[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) outv[mt] = stage[mt * kP2pPsStage + (lane >> 3) * 9 + (lane & 7)];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int mbase = (wave * MT + mt) * 16;
            {               // lane = (channel, group of 4 frames)
                const int m = mbase + 2 * (lane & 7);
                float* o = a.dst + prev_base + static_cast<long long>(lane >> 3) * a.H * T + 2 * m;
                if (m + 1 < prev_mblk) *reinterpret_cast<uint4*>(o) = outv[mt];
                else if (m < prev_mblk) { o[0] = __uint_as_float(outv[mt].x); o[1] = __uint_as_float(outv[mt].y); }
            }
        }
    };
    // bias + LeakyReLU, transposed into the wave's staging slab
    typedef float f32x4c __attribute__((ext_vector_type(4)));
    auto epilogue = [&](const f32x4c (&acc)[MT], const f32x4c (&accl)[MT], int obuf) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float x = fmaf(kP2pProducts == 2 ? fmaf(accl[mt][i], kP2pLoInv, acc[mt][i]) : acc[mt][i], iscale, bias);
                v[i] = fmaxf(x, x * kSlope);            // LeakyReLU (slope < 1)
            }
            if (OUT_CL || OUT_SEMI) {
                // every lane writes its own channel's four rows as 2-byte LDS stores ([m][tau][8 co] halves: channels-last).  (Round 1
                // traded halves between the lanes (co, co ^ 1) to write whole dwords: 2 DPP moves + 6 selects per M-tile -- vector
                // instructions, which on this chip are paid in full next to the MFMAs, for LDS stores that cost next to nothing.)
                // OUT 1: the M-tile's own slab; OUT 2 / 3: the tile-wide patch, m counted over the tile (= position 2m + tau: T = 2J)
                unsigned short* st = OUT_SEMI ? reinterpret_cast<unsigned short*>(opatch + obuf * kOP) + (wave * MT + mt) * 256
                                              : reinterpret_cast<unsigned short*>(stage + mt * kP2pPsStage);
                const f32x2e x01 = {v[0], v[1]}, x23 = {v[2], v[3]};
                const unsigned int p01 = __builtin_bit_cast(unsigned int, __builtin_convertvector(x01, f16x2c));
                const unsigned int p23 = __builtin_bit_cast(unsigned int, __builtin_convertvector(x23, f16x2c));
                const int e = (8 * q + tau) * 8 + co;                    // row m = 4q + i: + 16 i
                st[e] = static_cast<unsigned short>(p01);
                st[e + 16] = static_cast<unsigned short>(p01 >> 16);
                st[e + 32] = static_cast<unsigned short>(p23);
                st[e + 48] = static_cast<unsigned short>(p23 >> 16);
            } else {
                float* st = reinterpret_cast<float*>(stage + mt * kP2pPsStage);
#pragma unroll
                for (int i = 0; i < 4; ++i) st[co * 36 + (4 * q + i) * 2 + tau] = v[i];
            }
        }
    };
    // ---- OUT 2: semitone conv over the finished tile (one M-tile of 16 (semitone row, frame pair) positions per wave) ----
    uint4 sreg[6];
    float sbias = 0.f, siscale = 1.f;
    int sbase = 0;
    if (OUT_SEMI) {
#pragma unroll
        for (int i = 0; i < 6; ++i) sreg[i] = a.sfrag[i * 64 + lane];
        sbias = a.sbias[co];
        siscale = reinterpret_cast<const float*>(a.sfrag + 6 * 64)[co];
        int ms = wave * 16 + r16;
        ms = ms < (a.R / 3) * J ? ms : (a.R / 3) * J - 1;
        const int srow = ms / J, sj = ms - srow * J;
        sbase = 3 * srow * T + wrap(2 * sj - 1 + q, T);               // A[m][k = (position q, ci)] = X[3s + dy][2j - 1 + q][ci]
    }
    float smax[4] = {0.f, 0.f, 0.f, 0.f};                             // OUT 3: running maximum over the octaves of a unit
    auto semi_stage = [&](int obuf, long long base, int mblk, int oct) {   // base: element offset of (clip, channel 0, first semitone row) in dst
        const uint4* const oH = opatch + obuf * kOP;
        f32x4c sacc = {0.f, 0.f, 0.f, 0.f}, saccl = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const f16x8c ah = __builtin_bit_cast(f16x8c, oH[sbase + dy * T]);
            const f16x8c bh = __builtin_bit_cast(f16x8c, sreg[2 * dy]), bl = __builtin_bit_cast(f16x8c, sreg[2 * dy + 1]);
            sacc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, sacc, 0, 0, 0);
            if (kP2pProducts == 2) saccl = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, saccl, 0, 0, 0);
        }
        const int S = OUT_FOLD ? 12 : a.H / 3;
        float* const o = a.dst + base + static_cast<long long>(co) * S * T + tau;
#pragma unroll
        for (int i = 0; i < 4; ++i) {                                 // D[m = 4q + i][n = (tau, co)]: semitone position 2m + tau of the tile
            const int m = wave * 16 + 4 * q + i;
            const float x = fmaf(kP2pProducts == 2 ? fmaf(saccl[i], kP2pLoInv, sacc[i]) : sacc[i], siscale, sbias);
            const float v = fmaxf(x, x * kSlope);
            if (!OUT_FOLD) {
                if (3 * m < mblk) o[2 * m] = v;
            } else {
                smax[i] = oct == 0 ? v : fmaxf(smax[i], v);
                if (oct == a.n_oct - 1 && 3 * m < mblk) o[2 * m] = smax[i];
            }
        }
    };
    // The waves 4..7 run their epilogue one barrier late (the accumulators wait in registers): each SIMD holds one wave of either
    // half, so one half's epilogue, loads and stores issue under the other half's MFMAs instead of all eight waves leaving the
    // matrix pipe idle together
    const bool late = !OUT_SEMI && wave >= 4;
    f32x4c acc[MT], accl[MT];       // products with the hi / the (scaled) lo weight plane
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) { acc[mt] = f32x4c{0.f, 0.f, 0.f, 0.f}; accl[mt] = f32x4c{0.f, 0.f, 0.f, 0.f}; }
    int cur = 0;
    unsigned long long sm[6] = {0, 0, 0, 0, 0, 0}, ts[6];
    int unit = first, oct = 0, prev_oct = 0;
    for (int tile = tile0; tile >= 0; cur ^= 1) {
        int next, n_unit = unit, n_oct_i = oct;
        if (OUT_FOLD) {
            if (++n_oct_i == a.n_oct) { n_oct_i = 0; n_unit += nwg; }
            next = n_unit < a.n_units ? fold_tile(n_unit, n_oct_i) : -1;
        } else next = tile + nwg < a.n_tiles ? tile + nwg : -1;
        if (STAMP) ts[0] = p2p_stamp();
        // this wave's share of the tile's patch has landed (and its stores have left).  The builtin, not asm: hipcc then knows that
        // nothing of its own is pending at the loop top and places no vmcnt wait inside the loop that would also drain the LDS-DMA
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
        if (STAMP) ts[1] = p2p_stamp();
        __syncthreads();              // ... every wave's; and every wave is done with the other half
        if (STAMP) ts[2] = p2p_stamp();
        const bool more = next >= 0;
        if (has_prev) {
            if (OUT_SEMI) semi_stage(cur ^ 1, prev_base, prev_mblk, prev_oct);
            else if (late) epilogue(acc, accl, 0);
        }
        const uint4* const pH = lds4 + cur * a.plane_pos;
        if (STAMP) ts[3] = p2p_stamp();
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) { acc[mt] = f32x4c{0.f, 0.f, 0.f, 0.f}; accl[mt] = f32x4c{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int ks = 0; ks < 14; ++ks) {
            // the next tile's loads and the previous tile's stores are issued from inside the multiply loop rather than in front of it,
            // where both waves of a SIMD sit right after the barrier with the matrix pipe idle (-2 % per launch; switched off
            // altogether, loads and stores account for 11 + 15 us of a 128 us launch wherever they are placed -- see DESIGN.md)
            if (ks == 2 && more) {
                if (IN_NCHW) load_regs(next);
                else issue_loads(next, cur ^ 1);
            }
            if (ks == 6 && !OUT_SEMI && has_prev) store_pending();
            const int dy = ks >> 1, h = ks & 1;
            const f16x8c bh = __builtin_bit_cast(f16x8c, breg[2 * ks]), bl = __builtin_bit_cast(f16x8c, breg[2 * ks + 1]);
            f16x8c ah[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) ah[mt] = __builtin_bit_cast(f16x8c, pH[abase[mt] + dy * Tp + 4 * h]);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt], bh, acc[mt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                if (kP2pProducts == 2) accl[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt], bl, accl[mt], 0, 0, 0);
        }
        if (STAMP) ts[4] = p2p_stamp();
        if (!late) epilogue(acc, accl, cur);
        if (IN_NCHW && more) write_lds(cur ^ 1);
        {
            const int clip = tile / a.n_row_tiles;
            const int y0 = (tile - clip * a.n_row_tiles) * a.R;
            const int rows_here = a.H - y0 < a.R ? a.H - y0 : a.R;
            prev_mblk = rows_here * J;
            prev_base = OUT_CL ? static_cast<long long>(clip) * a.H * T + static_cast<long long>(y0) * T
                      : OUT_FOLD ? clip * a.dst_clip_stride + static_cast<long long>((y0 % 36) / 3) * T
                      : OUT_SEMI ? clip * a.dst_clip_stride + static_cast<long long>(y0 / 3) * T
                                 : clip * a.dst_clip_stride + static_cast<long long>(y0) * T;
            prev_oct = oct;
        }
        has_prev = true;
        tile = next; unit = n_unit; oct = n_oct_i;
        if (STAMP) {
            ts[5] = p2p_stamp();
#pragma unroll
            for (int i = 0; i < 5; ++i) sm[i] += ts[i + 1] - ts[i];
            sm[5] += 1;
        }
    }
    if (STAMP && blockIdx.x == 0 && lane == 0 && a.stamps) {
#pragma unroll
        for (int i = 0; i < 6; ++i) a.stamps[wave * 8 + i] = sm[i];
    }
    if (has_prev) {
        if (OUT_SEMI) {
            __syncthreads();          // every wave's share of the last tile is in the output patch
            semi_stage(cur ^ 1, prev_base, prev_mblk, prev_oct);
        } else {
            if (late) epilogue(acc, accl, 0);
            store_pending();
        }
    }
}

// ==========================================================================================
// Training-mode form of the persistent pitch convolution ("f16 x 3"): the same tiling and MFMA form as conv_p2p_f16_ps_kernel, but with
// f32-equivalent products -- the step's gradients are held to 2e-5 of float64 autograd, so nothing may be rounded to 11 bits here.
//   x = xh + xl, w = wh + wl, all f16, the second halves stored times 2^11 (normal numbers):  x * w ~ xh*wh + 2^-11 (xl'*wh + xh*wl')
//   (three MFMAs, two accumulator sets; the dropped xl*wl is 2^-22 of the product).
// Used for (a) the train-mode forward: input = the previous layer's RAW output with its pending BatchNorm + LeakyReLU applied on load
// (per-channel (scale, shift, slope) table, as conv_mfma_kernel<.., TRAIN>), output = raw f32 NCHW + per-channel sum / sum of squares
// for this layer's BatchNorm; (b) the data gradient: input = dz, weights = the forward's transposed and flipped, no table, no bias, no
// statistics.  Replaces conv_mfma_kernel (f32 MFMA, the vector rate) for these shapes: 0.53 -> 0.2 ms per convolution and 256 clips.
// ==========================================================================================
struct P2pTrArgs {
    const float* p;               // input channels [0, c0): [clip][c0][H][T]
    const float* u;               // input channels [c0, c0 + c1): [clip][c1][h1][T], row % h1 (the repeated up_sixth map), or unused (c1 = 0)
    int c0, c1, h1;
    const float* in_aff;          // [c0 + c1][3] (scale, shift, negative slope) applied on load, or null
    const uint4* bfrag;           // [14 k-steps][hi | lo * 2^11][64 lanes] x 8 f16, then the 8 inverse channel scales
    const float* bias;            // [cout] or null
    float* dst;                   // raw output, NCHW f32 [clip][cout][H][T]
    long long dst_clip_stride;
    int cout;
    double* stats;                // [kStatSlots][stats_stride]: (sum, sum of squares) of channel c at 2c, 2c + 1 (fixed point), or null
    int stats_stride;
    int H, T, R, J, Tp, n_row_tiles, n_tiles, plane_pos;
    const unsigned int* in_amax;  // data gradient (in_aff == null): bits of the largest |dz|; dz is staged times f16_weight_scale(max), divided out in the epilogue
    int lrelu;                    // 1: LeakyReLU on the stored value (inference in the f32x3 precision mode: BatchNorm is folded into bfrag / bias)
};

__global__ __launch_bounds__(512) void conv_p2p_f16x3_kernel(P2pTrArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint4 lds4[];
    constexpr int MT = kP2pMT, NV = 8;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r16 = lane & 15, q = lane >> 4;
    const int Tp = a.Tp, J = a.J, T = a.T;
    const int npos = (a.R + 6) * Tp;
    const int nwg = gridDim.x, per_xcd = nwg >> 3;
    const int first = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    const int ctot = a.c0 + a.c1;
    // LDS: [2 buffers][hi | lo][plane_pos] positions of 8 channels, then the waves' staging slabs
    float vin[3][NV];
    int pn[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int i = threadIdx.x + 512 * k;
        const int ic = i < npos ? i : npos - 1;
        const int rj = ic / Tp, f = ic - rj * Tp;
        pn[k] = (rj << 16) | wrap(f - 3, T);
    }
    auto load_regs = [&](int tile) {
        const int clip = tile / a.n_row_tiles;
        const int y0 = (tile - clip * a.n_row_tiles) * a.R;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            int row = y0 - 3 + (pn[k] >> 16);
            row += row < 0 ? a.H : 0;
            row -= row >= a.H ? a.H : 0;
            const int t = pn[k] & 0xffff;
            const float* pp = a.p + (static_cast<long long>(clip) * a.c0 * a.H + row) * T + t;
            const float* pu = a.u + (static_cast<long long>(clip) * a.c1 * a.h1 + row % a.h1) * T + t;
#pragma unroll
            for (int c = 0; c < NV; ++c) {
                const int cc = c < ctot ? c : ctot - 1;
                const float* src = cc < a.c0 ? pp + static_cast<long long>(cc) * a.H * T : pu + static_cast<long long>(cc - a.c0) * a.h1 * T;
                vin[k][c] = *src;
            }
        }
    };
    // the pending BatchNorm + LeakyReLU of the input, per channel (registers: the table is tiny and read once)
    float asc[NV], ash[NV], ang[NV];
#pragma unroll
    for (int c = 0; c < NV; ++c) {
        const bool on = a.in_aff != nullptr && c < ctot;
        asc[c] = on ? a.in_aff[3 * c] : 1.f; ash[c] = on ? a.in_aff[3 * c + 1] : 0.f; ang[c] = on ? a.in_aff[3 * c + 2] : 1.f;
    }
    const float in_mul = (a.in_amax && !a.in_aff) ? f16_weight_scale(amax_load(a.in_amax)) : 1.f;
    if (in_mul != 1.f) {
#pragma unroll
        for (int c = 0; c < NV; ++c) asc[c] = in_mul;
    }
    auto write_lds = [&](int buf) {
        uint4* const wH = lds4 + (2 * buf) * a.plane_pos;
        uint4* const wL = wH + a.plane_pos;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            unsigned int hi[4] = {0, 0, 0, 0}, lo[4] = {0, 0, 0, 0};
#pragma unroll
            for (int c = 0; c < NV; ++c) {
                float x = 0.f;
                if (c < ctot) {
                    x = fmaf(vin[k][c], asc[c], ash[c]);
                    x = x > 0.f ? x : x * ang[c];
                }
                const _Float16 h = static_cast<_Float16>(x);
                const unsigned int hb = __builtin_bit_cast(unsigned short, h);
                const unsigned int lb = f16_bits((x - static_cast<float>(h)) * kP2pLoScale);
                hi[c >> 1] |= hb << (16 * (c & 1));
                lo[c >> 1] |= lb << (16 * (c & 1));
            }
            const int i = threadIdx.x + 512 * k;
            if (i < npos) { wH[i] = make_uint4(hi[0], hi[1], hi[2], hi[3]); wL[i] = make_uint4(lo[0], lo[1], lo[2], lo[3]); }
        }
    };
    const int tile0 = first < a.n_tiles ? first : -1;
    if (tile0 >= 0) { load_regs(tile0); write_lds(0); }
    uint4 breg[28];
#pragma unroll
    for (int i = 0; i < 28; ++i) breg[i] = a.bfrag[i * 64 + lane];
    int abase[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int m = (wave * MT + mt) * 16 + r16;
        m = m < a.R * J ? m : a.R * J - 1;
        const int r = m / J, j = m - r * J;
        abase[mt] = r * Tp + 2 * j + q;
    }
    const int tau = r16 >> 3, co = r16 & 7;
    const float bias = (a.bias && co < a.cout) ? a.bias[co] : 0.f;
    const float iscale = reinterpret_cast<const float*>(a.bfrag + kP2pFragScale)[co] / in_mul;
    uint4* const stage = lds4 + 4 * a.plane_pos + wave * (MT * kP2pPsStage);
    typedef float f32x4c __attribute__((ext_vector_type(4)));
    long long prev_base = 0;
    int prev_mblk = 0;
    bool has_prev = false;
    auto store_pending = [&]() {          // lane = (channel, group of 4 frames), as the OUT = 0 form of conv_p2p_f16_ps_kernel
        uint4 outv[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) outv[mt] = stage[mt * kP2pPsStage + (lane >> 3) * 9 + (lane & 7)];
        if ((lane >> 3) >= a.cout) return;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = (wave * MT + mt) * 16 + 2 * (lane & 7);
            float* o = a.dst + prev_base + static_cast<long long>(lane >> 3) * a.H * T + 2 * m;
            if (m + 1 < prev_mblk) *reinterpret_cast<uint4*>(o) = outv[mt];
            else if (m < prev_mblk) { o[0] = __uint_as_float(outv[mt].x); o[1] = __uint_as_float(outv[mt].y); }
        }
    };
    ShiftStat sst;                        // this lane's share of channel co's statistics, over all tiles of the workgroup
    sst.init();
    int cur = 0;
    for (int tile = tile0; tile >= 0; cur ^= 1) {
        const int next = tile + nwg < a.n_tiles ? tile + nwg : -1;
        __syncthreads();                  // the patch of this tile is complete; every wave is done with the other buffer
        const uint4* const pH = lds4 + (2 * cur) * a.plane_pos;
        const uint4* const pL = pH + a.plane_pos;
        f32x4c acc[MT], accl[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) { acc[mt] = f32x4c{0.f, 0.f, 0.f, 0.f}; accl[mt] = f32x4c{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int ks = 0; ks < 14; ++ks) {
            if (ks == 2 && next >= 0) load_regs(next);
            if (ks == 6 && has_prev) store_pending();
            const int dy = ks >> 1, h = ks & 1;
            const f16x8c bh = __builtin_bit_cast(f16x8c, breg[2 * ks]), bl = __builtin_bit_cast(f16x8c, breg[2 * ks + 1]);
            f16x8c ah[MT], al[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int ad = abase[mt] + dy * Tp + 4 * h;
                ah[mt] = __builtin_bit_cast(f16x8c, pH[ad]);
                al[mt] = __builtin_bit_cast(f16x8c, pL[ad]);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt], bh, acc[mt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) accl[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[mt], bh, accl[mt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) accl[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt], bl, accl[mt], 0, 0, 0);
        }
        const int clip = tile / a.n_row_tiles;
        const int y0 = (tile - clip * a.n_row_tiles) * a.R;
        const int rows_here = a.H - y0 < a.R ? a.H - y0 : a.R;
        const int mblk = rows_here * J;
        // epilogue: raw value (+ bias), statistics, transposed into the wave's staging slab
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            float* st = reinterpret_cast<float*>(stage + mt * kP2pPsStage);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float v = fmaf(fmaf(accl[mt][i], kP2pLoInv, acc[mt][i]), iscale, bias);
                if ((wave * MT + mt) * 16 + 4 * q + i < mblk) sst.add(v);
                st[co * 36 + (4 * q + i) * 2 + tau] = a.lrelu ? fmaxf(v, v * kSlope) : v;
            }
        }
        if (next >= 0) write_lds(cur ^ 1);
        prev_mblk = mblk;
        prev_base = clip * a.dst_clip_stride + static_cast<long long>(y0) * T;
        has_prev = true;
        tile = next;
    }
    if (has_prev) store_pending();
    if (a.stats) {                        // lanes of one channel: r16 = co and co + 8 (tau), four q
        long long s1, s2;
        sst.fixed(s1, s2);
        s1 += shfl_xor_ll(s1, 8); s2 += shfl_xor_ll(s2, 8);
        s1 += shfl_xor_ll(s1, 16); s2 += shfl_xor_ll(s2, 16);
        s1 += shfl_xor_ll(s1, 32); s2 += shfl_xor_ll(s2, 32);
        if (lane < 8 && lane < a.cout) {
            double* st = a.stats + static_cast<size_t>((blockIdx.x + wave) & (kStatSlots - 1)) * a.stats_stride;
            fx_add_int(st + 2 * lane, s1);
            fx_add_int(st + 2 * lane + 1, s2);
        }
    }
}

// ==========================================================================================
// Equivariant pitch-class convolution (12 x 7 kernel, rows circular over the 12 pitch classes, time zero-padded or valid;
// models.py:36-47) on bf16 MFMA with split operands -- the PitchClass2PitchClass stacks and the first convolution of the
// key / tonic heads.  Same idea as conv_p2p_f16_kernel: channels-last activations [clip][12][T][16] as bf16 hi / lo planes,
//   m = (pitch class y, frame t)       A[m][k] = X[(y + dy) mod 12][t + dx - pad][ci]
//   n = output channel (NT tiles of 16)  B[k][n] = w[co][ci][dy][dx]
//   k-step = (dy, tap pair p): lane q holds tap dx = 2p + (q >> 1), channels 8 (q & 1) .. +7; the seventh taps of kernel rows dy and
//   dy + 1 (dy even) share the k-step p = 3 of row dy, which has none at odd dy -> 12 x 3 + 6 = 42 k-steps (a single kernel row: 4,
//   dx = 7 a zero tap), three MFMAs per k-step, M-tile and N-tile.  The whole clip (12 rows, all frames) is one LDS patch; the weight
//   fragments (96 KB per N-tile) come through a double-buffered LDS ring, one kernel row (4 k-steps) ahead, fetched once per workgroup.
// ==========================================================================================
struct PcBfArgs {
    const unsigned short* xh;     // [clip][12][T_in][16]
    const unsigned short* xl;
    const uint4* bfrag;           // [KH * 4 k-steps][NT][hi|lo][64 lanes] x 8 bf16
    const float* bias;            // [cout]
    float* dst;                   // NCHW f32 [clip][dst_ctot][12][T_out] (OUT_CL == false)
    long long dst_clip_stride;
    unsigned short* oh;           // channels-last planes [clip][12][T_out][cl_stride] (OUT_CL == true; cl_stride = cout = 16 or 32)
    unsigned short* ol;
    int T_in, T_out, pad_l, Tp, cout, lrelu, cl_stride;
    int KH, circular;             // kernel rows (12 circular for the pitch-class convs; 1, rows independent, for the genre head's first conv)
    // blockIdx.y == 1: a second convolution of the same geometry over the same input (the key and the tonic head in one launch)
    const uint4* bfrag2;
    const float* bias2;
    unsigned short* oh2;
    unsigned short* ol2;
    // F16X3 (training): per-channel statistics of the raw output, [kStatSlots][stats_stride] doubles (fixed point), (sum, sum of squares)
    // of channel c at 2c, 2c + 1; or null
    double* stats;
    int stats_stride;
    int accumulate;               // F16X3, NCHW output: dst += (a data gradient that arrives in two 16-channel halves, or from several heads)
    const unsigned int* in_amax;  // F16X3 data gradients: the input planes hold value * f16_weight_scale(*in_amax) (nchw_to_cl16_f16x2_kernel); or null
};

// F16X3 (training mode; forward with BatchNorm-on-load planes and the data gradients): the planes and fragments hold f16 hi and
// f16 lo * 2^11 instead of bf16 hi / lo, the products are xh*wh and (xl'*wh + xh*wl') in an accumulator of their own folded in with 2^-11
// (2^-22 of a product dropped: f32-equivalent, the gradient tests hold 2e-5), every output channel's weights are scaled by a power of
// two (inverse scales behind the fragments), and the raw output's per-channel sums go to the BatchNorm statistics.
template <int NT, bool OUT_CL, bool F16X3 = false>
__global__ __launch_bounds__(512) void conv_pc_bf16_kernel(PcBfArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint4 lds4[];
    const int clip = blockIdx.z;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float in_mul = (F16X3 && a.in_amax) ? f16_weight_scale(amax_load(a.in_amax)) : 1.f;      // data gradients: the planes hold dz * in_mul
    const int nw = blockDim.x >> 6;
    const int r16 = lane & 15, q = lane >> 4;
    const int Tp = a.Tp;
    constexpr int MT = 4;
    const int H_out = a.circular ? 12 : 12 - a.KH + 1;
    const int Mtot = H_out * a.T_out;
    uint4* const pH = lds4;                                  // [12][Tp][2 halves of 8 channels]
    uint4* const pL = lds4 + 12 * Tp * 2;
    // weight fragments of one kernel row (4 k-steps x NT x (hi | lo) x 64 lanes), double-buffered: fetched ONCE per workgroup by LDS-DMA,
    // one row ahead, and read from here by every wave (streamed per wave from L1 they cost a third of the fused stack's time, see
    // pc2pc_fused_kernel)
    uint4* const wring = lds4 + 2 * 12 * Tp * 2;
    constexpr int kRow = 4 * NT * 2 * 64;                    // uint4 per kernel row
    const bool second = blockIdx.y == 1;
    const uint4* const bfr = second ? a.bfrag2 : a.bfrag;
    auto fetch_row = [&](int dyn) {                          // pieces of 64 lanes x 16 bytes, dealt to the waves
        for (int pc = wave; pc < kRow / 64; pc += nw) {
            const uint4* src = bfr + dyn * kRow + pc * 64 + lane;
            const unsigned int lds_dst = static_cast<unsigned int>(reinterpret_cast<unsigned long long>(wring + (dyn & 1) * kRow + pc * 64));
            unsigned int keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(src), "s"(lds_dst) : "memory");
        }
    };
    fetch_row(0);
    {   // patch frame f <-> input frame f - pad_l, zeros outside [0, T_in)
        const long long cbase = static_cast<long long>(clip) * 12 * a.T_in * 2;
        const uint4* gh = reinterpret_cast<const uint4*>(a.xh) + cbase;
        const uint4* gl = reinterpret_cast<const uint4*>(a.xl) + cbase;
        const int n16 = 12 * Tp * 2;
        for (int i = threadIdx.x; i < n16; i += blockDim.x) {
            const int half = i & 1, pos = i >> 1;
            const int row = pos / Tp, f = pos - row * Tp;
            const int t = f - a.pad_l;
            uint4 vh = make_uint4(0, 0, 0, 0), vl = make_uint4(0, 0, 0, 0);
            if (t >= 0 && t < a.T_in) {
                const long long g = (static_cast<long long>(row) * a.T_in + t) * 2 + half;
                vh = gh[g]; vl = gl[g];
            }
            pH[i] = vh; pL[i] = vl;
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                       // vmcnt(0): this wave's share of row 0 has landed
    __syncthreads();
    const int tile0 = (blockIdx.x * nw + wave) * MT;          // first M-tile of this wave
    const bool active = tile0 * 16 < Mtot;                    // (idle waves still fetch and meet the barriers)
    int ay[MT], at[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int m = (tile0 + mt) * 16 + r16;
        if (m >= Mtot) m = Mtot - 1;
        ay[mt] = m / a.T_out;
        at[mt] = m - ay[mt] * a.T_out;
    }
    typedef float f32x4c __attribute__((ext_vector_type(4)));
    f32x4c acc[MT][NT], accl[F16X3 ? MT : 1][F16X3 ? NT : 1];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            acc[mt][nt] = f32x4c{0.f, 0.f, 0.f, 0.f};
            if (F16X3) accl[mt][nt] = f32x4c{0.f, 0.f, 0.f, 0.f};
        }
    const int dxq = q >> 1, half = q & 1;
    const bool pair = !(a.KH & 1);
    for (int dy = 0; dy < a.KH; ++dy) {
        if (dy + 1 < a.KH) fetch_row(dy + 1);                 // lands in the other half during this row's MFMAs
        const uint4* const wr = wring + (dy & 1) * kRow + lane;
        if (active) {
        int rowoff[MT], rowoff3[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            int row = ay[mt] + dy;
            row -= row >= 12 ? 12 : 0;
            rowoff[mt] = ((row * Tp + at[mt] + dxq) << 1) + half;
            // An even number of kernel rows: the seventh taps of rows dy and dy + 1 share ONE k-step (at even dy; lanes of the second tap
            // slot read the next input row at the same frame) instead of each filling half a k-step next to a zero tap: 7 k-steps per
            // pair of kernel rows instead of 8 (pack_pc_bf16_kernel / pack_pc_f16x3_body build the fragments to match)
            int row3 = ay[mt] + dy + dxq;
            row3 -= row3 >= 12 ? 12 : 0;
            rowoff3[mt] = pair ? ((row3 * Tp + at[mt] + 6) << 1) + half : rowoff[mt] + 12;
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            if (p == 3 && pair && (dy & 1)) break;
            uint4 bhu[NT], blu[NT], ahu[MT], alu[MT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                bhu[nt] = wr[((p * NT + nt) * 2 + 0) * 64];
                blu[nt] = wr[((p * NT + nt) * 2 + 1) * 64];
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                ahu[mt] = pH[p == 3 ? rowoff3[mt] : rowoff[mt] + 4 * p];
                alu[mt] = pL[p == 3 ? rowoff3[mt] : rowoff[mt] + 4 * p];
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if (F16X3) {
                    const f16x8c bh = __builtin_bit_cast(f16x8c, bhu[nt]), bl = __builtin_bit_cast(f16x8c, blu[nt]);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8c, ahu[mt]), bh, acc[mt][nt], 0, 0, 0);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) accl[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8c, alu[mt]), bh, accl[mt][nt], 0, 0, 0);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) accl[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8c, ahu[mt]), bl, accl[mt][nt], 0, 0, 0);
                } else {
                    const bf16x8c bh = __builtin_bit_cast(bf16x8c, bhu[nt]), bl = __builtin_bit_cast(bf16x8c, blu[nt]);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8c, ahu[mt]), bh, acc[mt][nt], 0, 0, 0);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8c, alu[mt]), bh, acc[mt][nt], 0, 0, 0);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8c, ahu[mt]), bl, acc[mt][nt], 0, 0, 0);
                }
            }
        }
        }
        if (dy + 1 < a.KH) {
            __builtin_amdgcn_s_waitcnt(0x0F70);               // this wave's share of the next row has landed ...
            __syncthreads();                                  // ... everybody's; and everybody is done with this row's half
        }
    }
    if (!active) return;
    // ---- epilogue: D[row m = 4q + i][col = co within the N-tile] ----
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int co = nt * 16 + r16;
        const float* const bptr = second ? a.bias2 : a.bias;
        const float bias = (co < a.cout && bptr) ? bptr[co] : 0.f;
        float iscale = (F16X3 && co < a.cout) ? reinterpret_cast<const float*>(bfr + a.KH * 4 * NT * 2 * 64)[co] : 1.f;
        if (F16X3 && a.in_amax) iscale /= in_mul;        // (powers of two: exact)
        ShiftStat sstat;
        sstat.init();
        unsigned short* const oh = second ? a.oh2 : a.oh;
        unsigned short* const ol = second ? a.ol2 : a.ol;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m0 = (tile0 + mt) * 16 + 4 * q;
            int y = m0 / a.T_out, t = m0 - y * a.T_out;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (m0 + i < Mtot && co < a.cout) {
                    float v = F16X3 ? fmaf(fmaf(accl[F16X3 ? mt : 0][F16X3 ? nt : 0][i], kP2pLoInv, acc[mt][nt][i]), iscale, bias) : acc[mt][nt][i] + bias;
                    if (F16X3) sstat.add(v);
                    if (a.lrelu) v = v > 0.f ? v : v * kSlope;
                    if (OUT_CL) {
                        const long long idx = ((static_cast<long long>(clip) * H_out + y) * a.T_out + t) * a.cl_stride + co;
                        const unsigned int hb = bf16_bits(v);
                        oh[idx] = static_cast<unsigned short>(hb);
                        ol[idx] = static_cast<unsigned short>(bf16_bits(v - __uint_as_float(hb << 16)));
                    } else {
                        float* const d = a.dst + clip * a.dst_clip_stride + (static_cast<long long>(co) * H_out + y) * a.T_out + t;
                        *d = (F16X3 && a.accumulate) ? *d + v : v;
                    }
                }
                if (++t == a.T_out) { t = 0; ++y; }
            }
        }
        if (F16X3 && a.stats) {   // channel co's statistics: the four q-groups of lanes hold different positions
            long long st1, st2;
            sstat.fixed(st1, st2);
            st1 += shfl_xor_ll(st1, 16); st2 += shfl_xor_ll(st2, 16);
            st1 += shfl_xor_ll(st1, 32); st2 += shfl_xor_ll(st2, 32);
            if (q == 0 && co < a.cout) {
                double* st = a.stats + static_cast<size_t>((blockIdx.x + 7 * blockIdx.z + wave) & (kStatSlots - 1)) * a.stats_stride;
                fx_add_int(st + 2 * co, st1);
                fx_add_int(st + 2 * co + 1, st2);
            }
        }
    }
}

// ==========================================================================================
// The last layer's PitchClass2PitchClass stack (cin <= 16 -> 16 -> 16 -> 16 channels on a 12 x T map) + the time pooling that
// follows it (models.py:393, 396) in ONE launch: one workgroup of 16 waves per clip keeps the maps in LDS as channels-last
// split planes (two ping-pong maps of 12 x (T + 8) x 16 channels), converts the NCHW f32 input while loading, runs the convs with
// the MFMA loop of conv_pc_bf16_kernel<1, *> (same arithmetic in the same order), and the last epilogue takes the max over frame
// pairs in registers and writes the pooled features twice: NCHW f32 (taps, f32 heads) and channels-last planes (bf16 heads).
// Measured on the per-conv launches (phases switched off one at a time): 13 us of launch + epilogue and 4 us of patch load per
// 21 us of MFMA loop, three times, plus the conversion and pooling passes.
// ==========================================================================================
struct Pc2pcFusedArgs {
    const float* src;             // NCHW f32 [clip][cin][12][T]
    long long src_clip_stride;
    int cin, n_conv;
    const uint4* bfrag[4];        // per conv: [48 k-steps][hi|lo][64 lanes] x 8 bf16
    const float* bias[4];         // per conv: [16]
    float* pooled;                // NCHW f32 [clip][16][12][T / 2]
    unsigned short* fh;           // channels-last planes of the pooled features [clip][12][T / 2][16], or null
    unsigned short* fl;
    int T, Tp;                    // T % 4 == 0, 12 * T <= 1024; Tp = T + 8
};

__global__ __launch_bounds__(1024) void pc2pc_fused_kernel(Pc2pcFusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint4 lds4[];
    const int clip = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r16 = lane & 15, q = lane >> 4;
    const int T = a.T, Tp = a.Tp;
    constexpr int MT = 4;
    const int n16 = 12 * Tp * 2;                              // uint4 per plane
    uint4* const map0 = lds4;                                 // two maps, each: hi plane, lo plane
    uint4* const map1 = lds4 + 2 * n16;
    // weight fragments of one kernel row dy (4 k-steps x (hi | lo) x 64 lanes = 8 KB), double-buffered: the workgroup fetches every
    // fragment ONCE and its 16 waves read it from here.  Streamed per wave from L1 they were 16 x 2 KB per k-step -- two thirds of the
    // L1's 64 B / clk next to 768 cycles of MFMAs -- and 49 of the launch's 130 us (measured by leaving the loads out)
    uint4* const wring = lds4 + 4 * n16;
    constexpr int kRow = 4 * 2 * 64;                          // uint4 per kernel row
    for (int i = threadIdx.x; i < 4 * n16; i += 1024) lds4[i] = make_uint4(0, 0, 0, 0);
    // ring chunk c = (p = c / 3, dq = c % 3): the fragments of k-step pair p of kernel rows 4 dq .. 4 dq + 3, [i][hi | lo][64 lanes];
    // thread x < 512 copies fragment (i = x >> 7, h = (x >> 6) & 1, lane x & 63) from the packed order [dy][p][h][lane]
    const int wsrc0 = (((threadIdx.x >> 7) * 4) * 2 + ((threadIdx.x >> 6) & 1)) * 64 + (threadIdx.x & 63);
    auto wsrc = [&](int p, int dq) { return wsrc0 + (16 * dq * 2 + p * 2) * 64; };
    if (threadIdx.x < kRow) wring[threadIdx.x] = a.bfrag[0][wsrc(0, 0)];
    __syncthreads();
    {   // input: NCHW f32 -> split channels-last, frame t at patch position t + 3
        const float* src = a.src + clip * a.src_clip_stride;
        for (int i = threadIdx.x; i < 12 * T * 2; i += 1024) {
            const int half = i & 1, pos = i >> 1;
            const int y = pos / T, t = pos - y * T;
            unsigned int hi[4] = {0, 0, 0, 0}, lo[4] = {0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int c = half * 8 + k;
                const float v = c < a.cin ? src[(static_cast<long long>(c) * 12 + y) * T + t] : 0.f;
                const unsigned int hb = bf16_bits(v);
                hi[k >> 1] |= hb << (16 * (k & 1));
                lo[k >> 1] |= bf16_bits(v - __uint_as_float(hb << 16)) << (16 * (k & 1));
            }
            const int idx = ((y * Tp + t + 3) << 1) + half;
            map0[idx] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
            map0[n16 + idx] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
        }
    }
    __syncthreads();
    // Wave = (16-frame tile tt, four output rows y0 .. y0 + 3).  Output row y at kernel row dy reads input row (y + dy) % 12, so the
    // A fragment tile mt used at dy is the one tile mt + 1 used at dy - 1: per kernel row ONE new fragment pair is read from LDS, not
    // four (the multiply loop was bound by its A-fragment reads: 10 ds_read_b128 per 12 MFMAs, 1 280 LDS cycles per CU next to 768
    // matrix cycles; now 4 per 12).  The k-step pair p is the outer loop -- the four slots hold one p -- and the weight ring moves
    // in chunks of (p, four kernel rows).
    const int n_tt = (T + 15) >> 4;
    const bool active = wave < 3 * n_tt;
    const int tt = wave / 3, y0 = 4 * (wave - 3 * tt);
    const int t0 = 16 * tt;
    typedef float f32x4c __attribute__((ext_vector_type(4)));
    const int dxq = q >> 1, half = q & 1;
    const int lane_off = ((t0 + r16 + dxq) << 1) + half;      // (frames past the row's end read the next row: those output rows are never stored)
    const int lane_off3 = ((t0 + r16 + 6) << 1) + half;
    // fragment s of k-step pair p: input row y0 + s for p < 3; p == 3 holds the paired seventh taps (conv_pc_bf16_kernel): the first tap
    // slot's lanes read row y0 + s, the second's row y0 + s + 1, both at frame + 6, and it exists for even kernel rows only
    auto frag_off = [&](int s, int p) {
        int row = y0 + s + (p == 3 ? dxq : 0);
        row -= row >= 12 ? 12 : 0;
        return ((row * Tp) << 1) + (p == 3 ? lane_off3 : lane_off + 4 * p);
    };
    const int Tf = T / 2;
    float* const pooled_c = a.pooled + static_cast<long long>(clip) * 16 * 12 * Tf;
    unsigned short* const fh_c = a.fh ? a.fh + static_cast<long long>(clip) * 12 * Tf * 16 : nullptr;
    unsigned short* const fl_c = a.fh ? a.fl + static_cast<long long>(clip) * 12 * Tf * 16 : nullptr;
    for (int j = 0; j < a.n_conv; ++j) {
        const uint4* const pH = (j & 1) ? map1 : map0;
        const uint4* const pL = pH + n16;
        unsigned short* const oH = reinterpret_cast<unsigned short*>((j & 1) ? map0 : map1);
        unsigned short* const oL = oH + n16 * 8;
        const bool last = j == a.n_conv - 1;
        // (selected, not indexed: a dynamically indexed member array sends the whole argument struct through scratch memory,
        //  and every load from it then drains the weight prefetch)
        const uint4* const bfrag_j = j == 0 ? a.bfrag[0] : (j == 1 ? a.bfrag[1] : (j == 2 ? a.bfrag[2] : a.bfrag[3]));
        const float* const bias_j = j == 0 ? a.bias[0] : (j == 1 ? a.bias[1] : (j == 2 ? a.bias[2] : a.bias[3]));
        const uint4* const bfrag_n = j == 0 ? a.bfrag[1] : (j == 1 ? a.bfrag[2] : a.bfrag[3]);     // the next convolution's (if any)
        {
            f32x4c acc[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4c{0.f, 0.f, 0.f, 0.f};
            bf16x8c fh[4], fl[4];                             // slot s: input row y0 + (fragment index == s mod 4)
            int c = 0;
            // one k-step pair p: three ring chunks of four kernel rows.  PAIR (p == 3): the paired seventh taps -- even kernel rows only,
            // the fragments move two slots per step
            auto run_p = [&](auto pair_c, const int p) __attribute__((always_inline)) {
                constexpr bool PAIR = decltype(pair_c)::value;
                if (active) {
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const int off = frag_off(s, PAIR ? 3 : p);
                        fh[s] = __builtin_bit_cast(bf16x8c, pH[off]);
                        fl[s] = __builtin_bit_cast(bf16x8c, pL[off]);
                    }
                }
#pragma unroll 1
                for (int dq = 0; dq < 3; ++dq, ++c) {
                    // the next chunk's fragments travel to registers during this chunk's MFMAs and into the other half of the ring after them
                    uint4 wpre = make_uint4(0, 0, 0, 0);
                    const bool fetch = threadIdx.x < kRow && (c + 1 < 12 || j + 1 < a.n_conv);
                    if (fetch) wpre = c + 1 < 12 ? bfrag_j[dq < 2 ? wsrc(p, dq + 1) : wsrc(p + 1, 0)] : bfrag_n[wsrc(0, 0)];
                    const uint4* const wr = wring + (c & 1) * kRow + lane;
                    if (active) {
#pragma unroll
                        for (int i = 0; i < 4; i += PAIR ? 2 : 1) {   // kernel row dy = 4 dq + i: tile mt reads fragment dy + mt = slot (mt + i) & 3
                            const bf16x8c bh = __builtin_bit_cast(bf16x8c, wr[(i * 2 + 0) * 64]), bl = __builtin_bit_cast(bf16x8c, wr[(i * 2 + 1) * 64]);
#pragma unroll
                            for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh[(mt + i) & 3], bh, acc[mt], 0, 0, 0);
#pragma unroll
                            for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fl[(mt + i) & 3], bh, acc[mt], 0, 0, 0);
#pragma unroll
                            for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh[(mt + i) & 3], bl, acc[mt], 0, 0, 0);
                            // slot i (tile 0's; PAIR: and slot i + 1, tile 1's) is free: the fragment(s) the next step's last tile(s) need
#pragma unroll
                            for (int u = 0; u < (PAIR ? 2 : 1); ++u) {
                                const int s_new = 4 * dq + i + 4 + u;
                                if (s_new < 12 + MT - (PAIR ? 2 : 1)) {
                                    const int off = frag_off(s_new, PAIR ? 3 : p);
                                    fh[i + u] = __builtin_bit_cast(bf16x8c, pH[off]);
                                    fl[i + u] = __builtin_bit_cast(bf16x8c, pL[off]);
                                }
                            }
                        }
                    }
                    // (12 chunks: chunk 11 read half 1, so the next convolution's chunk 0 lands in half 0, where it expects it)
                    if (fetch) wring[((c + 1) & 1) * kRow + threadIdx.x] = wpre;
                    if (c + 1 < 12) __syncthreads();          // (after the last chunk the barrier at the end of the convolution serves)
                }
            };
#pragma unroll 1
            for (int p = 0; p < 3; ++p) run_p(std::false_type{}, p);
            run_p(std::true_type{}, 3);
            if (active) {
            // ---- epilogue: D[row m = 4q + i][col = co]; T % 4 == 0: a lane's four positions are frames t0 + 4q .. + 3 of row y0 + mt ----
            const int co = r16;
            const float bias = bias_j[co];
            const int tq0 = t0 + 4 * q;
            if (tq0 < T) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                    const int y = y0 + mt;
                    float v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float x = acc[mt][i] + bias;
                        v[i] = x > 0.f ? x : x * kSlope;
                    }
                    if (!last) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int e = ((((y * Tp + tq0 + i + 3) << 1) + (co >> 3)) << 3) + (co & 7);
                            const unsigned int hb = bf16_bits(v[i]);
                            oH[e] = static_cast<unsigned short>(hb);
                            oL[e] = static_cast<unsigned short>(bf16_bits(v[i] - __uint_as_float(hb << 16)));
                        }
                    } else {                                             // nn.MaxPool2d((1, 2)), models.py:396
#pragma unroll
                        for (int i = 0; i < 4; i += 2) {
                            const float pv = fmaxf(v[i], v[i + 1]);
                            const int tq = (tq0 + i) >> 1;
                            pooled_c[(co * 12 + y) * Tf + tq] = pv;      // (uniform clip base + 32-bit offsets: few address registers)
                            if (a.fh) {
                                const int e = (y * Tf + tq) * 16 + co;
                                const unsigned int hb = bf16_bits(pv);
                                fh_c[e] = static_cast<unsigned short>(hb);
                                fl_c[e] = static_cast<unsigned short>(bf16_bits(pv - __uint_as_float(hb << 16)));
                            }
                        }
                    }
            }
            }
            }
        }
        __syncthreads();
    }
}

// ==========================================================================================
// Last convolution of the key / tonic heads: 32 channels -> ONE map, 12 x 7 over circular pitch classes, valid in time
// (models.py:730-731).  With a single output channel the N dimension of the MFMA is filled with 16 output FRAMES (Toeplitz in
// time), the K dimension with the 32 input channels of one tap -- channels-last split planes [clip][12][T_in][32] again:
//   m = (frame block jb, pitch class y)     A[m][k = ci] = X[(y + dy) mod 12][16 jb + dxe][ci]
//   n = tau (frame within the block)        B[k][n] = w[ci][dy][dxe - tau]   (zero outside the 7 taps)
//   k-step = (dy, dxe): 12 x 22 = 264, split over the 8 waves of the workgroup (one workgroup per clip and head); the partial
//   tiles are reduced through LDS.  The implicit-GEMM f32 kernel needed 100 us per launch for these 0.8 M MACs per clip.
// ==========================================================================================
struct Head1BfArgs {
    const unsigned short* xh[3];  // per head (key, tonic, genre): [clip][12][T_in][32]
    const unsigned short* xl[3];
    const uint4* bfrag[3];        // per head: [KH * 22 k-steps][hi|lo][64 lanes]
    const float* bias[3];
    float* dst[3];                // per head: [clip][H_out][T_out]
    int KH[3], H_out[3], circular[3];   // key / tonic: 12 rows over circular pitch classes; genre: 2 rows, valid (11 output rows)
    int T_in, T_out, Tp, JB;
    // masked temporal mean + sigmoid of the finished map (models.py:754-804) in the same launch: pout[head] = [clip][H_out] or null
    float* pout[3];
    const long long* seq;
    int n_pool_layers, tp, shrink, max_pool, clip0;
    int fin_off;                  // float offset of the finished map's LDS copy [H_out][T_out] (behind the reduction buffer)
};

constexpr int kHead1MT = 4;       // M-tiles (12 * JB positions / 16): T_out <= 80

__global__ __launch_bounds__(512) void conv_head1_bf16_kernel(Head1BfArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint4 lds4[];
    const int clip = blockIdx.x, head = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r16 = lane & 15, q = lane >> 4;
    const int Tp = a.Tp;
    uint4* const pH = lds4;                                  // [12][Tp][4 groups of 8 channels]
    uint4* const pL = lds4 + 12 * Tp * 4;
    float* const red = reinterpret_cast<float*>(lds4);       // [8 waves][kHead1MT][4][64]: reuses the patch's bytes after the multiply loop
    {
        const long long cbase = static_cast<long long>(clip) * 12 * a.T_in * 4;
        const uint4* gh = reinterpret_cast<const uint4*>(a.xh[head]) + cbase;
        const uint4* gl = reinterpret_cast<const uint4*>(a.xl[head]) + cbase;
        const int n16 = 12 * Tp * 4;
        for (int i = threadIdx.x; i < n16; i += blockDim.x) {
            const int grp = i & 3, pos = i >> 2;
            const int row = pos / Tp, f = pos - row * Tp;
            uint4 vh = make_uint4(0, 0, 0, 0), vl = make_uint4(0, 0, 0, 0);
            if (f < a.T_in) {
                const long long g = (static_cast<long long>(row) * a.T_in + f) * 4 + grp;
                vh = gh[g]; vl = gl[g];
            }
            pH[i] = vh; pL[i] = vl;
        }
    }
    __syncthreads();
    const int HO = a.H_out[head], circ = a.circular[head], nks = a.KH[head] * 22;
    const int Mtot = HO * a.JB;
    const int mtiles = (Mtot + 15) / 16;
    int ay[kHead1MT], af[kHead1MT];
#pragma unroll
    for (int mt = 0; mt < kHead1MT; ++mt) {
        int m = mt * 16 + r16;
        if (m >= Mtot) m = Mtot - 1;
        const int jb = m / HO;
        ay[mt] = m - jb * HO;
        af[mt] = 16 * jb;
    }
    typedef float f32x4c __attribute__((ext_vector_type(4)));
    f32x4c acc[kHead1MT];
#pragma unroll
    for (int mt = 0; mt < kHead1MT; ++mt) acc[mt] = f32x4c{0.f, 0.f, 0.f, 0.f};
    const uint4* __restrict__ bg = a.bfrag[head] + lane;
    // weight fragments three k-steps ahead (they come from L2: fetched at their use, every k-step paid a full round trip)
    constexpr int PF = 3;
    uint4 nbh[PF], nbl[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) {
        const int kn = wave + 8 * i < nks ? wave + 8 * i : wave;
        nbh[i] = bg[(2 * kn + 0) * 64]; nbl[i] = bg[(2 * kn + 1) * 64];
    }
    for (int ks = wave; ks < nks; ks += 8) {
        const int dy = ks / 22, dxe = ks - dy * 22;
        const bf16x8c bh = __builtin_bit_cast(bf16x8c, nbh[0]);
        const bf16x8c bl = __builtin_bit_cast(bf16x8c, nbl[0]);
#pragma unroll
        for (int i = 0; i + 1 < PF; ++i) { nbh[i] = nbh[i + 1]; nbl[i] = nbl[i + 1]; }
        {
            const int kn = ks + 8 * PF < nks ? ks + 8 * PF : ks;
            nbh[PF - 1] = bg[(2 * kn + 0) * 64]; nbl[PF - 1] = bg[(2 * kn + 1) * 64];
        }
#pragma unroll
        for (int mt = 0; mt < kHead1MT; ++mt) {
            if (mt < mtiles) {
                int row = ay[mt] + dy;
                row -= (circ && row >= 12) ? 12 : 0;
                const int ad = ((row * Tp + af[mt] + dxe) << 2) + q;
                const bf16x8c ah = __builtin_bit_cast(bf16x8c, pH[ad]);
                const bf16x8c al = __builtin_bit_cast(bf16x8c, pL[ad]);
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc[mt], 0, 0, 0);
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc[mt], 0, 0, 0);
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc[mt], 0, 0, 0);
            }
        }
    }
    __syncthreads();                                         // every wave is done with the patch
#pragma unroll
    for (int mt = 0; mt < kHead1MT; ++mt)
#pragma unroll
        for (int i = 0; i < 4; ++i) red[((wave * kHead1MT + mt) * 4 + i) * 64 + lane] = acc[mt][i];
    __syncthreads();
    // wave w finishes M-tile w: D[row m = 4q + i][col tau]
    if (wave < mtiles) {
        const float bias = a.bias[head][0];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v = bias;
#pragma unroll
            for (int w = 0; w < 8; ++w) v += red[((w * kHead1MT + wave) * 4 + i) * 64 + lane];
            const int m = wave * 16 + 4 * q + i;
            const int jb = m / HO, y = m - jb * HO;
            const int t = 16 * jb + r16;
            if (m < Mtot && t < a.T_out) {
                a.dst[head][(static_cast<long long>(clip) * HO + y) * a.T_out + t] = v;
                if (a.pout[head]) reinterpret_cast<float*>(lds4)[a.fin_off + y * a.T_out + t] = v;
            }
        }
    }
    if (a.pout[head] == nullptr) return;
    __syncthreads();
    if (threadIdx.x < HO) {                                   // one lane per output row, as head_pool_kernel
        bool use_max = a.max_pool != 0;
        const int L = pool_frames(a.seq, clip, a.T_out, a.n_pool_layers, a.tp, a.shrink, a.clip0, &use_max);
        const float* m = reinterpret_cast<const float*>(lds4) + a.fin_off + threadIdx.x * a.T_out;
        float v;
        if (use_max) {
            v = -INFINITY;
            for (int t = 0; t < L; ++t) v = fmaxf(v, m[t]);
        } else {
            float sum = 0.f;
            for (int t = 0; t < L; ++t) sum += m[t];
            v = sum / static_cast<float>(L > 0 ? L : 0);              // empty slice -> NaN, as torch.mean
        }
        if (head == 0) v = 1.f / (1.f + expf(-v));                    // self.sig(key_out), models.py:802
        a.pout[head][clip * HO + threadIdx.x] = v;
    }
}

// weight fragments of conv_head1_bf16_kernel from the VALU-layout eval pack [ci][12][7] (cout == 1): one thread per (k-step, lane)
__global__ void pack_head1_bf16_kernel(const float* __restrict__ w, uint4* __restrict__ out, int cin, int KH) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= KH * 22 * 64) return;
    const int lane = i & 63, ks = i >> 6;
    const int dy = ks / 22, dxe = ks - dy * 22;
    const int tau = lane & 15, qq = lane >> 4;
    const int dx = dxe - tau;
    unsigned int hi[4] = {0, 0, 0, 0}, lo[4] = {0, 0, 0, 0};
    for (int e = 0; e < 8; ++e) {
        const int ci = 8 * qq + e;
        float v = 0.f;
        if (dx >= 0 && dx < 7 && ci < cin) v = w[(ci * KH + dy) * 7 + dx];
        const unsigned int hb = bf16_bits(v);
        hi[e >> 1] |= hb << (16 * (e & 1));
        lo[e >> 1] |= bf16_bits(v - __uint_as_float(hb << 16)) << (16 * (e & 1));
    }
    out[(2 * ks + 0) * 64 + lane] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    out[(2 * ks + 1) * 64 + lane] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
}

// weight fragments of conv_pc_bf16_kernel from the VALU-layout eval pack [co group of CO][ci][12][7][CO]: one thread per (k-step, N-tile, lane)
__global__ void pack_pc_bf16_kernel(const float* __restrict__ w, uint4* __restrict__ out, int cin, int cout, int CO, int NT, int KH) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= KH * 4 * NT * 64) return;
    const int lane = i & 63, nt = (i >> 6) % NT, ks = i / (64 * NT);
    const int dy = ks >> 2, p = ks & 3;
    const int co = nt * 16 + (lane & 15), qq = lane >> 4;
    const int c8 = 8 * (qq & 1);
    int dx = 2 * p + (qq >> 1), dyw = dy;
    if (p == 3 && !(KH & 1)) {           // paired seventh taps (see conv_pc_bf16_kernel): [tap 6 of row dy | tap 6 of row dy + 1] at even dy
        dx = (dy & 1) ? 7 : 6;
        dyw = dy + (qq >> 1);
    }
    unsigned int hi[4] = {0, 0, 0, 0}, lo[4] = {0, 0, 0, 0};
    for (int e = 0; e < 8; ++e) {
        const int ci = c8 + e;
        float v = 0.f;
        if (dx < 7 && ci < cin && co < cout) v = w[((((co / CO) * cin + ci) * KH + dyw) * 7 + dx) * CO + (co % CO)];
        const unsigned int hb = bf16_bits(v);
        const unsigned int lb = bf16_bits(v - __uint_as_float(hb << 16));
        hi[e >> 1] |= hb << (16 * (e & 1));
        lo[e >> 1] |= lb << (16 * (e & 1));
    }
    out[((ks * NT + nt) * 2 + 0) * 64 + lane] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    out[((ks * NT + nt) * 2 + 1) * 64 + lane] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
}

// NCHW f32 [clip][C][12][T] (C <= 16) -> channels-last split planes [clip][12][T][16], channels >= C zero
__global__ void nchw_to_cl16_kernel(const float* __restrict__ src, long long src_clip_stride, int C, int T, unsigned short* __restrict__ xh,
                                    unsigned short* __restrict__ xl, long long npos) {
    const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;      // (clip, row, t)
    if (i >= npos) return;
    const int t = static_cast<int>(i % T);
    const long long r = i / T;
    const int y = static_cast<int>(r % 12);
    const long long clip = r / 12;
    const float* s = src + clip * src_clip_stride + static_cast<long long>(y) * T + t;
    unsigned int hi[8], lo[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { hi[k] = 0; lo[k] = 0; }
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const float v = c < C ? s[static_cast<long long>(c) * 12 * T] : 0.f;
        const unsigned int hb = bf16_bits(v);
        hi[c >> 1] |= hb << (16 * (c & 1));
        lo[c >> 1] |= bf16_bits(v - __uint_as_float(hb << 16)) << (16 * (c & 1));
    }
    uint4* oh = reinterpret_cast<uint4*>(xh) + i * 2;
    uint4* ol = reinterpret_cast<uint4*>(xl) + i * 2;
    oh[0] = make_uint4(hi[0], hi[1], hi[2], hi[3]); oh[1] = make_uint4(hi[4], hi[5], hi[6], hi[7]);
    ol[0] = make_uint4(lo[0], lo[1], lo[2], lo[3]); ol[1] = make_uint4(lo[4], lo[5], lo[6], lo[7]);
}

// training: NCHW f32 [clip][C][12][T] (C <= 16), with the pending BatchNorm + LeakyReLU of its producer applied (aff: [C][3] scale, shift,
// negative slope; or null) -> channels-last planes [clip][12][T][16] of f16 hi and f16 lo * 2^11 (conv_pc_bf16_kernel<.., F16X3>)
// amax (nullable, data gradients): bits of the tensor's largest |value|; every value is multiplied by f16_weight_scale(max) (a power of two:
// exact) so that gradients of 1e-6..1e-9 use f16's normal range; the reading convolution divides it out (PcBfArgs::in_amax).
__global__ void nchw_to_cl16_f16x2_kernel(const float* __restrict__ src, long long src_clip_stride, int C, int T, const float* __restrict__ aff,
                                          unsigned short* __restrict__ xh, unsigned short* __restrict__ xl, long long npos,
                                          const unsigned int* __restrict__ amax) {
    const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;      // (clip, row, t)
    const float mul = amax ? f16_weight_scale(amax_load(amax)) : 1.f;        // (before the bounds check: every lane of the wave takes part)
    if (i >= npos) return;
    const int t = static_cast<int>(i % T);
    const long long r = i / T;
    const int y = static_cast<int>(r % 12);
    const long long clip = r / 12;
    const float* s = src + clip * src_clip_stride + static_cast<long long>(y) * T + t;
    unsigned int hi[8], lo[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { hi[k] = 0; lo[k] = 0; }
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        float v = 0.f;
        if (c < C) {
            v = s[static_cast<long long>(c) * 12 * T] * mul;
            if (aff) { v = fmaf(v, aff[3 * c], aff[3 * c + 1]); v = v > 0.f ? v : v * aff[3 * c + 2]; }
        }
        const _Float16 h = static_cast<_Float16>(v);
        hi[c >> 1] |= static_cast<unsigned int>(__builtin_bit_cast(unsigned short, h)) << (16 * (c & 1));
        lo[c >> 1] |= f16_bits((v - static_cast<float>(h)) * kP2pLoScale) << (16 * (c & 1));
    }
    uint4* oh = reinterpret_cast<uint4*>(xh) + i * 2;
    uint4* ol = reinterpret_cast<uint4*>(xl) + i * 2;
    oh[0] = make_uint4(hi[0], hi[1], hi[2], hi[3]); oh[1] = make_uint4(hi[4], hi[5], hi[6], hi[7]);
    ol[0] = make_uint4(lo[0], lo[1], lo[2], lo[3]); ol[1] = make_uint4(lo[4], lo[5], lo[6], lo[7]);
}

// f16 hi / lo * 2^11 weight fragments of conv_pc_bf16_kernel<.., F16X3> (same order as pack_pc_bf16_kernel), every output channel scaled
// into [2^13, 2^14); the cout inverse scales follow the fragments.  From the VALU-layout TRAINING pack [co group of CO][ci][KH][7][CO].
// dy_rot: fragment row dy holds kernel row (dy + dy_rot) mod KH -- the data gradient of a convolution over 12 circular rows whose kernel
// starts AT the output row (py = 0) starts 11 rows before it, i.e. (mod 12) one row after: rotating the rows by 11 lets the same
// py = 0 kernel compute it.
// Inverse power-of-two scale of every output channel of a pack [group][ci][dy][dx][CO] (f16_weight_scale of its largest |w|), written
// behind the fragments pack_pc_f16x3_kernel fills afterwards.  One workgroup per output channel.
// Several packs per launch (round 3: a training step re-packs ~14 convolutions after every optimizer step, two tiny launches each -- 28 dependent
// launches, 0.39 ms of a 7 ms step, were launch latency): blockIdx.y picks the job, the jobs travel in the kernel argument.
struct PcPackJob { const float* w; uint4* out; int cin, cout, CO, NT, KH, dy_rot, ci_off; };      // (the inverse scales live behind the fragments: out + KH * 4 * NT * 2 * 64)
constexpr int kMaxPackJobs = 24;
struct PcPackJobs { int n; PcPackJob j[kMaxPackJobs]; };
struct P2pRawJob { const float* w; uint4* out; int cin_fwd, cout_fwd, transpose_flip; };
struct P2pRawJobs { int n; P2pRawJob j[kMaxPackJobs]; };

__device__ __forceinline__ void pc_weight_scale_body(const float* __restrict__ w, float* __restrict__ inv_scale, int cin, int cout, int CO, int KH, int bx) {
    __shared__ float red[4];
    const int co = bx;
    const int g = co / CO, c = co - g * CO;
    float m = 0.f;
    for (int r = threadIdx.x; r < cin * KH * 7; r += blockDim.x) m = fmaxf(m, fabsf(w[(static_cast<long long>(g) * cin * KH * 7 + r) * CO + c]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) inv_scale[co] = 1.f / f16_weight_scale(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
}
__global__ __launch_bounds__(256) void pc_weight_scale_kernel(const float* __restrict__ w, float* __restrict__ inv_scale, int cin, int cout, int CO, int KH) {
    pc_weight_scale_body(w, inv_scale, cin, cout, CO, KH, blockIdx.x);
}
__global__ __launch_bounds__(256) void pc_weight_scale_jobs_kernel(PcPackJobs js) {
    const PcPackJob& J = js.j[blockIdx.y];
    if (static_cast<int>(blockIdx.x) >= J.cout) return;                 // (uniform per block)
    pc_weight_scale_body(J.w, reinterpret_cast<float*>(J.out + J.KH * 4 * J.NT * 2 * 64), J.cin, J.cout, J.CO, J.KH, blockIdx.x);
}

// ci_off: the fragments take input channels [ci_off, ci_off + 16) of the pack (a 32-channel data gradient runs as two halves).
__device__ __forceinline__ void pack_pc_f16x3_body(const float* __restrict__ w, uint4* __restrict__ out, int cin, int cout, int CO, int NT, int KH, int dy_rot,
                                                   int ci_off, int bx) {
    // (the inverse channel scales behind the fragments were written by pc_weight_scale_kernel: every block of this kernel scanning all
    // weights for the channel maxima cost 28 us per pack, 0.5 ms per training step)
    const int i = bx * blockDim.x + threadIdx.x;
    if (i >= KH * 4 * NT * 64) return;
    const int lane = i & 63, nt = (i >> 6) % NT, ks = i / (64 * NT);
    const int dy = ks >> 2, p = ks & 3;
    const int co = nt * 16 + (lane & 15), qq = lane >> 4;
    const int c8 = 8 * (qq & 1);
    int dx = 2 * p + (qq >> 1), dyw = dy;
    if (p == 3 && !(KH & 1)) {           // paired seventh taps, as pack_pc_bf16_kernel
        dx = (dy & 1) ? 7 : 6;
        dyw = dy + (qq >> 1);
    }
    const float sc = co < cout ? 1.f / reinterpret_cast<const float*>(out + KH * 4 * NT * 2 * 64)[co] : 1.f;      // a power of two: exact
    unsigned int hi[4] = {0, 0, 0, 0}, lo[4] = {0, 0, 0, 0};
    for (int e = 0; e < 8; ++e) {
        const int ci = ci_off + c8 + e;
        float v = 0.f;
        if (dx < 7 && ci < cin && co < cout) v = sc * w[((((co / CO) * cin + ci) * KH + (dyw + dy_rot) % KH) * 7 + dx) * CO + (co % CO)];
        const _Float16 hv = static_cast<_Float16>(v);
        hi[e >> 1] |= static_cast<unsigned int>(__builtin_bit_cast(unsigned short, hv)) << (16 * (e & 1));
        lo[e >> 1] |= f16_bits((v - static_cast<float>(hv)) * kP2pLoScale) << (16 * (e & 1));
    }
    out[((ks * NT + nt) * 2 + 0) * 64 + lane] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    out[((ks * NT + nt) * 2 + 1) * 64 + lane] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
}
__global__ void pack_pc_f16x3_kernel(const float* __restrict__ w, uint4* __restrict__ out, int cin, int cout, int CO, int NT, int KH, int dy_rot,
                                     int ci_off) {
    pack_pc_f16x3_body(w, out, cin, cout, CO, NT, KH, dy_rot, ci_off, blockIdx.x);
}
__global__ void pack_pc_f16x3_jobs_kernel(PcPackJobs js) {
    const PcPackJob& J = js.j[blockIdx.y];
    pack_pc_f16x3_body(J.w, J.out, J.cin, J.cout, J.CO, J.NT, J.KH, J.dy_rot, J.ci_off, blockIdx.x);
}

// debug taps: channels-last split-bf16 planes (xl != null) or one f16 plane (xl == null) [clip][H][T][C] -> NCHW f32 [clip][C][H][T]
__global__ void cl_to_nchw_kernel(const unsigned short* __restrict__ xh, const unsigned short* __restrict__ xl, float* __restrict__ out, int C, int H,
                                  int T, long long total) {
    const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int t = static_cast<int>(i % T);
    long long r = i / T;
    const int y = static_cast<int>(r % H);
    r /= H;
    const int c = static_cast<int>(r % C);
    const long long clip = r / C;
    const long long src = ((clip * H + y) * T + t) * C + c;
    out[i] = xl ? __uint_as_float(static_cast<unsigned int>(xh[src]) << 16) + __uint_as_float(static_cast<unsigned int>(xl[src]) << 16)
                : static_cast<float>(__builtin_bit_cast(_Float16, xh[src]));
}

// B fragments of conv_p2p_f16_kernel from the VALU-layout eval pack [ci < cin][dy][dx][8 co] (BatchNorm already folded; input
// channels >= cin get zero weights): f16 hi = rn(w), f16 lo = rn((w - hi) * 2^11);
// one thread per (k-step, lane, element).

// max |w| of each of the 8 output channels of a pack [n_k][8 co], by the whole block (bit patterns of non-negative floats order as ints)
__device__ __forceinline__ void channel_absmax8(const float* __restrict__ w, int n_k, int* smax) {
    if (threadIdx.x < 8) smax[threadIdx.x] = 0;
    __syncthreads();
    for (int k = threadIdx.x; k < n_k * 8; k += blockDim.x) atomicMax(&smax[k & 7], __float_as_int(fabsf(w[k])));
    __syncthreads();
}
// ... of a pack [n_k][4 co]
__device__ __forceinline__ void channel_absmax4(const float* __restrict__ w, int n_k, int* smax) {
    if (threadIdx.x < 4) smax[threadIdx.x] = 0;
    __syncthreads();
    for (int k = threadIdx.x; k < n_k * 4; k += blockDim.x) atomicMax(&smax[k & 3], __float_as_int(fabsf(w[k])));
    __syncthreads();
}

__global__ void pack_p2p_f16_kernel(const float* __restrict__ w, uint4* __restrict__ out, int cin) {
    __shared__ int smax[8];
    channel_absmax8(w, cin * 49, smax);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;      // (ks, lane)
    if (i >= 14 * 64) return;
    const int ks = i / 64, lane = i - ks * 64;
    const int dy = ks >> 1, h = ks & 1;
    const int n = lane & 15, qq = lane >> 4;
    const int tau = n >> 3, co = n & 7;
    const int dx = 4 * h + qq - tau;
    const float sc = f16_weight_scale(__int_as_float(smax[co]));
    if (ks == 0 && qq == 0 && tau == 0) reinterpret_cast<float*>(out + kP2pFragScale)[co] = 1.f / sc;
    unsigned int hi[4] = {0, 0, 0, 0}, lo[4] = {0, 0, 0, 0};
    for (int ci = 0; ci < 8; ++ci) {
        float v = 0.f;
        if (dx >= 0 && dx < 7 && ci < cin) v = sc * w[((ci * 7 + dy) * 7 + dx) * 8 + co];
        // cin <= 5 (the stack's FIRST conv: channel 0 is the raw pitch stream = the log-CQT): the three idle channel slots make channel 0's
        // product f32-equivalent at no cost -- slot 5 = channel 0's weight again (meets the activation's LOW half), slot 6 = the weight's own
        // low half, true scale (meets the activation's high half): x w ~ xh wh + xl wh + xh wl (kP2pSplit0; loaders: the IN_NCHW forms).
        if (cin <= kP2pSplit0 && (ci == 5 || ci == 6) && dx >= 0 && dx < 7) {
            const float w0 = sc * w[((0 * 7 + dy) * 7 + dx) * 8 + co];
            const float w0h = static_cast<float>(static_cast<_Float16>(w0));
            v = ci == 5 ? w0h : w0 - w0h;
        }
        const _Float16 hv = static_cast<_Float16>(v);
        const unsigned int hb = __builtin_bit_cast(unsigned short, hv);
        const unsigned int lb = f16_bits((v - static_cast<float>(hv)) * kP2pLoScale);
        hi[ci >> 1] |= hb << (16 * (ci & 1));
        lo[ci >> 1] |= lb << (16 * (ci & 1));
    }
    out[(2 * ks + 0) * 64 + lane] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    out[(2 * ks + 1) * 64 + lane] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
}

// B fragments of the semitone conv fused into conv_p2p_f16_ps_kernel<2, ...>, from its eval pack [ci < 8][3 dy][3 dx][8 co]:
// k-step = dy, k = (position qq, ci), n = (tau, co), tap dx = qq - tau (the 4th position of either frame carries zero weights)
__global__ void pack_semi_f16_kernel(const float* __restrict__ w, uint4* __restrict__ out) {
    __shared__ int smax[8];
    channel_absmax8(w, 8 * 9, smax);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;      // (dy, lane)
    if (i >= 3 * 64) return;
    const int dy = i / 64, lane = i - dy * 64;
    const int n = lane & 15, qq = lane >> 4;
    const int tau = n >> 3, co = n & 7;
    const int dx = qq - tau;
    const float sc = f16_weight_scale(__int_as_float(smax[co]));
    if (dy == 0 && qq == 0 && tau == 0) reinterpret_cast<float*>(out + 6 * 64)[co] = 1.f / sc;
    unsigned int hi[4] = {0, 0, 0, 0}, lo[4] = {0, 0, 0, 0};
    for (int ci = 0; ci < 8; ++ci) {
        float v = 0.f;
        if (dx >= 0 && dx < 3) v = sc * w[((ci * 3 + dy) * 3 + dx) * 8 + co];
        const _Float16 hv = static_cast<_Float16>(v);
        const unsigned int hb = __builtin_bit_cast(unsigned short, hv);
        const unsigned int lb = f16_bits((v - static_cast<float>(hv)) * kP2pLoScale);
        hi[ci >> 1] |= hb << (16 * (ci & 1));
        lo[ci >> 1] |= lb << (16 * (ci & 1));
    }
    out[(2 * dy + 0) * 64 + lane] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    out[(2 * dy + 1) * 64 + lane] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
}

// Weight fragments of conv_p2p_f16x3_kernel from the TORCH layout w[co][ci][7][7] (the raw copy of the parameter).
// transpose_flip = 0: the forward (B[k = (tap, ci)][n = (tau, co)] = w[co][ci][dy][dx]); 1: the data gradient, a correlation of dz with
// w'[n_out = ci][n_in = co][dy][dx] = w[co][ci][6 - dy][6 - dx].  n_in / n_out: channel counts of the convolution being packed (<= 8).
__device__ __forceinline__ void pack_p2p_f16_raw_body(const float* __restrict__ w, uint4* __restrict__ out, int cin_fwd, int cout_fwd, int transpose_flip, int bx) {
    __shared__ int smax[8];
    const int n_in = transpose_flip ? cout_fwd : cin_fwd, n_out = transpose_flip ? cin_fwd : cout_fwd;
    auto wv = [&](int o, int i, int dy, int dx) {
        return transpose_flip ? w[((o < cin_fwd && i < cout_fwd ? i * cin_fwd + o : 0) * 7 + (6 - dy)) * 7 + (6 - dx)]
                              : w[((o < cout_fwd && i < cin_fwd ? o * cin_fwd + i : 0) * 7 + dy) * 7 + dx];
    };
    if (threadIdx.x < 8) smax[threadIdx.x] = 0;
    __syncthreads();
    for (int k = threadIdx.x; k < n_out * n_in * 49; k += blockDim.x) {
        const int o = k / (n_in * 49), r = k - o * n_in * 49, i = r / 49, t = r - i * 49;
        atomicMax(&smax[o], __float_as_int(fabsf(wv(o, i, t / 7, t % 7))));
    }
    __syncthreads();
    const int idx = bx * blockDim.x + threadIdx.x;      // (ks, lane)
    if (idx >= 14 * 64) return;
    const int ks = idx / 64, lane = idx - ks * 64;
    const int dy = ks >> 1, h = ks & 1;
    const int n = lane & 15, qq = lane >> 4;
    const int tau = n >> 3, co = n & 7;
    const int dx = 4 * h + qq - tau;
    const float sc = f16_weight_scale(__int_as_float(smax[co]));
    if (ks == 0 && qq == 0 && tau == 0) reinterpret_cast<float*>(out + kP2pFragScale)[co] = 1.f / sc;
    unsigned int hi[4] = {0, 0, 0, 0}, lo[4] = {0, 0, 0, 0};
    for (int ci = 0; ci < 8; ++ci) {
        float v = 0.f;
        if (dx >= 0 && dx < 7 && ci < n_in && co < n_out) v = sc * wv(co, ci, dy, dx);
        const _Float16 hv = static_cast<_Float16>(v);
        const unsigned int hb = __builtin_bit_cast(unsigned short, hv);
        const unsigned int lb = f16_bits((v - static_cast<float>(hv)) * kP2pLoScale);
        hi[ci >> 1] |= hb << (16 * (ci & 1));
        lo[ci >> 1] |= lb << (16 * (ci & 1));
    }
    out[(2 * ks + 0) * 64 + lane] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    out[(2 * ks + 1) * 64 + lane] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
}
__global__ void pack_p2p_f16_raw_kernel(const float* __restrict__ w, uint4* __restrict__ out, int cin_fwd, int cout_fwd, int transpose_flip) {
    pack_p2p_f16_raw_body(w, out, cin_fwd, cout_fwd, transpose_flip, blockIdx.x);
}
__global__ void pack_p2p_f16_raw_jobs_kernel(P2pRawJobs js) {
    const P2pRawJob& J = js.j[blockIdx.y];
    pack_p2p_f16_raw_body(J.w, J.out, J.cin_fwd, J.cout_fwd, J.transpose_flip, blockIdx.x);
}

// Pitch2PitchClassPool (models.py:95-106) of ready semitone maps [clip][C][S][T], S a multiple of 12: max over the octaves
__global__ void fold_max_kernel(const float* __restrict__ smap, int C, int S, int T, float* __restrict__ dst, long long dst_clip_stride, int dst_coff,
                                long long total) {
    const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;   // (clip, c, p, t)
    if (i >= total) return;
    const int t = static_cast<int>(i % T);
    long long r = i / T;
    const int p = static_cast<int>(r % 12); r /= 12;
    const int c = static_cast<int>(r % C);
    const long long clip = r / C;
    const float* src = smap + ((clip * C + c) * S + p) * T + t;
    float best = -INFINITY;
    for (int o = 0; o < S / 12; ++o) best = fmaxf(best, src[static_cast<long long>(12 * o) * T]);
    dst[clip * dst_clip_stride + (static_cast<long long>(dst_coff + c) * 12 + p) * T + t] = best;
}

// ==========================================================================================
// Layer 0 of the net in ONE launch (inference; models.py:361-374): semitone conv (1 channel) + octave fold, the
// PitchClass2PitchClass stack 1 -> NF -> NF -> NF (12 x 7, pitch classes circular, time zero-padded; NF <= 4), and layer 1's
// up_sixth (ConvTranspose (3,1)).  All of it is 2.9 M MAC on a 12 x T map per clip: five launches of 15-30 us each were
// mostly launch and tail.  One workgroup per clip keeps every map in LDS ([ch][12][T + 6 zero pad], 16-byte aligned rows);
// thread = (output-channel pair, pitch class, strip of 4 frames): 3 aligned 16-byte LDS reads feed 56 FMAs, the weights are
// wave-uniform (scalar loads).  The taps of the reference modules (fold output, conv outputs) are still written.
// ==========================================================================================
struct Layer0Args {
    const float* mel;            // [clip][1][H][T]
    const float* sw;             // semitone conv pack [1][3][3][1] and bias
    const float* sb;
    const float* w[4];           // conv packs [ci][12][7][4 co] and biases (BatchNorm folded)
    const float* b[4];
    float* dst[4];               // conv outputs [clip][dst_ctot][12][T] (the last one: channels [0, NF) of layer 1's concat buffer)
    long long dst_clip_stride[4];
    const float* uw;             // up_sixth pack [ci][co][3] and bias
    const float* ub;
    float* fold0;                // [clip][1][12][T]
    float* psix;                 // [clip][NF][36][T]
    int H, T, RP, NF, n_conv;    // RP: LDS row pitch in floats (T + 6 rounded up to 4)
    const uint4* frag[4];        // layer0_mfma_kernel: B fragments of the convs [12 dy][hi|lo][64 lanes] x 8 f16, then the 4 inverse channel scales (pack_l0_f16_kernel)
    int taps;                    // layer0_mfma_kernel: also write the intermediate conv outputs dst[0 .. n_conv - 2] (debug taps; ake_debug_keep_taps)
    int RPp;                     // layer0_mfma_kernel: row pitch of the channels-last maps, in positions (even, >= T + 8)
    int mel_fm;                  // layer0_mfma_kernel: mel is frames-major, [clip][T][H] (the CQT filter bank's own output order)
    uint2* psix_h;               // layer0_mfma_kernel: up_sixth's output as four f16 channels per (row, frame), [clip][36][T] x 8 bytes (channels >= NF
                                 // zero), INSTEAD of psix: what conv_p2p_f16_ps_kernel<1, 3> loads (one 8-byte load per patch position, no conversion)
    unsigned int* melh;          // (with psix_h) the log-CQT as f16 hi | f16 lo << 16 words in [clip][H][T] order, for the same reader: the transposition of
                                 // a frames-major mel is paid once, here, where the clip passes through the LDS anyway
};

__global__ __launch_bounds__(512) void layer0_fused_kernel(Layer0Args a) {
    extern __shared__ __attribute__((aligned(16))) float l0[];
    const int clip = blockIdx.x, tid = threadIdx.x;
    const int T = a.T, RP = a.RP, NF = a.NF;
    float* const f0 = l0;                        // [12][RP]
    float* const actA = l0 + 12 * RP;            // [4][12][RP]
    float* const actB = actA + 4 * 12 * RP;
    for (int i = tid; i < 9 * 12 * RP; i += 512) l0[i] = 0.f;
    __syncthreads();
    // ---- semitone conv + BN + LeakyReLU + octave fold (models.py:313-315, 95-106) ----
    // the clip's CQT (H x T floats) is streamed into LDS first: every load in flight at once, instead of eight dependent round trips
    {
        float* const ml = actB + 4 * 12 * RP;                      // [H][T]
        const float4* mel4 = reinterpret_cast<const float4*>(a.mel + static_cast<long long>(clip) * a.H * T);
        const int n4 = a.H * T / 4;                                // (the launcher requires H * T % 4 == 0)
        for (int i = tid; i < n4; i += 512) reinterpret_cast<float4*>(ml)[i] = mel4[i];
        __syncthreads();
        float w9[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) w9[i] = a.sw[i];
        const float sb = a.sb[0];
        const int n_oct = a.H / 36;
        for (int i = tid; i < 12 * T; i += 512) {
            const int p = i / T, t = i - p * T;
            const int tm = t == 0 ? T - 1 : t - 1, tp = t == T - 1 ? 0 : t + 1;
            float best = -INFINITY;
            for (int o = 0; o < n_oct; ++o) {
                const float* r = ml + 3 * (p + 12 * o) * T;
                float acc = 0.f;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    acc = fmaf(r[dy * T + tm], w9[dy * 3 + 0], acc);
                    acc = fmaf(r[dy * T + t], w9[dy * 3 + 1], acc);
                    acc = fmaf(r[dy * T + tp], w9[dy * 3 + 2], acc);
                }
                float v = acc + sb;
                v = v > 0.f ? v : v * kSlope;
                best = fmaxf(best, v);
            }
            f0[p * RP + 3 + t] = best;
            a.fold0[(static_cast<long long>(clip) * 12 + p) * T + t] = best;
        }
    }
    __syncthreads();
    // ---- the convolution stack ----
    const int nst = (T + 3) / 4;                                   // strips of 4 frames per row
    const int cp = __builtin_amdgcn_readfirstlane(tid >> 8);       // output channels 2cp, 2cp + 1 (wave-uniform)
    const int item = tid & 255;
    const float* in = f0;
    float* out = actA;
    int cin = 1;
    for (int j = 0; j < a.n_conv; ++j) {
        // the weights through the constant address space: hipcc then knows that the kernel's own stores cannot change them and
        // fetches them with scalar loads (as plain global pointers they became 14 vector loads + waits per 56 FMAs)
        typedef const float __attribute__((address_space(4))) cfloat;
        cfloat* w = (cfloat*)(a.w[j] + 2 * cp);
        for (int it = item; it < 12 * nst; it += 256) {
            const int p = it / nst, t0 = 4 * (it - p * nst);
            float acc[2][4];
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[c][k] = 0.f;
            for (int ci = 0; ci < cin; ++ci) {
#pragma unroll 4
                for (int dy = 0; dy < 12; ++dy) {   // (unrolled: the scalar weight loads of four rows are requested together)
                    int row = p + dy;
                    row -= row >= 12 ? 12 : 0;
                    const float4* rp = reinterpret_cast<const float4*>(in + (ci * 12 + row) * RP + t0);   // padded index t0 = frame t0 - 3
                    const float4 x0 = rp[0], x1 = rp[1], x2 = rp[2];
                    const float x[12] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w, x2.x, x2.y, x2.z, x2.w};
                    cfloat* wr = w + (ci * 12 + dy) * 28;
#pragma unroll
                    for (int dx = 0; dx < 7; ++dx) {
                        const float w0 = wr[dx * 4], w1 = wr[dx * 4 + 1];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            acc[0][k] = fmaf(x[k + dx], w0, acc[0][k]);
                            acc[1][k] = fmaf(x[k + dx], w1, acc[1][k]);
                        }
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int co = 2 * cp + c;
                if (co < NF) {
                    const float bias = a.b[j][co];
                    float* g = a.dst[j] + clip * a.dst_clip_stride[j] + (static_cast<long long>(co) * 12 + p) * T;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if (t0 + k < T) {
                            float v = acc[c][k] + bias;
                            v = v > 0.f ? v : v * kSlope;
                            out[(co * 12 + p) * RP + 3 + t0 + k] = v;
                            g[t0 + k] = v;
                        }
                    }
                }
            }
        }
        __syncthreads();
        in = out;
        out = out == actA ? actB : actA;
        cin = NF;
    }
    // ---- layer 1's up_sixth + BN + LeakyReLU (models.py:325-327): out[co][3p + jj][t] = lrelu(b[co] + sum_ci in[ci][p][t] w[ci][co][jj]) ----
    if (a.psix) {   // thread = (pitch class, frame): the 4 inputs once, then the NF x 3 outputs with wave-uniform (scalar) weights
        typedef const float __attribute__((address_space(4))) cfloat;
        cfloat* uw = (cfloat*)a.uw;
        cfloat* ub = (cfloat*)a.ub;
        float* ps = a.psix + static_cast<long long>(clip) * NF * 36 * T;
        for (int i = tid; i < 12 * T; i += 512) {
            const int p = i / T, t = i - p * T;
            float x[4];
#pragma unroll
            for (int ci = 0; ci < 4; ++ci) x[ci] = in[(ci * 12 + p) * RP + 3 + t];       // channels >= NF are zero
            for (int co = 0; co < NF; ++co) {
#pragma unroll
                for (int jj = 0; jj < 3; ++jj) {
                    float acc = ub[co];
#pragma unroll
                    for (int ci = 0; ci < 4; ++ci)
                        if (ci < NF) acc = fmaf(x[ci], uw[(ci * NF + co) * 3 + jj], acc);
                    ps[(static_cast<long long>(co) * 36 + 3 * p + jj) * T + t] = acc > 0.f ? acc : acc * kSlope;
                }
            }
        }
    }
}

// --local (models.py:720-722, 805-810): the key / tonic heads end in MaxPool2d((1, W), stride 1) and the maps are returned per
// frame.  The reference then *reshapes* [B][1][12][T'] to (B, T', 12) -- the same bytes -- so the outputs are written in the maps'
// own order: out[(clip * 12 + p) * Tq + t] = max_{w < W} map[(clip * 12 + p) * Tm + t + w]; sigmoid on key; genre is the map itself.
struct LocalPoolArgs {
    const float* maps[3];     // key, tonic, genre (genre may be null)
    float* outs[3];
    int Tm, Tq, W, batch;
};

__global__ void local_pool_kernel(LocalPoolArgs a) {
    const int which = blockIdx.y;
    if (!a.maps[which]) return;
    const int rows = which == 2 ? 11 : 12;
    const int To = which == 2 ? a.Tm : a.Tq;
    const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= static_cast<long long>(a.batch) * rows * To) return;
    const int t = static_cast<int>(i % To);
    const long long row = i / To;
    const float* m = a.maps[which] + row * a.Tm + t;
    float v = m[0];
    if (which < 2) {
        for (int w = 1; w < a.W; ++w) v = fmaxf(v, m[w]);
        if (which == 0) v = 1.f / (1.f + expf(-v));
    }
    a.outs[which][i] = v;
}

// ---- the same launch with the convolution stack on bf16 MFMA (split operands) ------------------------------------------------
// The VALU form above spends 2.9 M FMAs per clip at the vector rate (0.07 ms per 256 clips, the kernel is VALU-bound).  Here the
// maps live in LDS channels-last, [12 rows][T + 8 positions][4 ch] f16 (8 bytes per position, 3 zero positions
// either side = the convs' zero padding; ONE f16 value per activation and per weight, as in the pitch convs: this stack's operand
// rounding moves the outputs by 2e-6, tests/tools/split_precision_proto.py), and a conv is the MFMA form of conv_p2p_f16_kernel with 4 channels:
//   m = (pitch class p, frame pair j)     A[m][k = (position q', ci)] = X[(p + dy) mod 12][2j + q'][ci]    (8 positions x 4 channels = 32)
//   n = (tau, co) = 4 * tau + co < 8      B[k][n] = w[co][ci][dy][q' - tau]                                  (columns 8..15 idle)
//   k-step = dy: 12 MFMAs per 16 x 16 tile in two independent accumulator chains, ~29 tiles per clip.
// One A fragment = one aligned 16-byte LDS read (2 positions x 4 channels); the 12 weight fragments of a layer sit in registers and the
// next layer's are requested before the epilogue.  (Round 1: split bf16 x 3, one chain of 36 dependent MFMAs per tile, fragments
// fetched behind the barrier: the three convs were 24 of the kernel's 45 us.)
__global__ void pack_l0_f16_kernel(const float* __restrict__ w, uint4* __restrict__ out, int cin, int cout) {
    __shared__ int smax[4];
    channel_absmax4(w, cin * 84, smax);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;      // (dy, lane)
    if (i >= 12 * 64) return;
    const int dy = i / 64, lane = i - dy * 64;
    const int n = lane & 15, qq = lane >> 4;
    const int tau = n >> 2, co = n & 3;
    const float sc = f16_weight_scale(__int_as_float(smax[co]));
    if (dy == 0 && qq == 0 && tau == 0) reinterpret_cast<float*>(out + 24 * 64)[co] = 1.f / sc;
    unsigned int hi[4] = {0, 0, 0, 0}, lo[4] = {0, 0, 0, 0};
    for (int e = 0; e < 8; ++e) {
        const int pos = 2 * qq + (e >> 2), ci = e & 3;
        const int dx = pos - tau;
        float v = 0.f;
        if (n < 8 && dx >= 0 && dx < 7 && ci < cin && co < cout) v = sc * w[((ci * 12 + dy) * 7 + dx) * 4 + co];
        const _Float16 hv = static_cast<_Float16>(v);
        const unsigned int hb = __builtin_bit_cast(unsigned short, hv);
        const unsigned int lb = f16_bits((v - static_cast<float>(hv)) * kP2pLoScale);
        hi[e >> 1] |= hb << (16 * (e & 1));
        lo[e >> 1] |= lb << (16 * (e & 1));
    }
    out[(2 * dy + 0) * 64 + lane] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    out[(2 * dy + 1) * 64 + lane] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
}

__global__ __launch_bounds__(512) void layer0_mfma_kernel(Layer0Args a) {
    extern __shared__ __attribute__((aligned(16))) float l0[];
    const int clip = blockIdx.x, tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, q = lane >> 4;
    const int T = a.T, RP = a.RP, RPp = a.RPp, NF = a.NF;
    f16_saturate_mode();
    // LDS: f32 map of the latest conv output [4][12][RP] (for up_sixth) | two channels-last maps of [12][RPp] positions x 4 f16 (sized
    // as in round 1, when each had a hi and a lo plane) | the clip's CQT [H][T] (semitone phase only)
    float* const fmap = l0;
    unsigned short* const mapA = reinterpret_cast<unsigned short*>(l0 + 4 * 12 * RP);
    const int plane = 12 * RPp * 4;                                   // f16 elements per map
    unsigned short* const mapB = mapA + 2 * plane;
    float* const ml = reinterpret_cast<float*>(mapB + 2 * plane);
    for (int i = tid; i < 4 * 12 * RP + 2 * plane; i += 512) l0[i] = 0.f;
    // ---- semitone conv + BN + LeakyReLU + octave fold -> channel 0 of map A ----
    {
        const float4* mel4 = reinterpret_cast<const float4*>(a.mel + static_cast<long long>(clip) * a.H * T);
        const int n4 = a.H * T / 4;
        if (a.mel_fm) {          // [T][H] -> the [H][T] image: 16-byte reads along the bins, transposed on the way into the LDS
            const int h4 = a.H / 4;
            for (int i = tid; i < n4; i += 512) {
                const int t = i / h4, r4 = (i - t * h4) * 4;
                const float4 v = mel4[i];
                ml[(r4 + 0) * T + t] = v.x; ml[(r4 + 1) * T + t] = v.y; ml[(r4 + 2) * T + t] = v.z; ml[(r4 + 3) * T + t] = v.w;
            }
        } else
            for (int i = tid; i < n4; i += 512) reinterpret_cast<float4*>(ml)[i] = mel4[i];
        __syncthreads();
        if (a.melh) {
            uint4* const mh = reinterpret_cast<uint4*>(a.melh + static_cast<long long>(clip) * a.H * T);
            auto word = [](float v) {
                const _Float16 h = static_cast<_Float16>(v);
                return static_cast<unsigned int>(__builtin_bit_cast(unsigned short, h)) | (f16_bits(v - static_cast<float>(h)) << 16);
            };
            for (int i = tid; i < n4; i += 512) {
                const float4 v = reinterpret_cast<const float4*>(ml)[i];
                mh[i] = make_uint4(word(v.x), word(v.y), word(v.z), word(v.w));
            }
        }
        float w9[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) w9[i] = a.sw[i];
        const float sb = a.sb[0];
        const int n_oct = a.H / 36;
        for (int i = tid; i < 12 * T; i += 512) {
            const int p = i / T, t = i - p * T;
            const int tm = t == 0 ? T - 1 : t - 1, tp = t == T - 1 ? 0 : t + 1;
            float best = -INFINITY;
            for (int o = 0; o < n_oct; ++o) {
                const float* r = ml + 3 * (p + 12 * o) * T;
                float acc = 0.f;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    acc = fmaf(r[dy * T + tm], w9[dy * 3 + 0], acc);
                    acc = fmaf(r[dy * T + t], w9[dy * 3 + 1], acc);
                    acc = fmaf(r[dy * T + tp], w9[dy * 3 + 2], acc);
                }
                float v = acc + sb;
                v = v > 0.f ? v : v * kSlope;
                best = fmaxf(best, v);
            }
            const _Float16 bh16 = static_cast<_Float16>(best);
            mapA[(p * RPp + 3 + t) * 4] = __builtin_bit_cast(unsigned short, bh16);
            mapA[12 * RPp * 4 + (p * RPp + 3 + t) * 4] = static_cast<unsigned short>(f16_bits((best - static_cast<float>(bh16)) * kP2pLoScale));
            a.fold0[(static_cast<long long>(clip) * 12 + p) * T + t] = best;
        }
    }
    __syncthreads();
    // ---- the convolution stack on MFMA ----
    typedef float f32x4c __attribute__((ext_vector_type(4)));
    const int J = (T + 1) / 2, M = 12 * J, n_tiles = (M + 15) / 16;
    const int tau = (r16 >> 2) & 1, co = r16 & 3;
    const unsigned short* in = mapA;
    unsigned short* out = mapB;
    // Round 3: f32-equivalent products again (activations AND weights as f16 hi + f16 lo x 2^11, three MFMAs: ah*wh, and al'*wh + ah*wl' in an
    // accumulator of its own folded in with 2^-11).  Round 2 ran this stack on ONE product (f16 x f16): 2e-6 of the outputs on seeded-random
    // weights, but on a TRAINED net (tests/test_gpu_training.py::test_mixed_precision_inference_on_trained_weights) it was the largest single
    // contributor to the outputs' error, 6.7e-4 of a 7.9e-4 total against the 1e-3 budget.  The stack is 1 % of the network's MACs: the two
    // extra products cost microseconds.
    uint4 breg[12], bregl[12], bnext[12], bnextl[12];                     // this conv's weight fragments (hi, lo), the next one's
    auto frags_of = [&](int j) { return j == 0 ? a.frag[0] : j == 1 ? a.frag[1] : j == 2 ? a.frag[2] : a.frag[3]; };   // (no runtime index into the argument struct)
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        bnext[i] = a.n_conv > 0 ? frags_of(0)[2 * i * 64 + lane] : make_uint4(0, 0, 0, 0);
        bnextl[i] = a.n_conv > 0 ? frags_of(0)[(2 * i + 1) * 64 + lane] : make_uint4(0, 0, 0, 0);
    }
    for (int j = 0; j < a.n_conv; ++j) {
#pragma unroll
        for (int i = 0; i < 12; ++i) { breg[i] = bnext[i]; bregl[i] = bnextl[i]; }
        const float bias = (r16 < 8 && co < NF) ? a.b[j][co] : 0.f;
        const float iscale = reinterpret_cast<const float*>(frags_of(j) + 24 * 64)[co];
        const bool write_dst = a.taps || j == a.n_conv - 1;
        float* const g = a.dst[j] + clip * a.dst_clip_stride[j];
        if (j + 1 < a.n_conv) {                                           // in flight during this conv's multiply loop
#pragma unroll
            for (int i = 0; i < 12; ++i) { bnext[i] = frags_of(j + 1)[2 * i * 64 + lane]; bnextl[i] = frags_of(j + 1)[(2 * i + 1) * 64 + lane]; }
        }
        for (int tile = wave; tile < n_tiles; tile += 8) {
            int m = tile * 16 + r16;
            m = m < M ? m : M - 1;
            const int p = m / J, jj = m - p * J;
            f32x4c acc = {0.f, 0.f, 0.f, 0.f}, accl = {0.f, 0.f, 0.f, 0.f};   // hi x hi, and the two cross terms (x 2^11)
#pragma unroll
            for (int dy = 0; dy < 12; ++dy) {
                int row = p + dy;
                row -= row >= 12 ? 12 : 0;
                const int pos = row * RPp + 2 * jj + 2 * q;              // padded position of frame 2j - 3 + 2q
                const f16x8c ah = __builtin_bit_cast(f16x8c, *reinterpret_cast<const uint4*>(in + pos * 4));
                const f16x8c al = __builtin_bit_cast(f16x8c, *reinterpret_cast<const uint4*>(in + plane + pos * 4));
                const f16x8c bh = __builtin_bit_cast(f16x8c, breg[dy]), bl = __builtin_bit_cast(f16x8c, bregl[dy]);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0);
                accl = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, accl, 0, 0, 0);
                accl = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, accl, 0, 0, 0);
            }
            if (r16 < 8 && co < NF) {                                    // D[m = 4q + i][n = (tau, co)]
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int mm = tile * 16 + 4 * q + i;
                    const int pp = mm / J, t = 2 * (mm - pp * J) + tau;
                    if (mm < M && t < T) {
                        float v = fmaf(fmaf(accl[i], kP2pLoInv, acc[i]), iscale, bias);
                        v = fmaxf(v, v * kSlope);
                        const _Float16 hv = static_cast<_Float16>(v);
                        out[(pp * RPp + 3 + t) * 4 + co] = __builtin_bit_cast(unsigned short, hv);
                        out[plane + (pp * RPp + 3 + t) * 4 + co] = static_cast<unsigned short>(f16_bits((v - static_cast<float>(hv)) * kP2pLoScale));
                        fmap[(co * 12 + pp) * RP + 3 + t] = v;
                        if (write_dst) g[(static_cast<long long>(co) * 12 + pp) * T + t] = v;
                    }
                }
            }
        }
        __syncthreads();
        const unsigned short* tmp = in;
        in = out;
        out = const_cast<unsigned short*>(tmp);
    }
    // ---- layer 1's up_sixth + BN + LeakyReLU from the f32 map of the last conv ----
    if (a.psix_h) {          // ... as f16 x 4 per position (the pitch conv that reads it multiplies f16 anyway: the same values, rounded here)
        typedef const float __attribute__((address_space(4))) cfloat;
        cfloat* uw = (cfloat*)a.uw;
        cfloat* ub = (cfloat*)a.ub;
        uint2* ph = a.psix_h + static_cast<long long>(clip) * 36 * T;
        for (int i = tid; i < 12 * T; i += 512) {
            const int p = i / T, t = i - p * T;
            float x[4];
#pragma unroll
            for (int ci = 0; ci < 4; ++ci) x[ci] = fmap[(ci * 12 + p) * RP + 3 + t];
#pragma unroll
            for (int jj = 0; jj < 3; ++jj) {
                unsigned int h[4];
#pragma unroll
                for (int co2 = 0; co2 < 4; ++co2) {
                    float acc = co2 < NF ? ub[co2 < NF ? co2 : 0] : 0.f;
#pragma unroll
                    for (int ci = 0; ci < 4; ++ci)
                        if (ci < NF && co2 < NF) acc = fmaf(x[ci], uw[(ci * NF + co2) * 3 + jj], acc);
                    acc = acc > 0.f ? acc : acc * kSlope;
                    h[co2] = __builtin_bit_cast(unsigned short, static_cast<_Float16>(acc));
                }
                ph[(3 * p + jj) * T + t] = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
            }
        }
    } else if (a.psix) {
        typedef const float __attribute__((address_space(4))) cfloat;
        cfloat* uw = (cfloat*)a.uw;
        cfloat* ub = (cfloat*)a.ub;
        float* ps = a.psix + static_cast<long long>(clip) * NF * 36 * T;
        for (int i = tid; i < 12 * T; i += 512) {
            const int p = i / T, t = i - p * T;
            float x[4];
#pragma unroll
            for (int ci = 0; ci < 4; ++ci) x[ci] = fmap[(ci * 12 + p) * RP + 3 + t];
            for (int co2 = 0; co2 < NF; ++co2) {
#pragma unroll
                for (int jj = 0; jj < 3; ++jj) {
                    float acc = ub[co2];
#pragma unroll
                    for (int ci = 0; ci < 4; ++ci)
                        if (ci < NF) acc = fmaf(x[ci], uw[(ci * NF + co2) * 3 + jj], acc);
                    ps[(static_cast<long long>(co2) * 36 + 3 * p + jj) * T + t] = acc > 0.f ? acc : acc * kSlope;
                }
            }
        }
    }
}

// --pc2p_mem (PitchClass2Pitch_MemoryVariant, models.py:145-166): the up_sixth map is summed over groups of its channels and ADDED to
// the pitch stream instead of being concatenated to it.  The reference reshapes the P rows to (36, P / 36): row r takes
// third-semitone index r / (P / 36) -- eight consecutive rows share one -- kept as it is.
// psix_aff (training, nullable): psix is up_sixth's RAW output, its BatchNorm + LeakyReLU is applied while loading.
__global__ void pc2p_mem_kernel(const float* __restrict__ p, const float* __restrict__ psix, float* __restrict__ out, int cp, int ratio, int P, int T,
                                long long total, const float* __restrict__ psix_aff) {
    const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;      // (clip, c, row, t)
    if (i >= total) return;
    const int t = static_cast<int>(i % T);
    long long q = i / T;
    const int r = static_cast<int>(q % P);
    q /= P;
    const int c = static_cast<int>(q % cp);
    const long long clip = q / cp;
    const int k = r / (P / 36);
    const float* s = psix + (((clip * cp + c) * ratio) * 36 + k) * T + t;
    float acc = p[i];
    for (int g = 0; g < ratio; ++g) {
        float v = s[static_cast<long long>(g) * 36 * T];
        if (psix_aff) {
            const float* a3 = psix_aff + 3 * (c * ratio + g);
            const float y = fmaf(v, a3[0], a3[1]);
            v = y > 0.f ? y : y * a3[2];
        }
        acc += v;
    }
    out[i] = acc;
}

// --p2pc_conv (Pitch2PitchClassConv, models.py:108-133): the octave fold as a learned convolution over the octaves and channels
// (kernel (n_oct, 1), dilation (12, 1)) + BN (folded) + LeakyReLU, instead of the max.  src: semitone maps [clip][C][12 * n_oct][T],
// either final activations or the raw pre-activation conv output (in_lrelu = 1 applies the LeakyReLU while loading);
// w: [co][ci][n_oct], dst: channels [coff, coff + C) of [clip][ctot][12][T].
__global__ void fold_conv_kernel(const float* __restrict__ src, const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ dst,
                                 int C, int n_oct, int T, int in_lrelu, long long dst_clip_stride, int dst_coff, long long total) {
    const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;      // (clip, co, p, t)
    if (i >= total) return;
    const int t = static_cast<int>(i % T);
    long long r = i / T;
    const int p = static_cast<int>(r % 12); r /= 12;
    const int co = static_cast<int>(r % C);
    const long long clip = r / C;
    const float* s = src + (clip * C * 12 * n_oct + p) * T + t;
    float acc = bias[co];
    for (int ci = 0; ci < C; ++ci)
        for (int o = 0; o < n_oct; ++o) {
            float x = s[(static_cast<long long>(ci) * 12 * n_oct + 12 * o) * T];
            if (in_lrelu) x = x > 0.f ? x : x * kSlope;
            acc = fmaf(x, w[(co * C + ci) * n_oct + o], acc);
        }
    dst[clip * dst_clip_stride + (static_cast<long long>(dst_coff + co) * 12 + p) * T + t] = acc > 0.f ? acc : acc * kSlope;
}

// channels [0, C) of a wider buffer [clip][ctot][HT] -> dense [clip][C][HT]
__global__ void slice_channels_kernel(const float* __restrict__ src, long long src_clip_stride, float* __restrict__ dst, long long per_clip, long long total) {
    const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const long long clip = i / per_clip;
    dst[i] = src[clip * src_clip_stride + (i - clip * per_clip)];
}

// ==========================================================================================
// Training-mode forward helpers (BatchNorm with batch statistics, nn.BatchNorm2d in train(), models.py:196 etc.)
// ==========================================================================================

__device__ __forceinline__ float affine_act(float x, const float* aff, int c) {
    if (!aff) return x;
    const float y = fmaf(x, aff[3 * c], aff[3 * c + 1]);
    return y > 0.f ? y : y * aff[3 * c + 2];
}

// wave-level (sum, sumsq) -> one double atomic pair per wave
__device__ __forceinline__ void stats_commit(double* stats, int stats_stride, int c, const ShiftStat& ss) {
    long long s1, s2;
    ss.fixed(s1, s2);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s1 += shfl_xor_ll(s1, o); s2 += shfl_xor_ll(s2, o); }
    if ((threadIdx.x & 63) == 0) {
        double* st = stats + static_cast<size_t>((blockIdx.x + 7 * blockIdx.z + (threadIdx.x >> 6)) & (kStatSlots - 1)) * stats_stride;
        fx_add_int(st + 2 * c, s1);
        fx_add_int(st + 2 * c + 1, s2);
    }
}

// Semitone conv (models.py:313 / :337) WITHOUT its BatchNorm: raw output [B][C][H/3][T] + per-channel statistics.
// Same thread mapping as semi_fold_kernel; the input may carry a pending BatchNorm+LeakyReLU (in_affine).
struct SemiTrainArgs {
    SemiArgs s;               // dst = raw [B][C][H/3][T] (dst_coff / dst_clip_stride unused: dense)
    const float* in_affine;   // [C][3] or null
    double* stats;            // [kStatSlots][..][2]
    int stats_stride;
    int out_lrelu;            // 1: store the activated map (inference uses: --stay_sixth's pitch stream)
};

template <int CO>
__global__ void semi_conv_stats_kernel(SemiTrainArgs ta) {
    // thread = (semitone row, strip of TW frames): H / 3 x n_strips items per clip (the first version looped over the octaves in
    // every thread -- a quarter of a wave slot per SIMD, every load latency exposed -- and its early-out `break`s in the unrolled
    // channel loops turned the accumulator arrays into scratch memory: 0.58 ms per 256 clips)
    const SemiArgs& a = ta.s;
    const int item = blockIdx.x * blockDim.x + threadIdx.x;
    const int rows_out = a.H / 3;
    const int per_clip = rows_out * a.n_strips;
    const bool live = item < per_clip;
    const int srow = live ? item / a.n_strips : 0;
    const int sidx = live ? item - srow * a.n_strips : 0;
    const int grp = blockIdx.y;
    const int clip = blockIdx.z;
    const int t0 = sidx * TW;
    int tix[TW + 2];
#pragma unroll
    for (int j = 0; j < TW + 2; ++j) tix[j] = wrap(t0 - 1 + j, a.T);
    const float* src = a.src + clip * a.src_clip_stride;
    const float* __restrict__ wg = a.w + static_cast<long long>(grp) * a.C * (9 * CO);
    float* d = a.dst + static_cast<long long>(clip) * a.C * rows_out * a.T;
    const int row0 = 3 * srow;
    float acc[CO][TW];
#pragma unroll
    for (int co = 0; co < CO; ++co)
#pragma unroll
        for (int j = 0; j < TW; ++j) acc[co][j] = 0.f;
    for (int ci = 0; ci < a.C; ++ci) {
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const float* rowp = src + (static_cast<long long>(ci) * a.H + row0 + dy) * a.T;
            float in[TW + 2];
#pragma unroll
            for (int j = 0; j < TW + 2; ++j) in[j] = affine_act(rowp[tix[j]], ta.in_affine, ci);
            const float* __restrict__ wp = wg + (ci * 3 + dy) * (3 * CO);
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                for (int co = 0; co < CO; ++co) {
                    const float wv = wp[dx * CO + co];
#pragma unroll
                    for (int j = 0; j < TW; ++j) acc[co][j] = fmaf(in[j + dx], wv, acc[co][j]);
                }
        }
    }
#pragma unroll
    for (int co = 0; co < CO; ++co) {
        const int c = grp * CO + co;
        const bool okc = c < a.C;
        const float b = okc ? a.bias[c] : 0.f;
        float* drow = d + (static_cast<long long>(okc ? c : 0) * rows_out + srow) * a.T;
        ShiftStat ss;
        ss.init();
#pragma unroll
        for (int j = 0; j < TW; ++j) {
            const float v = acc[co][j] + b;
            if (okc && live && t0 + j < a.T) { drow[t0 + j] = (ta.out_lrelu && v < 0.f) ? v * kSlope : v; ss.add(v); }
        }
        if (okc && ta.stats) stats_commit(ta.stats, ta.stats_stride, c, ss);      // (null: inference use by --p2pc_conv, no statistics)
    }
}

// BatchNorm + LeakyReLU (as `aff`) then Pitch2PitchClassPool (models.py:95-106): raw [B][C][12*n_oct][T] -> [B][ctot][12][T]
__global__ void fold_affine_kernel(const float* __restrict__ src, const float* __restrict__ aff, float* __restrict__ dst, int C,
                                   int n_oct, int T, int dst_ctot, int dst_coff, long long total) {
    const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int t = static_cast<int>(i % T);
    long long q = i / T;
    const int p = static_cast<int>(q % 12);
    q /= 12;
    const int c = static_cast<int>(q % C);
    const long long clip = q / C;
    const float* s = src + ((clip * C + c) * (12 * n_oct) + p) * T + t;
    float m = -INFINITY;
    for (int o = 0; o < n_oct; ++o) m = fmaxf(m, affine_act(s[static_cast<long long>(o) * 12 * T], aff, c));
    dst[((clip * dst_ctot + dst_coff + c) * 12 + p) * T + t] = m;
}

// up_sixth in training mode: raw ConvTranspose2d output + statistics; input carries a pending affine.
__global__ void up_sixth_train_kernel(const float* __restrict__ src, long long src_clip_stride, const float* __restrict__ in_aff,
                                      const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ dst,
                                      double* __restrict__ stats, int stats_stride, int C, int T, long long total) {
    __shared__ long long sh[2 * 128];                              // fixed point: the order of the threads' adds does not matter
    for (int k = threadIdx.x; k < 2 * C; k += blockDim.x) sh[k] = 0;
    __syncthreads();
    const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < total) {
        const int t = static_cast<int>(i % T);
        long long q = i / T;
        const int row = static_cast<int>(q % 36);
        q /= 36;
        const int co = static_cast<int>(q % C);
        const long long clip = q / C;
        const int p = row / 3, j = row - 3 * p;
        const float* s = src + clip * src_clip_stride + static_cast<long long>(p) * T + t;
        float acc = bias[co];
        for (int ci = 0; ci < C; ++ci) acc = fmaf(affine_act(s[static_cast<long long>(ci) * 12 * T], in_aff, ci), w[(ci * C + co) * 3 + j], acc);
        dst[i] = acc;
        fx_add(&sh[2 * co], acc, kFxStat);
        fx_add(&sh[2 * co + 1], static_cast<double>(acc) * acc, kFxStat);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < 2 * C; k += blockDim.x)
        if (sh[k] != 0) atomicAdd(reinterpret_cast<unsigned long long*>(stats + static_cast<size_t>(blockIdx.x & (kStatSlots - 1)) * stats_stride + k),
                                  static_cast<unsigned long long>(sh[k]));
}

// --p2pc_conv in training mode (Pitch2PitchClassConv, models.py:108-133): the octave-fold convolution (kernel (n_oct, 1), dilation
// (12, 1)) over the RAW semitone maps [clip][C][12 * n_oct][T], whose pending BatchNorm + LeakyReLU is applied while loading; raw output
// [clip][C][12][T] (dense) + the statistics of pool.bn.  w: the reference layout [co][ci][n_oct].
__global__ void fold_conv_train_kernel(const float* __restrict__ src, const float* __restrict__ in_aff, const float* __restrict__ w,
                                       const float* __restrict__ bias, float* __restrict__ dst, double* __restrict__ stats, int stats_stride, int C,
                                       int n_oct, int T, long long total) {
    __shared__ long long sh[2 * 128];                              // fixed point: the order of the threads' adds does not matter
    for (int k = threadIdx.x; k < 2 * C; k += blockDim.x) sh[k] = 0;
    __syncthreads();
    const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;      // (clip, co, p, t)
    if (i < total) {
        const int t = static_cast<int>(i % T);
        long long r = i / T;
        const int p = static_cast<int>(r % 12); r /= 12;
        const int co = static_cast<int>(r % C);
        const long long clip = r / C;
        const float* s = src + (clip * C * 12 * n_oct + p) * T + t;
        float acc = bias[co];
        for (int ci = 0; ci < C; ++ci)
            for (int o = 0; o < n_oct; ++o)
                acc = fmaf(affine_act(s[(static_cast<long long>(ci) * 12 * n_oct + 12 * o) * T], in_aff, ci), w[(co * C + ci) * n_oct + o], acc);
        dst[i] = acc;
        fx_add(&sh[2 * co], acc, kFxStat);
        fx_add(&sh[2 * co + 1], static_cast<double>(acc) * acc, kFxStat);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < 2 * C; k += blockDim.x)
        if (sh[k] != 0) atomicAdd(reinterpret_cast<unsigned long long*>(stats + static_cast<size_t>(blockIdx.x & (kStatSlots - 1)) * stats_stride + k),
                                  static_cast<unsigned long long>(sh[k]));
}

// dst[clip][coff + c][ht] (of dst_ctot channels) = LeakyReLU(BatchNorm(src[clip][c][ht])) with the pending table `aff`
__global__ void apply_affine_kernel(const float* __restrict__ src, const float* __restrict__ aff, float* __restrict__ dst, int C, long long HT,
                                    int dst_ctot, int dst_coff, long long total) {
    const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const long long ht = i % HT;
    const long long q = i / HT;
    const int c = static_cast<int>(q % C);
    const long long clip = q / C;
    dst[(clip * dst_ctot + dst_coff + c) * HT + ht] = affine_act(src[i], aff, c);
}

// The heads' LAST convolution in training mode (cin -> 1 channel, kh x 7, "valid" in time; key / tonic: 12 circular rows, genre: kh = 2 over
// valid rows; models.py:36-47, 716-729): one output channel leaves 15 of 16 MFMA columns empty and the generic kernel walks its 32 input
// channels in 16 LDS passes (0.085 ms per head and 256 clips, 0.1 ms at 8 clips).  Here one workgroup per clip stages the activated input
// (pending BatchNorm + LeakyReLU applied) and the raw weights [ci][kh][7] in LDS; a thread owns one output position.  Exact f32.
__global__ __launch_bounds__(384) void conv_head_last_kernel(const float* __restrict__ x, const float* __restrict__ x_aff, const float* __restrict__ w,
                                                             const float* __restrict__ bias, float* __restrict__ dst, int cin, int KH, int circular,
                                                             int T_in, int T_out) {
    extern __shared__ float hl_lds[];
    float* const xs = hl_lds;                        // [cin][12][T_in]
    float* const ws = hl_lds + cin * 12 * T_in;      // [cin][KH][7]
    const int clip = blockIdx.x;
    const float* xc = x + static_cast<long long>(clip) * cin * 12 * T_in;
    for (int i = threadIdx.x; i < cin * 12 * T_in; i += blockDim.x) xs[i] = affine_act(xc[i], x_aff, i / (12 * T_in));
    for (int i = threadIdx.x; i < cin * KH * 7; i += blockDim.x) ws[i] = w[i];
    __syncthreads();
    const int H_out = circular ? 12 : 12 - KH + 1;
    const float b0 = bias ? bias[0] : 0.f;
    for (int o = threadIdx.x; o < H_out * T_out; o += blockDim.x) {
        const int y = o / T_out, t = o - y * T_out;
        float acc = b0;
        for (int ci = 0; ci < cin; ++ci)
            for (int dy = 0; dy < KH; ++dy) {
                int row = y + dy;
                row -= row >= 12 ? 12 : 0;
                const float* xr = xs + (ci * 12 + row) * T_in + t;
                const float* wr = ws + (ci * KH + dy) * 7;
#pragma unroll
                for (int dx = 0; dx < 7; ++dx) acc = fmaf(xr[dx], wr[dx], acc);
            }
        dst[(static_cast<long long>(clip) * H_out + y) * T_out + t] = acc;
    }
}

// time pooling with a pending affine on the input
__global__ void time_pool_affine_kernel(const float* __restrict__ src, const float* __restrict__ aff, float* __restrict__ dst, int C,
                                        int H, int T, int tp, int dst_ctot, int dst_coff, long long total) {
    const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int To = T / tp;
    const int t = static_cast<int>(i % To);
    long long q = i / To;
    const int y = static_cast<int>(q % H);
    q /= H;
    const int c = static_cast<int>(q % C);
    const long long clip = q / C;
    const float* s = src + ((clip * C + c) * H + y) * T + static_cast<long long>(t) * tp;
    float m = affine_act(s[0], aff, c);
    for (int j = 1; j < tp; ++j) m = fmaxf(m, affine_act(s[j], aff, c));
    dst[((clip * dst_ctot + dst_coff + c) * H + y) * To + t] = m;
}

// (sum, sumsq) over `count` values per channel -> BatchNorm(train) as an affine triple, plus (batch mean, biased
// variance, count) for the running-statistics update done by the caller (momentum 0.1, unbiased variance: torch semantics).
__global__ void bn_finalize_kernel(const double* __restrict__ stats, int stats_stride, double count, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float* __restrict__ aff, float* __restrict__ batch_stats, int C, float slope) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    long long i1 = 0, i2 = 0;                                      // the slots hold fixed-point sums (fx_add): exact integer adds
    for (int k = 0; k < kStatSlots; ++k) {
        i1 += reinterpret_cast<const long long*>(stats)[static_cast<size_t>(k) * stats_stride + 2 * c];
        i2 += reinterpret_cast<const long long*>(stats)[static_cast<size_t>(k) * stats_stride + 2 * c + 1];
    }
    const double s1 = fx_checked(i1, kFxStat), s2 = fx_checked(i2, kFxStat);
    const double mean = s1 / count;
    double var = s2 / count - mean * mean;
    if (var < 0) var = 0;
    const double sc = static_cast<double>(gamma[c]) / sqrt(var + 1e-5);
    aff[3 * c] = static_cast<float>(sc);
    aff[3 * c + 1] = static_cast<float>(static_cast<double>(beta[c]) - mean * sc);
    aff[3 * c + 2] = slope;                                        // kSlope; 1 for a BatchNorm with no activation of its own (ResBlock b2)
    batch_stats[3 * c] = static_cast<float>(mean);
    batch_stats[3 * c + 1] = static_cast<float>(var);
    batch_stats[3 * c + 2] = static_cast<float>(count);
}

// --denseblock training: (sum, sum of squares) of channels [0, C) of a READY tensor [clip][ctot][HT] (a dense block's input: its first
// norm1 normalises features no convolution of the block produced).  grid (C, 1, clips).
__global__ __launch_bounds__(256) void channel_stats_kernel(const float* __restrict__ src, long long clip_stride, long long HT, double* __restrict__ stats,
                                                            int stats_stride) {
    const int c = blockIdx.x;
    const float* p = src + static_cast<long long>(blockIdx.z) * clip_stride + static_cast<long long>(c) * HT;
    ShiftStat ss;
    ss.init();
    for (long long i = threadIdx.x; i < HT; i += 256) ss.add(p[i]);
    stats_commit(stats, stats_stride, c, ss);
}

// ... the statistics of C channels copied from one BatchNorm layer's cells to another's (every norm1 of a dense block normalises the same
// features with its own gamma / beta: the sums are taken once)
__global__ void stats_copy_kernel(double* __restrict__ stats, int stats_stride, int src_ch, int dst_ch, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= kStatSlots * 2 * C) return;
    const int k = i / (2 * C), r = i - k * 2 * C;
    stats[static_cast<size_t>(k) * stats_stride + 2 * dst_ch + r] = stats[static_cast<size_t>(k) * stats_stride + 2 * src_ch + r];
}

// dst[clip][c][i] += src[clip][c][i] for the first C channels of two tensors with their own channel counts
__global__ void add_channels_kernel(float* __restrict__ dst, int dst_ctot, const float* __restrict__ src, int src_ctot, int C, long long HT, long long total) {
    const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const long long e = i % HT;
    const long long r = i / HT;
    const int c = static_cast<int>(r % C);
    const long long clip = r / C;
    dst[(clip * dst_ctot + c) * HT + e] += src[(clip * src_ctot + c) * HT + e];
}

__global__ void affine_identity_kernel(float* __restrict__ aff, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    aff[3 * c] = 1.f; aff[3 * c + 1] = 0.f; aff[3 * c + 2] = 1.f;
}

// --resblock, training (models.py:402-454): x_out = LeakyReLU(b2(z2) + x_in).  z2 is conv2's raw output with b2's pending (scale,
// shift) in aff2; x_in is read through its own pending table (the stack's first conv: raw + BatchNorm + LeakyReLU; a block's output:
// identity).  The sum is MATERIALISED (dense, or as channels [0, C) of a wider concat buffer): the next block reads it twice.
__global__ void res_add_act_kernel(const float* __restrict__ z2, const float* __restrict__ aff2, const float* __restrict__ x,
                                   const float* __restrict__ x_aff, float* __restrict__ dst, int C, int HT, int dst_ctot, long long total) {
    const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int ht = static_cast<int>(i % HT);
    const long long q = i / HT;
    const int c = static_cast<int>(q % C);
    const long long clip = q / C;
    const float s = fmaf(z2[i], aff2[3 * c], aff2[3 * c + 1]) + affine_act(x[i], x_aff, c);
    dst[(clip * dst_ctot + c) * HT + ht] = s > 0.f ? s : s * kSlope;
}

// masked temporal mean with a pending affine on nothing (the last head conv has no BatchNorm): reuse head_pool_kernel.

}  // namespace ake_k

// Backward-pass kernels of PitchClassNet (training, models.py:952-963 -> autograd in the reference).
// Included by pcnet.hip only.  First correct version: every op has its own kernel; BatchNorm backward is two
// passes (reduce, apply).  The convolution data gradients reuse conv_mfma_kernel with transposed / flipped
// weight fragments; the weight gradients use conv_wgrad_kernel below (f32 MFMA, reduction over positions).
//
// Notation per BatchNorm block   z = conv(x)   a = lrelu(gamma * zhat + beta),  zhat = (z - mu) * rstd:
//   ga  = dL/da            (arrives from the consumer's data-gradient)
//   g1  = ga * lrelu'(.)   S1 = sum g1,  S2 = sum g1 * zhat,  S3 = sum (z - mu)   (act_bwd_stats_kernel)
//   dz  = gamma*rstd * (g1 - S1/N - zhat * S2/N)                   (bn_bwd_coef_kernel + bn_bwd_apply_kernel)
//   dgamma = S2, dbeta = S1
#pragma once

#include "pcnet_kernels.h"

namespace ake_k {

// Weight gradients are reductions over (clip, position) finished with atomics.  Thousands of workgroups adding into the
// same few cache lines serialise in L2 (measured: 1.6 ms for 2.4 M atomics on 18 lines), so every writer adds into one of
// kGradSlots copies of the flat gradient buffer and grad_reduce_kernel sums the copies into the caller's buffer.
// The copies hold 64-bit fixed point (kFxGrad, pcnet_kernels.h): the sums are independent of the arrival order, a training
// step is bit-identical from run to run, and a data-parallel step equals its single-process emulation exactly.
constexpr int kGradSlots = 16;

__device__ __forceinline__ gfx_t* grad_slot(gfx_t* base, long long slot_stride) {
    return base + static_cast<long long>((blockIdx.x + 3 * blockIdx.y + 5 * blockIdx.z) & (kGradSlots - 1)) * slot_stride;
}
__device__ __forceinline__ void grad_add(gfx_t* cell, float v) { fx_add(cell, v, kFxGrad); }

// out[i] (+)= sum over slots (exact integer sum, one rounding to float)
__global__ __launch_bounds__(256) void grad_reduce_kernel(const gfx_t* __restrict__ slots, float* __restrict__ out, long long n, int accumulate) {
    const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    long long si = 0;
#pragma unroll
    for (int k = 0; k < kGradSlots; ++k) si += slots[k * n + i];
    const float s = static_cast<float>(fx_checked(si, kFxGrad));
    out[i] = accumulate ? out[i] + s : s;
}

// ---- masked temporal mean + sigmoid, backward (models.py:754-804) -----------------------------------------------
struct PoolHeadBwdArgs {
    const float* d_out[3];    // dL/d(key_out, tonic_out, genre_out)  [B][rows]
    const float* key_out;     // sigmoid output (for sigma')
    float* d_map[3];          // [B][12][Tm] each (genre: row 11 is written as zeros)
    int rows[3];
    int Tm;
    const long long* seq;
    int n_pool_layers, tp, shrink, batch;
    const float* maps[3];     // --max_pool: the forward's maps [B][rows (12 stored)][Tm]: the gradient goes to the (first) maximum
    int max_pool;
};

__global__ void head_pool_bwd_kernel(PoolHeadBwdArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int which = blockIdx.y;
    if (a.d_map[which] == nullptr || i >= a.batch * 12) return;
    const int clip = i / 12, row = i - clip * 12;
    float* dm = a.d_map[which] + static_cast<long long>(i) * a.Tm;
    if (row >= a.rows[which]) {
        for (int t = 0; t < a.Tm; ++t) dm[t] = 0.f;
        return;
    }
    int L = a.Tm;
    if (a.seq) {
        long long l = a.seq[clip];
        for (int k = 0; k < a.n_pool_layers; ++k) l = l / a.tp;
        L = static_cast<int>(l) - a.shrink;
        if (L > a.Tm) L = a.Tm;
        if (L < 0) L = a.Tm + L > 0 ? a.Tm + L : 0;
    }
    float g = a.d_out[which][clip * a.rows[which] + row];
    if (which == 0) {
        const float y = a.key_out[clip * 12 + row];
        g *= y * (1.f - y);
    }
    // models.py:764-797: torch.max over the frames for every clip without seq_length, for clip 0 only with it (the quirk kept)
    if (a.max_pool && (!a.seq || clip == 0)) {
        const float* m = a.maps[which] + (static_cast<long long>(clip) * a.rows[which] + row) * a.Tm;
        int best = 0;
        for (int t = 1; t < L; ++t) best = m[t] > m[best] ? t : best;
        for (int t = 0; t < a.Tm; ++t) dm[t] = (t == best && L > 0) ? g : 0.f;
        return;
    }
    const float gl = L > 0 ? g / static_cast<float>(L) : 0.f;
    for (int t = 0; t < a.Tm; ++t) dm[t] = t < L ? gl : 0.f;
}

// ---- --local heads, backward (models.py:720-722, 805-810) ---------------------------------------------------------
// Forward: out[(clip * 12 + p) * Tq + t] = max_{w < W} map[(clip * 12 + p) * Tm + t + w] (sigmoid on key), genre = the map.
// One thread per map element (clip, p, tau) gathers the gradients of the windows whose (first) maximum sits at tau -- MaxPool2d's
// backward -- in a fixed order, so the result does not depend on scheduling.  d_map is [B][12][Tm]; row 11 of the genre one is zero.
struct LocalPoolBwdArgs {
    const float* d_out[3];    // dL/d(key_out, tonic_out) in the maps' order [B][12][Tq]; dL/d(genre_out) [B][11][Tm]
    const float* key_out;     // sigmoid output [B][12][Tq]
    const float* maps[3];     // the forward's maps (key, tonic: [B][12][Tm])
    float* d_map[3];
    int Tm, Tq, W, batch;
};

__global__ void local_pool_bwd_kernel(LocalPoolBwdArgs a) {
    const int which = blockIdx.y;
    if (!a.d_map[which]) return;
    const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= static_cast<long long>(a.batch) * 12 * a.Tm) return;
    const int tau = static_cast<int>(i % a.Tm);
    const long long row = i / a.Tm;                       // clip * 12 + p
    if (which == 2) {
        const int p = static_cast<int>(row % 12);
        const long long clip = row / 12;
        a.d_map[2][i] = p < 11 ? a.d_out[2][(clip * 11 + p) * a.Tm + tau] : 0.f;
        return;
    }
    const float* m = a.maps[which] + row * a.Tm;
    const float mv = m[tau];
    float acc = 0.f;
    const int t0 = tau - a.W + 1 > 0 ? tau - a.W + 1 : 0;
    const int t1 = tau < a.Tq - 1 ? tau : a.Tq - 1;
    for (int t = t0; t <= t1; ++t) {
        // tau is the arg-max of window t iff nothing before it in the window is >= and nothing after it is >
        bool best = true;
        for (int w = t; w < tau && best; ++w) best = m[w] < mv;
        for (int w = tau + 1; w < t + a.W && best; ++w) best = m[w] <= mv;
        if (!best) continue;
        float g = a.d_out[which][row * a.Tq + t];
        if (which == 0) {
            const float y = a.key_out[row * a.Tq + t];
            g *= y * (1.f - y);
        }
        acc += g;
    }
    a.d_map[which][i] = acc;
}

// ---- LeakyReLU' and the BatchNorm reductions -------------------------------------------------------------------------
// g (in: ga, out: g1) and z share the layout [B][ctot][HT]; channels [coff, coff + C) are processed.
// grid = (C, B); stats2[c] += (sum g1, sum g1 * zhat, sum (z - mean_f32)), into one of kBwdStatSlots copies picked by the clip: 256 clips adding
// to the 3 cells of a channel serialised in one L2 channel (0.35 of this pass's 0.76 ms per step); the sums are exact integers (fixed
// point), so the slots add up to the same bits in bn_bwd_coef_kernel.
constexpr int kBwdStatSlots = 16;
// The third sum is what makes the gradient leave this block with sum(dz) == 0 to rounding: the consumers (weight
// gradients) multiply dz with activations that have a large common mean, so a per-channel offset of a few 1e-8 in dz
// would otherwise show up as a 1e-3 relative error in dW.
// (Round 3: this pass only READS.  It used to store g1 = ga * act'(z) back into g for the apply pass -- 179 MB per pitch convolution and step;
// bn_bwd_apply_kernel recomputes it from the same two values it loads anyway.)
__global__ __launch_bounds__(256) void act_bwd_stats_kernel(const float* __restrict__ g, const float* __restrict__ z,
                                                            const float* __restrict__ aff, const float* __restrict__ bstats,
                                                            double* __restrict__ stats2, long long slot_stride, int ctot, int coff, int HT) {
    const int c = blockIdx.x, clip = blockIdx.y;
    const long long base = (static_cast<long long>(clip) * ctot + coff + c) * HT;
    const float sc = aff[3 * c], sh = aff[3 * c + 1], ng = aff[3 * c + 2];
    const float mu = bstats[3 * c], rstd = rsqrtf(bstats[3 * c + 1] + 1e-5f);
    float s1 = 0.f, s2 = 0.f, s3 = 0.f;
    auto one = [&](float gg, float zz) {
        const float pre = fmaf(zz, sc, sh);
        const float g1 = gg * (pre > 0.f ? 1.f : ng);
        const float zc = zz - mu;
        s1 += g1;
        s2 = fmaf(g1, zc * rstd, s2);
        s3 += zc;
        return g1;
    };
    // 16-byte accesses when the slice allows it (the pass is memory-bound: 0.87 ms per step with 4-byte accesses)
    const bool vec = (HT & 3) == 0 && (base & 3) == 0 && ((reinterpret_cast<unsigned long long>(g) | reinterpret_cast<unsigned long long>(z)) & 15) == 0;
    if (vec) {
        const float4* g4 = reinterpret_cast<const float4*>(g + base);
        const float4* z4 = reinterpret_cast<const float4*>(z + base);
#pragma unroll 4
        for (int i = threadIdx.x; i < HT / 4; i += 256) {      // (unrolled: eight 16-byte loads in flight per thread)
            typedef float f32x4nt __attribute__((ext_vector_type(4)));
            const f32x4nt gn = __builtin_nontemporal_load(reinterpret_cast<const f32x4nt*>(g4) + i);
            const f32x4nt zn = __builtin_nontemporal_load(reinterpret_cast<const f32x4nt*>(z4) + i);
            const float4 gv = make_float4(gn[0], gn[1], gn[2], gn[3]), zv = make_float4(zn[0], zn[1], zn[2], zn[3]);
            (void)one(gv.x, zv.x); (void)one(gv.y, zv.y); (void)one(gv.z, zv.z); (void)one(gv.w, zv.w);
        }
    } else {
        for (int i = threadIdx.x; i < HT; i += 256) (void)one(g[base + i], z[base + i]);
    }
    __shared__ float r1[4], r2[4], r3[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); s3 += __shfl_xor(s3, o); }
    if ((threadIdx.x & 63) == 0) { r1[threadIdx.x >> 6] = s1; r2[threadIdx.x >> 6] = s2; r3[threadIdx.x >> 6] = s3; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double* const st = stats2 + static_cast<long long>(clip & (kBwdStatSlots - 1)) * slot_stride;
        fx_add(st + 3 * c, static_cast<double>(r1[0]) + r1[1] + r1[2] + r1[3], kFxGrad);
        fx_add(st + 3 * c + 1, static_cast<double>(r2[0]) + r2[1] + r2[2] + r2[3], kFxGrad);
        fx_add(st + 3 * c + 2, static_cast<double>(r3[0]) + r3[1] + r3[2] + r3[3], kFxStat);      // sum (z - mean): activation-sized
    }
}

// per channel: dz = c0 * g1 + c1 * (z - mean) + c2;  dgamma = S2, dbeta = S1.  coef[c] = (c0, c1, c2, mean)
// c2 is solved in double from the ROUNDED c0, c1 so that sum(dz) vanishes for the values the apply kernel really uses.
__global__ void bn_bwd_coef_kernel(const double* __restrict__ stats2, long long slot_stride, const float* __restrict__ bstats, const float* __restrict__ gamma,
                                   float* __restrict__ coef, gfx_t* __restrict__ d_gamma, gfx_t* __restrict__ d_beta, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double N = bstats[3 * c + 2];
    const double rstd = 1.0 / sqrt(static_cast<double>(bstats[3 * c + 1]) + 1e-5);
    long long i1 = 0, i2 = 0, i3 = 0;                              // the slots' fixed-point cells: exact integer adds
    for (int k = 0; k < kBwdStatSlots; ++k) {
        const long long* cell = reinterpret_cast<const long long*>(stats2 + k * slot_stride + 3 * c);
        i1 += cell[0]; i2 += cell[1]; i3 += cell[2];
    }
    const double S1 = static_cast<double>(i1) / kFxGrad, S2 = static_cast<double>(i2) / kFxGrad, S3 = static_cast<double>(i3) / kFxStat;
    const double k1 = gamma[c] * rstd;
    const float c0 = static_cast<float>(k1);
    const float c1 = static_cast<float>(-k1 * S2 / N * rstd);
    coef[4 * c] = c0;
    coef[4 * c + 1] = c1;
    coef[4 * c + 2] = static_cast<float>(-(static_cast<double>(c0) * S1 + static_cast<double>(c1) * S3) / N);
    coef[4 * c + 3] = bstats[3 * c];
    grad_add(d_gamma + c, static_cast<float>(S2));        // slot 0 of the gradient slots (the caller's buffer may be accumulating)
    grad_add(d_beta + c, static_cast<float>(S1));
}

// dz = c0*g1 + c1*(z - mean) + c2 in place on g; same layout / grid as act_bwd_stats_kernel.
// amax (nullable): the bits of the largest |dz| of the whole tensor (atomicMax on the bit pattern of a non-negative float: order-independent,
// so the step stays bit-reproducible).  The f16 x 3 data-gradient kernels that read this dz scale it by a power of two taken from it before
// the hi / lo split (at 256 clips x 76 frames most dz are 1e-6..1e-9, f16's subnormal range).
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(float* __restrict__ g, const float* __restrict__ z, const float* __restrict__ aff,
                                                           const float* __restrict__ coef, int ctot, int coff, int HT, unsigned int* __restrict__ amax) {
    const int c = blockIdx.x, clip = blockIdx.y;
    const long long base = (static_cast<long long>(clip) * ctot + coff + c) * HT;
    const float c0 = coef[4 * c], c1 = coef[4 * c + 1], c2 = coef[4 * c + 2], mu = coef[4 * c + 3];
    const float sc = aff[3 * c], sh = aff[3 * c + 1], ng = aff[3 * c + 2];
    auto g1 = [&](float ga, float zz) { return ga * (fmaf(zz, sc, sh) > 0.f ? 1.f : ng); };      // the activation's derivative, exactly as act_bwd_stats_kernel took it
    const bool vec = (HT & 3) == 0 && (base & 3) == 0 && ((reinterpret_cast<unsigned long long>(g) | reinterpret_cast<unsigned long long>(z)) & 15) == 0;
    float m = 0.f;
    if (vec) {
        float4* g4 = reinterpret_cast<float4*>(g + base);
        const float4* z4 = reinterpret_cast<const float4*>(z + base);
#pragma unroll 4
        for (int i = threadIdx.x; i < HT / 4; i += 256) {
            // (both operands are read for the last time here -- the result replaces g -- so the loads are streaming, like act_bwd_stats_kernel's:
            //  that pass 0.356 -> 0.306 ms per step with them)
            typedef float f32x4nt __attribute__((ext_vector_type(4)));
            const f32x4nt gn = __builtin_nontemporal_load(reinterpret_cast<const f32x4nt*>(g4) + i);
            const f32x4nt zn = __builtin_nontemporal_load(reinterpret_cast<const f32x4nt*>(z4) + i);
            float4 gv = make_float4(gn[0], gn[1], gn[2], gn[3]);
            const float4 zv = make_float4(zn[0], zn[1], zn[2], zn[3]);
            gv.x = fmaf(g1(gv.x, zv.x), c0, fmaf(zv.x - mu, c1, c2)); gv.y = fmaf(g1(gv.y, zv.y), c0, fmaf(zv.y - mu, c1, c2));
            gv.z = fmaf(g1(gv.z, zv.z), c0, fmaf(zv.z - mu, c1, c2)); gv.w = fmaf(g1(gv.w, zv.w), c0, fmaf(zv.w - mu, c1, c2));
            m = fmaxf(fmaxf(m, fmaxf(fabsf(gv.x), fabsf(gv.y))), fmaxf(fabsf(gv.z), fabsf(gv.w)));
            g4[i] = gv;
        }
    } else {
        for (int i = threadIdx.x; i < HT; i += 256) {
            const float v = fmaf(g1(g[base + i], z[base + i]), c0, fmaf(z[base + i] - mu, c1, c2));
            m = fmaxf(m, fabsf(v));
            g[base + i] = v;
        }
    }
    if (!amax) return;      // [kAmaxSlots] cells, one per group of clips: thousands of atomics on ONE address serialise in its L2 channel (1.5 ms per step)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    __shared__ float wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
        if (m > 0.f && m < INFINITY) atomicMax(amax + (clip & (kAmaxSlots - 1)), __float_as_uint(m));
    }
}

// sum over (clip, positions) of one channel slice -> bias gradient of a convolution without BatchNorm
__global__ __launch_bounds__(256) void channel_sum_kernel(const float* __restrict__ g, gfx_t* __restrict__ out, long long slot_stride, int ctot, int coff,
                                                          int HT) {
    const int c = blockIdx.x, clip = blockIdx.y;
    const long long base = (static_cast<long long>(clip) * ctot + coff + c) * HT;
    float s = 0.f;
    for (int i = threadIdx.x; i < HT; i += 256) s += g[base + i];
    __shared__ float r[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) r[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) grad_add(grad_slot(out, slot_stride) + c, r[0] + r[1] + r[2] + r[3]);
}

// ---- --pc2p_mem (models.py:145-166) backward: the pitch stream's gradient passes through unchanged; every channel of the activated
// up_sixth map that was summed into pitch channel c = cpc / ratio collects the rows r with r / (P / 36) == k of that channel's gradient.
__global__ void pc2p_mem_bwd_kernel(const float* __restrict__ g_pin, float* __restrict__ g_psix, int cp, int ratio, int P, int T, long long total) {
    const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;      // (clip, cpc, k, t)
    if (i >= total) return;
    const int t = static_cast<int>(i % T);
    long long q = i / T;
    const int k = static_cast<int>(q % 36);
    q /= 36;
    const int cpc = static_cast<int>(q % (cp * ratio));
    const long long clip = q / (cp * ratio);
    const int n = P / 36;
    const float* s = g_pin + ((clip * cp + cpc / ratio) * P + static_cast<long long>(k) * n) * T + t;
    float acc = 0.f;
    for (int j = 0; j < n; ++j) acc += s[static_cast<long long>(j) * T];
    g_psix[i] = acc;
}

// ---- --p2pc_conv (models.py:108-133) backward: the octave-fold convolution, w[co][ci][o], dz [clip][C][12][T] --------------------
// weight: dW[co][ci][o] += sum_{p,t} dz[co][p][t] * act(x[ci][12 o + p][t])          grid (C*C*n_oct, B)
__global__ __launch_bounds__(64) void fold_conv_bwd_weight_kernel(const float* __restrict__ dz, const float* __restrict__ x, const float* __restrict__ x_aff,
                                                                  gfx_t* __restrict__ dW, long long slot_stride, int C, int n_oct, int T) {
    const int widx = blockIdx.x;                 // (co, ci, o)
    const int clip = blockIdx.y;
    const int o = widx % n_oct, ci = (widx / n_oct) % C, co = widx / (n_oct * C);
    const float* d = dz + (static_cast<long long>(clip) * C + co) * 12 * T;
    const float* xs = x + ((static_cast<long long>(clip) * C + ci) * 12 * n_oct + 12 * o) * T;
    float acc = 0.f;
    for (int i = threadIdx.x; i < 12 * T; i += 64) acc = fmaf(d[i], affine_act(xs[i], x_aff, ci), acc);
#pragma unroll
    for (int k = 32; k > 0; k >>= 1) acc += __shfl_xor(acc, k);
    if (threadIdx.x == 0) grad_add(grad_slot(dW, slot_stride) + widx, acc);
}
// data: ga[ci][12 o + p][t] = sum_co dz[co][p][t] * w[co][ci][o]      (writes every element of ga)
__global__ void fold_conv_bwd_data_kernel(const float* __restrict__ dz, const float* __restrict__ w, float* __restrict__ ga, int C, int n_oct, int T,
                                          long long total) {
    const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;      // (clip, ci, row, t)
    if (i >= total) return;
    const int t = static_cast<int>(i % T);
    long long q = i / T;
    const int row = static_cast<int>(q % (12 * n_oct));
    q /= 12 * n_oct;
    const int ci = static_cast<int>(q % C);
    const long long clip = q / C;
    const int o = row / 12, p = row - 12 * o;
    float acc = 0.f;
    for (int co = 0; co < C; ++co) acc = fmaf(dz[((clip * C + co) * 12 + p) * T + t], w[(co * C + ci) * n_oct + o], acc);
    ga[i] = acc;
}

// ---- --resblock: the activation behind the residual add -----------------------------------------------------------------
// x_out = LeakyReLU(s), s = b2(z2) + x_in (sign(x_out) = sign(s)): g <- g * LeakyReLU'(s) is the gradient of BOTH summands; the copy in
// g_skip travels down the skip connection while g goes through b2 / conv2 / b1 / conv1.
__global__ void res_act_bwd_kernel(float* __restrict__ g, const float* __restrict__ x_out, float* __restrict__ g_skip, long long total) {
    const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const float v = g[i] * (x_out[i] > 0.f ? 1.f : kSlope);
    g[i] = v;
    g_skip[i] = v;
}

// ---- pooling / fold / repeat routing -------------------------------------------------------------------------------
// MaxPool2d((1,tp)) backward: the pooled input was a = act(z) (aff may be null = already final); the gradient goes to the
// first maximum of each window, everything else (and the floor tail) gets zero.  One thread per pooled element.  accumulate: ga
// already holds the gradient of the activation's other consumer (an inner layer's pitch stream also feeds pool_semi): add to it.
__global__ void time_pool_bwd_kernel(const float* __restrict__ gp, const float* __restrict__ z, const float* __restrict__ aff,
                                     float* __restrict__ ga, int C, int H, int T, int tp, int gp_ctot, int gp_coff, long long total, int accumulate) {
    const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int To = T / tp;
    const int nt = To + (T - To * tp > 0 ? 1 : 0);                 // one extra "window" zero-fills the tail
    const int t = static_cast<int>(i % nt);
    long long q = i / nt;
    const int y = static_cast<int>(q % H);
    q /= H;
    const int c = static_cast<int>(q % C);
    const long long clip = q / C;
    const long long zb = ((clip * C + c) * H + y) * T;
    if (t >= To) {
        for (int j = To * tp; j < T && !accumulate; ++j) ga[zb + j] = 0.f;
        return;
    }
    int best = 0;
    float bv = affine_act(z[zb + static_cast<long long>(t) * tp], aff, c);
    for (int j = 1; j < tp; ++j) {
        const float v = affine_act(z[zb + static_cast<long long>(t) * tp + j], aff, c);
        if (v > bv) { bv = v; best = j; }
    }
    const float g = gp[((clip * gp_ctot + gp_coff + c) * H + y) * To + t];
    if (accumulate) { ga[zb + static_cast<long long>(t) * tp + best] += g; return; }
    for (int j = 0; j < tp; ++j) ga[zb + static_cast<long long>(t) * tp + j] = j == best ? g : 0.f;
}

// Pitch2PitchClassPool backward: route gfold[c][p][t] (channel slice of a concat gradient) to the arg-max octave.
__global__ void fold_bwd_kernel(const float* __restrict__ gfold, const float* __restrict__ z, const float* __restrict__ aff,
                                float* __restrict__ ga, int C, int n_oct, int T, int g_ctot, int g_coff, long long total) {
    const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int t = static_cast<int>(i % T);
    long long q = i / T;
    const int p = static_cast<int>(q % 12);
    q /= 12;
    const int c = static_cast<int>(q % C);
    const long long clip = q / C;
    const long long zb = ((clip * C + c) * (12 * n_oct) + p) * T + t;
    int best = 0;
    float bv = affine_act(z[zb], aff, c);
    for (int o = 1; o < n_oct; ++o) {
        const float v = affine_act(z[zb + static_cast<long long>(o) * 12 * T], aff, c);
        if (v > bv) { bv = v; best = o; }
    }
    const float g = gfold[((clip * g_ctot + g_coff + c) * 12 + p) * T + t];
    for (int o = 0; o < n_oct; ++o) ga[zb + static_cast<long long>(o) * 12 * T] = o == best ? g : 0.f;
}

// PitchClass2Pitch backward: gpsix[c][r][t] = sum_o gin[cp + c][36*o + r][t]   (gin = gradient of the pitch-conv input)
__global__ void repeat_sum_kernel(const float* __restrict__ gin, float* __restrict__ gps, int cin_tot, int cp, int C, int P, int T,
                                  long long total) {
    const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int t = static_cast<int>(i % T);
    long long q = i / T;
    const int r = static_cast<int>(q % 36);
    q /= 36;
    const int c = static_cast<int>(q % C);
    const long long clip = q / C;
    const float* s = gin + ((clip * cin_tot + cp + c) * P + r) * T + t;
    float acc = 0.f;
    for (int o = 0; o < P / 36; ++o) acc += s[static_cast<long long>(o) * 36 * T];
    gps[i] = acc;
}

// --stay_sixth (models.py:322-323, 379-383): the pitch classes themselves were repeated over the octaves of the semitone-resolution pitch
// stream -- their gradient, summed over the octaves, joins what the pitch-class stack left in channels [0, C) of the concat gradient.
__global__ void repeat_sum_add_kernel(const float* __restrict__ gin, float* __restrict__ gcat, int cin_tot, int cp, int C, int H, int T, int g_ctot,
                                      long long total) {
    const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;      // (clip, c, pitch class, t)
    if (i >= total) return;
    const int t = static_cast<int>(i % T);
    long long q = i / T;
    const int r = static_cast<int>(q % 12);
    q /= 12;
    const int c = static_cast<int>(q % C);
    const long long clip = q / C;
    const float* s = gin + ((clip * cin_tot + cp + c) * H + r) * T + t;
    float acc = 0.f;
    for (int o = 0; o < H / 12; ++o) acc += s[static_cast<long long>(o) * 12 * T];
    gcat[((clip * g_ctot + c) * 12 + r) * T + t] += acc;
}

// dst[clip][c][i] += src[clip][c (of src_ctot)][i]: a second consumer's gradient joins a dense one (--stay_sixth: layer 0's semitone map is
// also layer 1's pitch stream)
__global__ void add_slice_kernel(float* __restrict__ dst, const float* __restrict__ src, int C, long long HT, int src_ctot, long long total) {
    const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const long long ht = i % HT;
    const long long q = i / HT;
    const int c = static_cast<int>(q % C);
    const long long clip = q / C;
    dst[i] += src[(clip * src_ctot + c) * HT + ht];
}

// ---- semitone conv (3x3, stride (3,1), frames circular) backward ----------------------------------------------------
// data: ga[ci][3s+dy][t] = sum_{co,dx} dz[co][s][(t - dx + 1) mod T] * w[co][ci][dy][dx]      (w in reference layout)
// A thread owns one position (clip, row, frame) and up to 8 input channels at a time: the three dz values of an output channel are read once
// for all of them (one thread per (channel, position) read every dz value 8 x 3 times through L2: 0.375 ms per step, latency-bound).  Every
// output still adds its products in the order (co, dx): bit-identical to the per-channel form.   total = clips * H * T
__global__ __launch_bounds__(256) void semi_bwd_data_kernel(const float* __restrict__ dz, const float* __restrict__ w, float* __restrict__ ga, int C, int H,
                                                            int T, long long total) {
    extern __shared__ float sw_lds[];                              // the raw weights [co][ci][3][3]
    for (int k = threadIdx.x; k < C * C * 9; k += blockDim.x) sw_lds[k] = w[k];
    __syncthreads();
    const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int t = static_cast<int>(i % T);
    const long long q = i / T;
    const int r = static_cast<int>(q % H);
    const long long clip = q / H;
    const int s = r / 3, dy = r - 3 * s;
    const int tp = wrap(t + 1, T), tm = wrap(t - 1, T);
    for (int ci0 = 0; ci0 < C; ci0 += 8) {
        const int nci = C - ci0 < 8 ? C - ci0 : 8;
        float acc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = 0.f;
        for (int co = 0; co < C; ++co) {
            const float* drow = dz + ((clip * C + co) * (H / 3) + s) * T;
            const float d0 = drow[tp], d1 = drow[t], d2 = drow[tm];            // dx = 0, 1, 2 <-> frame t - dx + 1
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (k < nci) {
                    const float* wp = sw_lds + ((co * C + ci0 + k) * 3 + dy) * 3;
                    acc[k] = fmaf(d2, wp[2], fmaf(d1, wp[1], fmaf(d0, wp[0], acc[k])));
                }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (k < nci) ga[((clip * C + ci0 + k) * H + r) * T + t] = acc[k];
    }
}

// weight: dW[co][ci][dy][dx] += sum_{s,t} dz[co][s][t] * act(x[ci][3s+dy][(t+dx-1) mod T])
// Workgroup = (clip, group of rows_per_wg <= kSemiRows output rows: fewer per workgroup at small batches, so that the chip fills), 256 threads = 4 waves; wave w takes rows w, w+4, ...: it stages the
// dz row of every output channel and the three activated input rows of every input channel in its own LDS slice (with
// the circular time halo), then lane (co, ci) accumulates its 9 taps over the frames.  One atomic per weight and workgroup.
constexpr int kSemiRows = 48;

__global__ __launch_bounds__(256) void semi_bwd_weight_kernel(const float* __restrict__ dz, const float* __restrict__ x,
                                                              const float* __restrict__ x_aff, gfx_t* __restrict__ dW, long long slot_stride, int C,
                                                              int H, int T, int rows_per_wg) {
    extern __shared__ float semi_lds[];
    const int clip = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int S = H / 3;
    const int Tp = T + 2;
    float* ldz = semi_lds + wave * (C * T + 3 * C * Tp);      // [co][T]
    float* lx = ldz + C * T;                                  // [ci][3][T + 2]   (index 0 <-> frame -1)
    const unsigned int magicT = 0xFFFFFFFFu / static_cast<unsigned int>(T) + 1u;        // floor(i / T) = umulhi(i, magic) for i < 2^16
    const unsigned int magicTp = 0xFFFFFFFFu / static_cast<unsigned int>(Tp) + 1u;
    const int pairs = C * C;
    const int pair = blockIdx.z * 64 + lane;                  // wider layers (C > 8): blockIdx.z walks the (co, ci) pairs 64 at a time
    const int co = pair / C, ci = pair - co * C;
    float acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = 0.f;
    const int s_end = min(S, static_cast<int>(blockIdx.x + 1) * rows_per_wg);
    for (int srow = blockIdx.x * rows_per_wg + wave; srow < s_end; srow += 4) {
        // (index splits by multiply-high with precomputed reciprocals: two integer divisions per staged value were a third of the kernel)
        // Eight loads in flight per lane (round 3): with one load per loop iteration every value waited out a full memory round trip before its
        // LDS store -- 40 dependent round trips per row and wave were this launch's 0.41 ms per step (its arithmetic is a tenth of that).  The
        // loads are branch-free (clamped index), only the stores are guarded.
        constexpr int kU = 8;
        for (int i0 = lane; i0 < C * T; i0 += 64 * kU) {
            float v[kU];
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                const int i = min(i0 + 64 * u, C * T - 1);
                const int c = static_cast<int>(__umulhi(static_cast<unsigned int>(i), magicT)), t = i - c * T;
                v[u] = dz[((static_cast<long long>(clip) * C + c) * S + srow) * T + t];
            }
#pragma unroll
            for (int u = 0; u < kU; ++u)
                if (i0 + 64 * u < C * T) ldz[i0 + 64 * u] = v[u];
        }
        for (int i0 = lane; i0 < 3 * C * Tp; i0 += 64 * kU) {
            float v[kU];
            int cs[kU];
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                const int i = min(i0 + 64 * u, 3 * C * Tp - 1);
                const int cr = static_cast<int>(__umulhi(static_cast<unsigned int>(i), magicTp));      // (c, r) = i / Tp
                const int c = static_cast<int>(__umulhi(static_cast<unsigned int>(cr), 0x55555556u)), r = cr - 3 * c, tj = i - cr * Tp;
                int t = tj - 1;
                t += t < 0 ? T : 0;
                t -= t >= T ? T : 0;
                cs[u] = c;
                v[u] = x[((static_cast<long long>(clip) * C + c) * H + 3 * srow + r) * T + t];
            }
#pragma unroll
            for (int u = 0; u < kU; ++u)
                if (i0 + 64 * u < 3 * C * Tp) lx[i0 + 64 * u] = affine_act(v[u], x_aff, cs[u]);
        }
        // a wave's LDS slice is private: no workgroup barrier, the waitcnt the compiler inserts for the reads is enough
        if (pair < pairs) {
            const float* dr = ldz + co * T;
            const float* xr = lx + ci * 3 * Tp;
            // sliding window over the frames: the three taps of a row share two of their three values with the previous frame (4 LDS reads per
            // frame instead of 10; every accumulator still receives its products in frame order: bit-identical sums)
            float x0[3], x1[3];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) { x0[dy] = xr[dy * Tp]; x1[dy] = xr[dy * Tp + 1]; }
#pragma unroll 4
            for (int t = 0; t < T; ++t) {                          // (unrolled: the 16 LDS reads of four frames are requested together)
                const float d = dr[t];
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const float x2 = xr[dy * Tp + t + 2];
                    acc[dy * 3 + 0] = fmaf(d, x0[dy], acc[dy * 3 + 0]);
                    acc[dy * 3 + 1] = fmaf(d, x1[dy], acc[dy * 3 + 1]);
                    acc[dy * 3 + 2] = fmaf(d, x2, acc[dy * 3 + 2]);
                    x0[dy] = x1[dy]; x1[dy] = x2;
                }
            }
        }
    }
    // one atomic per weight and WORKGROUP: 4096 adders on the 18 cache lines of dW serialise in L2 (measured 1.6 ms when
    // every wave added its own partial sums)
    __syncthreads();
    float* red = semi_lds;                                        // [4 waves][9][64]
    if (pair < pairs) {
#pragma unroll
        for (int k = 0; k < 9; ++k) red[(wave * 9 + k) * 64 + lane] = acc[k];
    }
    __syncthreads();
    if (wave == 0 && pair < pairs) {
#pragma unroll
        for (int k = 0; k < 9; ++k)
            grad_add(grad_slot(dW, slot_stride) + (co * C + ci) * 9 + k, (red[k * 64 + lane] + red[(9 + k) * 64 + lane]) + (red[(18 + k) * 64 + lane] + red[(27 + k) * 64 + lane]));
    }
}

// ---- up_sixth (ConvTranspose2d (3,1)/(3,1)) backward -----------------------------------------------------------------
// data: ga_pc[ci][p][t] += sum_{co,j} dz[co][3p+j][t] * w[ci][co][j]      (accumulates into a channel slice of a concat gradient)
__global__ void up_sixth_bwd_data_kernel(const float* __restrict__ dz, const float* __restrict__ w, float* __restrict__ gcat, int g_ctot, int C,
                                         int T, long long total) {
    const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int t = static_cast<int>(i % T);
    long long q = i / T;
    const int p = static_cast<int>(q % 12);
    q /= 12;
    const int ci = static_cast<int>(q % C);
    const long long clip = q / C;
    float acc = 0.f;
    for (int co = 0; co < C; ++co)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc = fmaf(dz[((clip * C + co) * 36 + 3 * p + j) * T + t], w[(ci * C + co) * 3 + j], acc);
    gcat[((clip * g_ctot + ci) * 12 + p) * T + t] += acc;
}

// weight: dW[ci][co][j] += sum_{p,t} dz[co][3p+j][t] * act(x[ci][p][t])     grid (C*C*3, B)
__global__ __launch_bounds__(64) void up_sixth_bwd_weight_kernel(const float* __restrict__ dz, const float* __restrict__ x, long long x_clip_stride,
                                                                 const float* __restrict__ x_aff, gfx_t* __restrict__ dW, long long slot_stride, int C,
                                                                 int T) {
    const int widx = blockIdx.x;                 // (ci, co, j)
    const int clip = blockIdx.y;
    const int j = widx % 3, co = (widx / 3) % C, ci = widx / (3 * C);
    const float* d = dz + (static_cast<long long>(clip) * C + co) * 36 * T;
    const float* xs = x + clip * x_clip_stride + static_cast<long long>(ci) * 12 * T;
    float acc = 0.f;
    for (int i = threadIdx.x; i < 12 * T; i += 64) {
        const int p = i / T, t = i - p * T;
        acc = fmaf(d[(3 * p + j) * T + t], affine_act(xs[i], x_aff, ci), acc);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (threadIdx.x == 0) grad_add(grad_slot(dW, slot_stride) + widx, acc);
}

// ---- convolution weight gradient on f32 MFMA --------------------------------------------------------------------------
//   dW[co][ci][dy][dx] = sum_{clip, y, t} dz[co][y][t] * act(in[ci][(y + dy - py) mod H][t + dx - pad_l])
// GEMM view per input channel: D[m = co][n = (dy,dx)] += sum_{k = position} A[m][k] * B[k][n]
//   A[co][pos] = dz tile in LDS,   B[pos][(dy,dx)] = the forward's input patch in LDS read at (y+dy, t+dx)
// Workgroup = (clip, group of row tiles); wave w owns input channel c_lo + w of the current chunk of <= 8 channels and keeps
// MTC x NTK accumulator tiles (16 output channels x 16 kernel taps each) across all row tiles; they are flushed with float
// atomics once per (clip group, chunk).
struct WgradArgs {
    ConvArgs c;               // forward geometry; c.dst = dz (read), c.dst_coff / dst_clip_stride address it; c.w unused
    gfx_t* dW;                // [cout][cin][KH][KW] (+=), slot 0
    long long slot_stride;    // floats between gradient slots
    int KH, KW;
    int rt_per_block;         // row tiles per workgroup
    int c_per_block;          // input channels per workgroup (cin: gridDim.y == 1; waves per workgroup: blockIdx.y walks the channel groups)
    int dbg_noflush;          // timing experiments only (AKE_WGRAD_NOFLUSH): skip the atomics
    // nullable: every workgroup stores its partial dW as plain floats at partial[(blockIdx.z * gridDim.x + blockIdx.x) * partial_stride + ...]
    // and wgrad_partial_reduce_kernel adds them up in workgroup order (deterministic; one atomic per weight instead of one per weight
    // and workgroup: 6.3 M 64-bit atomics per pitch-class convolution and step cost 0.3 of its 1.1 ms)
    float* partial;
    long long partial_stride;
};

// dW[i] += sum over the workgroups' partial sums in a fixed order: 16 groups of consecutive workgroups are summed side by side (a lane
// per weight and group: 43 k weights alone leave most of the chip idle behind 256 dependent loads), then combined in group order
__global__ __launch_bounds__(1024) void wgrad_partial_reduce_kernel(const float* __restrict__ partial, int n_wg, long long stride, gfx_t* __restrict__ dW) {
    __shared__ double part[16][64];
    const int li = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const long long i = static_cast<long long>(blockIdx.x) * 64 + li;
    const int per = (n_wg + 15) / 16;
    const int w0 = grp * per, w1 = w0 + per < n_wg ? w0 + per : n_wg;
    double s = 0.0;
    if (i < stride) {
        int w = w0;
        for (; w + 4 <= w1; w += 4) {
            const float a0 = partial[(w + 0) * stride + i], a1 = partial[(w + 1) * stride + i], a2 = partial[(w + 2) * stride + i],
                        a3 = partial[(w + 3) * stride + i];
            s += static_cast<double>(a0); s += static_cast<double>(a1); s += static_cast<double>(a2); s += static_cast<double>(a3);
        }
        for (; w < w1; ++w) s += static_cast<double>(partial[w * stride + i]);
    }
    part[grp][li] = s;
    __syncthreads();
    if (grp == 0 && i < stride) {
        double t = 0.0;
#pragma unroll
        for (int g = 0; g < 16; ++g) t += part[g][li];
        grad_add(dW + i, static_cast<float>(t));
    }
}

template <int MTC, int NTK>
__global__ __launch_bounds__(512) void conv_wgrad_kernel(WgradArgs wa) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const ConvArgs& a = wa.c;
    const int KH = wa.KH, KW = wa.KW, KK = KH * KW;
    const int cin = a.c0 + a.c1;
    const int clip = blockIdx.z;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    const int r16 = lane & 15, q = lane >> 4;
    const int R_in = a.R + KH - 1;
    const int Tp = a.Tp;                       // patch pitch (>= TT + KW - 1)
    const int TTp = a.TT;                      // dz tile pitch (multiple of 4)
    const int cstride = R_in * Tp;
    float* const ldsZ = lds + nw * cstride;    // dz tile [cout][R][TTp]
    const float* s0 = a.src0 + clip * a.src0_clip_stride;
    const float* s1 = a.src1 ? a.src1 + clip * a.src1_clip_stride : nullptr;
    const float* dzc = a.dst + clip * a.dst_clip_stride;
    const int ncb = (Tp + 63) >> 6;
    const bool fast_wrap = a.T_in >= Tp;
    // per-lane kernel-tap offsets of the B operand: column c16 of N-tile nt <-> tap k' = 16*nt + c16 -> (dy, dx)
    int koff[NTK];
#pragma unroll
    for (int nt = 0; nt < NTK; ++nt) {
        int kk = 16 * nt + r16;
        if (kk >= KK) kk = KK - 1;             // padded columns read a valid address, never stored
        const int dy = kk / KW;
        koff[nt] = dy * Tp + (kk - dy * KW);
    }
    const int tile0 = blockIdx.x * wa.rt_per_block;

    // small batches: blockIdx.y walks the groups of nw input channels (c_per_block = nw), otherwise one workgroup loops over all of them
    const int c_begin = blockIdx.y * wa.c_per_block;
    const int c_end = c_begin + wa.c_per_block < cin ? c_begin + wa.c_per_block : cin;
    for (int c_lo = c_begin; c_lo < c_end; c_lo += nw) {
        const int cc = c_end - c_lo < nw ? c_end - c_lo : nw;
        f32x4 acc[MTC][NTK];
#pragma unroll
        for (int m = 0; m < MTC; ++m)
#pragma unroll
            for (int nt = 0; nt < NTK; ++nt) acc[m][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int ti = 0; ti < wa.rt_per_block; ++ti) {
            const int tile = tile0 + ti;
            const int row_tile = tile / a.n_time_tiles;
            if (row_tile >= a.n_row_tiles) break;
            const int time_tile = tile - row_tile * a.n_time_tiles;
            const int y0 = row_tile * a.R, t0 = time_tile * a.TT;
            const int rows_here = a.H_out - y0 < a.R ? a.H_out - y0 : a.R;
            const int tt_here = a.T_out - t0 < a.TT ? a.T_out - t0 : a.TT;
            __syncthreads();
            // ---- stage the input patch of channels [c_lo, c_lo+cc) (same loader as the forward conv) ----
            {
                // (four patch rows in flight per wave -- up to 12 loads per lane -- instead of one dependent round trip per row; loads branch-free
                // through clamped addresses, validity re-derived for the stores)
                const int nrows = cc * R_in;
                constexpr int kU = 4, kH = 3;                          // (Tp <= 192: at most three 64-frame pieces per row)
                struct RowGeom { const float* srow; int cs; bool row_ok; };
                auto row_geom = [&](int rr) {
                    const int cl = rr / R_in, rj = rr - cl * R_in;
                    RowGeom g;
                    g.cs = c_lo + cl;
                    int row = y0 - a.py + rj;
                    g.row_ok = !a.rows_zero || (row >= 0 && row < a.H);       // rows_zero: zero padding instead of the circular wrap
                    row += row < 0 ? a.H : 0;
                    row -= row >= a.H ? a.H : 0;
                    g.srow = g.cs < a.c0 ? s0 + (static_cast<long long>(g.cs) * a.H + row) * a.T_in
                                         : s1 + (static_cast<long long>(g.cs - a.c0) * a.h1 + (row % a.h1)) * a.T_in;
                    return g;
                };
                auto frame_of = [&](int tj, bool& ok) {                // input frame of patch column tj (clamped into the row when outside: ok = false)
                    int tin = t0 - a.pad_l + tj;
                    ok = true;
                    if (a.time_circ) {
                        if (fast_wrap) { tin += tin < 0 ? a.T_in : 0; tin -= tin >= a.T_in ? a.T_in : 0; }
                        else tin = wrap(tin, a.T_in);
                    } else {
                        ok = tin >= 0 && tin < a.T_in;
                        tin = tin < 0 ? 0 : (tin >= a.T_in ? a.T_in - 1 : tin);
                    }
                    return tin;
                };
                for (int r0 = wave; r0 < nrows; r0 += nw * kU) {
                    float v[kU][kH];
#pragma unroll
                    for (int u = 0; u < kU; ++u) {
                        const RowGeom g = row_geom(min(r0 + nw * u, nrows - 1));
#pragma unroll
                        for (int h = 0; h < kH; ++h) {
                            bool ok;
                            v[u][h] = h < ncb ? g.srow[frame_of(min(lane + 64 * h, Tp - 1), ok)] : 0.f;
                        }
                    }
#pragma unroll
                    for (int u = 0; u < kU; ++u) {
                        const int rr = r0 + nw * u;
                        if (rr >= nrows) break;
                        const RowGeom g = row_geom(rr);
                        float asc = 1.f, ash = 0.f, ang = 1.f;
                        if (a.in_affine) { asc = a.in_affine[3 * g.cs]; ash = a.in_affine[3 * g.cs + 1]; ang = a.in_affine[3 * g.cs + 2]; }
#pragma unroll
                        for (int h = 0; h < kH; ++h) {
                            const int tj = lane + 64 * h;
                            if (h >= ncb || tj >= Tp) break;
                            bool ok;
                            (void)frame_of(tj, ok);
                            float val = 0.f;
                            if (ok && g.row_ok) { const float x = fmaf(v[u][h], asc, ash); val = x > 0.f ? x : x * ang; }
                            lds[rr * Tp + tj] = val;
                        }
                    }
                }
                // ---- dz tile, zero outside the valid rows / frames ----
                const int nz = a.cout * a.R;
                for (int r0 = wave; r0 < nz; r0 += nw * kU) {
                    float v[kU][2];                                    // (TTp <= 128: two pieces)
#pragma unroll
                    for (int u = 0; u < kU; ++u) {
                        const int rr = min(r0 + nw * u, nz - 1);
                        const int co = rr / a.R, ry = min(rr - co * a.R, rows_here - 1);
                        const float* zrow = dzc + (static_cast<long long>(a.dst_coff + co) * a.H_out + y0 + ry) * a.T_out + t0;
#pragma unroll
                        for (int h = 0; h < 2; ++h) v[u][h] = zrow[min(lane + 64 * h, tt_here - 1)];
                    }
#pragma unroll
                    for (int u = 0; u < kU; ++u) {
                        const int rr = r0 + nw * u;
                        if (rr >= nz) break;
                        const int co = rr / a.R, ry = rr - co * a.R;
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int tj = lane + 64 * h;
                            if (tj < TTp) ldsZ[rr * TTp + tj] = (ry < rows_here && tj < tt_here) ? v[u][h] : 0.f;
                        }
                    }
                }
            }
            __syncthreads();
            if (wave >= cc) continue;
            const float* patch = lds + wave * cstride;
            for (int y = 0; y < rows_here; ++y) {
                for (int ts = 0; ts < TTp; ts += 4) {
                    float av[MTC], bv[NTK];
#pragma unroll
                    for (int m = 0; m < MTC; ++m) {
                        const int co = 16 * m + r16;
                        av[m] = co < a.cout ? ldsZ[(co * a.R + y) * TTp + ts + q] : 0.f;
                    }
#pragma unroll
                    for (int nt = 0; nt < NTK; ++nt) bv[nt] = patch[y * Tp + ts + q + koff[nt]];
#pragma unroll
                    for (int m = 0; m < MTC; ++m)
#pragma unroll
                        for (int nt = 0; nt < NTK; ++nt) acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], bv[nt], acc[m][nt], 0, 0, 0);
                }
            }
        }
        // ---- flush: D[row = co 4q+reg][col = tap c16] ----
        gfx_t* const dWs = grad_slot(wa.dW, wa.slot_stride);
        if (wave < cc) {
            const int ci = c_lo + wave;
#pragma unroll
            for (int m = 0; m < MTC; ++m)
#pragma unroll
                for (int nt = 0; nt < NTK; ++nt) {
                    const int kk = 16 * nt + r16;
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const int co = 16 * m + 4 * q + reg;
                        if (co < a.cout && kk < KK && !wa.dbg_noflush) {
                            const long long idx = (static_cast<long long>(co) * cin + ci) * KK + kk;
                            if (wa.partial) wa.partial[(static_cast<long long>(blockIdx.z) * gridDim.x + blockIdx.x) * wa.partial_stride + idx] = acc[m][nt][reg];
                            else grad_add(dWs + idx, acc[m][nt][reg]);
                        }
                    }
                }
        }
    }
}

// ---- weight gradient of the 7x7 circular pitch convolution on bf16 MFMA with split operands ---------------------------------
// The f32 kernel above keeps one input channel per wave and gets 38 % useful MFMA work (8 of 16 rows, 49 of 64 columns) at the
// vector rate: 1.2 ms per convolution and 256 clips, the largest kernel of a training step.  Here ONE row of the activated input
// ("a-row" ya) and the 7 dz rows it meets are a complete 56 x 56 GEMM per row:
//   D[m = (dy, co)][n = (dx, ci)] += sum_t  dz[co][ya - dy + 3][t]  *  a[ci][ya][t + dx - 3]         (rows / frames circular)
// i.e. the whole dW (8 x 8 x 7 x 7) as 4 x 4 tiles of 16 x 16, K = the frames of the row (3 k-steps of 32), three MFMAs per
// tile and k-step.  The time shift dx is resolved when the a-row is staged: it is written to LDS seven times, rotated, so every
// B fragment is an aligned 16-byte read; dz rows sit in a ring of 8.  Both operands are read as raw f32 (the pending
// BatchNorm + LeakyReLU of the input applied on the fly) and split into bf16 hi / lo while staging.
// Workgroup = (clip, block of a-rows), 4 waves = the 4 M-tiles; partial sums go to the gradient slots with one atomic per weight.
constexpr int kWgKP = 104;                  // LDS row pitch in bf16 (>= 96 frames; 52 dwords: conflict-free 16-byte reads)
constexpr int kWgMaxT = 96;
constexpr int kWgARows = 56;                // rows of the rotated a-row copy: 7 dx x 8 ci

struct WgradBfArgs {
    const float* src0;        // [clip][c0][H][T] raw
    const float* src1;        // [clip][c1][h1][T] raw (rows repeat: row % h1), or null
    long long src0_clip_stride, src1_clip_stride;
    const float* in_affine;   // [cin][3] or null
    const float* dz;          // [clip][8][H][T]
    gfx_t* dW;                // [8][cin][7][7], slot 0
    long long slot_stride;
    int c0, c1, h1, cin, H, T, rows_per_wg;
    // nullable: every workgroup stores its partial dW as plain floats at partial[(clip * gridDim.x + blockIdx.x) * partial_stride + weight] and
    // wgrad_partial_reduce_kernel adds them in workgroup order -- instead of 3 136 fixed-point atomics per workgroup into the SAME cells from
    // 1 024 workgroups (round 3: that flush, not the multiply, was most of this launch's 0.41 ms per convolution)
    float* partial;
    long long partial_stride;
};

__global__ __launch_bounds__(256) void conv_wgrad_p2p_bf16_kernel(WgradBfArgs a) {
    // two copies of the rotated a-row [8 dx][8 ci][kWgKP] (hi, lo) -- the next row is written while this one is multiplied --,
    // the dz ring [8 slots][8 co][kWgKP] (hi, lo; 7 slots live, the 8th receives the next row), one zero row
    extern __shared__ __attribute__((aligned(16))) unsigned short wg_lds[];
    // ONE copy of the rotated a-row, 56 rows (7 dx x 8 ci): 50 KB per workgroup, three per CU.  (Two copies of 64 rows -- the next row
    // written while this one is multiplied, one barrier per row -- were 80 KB, two workgroups per CU: counters showed the matrix pipes
    // 17-20 % busy and 56 % of the wave cycles parked, profiles/r03_h_train_pmc_mfma.md; more waves per SIMD hide what a second
    // buffer hid, and the rest.)
    constexpr int kA = kWgARows * kWgKP;                           // one plane of the a-row copy
    constexpr int kZ = 64 * kWgKP;                                 // one plane of the dz ring
    unsigned short* const aBase = wg_lds;                          // [hi|lo][56 rows][kWgKP]
    unsigned short* const zH = wg_lds + 2 * kA;
    unsigned short* const zL = zH + kZ;
    const unsigned short* const zero = zL + kZ;
    const int clip = blockIdx.z;
    const int y0 = blockIdx.x * a.rows_per_wg;
    const int rows = a.H - y0 < a.rows_per_wg ? a.H - y0 : a.rows_per_wg;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, q = lane >> 4;
    for (int i = tid; i < (2 * kA + 2 * kZ + kWgKP) / 2; i += 256) reinterpret_cast<unsigned int*>(wg_lds)[i] = 0u;
    __syncthreads();
    const float* s0 = a.src0 + clip * a.src0_clip_stride;
    const float* s1 = a.src1 ? a.src1 + clip * a.src1_clip_stride : nullptr;
    const float* dzc = a.dz + static_cast<long long>(clip) * 8 * a.H * a.T;
    // staging map: thread -> (channel tid >> 5, frames tl, tl + 32, tl + 64): no divisions, 3 values per operand and row in registers
    const int ch = tid >> 5, tl = tid & 31;
    // Three rows of raw operands in flight (round 3): a row's loads are issued three rows before it is committed to the LDS.  With ONE row
    // ahead (round 2) every row waited out most of a memory latency behind ~900 cycles of multiply: 72 rows per workgroup x ~1.5 us was the
    // launch's time, three times its MFMA work.
    float avs[3][3], zvs[3][3];
    auto fetch = [&](int yl, float (&av)[3], float (&zv)[3]) {     // raw values of a-row y0 + yl and dz row l = yl + 3
        int zr = (y0 + yl + 3) % a.H;
        const int ya = y0 + yl;
        const bool a_ok = ch < a.cin && yl < rows;
        const float* ap = nullptr;
        if (a_ok) ap = ch < a.c0 ? s0 + (static_cast<long long>(ch) * a.H + ya) * a.T : s1 + (static_cast<long long>(ch - a.c0) * a.h1 + (ya % a.h1)) * a.T;
        const float* zp = dzc + (static_cast<long long>(ch) * a.H + zr) * a.T;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int t = tl + 32 * i;
            av[i] = (a_ok && t < a.T) ? ap[t] : 0.f;
            zv[i] = t < a.T ? zp[t] : 0.f;
        }
    };
    auto commit = [&](int yl, const float (&av)[3], const float (&zv)[3]) {   // registers -> LDS: the a-row copy, dz slot (yl + 3) & 7
        unsigned short* aH = aBase;
        unsigned short* aL = aH + kA;
        const int slot = (yl + 3 + 8) & 7;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int t = tl + 32 * i;
            if (t < a.T) {
                const unsigned int zb = bf16_bits(zv[i]);
                zH[(slot * 8 + ch) * kWgKP + t] = static_cast<unsigned short>(zb);
                zL[(slot * 8 + ch) * kWgKP + t] = static_cast<unsigned short>(bf16_bits(zv[i] - __uint_as_float(zb << 16)));
                if (ch < a.cin) {
                    const float v = affine_act(av[i], a.in_affine, ch);
                    const unsigned int hb = bf16_bits(v);
                    const unsigned short h = static_cast<unsigned short>(hb), l = static_cast<unsigned short>(bf16_bits(v - __uint_as_float(hb << 16)));
#pragma unroll
                    for (int dx = 0; dx < 7; ++dx) {
                        int k = t - dx + 3;                        // aS[dx][ci][k] = a[ci][ya][(k + dx - 3) mod T]
                        k += k < 0 ? a.T : 0;
                        k -= k >= a.T ? a.T : 0;
                        aH[(dx * 8 + ch) * kWgKP + k] = h;
                        aL[(dx * 8 + ch) * kWgKP + k] = l;
                    }
                }
            }
        }
    };
    typedef float f32x4w __attribute__((ext_vector_type(4)));
    typedef __bf16 bf16x8w __attribute__((ext_vector_type(8)));
    f32x4w acc[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) acc[ni] = f32x4w{0.f, 0.f, 0.f, 0.f};
    const int m = 16 * wave + r16;                                 // A row -> (dy = m >> 3, co = m & 7)
    const int dy = m >> 3, co = m & 7;
    const int ksteps = (a.T + 31) / 32;
    // prologue: dz rows l = -3 .. 2 (fetch(yl) brings row l = yl + 3), then row 0 of both operands; row 1 stays in registers
    for (int yl = -6; yl < 0; ++yl) {
        int zr = (y0 + yl + 3) % a.H;
        zr += zr < 0 ? a.H : 0;
        const float* zp = dzc + (static_cast<long long>(ch) * a.H + zr) * a.T;
        const int slot = (yl + 3 + 8) & 7;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int t = tl + 32 * i;
            if (t < a.T) {
                const float v = zp[t];
                const unsigned int zb = bf16_bits(v);
                zH[(slot * 8 + ch) * kWgKP + t] = static_cast<unsigned short>(zb);
                zL[(slot * 8 + ch) * kWgKP + t] = static_cast<unsigned short>(bf16_bits(v - __uint_as_float(zb << 16)));
            }
        }
    }
    fetch(0, avs[0], zvs[0]);
    commit(0, avs[0], zvs[0]);
    fetch(1, avs[1], zvs[1]);
    fetch(2, avs[2], zvs[2]);
    fetch(3, avs[0], zvs[0]);
    __syncthreads();
    auto multiply = [&](int yl) {
        const unsigned short* aH = aBase;
        const unsigned short* aL = aH + kA;
        const int slot = (yl - dy + 3 + 8) & 7;
        const unsigned short* zh = dy < 7 ? zH + (slot * 8 + co) * kWgKP : zero;
        const unsigned short* zl = dy < 7 ? zL + (slot * 8 + co) * kWgKP : zero;
        for (int s = 0; s < ksteps; ++s) {
            const int ko = 32 * s + 8 * q;
            const bf16x8w ah = __builtin_bit_cast(bf16x8w, *reinterpret_cast<const uint4*>(zh + ko));
            const bf16x8w al = __builtin_bit_cast(bf16x8w, *reinterpret_cast<const uint4*>(zl + ko));
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const int n = 16 * ni + r16;
                const bool ok = n < 56;                            // dx = n >> 3 < 7
                const bf16x8w bh = __builtin_bit_cast(bf16x8w, *reinterpret_cast<const uint4*>(ok ? aH + n * kWgKP + ko : zero));
                const bf16x8w bl = __builtin_bit_cast(bf16x8w, *reinterpret_cast<const uint4*>(ok ? aL + n * kWgKP + ko : zero));
                acc[ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc[ni], 0, 0, 0);
                acc[ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc[ni], 0, 0, 0);
                acc[ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc[ni], 0, 0, 0);
            }
        }
    };
    // row yl multiplies; then row yl + 1 (register set (yl + 1) % 3, loaded three rows ago) replaces it and takes the free ring slot; its
    // registers then take row yl + 4
#define AKE_WG_ROW(J_)                                                          \
        if (yl < rows) {                                                         \
            multiply(yl);                                                        \
            __syncthreads();                                                     \
            if (yl + 1 < rows) {                                                 \
                commit(yl + 1, avs[J_], zvs[J_]);                                \
                fetch(yl + 4, avs[J_], zvs[J_]);                                 \
            }                                                                    \
            __syncthreads();                                                     \
            ++yl;                                                                \
        }
    for (int yl = 0; yl < rows;) {
        AKE_WG_ROW(1) AKE_WG_ROW(2) AKE_WG_ROW(0)
    }
#undef AKE_WG_ROW
    // flush: D[row mm = 16 * wave + 4q + i][col n = 16 * ni + r16]
    gfx_t* const dWs = grad_slot(a.dW, a.slot_stride);
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        const int n = 16 * ni + r16;
        const int dx = n >> 3, ci = n & 7;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int mm = 16 * wave + 4 * q + i;
            const int dyy = mm >> 3, coo = mm & 7;
            if (dyy < 7 && dx < 7 && ci < a.cin) {
                const long long idx = ((static_cast<long long>(coo) * a.cin + ci) * 7 + dyy) * 7 + dx;
                if (a.partial) a.partial[(static_cast<long long>(clip) * gridDim.x + blockIdx.x) * a.partial_stride + idx] = acc[ni][i];
                else grad_add(dWs + idx, acc[ni][i]);
            }
        }
    }
}

// ---- weight gradient of the 12 x 7 pitch-class convolutions (rows circular, frames zero-padded) on f16 MFMA with split operands -------
//   dW[co][ci][dy][dx] = sum_{clip, y, t} dz[co][y][t] * act(x[ci][(y + dy) mod 12][t + dx - pad])
// The f32 kernel (conv_wgrad_kernel<1|2, 6>) keeps one input channel per wave on the 16x16x4 f32 MFMA: 0.27 ms per convolution and 256
// clips, 1.7 ms of a 8.3 ms training step for the pitch-class stack and the heads.  Here a workgroup owns (clip, 16 output channels, 16
// input channels); for every kernel tap (dy, dx) the 16 x 16 block D[co][ci] is a GEMM over the clip's positions, K = the frames of one
// row at a time (k-steps of 32 frames):
//   A[m = co][k = t] = dz[co][y][t]                      (aligned 16-byte reads from the dz planes)
//   B[k = t][n = ci] = a[ci][(y + dy) mod 12][t + dx - pad]
// Wave w < 7 owns the tap column dx = w and keeps the 12 accumulator tiles of its dy; the time shift of B is resolved in registers: two
// aligned 16-byte reads and, for odd shifts, four v_alignbit per plane (the shift is a compile-time constant of the wave's code path, so
// even shifts are plain register picks).  Both operands are staged once per workgroup: raw f32 -> (pending BatchNorm + LeakyReLU for
// the input) -> f16 hi / lo * 2^11 planes in LDS, frames padded with zeros on both sides; the gradient tile is first scaled by a power of
// two that brings its largest magnitude into f16's upper range (gradients are 1e-3 .. 1e-9).  Three MFMAs per tile and k-step: hi*hi into
// one accumulator, lo*hi + hi*lo into a second one folded in with 2^-11 (2^-22 of a product dropped: f32-equivalent; bf16 hi / lo planes,
// 16 mantissa bits, left 3e-4 on the cancelling sum of a 1-channel layer whose input carries a DC offset).  The block's partial sums go to the clip's slot of the partial buffer (wgrad_partial_reduce_kernel adds the clips in a fixed
// order) or, without one, to the gradient slots with one fixed-point atomic per weight.
constexpr int kWpSeg = 64;                  // output frames per workgroup (two k-steps): longer rows run as segments, blockIdx.x = (segment, co block)

struct WgradPcArgs {
    const float* x;           // [clip][x_ctot][12][T_in] raw, first cin channels
    long long x_clip_stride;
    const float* in_affine;   // [cin][3] or null
    const float* dz;          // [clip][dz_ctot][12][T_out], channels [dz_coff, dz_coff + cout)
    long long dz_clip_stride;
    int dz_coff, cin, cout, T_in, T_out, pad;
    int AP, ZP;               // plane pitches (elements)
    int n_co_blocks, n_seg;   // gridDim.x = n_seg * n_co_blocks
    float* partial;           // nullable, see above
    long long partial_stride;
    gfx_t* dW;                // [cout][cin][12][7], slot 0
    long long slot_stride;
};

typedef float f32x4p __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8p __attribute__((ext_vector_type(8)));

// elements [R, R + 8) of the 16 halves in (c0 | c1)
template <int R>
__device__ __forceinline__ f16x8p shifted8(const uint4& c0, const uint4& c1) {
    const unsigned int w[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
    constexpr int m = R >> 1;
    uint4 o;
    if (R & 1) {
        o.x = __builtin_amdgcn_alignbit(w[m + 1], w[m], 16);
        o.y = __builtin_amdgcn_alignbit(w[m + 2], w[m + 1], 16);
        o.z = __builtin_amdgcn_alignbit(w[m + 3], w[m + 2], 16);
        o.w = __builtin_amdgcn_alignbit(w[(m + 4) & 7], w[m + 3], 16);
    } else {
        o.x = w[m]; o.y = w[m + 1]; o.z = w[m + 2]; o.w = w[m + 3];
    }
    return __builtin_bit_cast(f16x8p, o);
}

template <int R>
__device__ __forceinline__ void wgrad_pc_multiply(const WgradPcArgs& a, const unsigned short* aH, const unsigned short* aL, const unsigned short* zH,
                                                  const unsigned short* zL, int base0, int ksteps, int r16, int q, f32x4p (&acc)[12], f32x4p (&accl)[12]) {
    // The 12 dz rows of a k-step stay in REGISTERS (96 of the 256 a wave of this one-workgroup-per-CU kernel may hold) and meet every
    // input row's shifted operand, which is built once per (input row, k-step): 24 + 12 x 4 LDS reads per 432 MFMAs.  (Reading the dz
    // fragment again for every (input row, dy) was 2 reads per 3 MFMAs: the LDS array busier than the matrix pipe.)
    for (int ks = 0; ks < ksteps; ++ks) {
        f16x8p zh[12], zl[12];
#pragma unroll
        for (int y = 0; y < 12; ++y) {
            const int zo = (y * 16 + r16) * a.ZP + 32 * ks + 8 * q;
            zh[y] = __builtin_bit_cast(f16x8p, *reinterpret_cast<const uint4*>(zH + zo));
            zl[y] = __builtin_bit_cast(f16x8p, *reinterpret_cast<const uint4*>(zL + zo));
        }
#pragma unroll
        for (int row = 0; row < 12; ++row) {
            const int o = (row * 16 + r16) * a.AP + base0 + 32 * ks + 8 * q;       // aligned start of the shifted window inside the plane row
            const uint4 h0 = *reinterpret_cast<const uint4*>(aH + o);
            const uint4 l0 = *reinterpret_cast<const uint4*>(aL + o);
            uint4 h1 = h0, l1 = l0;
            if (R != 0) { h1 = *reinterpret_cast<const uint4*>(aH + o + 8); l1 = *reinterpret_cast<const uint4*>(aL + o + 8); }
            const f16x8p bh = shifted8<R>(h0, h1), bl = shifted8<R>(l0, l1);
#pragma unroll
            for (int dy = 0; dy < 12; ++dy) acc[dy] = __builtin_amdgcn_mfma_f32_16x16x32_f16(zh[(row - dy + 12) % 12], bh, acc[dy], 0, 0, 0);
#pragma unroll
            for (int dy = 0; dy < 12; ++dy) accl[dy] = __builtin_amdgcn_mfma_f32_16x16x32_f16(zl[(row - dy + 12) % 12], bh, accl[dy], 0, 0, 0);
#pragma unroll
            for (int dy = 0; dy < 12; ++dy) accl[dy] = __builtin_amdgcn_mfma_f32_16x16x32_f16(zh[(row - dy + 12) % 12], bl, accl[dy], 0, 0, 0);
            if (row & 1) __builtin_amdgcn_sched_barrier(0);   // (fully unrolled, hipcc hoists every row's reads to the top: 256 registers + spills)
        }
    }
}

__global__ __launch_bounds__(512) void conv_wgrad_pc_f16x3_kernel(WgradPcArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned short wp_lds[];
    // planes: activated input [12 rows][16 ci][AP] (frame f at element f + 8, zeros around), dz [12 rows][16 co][ZP] (zeros behind T_out)
    const int nA = 12 * 16 * a.AP, nZ = 12 * 16 * a.ZP;
    unsigned short* const aH = wp_lds;
    unsigned short* const aL = aH + nA;
    unsigned short* const zH = aL + nA;
    unsigned short* const zL = zH + nZ;
    const int clip = blockIdx.z, ci0 = 16 * blockIdx.y;
    const int seg = blockIdx.x / a.n_co_blocks, co0 = 16 * (blockIdx.x - seg * a.n_co_blocks);
    const int t0 = kWpSeg * seg;                                   // first output frame of this workgroup
    const int T_seg = a.T_out - t0 < kWpSeg ? a.T_out - t0 : kWpSeg;
    const int ksteps = (T_seg + 31) / 32;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, q = lane >> 4;
    __shared__ float zmax_red[8];
    f16_saturate_mode();
    for (int i = tid; i < (nA + nZ); i += 512) reinterpret_cast<unsigned int*>(wp_lds)[i] = 0u;      // (2 (nA + nZ) elements = nA + nZ dwords)
    // the gradient tile's scale: a power of two that brings its largest magnitude to [2^13, 2^14) (gradients are tiny -- 1e-3 .. 1e-9 -- and
    // f16 hi + lo * 2^11 holds 22 bits below the largest value whatever the scale)
    const float* const zc = a.dz + clip * a.dz_clip_stride + static_cast<long long>(a.dz_coff + co0) * 12 * a.T_out;
    const int nzl = (a.cout - co0 < 16 ? a.cout - co0 : 16) * 12;
    // Staging, eight lines in flight per wave (round 3): with one (channel, row) line per loop iteration every global load waited out a full
    // memory round trip before its LDS stores -- 24 lines x 3 passes = 72 dependent round trips per workgroup, about half of this launch.
    // A line of dz has T_seg <= 64 frames (one value per lane), a line of the input plane AP <= 128 elements (two); the loads are branch-free
    // (clamped addresses), the stores guarded.
    constexpr int kU = 8;
    {
        float m = 0.f;
        for (int l0 = wave; l0 < nzl; l0 += 8 * kU) {
            float v[kU];
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                const int line = min(l0 + 8 * u, nzl - 1);
                v[u] = zc[static_cast<long long>(line) * a.T_out + t0 + (lane < T_seg ? lane : T_seg - 1)];      // line = (channel, row): rows of T_out frames follow each other
            }
#pragma unroll
            for (int u = 0; u < kU; ++u) m = fmaxf(m, fabsf(v[u]));                                               // (clamped duplicates do not change a maximum)
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if (lane == 0) zmax_red[wave] = m;
    }
    __syncthreads();
    float zs;
    {
        float m = zmax_red[0];
#pragma unroll
        for (int w = 1; w < 8; ++w) m = fmaxf(m, zmax_red[w]);
        zs = f16_weight_scale(m);
    }
    {   // stage the input: one (channel, row) line of T_in frames per 64 threads' pass
        const float* xc = a.x + clip * a.x_clip_stride;
        const int ncl = (a.cin - ci0 < 16 ? a.cin - ci0 : 16) * 12;
        for (int l0 = wave; l0 < ncl; l0 += 8 * kU) {
            float v[kU][2];
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                const int line = min(l0 + 8 * u, ncl - 1);
                const int c = line / 12, row = line - 12 * c;
                const float* xr = xc + (static_cast<long long>(ci0 + c) * 12 + row) * a.T_in;
#pragma unroll
                for (int h = 0; h < 2; ++h) {                          // plane element e0 <-> input frame t0 + e0 - 8
                    int t = t0 + lane + 64 * h - 8;
                    t = t < 0 ? 0 : (t >= a.T_in ? a.T_in - 1 : t);
                    v[u][h] = xr[t];
                }
            }
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                const int line = l0 + 8 * u;
                if (line >= ncl) break;
                const int c = line / 12, row = line - 12 * c;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int e0 = lane + 64 * h, t = t0 + e0 - 8;
                    if (e0 >= a.AP || t < 0 || t >= a.T_in) continue;
                    const float x = affine_act(v[u][h], a.in_affine, ci0 + c);
                    const _Float16 hv = static_cast<_Float16>(x);
                    const int e = (row * 16 + c) * a.AP + e0;
                    aH[e] = __builtin_bit_cast(unsigned short, hv);
                    aL[e] = static_cast<unsigned short>(f16_bits((x - static_cast<float>(hv)) * kP2pLoScale));
                }
            }
        }
        for (int l0 = wave; l0 < nzl; l0 += 8 * kU) {
            float v[kU];
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                const int line = min(l0 + 8 * u, nzl - 1);
                v[u] = zc[static_cast<long long>(line) * a.T_out + t0 + (lane < T_seg ? lane : T_seg - 1)];
            }
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                const int line = l0 + 8 * u;
                if (line >= nzl || lane >= T_seg) continue;
                const int c = line / 12, row = line - 12 * c;
                const float x = v[u] * zs;
                const _Float16 hv = static_cast<_Float16>(x);
                const int e = (row * 16 + c) * a.ZP + lane;
                zH[e] = __builtin_bit_cast(unsigned short, hv);
                zL[e] = static_cast<unsigned short>(f16_bits((x - static_cast<float>(hv)) * kP2pLoScale));
            }
        }
    }
    __syncthreads();
    if (wave >= 7) return;                                         // (no barrier below)
    const int dx = wave;
    f32x4p acc[12], accl[12];
#pragma unroll
    for (int dy = 0; dy < 12; ++dy) { acc[dy] = f32x4p{0.f, 0.f, 0.f, 0.f}; accl[dy] = f32x4p{0.f, 0.f, 0.f, 0.f}; }
    const int shift = 8 + dx - a.pad;                              // element of frame (t + dx - pad) at t = 0
    const int base0 = shift & ~7;
    switch (shift & 7) {                                           // wave-uniform: every wave runs one specialisation
        case 0: wgrad_pc_multiply<0>(a, aH, aL, zH, zL, base0, ksteps, r16, q, acc, accl); break;
        case 1: wgrad_pc_multiply<1>(a, aH, aL, zH, zL, base0, ksteps, r16, q, acc, accl); break;
        case 2: wgrad_pc_multiply<2>(a, aH, aL, zH, zL, base0, ksteps, r16, q, acc, accl); break;
        case 3: wgrad_pc_multiply<3>(a, aH, aL, zH, zL, base0, ksteps, r16, q, acc, accl); break;
        case 4: wgrad_pc_multiply<4>(a, aH, aL, zH, zL, base0, ksteps, r16, q, acc, accl); break;
        case 5: wgrad_pc_multiply<5>(a, aH, aL, zH, zL, base0, ksteps, r16, q, acc, accl); break;
        case 6: wgrad_pc_multiply<6>(a, aH, aL, zH, zL, base0, ksteps, r16, q, acc, accl); break;
        default: wgrad_pc_multiply<7>(a, aH, aL, zH, zL, base0, ksteps, r16, q, acc, accl); break;
    }
    // flush: D[row = co 4q + i][col = ci r16] of tap (dy, dx)
    gfx_t* const dWs = grad_slot(a.dW, a.slot_stride);
    const int ci = ci0 + r16;
    const float inv_zs = 1.f / zs;
#pragma unroll
    for (int dy = 0; dy < 12; ++dy)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int co = co0 + 4 * q + i;
            if (co < a.cout && ci < a.cin) {
                const long long idx = ((static_cast<long long>(co) * a.cin + ci) * 12 + dy) * 7 + dx;
                const float v = fmaf(accl[dy][i], kP2pLoInv, acc[dy][i]) * inv_zs;
                if (a.partial) a.partial[(static_cast<long long>(clip) * a.n_seg + seg) * a.partial_stride + idx] = v;
                else grad_add(dWs + idx, v);
            }
        }
}

}  // namespace ake_k

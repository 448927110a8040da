// Micro-benchmark (gfx950): do a wave's VALU instructions issue while the SIMD's matrix pipe executes ANOTHER wave's MFMAs?
// One workgroup of 512 threads per CU = two waves per SIMD (w and w + 4).  Modes:
//   a  waves 0-3 MFMA, waves 4-7 idle        b  waves 0-3 idle, waves 4-7 VALU        c  waves 0-3 MFMA, waves 4-7 VALU
//   d  all eight waves MFMA                   e  every wave MFMA and VALU in one instruction stream (1 MFMA : 4 FMA)
// overlap  <=>  t(c) ~ max(t(a), t(b));  no overlap  <=>  t(c) ~ t(a) + t(b).
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_valu_overlap.hip -o tools/micro/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

typedef float f32x16 __attribute__((ext_vector_type(16)));

// the same question for v_mfma_f32_32x32x16_f16 (twice the MACs per instruction: 8 passes instead of 4): 3 MFMAs + 24 FMAs per iteration =
// the same matrix work and the same vector work as the 16x16x32 body, so the two tables compare line by line
template <bool MF, bool VA>
__device__ __forceinline__ void body32(int iters, float seed, float* out) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = static_cast<_Float16>(seed + i); b[i] = static_cast<_Float16>(0.5f * seed - i); }
    f32x16 acc[3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 16; ++j) acc[i][j] = seed;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed + i;
    const float m = 1.0001f, c = 0.001f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            if (MF) acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[u], 0, 0, 0);
            if (VA) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = fmaf(v[k], m, c);
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 16; ++j) s += acc[i][j];
    for (int i = 0; i < 8; ++i) s += v[i];
    if (s == 12345.678f) *out = s;
}

template <bool MF, bool VA>
__device__ __forceinline__ void body(int iters, float seed, float* out) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = static_cast<_Float16>(seed + i); b[i] = static_cast<_Float16>(0.5f * seed - i); }
    f32x4 acc[6];
    for (int i = 0; i < 6; ++i) acc[i] = f32x4{seed, seed, seed, seed};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed + i;
    const float m = 1.0001f, c = 0.001f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            if (MF) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[u], 0, 0, 0);
            if (VA) {
#pragma unroll
                for (int k = 0; k < 4; ++k) v[(4 * u + k) & 7] = fmaf(v[(4 * u + k) & 7], m, c);
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 6; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += v[i];
    if (s == 12345.678f) *out = s;
}

__global__ __launch_bounds__(512) void k(int mode, int iters, float seed, float* out) {
    const int wave = threadIdx.x >> 6;
    const bool lo = wave < 4;
    switch (mode) {
    case 0: if (lo) body<true, false>(iters, seed, out); break;
    case 1: if (!lo) body<false, true>(iters, seed, out); break;
    case 2: if (lo) body<true, false>(iters, seed, out); else body<false, true>(iters, seed, out); break;
    case 3: body<true, false>(iters, seed, out); break;
    case 4: body<true, true>(iters, seed, out); break;
    case 5: body<false, true>(iters, seed, out); break;
    }
}

__global__ __launch_bounds__(512) void k32(int mode, int iters, float seed, float* out) {
    const int wave = threadIdx.x >> 6;
    const bool lo = wave < 4;
    switch (mode) {
    case 0: if (lo) body32<true, false>(iters, seed, out); break;
    case 1: if (!lo) body32<false, true>(iters, seed, out); break;
    case 2: if (lo) body32<true, false>(iters, seed, out); else body32<false, true>(iters, seed, out); break;
    case 3: body32<true, false>(iters, seed, out); break;
    case 4: body32<true, true>(iters, seed, out); break;
    case 5: body32<false, true>(iters, seed, out); break;
    }
}

int main() {
    float* out;
    hipMalloc(&out, 4);
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, iters = 20000;
    const double mhz = p.clockRate / 1000.0;
    const char* names[] = {"a: waves 0-3 MFMA, 4-7 idle", "b: waves 0-3 idle, 4-7 VALU", "c: waves 0-3 MFMA, 4-7 VALU", "d: all waves MFMA",
                           "e: all waves MFMA + VALU interleaved", "f: all waves VALU"};
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%d CUs, %.0f MHz; per wave and iteration: 6 MFMA 16x16x32 f16 and / or 24 v_fma_f32\n", cus, mhz);
    for (int mode = 0; mode < 6; ++mode) {
        hipLaunchKernelGGL(k, dim3(cus), dim3(512), 0, 0, mode, 100, 1.0f, out);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(cus), dim3(512), 0, 0, mode, iters, 1.0f, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-40s %8.3f ms  = %6.1f cycles per iteration (at the nominal clock)\n", names[mode], ms, ms * 1e-3 * mhz * 1e6 / iters);
    }
    printf("\nthe same with v_mfma_f32_32x32x16_f16: per wave and iteration 3 MFMA 32x32x16 f16 (= the MACs of 6 16x16x32) and / or 24 v_fma_f32\n");
    for (int mode = 0; mode < 6; ++mode) {
        hipLaunchKernelGGL(k32, dim3(cus), dim3(512), 0, 0, mode, 100, 1.0f, out);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k32, dim3(cus), dim3(512), 0, 0, mode, iters, 1.0f, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-40s %8.3f ms  = %6.1f cycles per iteration (at the nominal clock)\n", names[mode], ms, ms * 1e-3 * mhz * 1e6 / iters);
    }
    return 0;
}

// Micro-probe: what stops v_mfma_f32_16x16x4_f32 from issuing every 32 cycles per SIMD?
//   mode 0: MFMAs only, operands in registers
//   mode 1: + A via ds_read2_b32 each step (3 tiles), used in the same step
//   mode 2: + B via ds_read2st64 each step
//   mode 3: A/B prefetched one step ahead (software pipelined by hand with asm barriers)
// hipcc --offload-arch=gfx950 -O3 mfma_probe.hip -o mfma_probe && ./mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int MT>
__global__ __launch_bounds__(512) void probe(float* out, int steps, int lds_floats) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < lds_floats; i += blockDim.x) lds[i] = 0.001f * (i & 255);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f32x4 acc[MT], acc2[MT];
    for (int m = 0; m < MT; ++m) { acc[m] = f32x4{0, 0, 0, 0}; acc2[m] = f32x4{0, 0, 0, 0}; }
    float a0[MT], a1[MT], b0 = 1.f + lane, b1 = 2.f + lane;
    for (int m = 0; m < MT; ++m) { a0[m] = lane + m; a1[m] = lane - m; }
    int off = lane;
    for (int st = 0; st < steps; ++st) {
        if (MODE >= 1) {
#pragma unroll
            for (int m = 0; m < MT; ++m) { a0[m] = lds[off + m * 97]; a1[m] = lds[off + m * 97 + 4]; }
        }
        if (MODE >= 2) { b0 = lds[off + 1024]; b1 = lds[off + 1088]; }
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[m], b0, acc[m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < MT; ++m) acc2[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[m], b1, acc2[m], 0, 0, 0);
        off += 84;
        if (off > lds_floats - 2048) off = lane;
    }
    f32x4 r = f32x4{0, 0, 0, 0};
    for (int m = 0; m < MT; ++m) r += acc[m] + acc2[m];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r[0] + r[1] + r[2] + r[3];
}

template <int MODE, int MT>
void run(const char* name, int blocks, int threads, int lds_bytes, int steps) {
    float* out;
    hipMalloc(&out, sizeof(float) * blocks * threads);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe<MODE, MT>), dim3(blocks), dim3(threads), lds_bytes, 0, out, steps, lds_bytes / 4);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfma = double(blocks) * (threads / 64) * steps * 2 * MT;
    const double cyc_per_mfma_per_simd = ms * 1e-3 * 2.4e9 / (mfma / 1024.0);
    printf("%-34s blocks=%5d thr=%3d lds=%6d steps=%d  %.3f ms  -> %.1f cycles/MFMA/SIMD @2.4GHz (32 = peak), %.1f TFLOP/s\n", name, blocks,
           threads, lds_bytes, steps, ms, cyc_per_mfma_per_simd, mfma * 2048 / (ms * 1e-3) / 1e12);
    hipFree(out);
}

int main() {
    const int steps = 2000;
    run<0, 3>("mode0 regs only MT3 1blk/CU 4w", 256, 256, 16384, steps);
    run<0, 3>("mode0 regs only MT3 8w/CU", 256, 512, 16384, steps);
    run<0, 3>("mode0 regs only MT3 32w/CU", 1024, 512, 36000, steps);
    run<1, 3>("mode1 +A lds MT3 8w/CU", 256, 512, 36000, steps);
    run<1, 3>("mode1 +A lds MT3 32w/CU", 1024, 512, 36000, steps);
    run<2, 3>("mode2 +A+B lds MT3 8w/CU", 256, 512, 36000, steps);
    run<2, 3>("mode2 +A+B lds MT3 16w/CU", 512, 512, 36000, steps);
    run<2, 3>("mode2 +A+B lds MT3 32w/CU", 1024, 512, 36000, steps);
    run<2, 6>("mode2 +A+B lds MT6 16w/CU", 512, 512, 36000, steps);
    run<2, 6>("mode2 +A+B lds MT6 32w/CU", 1024, 512, 36000, steps);
    run<2, 3>("mode2 MT3 32w/CU 1.8 rounds", 1856, 512, 36000, 56);
    return 0;
}

// Micro-benchmark (gfx950): does the MFMA rate depend on the OPERAND DATA?  The same instruction stream (every wave: 8 independent accumulators,
// v_mfma_f32_16x16x32_f16 back to back, 4 waves per SIMD) with (a) all-zero operands, (b) one constant value, (c) random f16 values in [-2, 2).
// If (c) is slower than (a), a kernel's ceiling under live data is not the data-sheet rate, and ablations that zero a kernel's inputs overstate
// what the removed phase cost.   build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_data_power.hip -o tools/micro/mfma_data_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <time.h>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k(const uint4* __restrict__ ops, int iters, float* out) {
    const int lane = threadIdx.x & 63;
    f16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = __builtin_bit_cast(f16x8, ops[(i * 2 + 0) * 64 + lane]);
        b[i] = __builtin_bit_cast(f16x8, ops[(i * 2 + 1) * 64 + lane]);
    }
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[u & 3], b[(u + 1) & 3], acc[u], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) *out = s;
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    uint4* ops; float* out;
    hipMalloc(&ops, 8 * 64 * sizeof(uint4)); hipMalloc(&out, 4);
    const int iters = 400000;          // 8 MFMAs x 16 cycles x 4 waves per SIMD = 512 cycles per iteration and SIMD: ~85 ms at 2.4 GHz
    const char* names[] = {"all-zero operands", "one constant (1.0)", "random f16 in [-2, 2)"};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%d CUs; 4 workgroups of 4 waves per CU (4 waves per SIMD), %d iterations x 8 MFMA 16x16x32 f16 per wave\n", cus, iters);
    for (int rep = 0; rep < 2; ++rep)
    for (int mode = 0; mode < 3; ++mode) {
        std::vector<unsigned short> h(8 * 64 * 8);
        srand(1);
        for (auto& v : h) {
            _Float16 x = mode == 0 ? (_Float16)0.f : mode == 1 ? (_Float16)1.f : (_Float16)(4.f * rand() / RAND_MAX - 2.f);
            v = __builtin_bit_cast(unsigned short, x);
        }
        hipMemcpy(ops, h.data(), h.size() * 2, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(cus * 4), dim3(256), 0, 0, ops, 1000, out);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(cus * 4), dim3(256), 0, 0, ops, iters, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
        const double flops = 2.0 * 16 * 16 * 32 * 8.0 * iters * 4 * 4 * cus;
        printf("%-24s %8.2f ms  %7.1f TFLOP/s  = an MFMA clock of %.2f GHz\n", names[mode], ms, flops / ms * 1e-9, 512.0 * iters / (ms * 1e-3) * 1e-9);
    }
    // the same stream with random operands at several launch lengths: where does the rate settle?  (each after 50 ms of idling)
    printf("\nrandom operands, launch length sweep (one launch each, 50 ms idle before):\n");
    for (int it2 : {500, 2000, 8000, 32000, 128000, 400000}) {
        hipDeviceSynchronize();
        struct timespec ts = {0, 50000000}; nanosleep(&ts, nullptr);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(cus * 4), dim3(256), 0, 0, ops, it2, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
        const double flops = 2.0 * 16 * 16 * 32 * 8.0 * it2 * 4 * 4 * cus;
        printf("%7d iterations  %8.3f ms  %7.1f TFLOP/s  = an MFMA clock of %.2f GHz\n", it2, ms, flops / ms * 1e-9, 512.0 * it2 / (ms * 1e-3) * 1e-9);
    }
    return 0;
}

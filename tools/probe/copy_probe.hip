// Calibration for the rocprofv3 FETCH_SIZE / WRITE_SIZE counters on gfx950 (MI355X_MICROARCH.md, HBM section):
// stream a buffer far larger than the 256 MiB Infinity Cache once, with the access widths the library kernels use.
//   read4 : 4 B per lane coalesced loads  (conv patch staging)          read16: 16 B per lane (decimator, W fragments)
//   write4: 4 B per lane stores (conv epilogue)                         write16: 16 B per lane (decimator)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void read4(const float* p, float* sink, size_t n) {
    float acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 12345.678f) *sink = acc;
}
__global__ void read16(const float4* p, float* sink, size_t n4) {
    float acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) { float4 v = p[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 12345.678f) *sink = acc;
}
__global__ void write4(float* p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 1.f;
}
__global__ void write16(float4* p, size_t n4) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) p[i] = make_float4(1, 2, 3, 4);
}
int main() {
    const size_t bytes = 1ull << 30;   // 1 GiB
    float *a, *sink;
    hipMalloc(&a, bytes); hipMalloc(&sink, 4);
    hipMemset(a, 0, bytes);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(read4, dim3(2048), dim3(256), 0, 0, a, sink, bytes / 4);
    hipLaunchKernelGGL(read16, dim3(2048), dim3(256), 0, 0, (const float4*)a, sink, bytes / 16);
    hipLaunchKernelGGL(write4, dim3(2048), dim3(256), 0, 0, a, bytes / 4);
    hipLaunchKernelGGL(write16, dim3(2048), dim3(256), 0, 0, (float4*)a, bytes / 16);
    hipDeviceSynchronize();
    printf("each kernel moves %zu bytes\n", bytes);
    return 0;
}

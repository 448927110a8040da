// Fused Adam over the flat parameter buffer (reference optimizer: models.py:1017-1027, torch.optim.Adam with
// betas (0.9, 0.999), lr --lr, weight_decay --reg, followed by ExponentialLR(gamma) once per epoch).
// One launch per optimizer step instead of torch's per-tensor (or foreach) passes over ~100 small tensors; HBM-bound:
// 4 reads + 3 writes of 4 B per parameter (668 KB of parameters -> the launch latency dominates).
#include "common.h"

#include <cmath>

namespace {

// Operation order follows torch/optim/adam.py::_single_tensor_adam so that the result agrees with the reference
// optimizer to the last bit or two (lerp for exp_avg; mul + addcmul for exp_avg_sq; sqrt / bias_correction2_sqrt + eps).
__global__ __launch_bounds__(256) void adam_step_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, const unsigned char* __restrict__ trainable, size_t n,
                                                        float step_size, float beta1, float beta2, float eps, float weight_decay,
                                                        float bc2_sqrt, float grad_scale) {
    const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (trainable && !trainable[i]) return;
    const float pi = p[i];
    float gi = g[i] * grad_scale;
    if (weight_decay != 0.f) gi = fmaf(weight_decay, pi, gi);
    const float mi = m[i] + (1.f - beta1) * (gi - m[i]);          // torch.lerp (weight < 0.5 branch)
    const float vi = fmaf(1.f - beta2, gi * gi, v[i] * beta2);
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = pi - step_size * (mi / denom);
}

}  // namespace

extern "C" int ake_adam_step_f32(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, const unsigned char* trainable,
                                 size_t count, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                                 float grad_scale, ake_stream_t stream) {
    AKE_REQUIRE(params && grads && exp_avg && exp_avg_sq, AKE_ERR_INVALID, "adam: null argument");
    AKE_REQUIRE(step >= 1, AKE_ERR_INVALID, "adam: step counts from 1 (got %d)", step);
    if (count == 0) return AKE_OK;
    const double bc1 = 1.0 - std::pow(static_cast<double>(beta1), step);
    const double bc2 = 1.0 - std::pow(static_cast<double>(beta2), step);
    hipStream_t s = static_cast<hipStream_t>(stream);
    ake::ProfScope ps("adam_step_kernel", s);
    hipLaunchKernelGGL(adam_step_kernel, dim3(static_cast<unsigned>((count + 255) / 256)), dim3(256), 0, s, params, grads, exp_avg, exp_avg_sq,
                       trainable, count, static_cast<float>(lr / bc1), beta1, beta2, eps, weight_decay, static_cast<float>(std::sqrt(bc2)),
                       grad_scale);
    AKE_HIP_CHECK(hipGetLastError());
    return AKE_OK;
}

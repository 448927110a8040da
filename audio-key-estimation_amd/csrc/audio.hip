// Device-side audio preparation in front of the CQT: channel selection / mono mix-down and
// polyphase resampling, for ragged batches.
//
// The reference does neither: it takes channel 0 of whatever torchaudio.load returns, at the file's own sample rate, and sets
// hop = round(rate / frames) (KeyDataset.py:479-485).  With channel = 0 and rate_in == rate_out this stage is the identity on
// channel 0, i.e. the reference's behaviour; everything else is opt-in, for serving pipelines that hold decoded multi-channel
// audio of mixed rates on the GPU and want ONE CQT plan (one hop, one set of filter tables) for all of it.
//
// Resampling = scipy.signal.resample_poly(x, up, down) with its default filter (the checker, oracle/resample_oracle.py, is pinned
// on scipy itself):  up, down = rate_out, rate_in reduced by their gcd;  h = firwin(2 * half + 1, 1 / max(up, down),
// window = ("kaiser", 5.0)) * up  with  half = 10 * max(up, down);  y[k] = sum_i x[i] h[k * down - i * up + half],
// k < ceil(n * up / down).  One thread per output sample: ~20 * max(up, down) / up + 1 taps (41 for 44.1 -> 22.05 kHz, 44 for
// 48 -> 22.05 kHz), input and filter served from L1/L2 -- a streaming kernel, bound by HBM (4 B read per input sample and channel,
// 4 B written per output sample).
#include "common.h"

#include <cmath>
#include <numeric>

struct ake_resampler {
    int rate_in, rate_out, up, down, half;
    float* h_dev = nullptr;     // [2 * half + 1]
};

namespace {

double bessel_i0d(double x) {
    double sum = 1.0, term = 1.0;
    const double q = x * x / 4.0;
    for (int k = 1; k < 500; ++k) {
        term *= q / (static_cast<double>(k) * k);
        sum += term;
        if (term < 1e-18 * sum) break;
    }
    return sum;
}

struct ResampleArgs {
    const float* in;            // [batch][channels][n_in] by strides
    long long clip_stride, channel_stride;
    int channels, channel;      // channel >= 0: that channel; -1: mean over the channels
    long long n_in;
    const long long* n_in_clip; // ragged: samples of each clip (<= n_in), or null
    float* out;                 // [batch][out_stride]
    long long out_stride, n_out_max;
    long long* n_out_clip;      // per-clip output length written here (or null)
    const float* h;
    int up, down, half;
};

__global__ __launch_bounds__(256) void resample_kernel(ResampleArgs a) {
    const int clip = blockIdx.y;
    long long n = a.n_in;
    if (a.n_in_clip) { const long long nc = a.n_in_clip[clip]; n = nc < 0 ? 0 : (nc < n ? nc : n); }
    const long long n_out = (n * a.up + a.down - 1) / a.down;
    if (blockIdx.x == 0 && threadIdx.x == 0 && a.n_out_clip) a.n_out_clip[clip] = n_out;
    const long long k = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (k >= a.n_out_max) return;
    float* o = a.out + clip * a.out_stride + k;
    if (k >= n_out) { *o = 0.f; return; }                         // zero padding behind a shorter clip
    const float* x = a.in + clip * a.clip_stride;
    const int nch = a.channel >= 0 ? 1 : a.channels;
    const long long c0 = a.channel >= 0 ? a.channel : 0;
    const float scale = a.channel >= 0 ? 1.f : 1.f / a.channels;
    float acc = 0.f;
    if (a.up == a.down) {                                         // same rate: channel selection / mix only
        for (int c = 0; c < nch; ++c) acc += x[(c0 + c) * a.channel_stride + k];
        *o = acc * scale;
        return;
    }
    const long long t = k * a.down;
    long long i_lo = t - a.half <= 0 ? 0 : (t - a.half + a.up - 1) / a.up;
    long long i_hi = (t + a.half) / a.up;
    i_hi = i_hi < n - 1 ? i_hi : n - 1;
    int j = static_cast<int>(t - i_lo * a.up + a.half);           // tap of input sample i_lo; the next sample's is up less
    for (long long i = i_lo; i <= i_hi; ++i, j -= a.up) {
        float v = 0.f;
        for (int c = 0; c < nch; ++c) v += x[(c0 + c) * a.channel_stride + i];
        acc = fmaf(v, a.h[j], acc);
    }
    *o = acc * scale;
}

}  // namespace

extern "C" {

int ake_resampler_create(int rate_in, int rate_out, ake_resampler** out) {
    AKE_REQUIRE(out && rate_in > 0 && rate_out > 0, AKE_ERR_INVALID, "ake_resampler_create: bad argument");
    auto* r = new ake_resampler();
    r->rate_in = rate_in; r->rate_out = rate_out;
    const int g = std::gcd(rate_in, rate_out);
    r->up = rate_out / g; r->down = rate_in / g;
    const int mr = std::max(r->up, r->down);
    r->half = 10 * mr;
    AKE_REQUIRE(r->half <= (1 << 22), AKE_ERR_UNSUPPORTED, "resampler: %d -> %d Hz needs a %d-tap filter", rate_in, rate_out, 2 * r->half + 1);
    // scipy.signal.firwin(2 * half + 1, 1 / mr, window = ("kaiser", 5.0)), scaled by `up` (resample_poly)
    const int N = 2 * r->half + 1;
    std::vector<double> h(N);
    const double fc = 1.0 / mr, beta = 5.0, i0b = bessel_i0d(beta);
    double sum = 0.0;
    for (int i = 0; i < N; ++i) {
        const double m = i - r->half;
        const double sinc = m == 0 ? 1.0 : std::sin(M_PI * fc * m) / (M_PI * fc * m);
        const double rr = 2.0 * i / (N - 1) - 1.0;
        const double win = bessel_i0d(beta * std::sqrt(std::max(0.0, 1.0 - rr * rr))) / i0b;
        h[i] = fc * sinc * win;
        sum += h[i];
    }
    std::vector<float> hf(N);
    for (int i = 0; i < N; ++i) hf[i] = static_cast<float>(h[i] / sum * r->up);
    hipError_t e = hipMalloc(&r->h_dev, N * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(r->h_dev, hf.data(), N * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        ake::set_error("resampler: filter upload failed: %s", hipGetErrorString(e));
        ake_resampler_destroy(r);
        return AKE_ERR_HIP;
    }
    *out = r;
    return AKE_OK;
}

void ake_resampler_destroy(ake_resampler* r) {
    if (!r) return;
    if (r->h_dev) (void)hipFree(r->h_dev);
    delete r;
}

int64_t ake_resampler_out_len(const ake_resampler* r, int64_t n_in) {
    if (!r || n_in < 0) return -1;
    return (n_in * r->up + r->down - 1) / r->down;
}

int ake_resample_f32(const ake_resampler* r, const float* in_dev, int batch, int channels, int64_t n_in, int64_t clip_stride,
                     int64_t channel_stride, int channel, const int64_t* n_in_clip_dev, float* out_dev, int64_t out_stride,
                     int64_t* n_out_clip_dev, ake_stream_t stream) {
    AKE_REQUIRE(r && in_dev && out_dev, AKE_ERR_INVALID, "ake_resample_f32: null argument");
    AKE_REQUIRE(batch > 0 && channels > 0 && n_in > 0 && channel >= -1 && channel < channels, AKE_ERR_INVALID, "resample: bad batch / channels / channel");
    const int64_t n_out = ake_resampler_out_len(r, n_in);
    AKE_REQUIRE(out_stride >= n_out, AKE_ERR_INVALID, "resample: out_stride %lld < %lld output samples", static_cast<long long>(out_stride), static_cast<long long>(n_out));
    ResampleArgs a;
    a.in = in_dev; a.clip_stride = clip_stride; a.channel_stride = channel_stride; a.channels = channels; a.channel = channel;
    a.n_in = n_in; a.n_in_clip = reinterpret_cast<const long long*>(n_in_clip_dev);
    a.out = out_dev; a.out_stride = out_stride; a.n_out_max = n_out; a.n_out_clip = reinterpret_cast<long long*>(n_out_clip_dev);
    a.h = r->h_dev; a.up = r->up; a.down = r->down; a.half = r->half;
    hipStream_t s = static_cast<hipStream_t>(stream);
    ake::ProfScope ps("resample_kernel", s);
    hipLaunchKernelGGL(resample_kernel, dim3(static_cast<unsigned>((n_out + 255) / 256), batch), dim3(256), 0, s, a);
    AKE_HIP_CHECK(hipGetLastError());
    return AKE_OK;
}

}  // extern "C"

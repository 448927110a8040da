// Whole hot path for a batch of equal-length clips: CQT -> PitchClassNet forward.
// (DatasetLoader.get_all -> KeyDataset.__getitem__ -> general_step's forward:
//  KeyDataset.py:469-509, 242-256, models.py:846.)
#include "common.h"

namespace {

__global__ void fill_i64_kernel(long long* dst, long long v, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = v;
}

__global__ void frames_i64_kernel(long long* dst, const long long* __restrict__ n_clip, int hop, long long t_max, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const long long nc = n_clip[i];
        const long long t = nc < 0 ? 0 : 1 + nc / hop;                 // librosa center=True framing, as ake_cqt_num_frames
        dst[i] = t < t_max ? t : t_max;
    }
}

struct PipeCarve {
    float* mel;
    long long* seq;
    void* cqt_ws;
    size_t cqt_bytes;
    void* net_ws;
    size_t net_bytes;
    size_t total;
};

int carve(const ake_cqt_plan* plan, const ake_pcnet* net, int batch, int64_t n, int n_bins, void* ws, PipeCarve* pc) {
    const int64_t T = ake_cqt_num_frames(plan, n);
    AKE_REQUIRE(T > 0, AKE_ERR_INVALID, "pipeline: bad n_samples");
    ake::Carver c(ws, 0);
    pc->mel = c.take<float>(static_cast<size_t>(batch) * n_bins * T);
    pc->seq = c.take<long long>(batch);
    pc->cqt_bytes = ake_cqt_workspace_bytes(plan, batch, n);
    pc->cqt_ws = c.take<char>(pc->cqt_bytes);
    pc->net_bytes = ake_pcnet_workspace_bytes(net, batch, static_cast<int>(T));
    pc->net_ws = c.take<char>(pc->net_bytes);
    pc->total = ake::align_up(c.off, 256);
    return AKE_OK;
}

}  // namespace

extern "C" {

size_t ake_pipeline_workspace_bytes(const ake_cqt_plan* plan, const ake_pcnet* net, int batch, int64_t n_samples) {
    if (!plan || !net || batch <= 0 || n_samples <= 0) return 0;
    PipeCarve pc;
    if (carve(plan, net, batch, n_samples, ake_cqt_plan_n_bins(plan), nullptr, &pc) != AKE_OK) return 0;
    return pc.total;
}

static int pipeline_impl(const ake_cqt_plan* plan, const ake_pcnet* net, const float* audio_dev, int batch, int64_t n_samples,
                         int64_t audio_stride, const int64_t* n_clip_dev, int hop, float* key_out_dev, float* tonic_out_dev,
                         float* genre_out_dev, void* workspace, size_t workspace_bytes, ake_stream_t stream);

int ake_pipeline_forward_f32(const ake_cqt_plan* plan, const ake_pcnet* net, const float* audio_dev, int batch,
                             int64_t n_samples, int64_t audio_stride, float* key_out_dev, float* tonic_out_dev,
                             float* genre_out_dev, void* workspace, size_t workspace_bytes, ake_stream_t stream) {
    return pipeline_impl(plan, net, audio_dev, batch, n_samples, audio_stride, nullptr, 0, key_out_dev, tonic_out_dev, genre_out_dev, workspace,
                         workspace_bytes, stream);
}

int ake_pipeline_forward_ragged_f32(const ake_cqt_plan* plan, const ake_pcnet* net, const float* audio_dev, int batch,
                                    int64_t n_max, int64_t audio_stride, const int64_t* n_samples_dev, float* key_out_dev,
                                    float* tonic_out_dev, float* genre_out_dev, void* workspace, size_t workspace_bytes,
                                    ake_stream_t stream) {
    AKE_REQUIRE(n_samples_dev, AKE_ERR_INVALID, "ake_pipeline_forward_ragged_f32: null n_samples_dev");
    AKE_REQUIRE(plan && n_max > 0, AKE_ERR_INVALID, "ake_pipeline_forward_ragged_f32: bad argument");
    return pipeline_impl(plan, net, audio_dev, batch, n_max, audio_stride, n_samples_dev, ake_cqt_plan_hop(plan), key_out_dev, tonic_out_dev,
                         genre_out_dev, workspace, workspace_bytes, stream);
}

static int pipeline_impl(const ake_cqt_plan* plan, const ake_pcnet* net, const float* audio_dev, int batch, int64_t n_samples,
                         int64_t audio_stride, const int64_t* n_clip_dev, int hop, float* key_out_dev, float* tonic_out_dev,
                         float* genre_out_dev, void* workspace, size_t workspace_bytes, ake_stream_t stream) {
    AKE_REQUIRE(plan && net && audio_dev, AKE_ERR_INVALID, "ake_pipeline_forward_f32: null argument");
    const int n_bins = ake_cqt_plan_n_bins(plan);
    AKE_REQUIRE(n_bins == ake_pcnet_pitches(net), AKE_ERR_INVALID, "pipeline: CQT has %d bins but the net expects %d pitches",
                n_bins, ake_pcnet_pitches(net));
    PipeCarve pc;
    int rc = carve(plan, net, batch, n_samples, n_bins, workspace, &pc);
    if (rc) return rc;
    AKE_REQUIRE(workspace && workspace_bytes >= pc.total, AKE_ERR_WORKSPACE, "pipeline: workspace %zu < %zu bytes", workspace_bytes, pc.total);
    const int64_t T = ake_cqt_num_frames(plan, n_samples);
    hipStream_t s = static_cast<hipStream_t>(stream);
    // equal-length clips through the default net: the CQT stays in the filter bank's own [clip][frame][bin] order and the net's two
    // readers transpose while they stage it -- no transpose pass, same results bit for bit (AKE_PIPE_FRAMES_MAJOR=0 switches it off)
    static const bool fm_off = ake::diag_env("AKE_PIPE_FRAMES_MAJOR") != nullptr && std::atoi(ake::diag_env("AKE_PIPE_FRAMES_MAJOR")) == 0;
    const bool fm = !fm_off && !n_clip_dev && ake_cqt_frames_major_supported(plan) && ake_pcnet_accepts_frames_major(net, batch, static_cast<int>(T));
    rc = n_clip_dev ? ake_cqt_logmag_ragged_f32(plan, audio_dev, batch, n_samples, audio_stride, n_clip_dev, pc.mel, T, pc.cqt_ws, pc.cqt_bytes, stream)
         : fm       ? ake_cqt_logmag_frames_major_f32(plan, audio_dev, batch, n_samples, audio_stride, pc.mel, pc.cqt_ws, pc.cqt_bytes, stream)
                    : ake_cqt_logmag_f32(plan, audio_dev, batch, n_samples, audio_stride, pc.mel, T, pc.cqt_ws, pc.cqt_bytes, stream);
    if (rc) return rc;
    if (n_clip_dev)   // seq_length of every clip = its own frame count (KeyDataset.py:248: mel.shape[2] before padding)
        hipLaunchKernelGGL(frames_i64_kernel, dim3((batch + 255) / 256), dim3(256), 0, s, pc.seq, reinterpret_cast<const long long*>(n_clip_dev), hop,
                           static_cast<long long>(T), batch);
    else
        hipLaunchKernelGGL(fill_i64_kernel, dim3((batch + 255) / 256), dim3(256), 0, s, pc.seq, static_cast<long long>(T), batch);
    if (fm)
        return ake_pcnet_forward_frames_major_f32(net, pc.mel, batch, static_cast<int>(T), reinterpret_cast<const int64_t*>(pc.seq), key_out_dev,
                                                  tonic_out_dev, genre_out_dev, pc.net_ws, pc.net_bytes, stream);
    return ake_pcnet_forward_f32(net, pc.mel, batch, static_cast<int>(T), reinterpret_cast<const int64_t*>(pc.seq), key_out_dev,
                                 tonic_out_dev, genre_out_dev, pc.net_ws, pc.net_bytes, stream);
}

}  // extern "C"

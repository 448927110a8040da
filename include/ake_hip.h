/*
 * ake_hip.h -- C ABI of libake_hip.so, the MI355X (gfx950) implementation of the
 * key-estimation hot path:  waveform -> CQT log-magnitude -> PitchClassNet forward.
 *
 * The reference (flo-stilz/Audio-Key-Estimation) is pure Python and defines no FFI;
 * each entry point below names the reference call it stands in for (file:line into
 * the reference tree).  INTEGRATION.md shows the ctypes binding a maintainer adds.
 *
 * Conventions
 *   - every function returns AKE_OK (0) or a negative error code; no C++ exception
 *     crosses the ABI; ake_last_error() returns a thread-local message;
 *   - "dev" pointers are device (HIP) pointers, "host" pointers are host memory;
 *   - compute entry points never allocate and never synchronise: the caller passes
 *     a workspace (size from the matching *_workspace_bytes) and a hipStream_t;
 *     they are safe to capture into a hipGraph;
 *   - one handle may be used from one stream at a time (the workspace is the
 *     per-call state; handles are immutable after creation / finalize);
 *   - all tensors are dense row-major float32, NCHW as in the reference.
 */
#ifndef AKE_HIP_H
#define AKE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AKE_OK 0
#define AKE_ERR_INVALID (-1)     /* bad argument / shape */
#define AKE_ERR_HIP (-2)         /* a HIP runtime call failed */
#define AKE_ERR_STATE (-3)       /* handle not finalized, tensor missing, ... */
#define AKE_ERR_WORKSPACE (-4)   /* workspace too small */
#define AKE_ERR_UNSUPPORTED (-5) /* architecture flag outside the default family */

typedef void* ake_stream_t; /* hipStream_t */

int ake_version(void);
const char* ake_last_error(void);
/* 1 if the library was built with -DAKE_DIAG (`AKE_DIAG=1 csrc/build.sh`): only then are the AKE_* environment switches of the kernel
 * experiments (A/B kernel selection, phase ablation, in-kernel cycle stamps) read at all.  The shipped build returns 0 and never looks at
 * the environment: nothing a process inherits can change results or precision. */
int ake_build_has_diag(void);

/* ------------------------------------------------------------------------------------------
 * CQT front end.  Replaces
 *     librosa.cqt(y, sr, hop_length=round(sr/frames), bins_per_octave=36, n_bins=36*octaves)
 *     -> torch.abs -> torch.log(1 + .)                 KeyDataset.py:485, 490-499
 * (also equivariance_test.py:161-169).  Definition: oracle/cqt_oracle.py (direct form);
 * evaluated here octave-recursively (half-band decimators + per-phase filter banks).
 * ---------------------------------------------------------------------------------------- */
typedef struct ake_cqt_plan ake_cqt_plan;

typedef struct ake_cqt_config {
    int sample_rate;      /* Hz, e.g. 22050 */
    int hop_length;       /* samples between frame centres; KeyDataset.py:485 */
    int n_bins;           /* 36 * octaves; KeyDataset.py:491 */
    int bins_per_octave;  /* 36 */
    double fmin;          /* Hz of bin 0; <=0 selects librosa's default C1 = 32.7032 Hz */
    int q_mode;           /* 0: Q = (r^2+1)/(r^2-1) (librosa >= 0.10); 1: Q = 1/(r-1) (<= 0.9) */
    int decim_half_len;   /* half length of the half-band decimator; <=0 selects 23 (47 taps) */
    double decim_beta;    /* Kaiser beta of the decimator; <=0 selects 8.0 */
    int engine;           /* 0: fastest available (3 where it applies); 1: one kernel per decimation stage + f32 filter bank (first version, kept
                             as the in-library cross-check); 2: fused decimator cascade + f32 bank (bit-identical to 1);
                             3: fused cascade writing split-bf16 level signals + bf16x3 MFMA bank (needs <= 8 octaves);
                             5 (opt-in): the half-band stages of levels 0-4 as Toeplitz products on bf16 MFMA with the clips as the N
                             dimension, chained through registers (cqt_stream.h), deeper levels and the bank as engine 3 (6-8 octaves,
                             47-tap decimator; measured level with engine 3 at 256 clips, so 0 never selects it).  4 was removed. */
} ake_cqt_config;

/* hop = round(sample_rate / frames_per_second) (KeyDataset.py:485), n_bins = 36 * octaves. */
int ake_cqt_default_config(ake_cqt_config* cfg, int sample_rate, int frames_per_second, int octaves);
/* Builds the filter tables in double precision on the host and uploads them to the current device. */
int ake_cqt_plan_create(const ake_cqt_config* cfg, ake_cqt_plan** out);
void ake_cqt_plan_destroy(ake_cqt_plan* plan);
int ake_cqt_plan_n_bins(const ake_cqt_plan* plan);
int ake_cqt_plan_hop(const ake_cqt_plan* plan);
/* 1 + n_samples / hop  (librosa center=True framing). */
int64_t ake_cqt_num_frames(const ake_cqt_plan* plan, int64_t n_samples);
size_t ake_cqt_workspace_bytes(const ake_cqt_plan* plan, int batch, int64_t n_samples);
/*
 * audio_dev : [batch][audio_stride] float32, first n_samples of each row are the clip
 * out_dev   : [batch][n_bins][out_frames] float32 = log(1 + |CQT|); frames >= num_frames are
 *             written as 0 (the zero padding KeyDataset.__getitem__ appends, KeyDataset.py:245)
 */
int ake_cqt_logmag_f32(const ake_cqt_plan* plan, const float* audio_dev, int batch, int64_t n_samples,
                       int64_t audio_stride, float* out_dev, int64_t out_frames, void* workspace,
                       size_t workspace_bytes, ake_stream_t stream);

/* Ragged batch (clips of different lengths in one call).  Row i holds n_samples_dev[i] <= n_max samples
 * (device array, int64); whatever follows them in the row is never read (hardware range checking returns 0, the transform's
 * zero padding).  Clip i gets 1 + n_samples[i] / hop frames; frames beyond them are written as 0, as KeyDataset.__getitem__
 * pads to the longest clip (KeyDataset.py:245).  Workspace as for (batch, n_max).  Engine 3 (the default up to 8 octaves). */
int ake_cqt_logmag_ragged_f32(const ake_cqt_plan* plan, const float* audio_dev, int batch, int64_t n_max,
                              int64_t audio_stride, const int64_t* n_samples_dev, float* out_dev, int64_t out_frames,
                              void* workspace, size_t workspace_bytes, ake_stream_t stream);

/* Frames-major output: the same transform left as the filter bank writes it, out_dev = [batch][num_frames][n_bins] -- no transpose
 * pass (a 48 MB round trip per 256 clips).  For consumers that can read that order: ake_pcnet_forward_frames_major_f32, and
 * ake_pipeline_forward_f32 uses the pair internally.  Engine 3, equal-length clips (ake_cqt_frames_major_supported). */
int ake_cqt_frames_major_supported(const ake_cqt_plan* plan);
int ake_cqt_logmag_frames_major_f32(const ake_cqt_plan* plan, const float* audio_dev, int batch, int64_t n_samples,
                                    int64_t audio_stride, float* out_dev, void* workspace, size_t workspace_bytes,
                                    ake_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * PitchClassNet forward (inference: eval-mode BatchNorm folded into the convolutions).
 * Replaces  PitchClassNet.forward(mel, seq_length)   models.py:747-817
 * and everything it calls (models.py:22-106, 135-143, 168-243, 246-399).
 * ---------------------------------------------------------------------------------------- */
typedef struct ake_pcnet ake_pcnet;

typedef struct ake_pcnet_config {
    int pitches;        /* CQT bins = 36*octaves; models.py:653 */
    int pitch_classes;  /* 12 */
    int num_layers;     /* opt.num_layers, default 2 */
    int kernel_size;    /* opt.kernel_size, default 7; 3 and 5 are built too (every convolution then runs the generic kernels) */
    int conv_layers;    /* opt.conv_layers, default 3 */
    int n_filters;      /* opt.n_filters, default 4 */
    int head_layers;    /* opt.head_layers, default 2 */
    int time_pool_size; /* opt.time_pool_size, default 2 */
    int genre;          /* opt.genre: 1 adds the 11-way genre head */
    int max_pool;       /* opt.max_pool (models.py:766-797, sample-0 quirk kept) */
    /* Non-default architecture variants (models.py:108-133,145-166,456-648): only_semitones must be 0 (it does not run in the reference either),
     * ake_pcnet_create returns AKE_ERR_UNSUPPORTED otherwise. */
    int resblock;       /* opt.resblock (models.py:181-187, 218-224, 402-454): 1 builds the residual-block stacks (inference and training) */
    int denseblock;     /* opt.denseblock (models.py:188-189, 225-226, 456-648): 1 builds the DenseNet-style stacks (pre-activation BatchNorm applied on
                         * load, 1-wide bottlenecks, features concatenated in place); not combinable with the other variants; inference only */
    int stay_sixth;     /* opt.stay_sixth (models.py:322-323, 336, 366-367): 1 keeps the pitch stream at semitone resolution after layer 0 */
    int only_semitones;
    int p2pc_conv;      /* opt.p2pc_conv (models.py:108-133): 1 folds the octaves with a learned dilated conv + BN + LeakyReLU instead of the max */
    int pc2p_mem;       /* opt.pc2p_mem (models.py:145-166): 1 adds the summed up_sixth map to the pitch stream instead of concatenating */
    /* opt.local (sliding-window key tracking, models.py:720-722): 0 = off, else the heads' pooling window
     * W = opt.frames * opt.loc_window_size - head_layers * (kernel_size - 1).  The layers then do not pool over time
     * (models.py:348, 394: time_pool_size is ignored) and the forward is ake_pcnet_forward_local_f32.  Inference only. */
    int local;
    /* Arithmetic of the INFERENCE convolutions (training always uses f32-equivalent products).  Storage, accumulation and every
     * non-convolution op are f32 in both modes.
     *   AKE_PRECISION_MIXED (0, default): the 7x7 pitch convolutions, the semitone convolutions and layer 0's pitch-class stack multiply f16
     *       activations with f16 weights (per-channel power-of-two scaled), one MFMA product; the last layer's pitch-class stack and the heads
     *       use 3-term split-bf16 products.  Outputs within ~2e-5 of the float64 reference on seeded and on trained weights (budget 1e-3).
     *   AKE_PRECISION_F32X3 (1): no operand is rounded below 2^-17: exact-f32 MFMA / VALU kernels for the pitch, semitone and layer-0
     *       convolutions, 3-term split-bf16 products (hi*hi + lo*hi + hi*lo) for the pitch-class stack and the heads.  ~4x slower. */
    int precision;
} ake_pcnet_config;
#define AKE_PRECISION_MIXED 0
#define AKE_PRECISION_F32X3 1
/* the precision a handle runs its inference convolutions in (AKE_PRECISION_*) */
int ake_pcnet_precision(const ake_pcnet* net);

int ake_pcnet_default_config(ake_pcnet_config* cfg, int octaves, int genre);
int ake_pcnet_create(const ake_pcnet_config* cfg, ake_pcnet** out);
void ake_pcnet_destroy(ake_pcnet* net);

int ake_pcnet_pitches(const ake_pcnet* net);

/* The float entries of the reference state_dict this configuration expects
 * (models.py:993 / eval.py:115 strict=True): names and shapes, for host-side validation. */
int ake_pcnet_num_tensors(const ake_pcnet* net);
int ake_pcnet_tensor_info(const ake_pcnet* net, int index, const char** name, int64_t shape[4], int* ndim);
/* Hand one state_dict entry (host float32, reference layout) to the handle. */
int ake_pcnet_set_tensor(ake_pcnet* net, const char* name, const float* host_data, const int64_t* shape, int ndim);
/* All tensors set -> fold BatchNorm (running stats, eps 1e-5), repack for the kernels, upload. */
int ake_pcnet_finalize(ake_pcnet* net);
/* Device-resident parameters (training: models.py:1017-1027 updates the weights every optimizer step).  params_dev is
 * ONE flat float32 buffer holding every float entry of the state_dict at ake_pcnet_grad_offset(name)
 * (ake_pcnet_grad_floats() floats; valid right after ake_pcnet_create).  Rebuilds every packed weight -- eval-mode
 * BatchNorm folding included, bit-identical to set_tensor + finalize -- with two kernels on `stream`; replaces
 * set_tensor/finalize for callers whose weights live on the device.  The buffer is only read during the call. */
int ake_pcnet_load_from_device_f32(ake_pcnet* net, const float* params_dev, ake_stream_t stream);
/* The same, but only what the TRAINING-mode entry points read (ake_pcnet_forward_train_f32, ake_pcnet_backward_f32): the MFMA fragments of the
 * inference kernels are left stale (half of the repack launches of a training step) and inference calls return AKE_ERR_STATE until the
 * next ake_pcnet_load_from_device_f32.  models.py:1017-1027 updates the weights every optimizer step; eval happens once per epoch. */
int ake_pcnet_load_for_training_f32(ake_pcnet* net, const float* params_dev, ake_stream_t stream);
/* nn.BatchNorm2d's train-mode side effect, on the flat parameter buffer: running_mean/var <- (1-m)*running + m*(batch mean,
 * UNBIASED batch variance), from the bn_stats a training forward returned.  Reference default momentum 0.1. */
int ake_pcnet_update_running_stats_f32(const ake_pcnet* net, const float* bn_stats_dev, float* params_dev, float momentum,
                                       ake_stream_t stream);
/* --denseblock: the reference checkpoints norm1 + conv1 of every dense layer (torch.utils.checkpoint, models.py:484-489, 553), so autograd's
 * BACKWARD runs that half a second time in train mode and its BatchNorm blends the batch statistics into the running ones once more
 * (num_batches_tracked counts 2 per step for those layers).  Call this after ake_pcnet_backward_f32 with the bn_stats of the forward the
 * backward belonged to: it repeats the blend for exactly those layers.  *channels (nullable) receives how many BatchNorm channels that is;
 * a no-op (0 channels) for every other configuration. */
int ake_pcnet_update_recomputed_running_stats_f32(const ake_pcnet* net, const float* bn_stats_dev, float* params_dev, float momentum, int* channels,
                                                  ake_stream_t stream);

size_t ake_pcnet_workspace_bytes(const ake_pcnet* net, int batch, int frames);
/*
 * mel_dev        : [batch][1][pitches][frames] float32 (log-CQT, zero padded to `frames`)
 * seq_length_dev : [batch] int64 valid frames per clip, or NULL (models.py:786-797 branch)
 * key_out_dev    : [batch][12]  sigmoid pitch-class membership           (models.py:802)
 * tonic_out_dev  : [batch][12]  tonic logits                             (models.py:800)
 * genre_out_dev  : [batch][11]  genre logits, ignored unless cfg.genre   (models.py:804)
 */
int ake_pcnet_forward_f32(const ake_pcnet* net, const float* mel_dev, int batch, int frames,
                          const int64_t* seq_length_dev, float* key_out_dev, float* tonic_out_dev,
                          float* genre_out_dev, void* workspace, size_t workspace_bytes, ake_stream_t stream);

/* --local forward (net created with cfg.local = W > 0; models.py:805-810).  Tm = frames - head_layers * (kernel_size - 1) map
 * frames, Tq = Tm - W + 1 pooled frames (ake_pcnet_local_frames).  Outputs, in the reference's memory order (it *reshapes*
 * [B][1][12][Tq] to (B, Tq, 12) -- same bytes):
 *   key_out_dev, tonic_out_dev : [batch][12 * Tq]   sliding max over W frames of the head maps; sigmoid on key
 *   genre_out_dev              : [batch][11 * Tm]   the genre head's map */
int ake_pcnet_local_frames(const ake_pcnet* net, int frames, int* pooled_frames, int* map_frames);
/* The forward with mel as ake_cqt_logmag_frames_major_f32 leaves it, [batch][frames][pitches]: the two kernels that read the CQT (layer 0
 * and the first pitch convolution) transpose while staging.  Only the default architecture on the fused inference path takes it:
 * ake_pcnet_accepts_frames_major(net, batch, frames) != 0; otherwise AKE_ERR_UNSUPPORTED.  Same results as ake_pcnet_forward_f32 on the
 * transposed tensor, bit for bit. */
int ake_pcnet_accepts_frames_major(const ake_pcnet* net, int batch, int frames);
int ake_pcnet_forward_frames_major_f32(const ake_pcnet* net, const float* mel_frames_major_dev, int batch, int frames,
                                       const int64_t* seq_length_dev, float* key_out_dev, float* tonic_out_dev, float* genre_out_dev,
                                       void* workspace, size_t workspace_bytes, ake_stream_t stream);

int ake_pcnet_forward_local_f32(const ake_pcnet* net, const float* mel_dev, int batch, int frames, float* key_out_dev,
                                float* tonic_out_dev, float* genre_out_dev, void* workspace, size_t workspace_bytes,
                                ake_stream_t stream);

/* Training-mode forward: BatchNorm uses the statistics of this batch (nn.BatchNorm2d in train(), as
 * equivariance_test.py:178 runs the net and as training_step does, models.py:952).  Convolutions keep their raw
 * outputs and per-channel sums; normalisation + LeakyReLU are applied by the next reader, never as a pass of their own.
 * bn_stats_out (optional, device): [sum of BN channels][3] = batch mean, biased batch variance, elements per channel, in
 * the layer order of ake_pcnet_bn_info -- what the caller needs for torch's running_mean / running_var update
 * (momentum 0.1, unbiased variance).  The whole batch is processed in one pass (no chunking).
 * A net created with local > 0 (--local, models.py:805-810) takes seq_length_dev = NULL and writes the per-frame outputs of
 * ake_pcnet_forward_local_f32: key / tonic (B, T', 12), genre (B, Tm, 11) with (T', Tm) from ake_pcnet_local_frames. */
int ake_pcnet_num_bn(const ake_pcnet* net);
int ake_pcnet_bn_info(const ake_pcnet* net, int index, const char** name, int* channels, int* channel_offset);
size_t ake_pcnet_train_workspace_bytes(const ake_pcnet* net, int batch, int frames);
int ake_pcnet_forward_train_f32(const ake_pcnet* net, const float* mel_dev, int batch, int frames,
                                const int64_t* seq_length_dev, float* key_out_dev, float* tonic_out_dev,
                                float* genre_out_dev, float* bn_stats_out_dev, void* workspace, size_t workspace_bytes,
                                ake_stream_t stream);

/* Backward pass of the training-mode forward (what autograd does for the reference's training_step, models.py:952-963).
 * Must follow ake_pcnet_forward_train_f32 on the same (mel, batch, frames, seq_length, workspace): the workspace holds the raw
 * convolution outputs and BatchNorm batch statistics it needs.  d_*_dev are dLoss/d(key_out, tonic_out, genre_out); key_out_dev
 * is the forward's key output (for the sigmoid derivative).  grads_out_dev receives dLoss/d(parameter) for every float entry of
 * the state_dict, flat, at ake_pcnet_grad_offset(name) (ake_pcnet_grad_floats() floats in total; running statistics get zeros).
 * accumulate != 0 adds to grads_out_dev instead of overwriting it (accumulate_grad_batches, train_model.py:118).
 * For a --local net d_* and key_out carry the per-frame shapes of the local forward (the sliding-window max routes each frame's
 * gradient to the first maximum of its window, as nn.MaxPool2d does).
 * Any num_layers; --resblock / --pc2p_mem / --p2pc_conv / --stay_sixth train too; --denseblock: AKE_ERR_UNSUPPORTED. */
size_t ake_pcnet_grad_floats(const ake_pcnet* net);
int64_t ake_pcnet_grad_offset(const ake_pcnet* net, const char* name);
int ake_pcnet_backward_f32(const ake_pcnet* net, const float* mel_dev, int batch, int frames, const int64_t* seq_length_dev,
                           const float* key_out_dev, const float* d_key_dev, const float* d_tonic_dev, const float* d_genre_dev,
                           float* grads_out_dev, int accumulate, void* workspace, size_t workspace_bytes, ake_stream_t stream);

/* Fused Adam over flat buffers, torch.optim.Adam semantics as the reference configures it (models.py:1017-1027:
 * betas (0.9, 0.999), eps 1e-8, weight_decay = --reg as L2 added to the gradient, bias correction with `step` = 1, 2, ...):
 *   g = grad * grad_scale + weight_decay * p;  m = b1*m + (1-b1)*g;  v = b2*v + (1-b2)*g*g
 *   p -= lr / (1 - b1^step) * m / (sqrt(v) / sqrt(1 - b2^step) + eps)
 * trainable_dev (nullable): one byte per element, 0 = leave untouched (running statistics inside the flat buffer).
 * grad_scale folds the 1/world_size of a summed all-reduce (and 1/accumulate_grad_batches if the caller did not scale the loss). */
int ake_adam_step_f32(float* params_dev, const float* grads_dev, float* exp_avg_dev, float* exp_avg_sq_dev,
                      const unsigned char* trainable_dev, size_t count, float lr, float beta1, float beta2, float eps,
                      float weight_decay, int step, float grad_scale, ake_stream_t stream);

/* Everything PitchClassNet.general_step computes behind the forward, in one launch (replaces models.py:826-905: label argmax / genre
 * mask, key_weight * BCE(key) + tonic_weight * CE(tonic) + genre_weight * CE(genre rows whose one-hot label sums to 1; dropped when there
 * is none) [+ 1 - mean cosine(key, key_labels) with use_cos], and models.py:1065-1116: the MIREX categories over the 21-row key-signature
 * table, utils/key_signatures.py:19-42).  One-hot label tensors are float32 or int64 (the *_i64 flags); genre_* nullable (--genre off).
 * scalars_out_dev[10] = loss, accuracy, mirex_score, correct, fifths, relative, parallel, other, accuracy_tonic, accuracy_genre
 * (general_step's return order).  d_*_dev (all or none): dloss/d(key_out | tonic_out | genre_out), what loss.backward() hands to
 * ake_pcnet_backward_f32. */
int ake_general_step_f32(const float* key_out_dev, const float* tonic_out_dev, const float* genre_out_dev, const float* key_labels_dev,
                         const void* tonic_labels_dev, int tonic_labels_i64, const void* genre_labels_dev, int genre_labels_i64,
                         const void* key_signature_id_dev, int key_signature_i64, int batch, float key_weight, float tonic_weight,
                         float genre_weight, int use_cos, float* scalars_out_dev, float* d_key_dev, float* d_tonic_dev, float* d_genre_dev,
                         ake_stream_t stream);

/* Debug tap: copy an intermediate activation of the LAST forward call out of the workspace.
 * name is the reference module path whose output it is (e.g. "model.1.p2p.layer.8"). */
int ake_pcnet_tap_info(const ake_pcnet* net, const char* name, int batch, int frames, int64_t shape[4]);
/* Inference fuses the semitone conv into the last pitch conv of a stack and runs the last layer's pitch-class stack as one
 * launch, so "model.i.p2p.layer.8" and the last layer's "pc2pc.layer.{2,5,8}" are never written (tap_info says so).
 * ake_debug_keep_taps(1) (process-wide, before the forward) keeps every nameable activation in memory (slower); returns the old value. */
int ake_debug_keep_taps(int on);
int ake_pcnet_tap_copy(const ake_pcnet* net, const char* name, int batch, int frames, const void* workspace,
                       float* out_dev, ake_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Audio preparation in front of the CQT (not in the reference, which takes channel 0 at the file's own
 * rate: KeyDataset.py:479-485 -- channel = 0 and rate_in == rate_out reproduce exactly that): channel selection or mono
 * mix-down and polyphase resampling = scipy.signal.resample_poly(x, up, down) with its default Kaiser(5.0) filter, on the
 * device, for ragged batches.
 * in_dev[clip * clip_stride + c * channel_stride + i]; channel >= 0 selects one channel, -1 takes the mean of all;
 * n_in_clip_dev (or null): samples of each clip (<= n_in); out_dev[clip * out_stride + k], k < ake_resampler_out_len(n_in),
 * zero behind a shorter clip; n_out_clip_dev (or null) receives every clip's output length ceil(n * up / down).
 * ---------------------------------------------------------------------------------------- */
typedef struct ake_resampler ake_resampler;
int ake_resampler_create(int rate_in, int rate_out, ake_resampler** out);
void ake_resampler_destroy(ake_resampler* r);
int64_t ake_resampler_out_len(const ake_resampler* r, int64_t n_in);
int ake_resample_f32(const ake_resampler* r, const float* in_dev, int batch, int channels, int64_t n_in, int64_t clip_stride,
                     int64_t channel_stride, int channel, const int64_t* n_in_clip_dev, float* out_dev, int64_t out_stride,
                     int64_t* n_out_clip_dev, ake_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Whole hot path for a batch of equal-length clips: CQT then forward
 * (DatasetLoader.get_all -> __getitem__ -> general_step's forward; KeyDataset.py:469-509,
 * 242-256, models.py:846).  seq_length of every clip = num_frames(n_samples).
 * ---------------------------------------------------------------------------------------- */
size_t ake_pipeline_workspace_bytes(const ake_cqt_plan* plan, const ake_pcnet* net, int batch, int64_t n_samples);
int ake_pipeline_forward_f32(const ake_cqt_plan* plan, const ake_pcnet* net, const float* audio_dev, int batch,
                             int64_t n_samples, int64_t audio_stride, float* key_out_dev, float* tonic_out_dev,
                             float* genre_out_dev, void* workspace, size_t workspace_bytes, ake_stream_t stream);
/* The same for a ragged batch (ake_cqt_logmag_ragged_f32): seq_length of clip i = 1 + n_samples[i] / hop, computed on the
 * device; workspace as ake_pipeline_workspace_bytes(plan, net, batch, n_max). */
int ake_pipeline_forward_ragged_f32(const ake_cqt_plan* plan, const ake_pcnet* net, const float* audio_dev, int batch,
                                    int64_t n_max, int64_t audio_stride, const int64_t* n_samples_dev, float* key_out_dev,
                                    float* tonic_out_dev, float* genre_out_dev, void* workspace, size_t workspace_bytes,
                                    ake_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Per-kernel timing with hipEvents recorded on the launch stream (bench.py roofline leg).
 * ---------------------------------------------------------------------------------------- */
int ake_prof_enable(const char* name_filter /* substring, NULL or "" = all */, int on);
int ake_prof_collect(void);                 /* synchronises the recorded events, accumulates */
int ake_prof_reset(void);
int ake_prof_num_entries(void);
int ake_prof_entry(int index, const char** kernel_name, double* total_ms, int64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* AKE_HIP_H */

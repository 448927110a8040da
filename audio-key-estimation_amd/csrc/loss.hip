// Everything PitchClassNet.general_step computes behind the forward, in ONE launch (reference: models.py:826-905 -- label
// preparation, BCE / cross-entropy losses with the genre mask and the optional cosine term -- and models.py:1065-1116, the MIREX
// categories over the 21-row key-signature table of utils/key_signatures.py:19-42), plus the gradient of the loss with respect to
// the three network outputs.  With torch ops this is ~110 tiny kernels forward and ~50 in autograd's backward per step: at the
// reference's batch size (8 clips) that was a quarter of the step, at 256 clips the host could not issue them as fast as the GPU
// finished them.  One workgroup; a thread owns whole rows (a clip's 12 + 12 + 11 outputs), sums in double, block reduction in a
// fixed order (bit-reproducible).
#include "common.h"

#include <cstdint>

namespace {

struct StepLossArgs {
    const float* key;          // [B][12] sigmoid outputs
    const float* tonic;        // [B][12] logits
    const float* genre;        // [B][11] logits, nullable
    const float* key_labels;   // [B][12]
    const void* tonic_lab;     // [B][12] one-hot, float32 or int64
    const void* genre_lab;     // [B][11] one-hot (rows that do not sum to 1 carry no genre label), nullable
    const void* sig_lab;       // [B][24] one-hot key-signature id
    int tonic_i64, genre_i64, sig_i64;
    int B;
    float key_w, tonic_w, genre_w;
    int use_cos;
    float* scalars;            // [10]: loss, accuracy, mirex, correct, fifths, relative, parallel, other, accuracy_tonic, accuracy_genre
    float* d_key;              // nullable (all three or none)
    float* d_tonic;
    float* d_genre;
};

// label element; float32 tonic / genre labels go through the reference's .long() (models.py:826, 831: truncation toward zero)
__device__ __forceinline__ double lab(const void* p, int is_i64, long long i, bool as_long = true) {
    if (is_i64) return static_cast<double>(static_cast<const long long*>(p)[i]);
    const double v = static_cast<double>(static_cast<const float*>(p)[i]);
    return as_long ? trunc(v) : v;
}

// first maximum of a one-hot row (torch.argmax on the labels, models.py:832 / :840 / :1090)
__device__ __forceinline__ int lab_argmax(const void* p, int is_i64, long long base, int n, bool as_long = true) {
    int best = 0;
    double bv = lab(p, is_i64, base, as_long);
    for (int j = 1; j < n; ++j) {
        const double v = lab(p, is_i64, base + j, as_long);
        if (v > bv) { bv = v; best = j; }
    }
    return best;
}

// tonic (pitch class of the major scale's first degree) of row k of the key-signature table: circle of fifths Cb .. C# (15 rows),
// then the six enharmonic duplicates (utils/key_signatures.py:19-42)
__device__ __forceinline__ int table_tonic(int k) {
    const int dup[6] = {9, 11, 10, 4, 3, 5};
    const int i = k < 15 ? k : dup[k - 15];
    return ((7 * (i - 7)) % 12 + 12) % 12;
}
__device__ __forceinline__ bool in_major_scale(int pc, int tonic) {
    const int d = ((pc - tonic) % 12 + 12) % 12;
    return d == 0 || d == 2 || d == 4 || d == 5 || d == 7 || d == 9 || d == 11;
}

constexpr int kNS = 12;        // per-thread sums: bce, ce_tonic, ce_genre (masked), genre count, genre correct, cos, tonic ok, full, correct, fifths, relative, parallel

__global__ __launch_bounds__(256) void general_step_kernel(StepLossArgs a) {
    __shared__ double red[kNS][256];
    const int B = a.B;
    double sum[kNS];
#pragma unroll
    for (int k = 0; k < kNS; ++k) sum[k] = 0.0;
    for (int r = threadIdx.x; r < B; r += blockDim.x) {
        // ---- key: BCE (log clamped at -100 as torch does), models.py:855, 878 ----
        double p[12], y[12];
        double pp = 0.0, yy = 0.0, py = 0.0;
        for (int j = 0; j < 12; ++j) {
            p[j] = static_cast<double>(a.key[r * 12 + j]);
            y[j] = static_cast<double>(a.key_labels[r * 12 + j]);
            const double lp = fmax(log(p[j]), -100.0), l1p = fmax(log1p(-p[j]), -100.0);
            sum[0] -= y[j] * lp + (1.0 - y[j]) * l1p;
            pp += p[j] * p[j]; yy += y[j] * y[j]; py += p[j] * y[j];
        }
        const double pn = fmax(sqrt(pp), 1e-8), yn = fmax(sqrt(yy), 1e-8);
        if (a.use_cos) sum[5] += py / (pn * yn);                                   // models.py:885-887
        if (a.d_key) {
            for (int j = 0; j < 12; ++j) {
                double g = a.key_w * (p[j] - y[j]) / fmax((1.0 - p[j]) * p[j], 1e-12) / (12.0 * B);      // torch's binary_cross_entropy_backward
                if (a.use_cos) g -= (y[j] / (pn * yn) - py * p[j] / (pn * pn * pn * yn)) / B;
                a.d_key[r * 12 + j] = static_cast<float>(g);
            }
        }
        // ---- tonic: cross entropy on the logits, models.py:856, 879 ----
        const int t_idx = lab_argmax(a.tonic_lab, a.tonic_i64, static_cast<long long>(r) * 12, 12);
        bool tok;                                                                  // the predicted tonic is the labelled one
        {
            double z[12], zmax = -1e300;
            int zarg = 0;
            for (int j = 0; j < 12; ++j) {
                z[j] = static_cast<double>(a.tonic[r * 12 + j]);
                if (z[j] > zmax) { zmax = z[j]; zarg = j; }
            }
            double se = 0.0;
            for (int j = 0; j < 12; ++j) se += exp(z[j] - zmax);
            sum[1] -= z[t_idx] - zmax - log(se);
            if (a.d_tonic)
                for (int j = 0; j < 12; ++j) a.d_tonic[r * 12 + j] = static_cast<float>(a.tonic_w * (exp(z[j] - zmax) / se - (j == t_idx ? 1.0 : 0.0)) / B);
            tok = zarg == t_idx;
            sum[6] += tok ? 1.0 : 0.0;
        }
        // ---- genre: cross entropy over the rows that carry a label, models.py:839-840, 881-883, 892-893 ----
        if (a.genre) {
            double ls = 0.0;
            for (int j = 0; j < 11; ++j) ls += lab(a.genre_lab, a.genre_i64, static_cast<long long>(r) * 11 + j);
            const double m = ls == 1.0 ? 1.0 : 0.0;
            const int g_idx = lab_argmax(a.genre_lab, a.genre_i64, static_cast<long long>(r) * 11, 11);
            double z[11], zmax = -1e300;
            int zarg = 0;
            for (int j = 0; j < 11; ++j) {
                z[j] = static_cast<double>(a.genre[r * 11 + j]);
                if (z[j] > zmax) { zmax = z[j]; zarg = j; }
            }
            double se = 0.0;
            for (int j = 0; j < 11; ++j) se += exp(z[j] - zmax);
            sum[2] -= m * (z[g_idx] - zmax - log(se));
            sum[3] += m;
            sum[4] += m * (zarg == g_idx ? 1.0 : 0.0);
            if (a.d_genre)   // scaled by 1 / (number of labelled rows) after the reduction
                for (int j = 0; j < 11; ++j) a.d_genre[r * 11 + j] = static_cast<float>(a.genre_w * m * (exp(z[j] - zmax) / se - (j == g_idx ? 1.0 : 0.0)));
        }
        // ---- MIREX categories, models.py:1065-1116: first-maximum cosine match over the 21 table rows ----
        {
            int pred = 0;
            double best = -1e300;
            for (int k = 0; k < 21; ++k) {
                const int tk = table_tonic(k);
                double dot = 0.0;
                for (int j = 0; j < 12; ++j) dot += in_major_scale(j, tk) ? p[j] : 0.0;
                const double sim = dot / (pn * fmax(sqrt(7.0), 1e-8));
                if (sim > best) { best = sim; pred = k; }
            }
            const int tp = table_tonic(pred);
            bool full = true;
            for (int j = 0; j < 12; ++j) full = full && ((in_major_scale(j, tp) ? 1.0 : 0.0) == y[j]);
            const int label_id = lab_argmax(a.sig_lab, a.sig_i64, static_cast<long long>(r) * 24, 24, false);   // (models.py:1090: no .long())
            const int diff = pred > label_id ? pred - label_id : label_id - pred;
            const bool fifths = diff == 1 && !(tok && full);
            const bool correct = tok && full && !fifths;
            const bool relative = full && !tok && !fifths;
            const bool parallel = tok && !full && !fifths;
            sum[7] += full ? 1.0 : 0.0;
            sum[8] += correct ? 1.0 : 0.0;
            sum[9] += fifths ? 1.0 : 0.0;
            sum[10] += relative ? 1.0 : 0.0;
            sum[11] += parallel ? 1.0 : 0.0;
        }
    }
#pragma unroll
    for (int k = 0; k < kNS; ++k) red[k][threadIdx.x] = sum[k];
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (static_cast<int>(threadIdx.x) < w)
#pragma unroll
            for (int k = 0; k < kNS; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + w];
        __syncthreads();
    }
    const double cnt = red[3][0];
    const double inv_cnt = 1.0 / fmax(cnt, 1.0);
    if (a.genre && a.d_genre)
        for (int r = threadIdx.x; r < B; r += blockDim.x)          // the rows this thread wrote above
            for (int j = 0; j < 11; ++j) a.d_genre[r * 11 + j] = static_cast<float>(static_cast<double>(a.d_genre[r * 11 + j]) * inv_cnt);
    if (threadIdx.x == 0) {
        double loss = a.key_w * red[0][0] / (12.0 * B) + a.tonic_w * red[1][0] / B;                 // models.py:889
        if (a.genre) loss += a.genre_w * red[2][0] * inv_cnt;                                       // an exact zero when no row carries a label (:892)
        if (a.use_cos) loss += 1.0 - red[5][0] / B;
        const double n = B;
        const double correct = red[8][0], fifths = red[9][0], relative = red[10][0], parallel = red[11][0];
        a.scalars[0] = static_cast<float>(loss);
        a.scalars[1] = static_cast<float>(red[7][0] / n);                                           // accuracy = all 12 key bits right
        a.scalars[2] = static_cast<float>((1.0 * correct + 0.5 * fifths + 0.3 * relative + 0.2 * parallel) / n);
        a.scalars[3] = static_cast<float>(correct / n);
        a.scalars[4] = static_cast<float>(fifths / n);
        a.scalars[5] = static_cast<float>(relative / n);
        a.scalars[6] = static_cast<float>(parallel / n);
        a.scalars[7] = static_cast<float>((n - correct - fifths - relative - parallel) / n);
        a.scalars[8] = static_cast<float>(red[6][0] / n);
        a.scalars[9] = static_cast<float>(a.genre ? red[4][0] * inv_cnt : 0.0);
    }
}

}  // namespace

extern "C" int ake_general_step_f32(const float* key_out, const float* tonic_out, const float* genre_out, const float* key_labels,
                                    const void* tonic_labels, int tonic_labels_i64, const void* genre_labels, int genre_labels_i64,
                                    const void* key_signature_id, int key_signature_i64, int batch, float key_weight, float tonic_weight,
                                    float genre_weight, int use_cos, float* scalars_out, float* d_key, float* d_tonic, float* d_genre,
                                    ake_stream_t stream) {
    AKE_REQUIRE(key_out && tonic_out && key_labels && tonic_labels && key_signature_id && scalars_out, AKE_ERR_INVALID, "general_step: null argument");
    AKE_REQUIRE(batch >= 1, AKE_ERR_INVALID, "general_step: batch %d", batch);
    AKE_REQUIRE(!genre_out || genre_labels, AKE_ERR_INVALID, "general_step: genre outputs without genre labels");
    const bool grads = d_key || d_tonic || d_genre;
    AKE_REQUIRE(!grads || (d_key && d_tonic && (d_genre || !genre_out)), AKE_ERR_INVALID, "general_step: pass every gradient buffer or none");
    StepLossArgs a;
    a.key = key_out; a.tonic = tonic_out; a.genre = genre_out; a.key_labels = key_labels;
    a.tonic_lab = tonic_labels; a.genre_lab = genre_labels; a.sig_lab = key_signature_id;
    a.tonic_i64 = tonic_labels_i64; a.genre_i64 = genre_labels_i64; a.sig_i64 = key_signature_i64;
    a.B = batch; a.key_w = key_weight; a.tonic_w = tonic_weight; a.genre_w = genre_weight; a.use_cos = use_cos;
    a.scalars = scalars_out; a.d_key = d_key; a.d_tonic = d_tonic; a.d_genre = genre_out ? d_genre : nullptr;
    hipStream_t s = static_cast<hipStream_t>(stream);
    ake::ProfScope ps("general_step_kernel", s);
    hipLaunchKernelGGL(general_step_kernel, dim3(1), dim3(256), 0, s, a);
    AKE_HIP_CHECK(hipGetLastError());
    return AKE_OK;
}

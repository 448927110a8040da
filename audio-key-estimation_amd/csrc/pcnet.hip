// PitchClassNet forward on gfx950: handle, weight registry, BatchNorm folding, packing, launch plan.
//
// Replaces PitchClassNet.forward (models.py:747-817) for the default architecture family
// (models.py:190-197, 227-234, 311-350).  Channel algebra follows models.py:279-308 / 693-710.
//
// HBM layout (all fp32, NCHW; the pitch stream runs in chunks of <= chunk_clips clips (AKE_PCNET_CHUNK, default
// 256): measured on MI355X a larger launch beats Infinity-Cache residency of the 700 KB/clip activations):
//   mel      [B][1][P][T]                    caller's
//   fold0    [c][1][12][T]                   semitone conv + octave fold of layer 0
//   cat_i    [c][prev_pc + out_p][12][T_i]   concat buffer of layer i: producers write their channel slice
//   psix     [c][prev_pc][36][T_i]           up_sixth output; the x(P/36) row repeat is never materialised
//   pa / pb  [c][out_p][P][T_i]              pitch-conv ping-pong
//   pc a/b   [c][out_pc][12][T_i]            pitch-class conv ping-pong
//   pcf      [c][final][12][T_f]             time-pooled features feeding the heads
//   hid      [c][2*final][12][.]             head hidden maps;  maps [c][rows][T_m]
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "common.h"
#include "pcnet_kernels.h"
#include "pcnet_bwd_kernels.h"

using namespace ake_k;

namespace {

constexpr double kBnEps = 1e-5;          // nn.BatchNorm2d default
constexpr size_t kLdsBudget = 64 * 1024; // dynamic LDS per workgroup we allow ourselves
constexpr size_t kBfFragsPerConv = 14 * 2 * 64 + 64;   // uint4 entries per 8->8 pitch convolution: f16 weight fragments [14 k-steps][hi|lo][64 lanes], then the 8 inverse channel scales

struct TensorSpec {
    std::string name;
    int64_t shape[4];
    int ndim;
};

struct HostTensor {
    std::vector<float> data;
    bool set = false;
};

struct PackedConv {          // device-resident folded + packed convolution
    int cin = 0, cout = 0, kh = 0, kw = 0, co = 1, groups = 1;
    size_t w_off = 0, b_off = 0;   // float offsets into the blob (VALU layout [group][ci][dy][dx][CO])
    // f32-MFMA fragment layout [ci][dy][s][ntile][64 lanes] (Toeplitz over TB frames), see pcnet_kernels.h
    int tb = 0, ku = 0, ntiles = 0, nt = 1;
    int row_k = 0;                 // 1: 12 x 1 kernel in the row-k fragment form (k = four consecutive rows, kh / 4 steps per channel)
    size_t f_off = 0;
    long long bf_off = -1;         // 16-byte offset of the bf16x3 fragments in ake_pcnet::bf_frags_dev (8 -> 8 channel 7x7 pitch convs), or -1
    long long l0_off = -1;         // ... of the layer-0 form (layer0_mfma_kernel: <= 4 channels, 12 x 7), or -1
    long long bf_off2 = -1;        // ... of input channels [16, 32) of a 32-channel data-gradient pack (heads' first conv, f16 x 3), or -1
};

struct LayerDims {
    int prev_p = 0, prev_pc = 0, out_p = 0, out_pc = 0;
};

// --denseblock: one _DenseLayer / _DenseLayerEquivariant (models.py:456-582).  Both convolutions keep their raw weights; the BatchNorm +
// activation in FRONT of each is a row table (scale, shift, negative slope) in ake_pcnet::dense_aff_dev that the kernel applies on load.
struct DensePack {
    PackedConv c1, c2;       // 1-wide bottleneck (stored as 7 taps, centre one non-zero), k-wide convolution
    size_t aff1 = 0, aff2 = 0;   // float offsets of the two (eval-mode) tables
    // training: the two BatchNorm layers (indices into ake_pcnet::bns), the data-gradient packs (transposed + flipped weights, stored like c1 / c2)
    // and the parameter names
    int bn1 = -1, bn2 = -1;
    PackedConv d1, d2;
    std::string norm1, norm2, w1, w2, b2;       // b2 empty: bias=False (the plain Conv2d form)
};

int pick_co(int cout) { return cout >= 8 ? 8 : (cout >= 2 ? 4 : 1); }

}  // namespace

struct ake_pcnet {
    ake_pcnet_config cfg;
    std::vector<LayerDims> dims;
    int final_ch = 0;
    std::vector<TensorSpec> specs;
    std::map<std::string, int> spec_index;
    std::vector<HostTensor> host;
    bool finalized = false;
    bool eval_frags_stale = false;   // ake_pcnet_load_for_training_f32 skipped the MFMA fragments only the inference kernels read
    int chunk_clips = 256;

    // packed parameters
    std::vector<float> blob;
    float* blob_dev = nullptr;
    std::vector<PackedConv> foldc, foldc_t;       // per layer: --p2pc_conv's octave-fold convolution [co][ci][n_oct]
    std::vector<PackedConv> semi;                 // per layer
    std::vector<std::vector<PackedConv>> pc2pc;   // per layer, conv_layers entries
    std::vector<PackedConv> up;                   // per layer (index 0 unused)
    std::vector<std::vector<PackedConv>> p2p;     // per layer (index 0 empty)
    std::vector<PackedConv> head_key, head_tonic, head_genre;

    // training-mode packs: raw convolution weights (no BatchNorm folding) + the BatchNorm layers themselves
    std::vector<PackedConv> semi_t, up_t, head_key_t, head_tonic_t, head_genre_t;
    std::vector<std::vector<PackedConv>> pc2pc_t, p2p_t;
    struct BnLayer { std::string name; int C; size_t gamma_off, beta_off; int ch_off; };
    std::vector<BnLayer> bns;
    std::map<std::string, int> bn_index;
    int bn_channels = 0;
    // backward: data-gradient packs (transposed + flipped raw weights, same fragment format) and the flat gradient layout
    std::vector<std::vector<PackedConv>> pc2pc_d, p2p_d;
    std::vector<PackedConv> head_key_d, head_tonic_d, head_genre_d;
    std::vector<size_t> grad_off;      // float offset of specs[i] in the flat gradient buffer
    size_t grad_floats = 0;
    std::vector<size_t> raw_w_off;     // float offset in the blob of specs[i]'s raw values (convolution weights used by small kernels)

    // device-side rebuild of the blob from a flat parameter buffer (same layout as the gradient buffer):
    //   folded[i] = eval-mode BatchNorm folded into params[i] (conv weights / biases in front of a BatchNorm), else params[i]
    //   blob[j]   = map[j] < 0 ? 0 : (map[j] & kMapFolded ? folded : params)[map[j] & kMapIndex]
    struct FoldRecord { std::string wkey, bkey, bn; int cout; size_t per_out; bool transposed; int cin; };
    std::vector<FoldRecord> fold_records;
    std::vector<int32_t> map_host;     // one entry per blob float
    int32_t* map_dev = nullptr;
    int32_t* fold_ch_dev = nullptr;    // per flat float: -1, or (eval BatchNorm channel << 1) | is_bias
    int32_t* fold_bn_dev = nullptr;    // per eval BatchNorm channel: flat offsets of gamma, beta, running_mean, running_var
    int32_t* run_off_dev = nullptr;    // per training BatchNorm channel (BnLayer::ch_off order): flat offsets of running_mean, running_var
    int32_t* run_off2_dev = nullptr;   // the same for the layers the reference's BACKWARD runs a second time (checkpointed halves: --denseblock's norm1), -1 elsewhere
    int recomputed_bn_channels = 0;
    float* folded_dev = nullptr;
    bool tracing = false;
    std::vector<std::vector<DensePack>> dense_pc, dense_p;   // --denseblock: per layer, conv_layers entries
    struct AffRec { std::string bn; int C; float slope; size_t off; };
    std::vector<AffRec> aff_recs;      // BatchNorm layers behind dense_aff_dev, in table order
    size_t dense_aff_floats = 0;
    float* dense_aff_dev = nullptr;
    int32_t* dense_aff_idx_dev = nullptr;
    uint4* bf_frags_dev = nullptr;     // split-bf16 weight fragments of conv_p2p_f16_kernel, rebuilt from the eval packs on the device
    size_t bf_frags_count = 0;         // uint4 entries
};

namespace {

void add_spec(ake_pcnet* n, const std::string& name, std::initializer_list<int64_t> shape) {
    TensorSpec s;
    s.name = name;
    s.ndim = static_cast<int>(shape.size());
    int i = 0;
    for (auto v : shape) s.shape[i++] = v;
    for (; i < 4; ++i) s.shape[i] = 1;
    n->spec_index[name] = static_cast<int>(n->specs.size());
    n->specs.push_back(s);
}

void add_conv_specs(ake_pcnet* n, const std::string& prefix, int cout, int cin, int kh, int kw) {
    add_spec(n, prefix + ".weight", {cout, cin, kh, kw});
    add_spec(n, prefix + ".bias", {cout});
}

void add_bn_specs(ake_pcnet* n, const std::string& prefix, int c) {
    for (const char* f : {".weight", ".bias", ".running_mean", ".running_var"}) add_spec(n, prefix + f, {c});
}

const std::vector<float>& T(const ake_pcnet* n, const std::string& name) {
    return n->host[n->spec_index.at(name)].data;
}

// conv weight [cout][cin][kh][kw] + bias, optionally followed by BatchNorm(bn_prefix) in eval mode:
//   y = (conv(x) - mean) * gamma / sqrt(var + eps) + beta  ==  conv'(x) with w' = w*s, b' = (b - mean)*s + beta
void fold(const ake_pcnet* n, const std::string& wkey, const std::string& bkey, const std::string& bn_prefix,
          int cout, size_t per_out, bool transposed_cin_first, int cin, std::vector<double>& w, std::vector<double>& b) {
    const auto& w32 = T(n, wkey);
    const auto& b32 = T(n, bkey);
    w.assign(w32.begin(), w32.end());
    b.assign(b32.begin(), b32.end());
    if (bn_prefix.empty()) return;
    if (n->tracing) const_cast<ake_pcnet*>(n)->fold_records.push_back({wkey, bkey, bn_prefix, cout, per_out, transposed_cin_first, cin});
    const auto& g = T(n, bn_prefix + ".weight");
    const auto& be = T(n, bn_prefix + ".bias");
    const auto& mu = T(n, bn_prefix + ".running_mean");
    const auto& var = T(n, bn_prefix + ".running_var");
    for (int co = 0; co < cout; ++co) {
        const double s = static_cast<double>(g[co]) / std::sqrt(static_cast<double>(var[co]) + kBnEps);
        if (!transposed_cin_first) {
            for (size_t i = 0; i < per_out; ++i) w[co * per_out + i] *= s;
        } else {   // ConvTranspose2d weight is [cin][cout][kh][kw]
            const size_t k = per_out / cin;   // kh*kw
            for (int ci = 0; ci < cin; ++ci)
                for (size_t i = 0; i < k; ++i) w[(static_cast<size_t>(ci) * cout + co) * k + i] *= s;
        }
        b[co] = (b[co] - mu[co]) * s + be[co];
    }
}

// [cout][cin][kh][kw] -> [group][cin][kh][kw][CO] (zero padded), bias padded to groups*CO
PackedConv pack_conv(ake_pcnet* n, const std::vector<double>& w, const std::vector<double>& b, int cout, int cin, int kh, int kw) {
    PackedConv p;
    p.cin = cin; p.cout = cout; p.kh = kh; p.kw = kw;
    p.co = pick_co(cout);
    p.groups = (cout + p.co - 1) / p.co;
    auto& blob = n->blob;
    blob.resize(ake::align_up(blob.size(), 64));
    p.w_off = blob.size();
    blob.resize(blob.size() + static_cast<size_t>(p.groups) * cin * kh * kw * p.co, 0.f);
    for (int g = 0; g < p.groups; ++g)
        for (int ci = 0; ci < cin; ++ci)
            for (int dy = 0; dy < kh; ++dy)
                for (int dx = 0; dx < kw; ++dx)
                    for (int c = 0; c < p.co; ++c) {
                        const int co = g * p.co + c;
                        const double v = co < cout ? w[((static_cast<size_t>(co) * cin + ci) * kh + dy) * kw + dx] : 0.0;
                        blob[p.w_off + ((((static_cast<size_t>(g) * cin + ci) * kh + dy) * kw + dx) * p.co) + c] = static_cast<float>(v);
                    }
    blob.resize(ake::align_up(blob.size(), 64));
    p.b_off = blob.size();
    blob.resize(blob.size() + static_cast<size_t>(p.groups) * p.co, 0.f);
    for (int co = 0; co < cout; ++co) blob[p.b_off + co] = static_cast<float>(b[co]);
    if (kw == 7 || kw == 5 || kw == 3) {   // MFMA fragments (Toeplitz in time over kw taps)
        p.tb = cout >= 9 ? 1 : (cout >= 5 ? 2 : (cout >= 2 ? 4 : 16));
        p.ku = (p.tb + kw - 1 + 3) / 4 * 4;
        p.ntiles = (cout * p.tb + 15) / 16;
        p.nt = (p.tb == 1 && p.ntiles % 2 == 0 && kh <= 2) ? 2 : 1;   // tall kernels: B fragments dominate LDS, keep one N-tile
        const int ks = p.ku / 4;
        blob.resize(ake::align_up(blob.size(), 64));
        p.f_off = blob.size();
        const size_t frag = static_cast<size_t>(cin) * kh * ks * p.ntiles * 64;
        blob.resize(blob.size() + frag + static_cast<size_t>(ks) * p.ntiles * 64, 0.f);   // + one (ci,dy) of zero padding (prefetch)
        for (int ci = 0; ci < cin; ++ci)
            for (int dy = 0; dy < kh; ++dy)
                for (int s = 0; s < ks; ++s)
                    for (int nt = 0; nt < p.ntiles; ++nt)
                        for (int l = 0; l < 64; ++l) {
                            const int u = 4 * s + (l >> 4);
                            const int nn = nt * 16 + (l & 15);
                            const int co = nn / p.tb, tau = nn % p.tb;
                            const int dx = u - tau;
                            double v = 0.0;
                            if (co < cout && dx >= 0 && dx < kw) v = w[((static_cast<size_t>(co) * cin + ci) * kh + dy) * kw + dx];
                            blob[p.f_off + ((((static_cast<size_t>(ci) * kh + dy) * ks + s) * p.ntiles + nt) * 64) + l] = static_cast<float>(v);
                        }
    }
    return p;
}

// Data-gradient pack of a convolution: the transposed + flipped raw weights as a forward-type conv (cin' = cout, cout' = cin)
PackedConv dgrad_pack(ake_pcnet* n, const std::string& wkey, int cout, int cin, int kh, int kw) {
    const auto& w32 = T(n, wkey);
    std::vector<double> w(static_cast<size_t>(cin) * cout * kh * kw), b(cin, 0.0);
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci)
            for (int dy = 0; dy < kh; ++dy)
                for (int dx = 0; dx < kw; ++dx)
                    w[((static_cast<size_t>(ci) * cout + co) * kh + (kh - 1 - dy)) * kw + (kw - 1 - dx)] =
                        w32[((static_cast<size_t>(co) * cin + ci) * kh + dy) * kw + dx];
    return pack_conv(n, w, b, cin, cout, kh, kw);
}

PackedConv fold_pack(ake_pcnet* n, const std::string& conv_prefix, const std::string& bn_prefix, int cout, int cin, int kh, int kw) {
    std::vector<double> w, b;
    fold(n, conv_prefix + ".weight", conv_prefix + ".bias", bn_prefix, cout, static_cast<size_t>(cin) * kh * kw, false, cin, w, b);
    return pack_conv(n, w, b, cout, cin, kh, kw);
}

// --denseblock: a convolution with its raw weights (no BatchNorm behind it); a 1-wide kernel is stored as 7 taps with only the centre
// one non-zero, so that the 7-tap Toeplitz kernels run it (bkey empty: bias=False)
PackedConv dense_conv_pack_w(ake_pcnet* n, const std::vector<float>& w32, const std::vector<float>* b32p, int cout, int cin, int kh, int kw_src);
PackedConv dense_conv_pack(ake_pcnet* n, const std::string& wkey, const std::string& bkey, int cout, int cin, int kh, int kw_src) {
    return dense_conv_pack_w(n, T(n, wkey), bkey.empty() ? nullptr : &T(n, bkey), cout, cin, kh, kw_src);
}
// the data-gradient pack of such a convolution: the weights transposed (cin <-> cout) and flipped on both axes, stored the same way
PackedConv dense_dgrad_pack(ake_pcnet* n, const std::string& wkey, int cout, int cin, int kh, int kw_src) {
    const auto& w32 = T(n, wkey);
    std::vector<float> w(static_cast<size_t>(cin) * cout * kh * kw_src);
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci)
            for (int dy = 0; dy < kh; ++dy)
                for (int dx = 0; dx < kw_src; ++dx)
                    w[((static_cast<size_t>(ci) * cout + co) * kh + (kh - 1 - dy)) * kw_src + (kw_src - 1 - dx)] =
                        w32[((static_cast<size_t>(co) * cin + ci) * kh + dy) * kw_src + dx];
    return dense_conv_pack_w(n, w, nullptr, cin, cout, kh, kw_src);
}
PackedConv dense_conv_pack_w(ake_pcnet* n, const std::vector<float>& w32, const std::vector<float>* b32p, int cout, int cin, int kh, int kw_src) {
    std::vector<double> w(static_cast<size_t>(cout) * cin * kh * 7, 0.0), b(cout, 0.0);
    for (size_t r = 0; r < static_cast<size_t>(cout) * cin * kh; ++r) {
        if (kw_src == 7) for (int dx = 0; dx < 7; ++dx) w[r * 7 + dx] = w32[r * 7 + dx];
        else w[r * 7 + 3] = w32[r];
    }
    if (b32p) { const auto& b32 = *b32p; for (int co = 0; co < cout; ++co) b[co] = b32[co]; }
    if (kw_src == 1 && kh % 4 == 0) {   // 12 x 1: fragments [ci][row group g < kh / 4][ntile][64 lanes], lane = (k = row 4g + (l >> 4), n = co)
        PackedConv p;
        p.cin = cin; p.cout = cout; p.kh = kh; p.kw = 1; p.co = pick_co(cout); p.groups = (cout + p.co - 1) / p.co;
        p.tb = 1; p.ku = 4; p.ntiles = (cout + 15) / 16; p.nt = 1; p.row_k = 1;
        auto& blob = n->blob;
        blob.resize(ake::align_up(blob.size(), 64));
        p.w_off = blob.size();                                   // (no VALU-layout copy: only the MFMA kernel runs this form)
        p.b_off = blob.size();
        blob.resize(blob.size() + static_cast<size_t>(p.groups) * p.co, 0.f);
        for (int co = 0; co < cout; ++co) blob[p.b_off + co] = static_cast<float>(b[co]);
        blob.resize(ake::align_up(blob.size(), 64));
        p.f_off = blob.size();
        const int steps = kh / 4;
        blob.resize(blob.size() + static_cast<size_t>(cin + 1) * steps * p.ntiles * 64, 0.f);   // + one channel of zero padding (prefetch)
        for (int ci = 0; ci < cin; ++ci)
            for (int g = 0; g < steps; ++g)
                for (int nt = 0; nt < p.ntiles; ++nt)
                    for (int l = 0; l < 64; ++l) {
                        const int dy = 4 * g + (l >> 4), co = nt * 16 + (l & 15);
                        if (co < cout) blob[p.f_off + ((static_cast<size_t>(ci) * steps + g) * p.ntiles + nt) * 64 + l] = static_cast<float>(w32[(static_cast<size_t>(co) * cin + ci) * kh + dy]);
                    }
        return p;
    }
    return pack_conv(n, w, b, cout, cin, kh, 7);
}

// ---- launch helpers ---------------------------------------------------------------------

struct Tile {
    int R, TT, Tp, n_row_tiles, n_time_tiles, threads;
    size_t lds;
};

// Workgroup shape of the MFMA conv: every wave owns MT M-tiles (16 positions each), so a workgroup of W <= 8
// waves covers up to W*MT*16 positions = R rows x J frame groups; input channels are staged in chunks that fit
// the LDS budget.  Score = useful tile slots / issued, times the row-tile fill.
struct MTile { int W, cin_chunk; };
int device_cus();
int tiling_cus(int which);

// want_tiles > 1 (small batches): tilings with fewer (row, time) tiles than that lose score, so that the launch fills the chip.
bool choose_tile(bool fullrows, int cin, int H, int KH, int T_out, int TB, int KU, int NT, int MT, Tile* t, MTile* mt_out, int want_tiles = 1) {
    const int cap = 8 * MT * 16;
    const int T4 = (T_out + TW - 1) / TW * TW;
    double best = -1;
    Tile bt{};
    MTile bm{};
    for (int n_tt = 1; n_tt <= T4 / TW; ++n_tt) {
        const int TT = ((T4 / TW + n_tt - 1) / n_tt) * TW;
        const int n_tt_eff = (T_out + TT - 1) / TT;
        const int J = (TT + TB - 1) / TB;
        const int Tp = (TB * J - TB + KU + 3) / 4 * 4;
        if (Tp > 192) continue;                                          // the kernel's loader walks a patch row in at most 3 x 64 frames
        for (int R = fullrows ? H : 1; R <= (fullrows ? H : std::min(H, 64)); ++R) {
            if (R * J > cap) break;
            const int R_in = R + KH - 1;                                 // the row halo is always materialised
            const size_t per_ch = (static_cast<size_t>(R_in) * Tp + static_cast<size_t>(KH) * (KU / 4) * NT * 64) * sizeof(float);   // A rows + B fragments
            if (per_ch > kLdsBudget) break;
            const int max_chunk = static_cast<int>(std::min<size_t>(cin, std::min<size_t>(kLdsBudget, 48 * 1024) / per_ch));
            if (max_chunk < 1) continue;
            const int n_chunks = (cin + max_chunk - 1) / max_chunk;
            const int chunk = (cin + n_chunks - 1) / n_chunks;           // even split, no 7+1
            const int tiles = (R * J + 15) / 16;
            const int W = (tiles + MT - 1) / MT;
            const int row_tiles = (H + R - 1) / R;
            // useful positions / issued slots; mild preferences: fewer staged halo bytes, fuller workgroups
            const double useful = static_cast<double>(H) * T_out / TB;
            const double issued = static_cast<double>(row_tiles) * n_tt_eff * W * MT * 16;
            const double halo = static_cast<double>(R) / R_in * TT / (TT + KU);
            double eff = useful / issued * (0.85 + 0.15 * halo) * (0.9 + 0.1 * W / 8.0);
            if (want_tiles > 1) eff *= 0.3 + 0.7 * std::min(1.0, static_cast<double>((fullrows ? 1 : row_tiles) * n_tt_eff) / want_tiles);
            if (eff > best + 1e-9) {
                best = eff;
                bt = Tile{R, TT, Tp, fullrows ? 1 : row_tiles, n_tt_eff, W * 64, per_ch * chunk};
                bm = MTile{W, chunk};
            }
        }
    }
    if (best < 0) return false;
    *t = bt; *mt_out = bm;
    return true;
}

template <bool TRAIN>
int launch_mfma_t(const PackedConv& pc, const MfmaArgs& a, int MT, dim3 grid, dim3 block, size_t lds, hipStream_t s) {
#define AKE_MFMA(KU_, NT_, MT_) hipLaunchKernelGGL((conv_mfma_kernel<KU_, NT_, MT_, TRAIN>), grid, block, lds, s, a); return AKE_OK
    if (pc.ku == 8 && pc.nt == 1 && MT == 6) { AKE_MFMA(8, 1, 6); }
    if (pc.ku == 8 && pc.nt == 1 && MT == 4) { AKE_MFMA(8, 1, 4); }
    if (pc.ku == 8 && pc.nt == 1) { AKE_MFMA(8, 1, 3); }
    if (pc.ku == 8 && pc.nt == 2) { AKE_MFMA(8, 2, 3); }
    if (pc.ku == 12 && pc.nt == 1) { AKE_MFMA(12, 1, 3); }
    if (pc.ku == 24 && pc.nt == 1) { AKE_MFMA(24, 1, 3); }
    if (pc.ku == 4 && pc.nt == 1) { AKE_MFMA(4, 1, 3); }
    if (pc.ku == 4 && pc.nt == 2) { AKE_MFMA(4, 2, 3); }          // kernel_size 3: the genre head's (1, 3) conv
    if (pc.ku == 20 && pc.nt == 1) { AKE_MFMA(20, 1, 3); }        // kernel_size 3 / 5: the heads' one-channel last convs (16 frames per column group)
#undef AKE_MFMA
    ake::set_error("conv: no MFMA kernel for KU=%d NT=%d", pc.ku, pc.nt);
    return AKE_ERR_UNSUPPORTED;
}

// inference launches carry none of the training-mode code (pending BatchNorm on load, statistics, accumulation)
int launch_mfma(const PackedConv& pc, const MfmaArgs& a, int MT, dim3 grid, dim3 block, size_t lds, hipStream_t s) {
    const bool train = a.c.in_affine || a.c.stats || a.c.accumulate || a.c.rows_zero;      // (rows_zero: only the TRAIN form's loader zero-pads the rows)
    return train ? launch_mfma_t<true>(pc, a, MT, grid, block, lds, s) : launch_mfma_t<false>(pc, a, MT, grid, block, lds, s);
}

struct Src {
    const float* p0; int c0; const float* p1; int c1; int h1;
    int ctot0 = 0;           // channels of the buffer p0 points into when only its first c0 are read (0: c0)
};

struct ConvGeom {            // explicit geometry for the data-gradient convolutions
    int py, pad_l, T_out, H_out, time_circ;
};

// One convolution of the net.  `kind`: 0 pitch conv (7x7 circular both axes), 1 equivariant pitch-class
// conv (12 x k, rows circular), 2 genre conv (kh in {1,2}, rows valid).
int run_conv(const ake_pcnet* n, const PackedConv& pc, int kind, Src src, int batch, int H, int T_in, bool same_time,
             bool lrelu, float* dst, int dst_ctot, int dst_coff, hipStream_t s, const char* name,
             const float* in_affine = nullptr, double* stats = nullptr, const ConvGeom* geom = nullptr, bool accumulate = false,
             const float* residual = nullptr, bool rows_zero = false) {
    ConvArgs a;
    std::memset(&a, 0, sizeof(a));
    AKE_REQUIRE(pc.kw == 7 || pc.kw == 5 || pc.kw == 3 || pc.row_k, AKE_ERR_UNSUPPORTED, "conv: kernel width %d not built (3, 5, 7)", pc.kw);
    AKE_REQUIRE(src.c0 + src.c1 == pc.cin, AKE_ERR_STATE, "conv %s: cin mismatch", name);
    a.src0 = src.p0; a.c0 = src.c0; a.src1 = src.p1; a.c1 = src.c1; a.h1 = src.h1 > 0 ? src.h1 : 1;
    a.H = H; a.T_in = T_in;
    a.src0_clip_stride = static_cast<long long>(src.ctot0 > 0 ? src.ctot0 : src.c0) * H * T_in;
    a.rows_zero = rows_zero ? 1 : 0;
    a.src1_clip_stride = static_cast<long long>(src.c1) * a.h1 * T_in;
    const bool fullrows = kind != 0;
    if (kind == 0) { a.py = pc.kh / 2; a.pad_l = pc.kw / 2; a.time_circ = 1; a.T_out = T_in; a.H_out = H; }
    else {
        a.py = 0; a.time_circ = 0;
        a.pad_l = same_time ? pc.kw / 2 : 0;
        a.T_out = same_time ? T_in : T_in - pc.kw + 1;
        a.H_out = kind == 1 ? H : H - pc.kh + 1;
    }
    if (geom) { a.py = geom->py; a.pad_l = geom->pad_l; a.T_out = geom->T_out; a.H_out = geom->H_out; a.time_circ = geom->time_circ; }
    AKE_REQUIRE(a.T_out > 0, AKE_ERR_INVALID, "conv %s: %d frames is too short for the valid head convolutions", name, T_in);
    a.w = n->blob_dev + pc.w_off; a.bias = n->blob_dev + pc.b_off; a.cout = pc.cout;
    a.dst = dst; a.dst_coff = dst_coff; a.dst_clip_stride = static_cast<long long>(dst_ctot) * a.H_out * a.T_out;
    a.lrelu = lrelu ? 1 : 0;
    a.in_affine = in_affine; a.stats = stats; a.stats_stride = 2 * n->bn_channels; a.accumulate = accumulate ? 1 : 0;
    a.residual = residual; a.residual_clip_stride = static_cast<long long>(pc.cout) * a.H_out * a.T_out;      // dense [B][cout][H][T]
    Tile t;
    MTile mtile;
    static const int mt_env = ake::diag_env("AKE_MT") ? std::atoi(ake::diag_env("AKE_MT")) : 3;
    const int MT = (pc.ku == 8 && pc.nt == 1 && kind == 0) ? mt_env : 3;
    const int cout_tiles = std::max(1, pc.ntiles / pc.nt);
    static const char* const tiling_only = ake::diag_env("AKE_TILING_ONLY");      // diagnostic: AKE_TILING_CUS only for launches whose name contains this
    const int t_cus = (tiling_only && !std::strstr(name, tiling_only)) ? device_cus() : tiling_cus(0);
    const int want_tiles = (std::max(t_cus, 1) + batch * cout_tiles - 1) / (batch * cout_tiles);      // 1 at the bench / training batch sizes
    AKE_REQUIRE(choose_tile(fullrows, pc.cin, H, pc.kh, a.T_out, pc.tb, pc.ku, pc.nt, MT, &t, &mtile, want_tiles), AKE_ERR_UNSUPPORTED,
                "conv %s: no tile fits LDS (cin=%d H=%d)", name, pc.cin, H);
    a.R = t.R; a.TT = t.TT; a.Tp = t.Tp; a.n_row_tiles = t.n_row_tiles; a.n_time_tiles = t.n_time_tiles;
    a.w = n->blob_dev + pc.f_off;
    MfmaArgs ma;
    ma.c = a; ma.TB = pc.tb; ma.KH = pc.row_k ? pc.kh / 4 : pc.kh; ma.ntiles_total = pc.ntiles; ma.cin_chunk = mtile.cin_chunk;
    ma.row_k = pc.row_k;
    ma.ksplit = 0;
    ma.h1_magic = (65536 + a.h1 - 1) / a.h1;
    static const int ablate = ake::diag_env("AKE_ABLATE") ? std::atoi(ake::diag_env("AKE_ABLATE")) : 0;
    ma.dbg = ablate;
    if (mtile.W == 1 && pc.cin * pc.kh >= 16) {   // tiny M (1-channel head convs): split the (channel, dy) steps over 8 waves instead
        ma.ksplit = 1;
        mtile.W = 8;
        t.threads = 8 * 64;
        t.lds = std::max<size_t>(t.lds, static_cast<size_t>(8) * MT * pc.nt * 64 * sizeof(float) * 4);
    }
    static const bool debug = ake::diag_env("AKE_DEBUG") != nullptr;
    if (debug)
        fprintf(stderr, "[ake] %-28s cin=%3d cout=%3d kh=%2d TB=%2d KU=%2d NT=%d | R=%2d TT=%3d Tp=%3d tiles=%dx%d waves=%d chunk=%d lds=%zu grid=(%d,%d,%d)\n",
                name, pc.cin, pc.cout, pc.kh, pc.tb, pc.ku, pc.nt, t.R, t.TT, t.Tp, t.n_row_tiles, t.n_time_tiles, mtile.W,
                mtile.cin_chunk, t.lds, t.n_row_tiles * t.n_time_tiles, pc.ntiles / pc.nt, batch);
    dim3 grid(t.n_row_tiles * t.n_time_tiles, pc.ntiles / pc.nt, batch), block(t.threads);
    ake::ProfScope ps(name, s);
    return launch_mfma(pc, ma, MT, grid, block, t.lds, s);
}

// does inference run layer i's Pitch2Pitch stack on the bf16 kernel (everything but its first conv)?
// (the bf16 kernels keep all frames of their row tile in one LDS patch: long clips fall back to the time-tiled f32 kernel)
bool p2p_uses_f16(const ake_pcnet* n, int i, int T) {
    static const bool f32_only = ake::diag_env("AKE_P2P_F32") != nullptr;
    const auto& c = n->cfg;
    if (f32_only || c.precision == AKE_PRECISION_F32X3 || c.resblock || i < 1 || c.conv_layers < 2 || n->dims[i].out_p != 8 || T > 146) return false;
    for (int j = 0; j < c.conv_layers; ++j)
        if (n->p2p[i][j].bf_off < 0) return false;
    return true;
}

// does inference run layer i's PitchClass2PitchClass stack on conv_pc_bf16_kernel?
constexpr int kPcBf16MaxFrames = 120;
bool pc2pc_uses_bf16(const ake_pcnet* n, int i, int T) {
    static const bool f32_only = ake::diag_env("AKE_PC_F32") != nullptr;
    if (f32_only || n->cfg.resblock || n->pc2pc[i].empty() || T > kPcBf16MaxFrames) return false;
    for (const PackedConv& pc : n->pc2pc[i])
        if (pc.bf_off < 0 || pc.cout != 16) return false;
    return true;
}

// Row pitch of the pitch convs' LDS patch, in positions (16 B each): >= 2J + 6 (three halo frames either side), and 2J + 16 so that
// an M-tile that runs over a row end (m = r * J + j: its lanes then sit in two patch rows) still reads 16 distinct 16-byte slots per
// ds_read_b128 lane group -- the A-fragment address is r * Tp + 2j + q, a row change adds Tp - 2J, and the LDS has 16 slots per clock.
// With 2J + 6 the straddling tiles (40 % of them at J = 38) paid two-way conflicts: SQ_LDS_BANK_CONFLICT was 26-28 % of the LDS cycles
// of the three launches (profiles/r02_c_pmc_lds.md) in a multiply loop that is bound by exactly those reads.
inline int p2p_pitch(int J) { return 2 * J + 16; }

// 8 -> 8 channel 7x7 pitch convolution on bf16 MFMA (conv_p2p_f16_kernel): channels-last split planes in, planes or NCHW f32 out
int run_p2p_f16(const ake_pcnet* n, const PackedConv& pc, const unsigned short* xh, const Src* nchw, int batch, int H, int T,
                 float* dst_nchw, int dst_ctot, unsigned short* oh, hipStream_t s, const char* name) {
    P2pBfArgs a;
    std::memset(&a, 0, sizeof(a));
    a.xh = xh;
    if (nchw) { a.p = nchw->p0; a.c0 = nchw->c0; a.u = nchw->p1; a.c1 = nchw->c1; a.h1 = nchw->h1 > 0 ? nchw->h1 : 1; } a.bfrag = n->bf_frags_dev + pc.bf_off; a.bias = n->blob_dev + pc.b_off;
    a.dst = dst_nchw; a.dst_clip_stride = static_cast<long long>(dst_ctot) * H * T; a.dst_coff = 0;
    a.oh = oh;
    a.H = H; a.T = T;
    a.J = (T + 1) / 2;
    a.Tp = p2p_pitch(a.J);
    a.R = std::max(1, std::min(H, 8 * kP2pMT * 16 / a.J));
    auto lds_of = [&](int R) { return (static_cast<size_t>(R + 6) * a.Tp + (kP2pStreamB ? 0 : kBfFragsPerConv)) * sizeof(uint4); };
    while (a.R > 1 && lds_of(a.R) > 76 * 1024) --a.R;
    AKE_REQUIRE(lds_of(a.R) <= 150 * 1024, AKE_ERR_UNSUPPORTED, "conv %s: %d frames do not fit the bf16 kernel's LDS patch", name, T);
    a.n_row_tiles = (H + a.R - 1) / a.R;
    static ake::DeviceOnce attr_set;
    if (attr_set.need()) {
        AKE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_p2p_f16_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        AKE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_p2p_f16_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        AKE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_p2p_f16_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        attr_set.mark();
    }
    dim3 grid(a.n_row_tiles, 1, batch), block(512);
    ake::ProfScope ps(name, s);
    AKE_REQUIRE(!nchw || oh, AKE_ERR_STATE, "conv %s: the assembling variant writes channels-last planes", name);
    if (nchw) hipLaunchKernelGGL((conv_p2p_f16_kernel<true, true>), grid, block, lds_of(a.R), s, a);
    else if (oh) hipLaunchKernelGGL((conv_p2p_f16_kernel<true, false>), grid, block, lds_of(a.R), s, a);
    else hipLaunchKernelGGL((conv_p2p_f16_kernel<false, false>), grid, block, lds_of(a.R), s, a);
    return AKE_OK;
}

// debug switch (ake_debug_keep_taps): keep every activation that ake_pcnet_tap_* can name in memory, i.e. no fusion across them
int g_keep_taps = 0;

int device_cus() {
    static int n_cus = 0;
    if (!n_cus) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n_cus = -1;
    }
    return n_cus;
}

// The CU count the TILING heuristics compare the batch with ("does this launch fill the chip?").  Diagnostic builds can override it
// (AKE_TILING_CUS): a 32-clip batch with AKE_TILING_CUS=32 runs the tilings a 256-clip batch gets on the 256-CU part, small enough for
// float64 autograd to check them (the grids of the persistent kernels keep using the real count).
int tiling_cus(int which) {          // which: 0 the convolution's tile choice, 1 the weight-gradient kernels, 2 the semitone weight gradient
    static const int over = ake::diag_env("AKE_TILING_CUS") ? std::atoi(ake::diag_env("AKE_TILING_CUS")) : 0;
    static const int mask = ake::diag_env("AKE_TILING_CUS_MASK") ? std::atoi(ake::diag_env("AKE_TILING_CUS_MASK")) : 7;
    return (over > 0 && ((mask >> which) & 1)) ? over : device_cus();
}

// row-tile height of the persistent pitch-conv kernel for H x T maps (0: the shape does not qualify); `semi`: the form fused with the
// semitone conv (tiles of 3k rows, no staging slabs but a double-buffered output patch)
int p2p_ps_rows(int H, int T, bool semi, int* plane_pos, size_t* lds) {
    if (T < 2 || (T & 1) || device_cus() < 8) return 0;
    const int J = T / 2, Tp = p2p_pitch(J);
    auto plane_of = [&](int R) { return ((R + 6) * Tp + 63) / 64 * 64; };
    auto lds_of = [&](int R) { return (static_cast<size_t>(2) * plane_of(R) + (semi ? 2 * 8 * kP2pMT * 16 * 2 : 8 * kP2pMT * kP2pPsStage)) * sizeof(uint4); };
    int R = std::max(1, std::min(H, 8 * kP2pMT * 16 / J));
    if (semi) R = R / 3 * 3;
    while (R >= (semi ? 3 : 1) && (lds_of(R) > 156 * 1024 || plane_of(R) / 64 > 8 * kP2pPieces)) R -= semi ? 3 : 1;     // the loader: 8 waves x kP2pPieces pieces
    if (R < (semi ? 3 : 1) || H < R + 6) return 0;
    if (semi && (H % 3 || (H / 3) % 12)) return 0;
    if (plane_pos) *plane_pos = plane_of(R);
    if (lds) *lds = lds_of(R);
    return R;
}

// does inference fuse the semitone conv of layer i into the last pitch conv of its stack (the pitch tensor is then never written)?
// Only in the net's last layer: an inner layer's pitch tensor is also the next layer's pitch stream (time_pool_p, models.py:395).
bool p2p_fuses_semi(const ake_pcnet* n, int i, int P, int T) {
    static const bool off = ake::diag_env("AKE_P2P_PS") != nullptr && std::atoi(ake::diag_env("AKE_P2P_PS")) == 0;
    return !off && !g_keep_taps && !n->cfg.p2pc_conv && !n->cfg.stay_sixth && i == n->cfg.num_layers - 1 && p2p_uses_f16(n, i, T) && static_cast<size_t>(i) < n->semi.size() && n->semi[i].bf_off >= 0 &&
           p2p_ps_rows(P, T, true, nullptr, nullptr) > 0;
}

// the same convolution as a persistent launch (conv_p2p_f16_ps_kernel): one workgroup per CU walks the row tiles.  Taken for even
// frame counts and enough tiles to give every CU at least two (always when `semi_pc` asks for the fused semitone conv: `dst` then
// receives the semitone maps [clip][8][H / 3][T]); returns false when the shape does not qualify (the caller then launches
// conv_p2p_f16_kernel)
// p_frames_major: the one pitch-stream channel of `nchw` is stored [clip][T][H] (ake_pcnet_forward_frames_major_f32)
// fold_coff >= 0 (with semi_pc): the fused semitone launch also takes the maximum over the octaves and writes channels
// [fold_coff, fold_coff + 8) of the concat buffer dst_nchw [clip][dst_ctot][12][T] (OUT = 3); false when the shape does not allow it
bool run_p2p_f16_ps(const ake_pcnet* n, const PackedConv& pc, const unsigned short* xh, const Src* nchw, int batch, int H, int T, float* dst_nchw,
                     int dst_ctot, unsigned short* oh, const PackedConv* semi_pc, hipStream_t s, const char* name, bool p_frames_major = false,
                     bool dry_run = false, int fold_coff = -1, bool u_f16x4 = false) {     // u_f16x4: nchw->p1 is layer 0's f16 x 4 form of the up_sixth map (Layer0Args::psix_h)
    static const bool off = ake::diag_env("AKE_P2P_PS") != nullptr && std::atoi(ake::diag_env("AKE_P2P_PS")) == 0;
    if (off) return false;
    P2pPsArgs a;
    std::memset(&a, 0, sizeof(a));
    size_t lds = 0;
    a.R = p2p_ps_rows(H, T, semi_pc != nullptr, &a.plane_pos, &lds);
    if (a.R <= 0) return false;
    const int n_cus = device_cus();
    a.xh = xh; a.bfrag = n->bf_frags_dev + pc.bf_off; a.bias = n->blob_dev + pc.b_off;
    if (nchw) {
        if (dst_nchw || nchw->c0 < 1 || nchw->c0 + nchw->c1 > 8) return false;
        a.p = nchw->p0; a.c0 = nchw->c0; a.u = nchw->p1 ? nchw->p1 : nchw->p0; a.c1 = nchw->p1 ? nchw->c1 : 0; a.h1 = nchw->h1 > 0 ? nchw->h1 : 1;
        if (p_frames_major && nchw->c0 != 1) return false;
        a.p_fm = p_frames_major ? 1 : 0;
        if (u_f16x4) {
            if (nchw->c0 != 1 || !nchw->p1 || nchw->c1 > 4) return false;
            a.uh = reinterpret_cast<const uint2*>(nchw->p1);
            a.ph = reinterpret_cast<const unsigned int*>(nchw->p0);
            a.p_fm = 0;
        }
    } else if (p_frames_major || u_f16x4) return false;
    a.dst = dst_nchw; a.dst_clip_stride = static_cast<long long>(dst_ctot) * (semi_pc ? H / 3 : H) * T; a.oh = oh;
    if (semi_pc) {
        if (!dst_nchw || semi_pc->bf_off < 0) return false;
        a.sfrag = n->bf_frags_dev + semi_pc->bf_off; a.sbias = n->blob_dev + semi_pc->b_off;
    }
    a.H = H; a.T = T; a.J = T / 2; a.Tp = p2p_pitch(a.J);
    a.n_row_tiles = (H + a.R - 1) / a.R;
    a.n_tiles = a.n_row_tiles * batch;
    static const bool fold_off = ake::diag_env("AKE_P2P_FOLD") != nullptr && std::atoi(ake::diag_env("AKE_P2P_FOLD")) == 0;
    const bool fold = fold_coff >= 0;
    if (fold && fold_off) return false;
    if (fold) {
        if (!semi_pc || H % 36 || 36 % a.R) return false;
        a.n_oct = H / 36; a.n_units = batch * (36 / a.R);
        if (a.n_units < 4 * n_cus) return false;                      // a unit is n_oct tiles in a row: small batches keep the tile-parallel form
        a.dst = dst_nchw + static_cast<long long>(fold_coff) * 12 * T;
        a.dst_clip_stride = static_cast<long long>(dst_ctot) * 12 * T;
    }
    if (!semi_pc && a.n_tiles < 2 * n_cus) return false;
    if (dst_nchw && !semi_pc) {   // 16-byte stores of 4 consecutive frames
        if ((a.R * T) % 4 || (static_cast<long long>(H) * T) % 4 || a.dst_clip_stride % 4 || (reinterpret_cast<uintptr_t>(dst_nchw) & 15)) return false;
    }
    static ake::DeviceOnce attr_set;
    if (attr_set.need()) {
        const void* fns[] = {reinterpret_cast<const void*>(conv_p2p_f16_ps_kernel<3, 0>),
                             reinterpret_cast<const void*>(conv_p2p_f16_ps_kernel<1, 0>), reinterpret_cast<const void*>(conv_p2p_f16_ps_kernel<1, 5>),
                             reinterpret_cast<const void*>(conv_p2p_f16_ps_kernel<1, 8>), reinterpret_cast<const void*>(conv_p2p_f16_ps_kernel<0, 0>),
                             reinterpret_cast<const void*>(conv_p2p_f16_ps_kernel<2, 0>), reinterpret_cast<const void*>(conv_p2p_f16_ps_kernel<1, 0, true>),
                             reinterpret_cast<const void*>(conv_p2p_f16_ps_kernel<1, 3>)};
        for (const void* f : fns)
            if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return false;
        attr_set.mark();
    }
    // two workgroups per CU when the LDS allows it (the kernel is built for 4 waves per SIMD): one's epilogue (vector work) and
    // barrier waits run under the other's multiply loop (bound by its LDS reads)
    if (dry_run) return true;                          // (the eligibility question of ake_pcnet_accepts_frames_major)
    static const int wg_per_cu_env = ake::diag_env("AKE_P2P_WG_PER_CU") ? std::atoi(ake::diag_env("AKE_P2P_WG_PER_CU")) : 2;
    const int wg_per_cu = (wg_per_cu_env >= 2 && (!nchw || a.uh) && lds <= 80 * 1024 && a.n_tiles >= 4 * n_cus) ? 2 : 1;
    dim3 grid(std::min(wg_per_cu * (n_cus / 8 * 8), (a.n_tiles + 7) / 8 * 8)), block(512);
    ake::ProfScope ps(name, s);
    if (fold) hipLaunchKernelGGL((conv_p2p_f16_ps_kernel<3, 0>), grid, block, lds, s, a);
    else if (semi_pc) hipLaunchKernelGGL((conv_p2p_f16_ps_kernel<2, 0>), grid, block, lds, s, a);
    else if (dst_nchw) hipLaunchKernelGGL((conv_p2p_f16_ps_kernel<0, 0>), grid, block, lds, s, a);
    else if (nchw && a.uh) hipLaunchKernelGGL((conv_p2p_f16_ps_kernel<1, 3>), grid, block, lds, s, a);
    else if (nchw && a.c0 + a.c1 <= 5) hipLaunchKernelGGL((conv_p2p_f16_ps_kernel<1, 5>), grid, block, lds, s, a);
    else if (nchw) hipLaunchKernelGGL((conv_p2p_f16_ps_kernel<1, 8>), grid, block, lds, s, a);
    else {
        static const bool stamp_env = ake::diag_env("AKE_P2P_STAMP") != nullptr;
        unsigned long long* sb = nullptr;
        if (stamp_env && hipMalloc(&sb, 64 * sizeof(unsigned long long)) == hipSuccess) {
            // diagnostic build: in-kernel cycle stamps of the tile loop's sections (workgroup 0), printed to stderr; never timed
            (void)hipMemsetAsync(sb, 0, 64 * sizeof(unsigned long long), s);
            a.stamps = sb;
            hipLaunchKernelGGL((conv_p2p_f16_ps_kernel<1, 0, true>), grid, block, lds, s, a);
            unsigned long long hb[64];
            (void)hipMemcpyAsync(hb, sb, sizeof(hb), hipMemcpyDeviceToHost, s);
            (void)hipStreamSynchronize(s);
            (void)hipFree(sb);
            for (int wv = 0; wv < 8; ++wv)
                fprintf(stderr, "p2p stamps wave %d: tiles %llu  cycles/tile: vmcnt-wait %.0f barrier %.0f late-epilogue %.0f multiply(+dma+stores) %.0f epilogue %.0f\n", wv,
                        hb[wv * 8 + 5], hb[wv * 8 + 0] / double(hb[wv * 8 + 5]), hb[wv * 8 + 1] / double(hb[wv * 8 + 5]), hb[wv * 8 + 2] / double(hb[wv * 8 + 5]),
                        hb[wv * 8 + 3] / double(hb[wv * 8 + 5]), hb[wv * 8 + 4] / double(hb[wv * 8 + 5]));
        } else hipLaunchKernelGGL((conv_p2p_f16_ps_kernel<1, 0>), grid, block, lds, s, a);
    }
    return true;
}

// 7 x 7 circular pitch convolution with f32-equivalent products (conv_p2p_f16x3_kernel): train-mode forward (in_aff, bias, statistics)
// and data gradient (none of them).  false when the shape does not qualify: the caller then runs conv_mfma_kernel.
bool run_p2p_f16x3(const ake_pcnet* n, long long frag_off, const Src& src, const float* in_aff, const float* bias, int batch, int H, int T, float* dst,
                   int cout, double* stats, int stats_stride, hipStream_t s, const char* name, const unsigned int* in_amax = nullptr, bool lrelu = false) {
    static const bool off = ake::diag_env("AKE_P2P_TRAIN_F32") != nullptr;
    if (off || frag_off < 0 || T < 2 || (T & 1) || src.c0 < 1 || src.c0 + src.c1 > 8 || cout > 8 || src.ctot0 != 0) return false;
    P2pTrArgs a;
    std::memset(&a, 0, sizeof(a));
    const int n_cus = device_cus();
    if (n_cus < 8) return false;
    a.J = T / 2; a.Tp = p2p_pitch(a.J);
    auto plane_of = [&](int R) { return ((R + 6) * a.Tp + 63) / 64 * 64; };
    auto lds_of = [&](int R) { return (static_cast<size_t>(4) * plane_of(R) + 8 * kP2pMT * kP2pPsStage) * sizeof(uint4); };
    int R = std::max(1, std::min(H, 8 * kP2pMT * 16 / a.J));
    while (R >= 1 && (lds_of(R) > 156 * 1024 || plane_of(R) > 3 * 512)) --R;         // (the loader: three positions per thread)
    if (R < 1 || H < R + 6) return false;
    a.R = R; a.plane_pos = plane_of(R);
    a.H = H; a.T = T;
    a.n_row_tiles = (H + R - 1) / R;
    a.n_tiles = a.n_row_tiles * batch;
    const long long clip_stride = static_cast<long long>(cout) * H * T;
    if ((R * T) % 4 || (static_cast<long long>(H) * T) % 4 || clip_stride % 4 || (reinterpret_cast<uintptr_t>(dst) & 15)) return false;
    a.p = src.p0; a.c0 = src.c0; a.u = src.p1 ? src.p1 : src.p0; a.c1 = src.p1 ? src.c1 : 0; a.h1 = src.h1 > 0 ? src.h1 : 1;
    a.in_aff = in_aff; a.bfrag = n->bf_frags_dev + frag_off; a.bias = bias;
    a.dst = dst; a.dst_clip_stride = clip_stride; a.cout = cout;
    a.stats = stats; a.stats_stride = stats_stride; a.in_amax = in_amax; a.lrelu = lrelu ? 1 : 0;
    static ake::DeviceOnce attr_set;
    if (attr_set.need()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_p2p_f16x3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return false;
        attr_set.mark();
    }
    dim3 grid(std::min(n_cus / 8 * 8, (a.n_tiles + 7) / 8 * 8)), block(512);
    ake::ProfScope ps(name, s);
    hipLaunchKernelGGL(conv_p2p_f16x3_kernel, grid, block, lds_of(R), s, a);
    return true;
}

// semitone maps [clip][C][S][T] -> channels [coff, coff + C) of the concat buffer [clip][ctot][12][T]: max over the octaves
int run_fold_max(const float* smap, int C, int S, int batch, int T, float* dst, int dst_ctot, int dst_coff, hipStream_t s) {
    const long long total = static_cast<long long>(batch) * C * 12 * T;
    ake::ProfScope ps("fold_max_kernel", s);
    hipLaunchKernelGGL(fold_max_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, s, smap, C, S, T, dst,
                       static_cast<long long>(dst_ctot) * 12 * T, dst_coff, total);
    return AKE_OK;
}

static const bool g_pc_f32_only = ake::diag_env("AKE_PC_F32") != nullptr;

// NCHW f32 [clip][C][12][T] -> channels-last split planes [clip][12][T][16] (hi plane, then lo plane, at `planes`)
int run_nchw_to_cl16(const float* src, int C, int batch, int T, unsigned short* planes, hipStream_t s) {
    const long long npos = static_cast<long long>(batch) * 12 * T;
    ake::ProfScope ps("nchw_to_cl16_kernel", s);
    hipLaunchKernelGGL(nchw_to_cl16_kernel, dim3(static_cast<unsigned>((npos + 255) / 256)), dim3(256), 0, s, src, static_cast<long long>(C) * 12 * T, C, T,
                       planes, planes + npos * 16, npos);
    return AKE_OK;
}

// does inference run the last layer's pitch-class stack + its time pooling as ONE launch (pc2pc_fused_kernel)?  The stack's
// intermediate activations then never leave LDS (ake_debug_keep_taps(1) keeps the per-conv launches for bisecting).
bool pc2pc_fuses(const ake_pcnet* n, int i, int T) {
    static const bool off = ake::diag_env("AKE_PC_FUSED") != nullptr && std::atoi(ake::diag_env("AKE_PC_FUSED")) == 0;
    const auto& c = n->cfg;
    if (off || g_keep_taps || i < 1 || i != c.num_layers - 1 || c.time_pool_size != 2 || c.conv_layers < 1 || c.conv_layers > 4) return false;
    if (!pc2pc_uses_bf16(n, i, T) || T % 4 || 12 * T > 1024 || 3 * ((T + 15) / 16) > 16) return false;   // (16 waves: 3 row groups x 16-frame tiles)
    for (const PackedConv& pc : n->pc2pc[i])
        if (pc.cout != 16 || pc.cin > 16 || pc.kh != 12) return false;
    return (static_cast<size_t>(4) * 12 * (T + 8) * 2 + 2 * 512) * sizeof(uint4) <= 160 * 1024;     // two maps + the weight ring
}

int run_pc2pc_fused(const ake_pcnet* n, int i, const float* src, int cin, int batch, int T, float* pooled, unsigned short* feat_cl, hipStream_t s) {
    Pc2pcFusedArgs a;
    std::memset(&a, 0, sizeof(a));
    a.src = src; a.src_clip_stride = static_cast<long long>(cin) * 12 * T; a.cin = cin;
    a.n_conv = static_cast<int>(n->pc2pc[i].size());
    for (int j = 0; j < a.n_conv; ++j) { a.bfrag[j] = n->bf_frags_dev + n->pc2pc[i][j].bf_off; a.bias[j] = n->blob_dev + n->pc2pc[i][j].b_off; }
    a.pooled = pooled;
    if (feat_cl) { a.fh = feat_cl; a.fl = feat_cl + static_cast<long long>(batch) * 12 * (T / 2) * 16; }
    a.T = T; a.Tp = T + 8;
    const size_t lds = (static_cast<size_t>(4) * 12 * a.Tp * 2 + 2 * 512) * sizeof(uint4);
    static ake::DeviceOnce attr_set;
    if (attr_set.need()) {
        AKE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(pc2pc_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set.mark();
    }
    ake::ProfScope ps("pc2pc_fused_kernel", s);
    hipLaunchKernelGGL(pc2pc_fused_kernel, dim3(batch), dim3(1024), lds, s, a);
    return AKE_OK;
}

// does the f16 x 3 training form of conv_pc_bf16_kernel take this convolution (fragments built, patch fits the LDS)?
bool pc_f16x3_ok(const PackedConv& pt, int T_in, bool same_time) {
    static const bool off = ake::diag_env("AKE_PC_TRAIN_F32") != nullptr;
    if (off || pt.bf_off < 0 || pt.cin > 16) return false;
    const int T_out = same_time ? T_in : T_in - pt.kw + 1;
    if (T_out < 1) return false;
    const size_t lds = (static_cast<size_t>(2) * 12 * (T_out + 8) * 2 + 2 * 4 * ((pt.cout + 15) / 16) * 2 * 64) * sizeof(uint4);
    return lds <= 150 * 1024;
}

// training: NCHW f32 (+ the producer's pending BatchNorm + LeakyReLU) -> f16 hi / lo * 2^11 planes [hi: batch * 12 * T * 16][lo: ...]
// (src_ctot > 0: src points at the first of C channels inside a tensor of src_ctot channels)
// amax (data gradients): the bits of the tensor's largest |value|; the planes then hold value * f16_weight_scale(max) and the convolution
// that reads them (PcBfArgs::in_amax) divides it out again
void run_nchw_to_cl16_f16x2(const float* src, int C, int batch, int T, const float* aff, unsigned short* planes, hipStream_t s, int src_ctot = 0,
                            const unsigned int* amax = nullptr) {
    const long long npos = static_cast<long long>(batch) * 12 * T;
    ake::ProfScope ps("nchw_to_cl16_f16x2_kernel", s);
    hipLaunchKernelGGL(nchw_to_cl16_f16x2_kernel, dim3(static_cast<unsigned>((npos + 255) / 256)), dim3(256), 0, s, src, static_cast<long long>(src_ctot > 0 ? src_ctot : C) * 12 * T, C, T,
                       aff, planes, planes + npos * 16, npos, amax);
}

// pitch-class convolution on bf16 MFMA (conv_pc_bf16_kernel): channels-last planes in; planes (cout == 16) or NCHW f32 out
int run_pc_bf16(const ake_pcnet* n, const PackedConv& pc, const unsigned short* planes_in, int batch, int T_in, bool same_time, bool lrelu,
                float* dst_nchw, unsigned short* planes_out, hipStream_t s, const char* name, const PackedConv* pc2 = nullptr,
                unsigned short* planes_out2 = nullptr, bool f16x3 = false, double* stats = nullptr, int stats_stride = 0,
                const unsigned int* in_amax = nullptr) {
    PcBfArgs a;
    std::memset(&a, 0, sizeof(a));
    const long long npos_in = static_cast<long long>(batch) * 12 * T_in;
    a.xh = planes_in; a.xl = planes_in + npos_in * 16;
    a.bfrag = n->bf_frags_dev + pc.bf_off; a.bias = n->blob_dev + pc.b_off;
    a.T_in = T_in; a.T_out = same_time ? T_in : T_in - pc.kw + 1; a.pad_l = same_time ? pc.kw / 2 : 0;
    AKE_REQUIRE(a.T_out > 0, AKE_ERR_INVALID, "conv %s: %d frames is too short for the valid head convolutions", name, T_in);
    a.Tp = a.T_out + 8;
    a.cout = pc.cout; a.lrelu = lrelu ? 1 : 0;
    a.KH = pc.kh; a.circular = pc.kh == 12 ? 1 : 0;
    const int H_out = a.circular ? 12 : 12 - pc.kh + 1;
    a.dst = dst_nchw; a.dst_clip_stride = static_cast<long long>(pc.cout) * H_out * a.T_out;
    a.cl_stride = pc.cout;
    if (planes_out) { a.oh = planes_out; a.ol = planes_out + static_cast<long long>(batch) * H_out * a.T_out * pc.cout; }
    if (pc2) {   // a second convolution of the same geometry over the same input, as blockIdx.y == 1 (planes out only)
        AKE_REQUIRE(planes_out && planes_out2 && pc2->cout == pc.cout && pc2->kh == pc.kh && pc2->kw == pc.kw && pc2->cin == pc.cin && pc2->bf_off >= 0,
                    AKE_ERR_STATE, "conv %s: the paired convolution differs in shape", name);
        a.bfrag2 = n->bf_frags_dev + pc2->bf_off; a.bias2 = n->blob_dev + pc2->b_off;
        a.oh2 = planes_out2; a.ol2 = planes_out2 + static_cast<long long>(batch) * H_out * a.T_out * pc.cout;
    }
    a.stats = stats; a.stats_stride = stats_stride; a.in_amax = in_amax;
    AKE_REQUIRE(!f16x3 || (dst_nchw && !planes_out && !pc2), AKE_ERR_STATE, "conv %s: the f16 x 3 form writes NCHW f32", name);
    const size_t lds = (static_cast<size_t>(2) * 12 * a.Tp * 2 + 2 * 4 * ((pc.cout + 15) / 16) * 2 * 64) * sizeof(uint4);   // patch + weight ring
    AKE_REQUIRE(lds <= 150 * 1024, AKE_ERR_UNSUPPORTED, "conv %s: %d frames do not fit the bf16 kernel's LDS patch", name, T_in);
    static ake::DeviceOnce attr_set;
    if (attr_set.need()) {
        AKE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_pc_bf16_kernel<1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        AKE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_pc_bf16_kernel<1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        AKE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_pc_bf16_kernel<2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        AKE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_pc_bf16_kernel<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        AKE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_pc_bf16_kernel<1, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        AKE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_pc_bf16_kernel<2, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        attr_set.mark();
    }
    const int tiles = (H_out * a.T_out + 15) / 16;
    const int waves = std::min(8, (tiles + 3) / 4);
    dim3 grid((tiles + waves * 4 - 1) / (waves * 4), pc2 ? 2 : 1, batch), block(waves * 64);
    ake::ProfScope ps(name, s);
    if (f16x3 && pc.cout <= 16) hipLaunchKernelGGL((conv_pc_bf16_kernel<1, false, true>), grid, block, lds, s, a);
    else if (f16x3) hipLaunchKernelGGL((conv_pc_bf16_kernel<2, false, true>), grid, block, lds, s, a);
    else if (pc.cout == 16 && planes_out) hipLaunchKernelGGL((conv_pc_bf16_kernel<1, true>), grid, block, lds, s, a);
    else if (pc.cout == 16) hipLaunchKernelGGL((conv_pc_bf16_kernel<1, false>), grid, block, lds, s, a);
    else if (planes_out) hipLaunchKernelGGL((conv_pc_bf16_kernel<2, true>), grid, block, lds, s, a);
    else hipLaunchKernelGGL((conv_pc_bf16_kernel<2, false>), grid, block, lds, s, a);
    return AKE_OK;
}

// f16 x 3 data gradient of a "valid" head convolution (32 -> 16 channels seen from the gradient): full correlation (pad 6 on both sides,
// T_out = T_dz + 6) of 16 gradient channels (planes) with one half of the transposed + flipped weights; dst [clip][16][12][T_out] (+)=.
int run_pc_f16x3_full(const ake_pcnet* n, long long frag_off, int kh, const unsigned short* planes, int batch, int T_dz, float* dst, bool accumulate,
                      hipStream_t s, const char* name, const unsigned int* in_amax = nullptr) {
    PcBfArgs a;
    std::memset(&a, 0, sizeof(a));
    const long long npos_in = static_cast<long long>(batch) * 12 * T_dz;
    a.xh = planes; a.xl = planes + npos_in * 16;
    a.bfrag = n->bf_frags_dev + frag_off; a.bias = nullptr;
    a.T_in = T_dz; a.T_out = T_dz + 6; a.pad_l = 6; a.Tp = a.T_out + 8;
    a.cout = 16; a.lrelu = 0; a.KH = kh; a.circular = kh == 12 ? 1 : 0;
    AKE_REQUIRE(kh == 12 || kh == 1, AKE_ERR_STATE, "conv %s: kernel rows %d", name, kh);
    a.dst = dst; a.dst_clip_stride = static_cast<long long>(16) * 12 * a.T_out; a.cl_stride = 16;
    a.accumulate = accumulate ? 1 : 0; a.in_amax = in_amax;
    const size_t lds = (static_cast<size_t>(2) * 12 * a.Tp * 2 + 2 * 4 * 2 * 64) * sizeof(uint4);
    AKE_REQUIRE(lds <= 150 * 1024, AKE_ERR_UNSUPPORTED, "conv %s: %d frames do not fit the kernel's LDS patch", name, T_dz);
    static ake::DeviceOnce attr_set;
    if (attr_set.need()) {
        AKE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_pc_bf16_kernel<1, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        attr_set.mark();
    }
    const int tiles = (12 * a.T_out + 15) / 16;
    const int waves = std::min(8, (tiles + 3) / 4);
    dim3 grid((tiles + waves * 4 - 1) / (waves * 4), 1, batch), block(waves * 64);
    ake::ProfScope ps(name, s);
    hipLaunchKernelGGL((conv_pc_bf16_kernel<1, false, true>), grid, block, lds, s, a);
    return AKE_OK;
}

int run_semi(const ake_pcnet* n, const PackedConv& pc, const float* src, int batch, int H, int Tn, float* dst,
             int dst_ctot, int dst_coff, hipStream_t s, const char* name) {
    SemiArgs a;
    a.src = src; a.C = pc.cin; a.H = H; a.T = Tn;
    a.src_clip_stride = static_cast<long long>(pc.cin) * H * Tn;
    a.w = n->blob_dev + pc.w_off; a.bias = n->blob_dev + pc.b_off;
    a.dst = dst; a.dst_coff = dst_coff; a.dst_clip_stride = static_cast<long long>(dst_ctot) * 12 * Tn;
    a.n_strips = (Tn + TW - 1) / TW;
    const int per_clip = 12 * a.n_strips;
    const int threads = per_clip >= 256 ? 256 : (per_clip + 63) / 64 * 64;
    dim3 grid((per_clip + threads - 1) / threads, pc.groups, batch), block(threads);
    ake::ProfScope ps(name, s);
    switch (pc.co) {
        case 8: hipLaunchKernelGGL((semi_fold_kernel<8>), grid, block, 0, s, a); break;
        case 4: hipLaunchKernelGGL((semi_fold_kernel<4>), grid, block, 0, s, a); break;
        case 1: hipLaunchKernelGGL((semi_fold_kernel<1>), grid, block, 0, s, a); break;
        default: ake::set_error("semi: bad CO"); return AKE_ERR_UNSUPPORTED;
    }
    return AKE_OK;
}

struct Buffers {           // workspace carve
    // per chunk (pitch stream): everything up to the last layer's semitone fold
    float* fold0 = nullptr;
    unsigned int* melh = nullptr;   // inference: the log-CQT as f16 hi | lo words, [batch][P][T] (layer0_mfma_kernel -> conv_p2p_f16_ps_kernel<1, 3>)
    float* smap = nullptr;     // --p2pc_conv only
    float* p0 = nullptr;       // --stay_sixth only
    std::vector<float*> pcd;   // --stay_sixth only
    std::vector<float*> cat, psix, pa, pb, pca, pcb, ppool, pin;
    // whole batch (pitch-class tail): last layer's concat buffer, its pc stack, pooled features, heads
    float *pcf = nullptr, *hid_k = nullptr, *hid_t = nullptr, *hid_g = nullptr;
    float *map_k = nullptr, *map_t = nullptr, *map_g = nullptr;
    std::vector<int> Tl;    // frames at layer i
    int Tf = 0;             // frames after the last layer
    size_t bytes = 0;
    // training-mode forward only: BatchNorm statistics, raw semitone-conv outputs and the pending-affine table
    // ([C][3] = scale, shift, negative slope) of every buffer that can hold a raw (pre-BatchNorm) tensor
    double* stats = nullptr;           // [kStatSlots][bn_channels][2]
    float* bstats = nullptr;           // [bn_channels][3] batch mean, biased variance, element count
    std::vector<float*> semi_raw, aff_semi, aff_cat, aff_p2pin;
    std::vector<float*> foldc_raw, aff_foldc, g_foldc;                // --p2pc_conv training: raw fold-conv output [B][C][12][T], pool.bn's table, its gradient
    // every convolution keeps its own raw output in training mode (the backward pass needs all of them)
    std::vector<std::vector<float*>> pst, aff_pst, pcst, aff_pcst;   // [layer][conv]
    std::vector<float*> hst[3], aff_hst[3];                            // [head][hidden conv]
    // backward
    double* stats2 = nullptr;          // [bn_channels][3]  (sum g1, sum g1*zhat, sum (z - mean)); then one 8-byte cell per BatchNorm layer: the bits of max |dz| (dz_amax)
    gfx_t* gslots = nullptr;           // [kGradSlots][grad_floats] partial weight gradients (backward), 64-bit fixed point
    float* wg_partial = nullptr;       // per-workgroup partial weight gradients of one convolution (conv_wgrad_kernel), reduced in order
    size_t wg_partial_floats = 0;
    unsigned short* feat_cl = nullptr; // [2 planes][B][12][Tf][16]
    float* coef = nullptr;             // [bn_channels][4]  (c0, c1, c2, mean)
    float* g_map[3] = {nullptr, nullptr, nullptr};
    float *g_hid = nullptr, *g_pcf = nullptr, *g_fold0 = nullptr;
    std::vector<float*> g_pc, g_cat, g_semi, g_p, g_pin, g_psix;      // per layer (g_pc / g_p: ping-pong pair packed as 2x)
    // --denseblock training: per block [pitch-class block of layer i | pitch block of layer i] and dense layer: the bottleneck map (kept for
    // the backward pass) and the two on-load tables (norm1 over the block's features so far, norm2 over the bottleneck map); one gradient
    // buffer of the widest bottleneck per block kind
    std::vector<std::vector<float*>> dn_bott_pc, dn_aff1_pc, dn_aff2_pc, dn_bott_p, dn_aff1_p, dn_aff2_p;
    float *dn_gbott_pc = nullptr, *dn_gbott_p = nullptr;
};

// chunk = clips per pitch-stream pass, batch = clips of the call (tail buffers)
int plan_buffers(const ake_pcnet* n, int batch, int chunk, int frames, void* ws, Buffers* b, bool train = false) {
    const auto& c = n->cfg;
    const int L = c.num_layers, P = c.pitches;
    ake::Carver cv(ws, 0);
    b->Tl.assign(L, frames);
    for (int i = 2; i < L; ++i) b->Tl[i] = b->Tl[i - 1] / c.time_pool_size;
    b->Tf = L > 1 ? b->Tl[L - 1] / c.time_pool_size : frames;
    AKE_REQUIRE(b->Tf >= 1, AKE_ERR_INVALID, "pcnet: %d frames vanish under the time pooling", frames);
    b->cat.assign(L + 1, nullptr); b->psix.assign(L, nullptr); b->pa.assign(L, nullptr); b->pb.assign(L, nullptr);
    b->pca.assign(L, nullptr); b->pcb.assign(L, nullptr); b->ppool.assign(L, nullptr); b->pin.assign(L, nullptr);
    const size_t C = chunk, B = batch;
    const size_t dg = c.denseblock ? static_cast<size_t>(c.n_filters) * c.conv_layers : 0;   // channels a dense block appends in place
    b->fold0 = cv.take<float>(B * (L == 1 ? 1 + dg : 1) * 12 * frames);
    if (!train && L > 1) b->melh = cv.take<unsigned int>(B * P * frames);
    if (c.stay_sixth) {  // --stay_sixth: layer 0's activated semitone map is the pitch stream; dense copies of the pitch-class stream for the repeat
        b->p0 = cv.take<float>(B * (P / 3) * frames);
        b->pcd.assign(L, nullptr);
        for (int i = 1; i < L; ++i) b->pcd[i] = cv.take<float>(C * n->dims[i].prev_pc * 12 * b->Tl[i]);
    }
    if (c.p2pc_conv) {   // --p2pc_conv: raw semitone maps between the semitone conv and the octave-fold conv (largest layer)
        size_t m = B * (P / 3) * frames;
        for (int i = 1; i < L; ++i) m = std::max(m, C * n->dims[i].out_p * (P / 3) * b->Tl[i]);
        b->smap = cv.take<float>(m);
    }
    for (int i = 0; i < L; ++i) {
        const int Ti = b->Tl[i];
        const auto& d = n->dims[i];
        const bool last = i == L - 1;
        const int pc_out = i == 0 ? c.n_filters : d.out_pc;
        if (i >= 1) {
            b->cat[i] = cv.take<float>((last || i == 1 ? B : C) * (d.prev_pc + d.out_p + dg) * 12 * Ti);
            b->psix[i] = cv.take<float>((i == 1 ? B : C) * d.prev_pc * 36 * Ti);
            if (c.pc2p_mem) b->pin[i] = cv.take<float>(C * d.prev_p * P * Ti);             // --pc2p_mem: pitch stream + summed up_sixth map
            b->pa[i] = cv.take<float>(C * d.out_p * P * Ti);
            b->pb[i] = cv.take<float>((c.resblock ? 2 : 1) * C * (c.denseblock ? static_cast<size_t>((d.prev_pc + d.prev_p) / 2) * c.n_filters : d.out_p) * P * Ti);   // --resblock: the blocks' 2C-channel hidden map; --denseblock: the bottleneck map
            if (!last) b->ppool[i] = cv.take<float>(C * d.out_p * P * (Ti / c.time_pool_size));
        }
        const size_t pca_ch = (c.denseblock && i >= 1) ? static_cast<size_t>((d.out_p + d.prev_pc) / 2) * c.n_filters : pc_out;   // --denseblock: bottleneck
        b->pca[i] = cv.take<float>((last || i == 0 ? B : C) * pca_ch * 12 * Ti);
        b->pcb[i] = cv.take<float>((c.resblock ? 2 : 1) * (last || i == 0 ? B : C) * pc_out * 12 * Ti);
    }
    b->pcf = cv.take<float>(B * n->final_ch * 12 * b->Tf);
    const size_t hid = B * 2 * n->final_ch * 12 * b->Tf;
    b->hid_k = cv.take<float>(2 * hid); b->hid_t = cv.take<float>(2 * hid);
    b->feat_cl = cv.take<unsigned short>(B * 12 * b->Tf * 16 * 2);     // channels-last split copy of the head input (bf16 head convs)
    b->map_k = cv.take<float>(B * 12 * b->Tf); b->map_t = cv.take<float>(B * 12 * b->Tf);
    if (c.genre) { b->hid_g = cv.take<float>(2 * hid); b->map_g = cv.take<float>(B * 12 * b->Tf); }
    if (train) {
        b->stats = cv.take<double>(static_cast<size_t>(n->bn_channels) * 2 * kStatSlots);
        b->stats2 = cv.take<double>(static_cast<size_t>(n->bn_channels) * 3 * kBwdStatSlots + n->bns.size() * (kAmaxSlots / 2));       // [slot][bn_channels][3], then [bn layer][kAmaxSlots] unsigned: partial maxima of |dz| (float bits)
        b->gslots = cv.take<gfx_t>(static_cast<size_t>(kGradSlots) * n->grad_floats);
        b->wg_partial_floats = static_cast<size_t>(B) * 98304;                         // 384 KB per clip: e.g. two workgroups per clip x the 43 K weights of a head conv
        b->wg_partial = cv.take<float>(b->wg_partial_floats);
        b->bstats = cv.take<float>(static_cast<size_t>(n->bn_channels) * 3);
        b->coef = cv.take<float>(static_cast<size_t>(n->bn_channels) * 4);
        for (auto* v : {&b->semi_raw, &b->aff_semi, &b->aff_cat, &b->aff_p2pin, &b->g_pc, &b->g_cat, &b->g_semi, &b->g_p, &b->g_pin, &b->g_psix,
                        &b->foldc_raw, &b->aff_foldc, &b->g_foldc})
            v->assign(L + 1, nullptr);
        b->pst.assign(L, {}); b->aff_pst.assign(L, {}); b->pcst.assign(L, {}); b->aff_pcst.assign(L, {});
        for (int i = 0; i < L; ++i) {
            const auto& d = n->dims[i];
            const int Ti = b->Tl[i];
            const int cs = i == 0 ? 1 : d.out_p, pc_out = i == 0 ? c.n_filters : d.out_pc;
            b->semi_raw[i] = cv.take<float>(B * cs * (P / 3) * Ti);
            b->g_semi[i] = cv.take<float>(B * cs * (P / 3) * Ti);
            if (c.p2pc_conv) {
                b->foldc_raw[i] = cv.take<float>(B * cs * 12 * Ti);
                b->g_foldc[i] = cv.take<float>(B * cs * 12 * Ti);
                b->aff_foldc[i] = cv.take<float>(3 * cs);
            }
            b->aff_semi[i] = cv.take<float>(3 * cs);
            // --resblock: [conv0, (conv1 (2C), conv2, block output) per block] (res_stack_train); gradients: g, skip copy, 2C hidden map
            const int n_st = c.denseblock ? 0 : (c.resblock ? 1 + 3 * c.conv_layers : c.conv_layers);
            auto width = [&](int j) { return c.resblock && j % 3 == 1 ? 2 : 1; };
            for (int j = 0; j < n_st; ++j) {
                b->pcst[i].push_back(cv.take<float>(width(j) * B * pc_out * 12 * Ti));
                b->aff_pcst[i].push_back(cv.take<float>(3 * width(j) * pc_out));
            }
            b->g_pc[i] = cv.take<float>((c.resblock ? 4 : 2) * B * (c.denseblock && i == 0 ? 1 + dg : pc_out) * 12 * Ti);   // (--denseblock, one layer: dL/d(fold | growth))
            if (i >= 1) {
                b->aff_cat[i] = cv.take<float>(3 * (d.prev_pc + d.out_p));
                b->aff_p2pin[i] = cv.take<float>(3 * (d.prev_pc + d.prev_p));
                b->g_cat[i] = cv.take<float>(B * (d.prev_pc + d.out_p + dg) * 12 * Ti);
                for (int j = 0; j < n_st; ++j) {
                    b->pst[i].push_back(cv.take<float>(width(j) * B * d.out_p * P * Ti));
                    b->aff_pst[i].push_back(cv.take<float>(3 * width(j) * d.out_p));
                }
                b->g_p[i] = cv.take<float>((c.resblock ? 4 : 2) * B * d.out_p * P * Ti);
                b->g_pin[i] = cv.take<float>(B * (d.prev_p + d.prev_pc) * P * Ti);
                b->g_psix[i] = cv.take<float>(B * d.prev_pc * 36 * Ti);
            }
        }
        if (c.denseblock) {
            for (auto* v : {&b->dn_bott_pc, &b->dn_aff1_pc, &b->dn_aff2_pc, &b->dn_bott_p, &b->dn_aff1_p, &b->dn_aff2_p}) v->assign(L, {});
            size_t gb_pc = 0, gb_p = 0;
            for (int i = 0; i < L; ++i) {
                const int Ti = b->Tl[i];
                const auto& d = n->dims[i];
                for (size_t j = 0; j < n->dense_pc[i].size(); ++j) {
                    const DensePack& dp = n->dense_pc[i][j];
                    b->dn_bott_pc[i].push_back(cv.take<float>(B * dp.c1.cout * 12 * Ti));
                    b->dn_aff1_pc[i].push_back(cv.take<float>(3 * static_cast<size_t>(dp.c1.cin)));
                    b->dn_aff2_pc[i].push_back(cv.take<float>(3 * static_cast<size_t>(dp.c1.cout)));
                    gb_pc = std::max(gb_pc, B * dp.c1.cout * 12 * Ti);
                }
                for (size_t j = 0; i >= 1 && j < n->dense_p[i].size(); ++j) {
                    const DensePack& dp = n->dense_p[i][j];
                    b->dn_bott_p[i].push_back(cv.take<float>(B * dp.c1.cout * P * Ti));
                    b->dn_aff1_p[i].push_back(cv.take<float>(3 * static_cast<size_t>(dp.c1.cin)));
                    b->dn_aff2_p[i].push_back(cv.take<float>(3 * static_cast<size_t>(dp.c1.cout)));
                    gb_p = std::max(gb_p, B * dp.c1.cout * P * Ti);
                }
                (void)d;
            }
            b->dn_gbott_pc = cv.take<float>(gb_pc);
            b->dn_gbott_p = cv.take<float>(gb_p);
        }
        b->g_fold0 = cv.take<float>(B * 12 * frames);
        b->g_pcf = cv.take<float>(B * n->final_ch * 12 * b->Tf);
        b->g_hid = cv.take<float>(hid);
        for (int h = 0; h < 3; ++h) {
            if (h == 2 && !c.genre) break;
            for (int j = 0; j + 1 < c.head_layers; ++j) {
                b->hst[h].push_back(cv.take<float>(hid));
                b->aff_hst[h].push_back(cv.take<float>(3 * 2 * n->final_ch));
            }
            b->g_map[h] = cv.take<float>(B * 12 * b->Tf);
        }
    }
    b->bytes = ake::align_up(cv.off, 256);
    return AKE_OK;
}

}  // namespace

extern "C" {

int ake_pcnet_precision(const ake_pcnet* n) { return n ? n->cfg.precision : AKE_ERR_INVALID; }

int ake_pcnet_default_config(ake_pcnet_config* cfg, int octaves, int genre) {
    AKE_REQUIRE(cfg && octaves > 0, AKE_ERR_INVALID, "ake_pcnet_default_config: bad argument");
    std::memset(cfg, 0, sizeof(*cfg));
    cfg->pitches = 36 * octaves;      // train_model.py:92-95
    cfg->pitch_classes = 12;
    cfg->num_layers = 2; cfg->kernel_size = 7; cfg->conv_layers = 3; cfg->n_filters = 4;   // train_model.py:190-197
    cfg->head_layers = 2; cfg->time_pool_size = 2;                                          // train_model.py:212-217
    cfg->genre = genre ? 1 : 0;
    return AKE_OK;
}

int ake_pcnet_create(const ake_pcnet_config* cfg, ake_pcnet** out) {
    AKE_REQUIRE(cfg && out, AKE_ERR_INVALID, "ake_pcnet_create: null argument");
    const ake_pcnet_config& c = *cfg;
    AKE_REQUIRE(!c.only_semitones, AKE_ERR_UNSUPPORTED, "pcnet: the only_semitones variant is not built");
    AKE_REQUIRE(!(c.denseblock && (c.resblock || c.pc2p_mem || c.p2pc_conv || c.stay_sixth || c.local || c.kernel_size != 7)), AKE_ERR_UNSUPPORTED,
                "pcnet: denseblock together with resblock / pc2p_mem / p2pc_conv / stay_sixth / local / kernel_size != 7 is not built");
    AKE_REQUIRE(!(c.stay_sixth && c.pc2p_mem), AKE_ERR_UNSUPPORTED, "pcnet: stay_sixth together with pc2p_mem is not built");
    AKE_REQUIRE(c.local >= 0, AKE_ERR_INVALID, "pcnet: local = pooling window of the --local heads (0: off)");
    AKE_REQUIRE(c.pitch_classes == 12, AKE_ERR_UNSUPPORTED, "pcnet: pitch_classes must be 12");
    AKE_REQUIRE(c.pitches > 0 && c.pitches % 36 == 0, AKE_ERR_INVALID, "pcnet: pitches must be a multiple of 36");
    // --kernel_size (train_model.py:194): 7 runs the kernels built for it; 3 and 5 run every convolution on the generic kernels (conv_mfma_kernel,
    // conv_wgrad_kernel: kernel width is a parameter of theirs), inference and training
    AKE_REQUIRE(c.kernel_size == 7 || c.kernel_size == 5 || c.kernel_size == 3, AKE_ERR_UNSUPPORTED, "pcnet: kernel_size %d is not built (3, 5, 7)", c.kernel_size);
    AKE_REQUIRE(c.num_layers >= 1 && c.num_layers <= 4 && c.conv_layers >= 1 && c.n_filters >= 1 && c.head_layers >= 1,
                AKE_ERR_INVALID, "pcnet: bad layer counts");
    AKE_REQUIRE(c.time_pool_size >= 1, AKE_ERR_INVALID, "pcnet: bad time_pool_size");
    AKE_REQUIRE(c.precision == AKE_PRECISION_MIXED || c.precision == AKE_PRECISION_F32X3, AKE_ERR_INVALID,
                "pcnet: precision must be AKE_PRECISION_MIXED (0) or AKE_PRECISION_F32X3 (1), got %d", c.precision);
    auto* n = new ake_pcnet();
    n->cfg = c;
    if (c.local > 0) n->cfg.time_pool_size = 1;           // --local: the layers do not pool over time (models.py:348, 394)
    if (const char* e = ake::diag_env("AKE_PCNET_CHUNK")) n->chunk_clips = std::max(1, std::atoi(e));
    const int nf = c.n_filters, L = c.num_layers, k = c.kernel_size;
    n->dims.resize(L);
    for (int i = 1; i < L; ++i) {           // models.py:281-308
        LayerDims d;
        if (i == 1) { d.prev_p = 1; d.prev_pc = nf; d.out_p = 2 * nf; d.out_pc = 2 * d.out_p; }
        else {
            d.prev_p = i == 2 ? 2 * nf : 2 * nf * static_cast<int>(std::pow(4, i - 2));
            d.prev_pc = 2 * d.prev_p; d.out_p = 4 * d.prev_p; d.out_pc = 4 * d.prev_pc;
        }
        n->dims[i] = d;
    }
    n->final_ch = L == 1 ? nf : n->dims[L - 1].out_pc;   // models.py:694-710
    const int growth_all = nf * c.conv_layers;           // channels one dense block appends
    if (c.denseblock) {                                  // models.py:267-278 (layers), :678-689 (heads)
        int pp = 1, ppc = 1 + growth_all;
        for (int i = 1; i < L; ++i) {
            LayerDims d;
            d.prev_p = pp; d.prev_pc = ppc;
            d.out_p = pp + growth_all + ppc;             // block input (pitch stream | repeated up_sixth map) + its growth
            d.out_pc = ppc + d.out_p + growth_all;       // block input (pitch classes | folded semitone maps) + its growth
            n->dims[i] = d;
            pp = d.out_p; ppc = d.out_pc;
        }
        n->final_ch = L == 1 ? 1 + growth_all : n->dims[L - 1].out_pc;
    }
    // one dense block: layer j reads cin + j * nf channels; bn_size = cin / 2 (1 for a single input channel), models.py:189, 226
    auto add_dense_specs = [&](const std::string& base, int cin, bool equiv) {
        const int bott = (cin > 1 ? cin / 2 : 1) * nf;
        for (int j = 0; j < c.conv_layers; ++j) {
            const std::string lp = base + "denselayer" + std::to_string(j + 1) + ".";
            const int cj = cin + j * nf;
            add_bn_specs(n, lp + "norm1", cj);
            if (equiv) add_conv_specs(n, lp + "conv1.conv2d", bott, cj, 12, 1);
            else add_spec(n, lp + "conv1.weight", {bott, cj, 1, 1});                  // bias=False, models.py:464
            add_bn_specs(n, lp + "norm2", bott);
            if (equiv) add_conv_specs(n, lp + "conv2.conv2d", nf, bott, 12, k);
            else add_spec(n, lp + "conv2.weight", {nf, bott, k, k});
        }
    };
    // ---- expected state_dict entries ----
    for (int i = 0; i < L; ++i) {
        const std::string m = "model." + std::to_string(i) + ".";
        const LayerDims& d = n->dims[i];
        const int cs = i == 0 ? 1 : d.out_p;
        if (i == 0 || !c.stay_sixth) {                                        // --stay_sixth: no semitone conv after layer 0 (models.py:336)
            add_conv_specs(n, m + "pool_semi", cs, cs, 3, 3);                 // models.py:313 / :337
            add_bn_specs(n, m + "pool_semi_b", cs);
        }
        if (c.p2pc_conv) {                                                    // models.py:118-119: Pitch2PitchClassConv
            add_conv_specs(n, m + "pool.conv", cs, cs, c.pitches / 36, 1);
            add_bn_specs(n, m + "pool.bn", cs);
        }
        const int pc_in = i == 0 ? 1 : d.out_p + d.prev_pc, pc_out = i == 0 ? nf : d.out_pc;
        if (c.denseblock) {                                                   // models.py:188-189, 225-226
            add_dense_specs(m + "pc2pc.layer.0.", pc_in, true);
            if (i >= 1) {
                add_spec(n, m + "up_sixth.weight", {d.prev_pc, d.prev_pc, 3, 1});
                add_spec(n, m + "up_sixth.bias", {d.prev_pc});
                add_bn_specs(n, m + "up_sixth_b", d.prev_pc);
                add_dense_specs(m + "p2p.layer.0.", d.prev_pc + d.prev_p, false);
            }
            continue;
        }
        if (c.resblock) {                                                     // models.py:181-187: conv + BN, then conv_layers ResBlockEquivariant
            add_conv_specs(n, m + "pc2pc.layer.0.conv2d", pc_out, pc_in, 12, k);
            add_bn_specs(n, m + "pc2pc.layer.1", pc_out);
            for (int r = 0; r < c.conv_layers; ++r) {                         // models.py:429-441
                const std::string bp = m + "pc2pc.layer." + std::to_string(3 + r) + ".";
                add_conv_specs(n, bp + "conv1.conv2d", 2 * pc_out, pc_out, 12, k);
                add_bn_specs(n, bp + "b1", 2 * pc_out);
                add_conv_specs(n, bp + "conv2.conv2d", pc_out, 2 * pc_out, 12, k);
                add_bn_specs(n, bp + "b2", pc_out);
            }
        }
        for (int j = 0; j < c.conv_layers && !c.resblock; ++j) {              // models.py:191-197
            add_conv_specs(n, m + "pc2pc.layer." + std::to_string(3 * j) + ".conv2d", pc_out, j == 0 ? pc_in : pc_out, 12, k);
            add_bn_specs(n, m + "pc2pc.layer." + std::to_string(3 * j + 1), pc_out);
        }
        if (i >= 1) {
            if (!c.stay_sixth) {                                              // --stay_sixth: plain repeat of the pitch classes (models.py:322-323)
                add_spec(n, m + "up_sixth.weight", {d.prev_pc, d.prev_pc, 3, 1}); // models.py:325
                add_spec(n, m + "up_sixth.bias", {d.prev_pc});
                add_bn_specs(n, m + "up_sixth_b", d.prev_pc);
            }
            if (c.resblock) {                                                 // models.py:218-224, 402-414
                add_conv_specs(n, m + "p2p.layer.0", d.out_p, c.pc2p_mem ? d.prev_p : d.prev_pc + d.prev_p, k, k);
                add_bn_specs(n, m + "p2p.layer.1", d.out_p);
                for (int r = 0; r < c.conv_layers; ++r) {
                    const std::string bp = m + "p2p.layer." + std::to_string(3 + r) + ".";
                    add_conv_specs(n, bp + "conv1", 2 * d.out_p, d.out_p, k, k);
                    add_bn_specs(n, bp + "b1", 2 * d.out_p);
                    add_conv_specs(n, bp + "conv2", d.out_p, 2 * d.out_p, k, k);
                    add_bn_specs(n, bp + "b2", d.out_p);
                }
            }
            for (int j = 0; j < c.conv_layers && !c.resblock; ++j) {          // models.py:228-234
                add_conv_specs(n, m + "p2p.layer." + std::to_string(3 * j), d.out_p, j == 0 ? (c.pc2p_mem ? d.prev_p : d.prev_pc + d.prev_p) : d.out_p, k, k);
                add_bn_specs(n, m + "p2p.layer." + std::to_string(3 * j + 1), d.out_p);
            }
        }
    }
    for (const char* head : {"tonic_classifier", "key_classifier", "genre_classifier"}) {   // models.py:716-737
        const bool g = std::strcmp(head, "genre_classifier") == 0;
        if (g && !c.genre) continue;
        int ch = n->final_ch;
        for (int i = 0; i < c.head_layers; ++i) {
            const std::string base = std::string(head) + "." + std::to_string(3 * i) + (g ? "" : ".conv2d");
            if (i == c.head_layers - 1) {
                add_conv_specs(n, base, 1, ch, g ? 2 : 12, k);
            } else {
                const int co = i == 0 ? 2 * ch : ch;
                add_conv_specs(n, base, co, ch, g ? 1 : 12, k);
                add_bn_specs(n, std::string(head) + "." + std::to_string(3 * i + 1), co);
                ch = co;
            }
        }
    }
    n->host.resize(n->specs.size());
    n->grad_off.assign(n->specs.size(), 0);
    size_t goff = 0;
    for (size_t i = 0; i < n->specs.size(); ++i) {
        n->grad_off[i] = goff;
        size_t cnt = 1;
        for (int d = 0; d < n->specs[i].ndim; ++d) cnt *= static_cast<size_t>(n->specs[i].shape[d]);
        goff += cnt;
    }
    n->grad_floats = goff;
    *out = n;
    return AKE_OK;
}

void ake_pcnet_destroy(ake_pcnet* n) {
    if (!n) return;
    if (n->blob_dev) (void)hipFree(n->blob_dev);
    for (void* p : {static_cast<void*>(n->map_dev), static_cast<void*>(n->fold_ch_dev), static_cast<void*>(n->fold_bn_dev),
                    static_cast<void*>(n->run_off_dev), static_cast<void*>(n->run_off2_dev), static_cast<void*>(n->folded_dev), static_cast<void*>(n->bf_frags_dev),
                    static_cast<void*>(n->dense_aff_dev), static_cast<void*>(n->dense_aff_idx_dev)})
        if (p) (void)hipFree(p);
    delete n;
}

int ake_pcnet_pitches(const ake_pcnet* n) { return n ? n->cfg.pitches : 0; }

int ake_pcnet_num_tensors(const ake_pcnet* n) { return n ? static_cast<int>(n->specs.size()) : 0; }

int ake_pcnet_tensor_info(const ake_pcnet* n, int index, const char** name, int64_t shape[4], int* ndim) {
    AKE_REQUIRE(n && index >= 0 && index < static_cast<int>(n->specs.size()), AKE_ERR_INVALID, "tensor_info: index out of range");
    const TensorSpec& s = n->specs[index];
    if (name) *name = s.name.c_str();
    if (shape) std::memcpy(shape, s.shape, sizeof(s.shape));
    if (ndim) *ndim = s.ndim;
    return AKE_OK;
}

int ake_pcnet_set_tensor(ake_pcnet* n, const char* name, const float* host_data, const int64_t* shape, int ndim) {
    AKE_REQUIRE(n && name && host_data && shape, AKE_ERR_INVALID, "set_tensor: null argument");
    auto it = n->spec_index.find(name);
    AKE_REQUIRE(it != n->spec_index.end(), AKE_ERR_INVALID, "set_tensor: unexpected key '%s' for this configuration", name);
    const TensorSpec& s = n->specs[it->second];
    AKE_REQUIRE(ndim == s.ndim, AKE_ERR_INVALID, "set_tensor: '%s' has %d dims, expected %d", name, ndim, s.ndim);
    size_t count = 1;
    for (int i = 0; i < ndim; ++i) {
        AKE_REQUIRE(shape[i] == s.shape[i], AKE_ERR_INVALID, "set_tensor: '%s' dim %d is %lld, expected %lld", name, i,
                    static_cast<long long>(shape[i]), static_cast<long long>(s.shape[i]));
        count *= static_cast<size_t>(shape[i]);
    }
    HostTensor& h = n->host[it->second];
    h.data.assign(host_data, host_data + count);
    h.set = true;
    n->finalized = false;
    return AKE_OK;
}

// Packs every convolution of the net into the blob.  train == false: eval-mode BatchNorm folded into w/b.
// train == true: raw weights; each BatchNorm is registered as a layer of its own (gamma/beta in the blob).
static void build_packs(ake_pcnet* n, bool train) {
    const auto& c = n->cfg;
    const int L = c.num_layers, k = c.kernel_size;
    auto& semi = train ? n->semi_t : n->semi;
    auto& up = train ? n->up_t : n->up;
    auto& pc2pc = train ? n->pc2pc_t : n->pc2pc;
    auto& p2p = train ? n->p2p_t : n->p2p;
    semi.assign(L, PackedConv()); up.assign(L, PackedConv());
    (train ? n->foldc_t : n->foldc).assign(L, PackedConv());
    pc2pc.assign(L, {}); p2p.assign(L, {});
    if (!train) { n->dense_pc.assign(L, {}); n->dense_p.assign(L, {}); }
    // --denseblock (eval packs only): layer j of a block reads cin + j * nf channels
    auto dense_block = [&](const std::string& base, int cin, bool equiv, std::vector<DensePack>& out) {
        const int nf = c.n_filters, bott = (cin > 1 ? cin / 2 : 1) * nf;
        auto table = [&](const std::string& bnp, int C, float slope) {
            const size_t off = n->dense_aff_floats;
            n->aff_recs.push_back({bnp, C, slope, off});
            n->dense_aff_floats += static_cast<size_t>(3) * C;
            return off;
        };
        for (int j = 0; j < c.conv_layers; ++j) {
            const std::string lp = base + "denselayer" + std::to_string(j + 1) + ".";
            const int cj = cin + j * nf;
            DensePack dp;
            dp.norm1 = lp + "norm1"; dp.norm2 = lp + "norm2";
            dp.w1 = lp + (equiv ? "conv1.conv2d.weight" : "conv1.weight"); dp.w2 = lp + (equiv ? "conv2.conv2d.weight" : "conv2.weight");
            dp.b2 = equiv ? lp + "conv2.conv2d.bias" : "";
            dp.aff1 = table(lp + "norm1", cj, 0.01f);                                  // relu1 = nn.LeakyReLU, models.py:463 / :526
            dp.c1 = equiv ? dense_conv_pack(n, lp + "conv1.conv2d.weight", lp + "conv1.conv2d.bias", bott, cj, 12, 1)
                          : dense_conv_pack(n, lp + "conv1.weight", "", bott, cj, 1, 1);
            dp.aff2 = table(lp + "norm2", bott, 0.f);                                  // relu2 = nn.ReLU, models.py:467 / :530
            dp.c2 = equiv ? dense_conv_pack(n, lp + "conv2.conv2d.weight", lp + "conv2.conv2d.bias", nf, bott, 12, k)
                          : dense_conv_pack(n, lp + "conv2.weight", "", nf, bott, k, k);
            out.push_back(dp);
        }
    };
    if (train) { n->pc2pc_d.assign(L, {}); n->p2p_d.assign(L, {}); n->head_key_d.clear(); n->head_tonic_d.clear(); n->head_genre_d.clear(); }
    auto bn = [&](const std::string& prefix, int C) -> std::string {
        if (!train) return prefix;
        ake_pcnet::BnLayer l;
        l.name = prefix; l.C = C; l.ch_off = n->bn_channels;
        n->blob.resize(ake::align_up(n->blob.size(), 64));
        l.gamma_off = n->blob.size();
        for (float v : T(n, prefix + ".weight")) n->blob.push_back(v);
        l.beta_off = n->blob.size();
        for (float v : T(n, prefix + ".bias")) n->blob.push_back(v);
        n->bn_index[prefix] = static_cast<int>(n->bns.size());
        n->bns.push_back(l);
        n->bn_channels += C;
        return "";
    };
    // --denseblock, training pass: the blocks' BatchNorm layers in forward order and the data-gradient packs (the convolutions themselves
    // keep their raw weights in both modes: the eval packs serve the training forward)
    auto dense_block_train = [&](int cin, bool equiv, std::vector<DensePack>& packs) {
        const int nf = c.n_filters;
        for (size_t j = 0; j < packs.size(); ++j) {
            DensePack& dp = packs[j];
            const int cj = cin + static_cast<int>(j) * nf, bott = dp.c1.cout;
            bn(dp.norm1, cj); dp.bn1 = n->bn_index.at(dp.norm1);
            bn(dp.norm2, bott); dp.bn2 = n->bn_index.at(dp.norm2);
            dp.d1 = dense_dgrad_pack(n, dp.w1, bott, cj, equiv ? 12 : 1, 1);
            dp.d2 = dense_dgrad_pack(n, dp.w2, nf, bott, equiv ? 12 : k, k);
        }
    };
    for (int i = 0; i < L; ++i) {
        const std::string m = "model." + std::to_string(i) + ".";
        const LayerDims& d = n->dims[i];
        const int cs = i == 0 ? 1 : d.out_p;
        if (i >= 1) {   // creation order mirrors the forward order: up_sixth, p2p, pool_semi, pc2pc
            if (!c.stay_sixth) {
            std::vector<double> w, b;
            fold(n, m + "up_sixth.weight", m + "up_sixth.bias", bn(m + "up_sixth_b", d.prev_pc), d.prev_pc,
                 static_cast<size_t>(d.prev_pc) * 3, true, d.prev_pc, w, b);
            PackedConv u;
            u.cin = u.cout = d.prev_pc; u.kh = 3; u.kw = 1;
            n->blob.resize(ake::align_up(n->blob.size(), 64));
            u.w_off = n->blob.size();
            for (double v : w) n->blob.push_back(static_cast<float>(v));   // stays [ci][co][3]
            n->blob.resize(ake::align_up(n->blob.size(), 64));
            u.b_off = n->blob.size();
            for (double v : b) n->blob.push_back(static_cast<float>(v));
            up[i] = u;
            }
            if (c.resblock) {   // [conv0, (conv1, conv2) per block]; the data-gradient packs in the same order
                const int cin0 = c.pc2p_mem ? d.prev_p : d.prev_pc + d.prev_p;
                p2p[i].push_back(fold_pack(n, m + "p2p.layer.0", bn(m + "p2p.layer.1", d.out_p), d.out_p, cin0, k, k));
                if (train) n->p2p_d[i].push_back(dgrad_pack(n, m + "p2p.layer.0.weight", d.out_p, cin0, k, k));
                for (int r = 0; r < c.conv_layers; ++r) {
                    const std::string bp = m + "p2p.layer." + std::to_string(3 + r) + ".";
                    p2p[i].push_back(fold_pack(n, bp + "conv1", bn(bp + "b1", 2 * d.out_p), 2 * d.out_p, d.out_p, k, k));
                    p2p[i].push_back(fold_pack(n, bp + "conv2", bn(bp + "b2", d.out_p), d.out_p, 2 * d.out_p, k, k));
                    if (train) {
                        n->p2p_d[i].push_back(dgrad_pack(n, bp + "conv1.weight", 2 * d.out_p, d.out_p, k, k));
                        n->p2p_d[i].push_back(dgrad_pack(n, bp + "conv2.weight", d.out_p, 2 * d.out_p, k, k));
                    }
                }
            }
            if (c.denseblock && !train) dense_block(m + "p2p.layer.0.", d.prev_pc + d.prev_p, false, n->dense_p[i]);
            if (c.denseblock && train) dense_block_train(d.prev_pc + d.prev_p, false, n->dense_p[i]);
            for (int j = 0; j < c.conv_layers && !c.resblock && !c.denseblock; ++j) {
                const int cin_j = j == 0 ? (c.pc2p_mem ? d.prev_p : d.prev_pc + d.prev_p) : d.out_p;
                p2p[i].push_back(fold_pack(n, m + "p2p.layer." + std::to_string(3 * j), bn(m + "p2p.layer." + std::to_string(3 * j + 1), d.out_p),
                                           d.out_p, cin_j, k, k));
                if (train) n->p2p_d[i].push_back(dgrad_pack(n, m + "p2p.layer." + std::to_string(3 * j) + ".weight", d.out_p, cin_j, k, k));
            }
        }
        if (i == 0 || !c.stay_sixth) semi[i] = fold_pack(n, m + "pool_semi", bn(m + "pool_semi_b", cs), cs, cs, 3, 3);
        if (c.p2pc_conv) {   // [co][ci][n_oct], BatchNorm folded (eval); plain layout as up_sixth's
            std::vector<double> w, bb;
            fold(n, m + "pool.conv.weight", m + "pool.conv.bias", bn(m + "pool.bn", cs), cs, static_cast<size_t>(cs) * (c.pitches / 36), false, cs, w, bb);
            PackedConv u;
            u.cin = u.cout = cs; u.kh = c.pitches / 36; u.kw = 1;
            n->blob.resize(ake::align_up(n->blob.size(), 64));
            u.w_off = n->blob.size();
            for (double v : w) n->blob.push_back(static_cast<float>(v));
            n->blob.resize(ake::align_up(n->blob.size(), 64));
            u.b_off = n->blob.size();
            for (double v : bb) n->blob.push_back(static_cast<float>(v));
            (train ? n->foldc_t : n->foldc)[i] = u;
        }
        const int pc_in = i == 0 ? 1 : d.out_p + d.prev_pc, pc_out = i == 0 ? c.n_filters : d.out_pc;
        if (c.resblock) {
            pc2pc[i].push_back(fold_pack(n, m + "pc2pc.layer.0.conv2d", bn(m + "pc2pc.layer.1", pc_out), pc_out, pc_in, 12, k));
            if (train) n->pc2pc_d[i].push_back(dgrad_pack(n, m + "pc2pc.layer.0.conv2d.weight", pc_out, pc_in, 12, k));
            for (int r = 0; r < c.conv_layers; ++r) {
                const std::string bp = m + "pc2pc.layer." + std::to_string(3 + r) + ".";
                pc2pc[i].push_back(fold_pack(n, bp + "conv1.conv2d", bn(bp + "b1", 2 * pc_out), 2 * pc_out, pc_out, 12, k));
                pc2pc[i].push_back(fold_pack(n, bp + "conv2.conv2d", bn(bp + "b2", pc_out), pc_out, 2 * pc_out, 12, k));
                if (train) {
                    n->pc2pc_d[i].push_back(dgrad_pack(n, bp + "conv1.conv2d.weight", 2 * pc_out, pc_out, 12, k));
                    n->pc2pc_d[i].push_back(dgrad_pack(n, bp + "conv2.conv2d.weight", pc_out, 2 * pc_out, 12, k));
                }
            }
        }
        if (c.denseblock && !train) dense_block(m + "pc2pc.layer.0.", pc_in, true, n->dense_pc[i]);
        if (c.denseblock && train) dense_block_train(pc_in, true, n->dense_pc[i]);
        for (int j = 0; j < c.conv_layers && !c.resblock && !c.denseblock; ++j) {
            pc2pc[i].push_back(fold_pack(n, m + "pc2pc.layer." + std::to_string(3 * j) + ".conv2d",
                                         bn(m + "pc2pc.layer." + std::to_string(3 * j + 1), pc_out), pc_out, j == 0 ? pc_in : pc_out, 12, k));
            if (train) n->pc2pc_d[i].push_back(dgrad_pack(n, m + "pc2pc.layer." + std::to_string(3 * j) + ".conv2d.weight", pc_out,
                                                          j == 0 ? pc_in : pc_out, 12, k));
        }
    }
    for (const char* head : {"tonic_classifier", "key_classifier", "genre_classifier"}) {
        const bool g = std::strcmp(head, "genre_classifier") == 0;
        if (g && !c.genre) continue;
        const bool is_key = std::strcmp(head, "key_classifier") == 0;
        auto& vec = train ? (g ? n->head_genre_t : (is_key ? n->head_key_t : n->head_tonic_t))
                          : (g ? n->head_genre : (is_key ? n->head_key : n->head_tonic));
        vec.clear();
        int ch = n->final_ch;
        for (int i = 0; i < c.head_layers; ++i) {
            const std::string base = std::string(head) + "." + std::to_string(3 * i) + (g ? "" : ".conv2d");
            auto& dvec = g ? n->head_genre_d : (is_key ? n->head_key_d : n->head_tonic_d);
            if (i == c.head_layers - 1) {
                vec.push_back(fold_pack(n, base, "", 1, ch, g ? 2 : 12, k));
                if (train) dvec.push_back(dgrad_pack(n, base + ".weight", 1, ch, g ? 2 : 12, k));
            } else {
                const int co = i == 0 ? 2 * ch : ch;
                vec.push_back(fold_pack(n, base, bn(std::string(head) + "." + std::to_string(3 * i + 1), co), co, ch, g ? 1 : 12, k));
                if (train) dvec.push_back(dgrad_pack(n, base + ".weight", co, ch, g ? 1 : 12, k));
                ch = co;
            }
        }
    }
    if (train) {   // raw copies of every tensor (small backward kernels read the reference layout) + the flat gradient layout
        n->raw_w_off.assign(n->specs.size(), 0);
        for (size_t i = 0; i < n->specs.size(); ++i) {
            n->blob.resize(ake::align_up(n->blob.size(), 64));
            n->raw_w_off[i] = n->blob.size();
            for (float v : n->host[i].data) n->blob.push_back(v);
        }
    }
}

constexpr int32_t kMapFolded = 1 << 30, kMapIndex = kMapFolded - 1;

namespace {

__global__ void fold_params_kernel(const float* __restrict__ params, const int32_t* __restrict__ fold_ch, const int32_t* __restrict__ fold_bn,
                                   float* __restrict__ folded, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = params[i];
    const int32_t f = fold_ch[i];
    if (f >= 0) {
        const int32_t* o = fold_bn + 4 * (f >> 1);
        const double s = static_cast<double>(params[o[0]]) / sqrt(static_cast<double>(params[o[3]]) + 1e-5);
        v = (f & 1) ? static_cast<float>((static_cast<double>(v) - static_cast<double>(params[o[2]])) * s + static_cast<double>(params[o[1]]))
                    : static_cast<float>(static_cast<double>(v) * s);
    }
    folded[i] = v;
}

__global__ void gather_blob_kernel(const float* __restrict__ params, const float* __restrict__ folded, const int32_t* __restrict__ map,
                                   float* __restrict__ blob, int n) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int32_t m = map[j];
    blob[j] = m < 0 ? 0.f : ((m & kMapFolded) ? folded : params)[m & kMapIndex];
}

// nn.BatchNorm2d's train-mode side effect on the flat parameter buffer: running <- (1-m)*running + m*(mean, unbiased var)
__global__ void running_stats_kernel(const float* __restrict__ bstats, const int32_t* __restrict__ run_off, float* __restrict__ params,
                                     float momentum, int n_ch) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_ch || run_off[2 * c] < 0) return;
    const float mean = bstats[3 * c], var = bstats[3 * c + 1], cnt = bstats[3 * c + 2];
    const float unbiased = var * cnt / fmaxf(cnt - 1.f, 1.f);
    float* rm = params + run_off[2 * c];
    float* rv = params + run_off[2 * c + 1];
    *rm = (1.f - momentum) * *rm + momentum * mean;
    *rv = (1.f - momentum) * *rv + momentum * unbiased;
}

void reset_packs(ake_pcnet* n) {
    n->blob.clear();
    n->bns.clear(); n->bn_index.clear(); n->bn_channels = 0;
    n->aff_recs.clear(); n->dense_aff_floats = 0;
}


// The blob layout is a pure function of the configuration: build it once with index-coded values to learn, for every
// blob float, which flat parameter it comes from.  (Indices + 1 < 2^24 are exact in f32; the eval fold is neutralised by
// gamma = 1, beta = mean = 0, var = 1 - eps so that folded values still round to the index.)
int trace_maps(ake_pcnet* n) {
    std::vector<std::vector<float>> saved(n->specs.size());
    for (size_t i = 0; i < n->specs.size(); ++i) saved[i] = n->host[i].data;
    AKE_REQUIRE(n->grad_floats + 1 < (1u << 24), AKE_ERR_UNSUPPORTED, "finalize: %zu parameters exceed the index-trace range", n->grad_floats);
    auto ends_with = [](const std::string& a, const char* suf) { const size_t l = std::strlen(suf); return a.size() >= l && a.compare(a.size() - l, l, suf) == 0; };
    auto is_bn = [&](size_t i) {     // BatchNorm tensors are the ones that come with running statistics
        const std::string& nm = n->specs[i].name;
        const std::string prefix = nm.substr(0, nm.rfind('.'));
        return n->spec_index.count(prefix + ".running_mean") > 0;
    };
    auto fill_index = [&](size_t i) { for (size_t j = 0; j < n->host[i].data.size(); ++j) n->host[i].data[j] = static_cast<float>(n->grad_off[i] + j + 1); };
    // pass A: eval packs
    for (size_t i = 0; i < n->specs.size(); ++i) {
        const std::string& nm = n->specs[i].name;
        if (!is_bn(i)) { fill_index(i); continue; }
        const float v = ends_with(nm, ".weight") ? 1.f : (ends_with(nm, ".running_var") ? 1.f - 1e-5f : 0.f);
        std::fill(n->host[i].data.begin(), n->host[i].data.end(), v);
    }
    reset_packs(n);
    n->fold_records.clear();
    n->tracing = true;
    build_packs(n, false);
    n->tracing = false;
    const size_t eval_end = n->blob.size();
    // pass B: training packs (raw values everywhere)
    for (size_t i = 0; i < n->specs.size(); ++i) fill_index(i);
    build_packs(n, true);
    n->blob.resize(ake::align_up(n->blob.size() + 64, 64), 0.f);
    n->map_host.assign(n->blob.size(), -1);
    for (size_t j = 0; j < n->blob.size(); ++j) {
        const long v = std::lround(static_cast<double>(n->blob[j]));
        if (v <= 0) continue;
        AKE_REQUIRE(static_cast<size_t>(v) <= n->grad_floats, AKE_ERR_STATE, "finalize: index trace out of range at blob[%zu]", j);
        n->map_host[j] = static_cast<int32_t>(v - 1) | (j < eval_end ? kMapFolded : 0);
    }
    for (size_t i = 0; i < n->specs.size(); ++i) n->host[i].data = saved[i];
    return AKE_OK;
}

int upload_i32(const std::vector<int32_t>& v, int32_t** dev) {
    if (*dev) { (void)hipFree(*dev); *dev = nullptr; }
    AKE_HIP_CHECK(hipMalloc(dev, std::max<size_t>(v.size(), 1) * sizeof(int32_t)));
    AKE_HIP_CHECK(hipMemcpy(*dev, v.data(), v.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    return AKE_OK;
}

// --denseblock: (re)compute the BatchNorm-on-load tables from the flat parameter buffer on the device
int rebuild_dense_affine(ake_pcnet* n, const float* params_dev, hipStream_t s) {
    if (n->aff_recs.empty()) return AKE_OK;
    const int rows = static_cast<int>(n->dense_aff_floats / 3);
    if (!n->dense_aff_dev) {
        std::vector<int32_t> idx;
        idx.reserve(static_cast<size_t>(rows) * 5);
        auto off = [&](const std::string& key) { return static_cast<int32_t>(n->grad_off[n->spec_index.at(key)]); };
        for (const auto& r : n->aff_recs)
            for (int ch = 0; ch < r.C; ++ch) {
                int32_t bits;
                std::memcpy(&bits, &r.slope, sizeof(bits));
                for (const char* f : {".weight", ".bias", ".running_mean", ".running_var"}) idx.push_back(off(r.bn + f) + ch);
                idx.push_back(bits);
            }
        int rc = upload_i32(idx, &n->dense_aff_idx_dev);
        if (rc) return rc;
        AKE_HIP_CHECK(hipMalloc(&n->dense_aff_dev, n->dense_aff_floats * sizeof(float)));
    }
    ake::ProfScope ps("dense_affine_kernel", s);
    hipLaunchKernelGGL(dense_affine_kernel, dim3((rows + 255) / 256), dim3(256), 0, s, params_dev, n->dense_aff_idx_dev, n->dense_aff_dev, rows);
    AKE_HIP_CHECK(hipGetLastError());
    return AKE_OK;
}

int build_fold_tables(ake_pcnet* n) {
    std::vector<int32_t> fold_ch(n->grad_floats, -1), fold_bn, run_off(static_cast<size_t>(n->bn_channels) * 2, 0), run_off2(static_cast<size_t>(n->bn_channels) * 2, -1);
    n->recomputed_bn_channels = 0;
    auto off = [&](const std::string& key) { return static_cast<int32_t>(n->grad_off[n->spec_index.at(key)]); };
    int base = 0;
    for (const auto& r : n->fold_records) {
        const int32_t w0 = off(r.wkey), b0 = off(r.bkey);
        const size_t total = r.per_out * r.cout, k = r.transposed ? r.per_out / r.cin : 0;
        for (size_t i = 0; i < total; ++i) {
            const int co = r.transposed ? static_cast<int>((i / k) % r.cout) : static_cast<int>(i / r.per_out);
            fold_ch[w0 + i] = (base + co) << 1;
        }
        for (int co = 0; co < r.cout; ++co) {
            fold_ch[b0 + co] = ((base + co) << 1) | 1;
            fold_bn.push_back(off(r.bn + ".weight") + co);
            fold_bn.push_back(off(r.bn + ".bias") + co);
            fold_bn.push_back(off(r.bn + ".running_mean") + co);
            fold_bn.push_back(off(r.bn + ".running_var") + co);
        }
        base += r.cout;
    }
    for (const auto& l : n->bns)
        for (int c = 0; c < l.C; ++c) {
            run_off[2 * (l.ch_off + c)] = off(l.name + ".running_mean") + c;
            run_off[2 * (l.ch_off + c) + 1] = off(l.name + ".running_var") + c;
            // --denseblock: the reference checkpoints norm1 + conv1 of every dense layer (models.py:484-489, 553): autograd's backward runs that
            // half again, in train mode -- its BatchNorm blends the batch statistics into the running ones a second time
            const bool again = n->cfg.denseblock && l.name.size() > 6 && l.name.compare(l.name.size() - 6, 6, ".norm1") == 0;
            if (again) {
                run_off2[2 * (l.ch_off + c)] = run_off[2 * (l.ch_off + c)];
                run_off2[2 * (l.ch_off + c) + 1] = run_off[2 * (l.ch_off + c) + 1];
                ++n->recomputed_bn_channels;
            }
        }
    int rc;
    if ((rc = upload_i32(fold_ch, &n->fold_ch_dev)) || (rc = upload_i32(fold_bn, &n->fold_bn_dev)) || (rc = upload_i32(run_off, &n->run_off_dev)) || (rc = upload_i32(run_off2, &n->run_off2_dev)) ||
        (rc = upload_i32(n->map_host, &n->map_dev)))
        return rc;
    if (n->folded_dev) { (void)hipFree(n->folded_dev); n->folded_dev = nullptr; }
    AKE_HIP_CHECK(hipMalloc(&n->folded_dev, n->grad_floats * sizeof(float)));
    return AKE_OK;
}

// Inference runs the 8 -> 8 channel 7x7 pitch convolutions (every conv of a Pitch2Pitch stack but the first) on bf16 MFMA
// with split operands; their weight fragments are derived on the device from the eval packs, after every (re)pack.
bool pc_bf16_eligible(const PackedConv& pc) { return (pc.kh == 12 || pc.kh == 1) && pc.kw == 7 && pc.cin <= 16 && (pc.cout == 16 || pc.cout == 32); }

int rebuild_bf16_frags(ake_pcnet* n, hipStream_t s, bool train_only = false) {
    size_t count = 0;
    for (auto& layer : n->p2p)
        for (size_t j = 0; j < layer.size(); ++j) {
            PackedConv& pc = layer[j];
            pc.bf_off = -1;
            if (pc.cin <= 8 && pc.cout == 8 && pc.kh == 7 && pc.kw == 7 && pc.co == 8) { pc.bf_off = static_cast<long long>(count); count += kBfFragsPerConv; }
        }
    // train-mode forward and data gradient of the pitch convs (conv_p2p_f16x3_kernel): f16 hi + lo fragments from the raw weights
    struct TrainFrag { PackedConv* pc; size_t raw; int cin, cout, flip; };
    std::vector<TrainFrag> tfr;
    if (!n->raw_w_off.empty())
        for (size_t i = 1; i < n->p2p_t.size() && i < n->p2p_d.size(); ++i)
            for (size_t j = 0; j < n->p2p_t[i].size() && j < n->p2p_d[i].size(); ++j) {
                PackedConv& pt = n->p2p_t[i][j];
                PackedConv& pd = n->p2p_d[i][j];
                pt.bf_off = pd.bf_off = -1;
                const auto it = n->spec_index.find("model." + std::to_string(i) + ".p2p.layer." + std::to_string(3 * j) + ".weight");
                if (it == n->spec_index.end() || pt.kh != 7 || pt.kw != 7 || pt.cin > 8 || pt.cout > 8 || n->cfg.resblock || n->cfg.denseblock) continue;
                const size_t raw = n->raw_w_off[it->second];
                pt.bf_off = static_cast<long long>(count); count += kBfFragsPerConv;
                tfr.push_back({&pt, raw, pt.cin, pt.cout, 0});
                pd.bf_off = static_cast<long long>(count); count += kBfFragsPerConv;
                tfr.push_back({&pd, raw, pt.cin, pt.cout, 1});
            }
    // training-mode forward of the pitch-class convs with 16 / 32 output channels (conv_pc_bf16_kernel<.., F16X3>): f16 hi + lo fragments
    // from the training packs (raw weights in the VALU layout)
    std::vector<PackedConv*> tpc;
    if (!n->raw_w_off.empty() && !n->cfg.resblock && !n->cfg.denseblock) {
        for (size_t i = 1; i < n->pc2pc_t.size(); ++i)
            for (PackedConv& pc : n->pc2pc_t[i]) tpc.push_back(&pc);
        if (!n->head_key_t.empty()) tpc.push_back(&n->head_key_t[0]);
        if (!n->head_tonic_t.empty()) tpc.push_back(&n->head_tonic_t[0]);
        if (n->head_genre_t.size() == 2) tpc.push_back(&n->head_genre_t[0]);
    }
    for (PackedConv* pc : tpc) {
        pc->bf_off = -1;
        if (pc_bf16_eligible(*pc)) { pc->bf_off = static_cast<long long>(count); count += static_cast<size_t>(pc->kh) * 4 * (pc->cout / 16) * 2 * 64 + 64; }
    }
    // data gradients of the pitch-class stacks of layers >= 1 (16 gradient channels in, <= 32 out: the first conv's is 24 wide) on the same
    // f16 x 3 kernel; the transposed + flipped weights are the data-gradient packs
    std::vector<PackedConv*> tpd;
    if (!n->raw_w_off.empty() && !n->cfg.resblock && !n->cfg.denseblock)
        for (size_t i = 1; i < n->pc2pc_d.size(); ++i)
            for (PackedConv& pc : n->pc2pc_d[i]) tpd.push_back(&pc);
    for (PackedConv* pc : tpd) {
        pc->bf_off = -1;
        if (pc->kh == 12 && pc->kw == 7 && pc->cin == 16 && pc->cout <= 32) {
            pc->bf_off = static_cast<long long>(count);
            count += static_cast<size_t>(pc->kh) * 4 * ((pc->cout + 15) / 16) * 2 * 64 + 64;
        }
    }
    // ... and of the heads' first convolutions (32 gradient channels -> 16 features: two 16-channel halves, bf_off / bf_off2)
    std::vector<PackedConv*> thd;
    if (!n->raw_w_off.empty() && !n->cfg.resblock && !n->cfg.denseblock && n->cfg.head_layers == 2) {
        if (!n->head_key_d.empty()) thd.push_back(&n->head_key_d[0]);
        if (!n->head_tonic_d.empty()) thd.push_back(&n->head_tonic_d[0]);
        if (!n->head_genre_d.empty()) thd.push_back(&n->head_genre_d[0]);
    }
    for (PackedConv* pc : thd) {
        pc->bf_off = pc->bf_off2 = -1;
        if ((pc->kh == 12 || pc->kh == 1) && pc->kw == 7 && pc->cin == 32 && pc->cout == 16) {
            const size_t one = static_cast<size_t>(pc->kh) * 4 * 2 * 64 + 64;
            pc->bf_off = static_cast<long long>(count); pc->bf_off2 = static_cast<long long>(count + one);
            count += 2 * one;
        }
    }
    std::vector<PackedConv*> pcs;                             // pitch-class convolutions: the PitchClass2PitchClass stacks and the heads' first conv
    for (auto& layer : n->pc2pc)
        for (PackedConv& pc : layer) pcs.push_back(&pc);
    if (!n->head_key.empty()) pcs.push_back(&n->head_key[0]);
    if (!n->head_tonic.empty()) pcs.push_back(&n->head_tonic[0]);
    if (n->head_genre.size() == 2) pcs.push_back(&n->head_genre[0]);            // 1 x 7: rows independent
    for (PackedConv* pc : pcs) {
        pc->bf_off = -1;
        if (pc_bf16_eligible(*pc)) { pc->bf_off = static_cast<long long>(count); count += static_cast<size_t>(pc->kh) * 4 * (pc->cout / 16) * 2 * 64; }
    }
    std::vector<PackedConv*> h1;                              // 32 -> 1 last convolutions of the key / tonic heads (2-conv heads only)
    if (n->head_key.size() == 2) h1.push_back(&n->head_key[1]);
    if (n->head_tonic.size() == 2) h1.push_back(&n->head_tonic[1]);
    if (n->head_genre.size() == 2) h1.push_back(&n->head_genre[1]);              // 2 x 7 over valid rows
    for (PackedConv* pc : h1) {
        pc->bf_off = -1;
        if ((pc->kh == 12 || pc->kh == 2) && pc->kw == 7 && pc->cout == 1 && pc->co == 1 && pc->cin == 32) { pc->bf_off = static_cast<long long>(count); count += static_cast<size_t>(pc->kh) * 22 * 2 * 64; }
    }
    if (!n->pc2pc.empty())                                   // layer 0's pitch-class stack (<= 4 channels): layer0_mfma_kernel
        for (PackedConv& pc : n->pc2pc[0]) {
            pc.l0_off = -1;
            if (n->cfg.num_layers > 1 && pc.kh == 12 && pc.kw == 7 && pc.co == 4 && pc.groups == 1 && pc.cout <= 4 && pc.cin <= 4) {
                pc.l0_off = static_cast<long long>(count); count += 24 * 64 + 64;     // ... and the 4 inverse channel scales
            }
        }
    for (size_t i = 1; i < n->semi.size(); ++i) {            // semitone convs that follow an 8-channel pitch stack: fused into its last conv
        PackedConv& pc = n->semi[i];
        pc.bf_off = -1;
        if (pc.cin == 8 && pc.co == 8 && pc.groups == 1 && i < n->p2p.size() && !n->p2p[i].empty() && n->p2p[i].back().bf_off >= 0) {
            pc.bf_off = static_cast<long long>(count); count += 6 * 64 + 64;      // ... and the 8 inverse channel scales
        }
    }
    if (count == 0) return AKE_OK;
    if (n->bf_frags_count != count) {
        if (n->bf_frags_dev) { (void)hipFree(n->bf_frags_dev); n->bf_frags_dev = nullptr; }
        AKE_HIP_CHECK(hipMalloc(&n->bf_frags_dev, count * sizeof(uint4)));
        n->bf_frags_count = count;
    }
    ake::ProfScope ps("pack_bf16_kernels", s);
    // train_only (ake_pcnet_load_for_training_f32, after every optimizer step): only the fragments the training-mode forward and the backward
    // read -- the inference kernels' (20 of the 40 launches; 0.2 ms of a 7 ms step) are rebuilt by the next full load
    n->eval_frags_stale = train_only;
    if (!train_only)
    for (const auto& layer : n->p2p)
        for (const PackedConv& pc : layer)
            if (pc.bf_off >= 0)
                hipLaunchKernelGGL(pack_p2p_f16_kernel, dim3((14 * 64 + 255) / 256), dim3(256), 0, s, n->blob_dev + pc.w_off, n->bf_frags_dev + pc.bf_off, pc.cin);
    if (!train_only)
    for (const PackedConv* pc : pcs)
        if (pc->bf_off >= 0) {
            const int NT = pc->cout / 16;
            hipLaunchKernelGGL(pack_pc_bf16_kernel, dim3((pc->kh * 4 * NT * 64 + 255) / 256), dim3(256), 0, s, n->blob_dev + pc->w_off,
                               n->bf_frags_dev + pc->bf_off, pc->cin, pc->cout, pc->co, NT, pc->kh);
        }
    if (!train_only && !n->pc2pc.empty())
        for (const PackedConv& pc : n->pc2pc[0])
            if (pc.l0_off >= 0)
                hipLaunchKernelGGL(pack_l0_f16_kernel, dim3(3), dim3(256), 0, s, n->blob_dev + pc.w_off, n->bf_frags_dev + pc.l0_off, pc.cin, pc.cout);
    {   // the training-mode packs (pitch-class stacks, their data gradients, the heads' data gradients; the pitch convs' raw forms): ONE launch per
        // kernel for all of them, the jobs in the kernel argument (28 dependent launches of a few microseconds of work each were 0.39 ms per step)
        std::vector<PcPackJob> jobs;
        auto add = [&](const PackedConv* pc, long long off, int NT, int dy_rot, int ci_off) {
            jobs.push_back(PcPackJob{n->blob_dev + pc->w_off, n->bf_frags_dev + off, pc->cin, pc->cout, pc->co, NT, pc->kh, dy_rot, ci_off});
        };
        for (const PackedConv* pc : tpc)
            if (pc->bf_off >= 0) add(pc, pc->bf_off, pc->cout / 16, 0, 0);
        for (const PackedConv* pc : tpd)
            if (pc->bf_off >= 0) add(pc, pc->bf_off, (pc->cout + 15) / 16, pc->kh - 1, 0);
        for (const PackedConv* pc : thd)
            if (pc->bf_off >= 0)
                for (int half = 0; half < 2; ++half) add(pc, half ? pc->bf_off2 : pc->bf_off, 1, pc->kh - 1, 16 * half);
        for (size_t j0 = 0; j0 < jobs.size(); j0 += kMaxPackJobs) {
            PcPackJobs js;
            js.n = static_cast<int>(std::min<size_t>(kMaxPackJobs, jobs.size() - j0));
            int max_cout = 1, max_blocks = 1;
            for (int k = 0; k < js.n; ++k) {
                js.j[k] = jobs[j0 + k];
                max_cout = std::max(max_cout, js.j[k].cout);
                max_blocks = std::max(max_blocks, (js.j[k].KH * 4 * js.j[k].NT * 64 + 255) / 256);
            }
            hipLaunchKernelGGL(pc_weight_scale_jobs_kernel, dim3(max_cout, js.n), dim3(256), 0, s, js);
            hipLaunchKernelGGL(pack_pc_f16x3_jobs_kernel, dim3(max_blocks, js.n), dim3(256), 0, s, js);
        }
        for (size_t j0 = 0; j0 < tfr.size(); j0 += kMaxPackJobs) {
            P2pRawJobs js;
            js.n = static_cast<int>(std::min<size_t>(kMaxPackJobs, tfr.size() - j0));
            for (int k = 0; k < js.n; ++k) {
                const TrainFrag& t = tfr[j0 + k];
                js.j[k] = P2pRawJob{n->blob_dev + t.raw, n->bf_frags_dev + t.pc->bf_off, t.cin, t.cout, t.flip};
            }
            hipLaunchKernelGGL(pack_p2p_f16_raw_jobs_kernel, dim3((14 * 64 + 255) / 256, js.n), dim3(256), 0, s, js);
        }
    }
    for (size_t i = 1; !train_only && i < n->semi.size(); ++i)
        if (n->semi[i].bf_off >= 0)
            hipLaunchKernelGGL(pack_semi_f16_kernel, dim3(1), dim3(192), 0, s, n->blob_dev + n->semi[i].w_off, n->bf_frags_dev + n->semi[i].bf_off);
    if (!train_only)
    for (const PackedConv* pc : h1)
        if (pc->bf_off >= 0)
            hipLaunchKernelGGL(pack_head1_bf16_kernel, dim3((pc->kh * 22 * 64 + 255) / 256), dim3(256), 0, s, n->blob_dev + pc->w_off,
                               n->bf_frags_dev + pc->bf_off, pc->cin, pc->kh);
    AKE_HIP_CHECK(hipGetLastError());
    return AKE_OK;
}

}  // namespace

int ake_pcnet_finalize(ake_pcnet* n) {
    AKE_REQUIRE(n, AKE_ERR_INVALID, "finalize: null handle");
    for (size_t i = 0; i < n->specs.size(); ++i)
        AKE_REQUIRE(n->host[i].set, AKE_ERR_STATE, "finalize: missing key '%s' (load_state_dict strict=True)", n->specs[i].name.c_str());
    int rc;
    if (n->map_host.empty()) {
        if ((rc = trace_maps(n))) return rc;
        if ((rc = build_fold_tables(n))) return rc;
    }
    reset_packs(n);
    build_packs(n, false);
    build_packs(n, true);
    n->blob.resize(ake::align_up(n->blob.size() + 64, 64), 0.f);
    AKE_REQUIRE(n->blob.size() == n->map_host.size(), AKE_ERR_STATE, "finalize: blob layout changed between builds");
    if (!n->blob_dev) AKE_HIP_CHECK(hipMalloc(&n->blob_dev, n->blob.size() * sizeof(float)));
    AKE_HIP_CHECK(hipMemcpy(n->blob_dev, n->blob.data(), n->blob.size() * sizeof(float), hipMemcpyHostToDevice));
    if ((rc = rebuild_bf16_frags(n, nullptr))) return rc;
    if (!n->aff_recs.empty()) {   // --denseblock: the tables come from the same kernel as in ake_pcnet_load_from_device_f32, fed a staged copy
        std::vector<float> flat(n->grad_floats);
        for (size_t i = 0; i < n->specs.size(); ++i) std::copy(n->host[i].data.begin(), n->host[i].data.end(), flat.begin() + n->grad_off[i]);
        float* tmp = nullptr;
        AKE_HIP_CHECK(hipMalloc(&tmp, flat.size() * sizeof(float)));
        AKE_HIP_CHECK(hipMemcpy(tmp, flat.data(), flat.size() * sizeof(float), hipMemcpyHostToDevice));
        rc = rebuild_dense_affine(n, tmp, nullptr);
        AKE_HIP_CHECK(hipStreamSynchronize(nullptr));
        (void)hipFree(tmp);
        if (rc) return rc;
    }
    AKE_HIP_CHECK(hipStreamSynchronize(nullptr));
    n->finalized = true;
    return AKE_OK;
}

// Device-resident parameters: (re)build every packed weight from a flat f32 parameter buffer on the device (layout =
// ake_pcnet_grad_offset).  No host round trip after the first call; asynchronous on `stream`.
namespace {
int load_from_device_impl(ake_pcnet* n, const float* params_dev, ake_stream_t stream, bool train_only);
}
int ake_pcnet_load_from_device_f32(ake_pcnet* n, const float* params_dev, ake_stream_t stream) { return load_from_device_impl(n, params_dev, stream, false); }
int ake_pcnet_load_for_training_f32(ake_pcnet* n, const float* params_dev, ake_stream_t stream) { return load_from_device_impl(n, params_dev, stream, true); }
namespace {
int load_from_device_impl(ake_pcnet* n, const float* params_dev, ake_stream_t stream, bool train_only) {
    AKE_REQUIRE(n && params_dev, AKE_ERR_INVALID, "load_from_device: null argument");
    if (n->map_host.empty()) {          // first use: learn the layout (values are irrelevant; host tensors only need their sizes)
        for (size_t i = 0; i < n->specs.size(); ++i) {
            size_t cnt = 1;
            for (int d = 0; d < n->specs[i].ndim; ++d) cnt *= static_cast<size_t>(n->specs[i].shape[d]);
            if (n->host[i].data.size() != cnt) n->host[i].data.assign(cnt, 0.f);
        }
        int rc;
        if ((rc = trace_maps(n))) return rc;
        if ((rc = build_fold_tables(n))) return rc;
        if (!n->blob_dev) AKE_HIP_CHECK(hipMalloc(&n->blob_dev, n->blob.size() * sizeof(float)));
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int P = static_cast<int>(n->grad_floats), NB = static_cast<int>(n->map_host.size());
    {
        ake::ProfScope ps("fold_params_kernel", s);
        hipLaunchKernelGGL(fold_params_kernel, dim3((P + 255) / 256), dim3(256), 0, s, params_dev, n->fold_ch_dev, n->fold_bn_dev, n->folded_dev, P);
    }
    {
        ake::ProfScope ps("gather_blob_kernel", s);
        hipLaunchKernelGGL(gather_blob_kernel, dim3((NB + 255) / 256), dim3(256), 0, s, params_dev, n->folded_dev, n->map_dev, n->blob_dev, NB);
    }
    AKE_HIP_CHECK(hipGetLastError());
    {
        int rc2 = rebuild_bf16_frags(n, s, train_only);
        if (!rc2) rc2 = rebuild_dense_affine(n, params_dev, s);
        if (rc2) return rc2;
    }
    n->finalized = true;
    return AKE_OK;
}
}  // namespace

// running_mean / running_var inside the flat parameter buffer <- momentum blend with the batch statistics that
// ake_pcnet_forward_train_f32 returned in bn_stats (torch semantics: unbiased variance; models use momentum 0.1).
int ake_pcnet_update_running_stats_f32(const ake_pcnet* n, const float* bn_stats_dev, float* params_dev, float momentum, ake_stream_t stream) {
    AKE_REQUIRE(n && bn_stats_dev && params_dev, AKE_ERR_INVALID, "update_running_stats: null argument");
    AKE_REQUIRE(n->finalized && n->run_off_dev, AKE_ERR_STATE, "update_running_stats: the handle has no parameters yet");
    hipStream_t s = static_cast<hipStream_t>(stream);
    ake::ProfScope ps("running_stats_kernel", s);
    hipLaunchKernelGGL(running_stats_kernel, dim3((n->bn_channels + 63) / 64), dim3(64), 0, s, bn_stats_dev, n->run_off_dev, params_dev, momentum,
                       n->bn_channels);
    AKE_HIP_CHECK(hipGetLastError());
    return AKE_OK;
}

// The BatchNorm layers whose forward the reference's BACKWARD pass runs a second time (checkpointed halves: every dense layer's norm1,
// models.py:484-489, 553): the same blend once more, with the batch statistics of the forward this backward belongs to.  Returns the number
// of channels it touches through *channels (nullable); a no-op for nets without such layers.
int ake_pcnet_update_recomputed_running_stats_f32(const ake_pcnet* n, const float* bn_stats_dev, float* params_dev, float momentum, int* channels,
                                                  ake_stream_t stream) {
    AKE_REQUIRE(n, AKE_ERR_INVALID, "update_recomputed_running_stats: null handle");
    if (channels) *channels = n->recomputed_bn_channels;
    if (n->recomputed_bn_channels == 0) return AKE_OK;
    AKE_REQUIRE(bn_stats_dev && params_dev, AKE_ERR_INVALID, "update_recomputed_running_stats: null argument");
    AKE_REQUIRE(n->finalized && n->run_off2_dev, AKE_ERR_STATE, "update_recomputed_running_stats: the handle has no parameters yet");
    hipStream_t s = static_cast<hipStream_t>(stream);
    ake::ProfScope ps("running_stats_kernel", s);
    hipLaunchKernelGGL(running_stats_kernel, dim3((n->bn_channels + 63) / 64), dim3(64), 0, s, bn_stats_dev, n->run_off2_dev, params_dev, momentum,
                       n->bn_channels);
    AKE_HIP_CHECK(hipGetLastError());
    return AKE_OK;
}

// BatchNorm layers of the training-mode forward, in forward order (valid after ake_pcnet_create; channel counts only
// need the configuration, names are the reference module paths, e.g. "model.1.p2p.layer.1").
int ake_pcnet_num_bn(const ake_pcnet* n) { return n ? static_cast<int>(n->bns.size()) : 0; }

int ake_pcnet_bn_info(const ake_pcnet* n, int index, const char** name, int* channels, int* channel_offset) {
    AKE_REQUIRE(n && index >= 0 && index < static_cast<int>(n->bns.size()), AKE_ERR_INVALID, "bn_info: index out of range (finalize first)");
    if (name) *name = n->bns[index].name.c_str();
    if (channels) *channels = n->bns[index].C;
    if (channel_offset) *channel_offset = n->bns[index].ch_off;
    return AKE_OK;
}

size_t ake_pcnet_workspace_bytes(const ake_pcnet* n, int batch, int frames) {
    if (!n || batch <= 0 || frames <= 0) return 0;
    Buffers b;
    if (plan_buffers(n, batch, std::min(batch, n->chunk_clips), frames, nullptr, &b) != AKE_OK) return 0;
    return b.bytes;
}

size_t ake_pcnet_train_workspace_bytes(const ake_pcnet* n, int batch, int frames) {
    if (!n || batch <= 0 || frames <= 0) return 0;
    Buffers b;
    if (plan_buffers(n, batch, batch, frames, nullptr, &b, true) != AKE_OK) return 0;   // batch statistics: no chunking
    return b.bytes;
}

namespace {

// One forward pass.  In eval mode BatchNorm is folded into the convolutions and every tensor is final.  In training
// mode a convolution followed by BatchNorm leaves its RAW output plus a pending (scale, shift, slope) table that the
// next reader applies while loading: `aff` travels next to every tensor pointer below (null = nothing pending).
struct Fwd {
    const ake_pcnet* n;
    Buffers& b;
    hipStream_t s;
    bool train;
    bool mel_fm = false;             // mel is frames-major [clip][T][P] (ake_pcnet_forward_frames_major_f32): only the fused default path can read it
    int chunk = 0;                   // clips per pitch_chunk call (the last one may be shorter)
    bool psix_f16 = false;           // layer 0's launch left layer 1's up_sixth map as f16 x 4 words in psix[1] (Layer0Args::psix_h), for conv_p2p_f16_ps_kernel<1, 3>

    int bn_of(const std::string& name) const { return n->bn_index.at(name); }

    // launch BatchNorm finalisation of layer `bn` (count values per channel) into the affine table `aff_out`
    void finalize_bn(int bn, double count, float* aff_out, float slope = kSlope) {
        const auto& l = n->bns[bn];
        ake::ProfScope ps("bn_finalize_kernel", s);
        hipLaunchKernelGGL(bn_finalize_kernel, dim3((l.C + 63) / 64), dim3(64), 0, s, b.stats + 2 * l.ch_off, 2 * n->bn_channels, count,
                           n->blob_dev + l.gamma_off, n->blob_dev + l.beta_off, aff_out, b.bstats + 3 * l.ch_off, l.C, slope);
        // the element count rides along for the unbiased running variance (written by the host-visible copy below)
    }
    void identity(float* aff, int C) {
        hipLaunchKernelGGL(affine_identity_kernel, dim3((C + 63) / 64), dim3(64), 0, s, aff, C);
    }

    // conv (+BatchNorm `bn_name` + LeakyReLU unless bn_name is empty).  Returns via *aff_out_used whether dst is raw.
    int conv(const PackedConv& pe, const PackedConv& pt, const std::string& bn_name, int kind, Src src, const float* in_aff,
             int B, int H, int T_in, bool same, float* dst, int ctot, int coff, float* aff_dst, const char* name, const float* residual = nullptr) {
        if (!train) return run_conv(n, pe, kind, src, B, H, T_in, same, !bn_name.empty(), dst, ctot, coff, s, name, nullptr, nullptr, nullptr, false, residual);
        const bool has_bn = !bn_name.empty();
        const int bn = has_bn ? bn_of(bn_name) : -1;
        int rc = run_conv(n, pt, kind, src, B, H, T_in, same, false, dst, ctot, coff, s, name, in_aff,
                          has_bn ? b.stats + 2 * n->bns[bn].ch_off : nullptr);
        if (rc) return rc;
        if (has_bn) {
            const int T_out = same ? T_in : T_in - pt.kw + 1;
            const int H_out = kind == 2 ? H - pt.kh + 1 : H;
            finalize_bn(bn, static_cast<double>(B) * H_out * T_out, aff_dst);
        }
        return AKE_OK;
    }

    // --denseblock (models.py:584-648), inference: one block IN PLACE on feat [B][ctot][H][T] whose channels [0, cin) are filled; layer j
    // reads channels [0, cin + j*nf) through its norm1 table, writes the bottleneck map to `bott`, and its k-wide convolution (input
    // through the norm2 table, ReLU) appends nf channels at cin + j*nf.  kind 0: plain Conv2d -- 1 x 1, then k x k ZERO-padded on both
    // axes; kind 1: equivariant 12 x 1 and 12 x k (rows circular, frames zero-padded).
    int dense_stack(const std::vector<DensePack>& packs, int kind, float* feat, int ctot, int cin, int B, int H, int T, float* bott, const char* label) {
        const int nf = n->cfg.n_filters, k = n->cfg.kernel_size;
        int rc;
        for (size_t j = 0; j < packs.size(); ++j) {
            const DensePack& dp = packs[j];
            const int cj = cin + static_cast<int>(j) * nf;
            AKE_REQUIRE(dp.c1.cin == cj && cj + nf <= ctot, AKE_ERR_STATE, "dense %s: channel bookkeeping", label);
            Src s1{feat, cj, nullptr, 0, 0, ctot};
            const ConvGeom g1{0, 3, T, H, 0};                  // the single tap sits at offset 3 of the 7 stored
            if ((rc = run_conv(n, dp.c1, kind, s1, B, H, T, true, false, bott, dp.c1.cout, 0, s, label, n->dense_aff_dev + dp.aff1, nullptr,
                               kind == 0 ? &g1 : nullptr)))
                return rc;
            const ConvGeom g2{k / 2, k / 2, T, H, 0};
            if ((rc = run_conv(n, dp.c2, kind, Src{bott, dp.c1.cout, nullptr, 0, 0}, B, H, T, true, false, feat, ctot, cj, s, label,
                               n->dense_aff_dev + dp.aff2, nullptr, kind == 0 ? &g2 : nullptr, false, nullptr, kind == 0)))
                return rc;
        }
        return AKE_OK;
    }

    // ... in TRAINING mode (batch statistics).  BatchNorm sits in FRONT of both convolutions here, so every norm1 needs the statistics of
    // features other kernels produced: the block input's are taken by channel_stats_kernel, a layer's new features leave theirs in the NEXT
    // layer's norm1 cells (conv2's statistics epilogue) and every later norm1 copies the prefix it shares (stats_copy_kernel).  The
    // bottleneck maps and both tables of every layer stay in the workspace for the backward pass.
    int dense_stack_train(const std::vector<DensePack>& packs, int kind, float* feat, int ctot, int cin, int B, int H, int T,
                          const std::vector<float*>& botts, const std::vector<float*>& aff1s, const std::vector<float*>& aff2s, const char* label) {
        const int nf = n->cfg.n_filters, k = n->cfg.kernel_size;
        const double count = static_cast<double>(B) * H * T;
        const int sstride = 2 * n->bn_channels;
        int rc;
        {
            ake::ProfScope ps("channel_stats_kernel", s);
            hipLaunchKernelGGL(channel_stats_kernel, dim3(cin, 1, B), dim3(256), 0, s, feat, static_cast<long long>(ctot) * H * T, static_cast<long long>(H) * T,
                               b.stats + 2 * n->bns[packs[0].bn1].ch_off, sstride);
        }
        for (size_t j = 0; j < packs.size(); ++j) {
            const DensePack& dp = packs[j];
            const int cj = cin + static_cast<int>(j) * nf;
            AKE_REQUIRE(dp.c1.cin == cj && cj + nf <= ctot && dp.bn1 >= 0 && dp.bn2 >= 0, AKE_ERR_STATE, "dense %s: channel bookkeeping (training)", label);
            if (j > 0) {   // the statistics of channels [0, cj - nf): the previous norm1 has them (its own copy or the block input's)
                const int C = cj - nf;
                ake::ProfScope ps("stats_copy_kernel", s);
                hipLaunchKernelGGL(stats_copy_kernel, dim3((kStatSlots * 2 * C + 255) / 256), dim3(256), 0, s, b.stats, sstride,
                                   n->bns[packs[j - 1].bn1].ch_off, n->bns[dp.bn1].ch_off, C);
            }
            finalize_bn(dp.bn1, count, aff1s[j], kSlope);                              // relu1 = nn.LeakyReLU
            Src s1{feat, cj, nullptr, 0, 0, ctot};
            const ConvGeom g1{0, 3, T, H, 0};
            if ((rc = run_conv(n, dp.c1, kind, s1, B, H, T, true, false, botts[j], dp.c1.cout, 0, s, label, aff1s[j], b.stats + 2 * n->bns[dp.bn2].ch_off,
                               kind == 0 ? &g1 : nullptr)))
                return rc;
            finalize_bn(dp.bn2, count, aff2s[j], 0.f);                                 // relu2 = nn.ReLU
            const ConvGeom g2{k / 2, k / 2, T, H, 0};
            double* next_stats = j + 1 < packs.size() ? b.stats + 2 * (n->bns[packs[j + 1].bn1].ch_off + cj) : nullptr;
            if ((rc = run_conv(n, dp.c2, kind, Src{botts[j], dp.c1.cout, nullptr, 0, 0}, B, H, T, true, false, feat, ctot, cj, s, label, aff2s[j], next_stats,
                               kind == 0 ? &g2 : nullptr, false, nullptr, kind == 0)))
                return rc;
        }
        return AKE_OK;
    }

    // --resblock stack (models.py:181-187 / 218-224, 402-454), inference: st = [conv0, (conv1, conv2) per block].  x lives in X
    // (C channels, dense), a block's hidden map (2C channels) in Hb; conv2 adds x before its LeakyReLU, in place -- the last block
    // may write channels [0, C) of a wider buffer instead (final_dst with final_ctot channels).
    int res_stack(const std::vector<PackedConv>& st, int kind, Src first, int B, int H, int T, float* X, float* Hb, float* final_dst,
                  int final_ctot, const char* label) {
        const int C = st[0].cout;
        int rc = conv(st[0], st[0], "bn", kind, first, nullptr, B, H, T, true, X, C, 0, nullptr, label);
        if (rc) return rc;
        const int nb = (static_cast<int>(st.size()) - 1) / 2;
        for (int r = 0; r < nb; ++r) {
            if ((rc = conv(st[1 + 2 * r], st[1 + 2 * r], "bn", kind, Src{X, C, nullptr, 0, 0}, nullptr, B, H, T, true, Hb, 2 * C, 0, nullptr, label))) return rc;
            const bool to_final = r == nb - 1 && final_dst;
            if ((rc = conv(st[2 + 2 * r], st[2 + 2 * r], "bn", kind, Src{Hb, 2 * C, nullptr, 0, 0}, nullptr, B, H, T, true, to_final ? final_dst : X,
                           to_final ? final_ctot : C, 0, nullptr, label, X)))
                return rc;
        }
        return AKE_OK;
    }

    // --resblock stack, training (batch statistics): st = the raw-weight packs [conv0, (conv1, conv2) per block], BatchNorms
    // `prefix`1 and `prefix`{3+r}.b1 / .b2.  Every tensor the backward needs stays: z = [conv0 raw, (conv1 raw (2C), conv2 raw,
    // block output) per block] with the pending tables in aff (b2's carries slope 1: no activation of its own; a block output's is
    // the identity).  The last block writes channels [0, C) of final_dst (final_ctot channels) instead of z.back() when given.
    int res_stack_train(const std::vector<PackedConv>& st, const std::string& prefix, int kind, Src first, const float* first_aff, int B, int H, int T,
                        const std::vector<float*>& z, const std::vector<float*>& aff, float* final_dst, int final_ctot, const char* label) {
        const int C = st[0].cout;
        const int nb = (static_cast<int>(st.size()) - 1) / 2;
        AKE_REQUIRE(static_cast<int>(z.size()) == 1 + 3 * nb && z.size() == aff.size(), AKE_ERR_STATE, "resblock %s: buffer bookkeeping", label);
        int rc = conv(st[0], st[0], prefix + "1", kind, first, first_aff, B, H, T, true, z[0], C, 0, aff[0], label);
        if (rc) return rc;
        const float* x = z[0];
        const float* x_aff = aff[0];
        const double count = static_cast<double>(B) * H * T;
        for (int r = 0; r < nb; ++r) {
            const std::string bp = prefix + std::to_string(3 + r) + ".";
            float* z1 = z[1 + 3 * r];
            float* z2 = z[2 + 3 * r];
            if ((rc = conv(st[1 + 2 * r], st[1 + 2 * r], bp + "b1", kind, Src{x, C, nullptr, 0, 0}, x_aff, B, H, T, true, z1, 2 * C, 0, aff[1 + 3 * r], label)))
                return rc;
            const int bn2 = bn_of(bp + "b2");
            if ((rc = run_conv(n, st[2 + 2 * r], kind, Src{z1, 2 * C, nullptr, 0, 0}, B, H, T, true, false, z2, C, 0, s, label, aff[1 + 3 * r],
                               b.stats + 2 * n->bns[bn2].ch_off)))
                return rc;
            finalize_bn(bn2, count, aff[2 + 3 * r], 1.f);
            const bool to_final = r == nb - 1 && final_dst;
            float* xo = to_final ? final_dst : z[3 + 3 * r];
            {
                const long long total = static_cast<long long>(B) * C * H * T;
                ake::ProfScope ps("res_add_act_kernel", s);
                hipLaunchKernelGGL(res_add_act_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, s, z2, aff[2 + 3 * r], x, x_aff, xo, C,
                                   H * T, to_final ? final_ctot : C, total);
            }
            identity(aff[3 + 3 * r], C);
            x = xo; x_aff = aff[3 + 3 * r];
        }
        return AKE_OK;
    }

    // pool_semi (+BN+LReLU) + octave fold of `src` [B][C][P][T] into channels [coff, coff+C) of dst [B][ctot][12][T]
    int semi(int layer, const float* src, const float* in_aff, int B, int P, int Tn, float* dst, int ctot, int coff, float* aff_cat_rows) {
        const char* nm = layer == 0 ? "semi_fold_kernel/L0" : "semi_fold_kernel/L1+";
        if (!train && n->cfg.p2pc_conv) {   // semitone conv (raw, BatchNorm folded) -> octave-fold convolution (LeakyReLU applied on load)
            const PackedConv& pc = n->semi[layer];
            SemiTrainArgs ta;
            std::memset(&ta, 0, sizeof(ta));
            SemiArgs& a = ta.s;
            a.src = src; a.C = pc.cin; a.H = P; a.T = Tn;
            a.src_clip_stride = static_cast<long long>(pc.cin) * P * Tn;
            a.w = n->blob_dev + pc.w_off; a.bias = n->blob_dev + pc.b_off;
            a.dst = b.smap; a.n_strips = (Tn + TW - 1) / TW;
            const int per_clip = (P / 3) * a.n_strips;
            const int threads = per_clip >= 256 ? 256 : (per_clip + 63) / 64 * 64;
            dim3 grid((per_clip + threads - 1) / threads, pc.groups, B), block(threads);
            {
                ake::ProfScope ps("semi_conv_stats_kernel", s);
                switch (pc.co) {
                    case 8: hipLaunchKernelGGL((semi_conv_stats_kernel<8>), grid, block, 0, s, ta); break;
                    case 4: hipLaunchKernelGGL((semi_conv_stats_kernel<4>), grid, block, 0, s, ta); break;
                    case 1: hipLaunchKernelGGL((semi_conv_stats_kernel<1>), grid, block, 0, s, ta); break;
                    default: ake::set_error("semi: bad CO"); return AKE_ERR_UNSUPPORTED;
                }
            }
            const PackedConv& fc = n->foldc[layer];
            const long long total = static_cast<long long>(B) * pc.cin * 12 * Tn;
            ake::ProfScope ps("fold_conv_kernel", s);
            hipLaunchKernelGGL(fold_conv_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, s, b.smap, n->blob_dev + fc.w_off,
                               n->blob_dev + fc.b_off, dst, pc.cin, P / 36, Tn, 1, static_cast<long long>(ctot) * 12 * Tn, coff, total);
            return AKE_OK;
        }
        if (!train) return run_semi(n, n->semi[layer], src, B, P, Tn, dst, ctot, coff, s, nm);
        const PackedConv& pc = n->semi_t[layer];
        const int bn = bn_of("model." + std::to_string(layer) + ".pool_semi_b");
        SemiTrainArgs ta;
        std::memset(&ta, 0, sizeof(ta));
        SemiArgs& a = ta.s;
        a.src = src; a.C = pc.cin; a.H = P; a.T = Tn;
        a.src_clip_stride = static_cast<long long>(pc.cin) * P * Tn;
        a.w = n->blob_dev + pc.w_off; a.bias = n->blob_dev + pc.b_off;
        a.dst = b.semi_raw[layer]; a.dst_coff = 0; a.dst_clip_stride = 0;
        a.n_strips = (Tn + TW - 1) / TW;
        ta.in_affine = in_aff; ta.stats = b.stats + 2 * n->bns[bn].ch_off; ta.stats_stride = 2 * n->bn_channels;
        const int per_clip = (P / 3) * a.n_strips;            // thread = (semitone row, strip of frames)
        const int threads = per_clip >= 256 ? 256 : (per_clip + 63) / 64 * 64;
        dim3 grid((per_clip + threads - 1) / threads, pc.groups, B), block(threads);
        {
            ake::ProfScope ps("semi_conv_stats_kernel", s);
            switch (pc.co) {
                case 8: hipLaunchKernelGGL((semi_conv_stats_kernel<8>), grid, block, 0, s, ta); break;
                case 4: hipLaunchKernelGGL((semi_conv_stats_kernel<4>), grid, block, 0, s, ta); break;
                case 1: hipLaunchKernelGGL((semi_conv_stats_kernel<1>), grid, block, 0, s, ta); break;
                default: ake::set_error("semi: bad CO"); return AKE_ERR_UNSUPPORTED;
            }
        }
        finalize_bn(bn, static_cast<double>(B) * (P / 3) * Tn, b.aff_semi[layer]);
        const long long total = static_cast<long long>(B) * pc.cin * 12 * Tn;
        if (n->cfg.p2pc_conv) {   // models.py:108-133: the fold is a learned convolution over the octaves + BatchNorm (batch statistics) + LeakyReLU
            const PackedConv& fc = n->foldc_t[layer];
            const int bnf = bn_of("model." + std::to_string(layer) + ".pool.bn");
            AKE_REQUIRE(pc.cin <= 64, AKE_ERR_UNSUPPORTED, "p2pc_conv training: %d channels", pc.cin);
            {
                ake::ProfScope ps("fold_conv_kernel", s);
                hipLaunchKernelGGL(fold_conv_train_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, s, b.semi_raw[layer],
                                   b.aff_semi[layer], n->blob_dev + fc.w_off, n->blob_dev + fc.b_off, b.foldc_raw[layer],
                                   b.stats + 2 * n->bns[bnf].ch_off, 2 * n->bn_channels, pc.cin, P / 36, Tn, total);
            }
            finalize_bn(bnf, static_cast<double>(B) * 12 * Tn, b.aff_foldc[layer]);
            {   // the folded channels are materialised as final activations: every reader of the concat buffer takes them as they are
                ake::ProfScope ps("apply_affine_kernel", s);
                hipLaunchKernelGGL(apply_affine_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, s, b.foldc_raw[layer],
                                   b.aff_foldc[layer], dst, pc.cin, static_cast<long long>(12) * Tn, ctot, coff, total);
            }
        } else {
            ake::ProfScope ps("fold_affine_kernel", s);
            hipLaunchKernelGGL(fold_affine_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, s, b.semi_raw[layer],
                               b.aff_semi[layer], dst, pc.cin, P / 36, Tn, ctot, coff, total);
        }
        if (aff_cat_rows) identity(aff_cat_rows, pc.cin);     // the folded channels are final activations
        return AKE_OK;
    }

    void up_sixth(int layer, const float* src, long long src_clip_stride, const float* in_aff, int B, int C, int Tn, float* dst, float* aff_dst) {
        const long long total = static_cast<long long>(B) * C * 36 * Tn;
        if (!train) {
            ake::ProfScope ps("up_sixth_kernel", s);
            hipLaunchKernelGGL(up_sixth_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, s, src, src_clip_stride,
                               n->blob_dev + n->up[layer].w_off, n->blob_dev + n->up[layer].b_off, dst, C, Tn, total);
            return;
        }
        const int bn = bn_of("model." + std::to_string(layer) + ".up_sixth_b");
        {
            ake::ProfScope ps("up_sixth_train_kernel", s);
            hipLaunchKernelGGL(up_sixth_train_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, s, src, src_clip_stride,
                               in_aff, n->blob_dev + n->up_t[layer].w_off, n->blob_dev + n->up_t[layer].b_off, dst,
                               b.stats + 2 * n->bns[bn].ch_off, 2 * n->bn_channels, C, Tn, total);
        }
        finalize_bn(bn, static_cast<double>(B) * 36 * Tn, aff_dst);
    }

    void time_pool(const float* src, const float* in_aff, int B, int C, int H, int Tn, float* dst, int ctot, int coff) {
        const int tp = n->cfg.time_pool_size;
        const long long total = static_cast<long long>(B) * C * H * (Tn / tp);
        ake::ProfScope ps("time_pool_kernel", s);
        if (train && in_aff)
            hipLaunchKernelGGL(time_pool_affine_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, s, src, in_aff, dst, C, H,
                               Tn, tp, ctot, coff, total);
        else
            hipLaunchKernelGGL(time_pool_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, s, src, dst, C, H, Tn, tp, ctot,
                               coff, total);
    }

    // inference: semitone conv + BN + LeakyReLU of `src` [B][C][P][T] as a map of its own [B][C][P / 3][T] (raw = before the LeakyReLU)
    int semi_map(int layer, const float* src, int B, int P, int Tn, float* dst, bool raw) {
        const PackedConv& pc = n->semi[layer];
        SemiTrainArgs ta;
        std::memset(&ta, 0, sizeof(ta));
        SemiArgs& a = ta.s;
        a.src = src; a.C = pc.cin; a.H = P; a.T = Tn;
        a.src_clip_stride = static_cast<long long>(pc.cin) * P * Tn;
        a.w = n->blob_dev + pc.w_off; a.bias = n->blob_dev + pc.b_off;
        a.dst = dst; a.n_strips = (Tn + TW - 1) / TW;
        ta.out_lrelu = raw ? 0 : 1;
        const int per_clip = (P / 3) * a.n_strips;
        const int threads = per_clip >= 256 ? 256 : (per_clip + 63) / 64 * 64;
        dim3 grid((per_clip + threads - 1) / threads, pc.groups, B), block(threads);
        ake::ProfScope ps("semi_conv_stats_kernel", s);
        switch (pc.co) {
            case 8: hipLaunchKernelGGL((semi_conv_stats_kernel<8>), grid, block, 0, s, ta); break;
            case 4: hipLaunchKernelGGL((semi_conv_stats_kernel<4>), grid, block, 0, s, ta); break;
            case 1: hipLaunchKernelGGL((semi_conv_stats_kernel<1>), grid, block, 0, s, ta); break;
            default: ake::set_error("semi: bad CO"); return AKE_ERR_UNSUPPORTED;
        }
        return AKE_OK;
    }

    // the octave fold of ready maps [B][C][S][T] into channels [coff, coff + C) of dst: max (models.py:95-106) or --p2pc_conv's convolution
    int fold_maps(int layer, const float* maps, int C, int S, int B, int Tn, float* dst, int ctot, int coff) {
        if (!n->cfg.p2pc_conv) return run_fold_max(maps, C, S, B, Tn, dst, ctot, coff, s);
        const PackedConv& fc = n->foldc[layer];
        const long long total = static_cast<long long>(B) * C * 12 * Tn;
        ake::ProfScope ps("fold_conv_kernel", s);
        hipLaunchKernelGGL(fold_conv_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, s, maps, n->blob_dev + fc.w_off,
                           n->blob_dev + fc.b_off, dst, C, S / 12, Tn, 0, static_cast<long long>(ctot) * 12 * Tn, coff, total);
        return AKE_OK;
    }

    // inference, default family: the whole of phase A as one launch, one workgroup per clip (layer0_fused_kernel)
    bool layer0_fused(const float* mel, int B, bool dry_run = false) {
        static const bool off = ake::diag_env("AKE_L0_FUSED") != nullptr && std::atoi(ake::diag_env("AKE_L0_FUSED")) == 0;
        const auto& c = n->cfg;
        const int P = c.pitches, T0 = b.Tl[0], NF = c.n_filters;
        if (off || c.resblock || c.denseblock || c.p2pc_conv || c.stay_sixth || NF < 2 || NF > 4 || c.conv_layers < 1 || c.conv_layers > 4 || c.kernel_size != 7 || P % 36 || T0 < 1) return false;
        const PackedConv& sp = n->semi[0];
        if (sp.cin != 1 || sp.co != 1) return false;
        for (int j = 0; j < c.conv_layers; ++j) {
            const PackedConv& pc = n->pc2pc[0][j];
            if (pc.co != 4 || pc.groups != 1 || pc.kh != 12 || pc.kw != 7 || pc.cout != NF || pc.cin != (j == 0 ? 1 : NF)) return false;
        }
        const LayerDims& d1 = n->dims[1];
        if (d1.prev_pc != NF) return false;
        Layer0Args a;
        std::memset(&a, 0, sizeof(a));
        a.RP = 4 * ((T0 + 3) / 4) + 8;
        const size_t lds = (static_cast<size_t>(9) * 12 * a.RP + static_cast<size_t>(P) * T0) * sizeof(float);   // maps + the clip's CQT
        if (lds > 150 * 1024 || (static_cast<long long>(P) * T0) % 4 || (reinterpret_cast<uintptr_t>(mel) & 15)) return false;
        a.mel = mel; a.sw = n->blob_dev + sp.w_off; a.sb = n->blob_dev + sp.b_off;
        const int ctot1 = d1.prev_pc + d1.out_p;
        for (int j = 0; j < c.conv_layers; ++j) {
            const PackedConv& pc = n->pc2pc[0][j];
            const bool lastj = j == c.conv_layers - 1;
            a.w[j] = n->blob_dev + pc.w_off; a.b[j] = n->blob_dev + pc.b_off;
            a.dst[j] = lastj ? b.cat[1] : ((j & 1) ? b.pcb[0] : b.pca[0]);
            a.dst_clip_stride[j] = static_cast<long long>(lastj ? ctot1 : NF) * 12 * T0;
        }
        a.uw = n->blob_dev + n->up[1].w_off; a.ub = n->blob_dev + n->up[1].b_off;
        a.fold0 = b.fold0; a.psix = b.psix[1];
        a.H = P; a.T = T0; a.NF = NF; a.n_conv = c.conv_layers;
        static ake::DeviceOnce attr_set;
        if (attr_set.need()) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(layer0_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(layer0_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess)
                return false;
            attr_set.mark();
        }
        // the convolution stack on bf16 MFMA when the fragments exist and the maps fit (AKE_PC_F32=1: the exact-f32 VALU form)
        bool mfma = !g_pc_f32_only && c.precision == AKE_PRECISION_MIXED;      // (the MFMA form multiplies f16 x f16)
        for (int j = 0; j < c.conv_layers; ++j) {
            mfma = mfma && n->pc2pc[0][j].l0_off >= 0;
            if (mfma) a.frag[j] = n->bf_frags_dev + n->pc2pc[0][j].l0_off;
        }
        a.RPp = (T0 + 8 + 1) / 2 * 2;
        const size_t lds_m = (static_cast<size_t>(4) * 12 * a.RP + static_cast<size_t>(2) * 12 * a.RPp * 4 + static_cast<size_t>(P) * T0) * sizeof(float);
        const bool take_mfma = mfma && lds_m <= 150 * 1024;
        if (mel_fm && (!take_mfma || P % 4)) return false;        // only the MFMA form's loader transposes
        if (dry_run) return true;
        a.mel_fm = mel_fm ? 1 : 0;
        a.taps = g_keep_taps ? 1 : 0;
        // the up_sixth map as f16 x 4 when the conv that reads it is the persistent f16 kernel for every chunk of this batch (it rounds to f16
        // itself otherwise: same values); psix[1]'s buffer holds either form
        static const bool uh_off = ake::diag_env("AKE_PSIX_F32") != nullptr;
        psix_f16 = false;
        if (take_mfma && b.melh && !uh_off && !g_keep_taps && !c.pc2p_mem && d1.prev_p == 1 && chunk > 0 && p2p_uses_f16(n, 1, b.Tl[1])) {
            Src sd{mel, 1, b.psix[1], d1.prev_pc, 36};
            unsigned short* oh = reinterpret_cast<unsigned short*>(b.pa[1]);
            psix_f16 = run_p2p_f16_ps(n, n->p2p[1][0], nullptr, &sd, std::min(B, chunk), P, b.Tl[1], nullptr, d1.out_p, oh, nullptr, s, "", mel_fm, true, -1, true) &&
                       (B <= chunk || B % chunk == 0 ||
                        run_p2p_f16_ps(n, n->p2p[1][0], nullptr, &sd, B % chunk, P, b.Tl[1], nullptr, d1.out_p, oh, nullptr, s, "", mel_fm, true, -1, true));
        }
        if (psix_f16) { a.psix_h = reinterpret_cast<uint2*>(b.psix[1]); a.psix = nullptr; a.melh = b.melh; }
        static const int dbg_skip = ake::diag_env("AKE_L0_SKIP") ? std::atoi(ake::diag_env("AKE_L0_SKIP")) : 0;   // timing experiments only (wrong results)
        if (dbg_skip & 1) a.n_conv = 0;
        if (dbg_skip & 2) { a.psix = nullptr; a.psix_h = nullptr; }
        ake::ProfScope ps("layer0_fused_kernel", s);
        if (take_mfma) hipLaunchKernelGGL(layer0_mfma_kernel, dim3(B), dim3(512), lds_m, s, a);
        else hipLaunchKernelGGL(layer0_fused_kernel, dim3(B), dim3(512), lds, s, a);
        return true;
    }

    // Phase A, whole batch: layer 0 (models.py:361-369) and layer 1's up_sixth (models.py:372-374).
    int entry(const float* mel, int B) {
        const auto& c = n->cfg;
        const int L = c.num_layers, P = c.pitches, T0 = b.Tl[0];
        int rc;
        if (!train && L > 1 && layer0_fused(mel, B)) return AKE_OK;
        AKE_REQUIRE(!mel_fm, AKE_ERR_UNSUPPORTED, "pcnet: this configuration does not take the frames-major input (ake_pcnet_accepts_frames_major)");
        if (c.stay_sixth && L > 1 && train) {   // training: the RAW semitone map (semi_raw[0]) + its pending table are the pitch stream
            if ((rc = semi(0, mel, nullptr, B, P, T0, b.fold0, 1, 0, nullptr))) return rc;
        } else if (c.stay_sixth && L > 1) {   // models.py:366-367: the activated semitone map is the pitch stream from here on; its fold feeds pc2pc
            if ((rc = semi_map(0, mel, B, P, T0, b.p0, false))) return rc;
            if ((rc = fold_maps(0, b.p0, 1, P / 3, B, T0, b.fold0, 1, 0))) return rc;
        } else if (c.denseblock) {   // the fold is channel 0 of the block's feature buffer (L == 1: fold0 itself, else layer 1's concat buffer)
            const int g = c.n_filters * c.conv_layers;
            const int ctot1 = L == 1 ? 1 + g : n->dims[1].prev_pc + n->dims[1].out_p + g;
            float* feat = L == 1 ? b.fold0 : b.cat[1];
            if ((rc = semi(0, mel, nullptr, B, P, T0, feat, ctot1, 0, nullptr))) return rc;
            if (L == 1) return AKE_OK;                           // its block runs in the tail
            if (train) {
                if ((rc = dense_stack_train(n->dense_pc[0], 1, feat, ctot1, 1, B, 12, T0, b.dn_bott_pc[0], b.dn_aff1_pc[0], b.dn_aff2_pc[0], "conv_mfma_kernel/pc2pc0")))
                    return rc;
                // the raw up_sixth map + its table (rows [prev_p, ..) of the pitch block's input table, as in the default net)
                up_sixth(1, feat, static_cast<long long>(ctot1) * 12 * T0, nullptr, B, n->dims[1].prev_pc, T0, b.psix[1], b.aff_p2pin[1] + 3 * n->dims[1].prev_p);
                return AKE_OK;
            }
            if ((rc = dense_stack(n->dense_pc[0], 1, feat, ctot1, 1, B, 12, T0, b.pca[0], "conv_mfma_kernel/pc2pc0"))) return rc;
            up_sixth(1, feat, static_cast<long long>(ctot1) * 12 * T0, nullptr, B, n->dims[1].prev_pc, T0, b.psix[1], nullptr);
            return AKE_OK;
        } else if ((rc = semi(0, mel, nullptr, B, P, T0, b.fold0, 1, 0, nullptr))) return rc;
        if (L == 1) return AKE_OK;                               // its pc2pc runs in the tail
        const LayerDims& d1 = n->dims[1];
        const int ctot1 = d1.prev_pc + d1.out_p;
        const float* src = b.fold0;
        const float* src_aff = nullptr;
        int cin = 1;
        const std::string m = "model.0.pc2pc.layer.";
        if (c.resblock && train) {   // the last block's output = channels [0, nf) of layer 1's concat buffer, final (identity table)
            if ((rc = res_stack_train(n->pc2pc_t[0], m, 1, Src{b.fold0, 1, nullptr, 0, 0}, nullptr, B, 12, T0, b.pcst[0], b.aff_pcst[0], b.cat[1], ctot1,
                                      "conv_mfma_kernel/pc2pc0")))
                return rc;
            identity(b.aff_cat[1], c.n_filters);
        } else if (c.resblock) {
            if ((rc = res_stack(n->pc2pc[0], 1, Src{b.fold0, 1, nullptr, 0, 0}, B, 12, T0, b.pca[0], b.pcb[0], b.cat[1], ctot1, "conv_mfma_kernel/pc2pc0"))) return rc;
        }
        for (int j = 0; j < c.conv_layers && !c.resblock; ++j) {
            const bool lastj = j == c.conv_layers - 1;           // the last conv writes channels [0, nf) of layer 1's concat buffer
            float* dst = lastj ? b.cat[1] : (train ? b.pcst[0][j] : ((j & 1) ? b.pcb[0] : b.pca[0]));
            float* aff = !train ? nullptr : (lastj ? b.aff_cat[1] : b.aff_pcst[0][j]);
            if ((rc = conv(n->pc2pc[0][j], train ? n->pc2pc_t[0][j] : n->pc2pc[0][j], m + std::to_string(3 * j + 1), 1,
                           Src{src, cin, nullptr, 0, 0}, src_aff, B, 12, T0, true, dst, lastj ? ctot1 : c.n_filters, 0, aff,
                           "conv_mfma_kernel/pc2pc0")))
                return rc;
            src = dst; src_aff = aff; cin = c.n_filters;
        }
        if (c.stay_sixth) return AKE_OK;                         // no up_sixth: the pitch classes are repeated as they are
        // psix's BatchNorm lands in rows [prev_p, ..) of the pitch-conv input table (row 0.. = the pitch stream itself)
        if (train) identity(b.aff_p2pin[1], d1.prev_p);
        up_sixth(1, b.cat[1], static_cast<long long>(ctot1) * 12 * T0, train ? b.aff_cat[1] : nullptr, B, d1.prev_pc, T0, b.psix[1],
                 train ? b.aff_p2pin[1] + 3 * d1.prev_p : nullptr);
        return AKE_OK;
    }

    // Phase B, per chunk of clips [c0, c0+B): pitch convs -> semitone fold for every layer >= 1 (plus pc2pc / pooling /
    // the next up_sixth for the inner layers of deeper nets).
    int pitch_chunk(const float* mel, int c0, int B) {
        const auto& c = n->cfg;
        const int L = c.num_layers, tp = c.time_pool_size;
        const int P = c.stay_sixth ? c.pitches / 3 : c.pitches;  // rows of the pitch stream (--stay_sixth: semitones)
        int rc;
        // pitch stream [B][cp][P][T]: a final activation, except --stay_sixth in training (layer 0's raw semitone map + aff_semi[0])
        const float* p_cur = c.stay_sixth ? (train ? b.semi_raw[0] : b.p0 + static_cast<size_t>(c0) * P * b.Tl[0]) : mel;
        int cp = 1;
        const float* pc_cur = nullptr;
        for (int i = 1; i < L; ++i) {
            const int Ti = b.Tl[i];
            const LayerDims& d = n->dims[i];
            const bool last = i == L - 1;
            if (c.denseblock) {   // models.py:370-396 with dense stacks: both blocks grow their concat buffers in place
                const int g = c.n_filters * c.conv_layers;
                const int ctd = d.prev_pc + d.out_p + g;                             // = d.out_pc
                float* catd = b.cat[i] + (last || i == 1 ? static_cast<size_t>(c0) * ctd * 12 * Ti : 0);
                float* psixd = b.psix[i] + (i == 1 ? static_cast<size_t>(c0) * d.prev_pc * 36 * Ti : 0);
                if (i > 1) up_sixth(i, pc_cur, static_cast<long long>(ctd) * 12 * Ti, nullptr, B, d.prev_pc, Ti, psixd, train ? b.aff_p2pin[i] + 3 * d.prev_p : nullptr);
                float* fp = b.pa[i];                                                  // [B][out_p][P][Ti]
                {
                    const long long total = static_cast<long long>(B) * (cp + d.prev_pc) * P * Ti;
                    ake::ProfScope ps("concat_repeat_kernel", s);
                    hipLaunchKernelGGL(concat_repeat_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, s, p_cur, cp, psixd, d.prev_pc, 36,
                                       fp, d.out_p, P, Ti, total, train ? b.aff_p2pin[i] + 3 * d.prev_p : nullptr);
                }
                if (train) {
                    if ((rc = dense_stack_train(n->dense_p[i], 0, fp, d.out_p, cp + d.prev_pc, B, P, Ti, b.dn_bott_p[i], b.dn_aff1_p[i], b.dn_aff2_p[i], "conv_mfma_kernel/p2p")))
                        return rc;
                } else if ((rc = dense_stack(n->dense_p[i], 0, fp, d.out_p, cp + d.prev_pc, B, P, Ti, b.pb[i], "conv_mfma_kernel/p2p"))) return rc;
                if ((rc = semi(i, fp, nullptr, B, P, Ti, catd, ctd, d.prev_pc, nullptr))) return rc;
                if (last) return AKE_OK;                                              // its pitch-class block + pooling + heads run batch-wide
                if (train) {
                    if ((rc = dense_stack_train(n->dense_pc[i], 1, catd, ctd, d.prev_pc + d.out_p, B, 12, Ti, b.dn_bott_pc[i], b.dn_aff1_pc[i], b.dn_aff2_pc[i],
                                                "conv_mfma_kernel/pc2pc")))
                        return rc;
                } else if ((rc = dense_stack(n->dense_pc[i], 1, catd, ctd, d.prev_pc + d.out_p, B, 12, Ti, b.pca[i], "conv_mfma_kernel/pc2pc"))) return rc;
                const LayerDims& dn = n->dims[i + 1];
                const int ctn = dn.prev_pc + dn.out_p + g;
                const int Tn = Ti / tp;
                float* catn = b.cat[i + 1] + (i + 1 == L - 1 ? static_cast<size_t>(c0) * ctn * 12 * Tn : 0);
                time_pool(catd, nullptr, B, ctd, 12, Ti, catn, ctn, 0);
                time_pool(fp, nullptr, B, d.out_p, P, Ti, b.ppool[i], d.out_p, 0);
                pc_cur = catn; p_cur = b.ppool[i]; cp = d.out_p;
                continue;
            }
            const int ctot = d.prev_pc + d.out_p;
            const std::string m = "model." + std::to_string(i) + ".";
            float* cat = b.cat[i] + (last || i == 1 ? static_cast<size_t>(c0) * ctot * 12 * Ti : 0);
            float* psix = b.psix[i] + (i == 1 ? static_cast<size_t>(c0) * d.prev_pc * 36 * Ti : 0);
            if (i > 1 && !c.stay_sixth) {   // layer 1's up_sixth ran batch-wide in entry(); pc_cur = pooled (final) features here
                if (train) identity(b.aff_p2pin[i], d.prev_p);
                up_sixth(i, pc_cur, static_cast<long long>(ctot) * 12 * Ti, nullptr, B, d.prev_pc, Ti, psix,
                         train ? b.aff_p2pin[i] + 3 * d.prev_p : nullptr);
            }
            // models.py:378-384  repeat + concat (never materialised) + pitch convs
            Src sdesc{p_cur, cp, psix, d.prev_pc, 36};
            if (c.stay_sixth) {   // models.py:322-323, 379-383: the pitch classes themselves, repeated over the octaves (a dense copy: they live
                                  // in channels [0, prev_pc) of a concat buffer)
                const float* pcs = i == 1 ? b.cat[1] + static_cast<size_t>(c0) * ctot * 12 * Ti : pc_cur;
                const long long per_clip = static_cast<long long>(d.prev_pc) * 12 * Ti, total = per_clip * B;
                {
                    ake::ProfScope ps("slice_channels_kernel", s);
                    hipLaunchKernelGGL(slice_channels_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, s, pcs,
                                       static_cast<long long>(ctot) * 12 * Ti, b.pcd[i], per_clip, total);
                }
                sdesc = Src{p_cur, cp, b.pcd[i], d.prev_pc, 12};
                if (train) {   // the stack's input table: the pitch stream's rows, then the pitch classes' (they were copied raw)
                    if (i == 1) AKE_HIP_CHECK(hipMemcpyAsync(b.aff_p2pin[i], b.aff_semi[0], sizeof(float) * 3 * cp, hipMemcpyDeviceToDevice, s));
                    else identity(b.aff_p2pin[i], cp);
                    AKE_HIP_CHECK(hipMemcpyAsync(b.aff_p2pin[i] + 3 * cp, b.aff_cat[i], sizeof(float) * 3 * d.prev_pc, hipMemcpyDeviceToDevice, s));
                }
            }
            if (c.pc2p_mem) {   // models.py:376-377: no concat, the summed up_sixth map is added to the pitch stream
                const long long total = static_cast<long long>(B) * cp * P * Ti;
                ake::ProfScope ps("pc2p_mem_kernel", s);
                // (training: psix is raw, its up_sixth_b table sits behind the pitch stream's identity rows; the sum itself is final)
                hipLaunchKernelGGL(pc2p_mem_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, s, p_cur, psix, b.pin[i], cp,
                                   d.prev_pc / cp, P, Ti, total, train ? b.aff_p2pin[i] + 3 * d.prev_p : nullptr);
                sdesc = Src{b.pin[i], cp, nullptr, 0, 0};
            }
            const float* in_aff = train ? b.aff_p2pin[i] : nullptr;
            float* out = nullptr;
            float* out_aff = nullptr;
            // inference: the stack runs on f16 MFMA (f16 activations x hi + lo f16 weights, see conv_p2p_f16_kernel); the activations
            // between its convs are ONE channels-last f16 plane (16 B per position) in the same ping-pong buffers
            const bool bf = !train && p2p_uses_f16(n, i, Ti);
            AKE_REQUIRE(!mel_fm || (bf && i == 1 && L == 2 && !c.resblock && !c.pc2p_mem && !c.stay_sixth), AKE_ERR_UNSUPPORTED,
                        "pcnet: this configuration does not take the frames-major input (ake_pcnet_accepts_frames_major)");
            bool fused_semi = false, fused_fold = false;
            if (c.resblock && train) {
                if ((rc = res_stack_train(n->p2p_t[i], m + "p2p.layer.", 0, sdesc, in_aff, B, P, Ti, b.pst[i], b.aff_pst[i], nullptr, 0, "conv_mfma_kernel/p2p")))
                    return rc;
                out = b.pst[i].back(); out_aff = b.aff_pst[i].back();
            } else if (c.resblock) {
                if ((rc = res_stack(n->p2p[i], 0, sdesc, B, P, Ti, b.pa[i], b.pb[i], nullptr, 0, "conv_mfma_kernel/p2p"))) return rc;
                out = b.pa[i];
            }
            for (int j = 0; j < c.conv_layers && !c.resblock; ++j) {
                out = train ? b.pst[i][j] : ((j & 1) ? b.pb[i] : b.pa[i]);
                out_aff = !train ? nullptr : b.aff_pst[i][j];
                if (bf) {
                    unsigned short* oh = reinterpret_cast<unsigned short*>(out);
                    const bool last_conv = j == c.conv_layers - 1;
                    if (j == 0) {   // the stack's input (pitch stream | repeated up_sixth output) is assembled by the kernel's own loader
                        if (i == 1 && psix_f16) {   // layer 0 left the up_sixth map as f16 x 4 words (8 bytes per position, clip stride 36 T words)
                            Src sh{reinterpret_cast<const float*>(b.melh + static_cast<size_t>(c0) * P * Ti), 1,
                                   reinterpret_cast<const float*>(reinterpret_cast<const uint2*>(b.psix[1]) + static_cast<size_t>(c0) * 36 * Ti), d.prev_pc, 36};
                            AKE_REQUIRE(run_p2p_f16_ps(n, n->p2p[i][0], nullptr, &sh, B, P, Ti, nullptr, d.out_p, oh, nullptr, s, "conv_p2p_f16_kernel", false, false, -1, true),
                                        AKE_ERR_STATE, "pcnet: the f16 up_sixth map has no reader for this chunk (%d clips)", B);
                            continue;
                        }
                        if (run_p2p_f16_ps(n, n->p2p[i][0], nullptr, &sdesc, B, P, Ti, nullptr, d.out_p, oh, nullptr, s, "conv_p2p_f16_kernel", mel_fm && i == 1)) continue;
                        AKE_REQUIRE(!mel_fm, AKE_ERR_UNSUPPORTED, "pcnet: this shape does not take the frames-major input (ake_pcnet_accepts_frames_major)");
                        if ((rc = run_p2p_f16(n, n->p2p[i][0], nullptr, &sdesc, B, P, Ti, nullptr, d.out_p, oh, s, "conv_p2p_f16_kernel")))
                            return rc;
                    } else {
                        const unsigned short* xh = reinterpret_cast<const unsigned short*>(((j - 1) & 1) ? b.pb[i] : b.pa[i]);
                        if (last_conv && p2p_fuses_semi(n, i, P, Ti) &&
                            run_p2p_f16_ps(n, n->p2p[i][j], xh, nullptr, B, P, Ti, cat, ctot, nullptr, &n->semi[i], s, "conv_p2p_f16_kernel", false, false,
                                           d.prev_pc)) {
                            fused_semi = fused_fold = true;    // semitone conv AND octave fold inside the launch: the folded maps are in `cat`
                            continue;
                        }
                        if (last_conv && p2p_fuses_semi(n, i, P, Ti) &&
                            run_p2p_f16_ps(n, n->p2p[i][j], xh, nullptr, B, P, Ti, out, d.out_p, nullptr, &n->semi[i], s, "conv_p2p_f16_kernel")) {
                            fused_semi = true;    // `out` holds the semitone maps [clip][8][P / 3][T], not the pitch tensor
                            continue;
                        }
                        if (run_p2p_f16_ps(n, n->p2p[i][j], xh, nullptr, B, P, Ti, last_conv ? out : nullptr, d.out_p, last_conv ? nullptr : oh, nullptr, s,
                                            "conv_p2p_f16_kernel"))
                            continue;
                        if ((rc = run_p2p_f16(n, n->p2p[i][j], xh, nullptr, B, P, Ti, last_conv ? out : nullptr, d.out_p, last_conv ? nullptr : oh, s,
                                               "conv_p2p_f16_kernel")))
                            return rc;
                    }
                    continue;
                }
                if (train && !c.pc2p_mem && !c.stay_sixth) {   // f16 x 3 on the persistent form (f32-equivalent products); else the f32 MFMA kernel
                    const PackedConv& pt = n->p2p_t[i][j];
                    const int bn = bn_of(m + "p2p.layer." + std::to_string(3 * j + 1));
                    if (run_p2p_f16x3(n, pt.bf_off, sdesc, in_aff, n->blob_dev + pt.b_off, B, P, Ti, out, d.out_p, b.stats + 2 * n->bns[bn].ch_off,
                                      2 * n->bn_channels, s, "conv_p2p_f16x3_kernel/p2p")) {
                        finalize_bn(bn, static_cast<double>(B) * P * Ti, out_aff);
                        sdesc = Src{out, d.out_p, nullptr, 0, 0};
                        in_aff = out_aff;
                        continue;
                    }
                }
                // inference in the f32x3 precision mode: the same persistent kernel with the EVAL fragments (BatchNorm folded; f16 hi + lo operands,
                // three products: f32-equivalent to 2^-22) and LeakyReLU in its epilogue, instead of the f32-MFMA kernel at the vector rate
                if (!train && c.precision == AKE_PRECISION_F32X3 && !c.pc2p_mem && !c.stay_sixth && !g_keep_taps && n->p2p[i][j].bf_off >= 0 &&
                    run_p2p_f16x3(n, n->p2p[i][j].bf_off, sdesc, nullptr, n->blob_dev + n->p2p[i][j].b_off, B, P, Ti, out, d.out_p, nullptr, 0, s,
                                  "conv_p2p_f16x3_kernel/p2p", nullptr, true)) {
                    sdesc = Src{out, d.out_p, nullptr, 0, 0};
                    continue;
                }
                if ((rc = conv(n->p2p[i][j], train ? n->p2p_t[i][j] : n->p2p[i][j], m + "p2p.layer." + std::to_string(3 * j + 1), 0, sdesc,
                               in_aff, B, P, Ti, true, out, d.out_p, 0, out_aff, "conv_mfma_kernel/p2p")))
                    return rc;
                sdesc = Src{out, d.out_p, nullptr, 0, 0};
                in_aff = out_aff;
            }
            // models.py:386-392  pool_semi -> fold, written next to pc in the concat buffer
            if (fused_fold) {
            } else if (fused_semi) {
                if ((rc = run_fold_max(out, d.out_p, P / 3, B, Ti, cat, ctot, d.prev_pc, s))) return rc;
            } else if (c.stay_sixth && train) {   // ... through the last conv's pending BatchNorm + LeakyReLU
                const long long total = static_cast<long long>(B) * d.out_p * 12 * Ti;
                ake::ProfScope ps("fold_affine_kernel", s);
                hipLaunchKernelGGL(fold_affine_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, s, out, out_aff, cat, d.out_p, P / 12, Ti,
                                   ctot, d.prev_pc, total);
                identity(b.aff_cat[i] + 3 * d.prev_pc, d.out_p);
            } else if (c.stay_sixth) {   // models.py:391: the stack's output is folded as it is (no semitone conv)
                if ((rc = fold_maps(i, out, d.out_p, P, B, Ti, cat, ctot, d.prev_pc))) return rc;
            } else if ((rc = semi(i, out, out_aff, B, P, Ti, cat, ctot, d.prev_pc, train ? b.aff_cat[i] + 3 * d.prev_pc : nullptr))) return rc;
            if (last) return AKE_OK;                             // pc2pc + pooling + heads run batch-wide
            // inner layers of deeper nets: pc2pc, then both time pools (models.py:393-396)
            const float* psrc = cat;
            const float* psrc_aff = train ? b.aff_cat[i] : nullptr;
            int cin = ctot;
            float* pdst = nullptr;
            float* pdst_aff = nullptr;
            if (c.resblock && train) {
                if ((rc = res_stack_train(n->pc2pc_t[i], m + "pc2pc.layer.", 1, Src{psrc, cin, nullptr, 0, 0}, psrc_aff, B, 12, Ti, b.pcst[i], b.aff_pcst[i],
                                          nullptr, 0, "conv_mfma_kernel/pc2pc")))
                    return rc;
                pdst = b.pcst[i].back(); pdst_aff = b.aff_pcst[i].back();
            } else if (c.resblock) {
                if ((rc = res_stack(n->pc2pc[i], 1, Src{psrc, cin, nullptr, 0, 0}, B, 12, Ti, b.pca[i], b.pcb[i], nullptr, 0, "conv_mfma_kernel/pc2pc"))) return rc;
                pdst = b.pca[i];
            }
            for (int j = 0; j < c.conv_layers && !c.resblock; ++j) {
                pdst = train ? b.pcst[i][j] : ((j & 1) ? b.pcb[i] : b.pca[i]);
                pdst_aff = !train ? nullptr : b.aff_pcst[i][j];
                if ((rc = conv(n->pc2pc[i][j], train ? n->pc2pc_t[i][j] : n->pc2pc[i][j], m + "pc2pc.layer." + std::to_string(3 * j + 1), 1,
                               Src{psrc, cin, nullptr, 0, 0}, psrc_aff, B, 12, Ti, true, pdst, d.out_pc, 0, pdst_aff, "conv_mfma_kernel/pc2pc")))
                    return rc;
                psrc = pdst; psrc_aff = pdst_aff; cin = d.out_pc;
            }
            const LayerDims& dn = n->dims[i + 1];
            const int ctn = dn.prev_pc + dn.out_p;
            const int Tn = Ti / tp;
            float* catn = b.cat[i + 1] + (i + 1 == L - 1 ? static_cast<size_t>(c0) * ctn * 12 * Tn : 0);
            time_pool(pdst, pdst_aff, B, d.out_pc, 12, Ti, catn, ctn, 0);
            if (train) identity(b.aff_cat[i + 1], d.out_pc);
            time_pool(out, out_aff, B, d.out_p, P, Ti, b.ppool[i], d.out_p, 0);
            pc_cur = catn; p_cur = b.ppool[i]; cp = d.out_p;
        }
        return AKE_OK;
    }

    // Phase C, whole batch: last layer's pc2pc, time pool, heads, masked mean.
    int tail(int B, const int64_t* seq, float* key_out, float* tonic_out, float* genre_out) {
        const auto& c = n->cfg;
        const int L = c.num_layers, tp = c.time_pool_size;
        const int i = L - 1;
        const int Ti = b.Tl[i];
        const LayerDims& d = n->dims[i];
        int rc;
        const float* psrc = L == 1 ? b.fold0 : b.cat[i];
        const float* psrc_aff = (train && L > 1) ? b.aff_cat[i] : nullptr;
        int cin = L == 1 ? 1 : d.prev_pc + d.out_p;
        const int cout_default = L == 1 ? c.n_filters : d.out_pc;
        const int cout = cout_default;
        const std::string m = "model." + std::to_string(i) + ".pc2pc.layer.";
        float* pdst = nullptr;
        float* pdst_aff = nullptr;
        // inference, 16-channel stacks: bf16 MFMA with split operands; the stack's input is converted to channels-last planes once,
        // the intermediate activations stay in that format (same 64 B per position as 16 f32 channels: the ping-pong buffers are
        // reused), the last convolution writes NCHW f32 for the pooling / heads
        const bool pc_bf = !train && pc2pc_uses_bf16(n, i, Ti);
        // heads on the bf16 kernels read a channels-last copy of the pooled features (decided here: the fused stack writes it itself)
        const bool head_bf = !train && !g_pc_f32_only && L > 1 && n->final_ch == 16 && c.head_layers >= 2 && n->head_key[0].bf_off >= 0 &&
                             n->head_tonic[0].bf_off >= 0 && b.Tf <= kPcBf16MaxFrames;
        const bool pc_fused = !train && L > 1 && !c.denseblock && pc2pc_fuses(n, i, Ti);
        if (pc_fused) {
            if ((rc = run_pc2pc_fused(n, i, psrc, cin, B, Ti, b.pcf, head_bf ? b.feat_cl : nullptr, s))) return rc;
        } else if (pc_bf) run_nchw_to_cl16(psrc, cin, B, Ti, reinterpret_cast<unsigned short*>(b.pcb[i]), s);
        if (c.resblock && train) {
            if ((rc = res_stack_train(n->pc2pc_t[i], m, 1, Src{psrc, cin, nullptr, 0, 0}, psrc_aff, B, 12, Ti, b.pcst[i], b.aff_pcst[i], nullptr, 0,
                                      L == 1 ? "conv_mfma_kernel/pc2pc0" : "conv_mfma_kernel/pc2pc")))
                return rc;
            pdst = b.pcst[i].back(); pdst_aff = b.aff_pcst[i].back();
        } else if (c.resblock) {
            if ((rc = res_stack(n->pc2pc[i], 1, Src{psrc, cin, nullptr, 0, 0}, B, 12, Ti, b.pca[i], b.pcb[i], nullptr, 0,
                                L == 1 ? "conv_mfma_kernel/pc2pc0" : "conv_mfma_kernel/pc2pc")))
                return rc;
            pdst = b.pca[i];
        }
        if (c.denseblock) {   // the last layer's pitch-class block, in place on its concat buffer: the features are the buffer itself
            float* feat_buf = L == 1 ? b.fold0 : b.cat[i];
            if (train) {
                if ((rc = dense_stack_train(n->dense_pc[i], 1, feat_buf, n->final_ch, cin, B, 12, Ti, b.dn_bott_pc[i], b.dn_aff1_pc[i], b.dn_aff2_pc[i],
                                            L == 1 ? "conv_mfma_kernel/pc2pc0" : "conv_mfma_kernel/pc2pc")))
                    return rc;
            } else if ((rc = dense_stack(n->dense_pc[i], 1, feat_buf, n->final_ch, cin, B, 12, Ti, b.pca[i],
                                         L == 1 ? "conv_mfma_kernel/pc2pc0" : "conv_mfma_kernel/pc2pc")))
                return rc;
            pdst = feat_buf; pdst_aff = nullptr;
        }
        for (int j = 0; j < c.conv_layers && !pc_fused && !c.resblock && !c.denseblock; ++j) {
            pdst = train ? b.pcst[i][j] : ((j & 1) ? b.pcb[i] : b.pca[i]);
            pdst_aff = !train ? nullptr : b.aff_pcst[i][j];
            if (pc_bf) {
                const bool last_conv = j == c.conv_layers - 1;
                const unsigned short* in = reinterpret_cast<const unsigned short*>((j & 1) ? b.pca[i] : b.pcb[i]);
                if ((rc = run_pc_bf16(n, n->pc2pc[i][j], in, B, Ti, true, true, last_conv ? pdst : nullptr,
                                      last_conv ? nullptr : reinterpret_cast<unsigned short*>(pdst), s, "conv_pc_bf16_kernel/pc2pc")))
                    return rc;
                continue;
            }
            if (train && L > 1 && pc_f16x3_ok(n->pc2pc_t[i][j], Ti, true)) {   // f16 x 3 MFMA (f32-equivalent products) instead of the f32 MFMA kernel
                const PackedConv& pt = n->pc2pc_t[i][j];
                unsigned short* planes = reinterpret_cast<unsigned short*>(b.pcb[i]);       // (the inference ping-pong buffer: idle in training)
                run_nchw_to_cl16_f16x2(psrc, cin, B, Ti, psrc_aff, planes, s);
                const int bn = bn_of(m + std::to_string(3 * j + 1));
                if ((rc = run_pc_bf16(n, pt, planes, B, Ti, true, false, pdst, nullptr, s, "conv_pc_f16x3_kernel/pc2pc", nullptr, nullptr, true,
                                      b.stats + 2 * n->bns[bn].ch_off, 2 * n->bn_channels)))
                    return rc;
                finalize_bn(bn, static_cast<double>(B) * 12 * Ti, pdst_aff);
                psrc = pdst; psrc_aff = pdst_aff; cin = cout;
                continue;
            }
            if ((rc = conv(n->pc2pc[i][j], train ? n->pc2pc_t[i][j] : n->pc2pc[i][j], m + std::to_string(3 * j + 1), 1,
                           Src{psrc, cin, nullptr, 0, 0}, psrc_aff, B, 12, Ti, true, pdst, cout, 0, pdst_aff,
                           L == 1 ? "conv_mfma_kernel/pc2pc0" : "conv_mfma_kernel/pc2pc")))
                return rc;
            psrc = pdst; psrc_aff = pdst_aff; cin = cout;
        }
        const float* feat = pdst;                 // features feeding the heads
        const float* feat_aff = pdst_aff;
        if (pc_fused) { feat = b.pcf; feat_aff = nullptr; }
        else if (L > 1) {   // models.py:396  (the pitch stream of the last layer feeds nothing: its pool is skipped)
            const int cout = c.denseblock ? n->final_ch : cout_default;
            time_pool(pdst, pdst_aff, B, cout, 12, Ti, b.pcf, cout, 0);
            feat = b.pcf; feat_aff = nullptr;
        }
        // ---- heads (models.py:750-753) ----
        const int Tf = b.Tf;
        struct HeadRun { const std::vector<PackedConv>* ce; const std::vector<PackedConv>* ct; float* hid; float* map; int kind; const char* nm; };
        HeadRun heads[3] = {{&n->head_key, &n->head_key_t, b.hid_k, b.map_k, 1, "key_classifier"},
                            {&n->head_tonic, &n->head_tonic_t, b.hid_t, b.map_t, 1, "tonic_classifier"},
                            {&n->head_genre, &n->head_genre_t, b.hid_g, b.map_g, 2, "genre_classifier"}};
        int Tm = Tf;
        bool head_planes_ready = false;           // training: the f16 hi / lo planes of the pooled features (b.feat_cl) exist
        int pooled_heads = 0;                     // heads whose outputs conv_head1_bf16_kernel already wrote
        // key / tonic heads: the first convolution (16 -> 32 channels, most of a head's work) on the bf16 kernel; both read the same
        // channels-last copy of the features
        unsigned short* feat_cl = b.feat_cl;
        if (head_bf && !pc_fused) run_nchw_to_cl16(feat, n->final_ch, B, Tf, feat_cl, s);
        // two-conv heads: conv0 leaves its 32 channels as channels-last planes and ONE launch of conv_head1_bf16_kernel finishes
        // both the key and the tonic map
        const int T1 = Tf - (c.kernel_size - 1), T2 = T1 - (c.kernel_size - 1);
        const bool head1_bf = head_bf && c.head_layers == 2 && n->head_key[1].bf_off >= 0 && n->head_tonic[1].bf_off >= 0 && T2 > 0 &&
                              (12 * ((T2 + 15) / 16) + 15) / 16 <= kHead1MT;
        const bool genre_bf = head1_bf && c.genre && n->head_genre.size() == 2 && n->head_genre[0].bf_off >= 0 && n->head_genre[1].bf_off >= 0 &&
                              n->head_genre[0].kh == 1 && n->head_genre[1].kh == 2;
        if (head1_bf) {
            Head1BfArgs ha;
            std::memset(&ha, 0, sizeof(ha));
            float* maps[3] = {b.map_k, b.map_t, b.map_g};
            float* hids[3] = {b.hid_k, b.hid_t, b.hid_g};
            const PackedConv* c0[3] = {&n->head_key[0], &n->head_tonic[0], genre_bf ? &n->head_genre[0] : nullptr};
            const PackedConv* c1[3] = {&n->head_key[1], &n->head_tonic[1], genre_bf ? &n->head_genre[1] : nullptr};
            const int nh = genre_bf ? 3 : 2;
            for (int h = 0; h < nh; ++h) {
                unsigned short* planes = reinterpret_cast<unsigned short*>(hids[h]);
                // the key and the tonic head read the same features with the same geometry: one launch, blockIdx.y picks the head
                if (h == 0 && (rc = run_pc_bf16(n, *c0[0], feat_cl, B, Tf, false, true, nullptr, planes, s, "conv_pc_bf16_kernel/head", c0[1],
                                                reinterpret_cast<unsigned short*>(hids[1]))))
                    return rc;
                if (h == 2 && (rc = run_pc_bf16(n, *c0[h], feat_cl, B, Tf, false, true, nullptr, planes, s, "conv_pc_bf16_kernel/genre_head")))
                    return rc;
                ha.xh[h] = planes; ha.xl[h] = planes + static_cast<long long>(B) * 12 * T1 * 32;
                ha.bfrag[h] = n->bf_frags_dev + c1[h]->bf_off; ha.bias[h] = n->blob_dev + c1[h]->b_off; ha.dst[h] = maps[h];
                ha.KH[h] = c1[h]->kh; ha.circular[h] = c1[h]->kh == 12 ? 1 : 0; ha.H_out[h] = ha.circular[h] ? 12 : 12 - c1[h]->kh + 1;
            }
            ha.T_in = T1; ha.T_out = T2; ha.JB = (T2 + 15) / 16; ha.Tp = 16 * (ha.JB - 1) + 22;
            // the patch, and after the multiply loop the partial tiles of the 8 waves in the same bytes (two workgroups per CU fit)
            size_t lds = std::max(static_cast<size_t>(2) * 12 * ha.Tp * 4 * sizeof(uint4), static_cast<size_t>(8) * kHead1MT * 4 * 64 * sizeof(float));
            if (c.local == 0) {   // the masked mean + sigmoid of each finished map in the same launch (its LDS copy sits behind the patch / partial tiles)
                ha.fin_off = static_cast<int>(lds / sizeof(float));
                lds += static_cast<size_t>(12) * T2 * sizeof(float);
                float* outs[3] = {key_out, tonic_out, genre_out};
                for (int h = 0; h < nh; ++h) ha.pout[h] = outs[h];
                ha.seq = reinterpret_cast<const long long*>(seq);
                ha.n_pool_layers = L - 1; ha.tp = tp; ha.shrink = (c.kernel_size - 1) * c.head_layers; ha.max_pool = c.max_pool; ha.clip0 = 0;
                pooled_heads = nh;
            }
            static ake::DeviceOnce h1_attr;
            if (h1_attr.need()) {
                AKE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_head1_bf16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
                h1_attr.mark();
            }
            AKE_REQUIRE(lds <= 150 * 1024, AKE_ERR_UNSUPPORTED, "heads: %d frames do not fit conv_head1_bf16_kernel", T1);
            ake::ProfScope ps("conv_head1_bf16_kernel", s);
            hipLaunchKernelGGL(conv_head1_bf16_kernel, dim3(B, nh), dim3(512), lds, s, ha);
            Tm = T2;
        }
        for (int h = head1_bf ? (genre_bf ? 3 : 2) : 0; h < (c.genre ? 3 : 2); ++h) {
            const float* src = feat;
            const float* src_aff = feat_aff;
            int hc = n->final_ch, Tcur = Tf;
            const size_t hid_half = static_cast<size_t>(B) * 2 * n->final_ch * 12 * Tf;
            for (int j = 0; j < c.head_layers; ++j) {
                const PackedConv& pe = (*heads[h].ce)[j];
                const bool lastj = j == c.head_layers - 1;
                float* dst = lastj ? heads[h].map : (train ? b.hst[h][j] : heads[h].hid + (j & 1) * hid_half);
                float* aff = (!train || lastj) ? nullptr : b.aff_hst[h][j];
                if (head_bf && h < 2 && j == 0) {
                    if ((rc = run_pc_bf16(n, pe, feat_cl, B, Tcur, false, true, dst, nullptr, s, "conv_pc_bf16_kernel/head"))) return rc;
                    src = dst; src_aff = aff; hc = pe.cout; Tcur -= c.kernel_size - 1;
                    continue;
                }
                if (train && j == 0 && !lastj && pc_f16x3_ok((*heads[h].ct)[0], Tcur, false)) {   // the heads' first convs: f16 x 3 MFMA
                    const PackedConv& pt = (*heads[h].ct)[0];
                    if (!head_planes_ready) {               // one conversion of the pooled features serves the three heads
                        run_nchw_to_cl16_f16x2(src, hc, B, Tcur, src_aff, b.feat_cl, s);
                        head_planes_ready = true;
                    }
                    const int bn = bn_of(std::string(heads[h].nm) + ".1");
                    if ((rc = run_pc_bf16(n, pt, b.feat_cl, B, Tcur, false, false, dst, nullptr, s,
                                          h == 2 ? "conv_pc_f16x3_kernel/genre_head" : "conv_pc_f16x3_kernel/head", nullptr, nullptr, true,
                                          b.stats + 2 * n->bns[bn].ch_off, 2 * n->bn_channels)))
                        return rc;
                    const int H_out = heads[h].kind == 2 ? 12 - pt.kh + 1 : 12;
                    finalize_bn(bn, static_cast<double>(B) * H_out * (Tcur - pt.kw + 1), aff);
                    src = dst; src_aff = aff; hc = pe.cout; Tcur -= c.kernel_size - 1;
                    continue;
                }
                if (train && lastj && j > 0 && pe.cout == 1 && pe.kw == 7 && !n->raw_w_off.empty()) {   // cin -> 1: one workgroup per clip, f32 VALU
                    static const bool off = ake::diag_env("AKE_HEAD_LAST_MFMA") != nullptr;
                    const std::string wn = std::string(heads[h].nm) + "." + std::to_string(3 * j) + (heads[h].kind == 2 ? "" : ".conv2d");
                    const auto wi = n->spec_index.find(wn + ".weight"), bi = n->spec_index.find(wn + ".bias");
                    const size_t lds = (static_cast<size_t>(hc) * 12 * Tcur + static_cast<size_t>(hc) * pe.kh * 7) * sizeof(float);
                    if (!off && wi != n->spec_index.end() && bi != n->spec_index.end() && lds <= 150 * 1024 && Tcur - 6 >= 1) {
                        static ake::DeviceOnce hl_attr;
                        if (hl_attr.need()) {
                            AKE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_head_last_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
                            hl_attr.mark();
                        }
                        ake::ProfScope ps(h == 2 ? "conv_head_last_kernel/genre" : "conv_head_last_kernel", s);
                        hipLaunchKernelGGL(conv_head_last_kernel, dim3(B), dim3(384), lds, s, src, src_aff, n->blob_dev + n->raw_w_off[wi->second],
                                           n->blob_dev + n->raw_w_off[bi->second], dst, hc, pe.kh, heads[h].kind == 2 ? 0 : 1, Tcur, Tcur - 6);
                        src = dst; src_aff = aff; hc = pe.cout; Tcur -= c.kernel_size - 1;
                        continue;
                    }
                }
                if ((rc = conv(pe, train ? (*heads[h].ct)[j] : pe, lastj ? "" : std::string(heads[h].nm) + "." + std::to_string(3 * j + 1),
                               heads[h].kind, Src{src, hc, nullptr, 0, 0}, src_aff, B, 12, Tcur, false, dst, pe.cout, 0, aff,
                               h == 2 ? "conv_mfma_kernel/genre_head" : "conv_mfma_kernel/head")))
                    return rc;
                src = dst; src_aff = aff; hc = pe.cout; Tcur -= c.kernel_size - 1;
            }
            Tm = Tcur;
        }
        if (c.local > 0) {   // ---- --local: sliding-window max over the maps, per-frame outputs (models.py:720-722, 805-810) ----
            AKE_REQUIRE(Tm >= c.local, AKE_ERR_INVALID, "pcnet --local: %d map frames are fewer than the pooling window %d", Tm, c.local);
            LocalPoolArgs la;
            std::memset(&la, 0, sizeof(la));
            la.maps[0] = b.map_k; la.maps[1] = b.map_t; la.maps[2] = c.genre ? b.map_g : nullptr;
            la.outs[0] = key_out; la.outs[1] = tonic_out; la.outs[2] = genre_out;
            la.Tm = Tm; la.Tq = Tm - c.local + 1; la.W = c.local; la.batch = B;
            ake::ProfScope ps("local_pool_kernel", s);
            hipLaunchKernelGGL(local_pool_kernel, dim3(static_cast<unsigned>((static_cast<long long>(B) * 12 * Tm + 255) / 256), 3), dim3(256), 0, s, la);
            AKE_HIP_CHECK(hipGetLastError());
            return AKE_OK;
        }
        // ---- masked temporal mean, sigmoid (models.py:754-804) ----
        PoolHeadArgs pa;
        std::memset(&pa, 0, sizeof(pa));
        pa.maps[0] = b.map_k; pa.maps[1] = b.map_t; pa.maps[2] = c.genre ? b.map_g : nullptr;
        for (int h = 0; h < pooled_heads; ++h) pa.maps[h] = nullptr;
        if (!pa.maps[0] && !pa.maps[1] && !pa.maps[2]) return AKE_OK;
        pa.outs[0] = key_out; pa.outs[1] = tonic_out; pa.outs[2] = genre_out;
        pa.rows[0] = 12; pa.rows[1] = 12; pa.rows[2] = 11;
        pa.Tm = Tm; pa.seq = reinterpret_cast<const long long*>(seq);
        pa.n_pool_layers = L - 1; pa.tp = tp; pa.shrink = (c.kernel_size - 1) * c.head_layers;
        pa.max_pool = c.max_pool; pa.batch = B; pa.clip0 = 0;
        {
            ake::ProfScope ps("head_pool_kernel", s);
            hipLaunchKernelGGL(head_pool_kernel, dim3((B * 12 + 63) / 64, 3), dim3(64), 0, s, pa);
        }
        AKE_HIP_CHECK(hipGetLastError());
        return AKE_OK;
    }
};

int forward_impl(const ake_pcnet* n, bool train, const float* mel, int batch, int frames, const int64_t* seq_length, float* key_out,
                 float* tonic_out, float* genre_out, float* bn_stats_out, void* workspace, size_t ws_bytes, ake_stream_t stream,
                 bool mel_frames_major = false, bool dry_run = false) {
    AKE_REQUIRE(n && mel && key_out && tonic_out, AKE_ERR_INVALID, "pcnet forward: null argument");
    AKE_REQUIRE(n->finalized, AKE_ERR_STATE, "pcnet: ake_pcnet_finalize has not been called");
    AKE_REQUIRE(train || !n->eval_frags_stale, AKE_ERR_STATE,
                "pcnet: the last weight load was ake_pcnet_load_for_training_f32 (training fragments only): call ake_pcnet_load_from_device_f32 before inference");
    AKE_REQUIRE(batch > 0 && frames > 0, AKE_ERR_INVALID, "pcnet: bad batch/frames");
    AKE_REQUIRE(!n->cfg.genre || genre_out, AKE_ERR_INVALID, "pcnet: genre head enabled but genre_out is null");
    const int chunk = train ? batch : std::min(batch, n->chunk_clips);     // batch statistics need the whole batch at once
    Buffers b;
    int rc = plan_buffers(n, batch, chunk, frames, workspace, &b, train);
    if (rc) return rc;
    AKE_REQUIRE(workspace && ws_bytes >= b.bytes, AKE_ERR_WORKSPACE, "pcnet: workspace %zu < %zu bytes", ws_bytes, b.bytes);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (train) AKE_HIP_CHECK(hipMemsetAsync(b.stats, 0, sizeof(double) * 2 * n->bn_channels * kStatSlots, s));
    Fwd f{n, b, s, train};
    f.mel_fm = mel_frames_major;
    f.chunk = chunk;
    if (dry_run) {   // ake_pcnet_accepts_frames_major: would the two readers of mel take that layout for this shape?  (no launch)
        const auto& c = n->cfg;
        if (train || c.num_layers != 2 || c.resblock || c.denseblock || c.pc2p_mem || c.p2pc_conv || c.stay_sixth || c.local ||
            batch > chunk || !p2p_uses_f16(n, 1, b.Tl[1]))
            return AKE_ERR_UNSUPPORTED;
        if (!f.layer0_fused(mel, batch, true)) return AKE_ERR_UNSUPPORTED;
        const LayerDims& d = n->dims[1];
        Src sdesc{mel, 1, b.psix[1], d.prev_pc, 36};
        return run_p2p_f16_ps(n, n->p2p[1][0], nullptr, &sdesc, batch, c.pitches, b.Tl[1], nullptr, d.out_p, reinterpret_cast<unsigned short*>(b.pa[1]), nullptr, s,
                              "conv_p2p_f16_kernel", true, true) ? AKE_OK : AKE_ERR_UNSUPPORTED;
    }
    if ((rc = f.entry(mel, batch))) return rc;
    for (int c0 = 0; c0 < batch && n->cfg.num_layers > 1; c0 += chunk) {
        const int B = std::min(chunk, batch - c0);
        if ((rc = f.pitch_chunk(mel + static_cast<size_t>(c0) * n->cfg.pitches * frames, c0, B))) return rc;
    }
    if ((rc = f.tail(batch, seq_length, key_out, tonic_out, genre_out))) return rc;
    if (train && bn_stats_out)
        AKE_HIP_CHECK(hipMemcpyAsync(bn_stats_out, b.bstats, sizeof(float) * 3 * n->bn_channels, hipMemcpyDeviceToDevice, s));
    return AKE_OK;
}

}  // namespace

#include "pcnet_backward.h"

// Flat gradient buffer: float entries of the state_dict in ake_pcnet_tensor_info order (running statistics get zeros).
size_t ake_pcnet_grad_floats(const ake_pcnet* n) { return n ? n->grad_floats : 0; }

int64_t ake_pcnet_grad_offset(const ake_pcnet* n, const char* name) {
    if (!n || !name) return -1;
    auto it = n->spec_index.find(name);
    if (it == n->spec_index.end()) return -1;
    return static_cast<int64_t>(n->grad_off[it->second]);
}

int ake_pcnet_backward_f32(const ake_pcnet* n, const float* mel, int batch, int frames, const int64_t* seq_length, const float* key_out,
                           const float* d_key, const float* d_tonic, const float* d_genre, float* grads_out, int accumulate,
                           void* workspace, size_t ws_bytes, ake_stream_t stream) {
    AKE_REQUIRE(n && mel && key_out && d_key && d_tonic && grads_out, AKE_ERR_INVALID, "pcnet backward: null argument");
    AKE_REQUIRE(n->finalized, AKE_ERR_STATE, "pcnet: ake_pcnet_finalize has not been called");
    AKE_REQUIRE(!n->cfg.genre || d_genre, AKE_ERR_INVALID, "pcnet backward: genre head enabled but d_genre is null");
    AKE_REQUIRE(!(n->cfg.p2pc_conv && n->cfg.stay_sixth), AKE_ERR_UNSUPPORTED, "pcnet: training a --p2pc_conv --stay_sixth net is not built");
    Buffers b;
    int rc = plan_buffers(n, batch, batch, frames, workspace, &b, true);
    if (rc) return rc;
    AKE_REQUIRE(workspace && ws_bytes >= b.bytes, AKE_ERR_WORKSPACE, "pcnet backward: workspace %zu < %zu bytes", ws_bytes, b.bytes);
    hipStream_t s = static_cast<hipStream_t>(stream);
    AKE_HIP_CHECK(hipMemsetAsync(b.gslots, 0, sizeof(gfx_t) * n->grad_floats * kGradSlots, s));
    AKE_HIP_CHECK(hipMemsetAsync(b.stats2, 0, sizeof(double) * (static_cast<size_t>(3) * n->bn_channels * kBwdStatSlots + n->bns.size() * (kAmaxSlots / 2)), s));
    Bwd bw{n, b, s, b.gslots, batch};
    rc = bw.run(mel, seq_length, d_key, d_tonic, d_genre, key_out);
    if (rc) return rc;
    {
        ake::ProfScope ps("grad_reduce_kernel", s);
        hipLaunchKernelGGL(grad_reduce_kernel, dim3(static_cast<unsigned>((n->grad_floats + 255) / 256)), dim3(256), 0, s, b.gslots, grads_out,
                           static_cast<long long>(n->grad_floats), accumulate ? 1 : 0);
    }
    AKE_HIP_CHECK(hipGetLastError());
    return AKE_OK;
}

int ake_pcnet_local_frames(const ake_pcnet* n, int frames, int* pooled_frames, int* map_frames) {
    AKE_REQUIRE(n && n->cfg.local > 0, AKE_ERR_STATE, "pcnet: the net was not created with local > 0");
    int t = frames;
    for (int i = 1; i < n->cfg.num_layers; ++i) t /= n->cfg.time_pool_size;
    t -= (n->cfg.kernel_size - 1) * n->cfg.head_layers;
    if (map_frames) *map_frames = t;
    if (pooled_frames) *pooled_frames = t - n->cfg.local + 1;
    return AKE_OK;
}

int ake_pcnet_forward_local_f32(const ake_pcnet* n, const float* mel, int batch, int frames, float* key_out, float* tonic_out, float* genre_out,
                                void* workspace, size_t ws_bytes, ake_stream_t stream) {
    AKE_REQUIRE(n && n->cfg.local > 0, AKE_ERR_STATE, "pcnet: the net was not created with local > 0");
    return forward_impl(n, false, mel, batch, frames, nullptr, key_out, tonic_out, genre_out, nullptr, workspace, ws_bytes, stream);
}

int ake_pcnet_forward_f32(const ake_pcnet* n, const float* mel, int batch, int frames, const int64_t* seq_length,
                          float* key_out, float* tonic_out, float* genre_out, void* workspace, size_t ws_bytes,
                          ake_stream_t stream) {
    AKE_REQUIRE(!n || n->cfg.local == 0, AKE_ERR_STATE, "pcnet: a --local net returns per-frame outputs: call ake_pcnet_forward_local_f32");
    return forward_impl(n, false, mel, batch, frames, seq_length, key_out, tonic_out, genre_out, nullptr, workspace, ws_bytes, stream);
}

// mel as the CQT filter bank writes it, [batch][frames][pitches] (ake_cqt_logmag_frames_major_f32): the two kernels that read the CQT
// (layer 0 and the first pitch convolution) transpose while they stage it, so the [batch][pitches][frames] tensor of the
// reference's interface -- a 48 MB round trip per 256 clips -- is never written.  Taken by the default architecture on the fused
// inference path only: ask ake_pcnet_accepts_frames_major first (other configurations return AKE_ERR_UNSUPPORTED).
int ake_pcnet_accepts_frames_major(const ake_pcnet* n, int batch, int frames) {
    if (!n || !n->finalized || batch <= 0 || frames <= 0) return 0;
    float dummy = 0.f;           // never dereferenced: the dry run stops before any launch
    Buffers b;
    if (plan_buffers(n, batch, std::min(batch, n->chunk_clips), frames, nullptr, &b, false) != AKE_OK) return 0;
    const int rc = forward_impl(n, false, reinterpret_cast<const float*>(16), batch, frames, nullptr, &dummy, &dummy, &dummy, nullptr,
                                reinterpret_cast<void*>(16), b.bytes, nullptr, true, true);
    return rc == AKE_OK ? 1 : 0;
}

int ake_pcnet_forward_frames_major_f32(const ake_pcnet* n, const float* mel_fm, int batch, int frames, const int64_t* seq_length,
                                       float* key_out, float* tonic_out, float* genre_out, void* workspace, size_t ws_bytes,
                                       ake_stream_t stream) {
    AKE_REQUIRE(!n || n->cfg.local == 0, AKE_ERR_UNSUPPORTED, "pcnet: a --local net does not take the frames-major input");
    return forward_impl(n, false, mel_fm, batch, frames, seq_length, key_out, tonic_out, genre_out, nullptr, workspace, ws_bytes, stream, true);
}

int ake_pcnet_forward_train_f32(const ake_pcnet* n, const float* mel, int batch, int frames, const int64_t* seq_length,
                                float* key_out, float* tonic_out, float* genre_out, float* bn_stats_out, void* workspace,
                                size_t ws_bytes, ake_stream_t stream) {
    AKE_REQUIRE(!n || !(n->cfg.p2pc_conv && n->cfg.stay_sixth), AKE_ERR_UNSUPPORTED, "pcnet: training a --p2pc_conv --stay_sixth net is not built (inference only)");
    return forward_impl(n, true, mel, batch, frames, seq_length, key_out, tonic_out, genre_out, bn_stats_out, workspace, ws_bytes, stream);
}

// ---- debug taps ---------------------------------------------------------------------------
static int tap_lookup(const ake_pcnet* n, const char* name, int batch, int frames, const void* ws, float** p, int64_t shape[4],
                      int* channels_last = nullptr) {
    if (channels_last) *channels_last = 0;
    AKE_REQUIRE(n && name, AKE_ERR_INVALID, "tap: null argument");
    AKE_REQUIRE(batch <= n->chunk_clips, AKE_ERR_INVALID, "tap: batch %d exceeds the chunk size %d", batch, n->chunk_clips);
    Buffers b;
    const bool tr = std::strncmp(name, "train:", 6) == 0;      // buffers of the training-mode workspace (debugging the backward pass)
    int rc = plan_buffers(n, batch, batch, frames, const_cast<void*>(ws), &b, tr);
    if (rc) return rc;
    const auto& c = n->cfg;
    if (tr) {
        const std::string t = name + 6;
        const int L1 = c.num_layers - 1;
        const LayerDims& dd = n->dims[L1];
        *p = nullptr;
        shape[0] = batch; shape[1] = 0;
        if (t == "g_p") { *p = b.g_p[L1]; shape[1] = dd.out_p; shape[2] = c.pitches; shape[3] = b.Tl[L1]; }
        if (t == "g_semi") { *p = b.g_semi[L1]; shape[1] = dd.out_p; shape[2] = c.pitches / 3; shape[3] = b.Tl[L1]; }
        if (t == "g_cat") { *p = b.g_cat[L1]; shape[1] = dd.prev_pc + dd.out_p; shape[2] = 12; shape[3] = b.Tl[L1]; }
        if (t == "g_pin") { *p = b.g_pin[L1]; shape[1] = dd.prev_pc + dd.prev_p; shape[2] = c.pitches; shape[3] = b.Tl[L1]; }
        if (t == "z_p_last") { *p = b.pst[L1].back(); shape[1] = dd.out_p; shape[2] = c.pitches; shape[3] = b.Tl[L1]; }
        AKE_REQUIRE(shape[1] > 0, AKE_ERR_INVALID, "tap: unknown training buffer '%s'", name);
        if (ake::diag_env("AKE_DEBUG")) fprintf(stderr, "[ake] tap %s -> %p\n", name, (void*)*p);
        return AKE_OK;
    }
    const int L = c.num_layers, P = c.pitches;
    const std::string nm = name;
    auto set = [&](float* ptr, int64_t C, int64_t H, int64_t Tn) { *p = ptr; shape[0] = batch; shape[1] = C; shape[2] = H; shape[3] = Tn; return AKE_OK; };
    if (nm == "model.0.pool") return set(b.fold0, 1, 12, frames);
    if (c.resblock && (std::strstr(name, "pc2pc.layer.") || std::strstr(name, "p2p.layer."))) {
        ake::set_error("tap: '%s': the stacks of a --resblock net are not tapped", name);
        return AKE_ERR_INVALID;
    }
    for (int i = 0; i < L; ++i) {
        const std::string m = "model." + std::to_string(i) + ".";
        const LayerDims& d = n->dims[i];
        const int Ti = b.Tl[i];
        const int last_j = c.conv_layers - 1;
        for (int j = 0; j < c.conv_layers; ++j) {
            // ping-pong buffers: only the last two outputs of a stack are still in memory after the forward
            if (nm == m + "pc2pc.layer." + std::to_string(3 * j + 2)) {
                const bool to_cat = i == 0 && L > 1;             // layer 0's last conv writes into cat[1]
                if (to_cat && j == last_j) break;                // strided inside the concat buffer: use "model.1.cat"
                if (j < last_j - (to_cat ? 2 : 1)) break;
                if (i == L - 1 && pc2pc_fuses(n, i, Ti)) {
                    ake::set_error("tap: '%s' stays in LDS (the stack runs as one launch); ake_debug_keep_taps(1) before the forward keeps it", name);
                    return AKE_ERR_INVALID;
                }
                if (j < last_j && i == L - 1 && pc2pc_uses_bf16(n, i, Ti) && channels_last) *channels_last = 1;
                if (i == 0 && L > 1 && !g_keep_taps && !g_pc_f32_only && c.precision == AKE_PRECISION_MIXED && !c.resblock && !c.denseblock && !c.p2pc_conv && !c.stay_sixth && c.n_filters >= 2 &&
                    c.n_filters <= 4) {
                    ake::set_error("tap: '%s' stays in LDS (layer 0 runs as one launch); ake_debug_keep_taps(1) before the forward writes it", name);
                    return AKE_ERR_INVALID;
                }
                return set((j & 1) ? b.pcb[i] : b.pca[i], i == 0 ? c.n_filters : d.out_pc, 12, Ti);
            }
            if (i >= 1 && nm == m + "p2p.layer." + std::to_string(3 * j + 2)) {
                if (j < last_j - 1) break;
                if (j == last_j && p2p_fuses_semi(n, i, P, Ti)) {
                    ake::set_error("tap: '%s' is fused with the semitone conv that follows it (never written); ake_debug_keep_taps(1) before the forward keeps it", name);
                    return AKE_ERR_INVALID;
                }
                // inference keeps the stack's intermediate activations as channels-last split-bf16 planes (conv_p2p_f16_kernel)
                if (j < last_j && p2p_uses_f16(n, i, Ti) && channels_last) *channels_last = 2;      // one f16 plane
                return set((j & 1) ? b.pb[i] : b.pa[i], d.out_p, P, Ti);
            }
        }
        if (i >= 1 && nm == m + "up_sixth_a") {
            if (i == 1 && !g_keep_taps && !g_pc_f32_only && c.precision == AKE_PRECISION_MIXED && !c.denseblock && !c.p2pc_conv && !c.stay_sixth && !c.pc2p_mem) {
                ake::set_error("tap: '%s' may be held as f16 words for the pitch conv that reads it; ake_debug_keep_taps(1) before the forward writes it as f32", name);
                return AKE_ERR_INVALID;
            }
            return set(b.psix[i], d.prev_pc, 36, Ti);
        }
        if (i >= 1 && nm == m + "cat") return set(b.cat[i], d.prev_pc + d.out_p, 12, Ti);
        if (i == L - 1 && i >= 1 && nm == m + "time_pool_pc") return set(b.pcf, d.out_pc, 12, b.Tf);
    }
    const int Tm = b.Tf - (c.kernel_size - 1) * c.head_layers;
    if (nm == "key_map") return set(b.map_k, 1, 12, Tm);
    if (nm == "tonic_map") return set(b.map_t, 1, 12, Tm);
    if (nm == "genre_map" && c.genre) return set(b.map_g, 1, 11, Tm);
    ake::set_error("tap: '%s' is not a materialised activation", name);
    return AKE_ERR_INVALID;
}

int ake_debug_keep_taps(int on) {
    const int old = g_keep_taps;
    g_keep_taps = on != 0;
    return old;
}

int ake_pcnet_tap_info(const ake_pcnet* n, const char* name, int batch, int frames, int64_t shape[4]) {
    float* p = nullptr;
    return tap_lookup(n, name, batch, frames, nullptr, &p, shape);
}

int ake_pcnet_tap_copy(const ake_pcnet* n, const char* name, int batch, int frames, const void* workspace, float* out_dev,
                       ake_stream_t stream) {
    AKE_REQUIRE(workspace && out_dev, AKE_ERR_INVALID, "tap_copy: null argument");
    float* p = nullptr;
    int64_t shape[4];
    int cl = 0;
    int rc = tap_lookup(n, name, batch, frames, workspace, &p, shape, &cl);
    if (rc) return rc;
    if (cl) {
        const long long total = shape[0] * shape[1] * shape[2] * shape[3];
        const unsigned short* h = reinterpret_cast<const unsigned short*>(p);
        hipLaunchKernelGGL(cl_to_nchw_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), h,
                           cl == 2 ? nullptr : h + total, out_dev, static_cast<int>(shape[1]), static_cast<int>(shape[2]), static_cast<int>(shape[3]), total);
        AKE_HIP_CHECK(hipGetLastError());
        return AKE_OK;
    }
    AKE_HIP_CHECK(hipMemcpyAsync(out_dev, p, sizeof(float) * shape[0] * shape[1] * shape[2] * shape[3], hipMemcpyDeviceToDevice,
                                 static_cast<hipStream_t>(stream)));
    return AKE_OK;
}

}  // extern "C"

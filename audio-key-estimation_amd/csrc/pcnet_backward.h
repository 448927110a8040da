// Backward pass orchestration (implementation include of pcnet.hip; lives in its translation unit).
//
// ake_pcnet_backward_f32 consumes the workspace left by ake_pcnet_forward_train_f32 (raw convolution outputs, BatchNorm
// batch statistics and affine tables) and the gradients of the loss with respect to the three outputs, and writes
// dL/d(parameter) for every float entry of the state_dict into one flat buffer (layout: ake_pcnet_grad_offset).
// Any num_layers >= 1 (the reference default is 2): the layer walk below loops over models.py:370-396 backwards.
#pragma once

namespace {

struct Bwd {
    const ake_pcnet* n;
    Buffers& b;
    hipStream_t s;
    gfx_t* grads;        // slot 0 of the workspace's gradient slots ([kGradSlots][grad_floats], fixed point); run() is followed by grad_reduce_kernel
    int B;
    unsigned short* planes_scratch = nullptr;   // f16 hi / lo planes of a 16-channel gradient (set around the pitch-class stacks of layers >= 1)
    // the tensor bn_block_backward last turned into dz, and the cell that holds the bits of its largest |dz| (the f16 x 3 data gradients
    // scale dz by a power of two taken from it before the hi / lo split)
    const float* amax_of = nullptr;
    const unsigned int* amax_cell = nullptr;
    const unsigned int* amax_for(const float* dz) const { return dz == amax_of ? amax_cell : nullptr; }

    gfx_t* grad_of(const std::string& key) const { return grads + n->grad_off[n->spec_index.at(key)]; }
    const float* raw_of(const std::string& key) const { return n->blob_dev + n->raw_w_off[n->spec_index.at(key)]; }
    int bn_of(const std::string& name) const { return n->bn_index.at(name); }

    // ga (gradient w.r.t. the activation a = lrelu(bn(z))) -> dz (gradient w.r.t. the raw conv output z), in place in `g`.
    // g and z share the layout [B][ctot][HT]; the BatchNorm layer `bn_name` owns channels [coff, coff + C).
    void bn_block_backward(const std::string& bn_name, float* g, const float* z, const float* aff, int ctot, int coff, int HT) {
        const int bn = bn_of(bn_name);
        const auto& l = n->bns[bn];
        double* st2 = b.stats2 + 3 * l.ch_off;
        const float* bst = b.bstats + 3 * l.ch_off;
        float* coef = b.coef + 4 * l.ch_off;
        dim3 grid(l.C, B);
        {
            ake::ProfScope ps("act_bwd_stats_kernel", s);
            hipLaunchKernelGGL(act_bwd_stats_kernel, grid, dim3(256), 0, s, g, z, aff, bst, st2, static_cast<long long>(3) * n->bn_channels, ctot, coff, HT);
        }
        {
            ake::ProfScope ps("bn_bwd_coef_kernel", s);
            hipLaunchKernelGGL(bn_bwd_coef_kernel, dim3((l.C + 63) / 64), dim3(64), 0, s, st2, static_cast<long long>(3) * n->bn_channels, bst, n->blob_dev + l.gamma_off, coef,
                               grad_of(bn_name + ".weight"), grad_of(bn_name + ".bias"), l.C);
        }
        unsigned int* cell = reinterpret_cast<unsigned int*>(b.stats2 + static_cast<size_t>(3) * n->bn_channels * kBwdStatSlots) + static_cast<size_t>(bn) * kAmaxSlots;
        {
            ake::ProfScope ps("bn_bwd_apply_kernel", s);
            hipLaunchKernelGGL(bn_bwd_apply_kernel, grid, dim3(256), 0, s, g, z, aff, coef, ctot, coff, HT, cell);
        }
        amax_of = g; amax_cell = cell;
    }

    // weight gradient of one convolution: dW += corr(act(input), dz)
    // zero_pad (kind 0 only): the convolution pads rows AND frames with zeros (--denseblock's plain Conv2d) instead of wrapping both
    int wgrad(const PackedConv& pc, int kind, Src src, const float* in_aff, int H, int T_in, bool same_time, const float* dz, int dz_ctot,
              int dz_coff, gfx_t* dW, const char* name, bool zero_pad = false) {
        static const bool wg_f32 = ake::diag_env("AKE_WGRAD_F32") != nullptr;
        if (!wg_f32 && !zero_pad && kind == 0 && pc.kh == 7 && pc.kw == 7 && pc.cout == 8 && pc.cin <= 8 && T_in <= kWgMaxT && dz_ctot == 8 && dz_coff == 0 && src.ctot0 == 0) {
            WgradBfArgs w;
            std::memset(&w, 0, sizeof(w));
            w.src0 = src.p0; w.c0 = src.c0; w.src1 = src.p1; w.c1 = src.c1; w.h1 = src.h1 > 0 ? src.h1 : 1;
            w.src0_clip_stride = static_cast<long long>(src.c0) * H * T_in;
            w.src1_clip_stride = static_cast<long long>(src.c1) * w.h1 * T_in;
            w.in_affine = in_aff; w.dz = dz; w.dW = dW; w.slot_stride = static_cast<long long>(n->grad_floats);
            w.cin = pc.cin; w.H = H; w.T = T_in;
            const int wgs_per_clip = std::max(3, std::min(H, (3 * std::max(tiling_cus(1), 1) + B - 1) / B));   // three workgroups fit a CU
            w.rows_per_wg = (H + wgs_per_clip - 1) / wgs_per_clip;
            dim3 grid((H + w.rows_per_wg - 1) / w.rows_per_wg, 1, B);
            const size_t lds = (static_cast<size_t>(2 * kWgARows + 2 * 64) * kWgKP + kWgKP) * sizeof(unsigned short);
            static ake::DeviceOnce attr_set;
            if (attr_set.need()) {
                AKE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_p2p_bf16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
                attr_set.mark();
            }
            const long long n_w = static_cast<long long>(8) * pc.cin * 49, n_wg = static_cast<long long>(grid.x) * B;
            static const bool partial_off = ake::diag_env("AKE_WGRAD_ATOMIC") != nullptr;
            const bool use_partial = !partial_off && b.wg_partial && n_wg * n_w <= static_cast<long long>(b.wg_partial_floats);
            if (use_partial) { w.partial = b.wg_partial; w.partial_stride = n_w; }
            {
                ake::ProfScope ps("conv_wgrad_p2p_bf16_kernel", s);
                hipLaunchKernelGGL(conv_wgrad_p2p_bf16_kernel, grid, dim3(256), lds, s, w);
            }
            if (use_partial) {
                ake::ProfScope ps("wgrad_partial_reduce_kernel", s);
                hipLaunchKernelGGL(wgrad_partial_reduce_kernel, dim3(static_cast<unsigned>((n_w + 63) / 64)), dim3(1024), 0, s, b.wg_partial, static_cast<int>(n_wg), n_w, dW);
            }
            return AKE_OK;
        }
        // 12 x 7 pitch-class convolutions (the pitch-class stacks, the heads' first convs): split-bf16 MFMA, one workgroup per (clip, 16 x 16
        // channel block), see conv_wgrad_pc_f16x3_kernel
        {
            const int T_out = kind == 0 ? T_in : (same_time ? T_in : T_in - pc.kw + 1);
            static const int wpc_min_cout = ake::diag_env("AKE_WPC_MIN_COUT") ? std::atoi(ake::diag_env("AKE_WPC_MIN_COUT")) : 1;
            if (!wg_f32 && kind == 1 && pc.kh == 12 && pc.kw == 7 && H == 12 && src.c1 == 0 && T_out >= 1 && pc.cout >= wpc_min_cout) {
                WgradPcArgs w;
                std::memset(&w, 0, sizeof(w));
                w.x = src.p0; w.x_clip_stride = static_cast<long long>(src.ctot0 > 0 ? src.ctot0 : src.c0) * 12 * T_in;
                w.in_affine = in_aff; w.dz = dz; w.dz_clip_stride = static_cast<long long>(dz_ctot) * 12 * T_out; w.dz_coff = dz_coff;
                w.cin = pc.cin; w.cout = pc.cout; w.T_in = T_in; w.T_out = T_out; w.pad = same_time ? pc.kw / 2 : 0;
                const int ks_max = T_out > 32 ? 2 : 1;             // k-steps of 32 output frames per workgroup (segments of kWpSeg frames)
                w.AP = ks_max == 1 ? 56 : 88;                      // >= 16 + 32 k-steps and 8 x odd: the 16 lanes of a fragment read hit 16 different LDS slots
                w.ZP = ks_max == 1 ? 40 : 72;
                w.n_co_blocks = (pc.cout + 15) / 16;
                w.n_seg = (T_out + kWpSeg - 1) / kWpSeg;
                w.dW = dW; w.slot_stride = static_cast<long long>(n->grad_floats);
                const long long n_w = static_cast<long long>(pc.cout) * pc.cin * 84;
                static const bool partial_off = ake::diag_env("AKE_WGRAD_ATOMIC") != nullptr;
                const long long n_wg = static_cast<long long>(B) * w.n_seg;
                const bool use_partial = !partial_off && b.wg_partial && n_wg * n_w <= static_cast<long long>(b.wg_partial_floats);
                if (use_partial) { w.partial = b.wg_partial; w.partial_stride = n_w; }
                const size_t lds = static_cast<size_t>(2) * 12 * 16 * (w.AP + w.ZP) * sizeof(unsigned short);
                static ake::DeviceOnce attr_set;
                if (attr_set.need()) {
                    AKE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_pc_f16x3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
                    attr_set.mark();
                }
                {
                    const char* slash = std::strrchr(name, '/');
                    const std::string pname = std::string("conv_wgrad_pc_f16x3_kernel") + (slash ? slash : "");
                    ake::ProfScope ps(pname.c_str(), s);
                    hipLaunchKernelGGL(conv_wgrad_pc_f16x3_kernel, dim3(w.n_seg * w.n_co_blocks, (pc.cin + 15) / 16, B), dim3(512), lds, s, w);
                }
                if (use_partial) {
                    ake::ProfScope ps("wgrad_partial_reduce_kernel", s);
                    hipLaunchKernelGGL(wgrad_partial_reduce_kernel, dim3(static_cast<unsigned>((n_w + 63) / 64)), dim3(1024), 0, s, b.wg_partial, static_cast<int>(n_wg), n_w, dW);
                }
                return AKE_OK;
            }
        }
        // the kernel's accumulator tiles cover <= 32 output channels: wider convolutions (deeper / wider nets) run as slices of 32
        for (int co0 = 0; co0 < pc.cout; co0 += 32) {
            const int co_n = std::min(32, pc.cout - co0);
            WgradArgs wa;
            std::memset(&wa, 0, sizeof(wa));
            ConvArgs& a = wa.c;
            a.src0 = src.p0; a.c0 = src.c0; a.src1 = src.p1; a.c1 = src.c1; a.h1 = src.h1 > 0 ? src.h1 : 1;
            a.H = H; a.T_in = T_in;
            a.src0_clip_stride = static_cast<long long>(src.ctot0 > 0 ? src.ctot0 : src.c0) * H * T_in;
            a.src1_clip_stride = static_cast<long long>(src.c1) * a.h1 * T_in;
            if (kind == 0) { a.py = pc.kh / 2; a.pad_l = pc.kw / 2; a.time_circ = zero_pad ? 0 : 1; a.rows_zero = zero_pad ? 1 : 0; a.T_out = T_in; a.H_out = H; }
            else {
                a.py = 0; a.time_circ = 0;
                a.pad_l = same_time ? pc.kw / 2 : 0;
                a.T_out = same_time ? T_in : T_in - pc.kw + 1;
                a.H_out = kind == 1 ? H : H - pc.kh + 1;
            }
            a.cout = co_n;
            a.dst = const_cast<float*>(dz); a.dst_coff = dz_coff + co0; a.dst_clip_stride = static_cast<long long>(dz_ctot) * a.H_out * a.T_out;
            a.in_affine = in_aff;
            wa.dW = dW + static_cast<long long>(co0) * (src.c0 + src.c1) * pc.kh * pc.kw; wa.slot_stride = static_cast<long long>(n->grad_floats); wa.KH = pc.kh; wa.KW = pc.kw;
            static const bool noflush = ake::diag_env("AKE_WGRAD_NOFLUSH") != nullptr;
            wa.dbg_noflush = noflush ? 1 : 0;
            const int KK = pc.kh * pc.kw;
            const int MTC = (co_n + 15) / 16, NTK = (KK + 15) / 16;
            // tile: rows x frames such that the patch of 8 channels + the dz tile fit the LDS budget
            const int T4 = (a.T_out + 3) / 4 * 4;
            int TT = std::min(T4, 128);
            int R = kind == 0 ? std::min(H, 16) : H;
            auto lds_of = [&](int R_, int TT_) {
                const int Tp = (TT_ + pc.kw - 1 + 3) / 4 * 4;
                return (static_cast<size_t>(8) * (R_ + pc.kh - 1) * Tp + static_cast<size_t>(co_n) * R_ * TT_) * sizeof(float);
            };
            while (lds_of(R, TT) > kLdsBudget && kind == 0 && R > 1) --R;
            while (lds_of(R, TT) > kLdsBudget && TT > 4) TT -= 4;
            // small batches (the reference trains with 8 clips per step): shorter time tiles and one workgroup per group of 8 input
            // channels, until the launch has about one workgroup per CU
            const int cin_all = src.c0 + src.c1;
            const int n_cus = std::max(tiling_cus(1), 1);
            auto tiles_of = [&](int R_, int TT_) { return ((a.H_out + R_ - 1) / R_) * ((a.T_out + TT_ - 1) / TT_); };
            const int c_groups = static_cast<long long>(B) * tiles_of(R, TT) < n_cus ? (cin_all + 7) / 8 : 1;
            while (static_cast<long long>(B) * tiles_of(R, TT) * c_groups < n_cus && TT > 16) TT = std::max(16, (TT / 2 + 3) / 4 * 4);
            AKE_REQUIRE(lds_of(R, TT) <= 160 * 1024, AKE_ERR_UNSUPPORTED, "wgrad %s: tile does not fit LDS", name);
            a.R = R; a.TT = TT; a.Tp = (TT + pc.kw - 1 + 3) / 4 * 4;
            a.n_row_tiles = (a.H_out + R - 1) / R;
            a.n_time_tiles = (a.T_out + TT - 1) / TT;
            const int tiles = a.n_row_tiles * a.n_time_tiles;
            // workgroups per clip: 4 at training batch sizes (fewer atomics), more for small batches so that the chip still fills
            const int wgs_per_clip = std::min(tiles, std::max(4, (2 * std::max(tiling_cus(1), 1) + B - 1) / B));
            wa.rt_per_block = std::max(1, (tiles + wgs_per_clip - 1) / wgs_per_clip);
            wa.c_per_block = c_groups > 1 ? 8 : cin_all;
            dim3 grid((tiles + wa.rt_per_block - 1) / wa.rt_per_block, c_groups, B), block(512);
            const size_t lds = lds_of(R, TT);
            // partial sums per workgroup + an ordered reduction when the scratch buffer holds them (see WgradArgs::partial)
            const long long n_w = static_cast<long long>(co_n) * (src.c0 + src.c1) * KK;
            const long long n_wg = static_cast<long long>(grid.x) * B;
            static const bool partial_off = ake::diag_env("AKE_WGRAD_ATOMIC") != nullptr;
            const bool use_partial = !partial_off && b.wg_partial && n_wg * n_w <= static_cast<long long>(b.wg_partial_floats);
            if (use_partial) { wa.partial = b.wg_partial; wa.partial_stride = n_w; }
            bool launched = false;
            {
                ake::ProfScope ps(name, s);
    #define AKE_WG(M_, N_) if (!launched && MTC == M_ && NTK == N_) { hipLaunchKernelGGL((conv_wgrad_kernel<M_, N_>), grid, block, lds, s, wa); launched = true; }
                AKE_WG(1, 4) AKE_WG(2, 4) AKE_WG(1, 6) AKE_WG(2, 6) AKE_WG(1, 1) AKE_WG(2, 1) AKE_WG(1, 2) AKE_WG(2, 2) AKE_WG(1, 3) AKE_WG(2, 3)
    #undef AKE_WG
            }
            if (!launched) {
                ake::set_error("wgrad %s: no kernel for cout=%d taps=%d", name, co_n, KK);
                return AKE_ERR_UNSUPPORTED;
            }
            if (use_partial) {
                ake::ProfScope ps("wgrad_partial_reduce_kernel", s);
                hipLaunchKernelGGL(wgrad_partial_reduce_kernel, dim3(static_cast<unsigned>((n_w + 63) / 64)), dim3(1024), 0, s, b.wg_partial, static_cast<int>(n_wg), n_w, wa.dW);
            }
        }
        return AKE_OK;
    }

    // data gradient of one convolution: dst(+)= conv(dz, flipped weights)
    int dgrad(const PackedConv& pd, const PackedConv& fwd, int kind, const float* dz, int H, int T_dz, int T_in, bool same_time, float* dst,
              int dst_ctot, int dst_coff, bool accumulate, const char* name) {
        ConvGeom g;
        g.py = fwd.kh - 1 - (kind == 0 ? fwd.kh / 2 : 0);
        g.time_circ = kind == 0 ? 1 : 0;
        const int pad_fwd = kind == 0 ? fwd.kw / 2 : (same_time ? fwd.kw / 2 : 0);
        g.pad_l = fwd.kw - 1 - pad_fwd;
        g.T_out = T_in;
        g.H_out = H;
        if (kind == 0 && !accumulate && dst_coff == 0 && dst_ctot == pd.cout && T_dz == T_in &&
            run_p2p_f16x3(n, pd.bf_off, Src{dz, pd.cin, nullptr, 0, 0}, nullptr, nullptr, B, H, T_dz, dst, pd.cout, nullptr, 0, s, "conv_p2p_f16x3_kernel/p2p_dgrad",
                          amax_for(dz)))
            return AKE_OK;
        // pitch-class stacks of layers >= 1 (same-size 12 x 7 convolutions, 16 gradient channels): the f16 x 3 form of conv_pc_bf16_kernel with
        // the transposed + flipped weights (f32-equivalent products; the f32 MFMA kernel took 0.17 ms per convolution and 256 clips)
        if (kind == 1 && same_time && planes_scratch && !accumulate && dst_coff == 0 && dst_ctot == pd.cout && T_dz == T_in && H == 12 &&
            pc_f16x3_ok(pd, T_dz, true)) {
            run_nchw_to_cl16_f16x2(dz, pd.cin, B, T_dz, nullptr, planes_scratch, s, 0, amax_for(dz));
            return run_pc_bf16(n, pd, planes_scratch, B, T_dz, true, false, dst, nullptr, s, "conv_pc_f16x3_kernel/pc2pc_dgrad", nullptr, nullptr, true,
                               nullptr, 0, amax_for(dz));
        }
        // the heads' first convolutions (16 features -> 32 channels, "valid" in time): 32 gradient channels as two 16-channel halves on the same
        // kernel, full correlation, the second half (and every head after the first) adding to the feature gradient
        if (kind == 1 && !same_time && planes_scratch && pd.bf_off >= 0 && pd.bf_off2 >= 0 && pd.cin == 32 && pd.cout == 16 && dst_coff == 0 &&
            dst_ctot == 16 && T_in == T_dz + 6 && H == 12) {
            const long long half = static_cast<long long>(B) * 12 * T_dz * 16 * 2;       // one set of hi + lo planes
            int rc2;
            for (int hf = 0; hf < 2; ++hf) {
                unsigned short* pl = planes_scratch + hf * half;
                run_nchw_to_cl16_f16x2(dz + static_cast<long long>(hf) * 16 * 12 * T_dz, 16, B, T_dz, nullptr, pl, s, 32, amax_for(dz));
                if ((rc2 = run_pc_f16x3_full(n, hf ? pd.bf_off2 : pd.bf_off, fwd.kh, pl, B, T_dz, dst, accumulate || hf == 1, s, "conv_pc_f16x3_kernel/head_dgrad",
                                             amax_for(dz))))
                    return rc2;
            }
            return AKE_OK;
        }
        return run_conv(n, pd, kind == 0 ? 0 : 1, Src{dz, pd.cin, nullptr, 0, 0}, B, H, T_dz, true, false, dst, dst_ctot, dst_coff, s, name,
                        nullptr, nullptr, &g, accumulate);
    }

    // --denseblock (models.py:456-648): one block backwards, in place on g_feat [B][ctot][H][T] = dL/d(features), complete for the channels the
    // block appended when this is called; on return channels [0, cin) hold dL/d(block input) ADDED to what they held.  Layer j (last first):
    //   its new features' gradient (channels [cj, cj + nf)) -> conv2 (weights, bias; data gradient into g_bott) -> norm2 + ReLU -> conv1 (weights;
    //   data gradient into the first cj channels of g_scr, a buffer of g_feat's shape) -> norm1 + LeakyReLU over ALL cj channels it read
    //   (in place in g_scr) -> added to g_feat[0, cj): the channel-sliced accumulate of a dense connection.
    int dense_block_backward(const std::vector<DensePack>& packs, int kind, const float* feat, int ctot, int cin, int H, int T, float* g_feat,
                             float* g_bott, float* g_scr, const std::vector<float*>& botts, const std::vector<float*>& aff1s,
                             const std::vector<float*>& aff2s, const char* wname, const char* dname) {
        const int nf = n->cfg.n_filters, k = n->cfg.kernel_size;
        const long long HT = static_cast<long long>(H) * T;
        int rc;
        for (int j = static_cast<int>(packs.size()) - 1; j >= 0; --j) {
            const DensePack& dp = packs[j];
            const int cj = cin + j * nf, bott = dp.c1.cout;
            // ---- conv2: new = conv2(relu(norm2(bott_j))) ----
            PackedConv t2 = dp.c2;                         // true kernel size for the weight gradient (the pack may store 7 taps)
            t2.kh = kind == 1 ? 12 : k; t2.kw = k;
            if ((rc = wgrad(t2, kind, Src{botts[j], bott, nullptr, 0, 0}, aff2s[j], H, T, true, g_feat, ctot, cj, grad_of(dp.w2), wname, kind == 0))) return rc;
            if (!dp.b2.empty()) bias_grad(g_feat, ctot, cj, nf, static_cast<int>(HT), grad_of(dp.b2));
            {
                const ConvGeom g2 = kind == 0 ? ConvGeom{k - 1 - k / 2, k - 1 - k / 2, T, H, 0} : ConvGeom{11, k - 1 - k / 2, T, H, 0};
                if ((rc = run_conv(n, dp.d2, kind, Src{g_feat + static_cast<long long>(cj) * HT, nf, nullptr, 0, 0, ctot}, B, H, T, true, false, g_bott, bott, 0, s,
                                   dname, nullptr, nullptr, &g2, false, nullptr, kind == 0)))
                    return rc;
            }
            bn_block_backward(dp.norm2, g_bott, botts[j], aff2s[j], bott, 0, static_cast<int>(HT));
            // ---- conv1: bott_j = conv1(lrelu(norm1(feat[0, cj)))) (its bias, where it has one, is removed by norm2: zero gradient) ----
            PackedConv t1 = dp.c1;
            t1.kh = kind == 1 ? 12 : 1; t1.kw = 1;
            if ((rc = wgrad(t1, kind, Src{feat, cj, nullptr, 0, 0, ctot}, aff1s[j], H, T, true, g_bott, bott, 0, grad_of(dp.w1), wname, kind == 0))) return rc;
            {
                const ConvGeom g1 = kind == 0 ? ConvGeom{0, 3, T, H, 0} : ConvGeom{11, 0, T, H, 0};   // (1 x 1: the centre of the 7 stored taps; 12 x 1: the row-k form, no frame extent)
                if ((rc = run_conv(n, dp.d1, kind, Src{g_bott, bott, nullptr, 0, 0}, B, H, T, true, false, g_scr, ctot, 0, s, dname, nullptr, nullptr,
                                   &g1, false, nullptr, kind == 0)))
                    return rc;
            }
            bn_block_backward(dp.norm1, g_scr, feat, aff1s[j], ctot, 0, static_cast<int>(HT));
            {
                const long long total = static_cast<long long>(B) * cj * HT;
                ake::ProfScope ps("add_channels_kernel", s);
                hipLaunchKernelGGL(add_channels_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, s, g_feat, ctot, g_scr, ctot, cj, HT, total);
            }
        }
        return AKE_OK;
    }

    // the layer walk of a --denseblock net (models.py:361-396 with dense stacks, backwards), num_layers >= 2.  g_cat[li] is dL/d(layer li's concat
    // buffer) = pitch classes of the layer below | folded semitone maps | the layer's growth; layer 0's block lives in the first channels of
    // cat[1].  An inner layer's two streams each have a second consumer, the time-pooled copies the layer above starts from (models.py:394-395).
    int run_dense(const float* mel) {
        const auto& c = n->cfg;
        const int L = c.num_layers, P = c.pitches, tp = c.time_pool_size, g = c.n_filters * c.conv_layers;
        int rc;
        for (int li = L - 1; li >= 1; --li) {
            const LayerDims& d = n->dims[li];
            const int Tl = b.Tl[li], ctd = d.prev_pc + d.out_p + g;
            const std::string m = "model." + std::to_string(li) + ".";
            {   // the time-pooled copy of the whole concat buffer: the heads' input (last layer) or the first channels of the next layer's buffer
                const long long total = static_cast<long long>(B) * ctd * 12 * ((Tl / tp) + (Tl % tp ? 1 : 0));
                const bool last = li == L - 1;
                const int up_ctot = last ? ctd : n->dims[li + 1].prev_pc + n->dims[li + 1].out_p + g;
                ake::ProfScope ps("time_pool_bwd_kernel", s);
                hipLaunchKernelGGL(time_pool_bwd_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, s, last ? b.g_pcf : b.g_cat[li + 1], b.cat[li],
                                   static_cast<const float*>(nullptr), b.g_cat[li], ctd, 12, Tl, tp, up_ctot, 0, total, 0);
            }
            float* scr_pc = b.g_pc[li];                      // [B][ctd][12][Tl] scratch (g_pc[li] holds two of them)
            if ((rc = dense_block_backward(n->dense_pc[li], 1, b.cat[li], ctd, d.prev_pc + d.out_p, 12, Tl, b.g_cat[li], b.dn_gbott_pc, scr_pc, b.dn_bott_pc[li],
                                           b.dn_aff1_pc[li], b.dn_aff2_pc[li], "conv_wgrad_kernel/pc2pc", "conv_mfma_kernel/pc2pc_dgrad")))
                return rc;
            // folded semitone maps -> pool_semi(li) -> dL/d(pitch features), all out_p channels
            float* g_p = b.g_p[li];
            float* scr_p = b.g_p[li] + static_cast<size_t>(B) * d.out_p * P * Tl;
            if ((rc = semi_backward(li, b.pa[li], nullptr, b.g_cat[li], ctd, d.prev_pc, g_p))) return rc;
            if (li < L - 1) {   // second consumer of an inner layer's pitch features: their time-pooled copy is the next pitch block's input (channels [0, out_p))
                const long long total = static_cast<long long>(B) * d.out_p * P * ((Tl / tp) + (Tl % tp ? 1 : 0));
                ake::ProfScope ps("time_pool_bwd_kernel", s);
                hipLaunchKernelGGL(time_pool_bwd_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, s, b.g_p[li + 1], b.pa[li], static_cast<const float*>(nullptr),
                                   g_p, d.out_p, P, Tl, tp, n->dims[li + 1].out_p, 0, total, 1);
            }
            if ((rc = dense_block_backward(n->dense_p[li], 0, b.pa[li], d.out_p, d.prev_p + d.prev_pc, P, Tl, g_p, b.dn_gbott_p, scr_p, b.dn_bott_p[li], b.dn_aff1_p[li],
                                           b.dn_aff2_p[li], "conv_wgrad_kernel/p2p", "conv_mfma_kernel/p2p_dgrad")))
                return rc;
            {   // repeat (x P / 36) backward: channels [prev_p, prev_p + prev_pc) of the pitch block's input gradient -> the up_sixth map
                const long long total = static_cast<long long>(B) * d.prev_pc * 36 * Tl;
                ake::ProfScope ps("repeat_sum_kernel", s);
                hipLaunchKernelGGL(repeat_sum_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, s, g_p, b.g_psix[li], d.out_p, d.prev_p, d.prev_pc, P, Tl, total);
            }
            bn_block_backward(m + "up_sixth_b", b.g_psix[li], b.psix[li], b.aff_p2pin[li] + 3 * d.prev_p, d.prev_pc, 0, 36 * Tl);
            {
                ake::ProfScope ps("up_sixth_bwd_weight_kernel", s);
                hipLaunchKernelGGL(up_sixth_bwd_weight_kernel, dim3(d.prev_pc * d.prev_pc * 3, B), dim3(64), 0, s, b.g_psix[li], b.cat[li], static_cast<long long>(ctd) * 12 * Tl,
                                   static_cast<const float*>(nullptr), grad_of(m + "up_sixth.weight"), static_cast<long long>(n->grad_floats), d.prev_pc, Tl);
            }
            {
                const long long total = static_cast<long long>(B) * d.prev_pc * 12 * Tl;
                ake::ProfScope ps("up_sixth_bwd_data_kernel", s);
                hipLaunchKernelGGL(up_sixth_bwd_data_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, s, b.g_psix[li], raw_of(m + "up_sixth.weight"),
                                   b.g_cat[li], ctd, d.prev_pc, Tl, total);
            }
        }
        // layer 0's block: channels [0, 1 + g) of layer 1's buffer, input = the fold (channel 0)
        const LayerDims& d1 = n->dims[1];
        const int T1 = b.Tl[1], ctd1 = d1.prev_pc + d1.out_p + g;
        if ((rc = dense_block_backward(n->dense_pc[0], 1, b.cat[1], ctd1, 1, 12, T1, b.g_cat[1], b.dn_gbott_pc, b.g_pc[1], b.dn_bott_pc[0], b.dn_aff1_pc[0],
                                       b.dn_aff2_pc[0], "conv_wgrad_kernel/pc2pc0", "conv_mfma_kernel/pc2pc0_dgrad")))
            return rc;
        return semi_backward(0, mel, nullptr, b.g_cat[1], ctd1, 0, nullptr);
    }

    void bias_grad(const float* dz, int ctot, int coff, int C, int HT, gfx_t* db) {
        ake::ProfScope ps("channel_sum_kernel", s);
        hipLaunchKernelGGL(channel_sum_kernel, dim3(C, B), dim3(256), 0, s, dz, db, static_cast<long long>(n->grad_floats), ctot, coff, HT);
    }

    // A stack of `nconv` convolutions (conv -> BN -> LReLU each).  g holds ga w.r.t. the last activation on entry and is
    // ping-ponged with g2.  On exit *g_in_out holds ga w.r.t. the stack input (shape of the input: in_ctot channels).
    int stack_backward(const std::vector<PackedConv>& fwd_t, const std::vector<PackedConv>& dpack, const std::string& prefix, bool conv2d_suffix,
                       int kind, Src stack_in, const float* stack_in_aff, const std::vector<float*>& z, const std::vector<float*>& aff,
                       int H, int Tn, float* g, float* g2, float* g_in, int g_in_ctot, const char* wname, const char* dname) {
        const int nconv = static_cast<int>(fwd_t.size());
        int rc;
        if (n->cfg.resblock)   // the gradient buffers of a --resblock net are 4 maps wide: g | skip copy | the blocks' 2C-channel hidden map
            return res_stack_backward(fwd_t, dpack, prefix, conv2d_suffix, kind, stack_in, stack_in_aff, z, aff, H, Tn, g, g2,
                                      g2 + static_cast<size_t>(B) * fwd_t[0].cout * H * Tn, g_in, g_in_ctot, wname, dname);
        for (int j = nconv - 1; j >= 0; --j) {
            const PackedConv& pc = fwd_t[j];
            const std::string cname = prefix + std::to_string(3 * j) + (conv2d_suffix ? ".conv2d" : "");
            bn_block_backward(prefix + std::to_string(3 * j + 1), g, z[j], aff[j], pc.cout, 0, H * Tn);
            // bias gradient under BatchNorm is exactly zero (the mean subtraction removes it); grads were zero-filled
            Src in = j == 0 ? stack_in : Src{z[j - 1], fwd_t[j - 1].cout, nullptr, 0, 0};
            const float* in_aff = j == 0 ? stack_in_aff : aff[j - 1];
            if ((rc = wgrad(pc, kind, in, in_aff, H, Tn, true, g, pc.cout, 0, grad_of(cname + ".weight"), wname))) return rc;
            float* dst = j == 0 ? g_in : g2;
            const int dst_ctot = j == 0 ? g_in_ctot : fwd_t[j - 1].cout;
            if (dst) {
                if ((rc = dgrad(dpack[j], pc, kind, g, H, Tn, Tn, true, dst, dst_ctot, 0, false, dname))) return rc;
            }
            if (j > 0) std::swap(g, g2);
        }
        return AKE_OK;
    }

    // --resblock stack (models.py:181-187 / 218-224, 402-454): conv0 + BN + LReLU, then per block x <- LReLU(x + b2(conv2(LReLU(b1(conv1(x)))))).
    // fwd_t / dpack = [conv0, (conv1, conv2) per block]; z / aff = res_stack_train's tensors [conv0 raw, (conv1 raw, conv2 raw, block
    // output) per block].  g: ga w.r.t. the last block's output on entry; gs: skip copy; gh: the 2C-channel hidden gradient.
    int res_stack_backward(const std::vector<PackedConv>& fwd_t, const std::vector<PackedConv>& dpack, const std::string& prefix, bool conv2d_suffix,
                           int kind, Src stack_in, const float* stack_in_aff, const std::vector<float*>& z, const std::vector<float*>& aff, int H, int Tn,
                           float* g, float* gs, float* gh, float* g_in, int g_in_ctot, const char* wname, const char* dname) {
        const int nb = (static_cast<int>(fwd_t.size()) - 1) / 2;
        const int C = fwd_t[0].cout;
        const std::string sfx = conv2d_suffix ? ".conv2d" : "";
        AKE_REQUIRE(static_cast<int>(z.size()) == 1 + 3 * nb && dpack.size() == fwd_t.size(), AKE_ERR_STATE, "resblock backward: buffer bookkeeping");
        int rc;
        for (int r = nb - 1; r >= 0; --r) {
            const std::string bp = prefix + std::to_string(3 + r) + ".";
            {
                const long long total = static_cast<long long>(B) * C * H * Tn;
                ake::ProfScope ps("res_act_bwd_kernel", s);
                hipLaunchKernelGGL(res_act_bwd_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, s, g, z[3 + 3 * r], gs, total);
            }
            // b2 has no activation of its own: its table carries slope 1, so the generic block applies LeakyReLU' = 1
            bn_block_backward(bp + "b2", g, z[2 + 3 * r], aff[2 + 3 * r], C, 0, H * Tn);
            if ((rc = wgrad(fwd_t[2 + 2 * r], kind, Src{z[1 + 3 * r], 2 * C, nullptr, 0, 0}, aff[1 + 3 * r], H, Tn, true, g, C, 0,
                            grad_of(bp + "conv2" + sfx + ".weight"), wname)))
                return rc;
            if ((rc = dgrad(dpack[2 + 2 * r], fwd_t[2 + 2 * r], kind, g, H, Tn, Tn, true, gh, 2 * C, 0, false, dname))) return rc;
            bn_block_backward(bp + "b1", gh, z[1 + 3 * r], aff[1 + 3 * r], 2 * C, 0, H * Tn);
            // the block's input: the previous block's output (final, identity table) or conv0's raw output + its BatchNorm
            if ((rc = wgrad(fwd_t[1 + 2 * r], kind, Src{z[3 * r], C, nullptr, 0, 0}, aff[3 * r], H, Tn, true, gh, 2 * C, 0,
                            grad_of(bp + "conv1" + sfx + ".weight"), wname)))
                return rc;
            if ((rc = dgrad(dpack[1 + 2 * r], fwd_t[1 + 2 * r], kind, gh, H, Tn, Tn, true, gs, C, 0, true, dname))) return rc;   // + the skip copy
            std::swap(g, gs);
        }
        bn_block_backward(prefix + "1", g, z[0], aff[0], C, 0, H * Tn);
        if ((rc = wgrad(fwd_t[0], kind, stack_in, stack_in_aff, H, Tn, true, g, C, 0, grad_of(prefix + "0" + sfx + ".weight"), wname))) return rc;
        if (g_in && (rc = dgrad(dpack[0], fwd_t[0], kind, g, H, Tn, Tn, true, g_in, g_in_ctot, 0, false, dname))) return rc;
        return AKE_OK;
    }

    int run(const float* mel, const int64_t* seq, const float* d_key, const float* d_tonic, const float* d_genre, const float* key_out) {
        const auto& c = n->cfg;
        const int L = c.num_layers, P = c.pitches, tp = c.time_pool_size;
        int rc;
        const int Tf = b.Tf;
        const int Tm = Tf - (c.kernel_size - 1) * c.head_layers;
        const int i = L - 1;
        const int Ti = b.Tl[i];
        const LayerDims& d = n->dims[i];
        const int fin = n->final_ch;

        // ---- masked mean + sigmoid ----
        PoolHeadBwdArgs pa;
        std::memset(&pa, 0, sizeof(pa));
        pa.d_out[0] = d_key; pa.d_out[1] = d_tonic; pa.d_out[2] = c.genre ? d_genre : nullptr;
        pa.key_out = key_out;
        pa.d_map[0] = b.g_map[0]; pa.d_map[1] = b.g_map[1]; pa.d_map[2] = c.genre ? b.g_map[2] : nullptr;
        pa.rows[0] = 12; pa.rows[1] = 12; pa.rows[2] = 11;
        pa.Tm = Tm; pa.seq = reinterpret_cast<const long long*>(seq);
        pa.n_pool_layers = L - 1; pa.tp = tp; pa.shrink = (c.kernel_size - 1) * c.head_layers; pa.batch = B;
        pa.maps[0] = b.map_k; pa.maps[1] = b.map_t; pa.maps[2] = c.genre ? b.map_g : nullptr; pa.max_pool = c.max_pool;
        if (c.local > 0) {   // --local: sliding-window max, per-frame outputs (no seq_length, no time pooling in the layers)
            AKE_REQUIRE(Tm >= c.local, AKE_ERR_INVALID, "pcnet --local backward: %d map frames are fewer than the pooling window %d", Tm, c.local);
            LocalPoolBwdArgs la;
            std::memset(&la, 0, sizeof(la));
            for (int h = 0; h < 3; ++h) { la.d_out[h] = pa.d_out[h]; la.maps[h] = pa.maps[h]; la.d_map[h] = pa.d_map[h]; }
            la.key_out = key_out;
            la.Tm = Tm; la.Tq = Tm - c.local + 1; la.W = c.local; la.batch = B;
            ake::ProfScope ps("local_pool_bwd_kernel", s);
            hipLaunchKernelGGL(local_pool_bwd_kernel, dim3(static_cast<unsigned>((static_cast<long long>(B) * 12 * Tm + 255) / 256), 3), dim3(256), 0, s, la);
        } else {
            ake::ProfScope ps("head_pool_bwd_kernel", s);
            hipLaunchKernelGGL(head_pool_bwd_kernel, dim3((B * 12 + 63) / 64, 3), dim3(64), 0, s, pa);
        }
        // features feeding the heads: pcf (final) for L > 1, the raw last pc2pc output (+affine) for L == 1
        const float* feat = L > 1 ? b.pcf : (c.denseblock ? b.fold0 : b.pcst[0].back());      // (--denseblock, one layer: the block's feature buffer itself)
        const float* feat_aff = (L > 1 || c.denseblock) ? nullptr : b.aff_pcst[0].back();
        float* g_feat = L > 1 ? b.g_pcf : b.g_pc[0];          // gradient w.r.t. the head input activation
        AKE_HIP_CHECK(hipMemsetAsync(g_feat, 0, sizeof(float) * B * fin * 12 * Tf, s));

        struct HeadRun { const std::vector<PackedConv>* ct; const std::vector<PackedConv>* cd; int kind; const char* nm; bool conv2d; };
        HeadRun heads[3] = {{&n->head_key_t, &n->head_key_d, 1, "key_classifier", true},
                            {&n->head_tonic_t, &n->head_tonic_d, 1, "tonic_classifier", true},
                            {&n->head_genre_t, &n->head_genre_d, 2, "genre_classifier", false}};
        {   // scratch for the f16 planes of the heads' 32-channel gradients: the last layer's inference ping-pong buffer (idle in training)
            const int T1 = Tf - (c.kernel_size - 1);
            const size_t need = static_cast<size_t>(2) * B * 12 * T1 * 16 * 2 * sizeof(unsigned short);
            const size_t have = static_cast<size_t>(B) * (i == 0 ? c.n_filters : d.out_pc) * 12 * Ti * sizeof(float);
            planes_scratch = (L > 1 && fin == 16 && !c.resblock && !c.denseblock && T1 > 0 && need <= have) ? reinterpret_cast<unsigned short*>(b.pcb[i]) : nullptr;
        }
        for (int h = 0; h < (c.genre ? 3 : 2); ++h) {
            const auto& ct = *heads[h].ct;
            const auto& cd = *heads[h].cd;
            const int nl = c.head_layers;
            float* g = b.g_map[h];                 // dz of the last conv (no BatchNorm): [B][1][12][Tm] (genre: row 11 zero)
            int Tcur = Tm;
            for (int j = nl - 1; j >= 0; --j) {
                const PackedConv& pc = ct[j];
                const std::string cname = std::string(heads[h].nm) + "." + std::to_string(3 * j) + (heads[h].conv2d ? ".conv2d" : "");
                const int T_in = Tcur + c.kernel_size - 1;
                const int H_out = heads[h].kind == 2 ? 12 - pc.kh + 1 : 12;
                if (j < nl - 1) bn_block_backward(std::string(heads[h].nm) + "." + std::to_string(3 * j + 1), g, b.hst[h][j], b.aff_hst[h][j], pc.cout, 0, 12 * Tcur);
                else bias_grad(g, pc.cout, 0, pc.cout, 12 * Tcur, grad_of(cname + ".bias"));   // row 11 of the genre map gradient is zero
                Src in = j == 0 ? Src{feat, fin, nullptr, 0, 0} : Src{b.hst[h][j - 1], ct[j - 1].cout, nullptr, 0, 0};
                const float* in_aff = j == 0 ? feat_aff : b.aff_hst[h][j - 1];
                // the genre gradient maps carry 12 rows (the last one zero), so circular rows reproduce its "valid" rows exactly
                const int kind_w = 1;
                if ((rc = wgrad(pc, kind_w, in, in_aff, 12, T_in, false, g, pc.cout, 0, grad_of(cname + ".weight"), "conv_wgrad_kernel/head"))) return rc;
                float* dst = j == 0 ? g_feat : b.g_hid;
                if ((rc = dgrad(cd[j], pc, 1, g, 12, Tcur, T_in, false, dst, j == 0 ? fin : ct[j - 1].cout, 0, j == 0, "conv_mfma_kernel/head_dgrad")))
                    return rc;
                g = b.g_hid;
                Tcur = T_in;
                (void)H_out;
            }
        }

        planes_scratch = nullptr;
        if (c.denseblock && L == 1) {   // g_pc[0] = dL/d(fold | the block's growth), [B][fin][12][T]; its second half is the block's scratch
            float* gf = b.g_pc[0];
            if ((rc = dense_block_backward(n->dense_pc[0], 1, b.fold0, fin, 1, 12, Ti, gf, b.dn_gbott_pc, gf + static_cast<size_t>(B) * fin * 12 * Ti, b.dn_bott_pc[0],
                                           b.dn_aff1_pc[0], b.dn_aff2_pc[0], "conv_wgrad_kernel/pc2pc0", "conv_mfma_kernel/pc2pc0_dgrad")))
                return rc;
            return semi_backward(0, mel, nullptr, gf, fin, 0, nullptr);
        }
        if (c.denseblock) return run_dense(mel);
        float* g_last = b.g_pc[i];                                   // ga w.r.t. the last pc2pc activation of the last layer
        float* g_last2 = b.g_pc[i] + static_cast<size_t>(B) * (i == 0 ? c.n_filters : d.out_pc) * 12 * Ti;
        if (L > 1) {
            const long long total = static_cast<long long>(B) * d.out_pc * 12 * ((Ti / tp) + (Ti % tp ? 1 : 0));
            ake::ProfScope ps("time_pool_bwd_kernel", s);
            hipLaunchKernelGGL(time_pool_bwd_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, s, b.g_pcf, b.pcst[i].back(),
                               b.aff_pcst[i].back(), g_last, d.out_pc, 12, Ti, tp, d.out_pc, 0, total, 0);
        }
        if (L == 1) {
            // single layer: pc2pc0 stack straight down to fold0
            if ((rc = stack_backward(n->pc2pc_t[0], n->pc2pc_d[0], "model.0.pc2pc.layer.", true, 1, Src{b.fold0, 1, nullptr, 0, 0}, nullptr,
                                     b.pcst[0], b.aff_pcst[0], 12, Ti, g_last, g_last2, b.g_fold0, 1, "conv_wgrad_kernel/pc2pc0", "conv_mfma_kernel/pc2pc0_dgrad")))
                return rc;
            return semi_backward(0, mel, nullptr, b.g_fold0, 1, 0, nullptr);
        }

        // ---- layers L-1 .. 1 (models.py:370-396 backwards).  g_pc[i] holds dL/d(activation of layer i's last pc2pc conv) on entry ----
        for (int li = L - 1; li >= 1; --li) {
            const LayerDims& dl = n->dims[li];
            const int Tl = b.Tl[li];
            const int Pp = c.stay_sixth ? P / 3 : P;              // rows of the pitch stream (--stay_sixth: semitone resolution)
            const int ctot = dl.prev_pc + dl.out_p;
            const bool inner = li < L - 1;
            const std::string m = "model." + std::to_string(li) + ".";
            float* gl = b.g_pc[li];
            float* gl2 = gl + static_cast<size_t>(B) * dl.out_pc * 12 * Tl;
            // pc2pc stack (input = concat buffer: pitch classes of the layer below | folded semitone maps)
            // (the inference ping-pong buffer of the layer is idle in training: the gradient's f16 planes go there)
            planes_scratch = (dl.out_pc == 16 && !c.resblock) ? reinterpret_cast<unsigned short*>(b.pcb[li]) : nullptr;
            rc = stack_backward(n->pc2pc_t[li], n->pc2pc_d[li], m + "pc2pc.layer.", true, 1, Src{b.cat[li], ctot, nullptr, 0, 0}, b.aff_cat[li],
                                b.pcst[li], b.aff_pcst[li], 12, Tl, gl, gl2, b.g_cat[li], ctot, "conv_wgrad_kernel/pc2pc", "conv_mfma_kernel/pc2pc_dgrad");
            planes_scratch = nullptr;
            if (rc) return rc;
            // channels [prev_pc, ctot) of g_cat: gradient of the folded semitone features -> pool_semi(li) -> pitch stream
            const std::vector<float*>& zp = b.pst[li];
            float* g_p = b.g_p[li];
            float* g_p2 = b.g_p[li] + static_cast<size_t>(B) * dl.out_p * P * Tl;
            if (c.stay_sixth) {   // models.py:391: the stack's output (semitone resolution) was folded as it is
                const long long total = static_cast<long long>(B) * dl.out_p * 12 * Tl;
                ake::ProfScope ps("fold_bwd_kernel", s);
                hipLaunchKernelGGL(fold_bwd_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, s, b.g_cat[li], zp.back(),
                                   b.aff_pst[li].back(), g_p, dl.out_p, Pp / 12, Tl, ctot, dl.prev_pc, total);
            } else if ((rc = semi_backward(li, zp.back(), b.aff_pst[li].back(), b.g_cat[li], ctot, dl.prev_pc, g_p))) return rc;
            if (inner) {   // second consumer of an inner layer's pitch stream: its time-pooled copy is the next layer's pitch input (models.py:395)
                const LayerDims& dn = n->dims[li + 1];
                const long long total = static_cast<long long>(B) * dl.out_p * Pp * ((Tl / tp) + (Tl % tp ? 1 : 0));
                ake::ProfScope ps("time_pool_bwd_kernel", s);
                hipLaunchKernelGGL(time_pool_bwd_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, s, b.g_pin[li + 1], zp.back(),
                                   b.aff_pst[li].back(), g_p, dl.out_p, Pp, Tl, tp, c.pc2p_mem ? dn.prev_p : dn.prev_p + dn.prev_pc, 0, total, 1);
            }
            // ---- pitch convs; input = (pitch stream | psix repeated) ----
            // (--pc2p_mem, models.py:145-166: the stack read the pitch stream + the summed up_sixth map, kept in b.pin by the forward)
            Src pin = c.pc2p_mem ? Src{b.pin[li], dl.prev_p, nullptr, 0, 0} : Src{li == 1 ? mel : b.ppool[li - 1], dl.prev_p, b.psix[li], dl.prev_pc, 36};
            if (c.stay_sixth) pin = Src{li == 1 ? b.semi_raw[0] : b.ppool[li - 1], dl.prev_p, b.pcd[li], dl.prev_pc, 12};   // (table: aff_p2pin, set by the forward)
            const int pin_ch = c.pc2p_mem ? dl.prev_p : dl.prev_p + dl.prev_pc;
            if ((rc = stack_backward(n->p2p_t[li], n->p2p_d[li], m + "p2p.layer.", false, 0, pin, b.aff_p2pin[li], zp, b.aff_pst[li], Pp, Tl, g_p, g_p2,
                                     b.g_pin[li], pin_ch, "conv_wgrad_kernel/p2p", "conv_mfma_kernel/p2p_dgrad")))
                return rc;
            // ---- repeat (x P/36) backward, then up_sixth ----
            if (c.stay_sixth) {   // no up_sixth: the pitch classes were repeated as they are
                const long long total = static_cast<long long>(B) * dl.prev_pc * 12 * Tl;
                ake::ProfScope ps("repeat_sum_kernel", s);
                hipLaunchKernelGGL(repeat_sum_add_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, s, b.g_pin[li], b.g_cat[li],
                                   dl.prev_p + dl.prev_pc, dl.prev_p, dl.prev_pc, Pp, Tl, ctot, total);
            } else if (c.pc2p_mem) {
                const long long total = static_cast<long long>(B) * dl.prev_pc * 36 * Tl;
                ake::ProfScope ps("pc2p_mem_bwd_kernel", s);
                hipLaunchKernelGGL(pc2p_mem_bwd_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, s, b.g_pin[li], b.g_psix[li],
                                   dl.prev_p, dl.prev_pc / dl.prev_p, P, Tl, total);
            } else {
                const long long total = static_cast<long long>(B) * dl.prev_pc * 36 * Tl;
                ake::ProfScope ps("repeat_sum_kernel", s);
                hipLaunchKernelGGL(repeat_sum_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, s, b.g_pin[li], b.g_psix[li],
                                   dl.prev_p + dl.prev_pc, dl.prev_p, dl.prev_pc, P, Tl, total);
            }
            if (!c.stay_sixth) {
                bn_block_backward(m + "up_sixth_b", b.g_psix[li], b.psix[li], b.aff_p2pin[li] + 3 * dl.prev_p, dl.prev_pc, 0, 36 * Tl);
                {
                    ake::ProfScope ps("up_sixth_bwd_weight_kernel", s);
                    hipLaunchKernelGGL(up_sixth_bwd_weight_kernel, dim3(dl.prev_pc * dl.prev_pc * 3, B), dim3(64), 0, s, b.g_psix[li], b.cat[li],
                                       static_cast<long long>(ctot) * 12 * Tl, b.aff_cat[li], grad_of(m + "up_sixth.weight"),
                                       static_cast<long long>(n->grad_floats), dl.prev_pc, Tl);
                }
                {
                    const long long total = static_cast<long long>(B) * dl.prev_pc * 12 * Tl;
                    ake::ProfScope ps("up_sixth_bwd_data_kernel", s);
                    hipLaunchKernelGGL(up_sixth_bwd_data_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, s, b.g_psix[li],
                                       raw_of(m + "up_sixth.weight"), b.g_cat[li], ctot, dl.prev_pc, Tl, total);
                }
            }
            if (li >= 2) {   // channels [0, prev_pc) of the concat buffer = the time-pooled pitch classes of the layer below (models.py:394)
                const LayerDims& dp = n->dims[li - 1];
                const int Tp = b.Tl[li - 1];
                const long long total = static_cast<long long>(B) * dp.out_pc * 12 * ((Tp / tp) + (Tp % tp ? 1 : 0));
                ake::ProfScope ps("time_pool_bwd_kernel", s);
                hipLaunchKernelGGL(time_pool_bwd_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, s, b.g_cat[li], b.pcst[li - 1].back(),
                                   b.aff_pcst[li - 1].back(), b.g_pc[li - 1], dp.out_pc, 12, Tp, tp, ctot, 0, total, 0);
            }
        }
        const int ctot = n->dims[1].prev_pc + n->dims[1].out_p;
        const int T0 = b.Tl[1];
        // ---- layer 0: its last conv's activation = channels [0, nf) of the concat buffer (two consumers, both summed into g_cat) ----
        {
            // move the slice into the dense pc0 gradient buffer, then the generic stack walk
            float* g0 = b.g_pc[0];
            AKE_HIP_CHECK(hipMemcpy2DAsync(g0, sizeof(float) * c.n_filters * 12 * T0, b.g_cat[1], sizeof(float) * ctot * 12 * T0,
                                           sizeof(float) * c.n_filters * 12 * T0, B, hipMemcpyDeviceToDevice, s));
            // the raw output of that conv lives inside cat[1] (strided): the stack walk needs it dense as well
            AKE_HIP_CHECK(hipMemcpy2DAsync(b.pcst[0].back(), sizeof(float) * c.n_filters * 12 * T0, b.cat[1], sizeof(float) * ctot * 12 * T0,
                                           sizeof(float) * c.n_filters * 12 * T0, B, hipMemcpyDeviceToDevice, s));
            std::vector<float*> aff0 = b.aff_pcst[0];
            aff0.back() = b.aff_cat[1];
            float* g02 = g0 + static_cast<size_t>(B) * c.n_filters * 12 * T0;
            if ((rc = stack_backward(n->pc2pc_t[0], n->pc2pc_d[0], "model.0.pc2pc.layer.", true, 1, Src{b.fold0, 1, nullptr, 0, 0}, nullptr, b.pcst[0],
                                     aff0, 12, T0, g0, g02, b.g_fold0, 1, "conv_wgrad_kernel/pc2pc0", "conv_mfma_kernel/pc2pc0_dgrad")))
                return rc;
        }
        // --stay_sixth: layer 0's activated semitone map is also layer 1's pitch stream (channels [0, 1) of its stack's input gradient)
        return semi_backward(0, mel, nullptr, b.g_fold0, 1, 0, nullptr, c.stay_sixth ? b.g_pin[1] : nullptr, n->dims[1].prev_p + n->dims[1].prev_pc);
    }

    // pool_semi(layer) + BatchNorm + LeakyReLU + octave fold, backward.  gfold: channel slice [g_coff, g_coff + C) of a [B][g_ctot][12][T]
    // gradient.  x / x_aff: the pitch tensor the semitone conv read.  ga_x (nullable): receives dL/d(act(x)).
    int semi_backward(int layer, const float* x, const float* x_aff, const float* gfold, int g_ctot, int g_coff, float* ga_x,
                      const float* extra_g = nullptr, int extra_ctot = 0) {
        const PackedConv& pc = n->semi_t[layer];
        const int C = pc.cin, P = n->cfg.pitches, Tn = b.Tl[layer];
        const std::string m = "model." + std::to_string(layer) + ".";
        float* g = b.g_semi[layer];
        if (n->cfg.p2pc_conv) {   // models.py:108-133: the fold is a convolution over the octaves + pool.bn + LeakyReLU
            const long long per_clip = static_cast<long long>(C) * 12 * Tn, total = per_clip * B;
            float* gf = b.g_foldc[layer];
            {   // the slice of the concat gradient that belongs to the folded channels, dense
                ake::ProfScope ps("slice_channels_kernel", s);
                hipLaunchKernelGGL(slice_channels_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, s,
                                   gfold + static_cast<long long>(g_coff) * 12 * Tn, static_cast<long long>(g_ctot) * 12 * Tn, gf, per_clip, total);
            }
            bn_block_backward(m + "pool.bn", gf, b.foldc_raw[layer], b.aff_foldc[layer], C, 0, 12 * Tn);
            {
                ake::ProfScope ps("fold_conv_bwd_weight_kernel", s);
                hipLaunchKernelGGL(fold_conv_bwd_weight_kernel, dim3(C * C * (P / 36), B), dim3(64), 0, s, gf, b.semi_raw[layer], b.aff_semi[layer],
                                   grad_of(m + "pool.conv.weight"), static_cast<long long>(n->grad_floats), C, P / 36, Tn);
            }
            {
                const long long tot = static_cast<long long>(B) * C * (P / 3) * Tn;
                ake::ProfScope ps("fold_conv_bwd_data_kernel", s);
                hipLaunchKernelGGL(fold_conv_bwd_data_kernel, dim3(static_cast<unsigned>((tot + 255) / 256)), dim3(256), 0, s, gf,
                                   raw_of(m + "pool.conv.weight"), g, C, P / 36, Tn, tot);
            }
        } else {
            const long long total = static_cast<long long>(B) * C * 12 * Tn;
            ake::ProfScope ps("fold_bwd_kernel", s);
            hipLaunchKernelGGL(fold_bwd_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, s, gfold, b.semi_raw[layer],
                               b.aff_semi[layer], g, C, P / 36, Tn, g_ctot, g_coff, total);
        }
        if (extra_g) {   // a second consumer of the activated semitone map: add its gradient (channels [0, C) of a wider tensor)
            const long long HT = static_cast<long long>(P / 3) * Tn, total = static_cast<long long>(B) * C * HT;
            ake::ProfScope ps("add_slice_kernel", s);
            hipLaunchKernelGGL(add_slice_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, s, g, extra_g, C, HT, extra_ctot, total);
        }
        bn_block_backward(m + "pool_semi_b", g, b.semi_raw[layer], b.aff_semi[layer], C, 0, (P / 3) * Tn);
        {
            ake::ProfScope ps("semi_bwd_weight_kernel", s);
            const size_t lds = std::max<size_t>(static_cast<size_t>(4) * (C * Tn + 3 * C * (Tn + 2)), 4 * 9 * 64) * sizeof(float);
            const size_t cap = 160 * 1024;          // one workgroup per CU beyond 64 KB (--local clips: no time pooling, more frames)
            AKE_REQUIRE(lds <= cap, AKE_ERR_UNSUPPORTED, "backward: pool_semi weight gradient stages %zu B of LDS per workgroup (got %zu: too many channels x frames)", cap, lds);
            static ake::DeviceOnce semi_attr;
            if (lds > kLdsBudget && semi_attr.need()) {
                AKE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(semi_bwd_weight_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(cap)));
                semi_attr.mark();
            }
            // rows per workgroup: kSemiRows at training batch sizes (few atomics), down to one row per wave when the batch is small
            const int S = P / 3, pair_groups = (C * C + 63) / 64;
            const int want = (std::max(tiling_cus(2), 1) + B * pair_groups - 1) / (B * pair_groups);     // workgroups per clip that fill the chip
            const int rows = std::min(kSemiRows, std::max(4, (S + want - 1) / want));
            hipLaunchKernelGGL(semi_bwd_weight_kernel, dim3((S + rows - 1) / rows, B, pair_groups), dim3(256), lds, s, g, x, x_aff,
                               grad_of(m + "pool_semi.weight"), static_cast<long long>(n->grad_floats), C, P, Tn, rows);
        }
        if (ga_x) {
            const long long total = static_cast<long long>(B) * P * Tn;
            ake::ProfScope ps("semi_bwd_data_kernel", s);
            hipLaunchKernelGGL(semi_bwd_data_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), sizeof(float) * C * C * 9, s, g,
                               raw_of(m + "pool_semi.weight"), ga_x, C, P, Tn, total);
        }
        return AKE_OK;
    }
};

}  // namespace

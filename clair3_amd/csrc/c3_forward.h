// c3_forward.h -- the launch sequences of the two forward passes (clair3/model.py:130-161 Clair3_P, :377-416 Clair3_F) on a
// stream, device pointers in and out.  Per layer: the product (fp16x3 on the 16-bit matrix instructions) while m->f16_ok, else
// its one fp32-MFMA form.
#pragma once
#include "c3_model.h"

// ------------------------------------------------------------------------------------------ the FC chain on its own stream (ring only)
// tail_begin: the stream the chain of THIS forward pass runs on -- tail_stream, behind everything queued on s so far, when the ring's
// submit asked for it (m->tail_now), else s itself.  tail_end: marks the chain's end.  tail_guard: called on s right before the first
// launch that overwrites what a chain reads (the pooled tensor / lstm2_out of the handle's one workspace): waits for the chain that may
// still be running -- by then it finished a whole network ago, so the wait costs nothing.
static int tail_guard(c3_model *m, hipStream_t s);
static int tail_begin(c3_model *m, hipStream_t s, hipStream_t *ts) {
    *ts = s;
    if (!m->tail_now) return tail_guard(m, s);  // (a chain on s itself: behind one that may still be running on tail_stream -- they share the partials)
    if (!m->tail_stream) {
        HIP_TRY(new_stream(m, &m->tail_stream));
        HIP_TRY(hipEventCreateWithFlags(&m->ev_body_done, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&m->ev_tail_done, hipEventDisableTiming));
    }
    HIP_TRY(hipEventRecord(m->ev_body_done, s));
    HIP_TRY(hipStreamWaitEvent(m->tail_stream, m->ev_body_done, 0));
    *ts = m->tail_stream;
    return 0;
}
static int tail_end(c3_model *m, hipStream_t ts) {
    if (ts != m->tail_stream || !m->tail_stream) return 0;
    HIP_TRY(hipEventRecord(m->ev_tail_done, ts));
    m->tail_pending = true;
    return 0;
}
static int tail_guard(c3_model *m, hipStream_t s) {
    if (!m->tail_pending) return 0;
    HIP_TRY(hipStreamWaitEvent(s, m->ev_tail_done, 0));
    m->tail_pending = false;
    return 0;
}

// ------------------------------------------------------------------------------------------ FC tail (both networks)
// L4 as a split-K contraction -> splitk_reduce_selu_kernel -> fc_tail_mfma_kernel (c3_tail.h); the decoder columns behind it
static int run_tail(c3_model *m, hipStream_t s0, const float *a, int64_t lda, int64_t n, float *y, const char *tag_l4,
                    const char *tag_tail) {
    hipStream_t s;
    TRY(tail_begin(m, s0, &s));  // (the ring: the chain on its own stream, behind the layers queued on s0)
    const int FC = m->FC, K4 = m->K4;
    const int nk_total = K4 / kBK;
    const int S = l4_splits(m);
    const bool l4_f16 = m->f16_ok && m->l4_wf;
    {
        ProfScope ps(m, s, tag_l4, 2.0 * n * FC * K4, 4.0 * (n * K4 + (double)FC * K4 + (double)S * n * FC));
        if (l4_f16) {  // partials carry the features' powers of two (l4_pre)
            L4Params lp{a, lda, m->l4_wf, m->part, (int)n, FC, K4 / 64, S, (int)((n + kL4BM - 1) / kL4BM), FC / kL4BN};
            ps.mfma(2.0 * lp.m_tiles * kL4BM * FC * K4 * 3, true);
            hipLaunchKernelGGL(l4_stream_kernel<0>, dim3((unsigned)(lp.m_tiles * lp.n_tiles * S)), dim3(kL4Threads), 0, s, lp);
            HIP_TRY(hipGetLastError());
        } else {
            DenseLoaderParams lp{a, lda};
            EpilogueParams ep{m->part, nullptr, nullptr, FC, n * FC};
            ps.mfma(2.0 * ((n + 127) / 128 * 128) * FC * K4, false);
            TRY((launch_gemm<DenseLoader<4>, EPI_PARTIAL, 128, 64>(s, lp, m->l4_w, K4, (int)n, FC, nk_total / S, S, ep)));
        }
    }
    {
        const double fl = 2.0 * n * (FC * 128.0 * m->nb + 128.0 * m->nout);
        ProfScope ps(m, s, tag_tail, fl, 4.0 * ((double)S * n * FC + n * m->nout));
        ps.mfma(2.0 * ((n + 15) / 16 * 16) * m->nb * (FC * 128.0 + 128.0 * 48.0), false);
        Tail2Params tp{m->l4dbg, m->w5f, m->b5, m->whf, m->bh48, y, (int)n, m->nb, m->row};
        if (m->tail_fused) {  // the split-K sum inside the tail kernel: two launches behind the last convolution / recurrence, not three
            tp.part = m->part, tp.S = S, tp.bias4 = m->l4_b, tp.l4out = m->l4dbg;
            if (l4_f16) tp.pre = m->l4_pre, tp.post = m->l4_post;
        } else {
            ReduceParams rp{m->part, m->l4_b, m->l4dbg, (int)n, FC, S};
            if (l4_f16) rp.pre = m->l4_pre, rp.post = m->l4_post;
            hipLaunchKernelGGL(splitk_reduce_selu_kernel, dim3((unsigned)((n * FC + 255) / 256)), dim3(256), 0, s, rp);
            HIP_TRY(hipGetLastError());
        }
        const dim3 grid((unsigned)((n + 15) / 16), m->nb);
        if (FC == 256)
            hipLaunchKernelGGL(fc_tail_mfma_kernel<256>, grid, dim3(256), 0, s, tp);
        else
            hipLaunchKernelGGL(fc_tail_mfma_kernel<128>, grid, dim3(256), 0, s, tp);
        HIP_TRY(hipGetLastError());
    }
    if (m->row > m->nout) {  // decoder columns behind the probabilities of every row (c3_decode.h)
        ProfScope ps(m, s, m->kind == C3_KIND_PILEUP ? "p.decode" : "fa.decode", 0.0, 4.0 * n * m->row);
        DecodeParams dp{y, m->row, nullptr, nullptr, nullptr, nullptr, y + m->nout, (int)n, m->nout == 90 ? 1 : 0};
        hipLaunchKernelGGL(outcome_maxima_kernel<true>, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, dp);
        HIP_TRY(hipGetLastError());
    }
    return tail_end(m, s);
}

// ------------------------------------------------------------------------------------------ full alignment
// PyramidPolling geometry, clair3/model.py:250-279: the bins of the three levels clipped to the image
static int spp_bins(const c3_model *m, int H, int W, SppParams &sp) {
    sp.H = H, sp.W = W, sp.C = 256;
    int nbins = 0;
    const int pools[3] = {3, 2, 1};
    for (int pi = 0; pi < 3; ++pi) {
        const int p = pools[pi];
        const int wh_ = (H + p - 1) / p, ww_ = (W + p - 1) / p;
        const int oh_n = (H + wh_ - 1) / wh_, ow_n = (W + ww_ - 1) / ww_;
        const int pad_h = std::max((oh_n - 1) * wh_ + wh_ - H, 0), pad_w = std::max((ow_n - 1) * ww_ + ww_ - W, 0);
        const int pt = pad_h / 2, pl = pad_w / 2;
        for (int oh = 0; oh < oh_n; ++oh)
            for (int ow = 0; ow < ow_n; ++ow) {
                if (nbins >= 16) return fail("unsupported geometry: more than 16 pyramid bins");
                const int a0 = oh * wh_ - pt, a1 = a0 + wh_, c0 = ow * ww_ - pl, c1 = c0 + ww_;
                sp.h0[nbins] = (short)std::max(a0, 0), sp.h1[nbins] = (short)std::min(a1, H);
                sp.w0[nbins] = (short)std::max(c0, 0), sp.w1[nbins] = (short)std::min(c1, W);
                sp.pad[nbins] = (a0 < 0 || a1 > H || c0 < 0 || c1 > W) ? 1 : 0;
                ++nbins;
            }
    }
    if (nbins * 256 != m->K4) return fail("unsupported geometry: %d pyramid bins (L4 expects %d inputs)", nbins, m->K4);
    sp.nbins = nbins;
    return 0;
}

// magic of fast_div (c3_gemm.h) for divisor d and dividends below n: 0 = "d is 1"; fails when n * d does not fit 32 bits
static int div_magic(int d, int64_t n, uint32_t *magic) {
    if (d <= 1) return *magic = 0u, 0;
    if (n * d >= ((int64_t)1 << 32)) return fail("batch too large for the 32-bit pixel arithmetic of the convolution kernels");
    return *magic = (uint32_t)((((uint64_t)1 << 32) / (uint64_t)d) + 1), 0;
}

// the plane pipeline needs every layer packed for it and images no wider than the halo tile is sized for
static bool fa_planes_ok(const c3_model *m) {
    if (!m->f16_ok) return false;
    int hh[10], ww[10];
    fa_geometry(m, hh, ww);
    for (int l = 1; l < 9; ++l)
        if (!m->pconv_w[l] || (kConvStride[l] == 1 && ww[l] > kPlMaxW)) return false;
    return m->C == 8 ? m->conv1_wfrag16 != nullptr : m->conv1_w16 != nullptr;
}

// activations as fp16 piece planes (c3_conv3.h), 8 convolution launches: conv1 inside res1a / res1b, the stride-2 convolutions
// on the chunk stream of c3_dense.h, the pyramid pooling as the epilogue of res3b
static int run_fa_planes(c3_model *m, hipStream_t s, const int8_t *x, int64_t n, float *y) {
    int hh[10], ww[10];
    fa_geometry(m, hh, ww);
    int cin = m->C;
    const bool sppf_ok = m->spp_fused && !m->keep && hh[9] == 12 && ww[9] == 5 && 14 * 256 == m->K4;
    for (int l = 0; l < 9; ++l) {
        const int Cout = kConvCout[l];
        const int M = (int)(n * hh[l + 1] * ww[l + 1]);
        // conv1 inside the first residual block (c3_conv3.h SRC8; 8-channel windows, or 9 with the dwell channel)
        const bool fuse1 = m->conv1_fused && (m->C == 8 || m->C == 9) && m->conv1_wfrag16 && !m->keep && ww[1] <= kPlMaxW && ww[0] >= 3;
        if (l == 0 && fuse1) {  // no launch, no conv1 planes: res1a computes its input rows, res1b its residual, from the windows
            cin = Cout;
            continue;
        }
        if (l == 8) TRY(tail_guard(m, s));  // res3b (and the pooling behind it) overwrites what a chain still on tail_stream reads
        double flops = 2.0 * M * Cout * 9.0 * cin;
        double bytes = (l == 0 ? 1.0 : 4.0) * n * hh[l] * ww[l] * cin + 4.0 * M * Cout * (l % 3 == 2 ? 2 : 1) + 4.0 * Cout * 9.0 * cin;
        if (fuse1 && l == 1) flops += 2.0 * M * 64.0 * 9.0 * m->C, bytes += 1.0 * n * hh[0] * ww[0] * m->C - 4.0 * M * 64;  // conv1's algorithmic work rides here
        if (fuse1 && l == 2) bytes += 1.0 * n * hh[0] * ww[0] * m->C - 4.0 * M * 64;  // residual from the windows, not from conv1 planes
        ProfScope ps(m, s, kFaLayerTag[l], flops, bytes);
        if (l == 0 && cin == 8) {
            ps.mfma(2.0 * ((M + 31) / 32 * 32) * 64.0 * 80.0 * 2, true);
            Conv1F16Params cp;
            cp.x = x, cp.wfrag = reinterpret_cast<const uint32_t *>(m->conv1_wfrag16), cp.bias = m->conv_b[0], cp.out = m->act[0];
            cp.range_flag = m->range_flag, cp.post = m->conv1_post;
            cp.B = (int)n, cp.H = hh[0], cp.W = ww[0], cp.OH = hh[1], cp.OW = ww[1], cp.M = M, cp.groups = (M + 31) / 32;
            const int grid = std::min((cp.groups + 3) / 4, m->wg_slots);
            hipLaunchKernelGGL(conv1_i8_f16_kernel<true>, dim3(grid), dim3(256), 0, s, cp);
            HIP_TRY(hipGetLastError());
        } else if (l == 0) {
            ps.mfma(2.0 * ((M + 127) / 128 * 128) * 64.0 * 96.0 * 3, true);
            Conv1LoaderParams lp{x, (const int8_t *)m->zeros, hh[0], ww[0], cin, hh[1], ww[1]};
            EpilogueParams ep{m->act[0], m->conv_b[0], nullptr, Cout, 0};
            ep.post = m->conv1_w16_post, ep.range_flag = m->range_flag;
            TRY((launch_gemm<Conv1Loader<4>, EPI_BIAS_RELU_PLANES, 128, 64, 2>(s, lp, m->conv_w[0], 96, M, Cout, 3, 1, ep, m->conv1_w16)));
        } else if (kConvStride[l] == 2) {
            S2ConvParams sp;
            sp.a = m->act[l - 1], sp.wf = m->pconv_w[l], sp.bias = m->conv_b[l], sp.post = m->pconv_post[l], sp.c = m->act[l], sp.range_flag = m->range_flag;
            sp.M = M, sp.N = Cout, sp.NK = 9 * cin / 64, sp.tiles_n = Cout / kS2BN, sp.tiles = (M + kS2BM - 1) / kS2BM * sp.tiles_n;
            sp.Hin = hh[l], sp.Win = ww[l], sp.Cin = cin, sp.Ho = hh[l + 1], sp.Wo = ww[l + 1];
            TRY(div_magic(hh[l + 1] * ww[l + 1], (int64_t)M + 2 * kS2BM, &sp.mg_hw));
            TRY(div_magic(ww[l + 1], hh[l + 1] * ww[l + 1], &sp.mg_w));
            ps.mfma(2.0 * ((M + kS2BM - 1) / kS2BM * kS2BM) * (double)Cout * 9.0 * cin * 3, true);
            // more tiles than CUs: two 512-thread workgroups (66 KB of LDS, 128 registers a lane) per CU; otherwise one, with twice the registers
            const bool pair = sp.tiles > m->wg_slots / 2;
            m->choice_s2[l == 3 ? 0 : 1] = pair ? "two-workgroups-per-cu" : "one-workgroup-per-cu";
            int g = sp.tiles;
            const int slots = pair ? m->wg_slots : m->wg_slots / 2, unit = 8 * sp.tiles_n;
            if (g > slots) g = std::max(unit, slots / unit * unit);
            if (pair) hipLaunchKernelGGL((conv3x3_s2_planes_kernel<0, true>), dim3(g), dim3(kS2Threads), 0, s, sp);
            else
                hipLaunchKernelGGL((conv3x3_s2_planes_kernel<0, false>), dim3(g), dim3(kS2Threads), 0, s, sp);
            HIP_TRY(hipGetLastError());
        } else {
            const bool res = l % 3 == 2;
            PlaneConvParams cp;
            cp.x = m->act[l - 1], cp.wf = m->pconv_w[l], cp.bias = m->conv_b[l], cp.res = res ? m->act[l - 2] : nullptr, cp.out = m->act[l];
            cp.range_flag = m->range_flag, cp.post = m->pconv_post[l], cp.pre = m->pconv_pre[l];
            cp.M = M, cp.H = hh[l], cp.W = ww[l];
            TRY(div_magic(hh[l] * ww[l], (int64_t)M + 2 * kPlBM, &cp.mg_hw));
            TRY(div_magic(ww[l], hh[l] * ww[l], &cp.mg_w));
            const int tiles_m = (M + kPlBM - 1) / kPlBM;
            cp.tiles = tiles_m * (Cout / 64);
            const bool src8 = fuse1 && (l == 1 || l == 2);
            // PyramidPolling as the epilogue of the last convolution (c3_conv3.h SPPF): 12 x 5 windows, two whole windows per tile
            const bool sppf = l == 8 && sppf_ok;
            constexpr int wpt = kPlBM / 60;  // windows per tile
            if (sppf) {
                cp.spp = m->spp;
                cp.tiles = (int)((n + wpt - 1) / wpt) * (Cout / 64);
            }
            if (src8) {
                cp.x8 = x, cp.c1w = reinterpret_cast<const uint32_t *>(m->conv1_wfrag16), cp.c1b = m->conv_b[0], cp.c1post = m->conv1_post, cp.Hin = hh[0], cp.Win = ww[0];
                if (l == 1) cp.x = nullptr;
                else cp.res = nullptr;
            }
            // SRC8: + conv1 for the halo rows of a tile in groups of 32 (res1a) / the tile's own pixels (res1b), two piece products of
            // K = 80 (96 for 9 channels)
            const double tiles_x = sppf ? (double)((n + wpt - 1) / wpt) : (double)tiles_m;  // pixel tiles the launch really runs
            const int c1_rows = l == 1 ? (kPlBM + 2 * ww[l] + 2 + 31) / 32 * 32 : kPlBM;
            ps.mfma(2.0 * tiles_x * kPlBM * (double)Cout * 9.0 * cin * 3 + (src8 ? 2.0 * tiles_m * c1_rows * 64.0 * (m->C == 8 ? 80.0 : 96.0) * 2 : 0.0), true);
            m->choice_s1[(l / 3) * 2 + (l % 3 - 1)] = 'd';
            // F(2,3) along H (c3_conv3w.h) for the plain plane-to-plane layers: 12 instead of 18 groups of piece products per output
            // pair.  Not where conv1 is computed inside the kernel (res1a / res1b) or the pyramid pooling is its epilogue (res3b):
            // those stay on the direct kernel.  C3HIP_WINO: 0 none, 1 the 64- / 128-channel ones, 2 (default) every plain one -- same-box
            // A/B of the whole step (profiles/r05_e_ab_wino_step*.txt): B = 256 707 k -> 729 k windows/s (res2a / res2b 48.3 / 50.3 -> 41.7 /
            // 43.8 us; res3a unchanged: 244 tiles = one workgroup per CU), B = 1000 774 k -> 804 k (res3a 172 -> 148.5 us as well).
            const bool wino_layer = m->wino >= 2 || (m->wino == 1 && (Cout == 64 || Cout == 128));
            if (wino_layer && !src8 && !sppf && m->wconv_w[l]) {
                WinoConvParams wp;
                wp.x = m->act[l - 1], wp.wf = m->wconv_w[l], wp.bias = m->conv_b[l], wp.post = m->wconv_post[l];
                wp.res = res ? m->act[l - 2] : nullptr, wp.out = m->act[l], wp.range_flag = m->range_flag;
                wp.M = M, wp.H = hh[l], wp.W = ww[l], wp.Hj = (hh[l] + 1) / 2, wp.Mp = (int)(n * wp.Hj * ww[l]);
                TRY(div_magic(wp.Hj * ww[l], (int64_t)wp.Mp + 2 * kWTM, &wp.mg_hjw));
                TRY(div_magic(ww[l], wp.Hj * ww[l], &wp.mg_w));
                const int tiles_w = (wp.Mp + kWTM - 1) / kWTM;
                wp.tiles = tiles_w * (Cout / 64);
                ps.mfma(2.0 * tiles_w * kWRows * (double)Cout * 12.0 * cin * 3, true);
                int gw = wp.tiles;
                const int wslots = m->wg_slots, wunit = 8 * (Cout / 64);
                if (gw > wslots) gw = std::max(wunit, wslots / wunit * wunit);
                const dim3 wgrid(gw), wblock(kPlThreads);
                if (Cout == 64) {
                    if (res) hipLaunchKernelGGL((conv3x3_wino_planes_kernel<64, true>), wgrid, wblock, 0, s, wp);
                    else hipLaunchKernelGGL((conv3x3_wino_planes_kernel<64, false>), wgrid, wblock, 0, s, wp);
                } else if (Cout == 128) {
                    if (res) hipLaunchKernelGGL((conv3x3_wino_planes_kernel<128, true>), wgrid, wblock, 0, s, wp);
                    else hipLaunchKernelGGL((conv3x3_wino_planes_kernel<128, false>), wgrid, wblock, 0, s, wp);
                } else {
                    if (res) hipLaunchKernelGGL((conv3x3_wino_planes_kernel<256, true>), wgrid, wblock, 0, s, wp);
                    else hipLaunchKernelGGL((conv3x3_wino_planes_kernel<256, false>), wgrid, wblock, 0, s, wp);
                }
                HIP_TRY(hipGetLastError());
                m->choice_s1[(l / 3) * 2 + (l % 3 - 1)] = 'w';
                cin = Cout;
                continue;
            }
            // persistent: one workgroup per tile when they all fit (two 256-thread workgroups, <= 70 KB of LDS each, per CU), else as
            // many as fit, rounded down so that a workgroup's tiles share their column tile (c3_conv3.h)
            int g = cp.tiles;
            const int slots = m->wg_slots, unit = 8 * (Cout / 64);
            if (g > slots) g = std::max(unit, slots / unit * unit);
            const dim3 grid(g), block(kPlThreads);
            if (Cout == 64 && src8 && m->C == 9) {
                if (res) hipLaunchKernelGGL((conv3x3_planes_kernel<64, true, 0, 2, false, 9>), grid, block, 0, s, cp);
                else hipLaunchKernelGGL((conv3x3_planes_kernel<64, false, 0, 1, false, 9>), grid, block, 0, s, cp);
            } else if (Cout == 64 && src8) {
                if (res) hipLaunchKernelGGL((conv3x3_planes_kernel<64, true, 0, 2>), grid, block, 0, s, cp);
                else hipLaunchKernelGGL((conv3x3_planes_kernel<64, false, 0, 1>), grid, block, 0, s, cp);
            } else if (Cout == 64) {
                if (res) hipLaunchKernelGGL((conv3x3_planes_kernel<64, true>), grid, block, 0, s, cp);
                else hipLaunchKernelGGL((conv3x3_planes_kernel<64, false>), grid, block, 0, s, cp);
            } else if (Cout == 128) {
                if (res) hipLaunchKernelGGL((conv3x3_planes_kernel<128, true>), grid, block, 0, s, cp);
                else hipLaunchKernelGGL((conv3x3_planes_kernel<128, false>), grid, block, 0, s, cp);
            } else {
                if (sppf) hipLaunchKernelGGL((conv3x3_planes_kernel<256, true, 0, 0, true>), grid, block, 0, s, cp);
                else if (res) hipLaunchKernelGGL((conv3x3_planes_kernel<256, true>), grid, block, 0, s, cp);
                else hipLaunchKernelGGL((conv3x3_planes_kernel<256, false>), grid, block, 0, s, cp);
            }
            HIP_TRY(hipGetLastError());
        }
        cin = Cout;
    }
    if (!sppf_ok) {
        ProfScope ps(m, s, "fa.spp", 0.0, 4.0 * n * (hh[9] * ww[9] * 256.0 + m->K4));
        if (hh[9] == 12 && ww[9] == 5) {
            if (14 * 256 != m->K4) return fail("unsupported geometry: L4 expects %d inputs", m->K4);
            const int grid = (int)std::min<int64_t>(n, 8192);
            hipLaunchKernelGGL((spp_planes_fixed_kernel<12, 5>), dim3(grid), dim3(256), 0, s, (const void *)m->act[8], m->spp, (int)n, 256);
        } else {
            SppParams sp;
            TRY(spp_bins(m, hh[9], ww[9], sp));
            sp.in = m->act[8], sp.out = m->spp, sp.B = (int)n;
            const int64_t total = n * m->K4;
            const int grid = (int)std::min<int64_t>((total + 255) / 256, 8192);
            hipLaunchKernelGGL(spp_planes_kernel, dim3(grid), dim3(256), 0, s, sp);
        }
        HIP_TRY(hipGetLastError());
    }
    m->last_planes = true;
    return run_tail(m, s, m->spp, m->K4, n, y, "fa.l4", "fa.tail");
}

// The fp32 form of the full-alignment network (range-guard fallback, C3HIP_FP32=1, geometries the plane kernels are not
// sized for): fp32 NHWC activations, every convolution an implicit GEMM on v_mfma_f32_32x32x2_f32 (c3_gemm.h ConvLoader /
// Conv1Loader), pooling and tail on fp32.
static int run_fa_fp32(c3_model *m, hipStream_t s, const int8_t *x, int64_t n, float *y) {
    m->last_planes = false;
    int hh[10], ww[10];
    fa_geometry(m, hh, ww);
    int cin = m->C;
    for (int l = 0; l < 9; ++l) {
        const int Cout = kConvCout[l];
        const int M = (int)(n * hh[l + 1] * ww[l + 1]);
        if (l == 8) TRY(tail_guard(m, s));
        const double flops = 2.0 * M * Cout * 9.0 * cin;
        const double bytes = (l == 0 ? 1.0 : 4.0) * n * hh[l] * ww[l] * cin + 4.0 * M * Cout * (l % 3 == 2 ? 2 : 1) + 4.0 * Cout * 9.0 * cin;
        ProfScope ps(m, s, kFaLayerTag[l], flops, bytes);
        const bool res = l % 3 == 2;
        EpilogueParams ep{m->act[l], m->conv_b[l], res ? m->act[l - 2] : nullptr, Cout, 0};
        if (l == 0) {
            Conv1LoaderParams lp{x, (const int8_t *)m->zeros, hh[0], ww[0], cin, hh[1], ww[1]};
            ps.mfma(2.0 * ((M + 127) / 128 * 128) * 64.0 * 96.0, false);
            TRY((launch_gemm<Conv1Loader<4>, EPI_BIAS_RELU, 128, 64>(s, lp, m->conv_w[0], 96, M, Cout, 3, 1, ep)));
        } else {
            ConvLoaderParams lp{m->act[l - 1], m->zeros, hh[l], ww[l], cin, hh[l + 1], ww[l + 1], kConvStride[l], cin / kBK};
            const int nk = 9 * cin / kBK;
            const int64_t ldb = 9 * cin;
            ps.mfma(2.0 * ((M + 127) / 128 * 128) * (double)Cout * 9.0 * cin, false);
            if (res)
                TRY((launch_gemm<ConvLoader<4>, EPI_BIAS_RES_RELU, 128, 64>(s, lp, m->conv_w[l], ldb, M, Cout, nk, 1, ep)));
            else
                TRY((launch_gemm<ConvLoader<4>, EPI_BIAS_RELU, 128, 64>(s, lp, m->conv_w[l], ldb, M, Cout, nk, 1, ep)));
        }
        cin = Cout;
    }
    {
        SppParams sp;
        TRY(spp_bins(m, hh[9], ww[9], sp));
        sp.in = m->act[8], sp.out = m->spp, sp.B = (int)n;
        ProfScope ps(m, s, "fa.spp", 0.0, 4.0 * n * (hh[9] * ww[9] * 256.0 + m->K4));
        const int64_t total = n * m->K4;
        const int grid = (int)std::min<int64_t>((total + 255) / 256, 8192);
        hipLaunchKernelGGL(spp_kernel, dim3(grid), dim3(256), 0, s, sp);
        HIP_TRY(hipGetLastError());
    }
    return run_tail(m, s, m->spp, m->K4, n, y, "fa.l4", "fa.tail");
}

static int run_fa(c3_model *m, hipStream_t s, const int8_t *x, int64_t n, float *y) {
    if (fa_planes_ok(m)) {
        m->choice_fa = "planes-f16x3";
        return run_fa_planes(m, s, x, n, y);
    }
    m->choice_fa = "fp32-mfma", m->choice_s2[0] = m->choice_s2[1] = "-";
    return run_fa_fp32(m, s, x, n, y);
}

// ------------------------------------------------------------------------------------------ pileup
// LSTM1 (input projection fused into the recurrence, h1 out as planes) -> LSTM2 projection (weights resident in registers) ->
// LSTM2 recurrence -> tail.  With `starts` the windows are gathered out of one region matrix (c3_predict_pileup_region).
template <typename T>
static int run_pileup_t(c3_model *m, hipStream_t s, const T *x, int64_t n, float *y, const int32_t *starts = nullptr) {
    const int Tn = m->positions;
    const int M = (int)(n * Tn);
    // int8 windows feed the fp16 projection fragments (counts are exact in fp16); int32 windows keep an fp32 projection inside
    // the fp16x3 recurrence kernel
    const bool l1_f16 = m->f16_ok && m->whh16[0] && (sizeof(T) != 1 || m->l1_wih16);
    const int beside = std::max(m->sharing, m->lane_sharing);  // other batches on the chip: the caller's handles, or this handle's other lanes (c3_model.h)
    const bool h1_planes = l1_f16 && m->proj2_pw;  // h1 leaves LSTM1 as fp16 piece planes for c3_dense.h
    m->last_planes = h1_planes;
    {
        ProfScope ps(m, s, "p.lstm1", 2.0 * M * 1024.0 * m->C + 2.0 * M * 2.0 * 512.0 * 128.0, sizeof(T) * (double)M * m->C + 4.0 * M * 256.0);
        // half tiles (8 windows per workgroup, c3_lstm_fused.h OPT bit 2) while the full tiles would leave half the CUs without a
        // workgroup (<= 1024 windows on 256 CUs; beyond that two half tiles share a CU and take twice as long: 1100 windows 94 us
        // against 60 us on full tiles)
        const bool half1 = l1_f16 && m->half_tiles && beside <= 1 && h1_planes && sizeof(T) == 1 && 2 * ((n + 15) / 16) <= m->wg_slots / 4;
        const double tiles = (double)(half1 ? (n + 7) / 8 * 16 : (n + 15) / 16 * 16) * Tn * 2;  // (window, step, direction) rows of the 16-row tiles
        // recurrent part 512 x 128 as fp16x3 (or fp32); input part: int8 windows 512 x 32 against two weight pieces, else 512 x 20 fp32
        ps.mfma(l1_f16 ? tiles * 2.0 * 512 * (128 * 3 + (sizeof(T) == 1 ? 32 * 2 : 0)) : tiles * 2.0 * 512 * (128 + 20), l1_f16);
        LstmFusedParams<T> lp{x, starts, m->l1_wih, m->l1_bias, m->whh[0], reinterpret_cast<const uint32_t *>(m->l1_wih16), m->h1, (int)n, Tn, m->C};
        const dim3 grid((unsigned)((n + 15) / 16), 2);
        if (l1_f16) {
            lp.whh = m->whh16[0];
            if (h1_planes) lp.hplanes = m->h1;
            m->choice_lstm1 = half1 ? "fused-f16x3-half-tiles" : "fused-f16x3-full-tiles";
            bool launched = false;
            if constexpr (sizeof(T) == 1) {
                if (half1) {
                    hipLaunchKernelGGL((lstm1_fused_kernel<T, true, 7>), dim3((unsigned)((n + 7) / 8), 2), dim3(512), 0, s, lp);
                    launched = true;
                }
            }
            if (!launched) {
                if (h1_planes) hipLaunchKernelGGL((lstm1_fused_kernel<T, true, 3>), grid, dim3(512), 0, s, lp);
                else hipLaunchKernelGGL((lstm1_fused_kernel<T, true>), grid, dim3(512), 0, s, lp);
            }
        } else {
            m->choice_lstm1 = "fused-fp32-mfma";
            hipLaunchKernelGGL(lstm1_fused_kernel<T>, grid, dim3(512), 0, s, lp);
        }
        HIP_TRY(hipGetLastError());
    }
    {
        ProfScope ps(m, s, "p.proj2", 2.0 * M * 1280.0 * 256.0, 4.0 * M * (256.0 + 1280.0));
        ps.mfma(2.0 * ((M + 127) / 128 * 128) * 1280.0 * 256.0 * (h1_planes ? 3 : 1), h1_planes);
        if (h1_planes && m->proj2_pwr && (M + kWrBM - 1) / kWrBM >= 2 * 8 * std::max(1, m->wg_slots / 16 / (1280 / kWrBN))) {
            // weights resident in registers (c3_dense.h): 8 XCDs x lanes x 5 column tiles of workgroups, each walking the row tiles of its lane
            DenseWresParams wp;
            wp.a = m->h1, wp.w = m->proj2_pwr, wp.bias = m->proj_b[1], wp.c = m->gx2, wp.post_scale = m->proj2_post_scale;
            wp.M = M, wp.N = 1280, wp.tiles_m = (M + kWrBM - 1) / kWrBM, wp.tiles_n = 1280 / kWrBN;
            wp.lanes_per_xcd = std::max(1, m->wg_slots / 16 / wp.tiles_n);  // CUs per XCD / column tiles (32 / 5 = 6)
            // beside other handles half as many, twice as long workgroups: 120 of them leave room for the 128 of another batch's
            // LSTM launch (three batches in flight 5.46 M -> 5.59 M windows/s; alone 4.9 M -> 4.3 M, hence the caller's hint)
            if (beside > 1) wp.lanes_per_xcd = std::max(1, wp.lanes_per_xcd / 2);
            m->choice_proj2 = beside > 1 ? "weights-resident-half-grid" : "weights-resident";
            static const bool gx2_nt = getenv("C3HIP_GX2_NT") && atoi(getenv("C3HIP_GX2_NT")) != 0;  // A/B knob (profiles/r05_*_ab_gx2_nt.txt)
            if (gx2_nt) hipLaunchKernelGGL(dense_planes_wres_kernel<64>, dim3(8 * wp.lanes_per_xcd * wp.tiles_n), dim3(kDnThreads), 0, s, wp);
            else hipLaunchKernelGGL(dense_planes_wres_kernel<0>, dim3(8 * wp.lanes_per_xcd * wp.tiles_n), dim3(kDnThreads), 0, s, wp);
            HIP_TRY(hipGetLastError());
        } else if (h1_planes) {  // batches below ~190 windows: fewer than two row tiles per lane
            DensePlanesParams dp;
            dp.a = m->h1, dp.w = m->proj2_pw, dp.bias = m->proj_b[1], dp.c = m->gx2, dp.post = m->proj2_post;
            dp.M = M, dp.N = 1280, dp.K = 256, dp.tiles_n = 1280 / kDnBN, dp.tiles = ((M + kDnBM - 1) / kDnBM) * dp.tiles_n;
            m->choice_proj2 = "128x128-chunk-stream";
            hipLaunchKernelGGL(dense_planes_pipe_kernel<0>, dim3(std::min(dp.tiles, m->wg_slots / 2)), dim3(kDnThreads), 0, s, dp);
            HIP_TRY(hipGetLastError());
        } else {
            DenseLoaderParams lp{m->h1, 256};
            EpilogueParams ep{m->gx2, m->proj_b[1], nullptr, 1280, 0};
            m->choice_proj2 = "fp32-mfma";
            TRY((launch_gemm<DenseLoader<4>, EPI_BIAS, 128, 128>(s, lp, m->proj_w[1], 256, M, 1280, 8, 1, ep)));
        }
    }
    TRY(tail_guard(m, s));  // LSTM2 overwrites lstm2_out, which a chain still on tail_stream reads
    {
        ProfScope ps(m, s, "p.lstm2", 2.0 * M * 2.0 * 640.0 * 160.0, 4.0 * M * (1280.0 + 320.0));
        const bool l2_f16 = m->f16_ok && m->whh16[1];
        const bool half2 = l2_f16 && m->half_tiles && beside <= 1 && 2 * ((n + 15) / 16) <= m->wg_slots / 4;
        ps.mfma((double)(half2 ? (n + 7) / 8 * 16 : (n + 15) / 16 * 16) * Tn * 2 * 2.0 * 640 * 160 * (l2_f16 ? 3 : 1), l2_f16);
        Lstm2Params lp{m->gx2, m->whh[1], m->h2, (int)n, Tn, 1280};
        if (l2_f16) {
            lp.whh = m->whh16[1];
            m->choice_lstm2 = half2 ? "f16x3-half-tiles" : "f16x3-full-tiles";
            if (half2) hipLaunchKernelGGL((lstm_recurrent_kernel_v2<160, true, 4>), dim3((unsigned)((n + 7) / 8), 2), dim3(512), 0, s, lp);
            else hipLaunchKernelGGL((lstm_recurrent_kernel_v2<160, true>), dim3((unsigned)((n + 15) / 16), 2), dim3(512), 0, s, lp);
        } else {
            m->choice_lstm2 = "fp32-mfma";
            hipLaunchKernelGGL(lstm_recurrent_kernel_v2<160>, dim3((unsigned)((n + 15) / 16), 2), dim3(512), 0, s, lp);
        }
        HIP_TRY(hipGetLastError());
    }
    return run_tail(m, s, m->h2, m->K4, n, y, "p.l4", "p.tail");
}

// ------------------------------------------------------------------------------------------ both
static int forward_device(c3_model *m, hipStream_t s, const void *x, int x_dtype, int64_t batch, float *y,
                          const int32_t *starts = nullptr) {
    if (!m->loaded) return fail("model has no weights: call c3_model_load first");
    if (batch < 0) return fail("negative batch");
    if (batch == 0) return 0;
    if (m->kind == C3_KIND_FULL_ALIGNMENT && x_dtype != C3_DTYPE_I8)
        return fail("full-alignment windows must be int8 (got dtype %d)", x_dtype);
    if (m->kind == C3_KIND_PILEUP && x_dtype != C3_DTYPE_I8 && x_dtype != C3_DTYPE_I32)
        return fail("pileup windows must be int8 or int32 (got dtype %d)", x_dtype);
    TRY(ensure_workspace(m, batch));
    const int64_t wbytes = c3_model_window_bytes(m, x_dtype);
    auto run = [&](hipStream_t st, const char *xp, const int32_t *sp, int64_t n, float *yp) -> int {
        if (m->kind == C3_KIND_FULL_ALIGNMENT) return run_fa(m, st, (const int8_t *)xp, n, yp);
        if (x_dtype == C3_DTYPE_I8) return run_pileup_t<int8_t>(m, st, (const int8_t *)xp, n, yp, sp);
        return run_pileup_t<int32_t>(m, st, (const int32_t *)xp, n, yp, sp);
    };
    for (int64_t off = 0; off < batch; off += m->cap) {
        const int64_t n = std::min<int64_t>(m->cap, batch - off);
        const char *xp = starts ? (const char *)x : (const char *)x + off * wbytes;  // region matrix is shared
        const int32_t *sp = starts ? starts + off : nullptr;
        float *yp = y + off * m->row;
        // DUO (C3HIP_DUO=1): the micro-batch as two halves on two streams.  Every layer is its own launch and every workgroup of a
        // launch is in the same phase, so ~10 us of head and tail per launch overlap nothing when one batch is alone on the chip
        // (DESIGN.md 3.8); windows are independent, so the second half's launches -- enqueued behind the first half's, i.e. half a
        // step out of phase -- fill them.  Each half works in its own part of the workspace (the buffers are sized for the whole
        // micro-batch); a window's row does not depend on the batch it travels in, so the rows are the undivided call's bit for bit.
        const int64_t duo_min = m->kind == C3_KIND_FULL_ALIGNMENT ? 192 : 768;
        if (m->duo > 0 && !m->keep && m->sharing <= 1 && n >= duo_min) {
            if (!m->duo_stream) {
                HIP_TRY(new_stream(m, &m->duo_stream));
                HIP_TRY(hipEventCreateWithFlags(&m->duo_fork, hipEventDisableTiming));
                HIP_TRY(hipEventCreateWithFlags(&m->duo_join, hipEventDisableTiming));
            }
            const int64_t n0 = ((n / 2 + 15) / 16) * 16, n1 = n - n0;  // whole 16-window tiles in the first half
            HIP_TRY(hipEventRecord(m->duo_fork, s));
            HIP_TRY(hipStreamWaitEvent(m->duo_stream, m->duo_fork, 0));
            TRY(run(s, xp, sp, n0, yp));
            // the second half: the same buffers, behind the first half's share of each
            float *const act0[9] = {m->act[0], m->act[1], m->act[2], m->act[3], m->act[4], m->act[5], m->act[6], m->act[7], m->act[8]};
            float *const spp0 = m->spp, *const part0 = m->part, *const dbg0 = m->l4dbg, *const h10 = m->h1, *const gx20 = m->gx2, *const h20 = m->h2;
            if (m->kind == C3_KIND_FULL_ALIGNMENT) {
                int hh[10], ww[10];
                fa_geometry(m, hh, ww);
                size_t biggest = 0;
                for (int l = 0; l < 9; ++l) biggest = std::max(biggest, (size_t)hh[l + 1] * ww[l + 1] * kConvCout[l]);
                for (int l = 0; l < 9; ++l) m->act[l] = act0[l] + biggest * (size_t)n0;
                m->spp = spp0 + (size_t)n0 * m->K4;
            } else {
                const size_t T = (size_t)m->positions;
                m->h1 = h10 + (size_t)n0 * T * 256, m->gx2 = gx20 + (size_t)n0 * T * 1280, m->h2 = h20 + (size_t)n0 * T * 320;
            }
            m->part = part0 + (size_t)l4_splits(m) * n0 * m->FC, m->l4dbg = dbg0 + (size_t)n0 * m->FC;
            const int rc = run(m->duo_stream, starts ? xp : xp + n0 * wbytes, sp ? sp + n0 : nullptr, n1, yp + n0 * m->row);
            for (int l = 0; l < 9; ++l) m->act[l] = act0[l];
            m->spp = spp0, m->part = part0, m->l4dbg = dbg0, m->h1 = h10, m->gx2 = gx20, m->h2 = h20;
            if (rc) return rc;
            HIP_TRY(hipEventRecord(m->duo_join, m->duo_stream));
            HIP_TRY(hipStreamWaitEvent(s, m->duo_join, 0));
            m->last_n = n0;
            continue;
        }
        TRY(run(s, xp, sp, n, yp));
        m->last_n = n;
    }
    return 0;
}

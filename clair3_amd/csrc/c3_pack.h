// c3_pack.h -- c3_model_load: the state_dict of the reference modules (clair3/model.py:96-125, :317-365) packed once into
// the layouts the kernels read.  Host arithmetic in double precision, rounded once to fp32; every tensor a fp16x3 kernel
// consumes exists a second time as two fp16 pieces (w * 2^s = h0 + h1, round to nearest even, subnormals kept) in that
// kernel's fragment / chunk order.
#pragma once
#include "c3_model.h"

// fp16x3: every OUTPUT ROW of a weight matrix (one output channel of a convolution with its BatchNorm folded in, one gate row
// of a projection, one L4 feature) is packed times its own power of two 2^k, chosen so that the row's largest weight lands in
// [4096, 8192): the high fp16 piece cannot overflow, and the low piece of a weight stays a NORMAL fp16 number down to 2^-16 of
// the row's maximum.  (With one factor per tensor -- rounds 1 and 2 -- a channel whose folded weights were 1e-4 of the tensor's
// maximum, as a BatchNorm with a small gamma / sigma produces, kept only ~13 bits of its weights: its low pieces were fp16
// subnormals.)  The factor is undone, exactly, by the per-channel multiplier of the bias FMA in the consuming kernel's
// epilogue (`post` = 2^-k next to the bias vector).
static void row_scales(const float *w, int rows, size_t cols, std::vector<float> &scale, std::vector<float> &post) {
    scale.assign(rows, 1.f), post.assign(rows, 1.f);
    for (int r = 0; r < rows; ++r) {
        float mx = 0.f;
        for (size_t i = 0; i < cols; ++i) mx = std::max(mx, std::fabs(w[(size_t)r * cols + i]));
        if (!(mx > 0.f) || !std::isfinite(mx)) continue;
        int e;
        (void)std::frexp(mx, &e);                         // mx = f * 2^e, f in [0.5, 1)
        const int k = std::min(100, std::max(-100, 13 - e));  // mx * 2^k in [4096, 8192)
        scale[r] = std::ldexp(1.f, k), post[r] = std::ldexp(1.f, -k);
    }
}

// A weight matrix [rows][cols] as fp16 pieces for the SPLIT form of gemm_mfma_kernel, layout [2][rows * cols] (uint16 payload
// carried in a float allocation): w * scale[row] = h0 + h1.
static int upload_split_pieces(c3_model *m, float **dst, const std::vector<float> &w, int rows, const std::vector<float> &scale) {
    const size_t n = w.size(), cols = n / rows;
    std::vector<float> pieces(n);  // 2 pieces x n x 2 bytes
    uint16_t *q = reinterpret_cast<uint16_t *>(pieces.data());
    for (size_t i = 0; i < n; ++i) {
        const float r = w[i] * scale[i / cols];  // exact
        const _Float16 h0 = (_Float16)r, h1 = (_Float16)(r - (float)h0);
        memcpy(&q[i], &h0, 2);
        memcpy(&q[n + i], &h1, 2);
    }
    return upload(m, dst, pieces);
}

// ------------------------------------------------------------------------------------------ weight packing
struct TensorView {
    const float *d;
    std::vector<int64_t> shape;
};
typedef std::map<std::string, TensorView> TensorMap;

static int want(const TensorMap &tm, const std::string &name, std::initializer_list<int64_t> shape, const float **out) {
    auto it = tm.find(name);
    if (it == tm.end()) return fail("Missing key in state_dict: \"%s\"", name.c_str());
    std::vector<int64_t> s(shape);
    if (it->second.shape != s) {
        std::string got, exp;
        for (auto v : it->second.shape) got += std::to_string(v) + ",";
        for (auto v : s) exp += std::to_string(v) + ",";
        return fail("size mismatch for %s: got (%s) expected (%s)", name.c_str(), got.c_str(), exp.c_str());
    }
    *out = it->second.d;
    return 0;
}

// l4_in_exp: the channel exponents of the tensor L4 reads (full alignment: the last stage's, column j reads channel j % 256 --
// PyramidPolling flattens (bin, channel)); nullptr for the pileup network
static int pack_tail(c3_model *m, const TensorMap &tm, const std::vector<int> *l4_in_exp = nullptr) {
    const int FC = m->FC, K4 = m->K4, nb = m->nb;
    const float *w0, *b;
    TRY(want(tm, "L4.weight", {FC, K4}, &w0));
    TRY(want(tm, "L4.bias", {FC}, &b));
    std::vector<float> w4(w0, w0 + (size_t)FC * K4);
    if (l4_in_exp)
        for (int f = 0; f < FC; ++f)
            for (int j = 0; j < K4; ++j) w4[(size_t)f * K4 + j] = std::ldexp(w4[(size_t)f * K4 + j], -(*l4_in_exp)[j % l4_in_exp->size()]);
    const float *w = w4.data();
    TRY(upload(m, &m->l4_w, w4));
    {
        std::vector<float> sc, post;
        row_scales(w, FC, (size_t)K4, sc, post);
        if (K4 % 64 == 0 && FC % kL4BN == 0) {
            // l4_stream_kernel (c3_l4.h): column tiles of 64 features, chunks of 64 inputs, 16 KB each in the FRAGMENT order of the
            // convolutions -- [feature half wn][k-step][piece hi | lo][lane] x 16 B, lane (n = lane & 31, kh = lane >> 5) holding inputs
            // 64 kc + 8 (2 ks + kh) .. + 7 of feature 64 tn + 32 wn + n, times the feature's power of two
            const int nk = K4 / 64;
            std::vector<float> pf((size_t)FC * K4);
            uint16_t *f16 = reinterpret_cast<uint16_t *>(pf.data());
            for (int tn = 0; tn < FC / 64; ++tn)
                for (int kc = 0; kc < nk; ++kc)
                    for (int wn = 0; wn < 2; ++wn)
                        for (int ks = 0; ks < 4; ++ks)
                            for (int lane = 0; lane < 64; ++lane)
                                for (int j = 0; j < 8; ++j) {
                                    const int f = tn * 64 + 32 * wn + (lane & 31), k = kc * 64 + 8 * (2 * ks + (lane >> 5)) + j;
                                    const float v = w[(size_t)f * K4 + k] * sc[f];  // exact
                                    const _Float16 h0 = (_Float16)v, h1 = (_Float16)(v - (float)h0);
                                    const size_t dst = ((((((size_t)tn * nk + kc) * 2 + wn) * 4 + ks) * 2) * 64 + lane) * 8 + j;
                                    memcpy(&f16[dst], &h0, 2);
                                    memcpy(&f16[dst + 64 * 8], &h1, 2);
                                }
            TRY(upload(m, &m->l4_wf, pf));
        }
        TRY(upload(m, &m->l4_pre, sc));
        TRY(upload(m, &m->l4_post, post));
    }
    TRY(upload(m, &m->l4_b, std::vector<float>(b, b + FC)));
    std::vector<float> w5t((size_t)FC * nb * 128), b5((size_t)nb * 128), wh((size_t)nb * 128 * 64, 0.f), bh((size_t)nb * 64, 0.f);
    for (int br = 0; br < nb; ++br) {
        const std::string l5 = "L5_" + std::to_string(br + 1);
        TRY(want(tm, l5 + ".weight", {128, FC}, &w));
        TRY(want(tm, l5 + ".bias", {128}, &b));
        for (int j = 0; j < 128; ++j) {
            b5[br * 128 + j] = b[j];
            for (int k = 0; k < FC; ++k) w5t[(size_t)k * nb * 128 + br * 128 + j] = w[(size_t)j * FC + k];
        }
        const std::string hd = kHeadName[br];
        TRY(want(tm, hd + ".weight", {kHeadN[br], 128}, &w));
        TRY(want(tm, hd + ".bias", {kHeadN[br]}, &b));
        for (int i = 0; i < kHeadN[br]; ++i) {
            bh[br * 64 + i] = b[i];
            for (int k = 0; k < 128; ++k) wh[((size_t)br * 128 + k) * 64 + i] = w[(size_t)i * 128 + k];
        }
    }
    {
        // fc_tail_mfma_kernel fragments: [br][wave][cb][q][lane][e] = L5_br[32 wave + 16 cb + (lane&15)][16 q + 4 (lane>>4) + e]
        //                                [br][cb][q][lane][e]       = head_br[16 cb + (lane&15)][16 q + 4 (lane>>4) + e]
        const int NQ = FC / 16;
        std::vector<float> w5f((size_t)nb * 128 * FC), whf((size_t)nb * 3 * 8 * 64 * 4, 0.f), bh48((size_t)nb * 48, 0.f);
        for (int br = 0; br < nb; ++br) {
            for (int wv = 0; wv < 4; ++wv)
                for (int cb = 0; cb < 2; ++cb)
                    for (int q = 0; q < NQ; ++q)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int e = 0; e < 4; ++e) {
                                const int j = 32 * wv + 16 * cb + (lane & 15), k = 16 * q + 4 * (lane >> 4) + e;
                                w5f[(((((size_t)br * 4 + wv) * 2 + cb) * NQ + q) * 64 + lane) * 4 + e] =
                                    w5t[(size_t)k * nb * 128 + br * 128 + j];
                            }
            for (int cb = 0; cb < 3; ++cb)
                for (int q = 0; q < 8; ++q)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 4; ++e) {
                            const int i = 16 * cb + (lane & 15), k = 16 * q + 4 * (lane >> 4) + e;
                            whf[((((size_t)br * 3 + cb) * 8 + q) * 64 + lane) * 4 + e] = wh[((size_t)br * 128 + k) * 64 + i];
                        }
            for (int i = 0; i < 48; ++i) bh48[br * 48 + i] = bh[br * 64 + i];
        }
        TRY(upload(m, &m->w5f, w5f));
        TRY(upload(m, &m->whf, whf));
        TRY(upload(m, &m->bh48, bh48));
    }
    TRY(upload(m, &m->b5, b5));
    return 0;
}

// LSTM layer `layer` (0/1): hidden H, input size `in`.
//   proj_w row n = dir*4H + wave*64 + gate*16 + unit  <->  PyTorch gate row gate*H + wave*16 + unit
//   whh fragments: [dir][wave][gate][q][lane][e] = W_hh[gate*H + wave*16 + (lane&15)][16q + 4*(lane>>4) + e]
// LSTM layer `layer` (0 = LSTM1: fused projection + recurrence, c3_lstm_fused.h; 1 = LSTM2: hoisted projection on c3_dense.h,
// recurrence lstm_recurrent_kernel_v2): hidden H, input size `in`, PyTorch gate row order i, f, g, o (gate*H + unit).
//   LSTM1 whh fragments: [dir][wave][gate][q][lane][e] = W_hh[gate*H + wave*16 + (lane&15)][16q + 4*(lane>>4) + e]
// What the load-time precision decision looks at (c3_model.h auto_fp32_at): the largest |w| of the recurrent matrices (and of
// LSTM2's input matrix, whose inputs are LSTM1's bounded outputs; LSTM1's input matrix multiplies raw counts of up to +-100 and is
// scaled accordingly by training: not comparable) and the largest absolute row sum of W_hh.
static int lstm_sensitivity(c3_model *m, const TensorMap &tm, const std::string &base, int layer, int H, int in) {
    for (int dir = 0; dir < 2; ++dir) {
        const std::string sfx = dir ? "_reverse" : "";
        const float *wih, *whh;
        TRY(want(tm, base + ".weight_ih_l0" + sfx, {4 * H, in}, &wih));
        TRY(want(tm, base + ".weight_hh_l0" + sfx, {4 * H, H}, &whh));
        for (int r = 0; r < 4 * H; ++r) {
            double row = 0.0;
            for (int k = 0; k < H; ++k) {
                const float a = std::fabs(whh[(size_t)r * H + k]);
                row += a;
                if (a > m->lstm_wmax) m->lstm_wmax = a;  // (NaN never compares greater: a NaN weight is the range guard's business)
            }
            if ((float)row > m->lstm_hh_norm) m->lstm_hh_norm = (float)row;
            if (layer == 1)
                for (int k = 0; k < in; ++k) m->lstm_wmax = std::max(m->lstm_wmax, std::fabs(wih[(size_t)r * in + k]));
        }
    }
    return 0;
}

static int pack_lstm(c3_model *m, const TensorMap &tm, int layer, int H, int in, int Kp) {
    const bool v2 = layer == 1;
    const std::string base = layer == 0 ? "LSTM1" : "LSTM2";
    const int NW = H / 16, NQ = H / 16;
    TRY(lstm_sensitivity(m, tm, base, layer, H, in));
    if (v2) {
        // lstm_recurrent_kernel_v2: projection rows in PyTorch order (n = dir*4H + gate*H + unit); W_hh fragments
        // [dir][block = n/16][q][lane][e] = W_hh[block*16 + (lane&15)][16q + 4*(lane>>4) + e]
        std::vector<float> pw((size_t)2 * 4 * H * Kp, 0.f), pb((size_t)2 * 4 * H), wf((size_t)2 * 4 * H * H);
        for (int dir = 0; dir < 2; ++dir) {
            const std::string sfx = dir ? "_reverse" : "";
            const float *wih, *whh, *bih, *bhh;
            TRY(want(tm, base + ".weight_ih_l0" + sfx, {4 * H, in}, &wih));
            TRY(want(tm, base + ".weight_hh_l0" + sfx, {4 * H, H}, &whh));
            TRY(want(tm, base + ".bias_ih_l0" + sfx, {4 * H}, &bih));
            TRY(want(tm, base + ".bias_hh_l0" + sfx, {4 * H}, &bhh));
            for (int r = 0; r < 4 * H; ++r) {
                const size_t n = (size_t)dir * 4 * H + r;
                pb[n] = (float)((double)bih[r] + (double)bhh[r]);
                for (int k = 0; k < in; ++k) pw[n * Kp + k] = wih[(size_t)r * in + k];
            }
            for (int blk = 0; blk < 4 * H / 16; ++blk)
                for (int q = 0; q < NQ; ++q)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 4; ++e) {
                            const int r = blk * 16 + (lane & 15);
                            const int k = 16 * q + 4 * (lane >> 4) + e;
                            wf[((((size_t)dir * (4 * H / 16) + blk) * NQ + q) * 64 + lane) * 4 + e] = whh[(size_t)r * H + k];
                        }
        }
        {
            // lstm_recurrent_kernel_v2<H, true>: slot q = 2 ks + piece of [dir][block][q][lane][8 fp16]:
            // piece of W_hh[block*16 + (lane&15)][32 ks + 8 (lane>>4) + j]   (the fp32 fragments' bytes and addressing)
            std::vector<float> wf16(wf.size());
            uint16_t *q16 = reinterpret_cast<uint16_t *>(wf16.data());
            for (int dir = 0; dir < 2; ++dir) {
                const float *whh;
                TRY(want(tm, base + ".weight_hh_l0" + (dir ? "_reverse" : ""), {4 * H, H}, &whh));
                for (int blk = 0; blk < 4 * H / 16; ++blk)
                    for (int ks = 0; ks < H / 32; ++ks)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int j = 0; j < 8; ++j) {
                                const float v = whh[(size_t)(blk * 16 + (lane & 15)) * H + 32 * ks + 8 * (lane >> 4) + j];
                                const _Float16 h0 = (_Float16)v, h1 = (_Float16)(v - (float)h0);
                                const size_t slot = (((size_t)dir * (4 * H / 16) + blk) * NQ + 2 * ks) * 64 * 8;
                                memcpy(&q16[slot + (size_t)lane * 8 + j], &h0, 2);
                                memcpy(&q16[slot + 64 * 8 + (size_t)lane * 8 + j], &h1, 2);
                            }
            }
            TRY(upload(m, &m->whh16[layer], wf16));
        }
        if (Kp == 256) {
            const int N = 2 * 4 * H;
            if (N % kDnBN == 0) {
                // dense_planes_pipe_kernel: chunk (column tile of 128, k chunk of 64) = 128 rows x 256 B; piece g < 8 = hi of
                // k 64 kc + 8 g .. + 7, g >= 8 = lo of the same k; times ONE power of two for the whole matrix (its rows are plain LSTM
                // weights -- no folded BatchNorm -- and dense_planes_wres_kernel has no register left for a per-row vector)
                std::vector<float> sc1, post1;
                row_scales(pw.data(), 1, pw.size(), sc1, post1);
                const std::vector<float> sc(N, sc1[0]), post(N, post1[0]);
                m->proj2_post_scale = post1[0];
                TRY(upload(m, &m->proj2_post, post));
                const int NKc = 256 / 64;
                std::vector<float> pk((size_t)N * 256);
                uint16_t *q16 = reinterpret_cast<uint16_t *>(pk.data());
                for (int tn = 0; tn < N / kDnBN; ++tn)
                    for (int kc = 0; kc < NKc; ++kc)
                        for (int r = 0; r < kDnBN; ++r)
                            for (int g = 0; g < 16; ++g)
                                for (int j = 0; j < 8; ++j) {
                                    const float v = pw[(size_t)(tn * kDnBN + r) * Kp + kc * 64 + 8 * (g & 7) + j] * sc[tn * kDnBN + r];  // exact
                                    const _Float16 h0 = (_Float16)v, h1 = (_Float16)(v - (float)h0);
                                    const _Float16 piece = g < 8 ? h0 : h1;
                                    memcpy(&q16[(((((size_t)tn * NKc + kc) * kDnBN + r) * 16 + g) * 8) + j], &piece, 2);
                                }
                TRY(upload(m, &m->proj2_pw, pk));
                if (N % kWrBN == 0 && Kp == kWrK) {
                    // dense_planes_wres_kernel: the weights of wave w of column tile tn in fragment order, [tn][w][k-step][piece][lane][8 fp16]:
                    // lane (n = lane & 31, kh = lane >> 5) holds k = 16 ks + 8 kh .. + 7 of row 256 tn + 32 w + n; the same power of two
                    std::vector<float> pr((size_t)N * 256);
                    uint16_t *r16 = reinterpret_cast<uint16_t *>(pr.data());
                    for (int tn = 0; tn < N / kWrBN; ++tn)
                        for (int w = 0; w < 8; ++w)
                            for (int ks = 0; ks < kWrKS; ++ks)
                                for (int lane = 0; lane < 64; ++lane)
                                    for (int j = 0; j < 8; ++j) {
                                        const float v = pw[(size_t)(tn * kWrBN + 32 * w + (lane & 31)) * Kp + 16 * ks + 8 * (lane >> 5) + j] *
                                                        sc[tn * kWrBN + 32 * w + (lane & 31)];  // exact
                                        const _Float16 h0 = (_Float16)v, h1 = (_Float16)(v - (float)h0);
                                        const size_t base = ((((size_t)tn * 8 + w) * kWrKS + ks) * 2) * 64 * 8;
                                        memcpy(&r16[base + (size_t)lane * 8 + j], &h0, 2);
                                        memcpy(&r16[base + 64 * 8 + (size_t)lane * 8 + j], &h1, 2);
                                    }
                    TRY(upload(m, &m->proj2_pwr, pr));
                }
            }
        }
        TRY(upload(m, &m->proj_w[layer], pw));
        TRY(upload(m, &m->proj_b[layer], pb));
        TRY(upload(m, &m->whh[layer], wf));
        return 0;
    }
    std::vector<float> wf((size_t)2 * 4 * H * H);
    for (int dir = 0; dir < 2; ++dir) {
        const std::string sfx = dir ? "_reverse" : "";
        const float *whh;
        TRY(want(tm, base + ".weight_hh_l0" + sfx, {4 * H, H}, &whh));
        for (int w = 0; w < NW; ++w)
            for (int g = 0; g < 4; ++g)
                for (int q = 0; q < NQ; ++q)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 4; ++e) {
                            const int r = g * H + w * 16 + (lane & 15);
                            const int k = 16 * q + 4 * (lane >> 4) + e;
                            wf[(((((size_t)dir * NW + w) * 4 + g) * NQ + q) * 64 + lane) * 4 + e] = whh[(size_t)r * H + k];
                        }
    }
    TRY(upload(m, &m->whh[layer], wf));
    {
        // lstm1_fused_kernel<TX, true>: slot q = 2 ks + piece of [dir][wave][gate][q][lane][8 fp16]:
        // piece of W_hh[gate*H + wave*16 + (lane&15)][32 ks + 8 (lane>>4) + j]
        std::vector<float> wf16(wf.size());
        uint16_t *q16 = reinterpret_cast<uint16_t *>(wf16.data());
        for (int dir = 0; dir < 2; ++dir) {
            const float *whh;
            TRY(want(tm, base + ".weight_hh_l0" + (dir ? "_reverse" : ""), {4 * H, H}, &whh));
            for (int w = 0; w < NW; ++w)
                for (int g = 0; g < 4; ++g)
                    for (int ks = 0; ks < H / 32; ++ks)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int j = 0; j < 8; ++j) {
                                const float v = whh[(size_t)(g * H + w * 16 + (lane & 15)) * H + 32 * ks + 8 * (lane >> 4) + j];
                                const _Float16 h0 = (_Float16)v, h1 = (_Float16)(v - (float)h0);
                                const size_t slot = ((((size_t)dir * NW + w) * 4 + g) * NQ + 2 * ks) * 64 * 8;
                                memcpy(&q16[slot + (size_t)lane * 8 + j], &h0, 2);
                                memcpy(&q16[slot + 64 * 8 + (size_t)lane * 8 + j], &h1, 2);
                            }
        }
        TRY(upload(m, &m->whh16[layer], wf16));
    }
    {
        // fused kernel: W_ih as 16x16x4 B fragments [dir][wave][gate][ks][lane] = W_ih[g*H + w*16 + (lane&15)][4ks + (lane>>4)]
        std::vector<float> fw((size_t)2 * NW * 4 * kFusedKS * 64, 0.f), fb((size_t)2 * NW * 4 * 16);
        for (int dir = 0; dir < 2; ++dir) {
            const std::string sfx = dir ? "_reverse" : "";
            const float *wih, *bih, *bhh;
            TRY(want(tm, base + ".weight_ih_l0" + sfx, {4 * H, in}, &wih));
            TRY(want(tm, base + ".bias_ih_l0" + sfx, {4 * H}, &bih));
            TRY(want(tm, base + ".bias_hh_l0" + sfx, {4 * H}, &bhh));
            for (int w = 0; w < NW; ++w)
                for (int g = 0; g < 4; ++g) {
                    for (int u = 0; u < 16; ++u) {
                        const int r = g * H + w * 16 + u;
                        fb[(((size_t)dir * NW + w) * 4 + g) * 16 + u] = (float)((double)bih[r] + (double)bhh[r]);
                    }
                    for (int ks = 0; ks < kFusedKS; ++ks)
                        for (int lane = 0; lane < 64; ++lane) {
                            const int r = g * H + w * 16 + (lane & 15), k = 4 * ks + (lane >> 4);
                            if (k < in) fw[((((size_t)dir * NW + w) * 4 + g) * kFusedKS + ks) * 64 + lane] = wih[(size_t)r * in + k];
                        }
                }
        }
        TRY(upload(m, &m->l1_wih, fw));
        TRY(upload(m, &m->l1_bias, fb));
        if (in % 2 == 0) {
            // [dir][wave][gate][piece][lane][8 fp16]: piece of 128 W_ih[g*H + w*16 + (lane&15)][8 (lane>>4) + j]
            std::vector<float> fw16((size_t)2 * NW * 4 * 2 * 64 * 4, 0.f);
            uint16_t *q16 = reinterpret_cast<uint16_t *>(fw16.data());
            for (int dir = 0; dir < 2; ++dir) {
                const float *wih;
                TRY(want(tm, base + ".weight_ih_l0" + (dir ? "_reverse" : ""), {4 * H, in}, &wih));
                for (int w = 0; w < NW; ++w)
                    for (int g = 0; g < 4; ++g)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int j = 0; j < 8; ++j) {
                                const int r = g * H + w * 16 + (lane & 15), k = 8 * (lane >> 4) + j;
                                const float v = k < in ? 128.f * wih[(size_t)r * in + k] : 0.f;
                                const _Float16 h0 = (_Float16)v, h1 = (_Float16)(v - (float)h0);
                                const size_t slot = ((((size_t)dir * NW + w) * 4 + g) * 2) * 64 * 8;
                                memcpy(&q16[slot + (size_t)lane * 8 + j], &h0, 2);
                                memcpy(&q16[slot + 64 * 8 + (size_t)lane * 8 + j], &h1, 2);
                            }
            }
            TRY(upload(m, &m->l1_wih16, fw16));
        }
    }
    return 0;
}

// conv layer l: fold BatchNorm2d(eval, eps=1e-3) into weight and bias (clair3/model.py:191,195-197):
//   scale = gamma / sqrt(var + eps);  w' = w * scale;  b' = (b - mean) * scale + beta
// layout [Cout][kh][kw][Cin]; conv1 additionally folds x/100 (model.py:378) and pads each kh to 32 slots.
// Channel equalisation (exact): ReLU is positively homogeneous, so the network computes the same function when an activation
// channel is multiplied by 2^k -- its producers' folded weights and bias times 2^k, the weights of its consumers that read it
// times 2^-k; powers of two, so every product and sum scales exactly and the rows are the ones of the checkpoint as given.
// A trained checkpoint may keep a channel at 1e-3 of its neighbours (a small BatchNorm gamma, compensated by large weights in
// the next layer); as fp16 piece planes such a channel would carry its values with the ABSOLUTE precision of fp16 subnormals
// (6e-8: 6e-5 relative at 1e-3).  k is chosen per channel so that |gamma| + |beta| of its producers -- the magnitude the
// BatchNorm gives it -- lands in [1, 2).  Channels of a stage are produced by the stage convolution and by the residual
// block's second convolution (the identity add: one k for both) and read by the block's first convolution and the next stage
// (the last stage: by L4 through the pooling); the channels inside a block by conv1+bn1 / conv2.
struct FaChannelExps {
    std::vector<int> stage[3], inner[3];  // per stage: exponents of the stage's channels / of the block's inner channels
    const std::vector<int> *out_of(int l) const { return l % 3 == 1 ? &inner[l / 3] : &stage[l / 3]; }
    const std::vector<int> *in_of(int l) const { return l == 0 ? nullptr : (l % 3 == 2 ? &inner[l / 3] : &stage[(l - 1) / 3]); }
};
static int fa_channel_exps(const TensorMap &tm, FaChannelExps &ex) {
    auto mags = [&](int l, std::vector<double> &mag) -> int {
        const int Cout = kConvCout[l];
        const float *g, *beta;
        TRY(want(tm, std::string(kBnName[l]) + ".weight", {Cout}, &g));
        TRY(want(tm, std::string(kBnName[l]) + ".bias", {Cout}, &beta));
        mag.resize(Cout);
        for (int c = 0; c < Cout; ++c) mag[c] = std::fabs((double)g[c]) + std::fabs((double)beta[c]);
        return 0;
    };
    auto exps = [](const std::vector<double> &mag, std::vector<int> &k) {
        k.assign(mag.size(), 0);
        for (size_t c = 0; c < mag.size(); ++c) {
            if (!(mag[c] > 0.0) || !std::isfinite(mag[c])) continue;
            int e;
            (void)std::frexp(mag[c], &e);                       // mag = f * 2^e, f in [0.5, 1)
            k[c] = std::min(40, std::max(-40, 1 - e));          // mag * 2^k in [1, 2)
        }
    };
    for (int s = 0; s < 3; ++s) {
        std::vector<double> a, b2, in;
        TRY(mags(3 * s, a));
        TRY(mags(3 * s + 2, b2));
        TRY(mags(3 * s + 1, in));
        for (size_t c = 0; c < a.size(); ++c) a[c] = std::max(a[c], b2[c]);
        exps(a, ex.stage[s]);
        exps(in, ex.inner[s]);
    }
    return 0;
}

static int pack_conv(c3_model *m, const TensorMap &tm, int l, int Cin, const FaChannelExps &ex) {
    const int Cout = kConvCout[l];
    const std::vector<int> &kout = *ex.out_of(l);
    const std::vector<int> *kin = ex.in_of(l);
    // fold(co) = gamma / sqrt(var + eps) * 2^k_out(co);  every weight additionally times 2^-k_in(ci)
    auto fold = [&](const float *g, const float *var, int co) { return std::ldexp((double)g[co] / std::sqrt((double)var[co] + 1e-3), kout[co]); };
    auto in_f = [&](int ci) { return kin ? std::ldexp(1.0, -(*kin)[ci]) : 1.0; };
    const float *w, *b, *g, *beta, *mean, *var;
    const std::string cv = kConvName[l], bn = kBnName[l];
    TRY(want(tm, cv + ".weight", {Cout, Cin, 3, 3}, &w));
    TRY(want(tm, cv + ".bias", {Cout}, &b));
    TRY(want(tm, bn + ".weight", {Cout}, &g));
    TRY(want(tm, bn + ".bias", {Cout}, &beta));
    TRY(want(tm, bn + ".running_mean", {Cout}, &mean));
    TRY(want(tm, bn + ".running_var", {Cout}, &var));
    const int ldb = l == 0 ? 96 : 9 * Cin;
    std::vector<float> pw((size_t)Cout * ldb, 0.f), pb(Cout);
    for (int co = 0; co < Cout; ++co) {
        const double scale = fold(g, var, co);
        pb[co] = (float)(((double)b[co] - (double)mean[co]) * scale + std::ldexp((double)beta[co], kout[co]));
        for (int ci = 0; ci < Cin; ++ci)
            for (int kh = 0; kh < 3; ++kh)
                for (int kw = 0; kw < 3; ++kw) {
                    const double v = (double)w[(((size_t)co * Cin + ci) * 3 + kh) * 3 + kw] * scale * in_f(ci);
                    if (l == 0)
                        pw[(size_t)co * ldb + kh * 32 + kw * Cin + ci] = (float)(v / 100.0);
                    else
                        pw[(size_t)co * ldb + (size_t)(kh * 3 + kw) * Cin + ci] = (float)v;
                }
    }
    TRY(upload(m, &m->conv_w[l], pw));
    TRY(upload(m, &m->conv_b[l], pb));
    if (l == 0 && Cin != 8)  // conv1 as its own launch runs on the tiled contraction unless the window has 8 channels
    {
        std::vector<float> sc, post;
        row_scales(pw.data(), Cout, (size_t)ldb, sc, post);
        TRY(upload_split_pieces(m, &m->conv1_w16, pw, Cout, sc));
        TRY(upload(m, &m->conv1_w16_post, post));
    }
    std::vector<float> c1s, c1post;  // conv1's fragments carry their output channel's power of two as well (undone by c1post in the kernels)
    if (l == 0 && (Cin == 8 || Cin == 9)) {
        std::vector<float> v1((size_t)64 * 9 * Cin);
        for (int co = 0; co < 64; ++co) {
            const double scale = fold(g, var, co);
            for (int i = 0; i < 9 * Cin; ++i) v1[(size_t)co * 9 * Cin + i] = (float)((double)w[(size_t)co * 9 * Cin + i] * scale * (128.0 / 100.0));
        }
        row_scales(v1.data(), 64, (size_t)9 * Cin, c1s, c1post);
        TRY(upload(m, &m->conv1_post, c1post));
    }
    if (l == 0 && Cin == 8) {
        {
            // conv1_i8_f16_kernel: lane (n = lane & 31, kh = lane >> 5) of k-step t holds channel j of tap 2 t + kh
            std::vector<float> pf16((size_t)5 * 2 * 2 * 64 * 4, 0.f);
            uint16_t *q16 = reinterpret_cast<uint16_t *>(pf16.data());
            for (int t = 0; t < 5; ++t)
                for (int cb = 0; cb < 2; ++cb)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 8; ++j) {
                            const int co = 32 * cb + (lane & 31), tap = 2 * t + (lane >> 5);
                            float v = 0.f;
                            if (tap < 9) {
                                const double scale = fold(g, var, co);
                                v = (float)((double)w[(((size_t)co * Cin + j) * 3 + tap / 3) * 3 + tap % 3] * scale * (128.0 / 100.0)) * c1s[co];  // the kernel feeds x / 128
                            }
                            const _Float16 h0 = (_Float16)v, h1 = (_Float16)(v - (float)h0);
                            memcpy(&q16[((((size_t)t * 2 + cb) * 2 + 0) * 64 + lane) * 8 + j], &h0, 2);
                            memcpy(&q16[((((size_t)t * 2 + cb) * 2 + 1) * 64 + lane) * 8 + j], &h1, 2);
                        }
            float mx = 0.f;  // the fp16 form needs its weights inside the fp16 range; a checkpoint with a degenerate BatchNorm stays on fp32
            for (size_t i = 0; i < pf16.size() * 2; ++i) {
                _Float16 h;
                memcpy(&h, &q16[i], 2);
                mx = std::max(mx, std::fabs((float)h));
            }
            if (mx < 16384.f) TRY(upload(m, &m->conv1_wfrag16, pf16));
        }
    }
    if (l == 0 && Cin == 9) {
        // conv1 inside conv3x3_planes_kernel<.., C1 = 9> (c3_conv3.h): k-step t = (patch row ky = t >> 1, half u = t & 1); lane
        // (n = lane & 31, kh = lane >> 5) holds the weights of bytes q = 16 u + 8 kh + j of the row's three 9-byte pixels
        // (pixel q / 9, channel q % 9; q >= 27: padding, zero), times 1.28 (the kernel feeds x / 128), as two fp16 pieces
        std::vector<float> pf16((size_t)6 * 2 * 2 * 64 * 4, 0.f);
        uint16_t *q16 = reinterpret_cast<uint16_t *>(pf16.data());
        float mx = 0.f;
        for (int t = 0; t < 6; ++t)
            for (int cb = 0; cb < 2; ++cb)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int co = 32 * cb + (lane & 31), q = 16 * (t & 1) + 8 * (lane >> 5) + j, ky = t >> 1;
                        float v = 0.f;
                        if (q < 27) {
                            const double scale = fold(g, var, co);
                            v = (float)((double)w[(((size_t)co * Cin + q % 9) * 3 + ky) * 3 + q / 9] * scale * (128.0 / 100.0)) * c1s[co];
                        }
                        const _Float16 h0 = (_Float16)v, h1 = (_Float16)(v - (float)h0);
                        memcpy(&q16[((((size_t)t * 2 + cb) * 2 + 0) * 64 + lane) * 8 + j], &h0, 2);
                        memcpy(&q16[((((size_t)t * 2 + cb) * 2 + 1) * 64 + lane) * 8 + j], &h1, 2);
                        mx = std::max(mx, std::fabs(v));
                    }
        if (mx < 16384.f) TRY(upload(m, &m->conv1_wfrag16, pf16));  // else: conv1 stays on the tiled GEMM
    }
    if (kConvStride[l] == 1 && Cin == Cout && Cin % 64 == 0) {
        // conv3x3_planes_kernel (c3_conv3.h): chunk (column tile tn, input slab, tap) = 64 couts x 64 channels x 2 pieces = 16 KB in
        // FRAGMENT order -- [cout half wn][k-step][piece hi | lo][lane] x 16 B: lane (n = lane & 31, kh = lane >> 5) holds channels
        // 64 slab + 8 (2 ks + kh) .. + 7 of cout 64 tn + 32 wn + n, times the output channel's power of two (row_scales) -- so a wave's
        // matrix operand of one k-step is one contiguous kilobyte per piece, loaded straight into registers
        const int NS = Cin / 64;
        std::vector<float> sc, post;
        row_scales(pw.data(), Cout, (size_t)ldb, sc, post);
        TRY(upload(m, &m->pconv_pre[l], sc));
        TRY(upload(m, &m->pconv_post[l], post));
        std::vector<float> pk((size_t)NS * NS * 9 * 64 * 64);  // 16 KB per chunk
        uint16_t *q16 = reinterpret_cast<uint16_t *>(pk.data());
        for (int tn = 0; tn < NS; ++tn)
            for (int slab = 0; slab < NS; ++slab)
                for (int tap = 0; tap < 9; ++tap)
                    for (int wn = 0; wn < 2; ++wn)
                        for (int ks = 0; ks < 4; ++ks)
                            for (int lane = 0; lane < 64; ++lane)
                                for (int j = 0; j < 8; ++j) {
                                    const int co = tn * 64 + 32 * wn + (lane & 31), ci = slab * 64 + 8 * (2 * ks + (lane >> 5)) + j;
                                    const float v = pw[(size_t)co * ldb + (size_t)tap * Cin + ci] * sc[co];  // exact
                                    const _Float16 h0 = (_Float16)v, h1 = (_Float16)(v - (float)h0);
                                    const size_t dst = (((((((size_t)tn * NS + slab) * 9 + tap) * 2 + wn) * 4 + ks) * 2) * 64 + lane) * 8 + j;
                                    memcpy(&q16[dst], &h0, 2);
                                    memcpy(&q16[dst + 64 * 8], &h1, 2);
                                }
        TRY(upload(m, &m->pconv_w[l], pk));
        // conv3x3_wino_planes_kernel (c3_conv3w.h): F(2,3) along the rows.  Per column tap kw the three row taps g0, g1, g2 of a
        // (cout, cin) pair become U0 = g0, U1 = (g0 + g1 + g2) / 2, U2 = (g0 - g1 + g2) / 2, U3 = g2 -- formed in double from the
        // folded fp32 weights, rounded to fp32 once, then every output row times its own power of two and split into the two fp16
        // pieces.  Chunk (column tile tn, 32-channel slab, tap = xi * 3 + kw) = 8 KB in fragment order [cout half wn][k-step][hi | lo][lane]
        // x 16 B: lane (n = lane & 31, kh = lane >> 5) holds channels 32 slab + 8 (2 ks + kh) .. + 7 of cout 64 tn + 32 wn + n.
        {
            const int NS32 = Cin / 32;
            std::vector<float> u((size_t)Cout * 12 * Cin);
            for (int co = 0; co < Cout; ++co)
                for (int kw = 0; kw < 3; ++kw)
                    for (int ci = 0; ci < Cin; ++ci) {
                        const double g0 = pw[(size_t)co * ldb + (size_t)(0 + kw) * Cin + ci], g1 = pw[(size_t)co * ldb + (size_t)(3 + kw) * Cin + ci],
                                     g2 = pw[(size_t)co * ldb + (size_t)(6 + kw) * Cin + ci];
                        const double uu[4] = {g0, 0.5 * (g0 + g1 + g2), 0.5 * (g0 - g1 + g2), g2};
                        for (int xi = 0; xi < 4; ++xi) u[((size_t)co * 12 + xi * 3 + kw) * Cin + ci] = (float)uu[xi];
                    }
            std::vector<float> scw, postw;
            row_scales(u.data(), Cout, (size_t)12 * Cin, scw, postw);
            TRY(upload(m, &m->wconv_post[l], postw));
            std::vector<float> pu((size_t)NS * NS32 * 12 * 2048);  // 8 KB per chunk
            uint16_t *u16 = reinterpret_cast<uint16_t *>(pu.data());
            for (int tn = 0; tn < NS; ++tn)
                for (int s32 = 0; s32 < NS32; ++s32)
                    for (int tap = 0; tap < 12; ++tap)
                        for (int wn = 0; wn < 2; ++wn)
                            for (int ks = 0; ks < 2; ++ks)
                                for (int lane = 0; lane < 64; ++lane)
                                    for (int j = 0; j < 8; ++j) {
                                        const int co = tn * 64 + 32 * wn + (lane & 31), ci = s32 * 32 + 8 * (2 * ks + (lane >> 5)) + j;
                                        const float v = u[((size_t)co * 12 + tap) * Cin + ci] * scw[co];  // exact
                                        const _Float16 h0 = (_Float16)v, h1 = (_Float16)(v - (float)h0);
                                        const size_t dst = (((((((size_t)tn * NS32 + s32) * 12 + tap) * 2 + wn) * 2 + ks) * 2) * 64 + lane) * 8 + j;
                                        memcpy(&u16[dst], &h0, 2);
                                        memcpy(&u16[dst + 64 * 8], &h1, 2);
                                    }
            TRY(upload(m, &m->wconv_w[l], pu));
        }
    }
    if (kConvStride[l] == 2 && l > 0 && Cin % 64 == 0 && Cout % kS2BN == 0) {
        // conv3x3_s2_planes_kernel (c3_conv3s2.h): column tiles of 64 couts, chunk kc = tap * Cin/64 + slab = 16 KB in the FRAGMENT order of
        // the stride-1 layers above -- [cout half wn][k-step][piece hi | lo][lane] x 16 B, lane (n = lane & 31, kh = lane >> 5) holding
        // channels 64 slab + 8 (2 ks + kh) .. + 7 of cout 64 tn + 32 wn + n, times the output channel's power of two
        const int NS = Cin / 64, NKc = 9 * NS;
        std::vector<float> sc, post;
        row_scales(pw.data(), Cout, (size_t)ldb, sc, post);
        TRY(upload(m, &m->pconv_pre[l], sc));
        TRY(upload(m, &m->pconv_post[l], post));
        std::vector<float> pf((size_t)Cout * NKc * 64);
        uint16_t *f16 = reinterpret_cast<uint16_t *>(pf.data());
        for (int tn = 0; tn < Cout / 64; ++tn)
            for (int kc = 0; kc < NKc; ++kc)
                for (int wn = 0; wn < 2; ++wn)
                    for (int ks = 0; ks < 4; ++ks)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int j = 0; j < 8; ++j) {
                                const int tap = kc / NS, slab = kc % NS;
                                const int co = tn * 64 + 32 * wn + (lane & 31), ci = slab * 64 + 8 * (2 * ks + (lane >> 5)) + j;
                                const float v = pw[(size_t)co * ldb + (size_t)tap * Cin + ci] * sc[co];  // exact
                                const _Float16 h0 = (_Float16)v, h1 = (_Float16)(v - (float)h0);
                                const size_t dst = ((((((size_t)tn * NKc + kc) * 2 + wn) * 4 + ks) * 2) * 64 + lane) * 8 + j;
                                memcpy(&f16[dst], &h0, 2);
                                memcpy(&f16[dst + 64 * 8], &h1, 2);
                            }
        TRY(upload(m, &m->pconv_w[l], pf));
    }
    return 0;
}


/*
 * c3_oracle.c -- TEST INFRASTRUCTURE ONLY. See c3_oracle.h.
 *
 * Everything is computed in double precision from the float32 parameters; outputs are rounded to
 * float32 once at the end.  This makes the oracle an independent "second opinion" next to the
 * float32 PyTorch reference (measured distance between the two: <= ~1e-7 on probabilities).
 */
#include "c3_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define SELU_SCALE 1.0507009873554804934193349852946
#define SELU_ALPHA 1.6732632423543772848170429916717

static inline double selu(double x) { return x > 0.0 ? SELU_SCALE * x : SELU_SCALE * SELU_ALPHA * expm1(x); }
static inline double sigmoid(double x) { return 1.0 / (1.0 + exp(-x)); }

/* y[o] = b[o] + sum_i w[o*in+i] * x[i]   (nn.Linear) */
static void linear(const c3o_linear *l, const double *x, int in, int out, double *y) {
    for (int o = 0; o < out; ++o) {
        double acc = l->b ? (double)l->b[o] : 0.0;
        const float *wr = l->w + (size_t)o * in;
        for (int i = 0; i < in; ++i) acc += (double)wr[i] * x[i];
        y[o] = acc;
    }
}

/* softmax(selu(logits)) -- clair3/model.py:142-150 applies SELU to the logits before softmax */
static void selu_softmax(const double *logits, int n, float *out) {
    double v[64];
    double m = -1e300;
    for (int i = 0; i < n; ++i) {
        v[i] = selu(logits[i]);
        if (v[i] > m) m = v[i];
    }
    double s = 0.0;
    for (int i = 0; i < n; ++i) {
        v[i] = exp(v[i] - m);
        s += v[i];
    }
    for (int i = 0; i < n; ++i) out[i] = (float)(v[i] / s);
}

/* FC tail shared by both networks: x (post-SELU L4 output, width fc) -> 24/90 probabilities.
 * clair3/model.py:139-159 / 394-414 */
static void fc_tail(const c3o_linear *L5, const c3o_linear *head, const double *x, int fc, int add_indel, float *y) {
    static const int head_n[4] = {21, 3, 33, 33};
    int nb = add_indel ? 4 : 2;
    int off = 0;
    for (int b = 0; b < nb; ++b) {
        double h[128], lg[64];
        linear(&L5[b], x, fc, 128, h);
        for (int i = 0; i < 128; ++i) h[i] = selu(h[i]);
        linear(&head[b], h, 128, head_n[b], lg);
        selu_softmax(lg, head_n[b], y + off);
        off += head_n[b];
    }
}

/* ---------------------------------------------------------------- pileup (Clair3_P) */

/* One direction of one LSTM layer over T steps (torch.nn.LSTM semantics, h0=c0=0):
 *   gates = W_ih x_t + b_ih + W_hh h_{t-1} + b_hh ; rows [i | f | g | o]
 *   c_t = sigmoid(f) c_{t-1} + sigmoid(i) tanh(g) ; h_t = sigmoid(o) tanh(c_t)
 * reverse=1 walks t = T-1..0 and stores h at its own t.
 * out: (T, out_stride) with this direction's H values at column offset out_off. */
static void lstm_dir(const c3o_lstm_dir *p, const double *x, int T, int in, int H, int reverse, double *out,
                     int out_stride, int out_off) {
    double *h = (double *)calloc((size_t)H, sizeof(double));
    double *c = (double *)calloc((size_t)H, sizeof(double));
    double *g = (double *)malloc(sizeof(double) * 4 * (size_t)H);
    for (int s = 0; s < T; ++s) {
        int t = reverse ? T - 1 - s : s;
        const double *xt = x + (size_t)t * in;
        for (int r = 0; r < 4 * H; ++r) {
            double acc = (double)p->b_ih[r] + (double)p->b_hh[r];
            const float *wi = p->w_ih + (size_t)r * in;
            for (int i = 0; i < in; ++i) acc += (double)wi[i] * xt[i];
            const float *wh = p->w_hh + (size_t)r * H;
            for (int j = 0; j < H; ++j) acc += (double)wh[j] * h[j];
            g[r] = acc;
        }
        for (int j = 0; j < H; ++j) {
            double ig = sigmoid(g[j]);
            double fg = sigmoid(g[H + j]);
            double gg = tanh(g[2 * H + j]);
            double og = sigmoid(g[3 * H + j]);
            c[j] = fg * c[j] + ig * gg;
            h[j] = og * tanh(c[j]);
        }
        for (int j = 0; j < H; ++j) out[(size_t)t * out_stride + out_off + j] = h[j];
    }
    free(h);
    free(c);
    free(g);
}

int c3o_pileup_forward(const c3o_pileup_weights *w, const void *x, int x_itemsize, long B, int T, int C,
                       int add_indel_length, float *y, const c3o_pileup_debug *dbg, int n_threads) {
    if (!w || !x || !y || B < 0 || (x_itemsize != 1 && x_itemsize != 4)) return -1;
    const int H1 = 128, H2 = 160;
    const int ny = add_indel_length ? 90 : 24;
    int fail = 0;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#else
    (void)n_threads;
#endif
#pragma omp parallel for schedule(dynamic, 4)
    for (long b = 0; b < B; ++b) {
        double *xin = (double *)malloc(sizeof(double) * (size_t)T * C);
        double *o1 = (double *)malloc(sizeof(double) * (size_t)T * 2 * H1);
        double *o2 = (double *)malloc(sizeof(double) * (size_t)T * 2 * H2);
        double l4[128];
        if (!xin || !o1 || !o2) {
            fail = 1;
            free(xin), free(o1), free(o2);
            continue;
        }
        /* x.float() -- clair3/model.py:131 */
        if (x_itemsize == 1) {
            const int8_t *xb = (const int8_t *)x + (size_t)b * T * C;
            for (int i = 0; i < T * C; ++i) xin[i] = (double)xb[i];
        } else {
            const int32_t *xb = (const int32_t *)x + (size_t)b * T * C;
            for (int i = 0; i < T * C; ++i) xin[i] = (double)xb[i];
        }
        /* LSTM1 / LSTM2 bidirectional, output = [h_fwd(t) || h_bwd(t)] -- model.py:132-133 */
        lstm_dir(&w->lstm1[0], xin, T, C, H1, 0, o1, 2 * H1, 0);
        lstm_dir(&w->lstm1[1], xin, T, C, H1, 1, o1, 2 * H1, H1);
        lstm_dir(&w->lstm2[0], o1, T, 2 * H1, H2, 0, o2, 2 * H2, 0);
        lstm_dir(&w->lstm2[1], o1, T, 2 * H1, H2, 1, o2, 2 * H2, H2);
        /* flatten (T*320) -> L4 -> SELU -- model.py:135-136 */
        linear(&w->L4, o2, T * 2 * H2, 128, l4);
        for (int i = 0; i < 128; ++i) l4[i] = selu(l4[i]);
        fc_tail(w->L5, w->head, l4, 128, add_indel_length, y + (size_t)b * ny);
        if (dbg) {
            if (dbg->lstm1_out)
                for (int i = 0; i < T * 2 * H1; ++i) dbg->lstm1_out[(size_t)b * T * 2 * H1 + i] = (float)o1[i];
            if (dbg->lstm2_out)
                for (int i = 0; i < T * 2 * H2; ++i) dbg->lstm2_out[(size_t)b * T * 2 * H2 + i] = (float)o2[i];
            if (dbg->l4_out)
                for (int i = 0; i < 128; ++i) dbg->l4_out[(size_t)b * 128 + i] = (float)l4[i];
        }
        free(xin), free(o1), free(o2);
    }
    return fail ? -2 : 0;
}

/* ---------------------------------------------------------------- full alignment (Clair3_F) */

static inline int conv_out(int n, int stride) { return (n + 2 - 3) / stride + 1; }

/* 3x3 conv, padding 1, NHWC activations in double; weights (out,in,3,3); then BN(eval, eps=1e-3);
 * optional residual add (before the ReLU); ReLU.  clair3/model.py:194-197, 225-235 */
static void conv_bn_relu(const c3o_convbn *p, const double *in, int H, int W, int Cin, int Cout, int stride,
                         const double *residual, double *out) {
    const int Ho = conv_out(H, stride), Wo = conv_out(W, stride);
    double *acc = (double *)malloc(sizeof(double) * (size_t)Cout);
    /* re-layout weights once as [kh][kw][ci][co] so the inner loop runs over co */
    double *wt = (double *)malloc(sizeof(double) * 9 * (size_t)Cin * Cout);
    for (int co = 0; co < Cout; ++co)
        for (int ci = 0; ci < Cin; ++ci)
            for (int k = 0; k < 9; ++k)
                wt[((size_t)k * Cin + ci) * Cout + co] = (double)p->w[((size_t)co * Cin + ci) * 9 + k];
    for (int oh = 0; oh < Ho; ++oh)
        for (int ow = 0; ow < Wo; ++ow) {
            for (int co = 0; co < Cout; ++co) acc[co] = (double)p->b[co];
            for (int kh = 0; kh < 3; ++kh) {
                int ih = oh * stride + kh - 1;
                if (ih < 0 || ih >= H) continue;
                for (int kw = 0; kw < 3; ++kw) {
                    int iw = ow * stride + kw - 1;
                    if (iw < 0 || iw >= W) continue;
                    const double *px = in + ((size_t)ih * W + iw) * Cin;
                    const double *wk = wt + (size_t)(kh * 3 + kw) * Cin * Cout;
                    for (int ci = 0; ci < Cin; ++ci) {
                        double v = px[ci];
                        const double *wr = wk + (size_t)ci * Cout;
                        for (int co = 0; co < Cout; ++co) acc[co] += v * wr[co];
                    }
                }
            }
            double *po = out + ((size_t)oh * Wo + ow) * Cout;
            const double *pr = residual ? residual + ((size_t)oh * Wo + ow) * Cout : 0;
            for (int co = 0; co < Cout; ++co) {
                double v = (acc[co] - (double)p->bn_mean[co]) / sqrt((double)p->bn_var[co] + 1e-3) *
                               (double)p->bn_w[co] +
                           (double)p->bn_b[co];
                if (pr) v += pr[co];
                po[co] = v > 0.0 ? v : 0.0;
            }
        }
    free(acc);
    free(wt);
}

/* PyramidPolling, clair3/model.py:250-279: bins 3,2,1; window=stride=ceil(n/bin); zero padding split
 * floor/ceil (top/left get the floor); flatten order (h, w, c); concatenated bin 3, 2, 1. */
static int spp(const double *in, int H, int W, int C, double *out) {
    static const int bins[3] = {3, 2, 1};
    int n = 0;
    for (int bi = 0; bi < 3; ++bi) {
        int p = bins[bi];
        int wh = (H + p - 1) / p, ww = (W + p - 1) / p;
        int oh_n = (H + wh - 1) / wh, ow_n = (W + ww - 1) / ww;
        int pad_h = (oh_n - 1) * wh + wh - H;
        int pad_w = (ow_n - 1) * ww + ww - W;
        if (pad_h < 0) pad_h = 0;
        if (pad_w < 0) pad_w = 0;
        int pad_top = pad_h / 2, pad_left = pad_w / 2;
        for (int oh = 0; oh < oh_n; ++oh)
            for (int ow = 0; ow < ow_n; ++ow)
                for (int c = 0; c < C; ++c) {
                    double m = -1e300;
                    for (int dh = 0; dh < wh; ++dh)
                        for (int dw = 0; dw < ww; ++dw) {
                            int ih = oh * wh + dh - pad_top, iw = ow * ww + dw - pad_left;
                            double v = (ih < 0 || ih >= H || iw < 0 || iw >= W) ? 0.0 /* F.pad value=0 */
                                                                                  : in[((size_t)ih * W + iw) * C + c];
                            if (v > m) m = v;
                        }
                    out[n++] = m;
                }
    }
    return n;
}

int c3o_fa_forward(const c3o_fa_weights *w, const int8_t *x, long B, int H, int W, int C, int add_indel_length,
                   float *y, const c3o_fa_debug *dbg, int n_threads) {
    if (!w || !x || !y || B < 0) return -1;
    const int ny = add_indel_length ? 90 : 24;
    /* layer table: {Cin, Cout, stride, residual-source layer (-1 none)} */
    const int cin[9] = {C, 64, 64, 64, 128, 128, 128, 256, 256};
    const int cout[9] = {64, 64, 64, 128, 128, 128, 256, 256, 256};
    const int strd[9] = {2, 1, 1, 2, 1, 1, 2, 1, 1};
    int hh[10], ww[10];
    hh[0] = H, ww[0] = W;
    for (int l = 0; l < 9; ++l) hh[l + 1] = conv_out(hh[l], strd[l]), ww[l + 1] = conv_out(ww[l], strd[l]);
    const int H3 = hh[9], W3 = ww[9];
    const int spp_n = 14 * 256;
    int fail = 0;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#else
    (void)n_threads;
#endif
#pragma omp parallel for schedule(dynamic, 1)
    for (long b = 0; b < B; ++b) {
        size_t max_act = (size_t)hh[1] * ww[1] * 64;
        if ((size_t)H * W * C > max_act) max_act = (size_t)H * W * C;
        double *buf[4];
        for (int i = 0; i < 4; ++i) buf[i] = (double *)malloc(sizeof(double) * max_act);
        double *pool = (double *)malloc(sizeof(double) * spp_n);
        double l4[256];
        if (!buf[0] || !buf[1] || !buf[2] || !buf[3] || !pool) {
            fail = 1;
            for (int i = 0; i < 4; ++i) free(buf[i]);
            free(pool);
            continue;
        }
        /* inputs.float() / NORMALIZE_NUM -- model.py:378, shared/param_f.py NORMALIZE_NUM = 100 */
        const int8_t *xb = x + (size_t)b * H * W * C;
        for (size_t i = 0; i < (size_t)H * W * C; ++i) buf[0][i] = (double)xb[i] / 100.0;
        /* rotate through buffers: cur -> a (conv s2) -> t (block conv1) -> o (block conv2 + a) */
        double *cur = buf[0], *a = buf[1], *t = buf[2], *o = buf[3];
        for (int s = 0; s < 3; ++s) {
            int l = 3 * s;
            conv_bn_relu(&w->conv[l], cur, hh[l], ww[l], cin[l], cout[l], strd[l], 0, a);
            conv_bn_relu(&w->conv[l + 1], a, hh[l + 1], ww[l + 1], cin[l + 1], cout[l + 1], 1, 0, t);
            conv_bn_relu(&w->conv[l + 2], t, hh[l + 2], ww[l + 2], cin[l + 2], cout[l + 2], 1, a, o);
            if (dbg) {
                double *src[3] = {a, t, o};
                for (int k = 0; k < 3; ++k)
                    if (dbg->act[l + k]) {
                        size_t n = (size_t)hh[l + k + 1] * ww[l + k + 1] * cout[l + k];
                        float *d = dbg->act[l + k] + (size_t)b * n;
                        for (size_t i = 0; i < n; ++i) d[i] = (float)src[k][i];
                    }
            }
            double *tmp = cur;
            cur = o;
            o = tmp;
        }
        int n = spp(cur, H3, W3, 256, pool);
        if (n != spp_n) fail = 1; /* ONT geometry gives exactly 14 bins */
        linear(&w->L4, pool, spp_n, 256, l4);
        for (int i = 0; i < 256; ++i) l4[i] = selu(l4[i]);
        fc_tail(w->L5, w->head, l4, 256, add_indel_length, y + (size_t)b * ny);
        if (dbg) {
            if (dbg->spp)
                for (int i = 0; i < spp_n; ++i) dbg->spp[(size_t)b * spp_n + i] = (float)pool[i];
            if (dbg->l4_out)
                for (int i = 0; i < 256; ++i) dbg->l4_out[(size_t)b * 256 + i] = (float)l4[i];
        }
        for (int i = 0; i < 4; ++i) free(buf[i]);
        free(pool);
    }
    return fail ? -2 : 0;
}

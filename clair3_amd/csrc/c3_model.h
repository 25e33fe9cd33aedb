// c3_model.h -- the model handle of libc3hip.so and the helpers every host-side unit shares: error convention, device
// memory, the profiling scope, the launch wrapper of the tiled contraction.
//
// The host side of the library is one translation unit (c3_model.hip) made of:
//   c3_model.h     this file
//   c3_pack.h      c3_model_load: BatchNorm folding, gate re-ordering, matrix-instruction fragment layouts, fp16 pieces
//   c3_forward.h   the launch sequences of the two forward passes (clair3/model.py:130-161 and :377-416)
//   c3_hostring.h  the host <-> device ring behind c3_predict / c3_predict_submit / _wait (staging, transfers, range guard)
//   c3_comm.h      the gather of a sharded job on RCCL
//   c3_debug.h     c3_debug_* / c3_profile_* (parity tests, bench.py)
//   c3_model.hip   create / geometry / device-resident entries / describe / destroy
// Every layer has exactly two forms: the product (fp16x3 split products on the 16-bit matrix instructions) and
// one fp32-MFMA form that the range guard falls back to (and that C3HIP_FP32=1 selects from the start).
#pragma once
#include <hip/hip_runtime.h>
#include <errno.h>
#include <sys/mman.h>

#include <algorithm>
#include <functional>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/c3hip.h"
#include "c3_gemm.h"
#include "c3_kernels.h"
#include "c3_conv1.h"
#include "c3_tail.h"
#include "c3_decode.h"
#include "c3_lstm_fused.h"
#include "c3_host.h"
#include "c3_conv3.h"
#include "c3_conv3s2.h"
#include "c3_conv3w.h"
#include "c3_l4.h"
#include "c3_dense.h"

using namespace c3;

// ------------------------------------------------------------------------------------------ errors
static thread_local std::string g_err;
static int fail(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return 1;
}
#define HIP_TRY(expr)                                                                                    \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) return fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)
#define TRY(expr)            \
    do {                     \
        int rc_ = (expr);    \
        if (rc_) return rc_; \
    } while (0)

// ------------------------------------------------------------------------------------------ model
static const int kHeadN[4] = {21, 3, 33, 33};
static const char *kHeadName[4] = {"Y_gt21_logits", "Y_genotype_logits", "Y_indel_length_logits_1",
                                   "Y_indel_length_logits_2"};
static const char *kConvName[9] = {"conv1.conv",         "res_block1.0.conv1", "res_block1.0.conv2",
                                   "conv3.conv",         "res_block2.0.conv1", "res_block2.0.conv2",
                                   "conv5.conv",         "res_block3.0.conv1", "res_block3.0.conv2"};
static const char *kBnName[9] = {"conv1.bn",         "res_block1.0.bn1", "res_block1.0.bn2",
                                 "conv3.bn",         "res_block2.0.bn1", "res_block2.0.bn2",
                                 "conv5.bn",         "res_block3.0.bn1", "res_block3.0.bn2"};
static const int kConvCout[9] = {64, 64, 64, 128, 128, 128, 256, 256, 256};
static const int kConvStride[9] = {2, 1, 1, 2, 1, 1, 2, 1, 1};
static const char *kFaLayerTag[9] = {"fa.conv1", "fa.res1a", "fa.res1b", "fa.conv3", "fa.res2a",
                                     "fa.res2b", "fa.conv5", "fa.res3a", "fa.res3b"};

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
};

struct ProfRec {
    std::string name;
    hipEvent_t a, b;
    double flops, bytes;
    double mfma_flops = 0.0;  // FLOP the matrix instructions of the launch EXECUTE (tile padding, piece products included)
    double mfma_peak = 0.0;   // dense peak (TFLOP/s) of the matrix instruction the launch issues: 2500 (16-bit) or 157.3 (fp32)
};

struct HostSlot {
    void *pin_x = nullptr;
    float *pin_y = nullptr;
    void *dev_x = nullptr;
    float *dev_y = nullptr;
    size_t cap_x = 0, cap_y = 0;
    hipEvent_t ev_h2d = nullptr, ev_out = nullptr;
    float *y_host = nullptr;
    float *y_dev_out = nullptr;  // c3_predict_submit_dev: the rows stay in the caller's device buffer
    size_t y_bytes = 0;
    int64_t batch = 0;  // what is in flight (for the fp32 re-run of c3_predict_wait)
    uint32_t *pin_flag = nullptr;  // pinned copy of the model's range_flag after this batch
    int x_dtype = 0;
    bool busy = false;
    bool used_f16 = false;  // the batch in flight was computed by the fp16x3 kernels (c3_predict_wait then checks its range)
    int lane = 0;           // the lane (c3_model::Lane) the batch in flight runs in
};

constexpr int kHostSlots = 4;  // batches in flight per handle through c3_predict_submit / _wait (C3_HOST_SLOTS)

struct c3_model {
    int kind = 0, C = 0, add_indel = 0, device = 0;
    int depth = 89, positions = 33;
    int nb = 2, nout = 24;
    int row = 24;  // floats per output row: nout, + kDecodeCols when c3_model_set_decode_columns is on
    bool loaded = false;
    hipStream_t stream = nullptr, h2d_stream = nullptr;  // kernels (and the rows on their way out); staged windows on their way in
    hipStream_t duo_stream = nullptr;                    // the second half of a micro-batch (c3_forward.h forward_device, C3HIP_DUO)
    // The FC chain of a batch of the submit / wait ring on a stream of its own (round 6, c3_forward.h tail_split): L4, the split-K sum and
    // the tail are three small launches (224 / few / 64 workgroups, 20 - 30 us) behind which the NEXT batch's first layers would wait
    // although they depend on nothing of them; on their own stream they run in the slots the next batch's under-filled launches leave
    // free.  ev_body_done: the last layer in front of the chain; ev_tail_done: the chain (the next batch waits for it before it overwrites
    // what the chain reads: the pooled tensor / lstm2_out).  Default: on for full alignment, off for pileup (c3_model_create says why); env
    // C3HIP_TAIL_STREAM=0 / 1.
    hipStream_t tail_stream = nullptr;
    hipEvent_t ev_body_done = nullptr, ev_tail_done = nullptr;
    bool tail_split = false;   // allowed (set in c3_model_create: the kind's default, or env)
    bool tail_pending = false;  // a chain is (or may still be) running on tail_stream
    bool tail_now = false;      // this forward pass puts its chain on tail_stream (set by the ring's submit around forward_device)
    hipEvent_t duo_fork = nullptr, duo_join = nullptr;

    // ---- packed weights (device) ----
    // pileup
    float *proj_w[2] = {nullptr, nullptr};  // LSTM input projections [2*4H][Kp] fp32 (layer 1: the fp32 form of the LSTM2 projection)
    float *proj_b[2] = {nullptr, nullptr};  // [2*4H], b_ih + b_hh
    float *whh[2] = {nullptr, nullptr};     // W_hh as fp32 matrix-instruction fragments (fp32 forms)
    float *whh16[2] = {nullptr, nullptr};   // W_hh as two fp16 pieces in the same fragment order (c3_kernels.h, c3_lstm_fused.h)
    float *l1_wih = nullptr, *l1_bias = nullptr;  // LSTM1 input projection as fragments of the fused kernel (fp32; int32 windows)
    float *l1_wih16 = nullptr;                    // the same as two fp16 pieces of 128 W_ih (int8 windows)
    float *proj2_pw = nullptr;   // LSTM2 projection weights as dense_planes_pipe_kernel chunks (c3_dense.h)
    float *proj2_pwr = nullptr;  // the same in the register-fragment order of dense_planes_wres_kernel
    float *proj2_post = nullptr; // [1280] x 2^-k, one power of two for the whole matrix (c3_pack.h), as the vector dense_planes_pipe_kernel reads
    float proj2_post_scale = 1.f;  // the same 2^-k for dense_planes_wres_kernel
    // full alignment
    float *conv_w[9] = {};   // [Cout][9 Cin] fp32, BatchNorm folded (conv1: [64][3][32], /100 folded): the fp32 forms
    float *conv_b[9] = {};
    float *conv1_w16 = nullptr;  // conv1 of a window with C != 8 as two fp16 pieces for the tiled contraction (keep mode of the 9-channel model)
    float *conv1_w16_post = nullptr;  // its [64] per-channel 2^-k
    float *conv1_wfrag16 = nullptr;  // conv1 as fragments of conv1_i8_f16_kernel / conv3x3_planes_kernel's SRC8 forms (C = 8 or 9)
    float *conv1_post = nullptr;     // their [64] per-channel 2^-k
    float *pconv_w[9] = {};  // stride-1 convs: conv3x3_planes_kernel chunks in fragment order [Cout/64][Cin/64][9][2][4][2][64 lanes x 16 B];
                             // stride-2 convs: conv3x3_s2_planes_kernel chunks, the same fragment order [Cout/64][9 Cin/64][2][4][2][64 lanes x 16 B]
    float *pconv_pre[9] = {}, *pconv_post[9] = {};  // [Cout] the output channels' powers of two 2^k / 2^-k (c3_pack.h row_scales)
    float *wconv_w[9] = {};     // stride-1 convs: the F(2,3)-along-H weights U of conv3x3_wino_planes_kernel (c3_conv3w.h) in fragment
                                // order [Cout/64][Cin/32][12 taps][2][2][2][64 lanes x 16 B]
    float *wconv_post[9] = {};  // [Cout] 2^-k of U's rows
    // shared FC tail
    float *l4_w = nullptr, *l4_b = nullptr;  // [FC][K4] native layout (fp32 form)
    float *l4_wf = nullptr;                  // the same as two fp16 pieces in fragment order (c3_l4.h), every feature row times its power of two
    float *l4_pre = nullptr, *l4_post = nullptr;  // [FC] 2^k / 2^-k
    float *b5 = nullptr;
    float *w5f = nullptr, *whf = nullptr, *bh48 = nullptr;  // L5 / head weights as fragments of fc_tail_mfma_kernel (c3_tail.h)
    std::vector<int> act_exp[9];  // channel equalisation (c3_pack.h): activation channel c of layer l lives on the device times 2^act_exp[l][c]
    float *zeros = nullptr;  // 256-byte zero page: padding taps of the conv loaders read from here
    int FC = 0, K4 = 0;

    // ---- arithmetic ----
    uint32_t *range_flag = nullptr;  // device word set by the fp16x3 kernels when an activation nears the fp16 range (c3_gemm.h kF16Range)
    uint32_t *pin_flag = nullptr;    // pinned copy of range_flag for c3_predict_device_checked
    bool f16_ok = true;              // cleared when a batch came back out of range / non-finite (or by C3HIP_FP32=1): every layer then runs its fp32-MFMA form
    // Precision escalation decided at c3_model_load (pileup only, c3_pack.h lstm_sensitivity): a recurrence whose weights hold an entry of
    // magnitude >= auto_fp32_at amplifies the 2^-22 of the fp16 piece pairs over its 33 steps beyond north_star's 1e-4 on rare windows
    // (tests/diag/sensitive_window.py), so such a handle STARTS on the fp32 matrix instructions.  C3HIP_FP32 set (0 or 1) is an explicit
    // choice and switches the automatism off; C3HIP_AUTO_FP32=<threshold> moves it (0 = never).
    bool precision_forced = false;   // C3HIP_FP32 was given
    float auto_fp32_at = 4.0f;
    float lstm_wmax = 0.f;           // max |w| over W_hh of both LSTMs and W_ih of LSTM2 (what the decision looked at)
    float lstm_hh_norm = 0.f;        // max abs row sum of W_hh (reported, not decided on: ordinary LSTMs reach ~6)
    const char *precision = "fp16x3";  // "fp16x3" | "fp32-forced" (C3HIP_FP32=1) | "fp32-auto" (this decision) | "fp32-range-guard"

    // ---- switches (README) ----
    bool spp_fused = true;    // PyramidPolling as the epilogue of res3b (c3_conv3.h SPPF; 12 x 5 windows); env C3HIP_SPP_FUSED
    int wino = 2;             // stride-1 convolutions on F(2,3) along H (c3_conv3w.h: 12 instead of 18 piece-product groups per output pair)
                              // where the layer is a plain plane-to-plane one (res2a, res2b, res3a); env C3HIP_WINO: 0 = the direct kernels
                              // everywhere, 1 = only the 64- / 128-channel layers
    bool conv1_fused = true;  // conv1 computed inside res1a / res1b (c3_conv3.h SRC8): no conv1 launch, no conv1 planes; env C3HIP_CONV1_FUSED
    bool half_tiles = true;   // LSTM recurrences on 8-window tiles while 16-window tiles would leave CUs without a workgroup; env C3HIP_HALF_TILES
    int sharing = 1;          // handles the CALLER says feed this GPU side by side (c3_model_set_sharing): beside other batches the chip is
                              // full, so the recurrences stay on full tiles and the projection launches half as many, twice as long workgroups
    // a batch of the ring that runs in a lane NEXT TO another batch of the same handle (c3_hostring.h predict_submit) is in the same position
    // as one beside another handle: its recurrences take full 16-window tiles (128 workgroups per 1024 windows, so that two batches fill the 256
    // CUs between them -- a half-tile launch alone owns every CU: 144 KB of LDS per workgroup, and the second lane's batch waits), the
    // LSTM2 projection half its grid.  Set around forward_device by the ring; 1 everywhere else.  C3HIP_LANE_SHARING=0: never.
    int lane_sharing = 1;
    bool lane_sharing_ok = true;
    unsigned lane_next = 0;     // the lane of the ring's next small batch (round robin over the submits)
    int stream_priority = 0;    // env C3HIP_STREAM_PRIORITY=-1/0/1: the priority every stream of this handle is created with (new_stream below)
    // As few streams as the work needs.  The runtime gives a process FOUR hardware queues (GPU_MAX_HW_QUEUES) and places every stream on the
    // least-used one; two streams on one queue run in submission order, and a wait between streams on two queues costs tens of microseconds.
    // Which streams meet on a queue depends on everything else the process created before -- measured (profiles/r06_o_*): the same ring of
    // 256-window batches ran at 650 k windows/s on the first handle of a fresh process and at 800 k on a second one, 620 k with 8 or 16 queues
    // (every stream alone: every dependency crosses queues).  So a small batch of the ring lives on its lane's stream ALONE -- staged windows in,
    // kernels, FC chain, rows out, in order, no event (lane_h2d; the batches of the other lanes are what its copy runs under) -- and the transfer
    // stream exists only from the first large batch on (lazy_h2d): three lanes + the null stream are the four queues.
    bool lazy_h2d = true;       // env C3HIP_LAZY_H2D_STREAM=0: the transfer stream is created with the handle
    bool lane_h2d = true;       // env C3HIP_LANE_H2D=0: a lane batch's staged windows travel on the transfer stream, an event in between
    bool lane_by_slot = false;  // env C3HIP_LANE_ORDER=slot: lane = slot % lanes (round 6's first form)
    int host_copy_kernel = 1;  // env C3HIP_HOST_COPY_KERNEL=0: every batch through the DMA engines on the transfer streams
    bool tail_fused = false;  // the split-K sum of L4 inside fc_tail_mfma_kernel (c3_tail.h) instead of its own launch: on for the pileup network (+0.7 %:
                              // 15 partials of 128 features), off for full alignment (-1 %: four branch workgroups re-read 28 partials of 256); env C3HIP_TAIL_FUSED
    int duo = 0;              // a micro-batch as two halves on two streams inside one call (c3_forward.h forward_device); env C3HIP_DUO
    int wg_slots = 512;       // co-resident 256-thread / 64 KiB-LDS workgroups on the device (2 per CU)

    void *decode_dev = nullptr;  // scratch of c3_outcome_maxima
    size_t decode_bytes = 0;

    // ---- workspace ----
    int64_t cap = 0;    // windows per micro-batch the workspace can hold
    bool keep = false;  // debug: one buffer per layer instead of the 3-buffer rotation
    bool last_planes = false;  // the last forward pass left plane activations in act[] / h1 (c3_debug_fetch converts)
    std::vector<DevBuf> bufs;
    float *act[9] = {};
    float *spp = nullptr, *part = nullptr, *l4dbg = nullptr;
    float *h1 = nullptr, *gx2 = nullptr, *h2 = nullptr;
    int64_t last_n = 0;  // windows of the last micro-batch (for debug fetch)

    HostSlot slot[kHostSlots];

    // ---- the ring's second lane (round 6) ----
    // Batches of the submit / wait ring used to run strictly one after the other: ONE workspace and ONE kernel stream per handle.  A lane is
    // everything a forward pass writes -- the workspace, the kernel stream, the tail stream and its events: with two of them consecutive
    // small batches (dealt to the lanes in submit order, c3_hostring.h) overlap on the chip (what three HANDLES in flight do, 876 k against 735 k windows/s at
    // B = 256, without a second copy of the weights) and fill each other's under-filled launches.  The fields above ARE the
    // active lane; use_lane() parks them and takes another lane's out of `parked`.  Rows do not depend on the lane (same kernels, same data).
    // env C3HIP_RING_LANES=1: one lane.
    struct Lane {
        int64_t cap = 0;
        bool last_planes = false, tail_pending = false;
        std::vector<DevBuf> bufs;
        float *act[9] = {};
        float *spp = nullptr, *part = nullptr, *l4dbg = nullptr, *h1 = nullptr, *gx2 = nullptr, *h2 = nullptr;
        int64_t last_n = 0;
        hipStream_t stream = nullptr, tail_stream = nullptr;
        hipEvent_t ev_body_done = nullptr, ev_tail_done = nullptr;
    };
    static constexpr int kMaxLanes = 3;  // = the batches a worker keeps in flight (ring of three slots)
    Lane parked[kMaxLanes];  // the lanes that are NOT active live here (parked[lane_cur] is stale: its contents ARE the fields above)
    int lane_cur = 0;  // which lane the fields above hold
    int ring_lanes = 1;  // 1 .. kMaxLanes (set in c3_model_create: the kind's default, or env)
    int64_t lane_max_batch = 0;  // batches up to this many windows take the lane of their slot, larger ones the first lane (env C3HIP_RING_LANES_MAX_BATCH)

    // which kernel forms the last forward pass took (c3_model_describe; bench.py reports it)
    const char *choice_lstm1 = "-", *choice_proj2 = "-", *choice_lstm2 = "-", *choice_fa = "-";
    const char *choice_s2[2] = {"-", "-"};  // conv3, conv5: one or two workgroups per CU (c3_conv3s2.h PAIR)
    char choice_s1[8] = "------";           // the six stride-1 convolutions res1a .. res3b: d = direct, w = F(2,3) along H

    bool prof = false;
    std::vector<ProfRec> recs;
};

static int conv_out(int n, int s) { return (n - 1) / s + 1; }

// Every stream of a handle: non-blocking, at the handle's priority.  The runtime keeps one pool of at most GPU_MAX_HW_QUEUES (4) hardware queues
// per priority and hands a new stream the least-used queue of its pool; two streams on one hardware queue run in submission order.
static hipError_t new_stream(const c3_model *m, hipStream_t *s) {
    return m->stream_priority ? hipStreamCreateWithPriority(s, hipStreamNonBlocking, m->stream_priority) : hipStreamCreateWithFlags(s, hipStreamNonBlocking);
}


// make lane k the active one (c3_model::Lane): park everything a forward pass writes and take lane k's out of the parking lot
static void lane_exchange(c3_model *m, c3_model::Lane &o) {
    std::swap(m->cap, o.cap), std::swap(m->last_planes, o.last_planes), std::swap(m->tail_pending, o.tail_pending);
    m->bufs.swap(o.bufs);
    for (int l = 0; l < 9; ++l) std::swap(m->act[l], o.act[l]);
    std::swap(m->spp, o.spp), std::swap(m->part, o.part), std::swap(m->l4dbg, o.l4dbg);
    std::swap(m->h1, o.h1), std::swap(m->gx2, o.gx2), std::swap(m->h2, o.h2), std::swap(m->last_n, o.last_n);
    std::swap(m->stream, o.stream), std::swap(m->tail_stream, o.tail_stream);
    std::swap(m->ev_body_done, o.ev_body_done), std::swap(m->ev_tail_done, o.ev_tail_done);
}
static int use_lane(c3_model *m, int k) {
    if (k == m->lane_cur) return 0;
    if (k < 0 || k >= c3_model::kMaxLanes) return fail("lane %d out of range", k);
    lane_exchange(m, m->parked[m->lane_cur]);  // the active fields -> their parking place (which held nothing that matters)
    lane_exchange(m, m->parked[k]);            // lane k's -> the active fields
    m->lane_cur = k;
    if (!m->stream) HIP_TRY(new_stream(m, &m->stream));  // (a further lane's kernel stream, on first use)
    return 0;
}

static void fa_geometry(const c3_model *m, int hh[10], int ww[10]) {
    hh[0] = m->depth, ww[0] = m->positions;
    for (int l = 0; l < 9; ++l) hh[l + 1] = conv_out(hh[l], kConvStride[l]), ww[l + 1] = conv_out(ww[l], kConvStride[l]);
}

// ------------------------------------------------------------------------------------------ profiling scope
// dense MFMA peaks of MI355X (MI355X_MICROARCH.md): v_mfma_f32_32x32x16_f16 / 16x16x32_f16 and the fp32-input forms
static constexpr double kPeakF16 = 2500.0, kPeakF32 = 157.3;
struct ProfScope {
    c3_model *m;
    hipStream_t s;
    ProfRec r;
    bool on;
    ProfScope(c3_model *m_, hipStream_t s_, const char *name, double flops, double bytes) : m(m_), s(s_), on(m_->prof) {
        if (!on) return;
        r.name = name, r.flops = flops, r.bytes = bytes;
        (void)hipEventCreate(&r.a);
        (void)hipEventCreate(&r.b);
        (void)hipEventRecord(r.a, s);
    }
    // executed matrix work of the launch and the roof of the instruction it uses (c3_kernel_stat.mfma_flops / mfma_peak_tflops)
    void mfma(double flops, bool f16) { r.mfma_flops = flops, r.mfma_peak = f16 ? kPeakF16 : kPeakF32; }
    ~ProfScope() {
        if (!on) return;
        (void)hipEventRecord(r.b, s);
        m->recs.push_back(r);
    }
};

// ------------------------------------------------------------------------------------------ launches
template <class Loader, int EPI, int BM, int BN, int SPLIT = 0>
static int launch_gemm(hipStream_t s, const typename Loader::Params &lp, const float *bt, int64_t ldb, int M, int N,
                       int nk, int splits, const EpilogueParams &ep, const float *bt16 = nullptr) {
    if (N % BN) return fail("internal: N=%d not a multiple of BN=%d", N, BN);
    if (M <= 0) return 0;
    GemmParams gp;
    gp.bt = bt, gp.ldb = ldb, gp.M = M, gp.N = N, gp.nk = nk;
    gp.bt3 = reinterpret_cast<const uint16_t *>(bt16);
    gp.tiles_n = N / BN;
    gp.tiles = ((M + BM - 1) / BM) * gp.tiles_n;
    dim3 grid(gp.tiles, splits);
    hipLaunchKernelGGL((gemm_mfma_kernel<Loader, EPI, BM, BN, SPLIT>), grid, dim3(kThreads), 0, s, lp, gp, ep);
    HIP_TRY(hipGetLastError());
    return 0;
}

// Split-K factor of the L4 GEMM (K = 10560 / 3584, N = 128 / 256: far too few output tiles to fill 256 CUs).
// It is a constant of the model, NOT a function of the batch size: the partial sums are added in a fixed
// order by the reduce kernel, so a window's probabilities are bit-identical whatever batch it travels in.
static int l4_splits(const c3_model *m) {
    const int nk = m->K4 % 64 == 0 ? m->K4 / 64 : m->K4 / kBK;  // l4_stream_kernel (c3_l4.h) walks chunks of 64 inputs, the fp32 form chunks of kBK
    static const int env = getenv("C3HIP_L4_SPLITS") ? atoi(getenv("C3HIP_L4_SPLITS")) : 0;  // A/B knob
    const int want = env > 0 ? env : m->kind == C3_KIND_PILEUP ? 15 : 28;  // measured against 22 / 30 / 33 (pileup) and 14 / 56 (full alignment)
    int best = 1;
    for (int s = 1; s <= nk && s <= want; ++s)
        if (nk % s == 0) best = s;
    return best;
}

// ------------------------------------------------------------------------------------------ memory
static int dev_alloc(c3_model *m, void **p, size_t bytes) {
    DevBuf b;
    b.bytes = bytes;
    HIP_TRY(hipMalloc(&b.p, std::max<size_t>(bytes, 256)));
    m->bufs.push_back(b);
    *p = b.p;
    return 0;
}
// ---- PAGEABLE host memory never meets the device directly.  A hipMemcpy from (or to) pageable memory makes the runtime pin the
// caller's pages, and it keeps them registered with the device for as long as the range stays mapped (measured, tools/
// fork_stall_probe.hip: after a 256 MB pageable upload the first kernel behind a fork() completes 3.3 s late -- fork write-protects
// the registered pages, the driver evicts the process's queues and revalidates every registered page; with the source freed, or with
// no pageable copy at all, nothing happens).  The reference's stage-B loop forks its decode pool right after its first model call
// (clair3/CallVariantsFromCffi.py:302): with the weights uploaded from the state dict's pageable arrays that fork cost ~0.3 s of
// every run of the loop (tests/diag/fork_stall.py: first call after the forks 320 ms, then 2.8 ms).  So every such copy goes through
// ONE pinned bounce buffer that forked children do not inherit.
// Handles must not be used in a forked child: their pinned buffers are not there (MADV_DONTFORK) and a touch faults.
static void keep_out_of_children(void *p, size_t bytes) {  // (what ibv_fork_init does for RDMA buffers; a child could not use the handle anyway)
    if (!p || !bytes) return;
    // madvise works on whole pages: every caller allocates page-multiples with hipHostMalloc (page-aligned, page-exclusive); anything
    // else would take a neighbour's bytes out of the children too, so it is refused here and said once
    static std::atomic<bool> said{false};
    if (((uintptr_t)p & 4095) != 0 || (bytes & 4095) != 0) {
        if (!said.exchange(true)) fprintf(stderr, "libc3hip: a pinned buffer of %zu bytes at %p is not page-granular: left visible to forked children\n", bytes, p);
        return;
    }
    if (madvise(p, bytes, MADV_DONTFORK) != 0 && !said.exchange(true))
        fprintf(stderr, "libc3hip: madvise(MADV_DONTFORK) failed (%s): pinned staging memory stays visible to forked children (a fork then stalls the device longer)\n", strerror(errno));
}
// One bounce buffer PER DEVICE, each with its own lock (round 5 had one for the process: weight uploads, c3_outcome_maxima and
// c3_decode_columns of handles on different GPUs serialised on it), portable (usable by every device's copies whatever device was
// current when it was made), in two halves: the host memcpy of chunk i + 1 runs under the DMA of chunk i.
struct BounceBuf {
    static constexpr size_t kBytes = (size_t)8 << 20, kHalf = kBytes / 2;
    std::mutex mu;
    void *pin = nullptr;
    hipEvent_t ev[2] = {nullptr, nullptr};
};
static BounceBuf &bounce_buf() {
    constexpr int kMaxDev = 64;
    static BounceBuf *b = new BounceBuf[kMaxDev];  // never destroyed (handles may outlive static destruction at exit)
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= kMaxDev) d = 0;
    return b[d];
}
static int bounce_ready(BounceBuf &b) {
    if (!b.pin) {
        HIP_TRY(hipHostMalloc(&b.pin, BounceBuf::kBytes, hipHostMallocPortable));
        keep_out_of_children(b.pin, BounceBuf::kBytes);
        for (hipEvent_t &e : b.ev) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    return 0;
}
// dev[0, bytes) = src[0, bytes): synchronous (src may be reused on return); s orders the copy behind the stream's work.  The device
// whose buffer is used is the CURRENT one: every caller has made the handle's device current (hipSetDevice) before it gets here.
static int h2d_staged(void *dev, const void *src, size_t bytes, hipStream_t s = nullptr) {
    BounceBuf &b = bounce_buf();
    std::lock_guard<std::mutex> lk(b.mu);
    TRY(bounce_ready(b));
    int i = 0;
    for (size_t off = 0; off < bytes; off += BounceBuf::kHalf, ++i) {
        const size_t n = std::min(BounceBuf::kHalf, bytes - off);
        char *half = (char *)b.pin + (i & 1) * BounceBuf::kHalf;
        if (i >= 2) HIP_TRY(hipEventSynchronize(b.ev[i & 1]));  // the DMA that read this half two chunks ago
        memcpy(half, (const char *)src + off, n);
        HIP_TRY(hipMemcpyAsync((char *)dev + off, half, n, hipMemcpyHostToDevice, s));
        HIP_TRY(hipEventRecord(b.ev[i & 1], s));
    }
    HIP_TRY(hipStreamSynchronize(s));
    return 0;
}
// dst[0, bytes) = dev[0, bytes) behind the stream's work: synchronous
static int d2h_staged(void *dst, const void *dev, size_t bytes, hipStream_t s = nullptr) {
    BounceBuf &b = bounce_buf();
    std::lock_guard<std::mutex> lk(b.mu);
    TRY(bounce_ready(b));
    int i = 0;
    size_t prev_off = 0, prev_n = 0;
    for (size_t off = 0; off < bytes; off += BounceBuf::kHalf, ++i) {
        const size_t n = std::min(BounceBuf::kHalf, bytes - off);
        char *half = (char *)b.pin + (i & 1) * BounceBuf::kHalf;
        HIP_TRY(hipMemcpyAsync(half, (const char *)dev + off, n, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipEventRecord(b.ev[i & 1], s));
        if (i >= 1) {  // chunk i - 1 leaves the other half while chunk i arrives
            HIP_TRY(hipEventSynchronize(b.ev[(i - 1) & 1]));
            memcpy((char *)dst + prev_off, (char *)b.pin + ((i - 1) & 1) * BounceBuf::kHalf, prev_n);
        }
        prev_off = off, prev_n = n;
    }
    if (i >= 1) {
        HIP_TRY(hipEventSynchronize(b.ev[(i - 1) & 1]));
        memcpy((char *)dst + prev_off, (char *)b.pin + ((i - 1) & 1) * BounceBuf::kHalf, prev_n);
    }
    return 0;
}

static int upload(c3_model *m, float **dst, const std::vector<float> &src) {
    void *p = nullptr;
    HIP_TRY(hipMalloc(&p, std::max<size_t>(src.size() * sizeof(float), 256)));
    TRY(h2d_staged(p, src.data(), src.size() * sizeof(float)));
    if (*dst) (void)hipFree(*dst);
    *dst = (float *)p;
    (void)m;
    return 0;
}

static void free_workspace(c3_model *m) {  // the active lane's
    for (auto &b : m->bufs) (void)hipFree(b.p);
    m->bufs.clear();
    m->cap = 0;
}
static void free_all_workspaces(c3_model *m) {  // both lanes' (geometry change, destruction)
    free_workspace(m);
    for (int k = 0; k < c3_model::kMaxLanes; ++k) {
        if (k == m->lane_cur) continue;
        for (auto &b : m->parked[k].bufs) (void)hipFree(b.p);
        m->parked[k].bufs.clear();
        m->parked[k].cap = 0;
    }
}

static int64_t max_microbatch(const c3_model *m) { return m->kind == C3_KIND_PILEUP ? 16384 : 2048; }

static int ensure_workspace(c3_model *m, int64_t n) {
    n = std::min<int64_t>(n, max_microbatch(m));
    if (n <= m->cap) return 0;
    HIP_TRY(hipDeviceSynchronize());
    free_workspace(m);
    if (m->kind == C3_KIND_FULL_ALIGNMENT) {
        int hh[10], ww[10];
        fa_geometry(m, hh, ww);
        size_t act_elems[9];
        size_t biggest = 0;
        for (int l = 0; l < 9; ++l) {
            act_elems[l] = (size_t)hh[l + 1] * ww[l + 1] * kConvCout[l];
            biggest = std::max(biggest, act_elems[l]);
        }
        if (m->keep) {
            for (int l = 0; l < 9; ++l) TRY(dev_alloc(m, (void **)&m->act[l], act_elems[l] * n * sizeof(float)));
        } else {
            float *rot[3];
            for (int i = 0; i < 3; ++i) TRY(dev_alloc(m, (void **)&rot[i], biggest * n * sizeof(float)));
            for (int l = 0; l < 9; ++l) m->act[l] = rot[l % 3];
        }
        TRY(dev_alloc(m, (void **)&m->spp, (size_t)n * m->K4 * sizeof(float)));
    } else {
        const int T = m->positions;
        TRY(dev_alloc(m, (void **)&m->h1, (size_t)n * T * 256 * sizeof(float)));
        TRY(dev_alloc(m, (void **)&m->gx2, (size_t)n * T * 1280 * sizeof(float)));
        TRY(dev_alloc(m, (void **)&m->h2, (size_t)n * T * 320 * sizeof(float)));
    }
    TRY(dev_alloc(m, (void **)&m->part, (size_t)l4_splits(m) * n * m->FC * sizeof(float)));  // [S][n][FC]
    TRY(dev_alloc(m, (void **)&m->l4dbg, (size_t)n * m->FC * sizeof(float)));
    m->cap = n;
    return 0;
}

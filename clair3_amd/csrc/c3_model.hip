// c3_model.hip -- host side of libc3hip.so: the C ABI of include/c3hip.h, weight packing (BatchNorm folding,
// gate re-ordering, MFMA fragment layouts), workspace management and the launch sequences of the two
// forward passes (clair3/model.py:130-161 and :377-416).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <map>
#include <chrono>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/c3hip.h"
#include "c3_gemm.h"
#include "c3_kernels.h"
#include "c3_wino.h"
#include "c3_wino_p.h"
#include "c3_conv1.h"
#include "c3_tail.h"
#include "c3_proj.h"
#include "c3_decode.h"
#include "c3_lstm_fused.h"
#include "c3_host.h"
#include "c3_conv3.h"
#include "c3_comm.h"
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

// ------------------------------------------------------------------------------------------ host staging
// The caller's windows are pageable numpy memory (clair3/CallVariantsFromCffi.py:112-133: np.load slices); they go
// through a pinned buffer, cut into pieces: the H2D transfer of a piece is queued as soon as it is staged, so the DMA of
// piece i runs under the memcpy of piece i + 1, and every piece's memcpy is split over the staging pool (c3_host.h).
// Buffers the caller has registered (c3_host_register: page-locked for the device) skip the staging copy altogether.
struct HostRange {
    const char *p;
    size_t n;
};
static std::vector<HostRange> g_registered;
static std::mutex g_registered_mu;
static bool is_registered(const void *p, size_t n) {
    std::lock_guard<std::mutex> lk(g_registered_mu);
    for (const HostRange &r : g_registered)
        if ((const char *)p >= r.p && (const char *)p + n <= r.p + r.n) return true;
    return false;
}

// Small batches (the pileup network: 594 B per window in, 96 B out) cross PCIe inside the COMPUTE stream instead: a copy kernel
// reads the pinned staging buffer / writes the pinned result buffer directly (both are device-mapped), so a batch is ONE chain
// of launches on one queue -- no copy engine, no cross-queue event waits, whose barrier packets cost a 210 us pileup batch
// ~60 us of idle GPU per batch.  The transfer is then serial with the kernels, which is only worth it while it is
// short: up to kKernelCopyMax bytes (~15 us at PCIe Gen5 rates); full-alignment batches (23.5 MB) keep the DMA engines.
constexpr size_t kKernelCopyMax = (size_t)2 << 20;
__global__ __launch_bounds__(256) void host_copy_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16,
                                                       const uint32_t *flag_src, uint32_t *flag_dst) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
    if (flag_dst && blockIdx.x == 0 && threadIdx.x == 0) *flag_dst = *flag_src;
}

// stage [src, src + bytes) through `pin` into `dev` on stream s, piecewise
static int stage_h2d(void *dev, void *pin, const void *src, size_t bytes, hipStream_t s, bool src_locked = false) {
    if (src_locked || is_registered(src, bytes)) {  // zero-copy: the DMA engine reads the caller's pages
        HIP_TRY(hipMemcpyAsync(dev, src, bytes, hipMemcpyHostToDevice, s));
        return 0;
    }
    // >= 4 MiB and at most four pieces: every queued transfer costs ~15 us of host time (2 MiB x 8 was slower again)
    const size_t piece = std::max<size_t>((size_t)4 << 20, ((bytes / 4) + 4095) & ~(size_t)4095);
    for (size_t off = 0; off < bytes; off += piece) {
        const size_t n = std::min(piece, bytes - off);
        StagePool::get().copy((char *)pin + off, (const char *)src + off, n);
        HIP_TRY(hipMemcpyAsync((char *)dev + off, (char *)pin + off, n, hipMemcpyHostToDevice, s));
    }
    return 0;
}

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
    double mfma_flops = 0.0;  // FLOP the matrix instructions of the launch EXECUTE (tile padding, piece products, Winograd reduction included)
    double mfma_peak = 0.0;   // dense peak (TFLOP/s) of the matrix instruction the launch issues: 2500 (16-bit) or 157.3 (fp32)
};

struct HostSlot {
    void *pin_x = nullptr;
    float *pin_y = nullptr;
    void *dev_x = nullptr;
    float *dev_y = nullptr;
    size_t cap_x = 0, cap_y = 0;
    hipEvent_t ev_h2d = nullptr, ev_compute = nullptr, ev_out = nullptr;
    float *y_host = nullptr;
    size_t y_bytes = 0;
    int64_t batch = 0;  // what is in flight (for the fp32 re-run of c3_predict_wait)
    uint32_t *pin_flag = nullptr;  // pinned copy of the model's range_flag after this batch
    int x_dtype = 0;
    bool busy = false;
    bool used_f16 = false;  // the batch in flight was computed by the fp16x3 kernels (c3_predict_wait then checks its range)
};

constexpr int kHostSlots = 4;  // batches in flight per handle through c3_predict_submit / _wait (C3_HOST_SLOTS)

struct c3_model {
    int kind = 0, C = 0, add_indel = 0, device = 0;
    int depth = 89, positions = 33;
    int nb = 2, nout = 24;
    int row = 24;  // floats per output row: nout, + kDecodeCols when c3_model_set_decode_columns is on
    bool loaded = false;
    hipStream_t stream = nullptr, h2d_stream = nullptr, d2h_stream = nullptr;

    // ---- packed weights (device) ----
    // pileup
    float *proj_w[2] = {nullptr, nullptr};  // [2*4H][Kp]
    float *proj2_frag = nullptr;            // LSTM2 projection weights as proj_stream_kernel fragments (c3_proj.h)
    bool proj2_stream = true;               // env C3HIP_PROJ2_STREAM
    float *proj_b[2] = {nullptr, nullptr};  // [2*4H]
    float *whh[2] = {nullptr, nullptr};     // fragment-packed W_hh
    float *whh16[2] = {nullptr, nullptr};   // W_hh as two fp16 pieces in the F16 kernels' fragment order (c3_kernels.h)
    bool lstm2_f16 = true;                  // LSTM2 recurrence on fp16x3 split products; env C3HIP_LSTM2_F16
    bool lstm1_f16 = true;                  // LSTM1 recurrence likewise; env C3HIP_LSTM1_F16
    float *l1_wih = nullptr, *l1_bias = nullptr;  // LSTM1 input projection as MFMA fragments (fused kernel)
    float *l1_wih16 = nullptr;                    // the same as two fp16 pieces of 128 W_ih for the F16 kernel (int8 windows)
    bool lstm1_fused = true;                // env C3HIP_LSTM1_FUSED=0 selects GEMM + recurrence
    int lstm_opt = 1;                       // env C3HIP_LSTM_OPT: bit 0 = LSTM1 (int8 windows, h1 as planes) widens the counts of step t + 1 at the
                                            // top of step t + 1 and stores its plane piece unconditionally (c3_lstm_fused.h OPT 3); bit 2 = half
                                            // tiles (8 windows per workgroup) while full tiles leave CUs idle: LSTM1 57 -> 52 us and +2 % for ONE
                                            // batch in flight, -5 % with three (twice the matrix work on a chip the others already fill): off
    bool concurrent = false;                // another handle of the process queued a forward pass in the last 2 ms (others_active)
    int lstm2_half = 1;                     // env C3HIP_LSTM2_HALF=0: LSTM2 never takes half tiles (C3HIP_LSTM_OPT bit 4 pins them on)
    int adaptive = 1;                       // env C3HIP_ADAPTIVE=0: the kernel choices that depend on `concurrent` follow C3HIP_LSTM_OPT / C3HIP_DENSE_MODE alone
    int lstm_trace_left = 0, lstm2_trace_left = 0;  // debug, env C3HIP_LSTM_TRACE=n: the n-th LSTM launches record a phase trace
    unsigned long long *lstm_trace_dev = nullptr;
    // full alignment
    float *conv_w[9] = {};
    float *conv_b[9] = {};
    float *wino_v[9] = {};   // Winograd-domain weights of the stride-1 convs (layers 1,2,4,5,7,8)
    float *wino_v16[9] = {};  // the same as two fp16 pieces per weight, fragment order of the F16 persistent kernel (c3_wino_p.h)
    unsigned wino_f16_mask = 0x1b6;  // Winograd layers on the fp16x3 split products; env C3HIP_WINOGRAD_F16MASK
    float *pconv_w[9] = {};  // stride-1 convs for conv3x3_planes_kernel (c3_conv3.h): [Cout/64][Cin/64][9][64][16 pieces of 16 B];
                             // stride-2 convs for dense_planes_kernel<true> (c3_dense.h): [Cout/128][9 Cin/64][128][16 pieces]
    float pconv_wscale[9] = {1, 1, 1, 1, 1, 1, 1, 1, 1};
    bool conv_s2_planes = true;  // conv3 / conv5 on dense_planes_kernel<true> (c3_dense.h); env C3HIP_CONV_S2_PLANES=0: tiled GEMM with PlaneConvLoader
    bool fa_planes = true;   // plane activations + direct fp16x3 convolutions (c3_conv3.h); env C3HIP_FA_PLANES=0 restores the fp32-activation kernels of round 1
    bool last_planes = false;  // the last full-alignment forward left plane activations in act[] (c3_debug_fetch converts)
    float *conv_w3[9] = {};  // direct-conv weights as three bf16 pieces [3][Cout][K] (uint16 payload), layers in conv_split_mask
    unsigned conv_split_mask = 0x48;  // stride-2 convs conv3 / conv5 on the split-precision path (c3_gemm.h SPLIT); env C3HIP_CONV_SPLITMASK
    // fp16x3: a weight tensor is packed times a power of two chosen per tensor (pick_wscale: as close to 256 as keeps
    // max |w| * scale below 16384, so the low piece is a normal fp16 number and the high piece cannot overflow)
    float conv_wscale[9] = {1, 1, 1, 1, 1, 1, 1, 1, 1}, wino_wscale[9] = {1, 1, 1, 1, 1, 1, 1, 1, 1}, l4_wscale = 1.f, proj2_wscale = 1.f;
    int split_kind = 2;               // 2 = fp16x3 (two fp16 pieces, three products), 1 = bf16x6 (three bf16 pieces, six products); env C3HIP_SPLIT_KIND
    bool use_wino[9] = {};
    // Measured on MI355X (B=256), direct implicit GEMM -> Winograd v1: res1 134/144 -> 105/112 us, res2 163/172 ->
    // 110/113 us, res3 171/179 -> 162/164 us (res3 gains little: 18 tiles per window quantise badly and its input
    // transform is recomputed for each of 8 N-tiles).  v2 (one workgroup per CU, transform interleaved into the MFMA
    // stream) is 10-15 % slower than v1 on res1/res2 and equal on res3; it stays selectable for experiments.
    unsigned conv_bn64_mask = 0x40; // direct-conv layers on 128x64 tiles instead of 128x128 (conv5: 480 workgroups fill 2 per CU); env C3HIP_CONV_BN64MASK
    void *decode_dev = nullptr;     // scratch of c3_outcome_maxima
    size_t decode_bytes = 0;
    bool conv1_direct = true;       // 8-channel conv1 through conv1_i8_kernel (c3_conv1.h); env C3HIP_CONV1_DIRECT
    float *conv1_wfrag = nullptr;   // its resident B fragments [36][2][64]
    uint32_t *range_flag = nullptr; // device word set by the fp16x3 kernels when an activation nears the fp16 range (c3_gemm.h kF16Range)
    uint32_t *pin_flag = nullptr;   // pinned copy of range_flag for c3_predict_device_checked
    bool f16_ok = true;             // cleared by c3_predict_wait when a batch came back non-finite: every layer then runs its fp32-MFMA form
    float *conv1_wfrag16 = nullptr; // conv1_i8_f16_kernel: [5][2][2 pieces][64][8 fp16]
    bool conv1_f16 = true;          // conv1 on fp16 matrix instructions (int8 inputs exact, weights as two pieces); env C3HIP_CONV1_F16
    bool spp_fused = true;          // PyramidPolling as the epilogue of res3b (c3_conv3.h SPPF; 12 x 5 windows); env C3HIP_SPP_FUSED
    bool conv1_fused = true;        // 8-channel conv1 computed inside res1a / res1b (c3_conv3.h SRC8): no conv1 launch, no conv1 planes; env C3HIP_CONV1_FUSED
    unsigned wino_p_mask = 0x1b6;   // layers using the persistent 32x64 kernel (c3_wino_p.h); env C3HIP_WINOGRAD_PMASK
    int wg_slots = 512;             // co-resident 256-thread / 64 KiB-LDS workgroups on the device (2 per CU)
    unsigned wino_n64_mask = 0x1b6; // layers using the 32-tile x 64-cout workgroup shape; env C3HIP_WINOGRAD_N64MASK
    bool lstm2_v2 = true;        // env C3HIP_LSTM2_V2=0 selects the streaming 10-wave kernel
    unsigned wino_mask = 0x1b6;  // layers run as Winograd (bit l): all six stride-1 convs; env C3HIP_WINOGRAD overrides
    // shared FC tail
    float *l4_w = nullptr, *l4_b = nullptr;  // [FC][K4] native layout
    float *l4_w3 = nullptr;                  // the same as three bf16 pieces (SPLIT path); env C3HIP_L4_SPLIT
    bool l4_split = true;
    float *proj2_pw = nullptr;               // LSTM2 projection weights as dense_planes_kernel chunks (c3_dense.h); env C3HIP_PROJ2_PLANES
    float *proj2_pwr = nullptr;              // the same in the register-fragment order of dense_planes_wres_kernel (weights resident; C3HIP_DENSE_MODE=6)
    float *proj2_pw32 = nullptr;             // the same as 32-channel chunks of 256 rows for dense_planes_big_kernel (C3HIP_DENSE_MODE=5)
    float proj2_pwscale = 1.f;
    bool proj2_planes = true;
    int dense_mode = 3;                      // dense kernels (c3_dense.h): 3 = default: dense_planes_pipe_kernel (chunk stream spread over the matrix
                                             // stream) for the stride-2 convolutions and, while `adaptive`, dense_planes_wres_kernel (weights resident
                                             // in registers) for the LSTM2 projection; 6 = that kernel pinned, 5 = 256 x 256 tiles, 4 =
                                             // dense_planes_ws_kernel (8 multiplying + 4 moving waves: the kernels themselves 5-8 % faster, the step
                                             // as a whole 2.5 % slower -- DESIGN.md 3.8), 1 / 0 = round-2 kernel with the direct / staged fp32
                                             // epilogue; env C3HIP_DENSE_MODE (setting it pins the projection to the pipe kernel for 3)
    float *proj2_w3 = nullptr;               // LSTM2 projection weights as bf16 pieces for the tiled SPLIT GEMM; env C3HIP_PROJ2_SPLIT
    bool proj2_split = true;
    float *w5t = nullptr, *b5 = nullptr, *wh = nullptr, *bh = nullptr;
    float *w5f = nullptr, *whf = nullptr, *bh48 = nullptr;  // MFMA fragment packing of the same weights (c3_tail.h)
    bool tail_mfma = true;                                   // env C3HIP_TAIL_MFMA
    float *zeros = nullptr;  // 256-byte zero page: padding taps of the conv loaders read from here
    int FC = 0, K4 = 0;

    // ---- workspace ----
    int64_t cap = 0;  // windows per micro-batch the workspace can hold
    bool keep = false;  // debug: one buffer per layer instead of the 3-buffer rotation
    std::vector<DevBuf> bufs;
    float *act[9] = {};
    float *spp = nullptr, *part = nullptr, *l4dbg = nullptr;
    float *gx1 = nullptr, *h1 = nullptr, *gx2 = nullptr, *h2 = nullptr;
    int splits = 1;
    int64_t last_n = 0;  // windows of the last micro-batch (for debug fetch)

    HostSlot slot[kHostSlots];

    // which of the bit-identical kernel forms the last forward pass took (c3_model_describe; bench.py reports it)
    const char *choice_lstm1 = "-", *choice_proj2 = "-", *choice_lstm2 = "-", *choice_fa = "-";

    int host_copy_kernel = 1;  // env C3HIP_HOST_COPY_KERNEL=0: every batch through the DMA engines on the transfer streams

    bool prof = false;
    std::vector<ProfRec> recs;
};

static int conv_out(int n, int s) { return (n - 1) / s + 1; }

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
                       int nk, int splits, const EpilogueParams &ep, const float *bt3 = nullptr) {
    if (N % BN) return fail("internal: N=%d not a multiple of BN=%d", N, BN);
    if (M <= 0) return 0;
    GemmParams gp;
    gp.bt = bt, gp.ldb = ldb, gp.M = M, gp.N = N, gp.nk = nk;
    gp.bt3 = reinterpret_cast<const uint16_t *>(bt3);
    gp.tiles_n = N / BN;
    gp.tiles = ((M + BM - 1) / BM) * gp.tiles_n;
    dim3 grid(gp.tiles, splits);
    hipLaunchKernelGGL((gemm_mfma_kernel<Loader, EPI, BM, BN, 0, SPLIT>), grid, dim3(kThreads), 0, s, lp, gp, ep);
    HIP_TRY(hipGetLastError());
    return 0;
}

// Split-K factor of the L4 GEMM (K = 10560 / 3584, N = 128 / 256: far too few output tiles to fill 256 CUs).
// It is a constant of the model, NOT a function of the batch size: the partial sums are added in a fixed
// order by the tail kernel, so a window's probabilities are bit-identical whatever batch it travels in.
static int l4_splits(const c3_model *m) {
    const int nk = m->K4 / kBK;
    const int want = m->kind == C3_KIND_PILEUP ? 15 : 28;  // measured against 22 / 30 / 33 (pileup) and 14 / 56 (full alignment), 128 x 128 tiles too
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
static int upload(c3_model *m, float **dst, const std::vector<float> &src) {
    void *p = nullptr;
    HIP_TRY(hipMalloc(&p, std::max<size_t>(src.size() * sizeof(float), 256)));
    HIP_TRY(hipMemcpy(p, src.data(), src.size() * sizeof(float), hipMemcpyHostToDevice));
    if (*dst) (void)hipFree(*dst);
    *dst = (float *)p;
    (void)m;
    return 0;
}

// A weight matrix as 16-bit pieces for the SPLIT paths of gemm_mfma_kernel, layout [pieces][n] (uint16 payload carried
// in a float allocation), every piece rounded to nearest even, the remainders exact in fp32:
//   kind 1: w = p0 + p1 + p2, bf16;   kind 2: w = h0 + h1, fp16 (subnormals kept).
static float pick_wscale(const float *w, size_t n) {
    float mx = 0.f;
    for (size_t i = 0; i < n; ++i) mx = std::max(mx, std::fabs(w[i]));
    float s = 256.f;
    while (s > 1.f / 65536.f && mx * s >= 16384.f) s *= 0.5f;
    return s;
}

static int upload_split_pieces(c3_model *m, float **dst, const std::vector<float> &w, float *scale_out) {
    const float wscale = m->split_kind == 2 ? pick_wscale(w.data(), w.size()) : 1.f;
    *scale_out = wscale;
    auto bf16_rne = [](float f) -> uint16_t {
        uint32_t u;
        memcpy(&u, &f, 4);
        u += 0x7fffu + ((u >> 16) & 1u);
        return (uint16_t)(u >> 16);
    };
    auto bf16_f32 = [](uint16_t h) -> float {
        const uint32_t u = (uint32_t)h << 16;
        float f;
        memcpy(&f, &u, 4);
        return f;
    };
    const size_t n = w.size();
    const int np = m->split_kind == 1 ? 3 : 2;
    std::vector<float> pieces((np * n + 1) / 2);
    uint16_t *q = reinterpret_cast<uint16_t *>(pieces.data());
    for (size_t i = 0; i < n; ++i) {
        float r = w[i];
        for (int lvl = 0; lvl < np; ++lvl) {
            if (m->split_kind == 1) {
                const uint16_t h = bf16_rne(r);
                q[lvl * n + i] = h;
                r -= bf16_f32(h);
            } else {
                if (lvl == 0) r *= wscale;  // exact; undone by post_scale in the kernels' epilogues
                const _Float16 h = (_Float16)r;  // round to nearest even, subnormals kept
                memcpy(&q[lvl * n + i], &h, 2);
                r -= (float)h;
            }
        }
    }
    return upload(m, dst, pieces);
}

// launch the SPLIT instantiation the model was packed for
#define LAUNCH_SPLIT(m, Loader, EPI, BM, BN, ...) \
    ((m)->split_kind == 1 ? launch_gemm<Loader, EPI, BM, BN, 1>(__VA_ARGS__) : launch_gemm<Loader, EPI, BM, BN, 2>(__VA_ARGS__))

static void free_workspace(c3_model *m) {
    for (auto &b : m->bufs) (void)hipFree(b.p);
    m->bufs.clear();
    m->cap = 0;
    m->gx1 = nullptr;
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
        if (!(m->lstm1_fused && m->l1_wih))  // the fused LSTM1 kernel never touches gx1 (2.2 GB at the 16384-window cap)
            TRY(dev_alloc(m, (void **)&m->gx1, (size_t)n * T * 1024 * sizeof(float)));
        TRY(dev_alloc(m, (void **)&m->h1, (size_t)n * T * 256 * sizeof(float)));
        TRY(dev_alloc(m, (void **)&m->gx2, (size_t)n * T * 1280 * sizeof(float)));
        TRY(dev_alloc(m, (void **)&m->h2, (size_t)n * T * 320 * sizeof(float)));
    }
    TRY(dev_alloc(m, (void **)&m->part, (size_t)l4_splits(m) * n * m->FC * sizeof(float)));  // [S][n][FC]
    TRY(dev_alloc(m, (void **)&m->l4dbg, (size_t)n * m->FC * sizeof(float)));
    m->cap = n;
    return 0;
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

static int pack_tail(c3_model *m, const TensorMap &tm) {
    const int FC = m->FC, K4 = m->K4, nb = m->nb;
    const float *w, *b;
    TRY(want(tm, "L4.weight", {FC, K4}, &w));
    TRY(want(tm, "L4.bias", {FC}, &b));
    TRY(upload(m, &m->l4_w, std::vector<float>(w, w + (size_t)FC * K4)));
    if (m->l4_split) TRY(upload_split_pieces(m, &m->l4_w3, std::vector<float>(w, w + (size_t)FC * K4), &m->l4_wscale));
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
    TRY(upload(m, &m->w5t, w5t));
    TRY(upload(m, &m->b5, b5));
    TRY(upload(m, &m->wh, wh));
    TRY(upload(m, &m->bh, bh));
    return 0;
}

// LSTM layer `layer` (0/1): hidden H, input size `in`.
//   proj_w row n = dir*4H + wave*64 + gate*16 + unit  <->  PyTorch gate row gate*H + wave*16 + unit
//   whh fragments: [dir][wave][gate][q][lane][e] = W_hh[gate*H + wave*16 + (lane&15)][16q + 4*(lane>>4) + e]
static int pack_lstm(c3_model *m, const TensorMap &tm, int layer, int H, int in, int Kp, bool v2) {
    const std::string base = layer == 0 ? "LSTM1" : "LSTM2";
    const int NW = H / 16, NQ = H / 16;
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
        if (layer == 1 && m->lstm2_f16 && H % 32 == 0) {
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
        if (layer == 1 && Kp == 256 && (2 * 4 * H) % 32 == 0) {
            // proj_stream_kernel: [cb][i][lane][e] = W[n = 32 cb + (lane&31)][k = 128 (lane>>5) + 4 i + e]
            const int N = 2 * 4 * H;
            std::vector<float> pf((size_t)N * 256);
            for (int cb = 0; cb < N / 32; ++cb)
                for (int i = 0; i < 32; ++i)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 4; ++e)
                            pf[((((size_t)cb * 32 + i) * 64) + lane) * 4 + e] =
                                pw[(size_t)(32 * cb + (lane & 31)) * Kp + 128 * (lane >> 5) + 4 * i + e];
            TRY(upload(m, &m->proj2_frag, pf));
            if (m->proj2_split) TRY(upload_split_pieces(m, &m->proj2_w3, pw, &m->proj2_wscale));
            if (m->proj2_planes && m->split_kind == 2 && N % kDnBN == 0) {
                // dense_planes_kernel: chunk (column tile of 128, k chunk of 64) = 128 rows x 256 B; piece g < 8 = hi of
                // k 64 kc + 8 g .. + 7, g >= 8 = lo of the same k; times a power of two (pick_wscale)
                const float sc = pick_wscale(pw.data(), pw.size());
                m->proj2_pwscale = sc;
                const int NKc = 256 / 64;
                std::vector<float> pk((size_t)N * 256);
                uint16_t *q16 = reinterpret_cast<uint16_t *>(pk.data());
                for (int tn = 0; tn < N / kDnBN; ++tn)
                    for (int kc = 0; kc < NKc; ++kc)
                        for (int r = 0; r < kDnBN; ++r)
                            for (int g = 0; g < 16; ++g)
                                for (int j = 0; j < 8; ++j) {
                                    const float v = pw[(size_t)(tn * kDnBN + r) * Kp + kc * 64 + 8 * (g & 7) + j] * sc;  // exact
                                    const _Float16 h0 = (_Float16)v, h1 = (_Float16)(v - (float)h0);
                                    const _Float16 piece = g < 8 ? h0 : h1;
                                    memcpy(&q16[(((((size_t)tn * NKc + kc) * kDnBN + r) * 16 + g) * 8) + j], &piece, 2);
                                }
                TRY(upload(m, &m->proj2_pw, pk));
                if (N % kBgBN == 0) {
                    // dense_planes_big_kernel: chunk (column tile of 256, k chunk of 32) = 256 rows x 128 B; piece g < 4 = hi of
                    // k 32 kc + 8 g .. + 7, g >= 4 = lo of the same k; the same power of two
                    const int NK32 = 256 / kBgKC;
                    std::vector<float> pb((size_t)N * 256);
                    uint16_t *b16 = reinterpret_cast<uint16_t *>(pb.data());
                    for (int tn = 0; tn < N / kBgBN; ++tn)
                        for (int kc = 0; kc < NK32; ++kc)
                            for (int r = 0; r < kBgBN; ++r)
                                for (int g = 0; g < 8; ++g)
                                    for (int j = 0; j < 8; ++j) {
                                        const float v = pw[(size_t)(tn * kBgBN + r) * Kp + kc * kBgKC + 8 * (g & 3) + j] * sc;  // exact
                                        const _Float16 h0 = (_Float16)v, h1 = (_Float16)(v - (float)h0);
                                        const _Float16 piece = g < 4 ? h0 : h1;
                                        memcpy(&b16[(((((size_t)tn * NK32 + kc) * kBgBN + r) * 8 + g) * 8) + j], &piece, 2);
                                    }
                    TRY(upload(m, &m->proj2_pw32, pb));
                }
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
                                        const float v = pw[(size_t)(tn * kWrBN + 32 * w + (lane & 31)) * Kp + 16 * ks + 8 * (lane >> 5) + j] * sc;  // exact
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
    std::vector<float> pw((size_t)2 * 4 * H * Kp, 0.f), pb((size_t)2 * 4 * H), wf((size_t)2 * 4 * H * H);
    for (int dir = 0; dir < 2; ++dir) {
        const std::string sfx = dir ? "_reverse" : "";
        const float *wih, *whh, *bih, *bhh;
        TRY(want(tm, base + ".weight_ih_l0" + sfx, {4 * H, in}, &wih));
        TRY(want(tm, base + ".weight_hh_l0" + sfx, {4 * H, H}, &whh));
        TRY(want(tm, base + ".bias_ih_l0" + sfx, {4 * H}, &bih));
        TRY(want(tm, base + ".bias_hh_l0" + sfx, {4 * H}, &bhh));
        for (int w = 0; w < NW; ++w)
            for (int g = 0; g < 4; ++g)
                for (int u = 0; u < 16; ++u) {
                    const int r = g * H + w * 16 + u;
                    const size_t n = (size_t)dir * 4 * H + w * 64 + g * 16 + u;
                    pb[n] = (float)((double)bih[r] + (double)bhh[r]);
                    for (int k = 0; k < in; ++k) pw[n * Kp + k] = wih[(size_t)r * in + k];
                }
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
    TRY(upload(m, &m->proj_w[layer], pw));
    TRY(upload(m, &m->proj_b[layer], pb));
    TRY(upload(m, &m->whh[layer], wf));
    if (layer == 0 && m->lstm1_f16 && H % 32 == 0) {
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
    if (layer == 0 && in <= 4 * kFusedKS) {
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
        if (m->lstm1_f16 && in <= 32 && in % 2 == 0) {
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
static int pack_conv(c3_model *m, const TensorMap &tm, int l, int Cin) {
    const int Cout = kConvCout[l];
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
        const double scale = (double)g[co] / std::sqrt((double)var[co] + 1e-3);
        pb[co] = (float)(((double)b[co] - (double)mean[co]) * scale + (double)beta[co]);
        for (int ci = 0; ci < Cin; ++ci)
            for (int kh = 0; kh < 3; ++kh)
                for (int kw = 0; kw < 3; ++kw) {
                    const double v = (double)w[(((size_t)co * Cin + ci) * 3 + kh) * 3 + kw] * scale;
                    if (l == 0)
                        pw[(size_t)co * ldb + kh * 32 + kw * Cin + ci] = (float)(v / 100.0);
                    else
                        pw[(size_t)co * ldb + (size_t)(kh * 3 + kw) * Cin + ci] = (float)v;
                }
    }
    TRY(upload(m, &m->conv_w[l], pw));
    TRY(upload(m, &m->conv_b[l], pb));
    if ((l > 0 && (m->conv_split_mask & (1u << l))) || (l == 0 && Cin != 8 && m->conv1_f16))  // (the 9-channel conv1 runs on the tiled GEMM)
        TRY(upload_split_pieces(m, &m->conv_w3[l], pw, &m->conv_wscale[l]));
    if (l == 0 && Cin == 8) {
        // conv1_i8_kernel: k-step s = 4 tap + j of lane (n = lane & 31, kk = lane >> 5) multiplies channel 4 kk + j of tap s / 4
        std::vector<float> pf((size_t)36 * 2 * 64);
        for (int s = 0; s < 36; ++s)
            for (int cb = 0; cb < 2; ++cb)
                for (int lane = 0; lane < 64; ++lane) {
                    const int co = 32 * cb + (lane & 31), ci = 4 * (lane >> 5) + s % 4, kh = (s / 4) / 3, kw = (s / 4) % 3;
                    const double scale = (double)g[co] / std::sqrt((double)var[co] + 1e-3);
                    pf[((size_t)s * 2 + cb) * 64 + lane] = (float)((double)w[(((size_t)co * Cin + ci) * 3 + kh) * 3 + kw] * scale / 100.0);
                }
        TRY(upload(m, &m->conv1_wfrag, pf));
        if (m->conv1_f16) {
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
                                const double scale = (double)g[co] / std::sqrt((double)var[co] + 1e-3);
                                v = (float)((double)w[(((size_t)co * Cin + j) * 3 + tap / 3) * 3 + tap % 3] * scale * (128.0 / 100.0));  // the kernel feeds x / 128
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
    if (l == 0 && Cin == 9 && m->conv1_f16) {
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
                            const double scale = (double)g[co] / std::sqrt((double)var[co] + 1e-3);
                            v = (float)((double)w[(((size_t)co * Cin + q % 9) * 3 + ky) * 3 + q / 9] * scale * (128.0 / 100.0));
                        }
                        const _Float16 h0 = (_Float16)v, h1 = (_Float16)(v - (float)h0);
                        memcpy(&q16[((((size_t)t * 2 + cb) * 2 + 0) * 64 + lane) * 8 + j], &h0, 2);
                        memcpy(&q16[((((size_t)t * 2 + cb) * 2 + 1) * 64 + lane) * 8 + j], &h1, 2);
                        mx = std::max(mx, std::fabs(v));
                    }
        if (mx < 16384.f) TRY(upload(m, &m->conv1_wfrag16, pf16));  // else: conv1 stays on the tiled GEMM
    }
    if (kConvStride[l] == 1 && Cin == Cout && Cin % 64 == 0 && m->fa_planes && m->split_kind == 2) {
        // conv3x3_planes_kernel: chunk (column tile tn, input slab, tap) = 64 couts x 256 B; piece g < 8 = hi of channels
        // 64 slab + 8 g .. + 7, g >= 8 = lo of channels 8 (g - 8) ..; times a power of two (pick_wscale), undone by post_scale
        const int NS = Cin / 64;
        const float sc = pick_wscale(pw.data(), pw.size());
        m->pconv_wscale[l] = sc;
        std::vector<float> pk((size_t)NS * NS * 9 * 64 * 64);  // 16 KB per chunk
        uint16_t *q16 = reinterpret_cast<uint16_t *>(pk.data());
        for (int tn = 0; tn < NS; ++tn)
            for (int slab = 0; slab < NS; ++slab)
                for (int tap = 0; tap < 9; ++tap)
                    for (int n = 0; n < 64; ++n)
                        for (int g = 0; g < 16; ++g)
                            for (int j = 0; j < 8; ++j) {
                                const int co = tn * 64 + n, ci = slab * 64 + 8 * (g & 7) + j;
                                const float v = pw[(size_t)co * ldb + (size_t)tap * Cin + ci] * sc;  // exact
                                const _Float16 h0 = (_Float16)v, h1 = (_Float16)(v - (float)h0);
                                const _Float16 piece = g < 8 ? h0 : h1;
                                memcpy(&q16[((((((size_t)tn * NS + slab) * 9 + tap) * 64 + n) * 16 + g) * 8) + j], &piece, 2);
                            }
        TRY(upload(m, &m->pconv_w[l], pk));
    }
    if (kConvStride[l] == 2 && l > 0 && Cin % 64 == 0 && Cout % kDnBN == 0 && m->fa_planes && m->conv_s2_planes && m->split_kind == 2) {
        // dense_planes_kernel<true>: chunk (column tile of 128, kc = tap * Cin/64 + slab) = 128 couts x 256 B, pieces as above
        const int NS = Cin / 64, NKc = 9 * NS;
        const float sc = pick_wscale(pw.data(), pw.size());
        m->pconv_wscale[l] = sc;
        std::vector<float> pk((size_t)Cout * NKc * 64);
        uint16_t *q16 = reinterpret_cast<uint16_t *>(pk.data());
        for (int tn = 0; tn < Cout / kDnBN; ++tn)
            for (int kc = 0; kc < NKc; ++kc)
                for (int r = 0; r < kDnBN; ++r)
                    for (int g = 0; g < 16; ++g)
                        for (int j = 0; j < 8; ++j) {
                            const int tap = kc / NS, slab = kc % NS;
                            const float v = pw[(size_t)(tn * kDnBN + r) * ldb + (size_t)tap * Cin + slab * 64 + 8 * (g & 7) + j] * sc;  // exact
                            const _Float16 h0 = (_Float16)v, h1 = (_Float16)(v - (float)h0);
                            const _Float16 piece = g < 8 ? h0 : h1;
                            memcpy(&q16[(((((size_t)tn * NKc + kc) * kDnBN + r) * 16 + g) * 8) + j], &piece, 2);
                        }
        TRY(upload(m, &m->pconv_w[l], pk));
    }
    if (kConvStride[l] == 1 && Cin % kWinoBK == 0 && Cout % kWinoNT == 0) {
        // Winograd F(2x2,3x3) weights V = G g' G^T (g' = BN-folded), in MFMA B-fragment order
        //   [Cout/32][xi = 4i+j][Cin/16][g][lane][e] = V_xi[n = nt*32 + (lane&31)][k = 16c + 8g + 4(lane>>5) + e]
        static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
        const int nch = Cin / kWinoBK;
        std::vector<float> pv((size_t)Cout * 16 * Cin);
        for (int n = 0; n < Cout; ++n) {
            const double scale = (double)g[n] / std::sqrt((double)var[n] + 1e-3);
            for (int k = 0; k < Cin; ++k) {
                double gg[3][3];
                for (int a = 0; a < 3; ++a)
                    for (int b2 = 0; b2 < 3; ++b2) gg[a][b2] = (double)w[(((size_t)n * Cin + k) * 3 + a) * 3 + b2] * scale;
                for (int i = 0; i < 4; ++i)
                    for (int j = 0; j < 4; ++j) {
                        double v = 0.0;
                        for (int a = 0; a < 3; ++a)
                            for (int b2 = 0; b2 < 3; ++b2) v += G[i][a] * gg[a][b2] * G[j][b2];
                        const int nt = n / 32, ln = n % 32, c = k / 16, kk = k % 16, gq = kk / 8, hi = (kk % 8) / 4, e = kk % 4;
                        const int lane = hi * 32 + ln;
                        pv[((((size_t)(nt * 16 + i * 4 + j) * nch + c) * 2 + gq) * 64 + lane) * 4 + e] = (float)v;
                    }
            }
        }
        TRY(upload(m, &m->wino_v[l], pv));
        if ((m->wino_f16_mask & (1u << l)) && Cout % 64 == 0) {
            // F16 kernel: fragment (nt, xi, chunk c, piece q), lane (n = lane & 31, hi = lane >> 5), 8 fp16:
            // piece q of V_xi[n][k = 16 c + 8 hi + j] -- the fp32 fragments' bytes and addressing, g replaced by q
            std::vector<float> pv16(pv.size());  // two 2-byte pieces per weight = the fp32 array's bytes
            m->wino_wscale[l] = pick_wscale(pv.data(), pv.size());
            uint16_t *q16 = reinterpret_cast<uint16_t *>(pv16.data());
            for (int nt = 0; nt < Cout / 32; ++nt)
                for (int xi = 0; xi < 16; ++xi)
                    for (int c = 0; c < nch; ++c)
                        for (int gq = 0; gq < 2; ++gq)
                            for (int hi = 0; hi < 2; ++hi)
                                for (int ln = 0; ln < 32; ++ln)
                                    for (int e = 0; e < 4; ++e) {
                                        const float v = pv[((((size_t)(nt * 16 + xi) * nch + c) * 2 + gq) * 64 + hi * 32 + ln) * 4 + e];
                                        const int kk = 8 * gq + 4 * hi + e;  // channel within the chunk (fp32 fragment order)
                                        const int hi16 = kk / 8, j = kk % 8;
                                        const float vs = v * m->wino_wscale[l];  // exact; wp.post_scale undoes it
                                        const _Float16 h0 = (_Float16)vs, h1 = (_Float16)(vs - (float)h0);
                                        const size_t base = (((size_t)(nt * 16 + xi) * nch + c) * 2) * 64 * 8;
                                        memcpy(&q16[base + (size_t)(0 * 64 + hi16 * 32 + ln) * 8 + j], &h0, 2);
                                        memcpy(&q16[base + (size_t)(1 * 64 + hi16 * 32 + ln) * 8 + j], &h1, 2);
                                    }
            TRY(upload(m, &m->wino_v16[l], pv16));
        }
        m->use_wino[l] = true;
    }
    return 0;
}

// ------------------------------------------------------------------------------------------ forward passes
static int run_tail(c3_model *m, hipStream_t s, const float *a, int64_t lda, int64_t n, float *y, const char *tag_l4,
                    const char *tag_tail) {
    const int FC = m->FC, K4 = m->K4;
    const int nk_total = K4 / kBK;
    const int S = l4_splits(m);
    {
        ProfScope ps(m, s, tag_l4, 2.0 * n * FC * K4, 4.0 * (n * K4 + (double)FC * K4 + (double)S * n * FC));
        DenseLoaderParams lp{a, lda};
        EpilogueParams ep{m->part, nullptr, nullptr, FC, n * FC};
        const bool l4_f16 = m->f16_ok && m->l4_split && m->l4_w3 && m->tail_mfma && m->w5f;
        ps.mfma(2.0 * ((n + 127) / 128 * 128) * FC * K4 * (l4_f16 ? (m->split_kind == 1 ? 6 : 3) : 1), l4_f16);
        if (l4_f16)  // (the scalar tail sums the partials itself and knows no scale)
            TRY(LAUNCH_SPLIT(m, DenseLoader<4>, EPI_PARTIAL, 128, 64, s, lp, m->l4_w, K4, (int)n, FC, nk_total / S, S, ep, m->l4_w3));  // partials carry l4_wscale
        else
            TRY((launch_gemm<DenseLoader<4>, EPI_PARTIAL, 128, 64>(s, lp, m->l4_w, K4, (int)n, FC, nk_total / S, S, ep)));
    }
    const double fl = 2.0 * n * (FC * 128.0 * m->nb + 128.0 * m->nout);
    if (m->tail_mfma && m->w5f) {
        ProfScope ps(m, s, tag_tail, fl, 4.0 * ((double)S * n * FC + n * m->nout));
        ps.mfma(2.0 * ((n + 15) / 16 * 16) * m->nb * (FC * 128.0 + 128.0 * 48.0), false);
        ReduceParams rp{m->part, m->l4_b, m->l4dbg, (int)n, FC, S};
        if (m->f16_ok && m->l4_split && m->l4_w3) rp.pre = m->l4_wscale, rp.post = 1.f / m->l4_wscale;  // same condition as the launch above (tail_mfma holds here)
        hipLaunchKernelGGL(splitk_reduce_selu_kernel, dim3((unsigned)((n * FC + 255) / 256)), dim3(256), 0, s, rp);
        HIP_TRY(hipGetLastError());
        Tail2Params tp{m->l4dbg, m->w5f, m->b5, m->whf, m->bh48, y, (int)n, m->nb, m->row};
        const dim3 grid((unsigned)((n + 15) / 16), m->nb);
        if (FC == 256)
            hipLaunchKernelGGL(fc_tail_mfma_kernel<256>, grid, dim3(256), 0, s, tp);
        else
            hipLaunchKernelGGL(fc_tail_mfma_kernel<128>, grid, dim3(256), 0, s, tp);
        HIP_TRY(hipGetLastError());
    } else {
        ProfScope ps(m, s, tag_tail, fl, 4.0 * ((double)S * n * FC + n * m->nout));
        TailParams tp{m->part, m->l4_b, m->w5t, m->b5, m->wh, m->bh, y, m->keep ? m->l4dbg : nullptr,
                      (int)n, S, m->nb, m->nout, m->row};
        const int grid = (int)((n + kTailWindows - 1) / kTailWindows);
        if (FC == 256)
            hipLaunchKernelGGL(fc_tail_kernel<256>, dim3(grid), dim3(256), 0, s, tp);
        else
            hipLaunchKernelGGL(fc_tail_kernel<128>, dim3(grid), dim3(256), 0, s, tp);
        HIP_TRY(hipGetLastError());
    }
    if (m->row > m->nout) {  // decoder columns behind the probabilities of every row (c3_decode.h)
        ProfScope ps(m, s, m->kind == C3_KIND_PILEUP ? "p.decode" : "fa.decode", 0.0, 4.0 * n * m->row);
        DecodeParams dp{y, m->row, nullptr, nullptr, nullptr, nullptr, y + m->nout, (int)n, m->nout == 90 ? 1 : 0};
        hipLaunchKernelGGL(outcome_maxima_kernel<true>, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, dp);
        HIP_TRY(hipGetLastError());
    }
    return 0;
}

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

// ---- plane-activation pipeline (c3_conv3.h): the default whenever the handle is on the fp16x3 kernels ----
// magic of fast_div (c3_gemm.h) for divisor d and dividends below n: 0 = "d is 1"; fails when n * d does not fit 32 bits
static int div_magic(int d, int64_t n, uint32_t *magic) {
    if (d <= 1) return *magic = 0u, 0;
    if (n * d >= ((int64_t)1 << 32)) return fail("batch too large for the 32-bit pixel arithmetic of the convolution kernels");
    return *magic = (uint32_t)((((uint64_t)1 << 32) / (uint64_t)d) + 1), 0;
}

static bool fa_planes_ok(const c3_model *m) {
    if (!m->fa_planes || !m->f16_ok || m->split_kind != 2) return false;
    int hh[10], ww[10];
    fa_geometry(m, hh, ww);
    for (int l : {1, 2, 4, 5, 7, 8})
        if (!m->pconv_w[l] || ww[l] > kPlMaxW) return false;
    for (int l : {3, 6})
        if (!m->conv_w3[l] && !m->pconv_w[l]) return false;
    if (m->C == 8 ? !(m->conv1_direct && m->conv1_f16 && m->conv1_wfrag16) : !m->conv_w3[0]) return false;
    return true;
}

static int run_fa_planes(c3_model *m, hipStream_t s, const int8_t *x, int64_t n, float *y) {
    int hh[10], ww[10];
    fa_geometry(m, hh, ww);
    int cin = m->C;
    for (int l = 0; l < 9; ++l) {
        const int Cout = kConvCout[l];
        const int M = (int)(n * hh[l + 1] * ww[l + 1]);
        // conv1 inside the first residual block (c3_conv3.h SRC8; 8-channel windows, or 9 with the dwell channel)
        const bool fuse1 = m->conv1_fused && (m->C == 8 || m->C == 9) && m->conv1_wfrag16 && !m->keep && ww[1] <= kPlMaxW && ww[0] >= 3;
        if (l == 0 && fuse1) {  // no launch, no conv1 planes: res1a computes its input rows, res1b its residual, from the windows
            cin = Cout;
            continue;
        }
        double flops = 2.0 * M * Cout * 9.0 * cin;
        double bytes = (l == 0 ? 1.0 : 4.0) * n * hh[l] * ww[l] * cin + 4.0 * M * Cout * (l % 3 == 2 ? 2 : 1) + 4.0 * Cout * 9.0 * cin;
        if (fuse1 && l == 1) flops += 2.0 * M * 64.0 * 9.0 * m->C, bytes += 1.0 * n * hh[0] * ww[0] * m->C - 4.0 * M * 64;  // conv1's algorithmic work rides here
        if (fuse1 && l == 2) bytes += 1.0 * n * hh[0] * ww[0] * m->C - 4.0 * M * 64;  // residual from the windows, not from conv1 planes
        ProfScope ps(m, s, kFaLayerTag[l], flops, bytes);
        if (l == 0 && cin == 8) {
            ps.mfma(2.0 * ((M + 31) / 32 * 32) * 64.0 * 80.0 * 2, true);
            Conv1F16Params cp;
            cp.x = x, cp.wfrag = reinterpret_cast<const uint32_t *>(m->conv1_wfrag16), cp.bias = m->conv_b[0], cp.out = m->act[0];
            cp.range_flag = m->range_flag;
            cp.B = (int)n, cp.H = hh[0], cp.W = ww[0], cp.OH = hh[1], cp.OW = ww[1], cp.M = M, cp.groups = (M + 31) / 32;
            const int grid = std::min((cp.groups + 3) / 4, m->wg_slots);
            hipLaunchKernelGGL(conv1_i8_f16_kernel<true>, dim3(grid), dim3(256), 0, s, cp);
            HIP_TRY(hipGetLastError());
        } else if (l == 0) {
            ps.mfma(2.0 * ((M + 127) / 128 * 128) * 64.0 * 96.0 * 3, true);
            Conv1LoaderParams lp{x, (const int8_t *)m->zeros, hh[0], ww[0], cin, hh[1], ww[1]};
            EpilogueParams ep{m->act[0], m->conv_b[0], nullptr, Cout, 0};
            ep.post_scale = 1.f / m->conv_wscale[0], ep.range_flag = m->range_flag;
            TRY((launch_gemm<Conv1Loader<4>, EPI_BIAS_RELU_PLANES, 128, 64, 2>(s, lp, m->conv_w[0], 96, M, Cout, 3, 1, ep, m->conv_w3[0])));
        } else if (kConvStride[l] == 2 && m->pconv_w[l]) {
            DensePlanesParams dp;
            dp.a = m->act[l - 1], dp.w = m->pconv_w[l], dp.bias = m->conv_b[l], dp.c = m->act[l], dp.post_scale = 1.f / m->pconv_wscale[l];
            dp.M = M, dp.N = Cout, dp.K = 9 * cin, dp.tiles_n = Cout / kDnBN, dp.tiles = (M + kDnBM - 1) / kDnBM * dp.tiles_n;
            dp.Hin = hh[l], dp.Win = ww[l], dp.Cin = cin, dp.Ho = hh[l + 1], dp.Wo = ww[l + 1], dp.stride = 2, dp.range_flag = m->range_flag;
            TRY(div_magic(hh[l + 1] * ww[l + 1], (int64_t)M + 2 * kDnBM, &dp.mg_hw));
            TRY(div_magic(ww[l + 1], hh[l + 1] * ww[l + 1], &dp.mg_w));
            ps.mfma(2.0 * ((M + kDnBM - 1) / kDnBM * kDnBM) * (double)Cout * 9.0 * cin * 3, true);
            const int grid = std::min(dp.tiles, m->wg_slots / 2);  // one 512-thread workgroup (136 KB of LDS) per CU
            if (m->dense_mode == 4) hipLaunchKernelGGL(dense_planes_ws_kernel<true>, dim3(grid), dim3(kWsThreads), 0, s, dp);
            else if (m->dense_mode == 3 || m->dense_mode == 5) hipLaunchKernelGGL(dense_planes_pipe_kernel<true>, dim3(grid), dim3(kDnThreads), 0, s, dp);
            else hipLaunchKernelGGL(dense_planes_kernel<true>, dim3(grid), dim3(kDnThreads), 0, s, dp);
            HIP_TRY(hipGetLastError());
        } else if (kConvStride[l] == 2) {
            ps.mfma(2.0 * ((M + 127) / 128 * 128) * (double)Cout * 9.0 * cin * 3, true);
            PlaneConvLoaderParams lp{m->act[l - 1], m->zeros, hh[l], ww[l], cin, hh[l + 1], ww[l + 1], 2, cin / kBK};
            EpilogueParams ep{m->act[l], m->conv_b[l], nullptr, Cout, 0};
            ep.post_scale = 1.f / m->conv_wscale[l], ep.range_flag = m->range_flag;
            const int nk = 9 * cin / kBK;
            const int64_t ldb = 9 * cin;
            if (!(m->conv_bn64_mask & (1u << l)))
                TRY((launch_gemm<PlaneConvLoader<4>, EPI_BIAS_RELU_PLANES, 128, 128, 2>(s, lp, m->conv_w[l], ldb, M, Cout, nk, 1, ep, m->conv_w3[l])));
            else
                TRY((launch_gemm<PlaneConvLoader<4>, EPI_BIAS_RELU_PLANES, 128, 64, 2>(s, lp, m->conv_w[l], ldb, M, Cout, nk, 1, ep, m->conv_w3[l])));
        } else {
            const bool res = l % 3 == 2;
            PlaneConvParams cp;
            cp.x = m->act[l - 1], cp.w = m->pconv_w[l], cp.bias = m->conv_b[l], cp.res = res ? m->act[l - 2] : nullptr, cp.out = m->act[l];
            cp.range_flag = m->range_flag, cp.post_scale = 1.f / m->pconv_wscale[l];
            cp.M = M, cp.H = hh[l], cp.W = ww[l];
            TRY(div_magic(hh[l] * ww[l], (int64_t)M + 2 * kPlBM, &cp.mg_hw));
            TRY(div_magic(ww[l], hh[l] * ww[l], &cp.mg_w));
            const int tiles_m = (M + kPlBM - 1) / kPlBM;
            cp.tiles = tiles_m * (Cout / 64);
            const bool src8 = fuse1 && (l == 1 || l == 2);
            // PyramidPolling as the epilogue of the last convolution (c3_conv3.h SPPF): 12 x 5 windows, four whole windows per tile
            const bool sppf = l == 8 && m->spp_fused && !m->keep && hh[9] == 12 && ww[9] == 5 && 14 * 256 == m->K4;
            if (sppf) {
                cp.spp = m->spp;
                cp.tiles = (int)((n + 3) / 4) * (Cout / 64);
            }
            if (src8) {
                cp.x8 = x, cp.c1w = reinterpret_cast<const uint32_t *>(m->conv1_wfrag16), cp.c1b = m->conv_b[0], cp.Hin = hh[0], cp.Win = ww[0];
                if (l == 1) cp.x = nullptr;
                else cp.res = nullptr;
            }
            // SRC8: + conv1 for 320 halo rows (res1a) / the tile's 256 pixels (res1b), two piece products of K = 80 (96 for 9 channels)
            const double tiles_x = sppf ? (double)((n + 3) / 4) : (double)tiles_m;  // pixel tiles the launch really runs
            ps.mfma(2.0 * tiles_x * kPlBM * (double)Cout * 9.0 * cin * 3 +
                        (src8 ? 2.0 * tiles_m * (l == 1 ? 320 : 256) * 64.0 * (m->C == 8 ? 80.0 : 96.0) * 2 : 0.0),
                    true);
            // persistent: one workgroup per tile when they all fit (2 per CU), else wg_slots rounded down so that a
            // workgroup's tiles share their column tile (c3_conv3.h)
            int g = cp.tiles;
            const int cus = m->wg_slots / 2, unit = 8 * (Cout / 64);  // one 512-thread workgroup (114 KB of LDS) per CU
            if (g > cus) g = std::max(unit, cus / unit * unit);
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
    if (!(m->spp_fused && !m->keep && hh[9] == 12 && ww[9] == 5 && 14 * 256 == m->K4)) {
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

static int run_fa(c3_model *m, hipStream_t s, const int8_t *x, int64_t n, float *y) {
    if (fa_planes_ok(m)) {
        m->choice_fa = "planes-f16x3";
        return run_fa_planes(m, s, x, n, y);
    }
    m->choice_fa = m->f16_ok ? "fp32-activations-winograd-f16x3" : "fp32-activations-fp32-mfma";
    m->last_planes = false;
    int hh[10], ww[10];
    fa_geometry(m, hh, ww);
    int cin = m->C;
    for (int l = 0; l < 9; ++l) {
        const int Cout = kConvCout[l];
        const int M = (int)(n * hh[l + 1] * ww[l + 1]);
        const double flops = 2.0 * M * Cout * 9.0 * cin;
        const double bytes = (l == 0 ? 1.0 : 4.0) * n * hh[l] * ww[l] * cin + 4.0 * M * Cout * (l % 3 == 2 ? 2 : 1) +
                             4.0 * Cout * 9.0 * cin;
        ProfScope ps(m, s, kFaLayerTag[l], flops, bytes);
        EpilogueParams ep{m->act[l], m->conv_b[l], l % 3 == 2 ? m->act[l - 2] : nullptr, Cout, 0};
        if (m->use_wino[l] && m->wino_mask & (1u << l)) {
            WinoParams wp;
            wp.x = m->act[l - 1], wp.zeros = m->zeros, wp.v = m->wino_v[l], wp.bias = m->conv_b[l];
            wp.res = l % 3 == 2 ? m->act[l - 2] : nullptr, wp.out = m->act[l];
            wp.B = (int)n, wp.H = hh[l], wp.W = ww[l], wp.Cin = cin, wp.Cout = Cout;
            wp.th = (hh[l] + 1) / 2, wp.tw = (ww[l] + 1) / 2, wp.P = (int)n * wp.th * wp.tw;
            wp.tiles_n = Cout / kWinoNT, wp.tiles = ((wp.P + kWinoPT - 1) / kWinoPT) * wp.tiles_n;
            if ((m->wino_p_mask & (1u << l)) && Cout % 64 == 0) {  // persistent 32 x 64 workgroups
                wp.tiles_n = Cout / 64, wp.tiles = ((wp.P + 31) / 32) * wp.tiles_n;
                const int grid = std::min(wp.tiles, m->wg_slots / wp.tiles_n * wp.tiles_n);
                const bool wf16 = m->f16_ok && m->wino_v16[l] && (m->wino_f16_mask & (1u << l));
                ps.mfma(2.0 * ((wp.P + 31) / 32 * 32) * 16.0 * cin * Cout * (wf16 ? 3 : 1), wf16);
                if (wf16) {
                    wp.v = m->wino_v16[l], wp.post_scale = 1.f / m->wino_wscale[l], wp.range_flag = m->range_flag;
                    if (wp.res)
                        hipLaunchKernelGGL((wino_conv_kernel_p<true, 0, 0, true>), dim3(grid), dim3(256), 0, s, wp);
                    else
                        hipLaunchKernelGGL((wino_conv_kernel_p<false, 0, 0, true>), dim3(grid), dim3(256), 0, s, wp);
                } else if (wp.res)
                    hipLaunchKernelGGL(wino_conv_kernel_p<true>, dim3(grid), dim3(256), 0, s, wp);
                else
                    hipLaunchKernelGGL(wino_conv_kernel_p<false>, dim3(grid), dim3(256), 0, s, wp);
            } else if ((m->wino_n64_mask & (1u << l)) && Cout % 64 == 0) {  // 32 tiles x 64 couts per workgroup
                wp.tiles_n = Cout / 64, wp.tiles = ((wp.P + 31) / 32) * wp.tiles_n;
                if (wp.res)
                    hipLaunchKernelGGL(wino_conv_kernel_n64<true>, dim3(wp.tiles), dim3(256), 0, s, wp);
                else
                    hipLaunchKernelGGL(wino_conv_kernel_n64<false>, dim3(wp.tiles), dim3(256), 0, s, wp);
            } else {
                if (wp.res)
                    hipLaunchKernelGGL(wino_conv_kernel<true>, dim3(wp.tiles), dim3(256), 0, s, wp);
                else
                    hipLaunchKernelGGL(wino_conv_kernel<false>, dim3(wp.tiles), dim3(256), 0, s, wp);
            }
            HIP_TRY(hipGetLastError());
        } else if (l == 0 && cin == 8 && m->conv1_direct && m->conv1_f16 && m->conv1_wfrag16 && m->f16_ok) {
            ps.mfma(2.0 * ((M + 31) / 32 * 32) * 64.0 * 80.0 * 2, true);  // 5 k-steps of 16, two weight pieces
            Conv1F16Params cp;
            cp.x = x, cp.wfrag = reinterpret_cast<const uint32_t *>(m->conv1_wfrag16), cp.bias = m->conv_b[0], cp.out = m->act[0];
            cp.range_flag = m->range_flag;
            cp.B = (int)n, cp.H = hh[0], cp.W = ww[0], cp.OH = hh[1], cp.OW = ww[1], cp.M = M, cp.groups = (M + 31) / 32;
            const int grid = std::min((cp.groups + 3) / 4, m->wg_slots);
            hipLaunchKernelGGL(conv1_i8_f16_kernel, dim3(grid), dim3(256), 0, s, cp);
            HIP_TRY(hipGetLastError());
        } else if (l == 0 && cin == 8 && m->conv1_direct && m->conv1_wfrag) {
            ps.mfma(2.0 * ((M + 31) / 32 * 32) * 64.0 * 72.0, false);
            Conv1Params cp;
            cp.x = x, cp.wfrag = m->conv1_wfrag, cp.bias = m->conv_b[0], cp.out = m->act[0];
            cp.B = (int)n, cp.H = hh[0], cp.W = ww[0], cp.OH = hh[1], cp.OW = ww[1], cp.M = M, cp.groups = (M + 31) / 32;
            const int grid = std::min((cp.groups + 3) / 4, m->wg_slots);
            hipLaunchKernelGGL(conv1_i8_kernel, dim3(grid), dim3(256), 0, s, cp);
            HIP_TRY(hipGetLastError());
        } else if (l == 0) {
            Conv1LoaderParams lp{x, (const int8_t *)m->zeros, hh[0], ww[0], cin, hh[1], ww[1]};
            const bool c1f16 = m->f16_ok && m->conv1_f16 && m->conv_w3[0];
            ps.mfma(2.0 * ((M + 127) / 128 * 128) * 64.0 * 96.0 * (c1f16 ? (m->split_kind == 1 ? 6 : 3) : 1), c1f16);
            if (c1f16) {
                ep.post_scale = 1.f / m->conv_wscale[0], ep.range_flag = m->range_flag;
                TRY(LAUNCH_SPLIT(m, Conv1Loader<4>, EPI_BIAS_RELU, 128, 64, s, lp, m->conv_w[0], 96, M, Cout, 3, 1, ep, m->conv_w3[0]));
            } else {
                TRY((launch_gemm<Conv1Loader<4>, EPI_BIAS_RELU, 128, 64>(s, lp, m->conv_w[0], 96, M, Cout, 3, 1, ep)));
            }
        } else {
            ConvLoaderParams lp{m->act[l - 1], m->zeros, hh[l], ww[l], cin, hh[l + 1], ww[l + 1], kConvStride[l], cin / kBK};
            const int nk = 9 * cin / kBK;
            const int64_t ldb = 9 * cin;
            const bool res = l % 3 == 2;
            const bool cf16 = m->f16_ok && !res && m->conv_w3[l] && (m->conv_split_mask & (1u << l));
            ps.mfma(2.0 * ((M + 127) / 128 * 128) * (double)Cout * 9.0 * cin * (cf16 ? (m->split_kind == 1 ? 6 : 3) : 1), cf16);
            if (cf16) {
                ep.post_scale = 1.f / m->conv_wscale[l], ep.range_flag = m->range_flag;
                // fp16x3 keeps the fp32 kernel's LDS footprint, so conv3 (N = 128) can use 128x128 tiles at two workgroups
                // per CU (43 -> 38 us); conv5 stays on 128x64 (480 workgroups), bf16x6 needs 72 KB per 128x64 tile
                if (m->split_kind == 2 && !(m->conv_bn64_mask & (1u << l)))
                    TRY((launch_gemm<ConvLoader<4>, EPI_BIAS_RELU, 128, 128, 2>(s, lp, m->conv_w[l], ldb, M, Cout, nk, 1, ep, m->conv_w3[l])));
                else
                    TRY(LAUNCH_SPLIT(m, ConvLoader<4>, EPI_BIAS_RELU, 128, 64, s, lp, m->conv_w[l], ldb, M, Cout, nk, 1, ep, m->conv_w3[l]));
            } else if (Cout == 64 || (m->conv_bn64_mask & (1u << l))) {
                if (res)
                    TRY((launch_gemm<ConvLoader<4>, EPI_BIAS_RES_RELU, 128, 64>(s, lp, m->conv_w[l], ldb, M, Cout, nk, 1, ep)));
                else
                    TRY((launch_gemm<ConvLoader<4>, EPI_BIAS_RELU, 128, 64>(s, lp, m->conv_w[l], ldb, M, Cout, nk, 1, ep)));
            } else {
                if (res)
                    TRY((launch_gemm<ConvLoader<4>, EPI_BIAS_RES_RELU, 128, 128>(s, lp, m->conv_w[l], ldb, M, Cout, nk, 1, ep)));
                else
                    TRY((launch_gemm<ConvLoader<4>, EPI_BIAS_RELU, 128, 128>(s, lp, m->conv_w[l], ldb, M, Cout, nk, 1, ep)));
            }
        }
        cin = Cout;
    }
    {
        SppParams sp;
        TRY(spp_bins(m, hh[9], ww[9], sp));
        sp.in = m->act[8], sp.out = m->spp, sp.B = (int)n;
        ProfScope ps(m, s, "fa.spp", 0.0, 4.0 * n * (hh[9] * ww[9] * 256.0 + m->K4));
        if (hh[9] == 12 && ww[9] == 5) {  // ONT geometry: fully unrolled specialisation
            const int grid = (int)std::min<int64_t>(n, 8192);  // one block = the 256 channels of one window
            hipLaunchKernelGGL((spp_kernel_fixed<12, 5>), dim3(grid), dim3(256), 0, s, m->act[8], m->spp, (int)n, 256);
        } else {
            const int64_t total = n * m->K4;
            const int grid = (int)std::min<int64_t>((total + 255) / 256, 8192);
            hipLaunchKernelGGL(spp_kernel, dim3(grid), dim3(256), 0, s, sp);
        }
        HIP_TRY(hipGetLastError());
    }
    return run_tail(m, s, m->spp, m->K4, n, y, "fa.l4", "fa.tail");
}

// debug (C3HIP_LSTM_TRACE): shader-clock stamps of workgroup (0, 0), every wave, four phase boundaries per step
static hipError_t lstm_trace_begin(c3_model *m) {
    constexpr size_t bytes = 8 * 64 * 4 * sizeof(unsigned long long);
    if (!m->lstm_trace_dev) {
        const hipError_t e = hipMalloc((void **)&m->lstm_trace_dev, bytes);
        if (e != hipSuccess) return e;
    }
    return hipMemset(m->lstm_trace_dev, 0, bytes);
}
static hipError_t lstm_trace_print(c3_model *m, hipStream_t s, const char *name, int T, const char *legend) {
    static unsigned long long h[8 * 64 * 4];
    hipError_t e = hipStreamSynchronize(s);
    if (e != hipSuccess) return e;
    e = hipMemcpy(h, m->lstm_trace_dev, sizeof(h), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return e;
    fprintf(stderr, "%s trace (workgroup 0, shader cycles; %s)\n", name, legend);
    for (int w = 0; w < 8; ++w) {
        for (int st = 8; st < 12 && st + 1 < T; ++st) {
            const unsigned long long *a = h + (w * 64 + st) * 4, *b = a + 4;
            fprintf(stderr, "  wave %d step %2d: %6llu %6llu %6llu %6llu | step %6llu | top vs wave 0: %+lld\n", w, st, a[1] - a[0], a[2] - a[1],
                    a[3] - a[2], b[0] - a[3], b[0] - a[0], (long long)(a[0] - h[st * 4]));
        }
    }
    fprintf(stderr, "  whole launch, wave 0: %llu cycles for %d steps\n", h[(T - 1) * 4 + 3] - h[0], T);
    return hipSuccess;
}

template <typename T>
static int run_pileup_t(c3_model *m, hipStream_t s, const T *x, int64_t n, float *y, const int32_t *starts = nullptr) {
    const int Tn = m->positions;
    const int M = (int)(n * Tn);
    const bool fused1 = m->lstm1_fused && m->l1_wih != nullptr;
    if (starts && !fused1) return fail("region gathering needs the fused LSTM1 kernel (input_channels <= 20, C3HIP_LSTM1_FUSED != 0)");
    // LSTM1 -> planes -> dense_planes_kernel (c3_dense.h) whenever the fp16x3 fused LSTM1 runs and the projection is packed for it
    const bool h1_planes = fused1 && m->f16_ok && m->lstm1_f16 && m->whh16[0] && (sizeof(T) != 1 || m->l1_wih16) && m->proj2_planes &&
                           m->proj2_pw && m->lstm2_v2;
    m->last_planes = h1_planes;
    if (fused1) {
        ProfScope ps(m, s, "p.lstm1", 2.0 * M * 1024.0 * m->C + 2.0 * M * 2.0 * 512.0 * 128.0, sizeof(T) * (double)M * m->C + 4.0 * M * 256.0);
        {
            const bool f16 = m->f16_ok && m->lstm1_f16 && m->whh16[0] && (sizeof(T) != 1 || m->l1_wih16);
            const bool half = f16 && (m->adaptive ? !m->concurrent : (m->lstm_opt & 4) != 0) && (m->lstm_opt & 1) && h1_planes && sizeof(T) == 1 &&
                              2 * ((n + 15) / 16) <= m->wg_slots / 2 && !(m->lstm_trace_left > 0);
            const double tiles = (double)(half ? (n + 7) / 8 * 16 : (n + 15) / 16 * 16) * Tn * 2;  // (window, step, direction) rows of the 16-row tiles
            // recurrent part 512 x 128 as fp16x3 (or fp32); input part: int8 windows 512 x 32 against two weight pieces, else 512 x 20 fp32
            ps.mfma(f16 ? tiles * 2.0 * 512 * (128 * 3 + (sizeof(T) == 1 ? 32 * 2 : 0)) : tiles * 2.0 * 512 * (128 + 20), f16);
        }
        LstmFusedParams<T> lp{x, starts, m->l1_wih, m->l1_bias, m->whh[0], reinterpret_cast<const uint32_t *>(m->l1_wih16), m->h1, (int)n, Tn, m->C};
        if (m->f16_ok && m->lstm1_f16 && m->whh16[0] && (sizeof(T) != 1 || m->l1_wih16)) {
            lp.whh = m->whh16[0];
            if (h1_planes) lp.hplanes = m->h1;  // h1 leaves as fp16 piece planes for dense_planes_kernel
            dim3 grid((unsigned)((n + 15) / 16), 2);
            bool launched = false;
            // half tiles (8 windows per workgroup) while the full tiles would leave CUs idle: c3_lstm_fused.h OPT bit 2
            const bool want_half = m->adaptive ? !m->concurrent : (m->lstm_opt & 4) != 0;
            const bool half1 = want_half && (m->lstm_opt & 1) && h1_planes && sizeof(T) == 1 && 2 * ((n + 15) / 16) <= m->wg_slots / 2 &&
                               !(m->lstm_trace_left > 0);
            if constexpr (sizeof(T) == 1) {
                m->choice_lstm1 = half1 ? "fused-f16x3-half-tiles" : "fused-f16x3-full-tiles";
                if (half1) {
                    grid = dim3((unsigned)((n + 7) / 8), 2);
                    hipLaunchKernelGGL((lstm1_fused_kernel<T, true, 7>), grid, dim3(512), 0, s, lp);
                    launched = true;
                } else
                if (m->lstm_trace_left > 0 && --m->lstm_trace_left == 0) {  // debug (C3HIP_LSTM_TRACE): this launch is traced
                    HIP_TRY(lstm_trace_begin(m));
                    lp.trace = m->lstm_trace_dev;
                    if ((m->lstm_opt & 1) && h1_planes) hipLaunchKernelGGL((lstm1_fused_kernel<T, true, 11>), grid, dim3(512), 0, s, lp);
                    else hipLaunchKernelGGL((lstm1_fused_kernel<T, true, 8>), grid, dim3(512), 0, s, lp);
                    HIP_TRY(lstm_trace_print(m, s, "lstm1", Tn, "top -> matrix instructions issued -> cell + h written -> barrier -> next top"));
                    launched = true;
                } else if ((m->lstm_opt & 1) && h1_planes) {
                    hipLaunchKernelGGL((lstm1_fused_kernel<T, true, 3>), grid, dim3(512), 0, s, lp);
                    launched = true;
                }
            }
            if (!launched && sizeof(T) != 1 && (m->lstm_opt & 1) && h1_planes && !(m->lstm_trace_left > 0)) {
                hipLaunchKernelGGL((lstm1_fused_kernel<T, true, 3>), grid, dim3(512), 0, s, lp);  // int32 windows: same deferral, fp32 projection
                launched = true;
            }
            if (!launched) hipLaunchKernelGGL((lstm1_fused_kernel<T, true>), grid, dim3(512), 0, s, lp);
        } else {
            hipLaunchKernelGGL(lstm1_fused_kernel<T>, dim3((unsigned)((n + 15) / 16), 2), dim3(512), 0, s, lp);
        }
        HIP_TRY(hipGetLastError());
    }
    if (!fused1) {
        ProfScope ps(m, s, "p.proj1", 2.0 * M * 1024.0 * m->C, sizeof(T) * (double)M * m->C + 4.0 * M * 1024);
        IntRowLoaderParams<T> lp{x, m->C};
        EpilogueParams ep{m->gx1, m->proj_b[0], nullptr, 1024, 0};
        TRY((launch_gemm<IntRowLoader<4, T>, EPI_BIAS, 128, 128>(s, lp, m->proj_w[0], 32, M, 1024, 1, 1, ep)));
    }
    if (!fused1) {
        ProfScope ps(m, s, "p.lstm1", 2.0 * M * 2.0 * 512.0 * 128.0, 4.0 * M * (1024.0 + 256.0));
        LstmParams lp{m->gx1, m->whh[0], m->h1, (int)n, Tn, 1024};
        hipLaunchKernelGGL((lstm_recurrent_kernel<128, true>), dim3((unsigned)((n + 15) / 16), 2), dim3(512), 0, s, lp);
        HIP_TRY(hipGetLastError());
    }
    {
        ProfScope ps(m, s, "p.proj2", 2.0 * M * 1280.0 * 256.0, 4.0 * M * (256.0 + 1280.0));
        const bool p2f16 = h1_planes || (m->f16_ok && m->proj2_split && m->proj2_w3 && m->lstm2_v2);
        ps.mfma(2.0 * ((M + 127) / 128 * 128) * 1280.0 * 256.0 * (p2f16 ? (m->split_kind == 1 && !h1_planes ? 6 : 3) : 1), p2f16);
        if (h1_planes) {
            DensePlanesParams dp;
            dp.a = m->h1, dp.w = m->proj2_pw, dp.bias = m->proj_b[1], dp.c = m->gx2, dp.post_scale = 1.f / m->proj2_pwscale;
            dp.M = M, dp.N = 1280, dp.K = 256, dp.tiles_n = 1280 / kDnBN, dp.tiles = ((M + kDnBM - 1) / kDnBM) * dp.tiles_n;
            const int grid = std::min(dp.tiles, m->wg_slots / 2);
            static int dtrace_left = getenv("C3HIP_DENSE_TRACE") ? atoi(getenv("C3HIP_DENSE_TRACE")) : 0;  // debug: that launch is traced
            if (dtrace_left > 0 && --dtrace_left == 0) {
                static long long *tdev = nullptr;
                static long long th[2 * 512 * 2];
                if (!tdev) HIP_TRY(hipMalloc((void **)&tdev, sizeof(th)));
                HIP_TRY(hipMemset(tdev, 0, sizeof(th)));
                dp.trace = tdev;
                if (m->dense_mode == 3) hipLaunchKernelGGL((dense_planes_pipe_kernel<false, true>), dim3(grid), dim3(kDnThreads), 0, s, dp);
                else if (m->dense_mode == 1) hipLaunchKernelGGL((dense_planes_kernel<false, true, true>), dim3(grid), dim3(kDnThreads), 0, s, dp);
                else hipLaunchKernelGGL((dense_planes_kernel<false, false, true>), dim3(grid), dim3(kDnThreads), 0, s, dp);
                HIP_TRY(hipStreamSynchronize(s));
                HIP_TRY(hipMemcpy(th, tdev, sizeof(th), hipMemcpyDeviceToHost));
                fprintf(stderr, "dense trace (proj2, workgroup 0, shader cycles since the first stamp; tags: 1 chunk top, 2 staged, 3 requested, 4 matrix instructions issued, 5 tile done)\n");
                for (int w = 0; w < 2; ++w) {
                    fprintf(stderr, "  wave %d:", w * 4);
                    for (int i = 0; i < 300 && th[(w * 512 + i) * 2]; ++i)
                        fprintf(stderr, "%s%lld:%lld", th[(w * 512 + i) * 2] == 1 ? "\n    " : " ", th[(w * 512 + i) * 2], th[(w * 512 + i) * 2 + 1] - th[1]);
                    fprintf(stderr, "\n");
                }
                dp.trace = nullptr;
            } else
            if ((m->dense_mode == 6 || (m->adaptive && m->dense_mode == 3)) && m->proj2_pwr &&
                (M + kWrBM - 1) / kWrBM >= 2 * 8 * std::max(1, m->wg_slots / 16 / (1280 / kWrBN))) {
                // weights resident in registers (c3_dense.h): 8 XCDs x lanes x 5 column tiles of workgroups, each walking the row tiles of its lane
                DenseWresParams wp;
                wp.a = m->h1, wp.w = m->proj2_pwr, wp.bias = m->proj_b[1], wp.c = m->gx2, wp.post_scale = 1.f / m->proj2_pwscale;
                wp.M = M, wp.N = 1280, wp.tiles_m = (M + kWrBM - 1) / kWrBM, wp.tiles_n = 1280 / kWrBN;
                wp.lanes_per_xcd = std::max(1, m->wg_slots / 16 / wp.tiles_n);  // CUs per XCD / column tiles (32 / 5 = 6)
                // beside other handles half as many, twice as long workgroups: 120 of them leave room for the 128 of an LSTM launch
                // of another batch (three batches in flight: 5.46 M -> 5.59 M windows/s; alone 4.9 M -> 4.3 M, hence the choice)
                if (m->adaptive && m->concurrent) wp.lanes_per_xcd = std::max(1, wp.lanes_per_xcd / 2);
                static const int env_lanes = getenv("C3HIP_WRES_LANES") ? atoi(getenv("C3HIP_WRES_LANES")) : 0;  // A/B
                if (env_lanes > 0) wp.lanes_per_xcd = env_lanes;
                m->choice_proj2 = wp.lanes_per_xcd >= std::max(1, m->wg_slots / 16 / wp.tiles_n) ? "weights-resident" : "weights-resident-half-grid";
                hipLaunchKernelGGL(dense_planes_wres_kernel<0>, dim3(8 * wp.lanes_per_xcd * wp.tiles_n), dim3(kDnThreads), 0, s, wp);
            } else
            if ((m->dense_mode == 5 || (m->adaptive && m->dense_mode == 3 && m->concurrent)) && m->proj2_pw32) {
                DenseBigParams bp;
                bp.a = m->h1, bp.w = m->proj2_pw32, bp.bias = m->proj_b[1], bp.c = m->gx2, bp.post_scale = 1.f / m->proj2_pwscale;
                bp.M = M, bp.N = 1280, bp.K = 256, bp.tiles_n = 1280 / kBgBN, bp.tiles = ((M + kBgBM - 1) / kBgBM) * bp.tiles_n;
                m->choice_proj2 = "256x256-tiles";
                hipLaunchKernelGGL(dense_planes_big_kernel, dim3(std::min(bp.tiles, m->wg_slots / 2)), dim3(kDnThreads), 0, s, bp);
            } else
            if (m->dense_mode == 4) hipLaunchKernelGGL(dense_planes_ws_kernel<false>, dim3(grid), dim3(kWsThreads), 0, s, dp);
            else if (m->dense_mode == 3) hipLaunchKernelGGL(dense_planes_pipe_kernel<false>, dim3(grid), dim3(kDnThreads), 0, s, dp);
            else if (m->dense_mode == 1) hipLaunchKernelGGL((dense_planes_kernel<false, true>), dim3(grid), dim3(kDnThreads), 0, s, dp);
            else hipLaunchKernelGGL(dense_planes_kernel<false>, dim3(grid), dim3(kDnThreads), 0, s, dp);
            HIP_TRY(hipGetLastError());
        } else if (p2f16) {
            DenseLoaderParams lp{m->h1, 256};
            EpilogueParams ep{m->gx2, m->proj_b[1], nullptr, 1280, 0};
            ep.post_scale = 1.f / m->proj2_wscale;
            // 128x64 tiles (72 KB of LDS, two workgroups per CU); 128x128 (96 KB, one per CU) measured 193 us
            TRY(LAUNCH_SPLIT(m, DenseLoader<4>, EPI_BIAS, 128, 64, s, lp, m->proj_w[1], 256, M, 1280, 8, 1, ep, m->proj2_w3));
        } else if (m->proj2_stream && m->proj2_frag && m->lstm2_v2) {
            ProjParams pp{m->h1, m->proj2_frag, m->proj_b[1], m->gx2, M, 1280, (M + 31) / 32, 40, nullptr};
            static int trace_left = getenv("C3HIP_PROJ_TRACE") ? 3 : 0;  // debug: the third launch is traced
            static unsigned long long *trace_dev = nullptr;
            if (trace_left && !trace_dev) HIP_TRY(hipMalloc((void **)&trace_dev, 24 * 8 * 8));
            if (trace_left == 1) pp.trace = trace_dev;
            const int64_t units = (int64_t)pp.row_blocks * pp.col_blocks;
            const int grid = (int)std::min<int64_t>(m->wg_slots, (units + 3) / 4);
            hipLaunchKernelGGL(proj_stream_kernel, dim3(grid), dim3(256), 0, s, pp);
            HIP_TRY(hipGetLastError());
            if (trace_left && trace_left-- == 1) {
                unsigned long long h[24 * 8];
                HIP_TRY(hipStreamSynchronize(s));
                HIP_TRY(hipMemcpy(h, trace_dev, sizeof(h), hipMemcpyDeviceToHost));
                fprintf(stderr, "proj trace (workgroup 0 wave 0; shader cycles): unit start -> q0 q1 q2 q3 | gap to next unit\n");
                for (int u = 0; u < 23; ++u)
                    fprintf(stderr, "  unit %2d: %6llu %6llu %6llu %6llu | %6llu\n", u, h[u * 8 + 1] - h[u * 8], h[u * 8 + 2] - h[u * 8 + 1],
                            h[u * 8 + 3] - h[u * 8 + 2], h[u * 8 + 4] - h[u * 8 + 3], h[(u + 1) * 8] - h[u * 8 + 4]);
            }
        } else {
            DenseLoaderParams lp{m->h1, 256};
            EpilogueParams ep{m->gx2, m->proj_b[1], nullptr, 1280, 0};
            TRY((launch_gemm<DenseLoader<4>, EPI_BIAS, 128, 128>(s, lp, m->proj_w[1], 256, M, 1280, 8, 1, ep)));
        }
    }
    {
        ProfScope ps(m, s, "p.lstm2", 2.0 * M * 2.0 * 640.0 * 160.0, 4.0 * M * (1280.0 + 320.0));
        // half tiles (8 windows per workgroup, c3_kernels.h OPT bit 2) while the full tiles would leave CUs without a workgroup and
        // no other handle is there to use them
        const bool half2 = m->lstm2_v2 && m->f16_ok && m->lstm2_f16 && m->whh16[1] && m->lstm2_half &&
                           (m->adaptive ? !m->concurrent : (m->lstm_opt & 16) != 0) && 2 * ((n + 15) / 16) <= m->wg_slots / 4 &&
                           !(m->lstm2_trace_left > 0);
        {
            const bool f16 = m->lstm2_v2 && m->f16_ok && m->lstm2_f16 && m->whh16[1];
            ps.mfma((double)(half2 ? (n + 7) / 8 * 16 : (n + 15) / 16 * 16) * Tn * 2 * 2.0 * 640 * 160 * (f16 ? 3 : 1), f16);
        }
        if (m->lstm2_v2) {
            Lstm2Params lp{m->gx2, m->whh[1], m->h2, (int)n, Tn, 1280};
            if (m->f16_ok && m->lstm2_f16 && m->whh16[1]) {
                lp.whh = m->whh16[1];
                if (m->lstm2_trace_left > 0 && --m->lstm2_trace_left == 0) {
                    HIP_TRY(lstm_trace_begin(m));
                    lp.trace = m->lstm_trace_dev;
                    hipLaunchKernelGGL((lstm_recurrent_kernel_v2<160, true, 8>), dim3((unsigned)((n + 15) / 16), 2), dim3(512), 0, s, lp);
                    HIP_TRY(lstm_trace_print(m, s, "lstm2", Tn, "top -> matrix instructions issued -> gates exchanged -> cell + h written -> barrier -> next top"));
                } else if (half2) {
                    m->choice_lstm2 = "f16x3-half-tiles";
                    hipLaunchKernelGGL((lstm_recurrent_kernel_v2<160, true, 4>), dim3((unsigned)((n + 7) / 8), 2), dim3(512), 0, s, lp);
                } else {
                    m->choice_lstm2 = "f16x3-full-tiles";
                    hipLaunchKernelGGL((lstm_recurrent_kernel_v2<160, true>), dim3((unsigned)((n + 15) / 16), 2), dim3(512), 0, s, lp);
                }
            } else {
                hipLaunchKernelGGL(lstm_recurrent_kernel_v2<160>, dim3((unsigned)((n + 15) / 16), 2), dim3(512), 0, s, lp);
            }
        } else {
            LstmParams lp{m->gx2, m->whh[1], m->h2, (int)n, Tn, 1280};
            hipLaunchKernelGGL((lstm_recurrent_kernel<160, false>), dim3((unsigned)((n + 15) / 16), 2), dim3(640), 0, s, lp);
        }
        HIP_TRY(hipGetLastError());
    }
    return run_tail(m, s, m->h2, m->K4, n, y, "p.l4", "p.tail");
}

// Is another handle of this process feeding the GPU right now?  Two kernel choices of the pileup path depend on it, in opposite
// directions: alone on the chip, a 1024-window batch leaves CUs idle and LSTM1 runs on half tiles (more
// workgroups, shorter cell phase) and the projection on 128 x 128 tiles; beside other batches the chip is full, the extra matrix
// work of half tiles costs the neighbours their clock, and the projection's 256 x 256 tiles (half the L2 and LDS traffic per
// product) win 7 %.  Every choice produces bit-identical rows (tests/test_parity_gpu.py), so only the speed depends on it.
// "Right now" = another handle queued a forward pass within the last 2 ms.
static std::mutex g_activity_mu;
static std::vector<std::pair<const c3_model *, int64_t>> g_activity;  // (handle, steady-clock ns of its last forward pass)
static bool others_active(const c3_model *m) {
    const int64_t now = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
    std::lock_guard<std::mutex> lk(g_activity_mu);
    bool busy = false, found = false;
    for (auto &e : g_activity) {
        if (e.first == m) e.second = now, found = true;
        else if (now - e.second < 2000000) busy = true;
    }
    if (!found) g_activity.emplace_back(m, now);
    return busy;
}
static void forget_activity(const c3_model *m) {
    std::lock_guard<std::mutex> lk(g_activity_mu);
    for (size_t i = 0; i < g_activity.size(); ++i)
        if (g_activity[i].first == m) {
            g_activity.erase(g_activity.begin() + i);
            return;
        }
}

static int forward_device(c3_model *m, hipStream_t s, const void *x, int x_dtype, int64_t batch, float *y,
                          const int32_t *starts = nullptr) {
    if (!m->loaded) return fail("model has no weights: call c3_model_load first");
    m->concurrent = others_active(m);
    if (batch < 0) return fail("negative batch");
    if (batch == 0) return 0;
    if (m->kind == C3_KIND_FULL_ALIGNMENT && x_dtype != C3_DTYPE_I8)
        return fail("full-alignment windows must be int8 (got dtype %d)", x_dtype);
    if (m->kind == C3_KIND_PILEUP && x_dtype != C3_DTYPE_I8 && x_dtype != C3_DTYPE_I32)
        return fail("pileup windows must be int8 or int32 (got dtype %d)", x_dtype);
    TRY(ensure_workspace(m, batch));
    const int64_t wbytes = c3_model_window_bytes(m, x_dtype);
    for (int64_t off = 0; off < batch; off += m->cap) {
        const int64_t n = std::min<int64_t>(m->cap, batch - off);
        const char *xp = starts ? (const char *)x : (const char *)x + off * wbytes;  // region matrix is shared
        const int32_t *sp = starts ? starts + off : nullptr;
        float *yp = y + off * m->row;
        if (m->kind == C3_KIND_FULL_ALIGNMENT)
            TRY(run_fa(m, s, (const int8_t *)xp, n, yp));
        else if (x_dtype == C3_DTYPE_I8)
            TRY(run_pileup_t<int8_t>(m, s, (const int8_t *)xp, n, yp, sp));
        else
            TRY(run_pileup_t<int32_t>(m, s, (const int32_t *)xp, n, yp, sp));
        m->last_n = n;
    }
    return 0;
}

// ------------------------------------------------------------------------------------------ C ABI
extern "C" {

#ifndef C3HIP_SRC_HASH
#define C3HIP_SRC_HASH "unknown"
#endif
const char *c3_version(void) { return "c3hip 0.3.0 (gfx950, fp32 data, fp16x3 split matrix products) srchash:" C3HIP_SRC_HASH; }
const char *c3_last_error(void) { return g_err.c_str(); }

int c3_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e == hipErrorNoDevice) return 0;
    if (e != hipSuccess) {
        fail("hipGetDeviceCount failed: %s", hipGetErrorString(e));
        return -1;
    }
    return n;
}

int c3_mem_info(int device, size_t *free_bytes, size_t *total_bytes) {
    int prev = 0;
    HIP_TRY(hipGetDevice(&prev));
    HIP_TRY(hipSetDevice(device));
    size_t f = 0, t = 0;
    hipError_t e = hipMemGetInfo(&f, &t);
    (void)hipSetDevice(prev);
    if (e != hipSuccess) return fail("hipMemGetInfo failed: %s", hipGetErrorString(e));
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return 0;
}

c3_model *c3_model_create(int kind, int in_channels, int add_indel_length, int device) {
    if (kind != C3_KIND_PILEUP && kind != C3_KIND_FULL_ALIGNMENT) {
        fail("unknown model kind %d", kind);
        return nullptr;
    }
    if (kind == C3_KIND_PILEUP && (in_channels < 1 || in_channels > 32)) {
        fail("pileup input_channels must be in [1,32], got %d", in_channels);
        return nullptr;
    }
    if (kind == C3_KIND_FULL_ALIGNMENT && (in_channels < 1 || in_channels > 10)) {
        fail("full-alignment input_channels must be in [1,10], got %d", in_channels);
        return nullptr;
    }
    int ndev = c3_device_count();
    if (ndev <= 0) {
        fail("no HIP device visible (libc3hip has no CPU fallback)");
        return nullptr;
    }
    if (device < 0 || device >= ndev) {
        fail("device %d out of range (%d visible)", device, ndev);
        return nullptr;
    }
    if (hipSetDevice(device) != hipSuccess) {
        fail("hipSetDevice(%d) failed", device);
        return nullptr;
    }
    c3_model *m = new c3_model();
    m->kind = kind, m->C = in_channels, m->add_indel = add_indel_length ? 1 : 0, m->device = device;
    if (hipMalloc((void **)&m->range_flag, 256) != hipSuccess || hipMemset(m->range_flag, 0, 256) != hipSuccess) {
        fail("c3_model_create: cannot allocate the range flag");
        delete m;
        return nullptr;
    }
    m->nb = m->add_indel ? 4 : 2, m->nout = m->add_indel ? 90 : 24;
    m->row = m->nout;
    m->FC = kind == C3_KIND_PILEUP ? 128 : 256;
    m->K4 = kind == C3_KIND_PILEUP ? m->positions * 320 : 14 * 256;
    if (hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&m->h2d_stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&m->d2h_stream, hipStreamNonBlocking) != hipSuccess) {
        fail("hipStreamCreate failed");
        delete m;
        return nullptr;
    }
    if (getenv("C3HIP_KEEP_ACTIVATIONS")) m->keep = true;
    if (const char *e = getenv("C3HIP_HOST_COPY_KERNEL")) m->host_copy_kernel = atoi(e);
    if (const char *e = getenv("C3HIP_WINOGRAD")) m->wino_mask = (unsigned)strtoul(e, nullptr, 0);
    if (const char *e = getenv("C3HIP_CONV_BN64MASK")) m->conv_bn64_mask = (unsigned)strtoul(e, nullptr, 0);
    if (const char *e = getenv("C3HIP_CONV_SPLITMASK")) m->conv_split_mask = (unsigned)strtoul(e, nullptr, 0);
    if (const char *e = getenv("C3HIP_L4_SPLIT")) m->l4_split = atoi(e) != 0;
    if (const char *e = getenv("C3HIP_SPLIT_KIND")) m->split_kind = atoi(e) == 1 ? 1 : 2;
    if (const char *e = getenv("C3HIP_WINOGRAD_F16MASK")) m->wino_f16_mask = (unsigned)strtoul(e, nullptr, 0);
    if (const char *e = getenv("C3HIP_PROJ2_SPLIT")) m->proj2_split = atoi(e) != 0;
    if (const char *e = getenv("C3HIP_PROJ2_PLANES")) m->proj2_planes = atoi(e) != 0;
    if (const char *e = getenv("C3HIP_PROJ2_STREAM")) m->proj2_stream = atoi(e) != 0;
    if (const char *e = getenv("C3HIP_DENSE_MODE")) m->dense_mode = atoi(e);
    if (const char *e = getenv("C3HIP_TAIL_MFMA")) m->tail_mfma = atoi(e) != 0;
    if (const char *e = getenv("C3HIP_CONV1_DIRECT")) m->conv1_direct = atoi(e) != 0;
    if (const char *e = getenv("C3HIP_WINOGRAD_PMASK")) m->wino_p_mask = (unsigned)strtoul(e, nullptr, 0);
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
            m->wg_slots = 2 * prop.multiProcessorCount;
    }
    if (const char *e = getenv("C3HIP_WINOGRAD_N64MASK")) m->wino_n64_mask = (unsigned)strtoul(e, nullptr, 0);
    if (const char *e = getenv("C3HIP_LSTM2_V2")) m->lstm2_v2 = atoi(e) != 0;
    if (const char *e = getenv("C3HIP_LSTM2_F16")) m->lstm2_f16 = atoi(e) != 0;
    if (const char *e = getenv("C3HIP_LSTM1_F16")) m->lstm1_f16 = atoi(e) != 0;
    if (const char *e = getenv("C3HIP_CONV1_F16")) m->conv1_f16 = atoi(e) != 0;
    if (const char *e = getenv("C3HIP_CONV1_FUSED")) m->conv1_fused = atoi(e) != 0;
    if (const char *e = getenv("C3HIP_SPP_FUSED")) m->spp_fused = atoi(e) != 0;
    if (const char *e = getenv("C3HIP_LSTM1_FUSED")) m->lstm1_fused = atoi(e) != 0;
    if (const char *e = getenv("C3HIP_LSTM_OPT")) m->lstm_opt = atoi(e), m->adaptive = 0;
    if (const char *e = getenv("C3HIP_LSTM2_HALF")) m->lstm2_half = atoi(e);
    if (getenv("C3HIP_DENSE_MODE")) m->adaptive = 0;
    if (const char *e = getenv("C3HIP_ADAPTIVE")) m->adaptive = atoi(e);
    if (const char *e = getenv("C3HIP_LSTM_TRACE")) m->lstm_trace_left = m->lstm2_trace_left = atoi(e);
    if (const char *e = getenv("C3HIP_FA_PLANES")) m->fa_planes = atoi(e) != 0;
    if (const char *e = getenv("C3HIP_CONV_S2_PLANES")) m->conv_s2_planes = atoi(e) != 0;
    if (hipMalloc((void **)&m->zeros, 256) != hipSuccess || hipMemset(m->zeros, 0, 256) != hipSuccess) {
        fail("hipMalloc(zero page) failed");
        c3_model_destroy(m);
        return nullptr;
    }
    return m;
}

int c3_model_set_geometry(c3_model *m, int depth, int positions) {
    if (!m) return fail("null model");
    if (positions < 1 || depth < 1) return fail("bad geometry %dx%d", depth, positions);
    HIP_TRY(hipSetDevice(m->device));
    m->depth = depth, m->positions = positions;
    if (m->kind == C3_KIND_PILEUP) m->K4 = positions * 320;
    if (m->K4 % kBK) return fail("unsupported geometry: L4 fan-in %d is not a multiple of %d", m->K4, kBK);
    free_workspace(m);
    m->loaded = false;
    return 0;
}

int c3_model_output_size(const c3_model *m) { return m ? m->nout : -1; }
int c3_model_row_size(const c3_model *m) { return m ? m->row : -1; }

int c3_model_set_decode_columns(c3_model *m, int enable) {
    if (!m) return fail("null model");
    for (const HostSlot &sl : m->slot)
        if (sl.busy) return fail("a prediction is in flight: call c3_predict_wait first");
    m->row = m->nout + (enable ? kDecodeCols : 0);
    return 0;
}

int64_t c3_model_window_bytes(const c3_model *m, int x_dtype) {
    if (!m) return -1;
    const int64_t item = x_dtype == C3_DTYPE_I32 ? 4 : 1;
    if (m->kind == C3_KIND_PILEUP) return item * m->positions * m->C;
    return item * m->depth * m->positions * m->C;
}

int c3_model_load(c3_model *m, const c3_tensor_desc *tensors, int n_tensors) {
    if (!m || (!tensors && n_tensors)) return fail("null argument");
    HIP_TRY(hipSetDevice(m->device));
    TensorMap tm;
    for (int i = 0; i < n_tensors; ++i) {
        const c3_tensor_desc &t = tensors[i];
        if (!t.name) return fail("tensor %d has no name", i);
        const std::string name = t.name;
        const bool nbt = name.size() > 19 && name.compare(name.size() - 19, 19, "num_batches_tracked") == 0;
        if (nbt) continue;  // BatchNorm bookkeeping, unused in eval()
        if (t.dtype != C3_DTYPE_F32) return fail("tensor %s: only float32 parameters are supported", t.name);
        if (t.ndim < 1 || t.ndim > 4 || !t.data) return fail("tensor %s: bad descriptor", t.name);
        TensorView v;
        v.d = (const float *)t.data;
        v.shape.assign(t.shape, t.shape + t.ndim);
        tm[name] = v;
    }
    size_t expected = 0;
    if (m->kind == C3_KIND_PILEUP) {
        TRY(pack_lstm(m, tm, 0, 128, m->C, 32, false));
        TRY(pack_lstm(m, tm, 1, 160, 256, 256, m->lstm2_v2));
        expected = 16;
    } else {
        int cin = m->C;
        if (3 * cin > 32) return fail("full-alignment input_channels %d not supported (3*C must be <= 32)", cin);
        for (int l = 0; l < 9; ++l) {
            TRY(pack_conv(m, tm, l, cin));
            cin = kConvCout[l];
        }
        expected = 54;
    }
    TRY(pack_tail(m, tm));
    expected += 2 + 4 * (size_t)m->nb;
    if (tm.size() != expected) {
        // strict like load_state_dict: report the first unexpected key
        return fail("Unexpected key(s) in state_dict: %zu tensors given, %zu expected", tm.size(), expected);
    }
    m->loaded = true;
    return 0;
}

int c3_predict_device(c3_model *m, const void *x_dev, int x_dtype, int64_t batch, float *y_dev, void *stream) {
    if (!m) return fail("null model");
    if (batch > 0 && (!x_dev || !y_dev)) return fail("null buffer");
    HIP_TRY(hipSetDevice(m->device));
    // NULL is the HIP null stream itself (what torch's default stream is): work queued there is ordered with the
    // caller's other default-stream work.  Mapping NULL to the model's private non-blocking stream would let a
    // following torch op (y.cpu(), an RCCL gather) overtake the kernels.
    return forward_device(m, (hipStream_t)stream, x_dev, x_dtype, batch, y_dev);
}

// one thread per output float: raises the handle's range flag when a probability is not finite
__global__ void rows_finite_kernel(const float *y, int64_t n, uint32_t *flag) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n && (__float_as_uint(y[i]) & 0x7f800000u) == 0x7f800000u) atomicOr(flag, 2u);
}

int c3_predict_device_checked(c3_model *m, const void *x_dev, int x_dtype, int64_t batch, float *y_dev, void *stream) {
    if (!m) return fail("null model");
    if (batch > 0 && (!x_dev || !y_dev)) return fail("null buffer");
    HIP_TRY(hipSetDevice(m->device));
    hipStream_t s = (hipStream_t)stream;
    const bool f16 = m->f16_ok;
    TRY(forward_device(m, s, x_dev, x_dtype, batch, y_dev));
    if (!f16 || batch == 0) return 0;
    if (!m->pin_flag) HIP_TRY(hipHostMalloc((void **)&m->pin_flag, 64, hipHostMallocDefault));
    const int64_t nf = batch * m->row;
    hipLaunchKernelGGL(rows_finite_kernel, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, s, y_dev, nf, m->range_flag);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(m->pin_flag, m->range_flag, 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (*m->pin_flag) {
        fprintf(stderr, "libc3hip: activations beyond the range of the fp16x3 kernels; this handle continues on fp32 matrix instructions\n");
        m->f16_ok = false;
        TRY(forward_device(m, s, x_dev, x_dtype, batch, y_dev));
        HIP_TRY(hipStreamSynchronize(s));
    }
    return 0;
}

int c3_model_range_status(c3_model *m, int *flag_out, int *on_fp32_out) {
    if (!m) return fail("null model");
    HIP_TRY(hipSetDevice(m->device));
    uint32_t f = 0;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(&f, m->range_flag, 4, hipMemcpyDeviceToHost));
    if (flag_out) *flag_out = (int)f;
    if (on_fp32_out) *on_fp32_out = m->f16_ok ? 0 : 1;
    return 0;
}

static int ensure_slot(c3_model *m, HostSlot &sl, size_t xb, size_t yb) {
    if (!sl.ev_h2d) {
        HIP_TRY(hipEventCreateWithFlags(&sl.ev_h2d, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&sl.ev_compute, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&sl.ev_out, hipEventDisableTiming));
    }
    if (!sl.pin_flag) HIP_TRY(hipHostMalloc((void **)&sl.pin_flag, 64, hipHostMallocDefault));
    if (xb > sl.cap_x) {
        if (sl.pin_x) (void)hipHostFree(sl.pin_x);
        if (sl.dev_x) (void)hipFree(sl.dev_x);
        sl.pin_x = sl.dev_x = nullptr, sl.cap_x = 0;
        HIP_TRY(hipHostMalloc(&sl.pin_x, xb, hipHostMallocDefault));
        HIP_TRY(hipMalloc(&sl.dev_x, xb));
        sl.cap_x = xb;
    }
    if (yb > sl.cap_y) {
        if (sl.pin_y) (void)hipHostFree(sl.pin_y);
        if (sl.dev_y) (void)hipFree(sl.dev_y);
        sl.pin_y = nullptr, sl.dev_y = nullptr, sl.cap_y = 0;
        HIP_TRY(hipHostMalloc((void **)&sl.pin_y, yb, hipHostMallocDefault));
        HIP_TRY(hipMalloc((void **)&sl.dev_y, yb));
        sl.cap_y = yb;
    }
    (void)m;
    return 0;
}

static int predict_submit(c3_model *m, const void *x_host, int x_dtype, int64_t batch, float *y_host, int slot, bool src_locked);
int c3_predict_submit(c3_model *m, const void *x_host, int x_dtype, int64_t batch, float *y_host, int slot) {
    return predict_submit(m, x_host, x_dtype, batch, y_host, slot, false);
}
// src_locked: the caller (c3_predict) has page-locked x_host for the duration of ITS call -- a private fact of that call, not
// published in g_registered, so no other thread or handle ever DMAs from pages that are about to be unlocked
static int predict_submit(c3_model *m, const void *x_host, int x_dtype, int64_t batch, float *y_host, int slot, bool src_locked) {
    if (!m) return fail("null model");
    if (slot < 0 || slot >= kHostSlots) return fail("slot must be in [0, %d)", kHostSlots);
    if (batch < 0) return fail("negative batch");
    if (batch > 0 && (!x_host || !y_host)) return fail("null buffer");
    HostSlot &sl = m->slot[slot];
    if (sl.busy) return fail("slot %d still in flight: call c3_predict_wait first", slot);
    HIP_TRY(hipSetDevice(m->device));
    if (!m->loaded) return fail("model has no weights: call c3_model_load first");
    const size_t xb = (size_t)(batch * c3_model_window_bytes(m, x_dtype));
    const size_t yb = (size_t)batch * m->row * sizeof(float);
    if (batch > 0 && m->host_copy_kernel && xb <= kKernelCopyMax && yb <= kKernelCopyMax && !src_locked && !is_registered(x_host, xb)) {
        TRY(ensure_slot(m, sl, (xb + 255) & ~(size_t)255, (yb + 255) & ~(size_t)255));
        memcpy(sl.pin_x, x_host, xb);
        hipLaunchKernelGGL(host_copy_kernel, dim3(128), dim3(256), 0, m->stream, (const uint4 *)sl.pin_x, (uint4 *)sl.dev_x, (xb + 15) / 16,
                           (const uint32_t *)nullptr, (uint32_t *)nullptr);
        HIP_TRY(hipGetLastError());
        const bool f16 = m->f16_ok;
        TRY(forward_device(m, m->stream, sl.dev_x, x_dtype, batch, sl.dev_y));
        hipLaunchKernelGGL(host_copy_kernel, dim3(32), dim3(256), 0, m->stream, (const uint4 *)sl.dev_y, (uint4 *)sl.pin_y, (yb + 15) / 16,
                           (const uint32_t *)m->range_flag, sl.pin_flag);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(sl.ev_out, m->stream));
        sl.used_f16 = f16;
    } else
    if (batch > 0) {
        // the slot becomes busy only once everything has been queued: a failure on the way leaves it free
        TRY(ensure_slot(m, sl, xb, yb));
        TRY(stage_h2d(sl.dev_x, sl.pin_x, x_host, xb, m->h2d_stream, src_locked));
        HIP_TRY(hipEventRecord(sl.ev_h2d, m->h2d_stream));
        HIP_TRY(hipStreamWaitEvent(m->stream, sl.ev_h2d, 0));
        const bool f16 = m->f16_ok;
        TRY(forward_device(m, m->stream, sl.dev_x, x_dtype, batch, sl.dev_y));
        HIP_TRY(hipEventRecord(sl.ev_compute, m->stream));
        HIP_TRY(hipStreamWaitEvent(m->d2h_stream, sl.ev_compute, 0));
        HIP_TRY(hipMemcpyAsync(sl.pin_y, sl.dev_y, yb, hipMemcpyDeviceToHost, m->d2h_stream));
        HIP_TRY(hipMemcpyAsync(sl.pin_flag, m->range_flag, 4, hipMemcpyDeviceToHost, m->d2h_stream));
        HIP_TRY(hipEventRecord(sl.ev_out, m->d2h_stream));
        sl.used_f16 = f16;
    }
    sl.y_host = y_host, sl.y_bytes = yb, sl.batch = batch, sl.x_dtype = x_dtype, sl.busy = true;
    return 0;
}

int c3_predict_wait(c3_model *m, int slot) {
    if (!m) return fail("null model");
    if (slot < 0 || slot >= kHostSlots) return fail("slot must be in [0, %d)", kHostSlots);
    HostSlot &sl = m->slot[slot];
    if (!sl.busy) return fail("slot %d has nothing in flight", slot);
    sl.busy = false;
    if (sl.y_bytes == 0) return 0;
    HIP_TRY(hipEventSynchronize(sl.ev_out));
    if (sl.used_f16) {
        // Safety net of the fp16x3 products, per batch: what matters is how THIS slot's rows were computed, not what the
        // handle does now (another slot's wait may have switched it to fp32 while this batch was in flight).  An
        // activation beyond the fp16 range (|x| >= 65504; never seen) surfaces as inf / NaN rows or as the
        // range flag (sticky: an overflow in any earlier fp16x3 batch also lands here, which only costs a re-run).
        const uint32_t *u = reinterpret_cast<const uint32_t *>(sl.pin_y);
        bool bad = *sl.pin_flag != 0;  // a conv stage produced a value near the fp16 range (kF16Range): its consumers may have overflowed
        for (size_t i = 0, n = sl.y_bytes / 4; i < n; ++i) bad |= (u[i] & 0x7f800000u) == 0x7f800000u;
        if (bad) {
            if (m->f16_ok)
                fprintf(stderr, "libc3hip: activations beyond the range of the fp16x3 kernels; this handle continues on fp32 matrix instructions\n");
            m->f16_ok = false;
            TRY(forward_device(m, m->stream, sl.dev_x, sl.x_dtype, sl.batch, sl.dev_y));
            HIP_TRY(hipMemcpyAsync(sl.pin_y, sl.dev_y, sl.y_bytes, hipMemcpyDeviceToHost, m->stream));
            HIP_TRY(hipStreamSynchronize(m->stream));
        }
    }
    memcpy(sl.y_host, sl.pin_y, sl.y_bytes);
    return 0;
}

// The synchronous call of the reference loop (_torch_predict: H2D, forward, D2H one after the other,
// clair3/CallVariantsFromCffi.py:48-52).  A batch well beyond one chunk (256 full-alignment / 4096 pileup windows; env
// C3HIP_PREDICT_CHUNK) is cut into chunks that travel through the submit / wait ring: the staging copy and H2D transfer of chunk
// i + 1 and the D2H transfer of chunk i - 1 run under the kernels of chunk i, so the caller's ONE blocking call costs little more
// than the kernels of the whole batch.  Rows do not depend on the cut (a window's row is independent of the batch it travels in).
static int64_t predict_chunk(const c3_model *m) {
    static const int64_t env = getenv("C3HIP_PREDICT_CHUNK") ? atoll(getenv("C3HIP_PREDICT_CHUNK")) : -1;
    if (env >= 0) return env;  // 0: never cut
    return m->kind == C3_KIND_PILEUP ? 4096 : 256;
}

int c3_predict(c3_model *m, const void *x_host, int x_dtype, int64_t batch, float *y_host) {
    if (!m) return fail("null model");
    const int64_t chunk = predict_chunk(m);
    if (chunk <= 0 || batch < 2 * chunk) {
        TRY(c3_predict_submit(m, x_host, x_dtype, batch, y_host, 0));
        return c3_predict_wait(m, 0);
    }
    constexpr int kRing = 3;
    const int64_t wbytes = c3_model_window_bytes(m, x_dtype);
    // A blocking call cannot hide its staging copy behind a previous batch, and that copy (pageable -> pinned, ~16 GB/s with the
    // staging pool) is as long as the kernels of a full-alignment batch.  So the caller's pages are page-locked for the duration of
    // the call (~0.1 ms per 24 MB, hipHostRegister) and the DMA engine reads them directly; if the range cannot be registered
    // (e.g. a read-only mapping) the chunks go through the staging buffer as before.
    static const bool want_reg = !getenv("C3HIP_PREDICT_REGISTER") || atoi(getenv("C3HIP_PREDICT_REGISTER")) != 0;
    const size_t xbytes = (size_t)(batch * wbytes);
    void *reg_base = nullptr;
    if (want_reg && xbytes >= ((size_t)4 << 20) && !is_registered(x_host, xbytes)) {
        const uintptr_t lo = (uintptr_t)x_host & ~(uintptr_t)4095, hi = ((uintptr_t)x_host + xbytes + 4095) & ~(uintptr_t)4095;
        (void)hipSetDevice(m->device);
        if (hipHostRegister((void *)lo, hi - lo, hipHostRegisterDefault) == hipSuccess) {
            reg_base = (void *)lo;
        } else {
            (void)hipGetLastError();  // not fatal: staged copy
        }
    }
    int64_t n_sub = 0, n_done = 0;  // chunks submitted / waited for
    int rc = 0;
    // chunk sizes grow (chunk / 2, chunk, 2 chunk, 4 chunk, 4 chunk, ...): the kernels start after a SHORT first transfer, the
    // later, longer transfers hide under ever longer kernel runs, and big chunks fill the chip better than small ones; a tail
    // shorter than half the next size joins the last chunk
    int64_t next = std::max<int64_t>(chunk / 2, 1);
    for (int64_t off = 0; off < batch && rc == 0; ++n_sub) {
        int64_t take = std::min(next, batch - off);
        if (batch - off - take < next / 2 || batch - off - take < chunk / 2) take = batch - off;
        take = std::min(take, max_microbatch(m));
        if (n_sub - n_done == kRing) rc = c3_predict_wait(m, (int)(n_done++ % kRing));
        if (rc == 0)
            rc = predict_submit(m, (const char *)x_host + off * wbytes, x_dtype, take, y_host + off * m->row, (int)(n_sub % kRing),
                                reg_base != nullptr);
        if (rc != 0) break;
        off += take;
        next = std::min(2 * next, 4 * chunk);
    }
    const std::string first_error = rc != 0 ? g_err : std::string();
    for (; n_done < n_sub; ++n_done) {  // drain, also after an error: no slot stays busy behind a failed call
        const int r = c3_predict_wait(m, (int)(n_done % kRing));
        if (rc == 0) rc = r;
    }
    if (reg_base) (void)hipHostUnregister(reg_base);  // every chunk has been waited for: nothing reads the pages any more
    if (!first_error.empty()) g_err = first_error;
    return rc;
}

int c3_predict_pileup_region(c3_model *m, const void *region_host, int x_dtype, int64_t n_cols, const int32_t *starts_host,
                             int64_t batch, float *y_host) {
    if (!m) return fail("null model");
    if (m->kind != C3_KIND_PILEUP) return fail("c3_predict_pileup_region needs a pileup model");
    if (batch < 0 || n_cols < 0) return fail("negative size");
    if (batch == 0) return 0;
    if (!region_host || !starts_host || !y_host) return fail("null buffer");
    if (x_dtype != C3_DTYPE_I8 && x_dtype != C3_DTYPE_I32 && x_dtype != C3_DTYPE_I64)
        return fail("pileup regions must be int8, int32 or int64 / size_t (got dtype %d)", x_dtype);
    // int64 = plp_data.matrix itself (size_t counts, src/clair3_pileup.h:113): narrowed to int32 on its way into the staging
    // buffer, which is what the reference's PIPE mode feeds the model (CreateTensorPileupFromCffi.py:143-146 -> int32 windows)
    const bool narrow = x_dtype == C3_DTYPE_I64;
    if (narrow) x_dtype = C3_DTYPE_I32;
    for (int64_t i = 0; i < batch; ++i)
        if (starts_host[i] < 0 || (int64_t)starts_host[i] + m->positions > n_cols)
            return fail("window %lld starts at column %d: outside the %lld-column region", (long long)i, starts_host[i], (long long)n_cols);
    HostSlot &sl = m->slot[0];
    if (sl.busy) return fail("slot 0 still in flight: call c3_predict_wait first");
    HIP_TRY(hipSetDevice(m->device));
    if (!m->loaded) return fail("model has no weights: call c3_model_load first");
    const size_t item = x_dtype == C3_DTYPE_I32 ? 4 : 1;
    const size_t rb = ((size_t)n_cols * m->C * item + 255) & ~(size_t)255;
    const size_t sb = (size_t)batch * sizeof(int32_t);
    const size_t yb = (size_t)batch * m->row * sizeof(float);
    TRY(ensure_slot(m, sl, rb + sb, yb));
    if (narrow) {
        const int64_t *src = static_cast<const int64_t *>(region_host);
        int32_t *dst = static_cast<int32_t *>(sl.pin_x);
        for (size_t i = 0, e = (size_t)n_cols * m->C; i < e; ++i) dst[i] = (int32_t)src[i];
    } else {
        memcpy(sl.pin_x, region_host, (size_t)n_cols * m->C * item);
    }
    memcpy((char *)sl.pin_x + rb, starts_host, sb);
    HIP_TRY(hipMemcpyAsync(sl.dev_x, sl.pin_x, rb + sb, hipMemcpyHostToDevice, m->stream));
    TRY(forward_device(m, m->stream, sl.dev_x, x_dtype, batch, sl.dev_y, (const int32_t *)((char *)sl.dev_x + rb)));
    HIP_TRY(hipMemcpyAsync(sl.pin_y, sl.dev_y, yb, hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    memcpy(y_host, sl.pin_y, yb);
    return 0;
}

int c3_outcome_maxima(c3_model *m, const float *y_host, int64_t batch, const uint8_t *ref21_host, float *maxp_host,
                      int32_t *argmax_host, uint8_t *early_host) {
    if (!m) return fail("null model");
    if (batch < 0) return fail("negative batch");
    if (batch == 0) return 0;
    if (!y_host || !ref21_host || !maxp_host || !argmax_host || !early_host) return fail("null buffer");
    for (int64_t i = 0; i < batch; ++i)
        if (ref21_host[i] != 0 && ref21_host[i] != 4 && ref21_host[i] != 7 && ref21_host[i] != 9)
            return fail("row %lld: reference gt21 index %d is not one of AA=0, CC=4, GG=7, TT=9", (long long)i, (int)ref21_host[i]);
    HIP_TRY(hipSetDevice(m->device));
    const size_t yb = (size_t)batch * m->nout * sizeof(float), rb = ((size_t)batch + 255) & ~(size_t)255;
    const size_t mb = (size_t)batch * kDecodeClasses * sizeof(float);
    const size_t total = yb + rb + 2 * mb + rb;
    if (m->decode_bytes < total) {
        if (m->decode_dev) (void)hipFree(m->decode_dev);
        m->decode_dev = nullptr, m->decode_bytes = 0;
        HIP_TRY(hipMalloc(&m->decode_dev, total));
        m->decode_bytes = total;
    }
    char *base = (char *)m->decode_dev;
    float *y = (float *)base;
    uint8_t *ref = (uint8_t *)(base + yb);
    float *maxp = (float *)(base + yb + rb);
    int32_t *arg = (int32_t *)(base + yb + rb + mb);
    uint8_t *early = (uint8_t *)(base + yb + rb + 2 * mb);
    HIP_TRY(hipMemcpyAsync(y, y_host, yb, hipMemcpyHostToDevice, m->stream));
    HIP_TRY(hipMemcpyAsync(ref, ref21_host, (size_t)batch, hipMemcpyHostToDevice, m->stream));
    DecodeParams dp{y, m->nout, ref, maxp, arg, early, nullptr, (int)batch, m->nout == 90 ? 1 : 0};
    hipLaunchKernelGGL(outcome_maxima_kernel<false>, dim3((unsigned)((batch + 3) / 4)), dim3(256), 0, m->stream, dp);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(maxp_host, maxp, mb, hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipMemcpyAsync(argmax_host, arg, mb, hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipMemcpyAsync(early_host, early, (size_t)batch, hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    return 0;
}

int c3_decode_columns(c3_model *m, const float *y_host, int64_t batch, float *rows_host) {
    if (!m) return fail("null model");
    if (batch < 0) return fail("negative batch");
    if (batch == 0) return 0;
    if (!y_host || !rows_host) return fail("null buffer");
    HIP_TRY(hipSetDevice(m->device));
    const int wide = m->nout + kDecodeCols;
    const size_t total = (size_t)batch * wide * sizeof(float);
    if (m->decode_bytes < total) {
        if (m->decode_dev) (void)hipFree(m->decode_dev);
        m->decode_dev = nullptr, m->decode_bytes = 0;
        HIP_TRY(hipMalloc(&m->decode_dev, total));
        m->decode_bytes = total;
    }
    float *rows = (float *)m->decode_dev;
    HIP_TRY(hipMemcpy2DAsync(rows, (size_t)wide * sizeof(float), y_host, (size_t)m->nout * sizeof(float),
                             (size_t)m->nout * sizeof(float), (size_t)batch, hipMemcpyHostToDevice, m->stream));
    DecodeParams dp{rows, wide, nullptr, nullptr, nullptr, nullptr, rows + m->nout, (int)batch, m->nout == 90 ? 1 : 0};
    hipLaunchKernelGGL(outcome_maxima_kernel<true>, dim3((unsigned)((batch + 3) / 4)), dim3(256), 0, m->stream, dp);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(rows_host, rows, total, hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    return 0;
}

// ---- the gather of the sharded job on RCCL (c3_comm.h) ----
int c3_comm_unique_id(void *id128) {
    if (!id128) return fail("null buffer");
    RcclApi &r = RcclApi::get();
    if (!r.load()) return fail("%s", r.error.c_str());
    static_assert(sizeof(ncclUniqueId) == 128, "c3_comm_unique_id hands out 128 bytes");
    ncclUniqueId id;
    const ncclResult_t rc = r.GetUniqueId(&id);
    if (rc != ncclSuccess) return fail("ncclGetUniqueId failed: %s", r.GetErrorString(rc));
    memcpy(id128, &id, 128);
    return 0;
}

c3_comm *c3_comm_create(const void *id128, int rank, int world, int device) {
    if (world < 1 || rank < 0 || rank >= world) {
        fail("bad rank %d of %d", rank, world);
        return nullptr;
    }
    if (hipSetDevice(device) != hipSuccess) {
        fail("hipSetDevice(%d) failed", device);
        return nullptr;
    }
    c3_comm *c = new c3_comm();
    c->rank = rank, c->world = world, c->device = device;
    if (world == 1) return c;  // nothing to talk to: c3_gather_rows is a device copy
    if (!id128) {
        fail("null unique id");
        delete c;
        return nullptr;
    }
    RcclApi &r = RcclApi::get();
    if (!r.load()) {
        fail("%s", r.error.c_str());
        delete c;
        return nullptr;
    }
    ncclUniqueId id;
    memcpy(&id, id128, 128);
    const ncclResult_t rc = r.CommInitRank(&c->nccl, world, id, rank);
    if (rc != ncclSuccess) {
        fail("ncclCommInitRank failed: %s", r.GetErrorString(rc));
        delete c;
        return nullptr;
    }
    return c;
}

int c3_comm_destroy(c3_comm *c) {
    if (!c) return 0;
    if (c->nccl) (void)RcclApi::get().CommDestroy(c->nccl);
    delete c;
    return 0;
}

int c3_gather_rows(c3_comm *c, const float *rows_dev, int row_floats, const int64_t *counts, float *all_dev, int dst, void *stream) {
    if (!c || !counts) return fail("null argument");
    if (dst < 0 || dst >= c->world || row_floats <= 0) return fail("bad arguments (dst %d of %d ranks, %d floats per row)", dst, c->world, row_floats);
    for (int r = 0; r < c->world; ++r)
        if (counts[r] < 0) return fail("negative row count for rank %d", r);
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    const size_t mine = (size_t)counts[c->rank] * row_floats;
    if (mine && !rows_dev) return fail("null rows");
    if (c->rank == dst && !all_dev) return fail("the destination rank needs the gathered buffer");
    if (c->world == 1) {
        if (mine && all_dev != rows_dev) HIP_TRY(hipMemcpyAsync(all_dev, rows_dev, mine * sizeof(float), hipMemcpyDeviceToDevice, s));
        return 0;
    }
    RcclApi &r = RcclApi::get();
    int rc = (int)r.GroupStart();
    if (rc) return fail("ncclGroupStart failed: %s", r.GetErrorString((ncclResult_t)rc));
    if (c->rank == dst) {
        size_t off = 0;
        for (int src = 0; src < c->world && !rc; ++src) {
            const size_t n = (size_t)counts[src] * row_floats;
            if (src == dst) {
                if (n && all_dev + off != rows_dev) {
                    hipError_t e = hipMemcpyAsync(all_dev + off, rows_dev, n * sizeof(float), hipMemcpyDeviceToDevice, s);
                    if (e != hipSuccess) rc = -1;
                }
            } else if (n) {
                rc = (int)r.Recv(all_dev + off, n, ncclFloat32, src, c->nccl, s);
            }
            off += n;
        }
    } else if (mine) {
        rc = (int)r.Send(rows_dev, mine, ncclFloat32, dst, c->nccl, s);
    }
    const ncclResult_t rc2 = r.GroupEnd();
    if (rc > 0) return fail("ncclSend/ncclRecv failed: %s", r.GetErrorString((ncclResult_t)rc));
    if (rc < 0) return fail("device copy inside the gather failed");
    if (rc2 != ncclSuccess) return fail("ncclGroupEnd failed: %s", r.GetErrorString(rc2));
    return 0;
}

int c3_comm_count(c3_comm *c, int *ranks_out, int *rank_out) {
    if (!c || !ranks_out) return fail("null argument");
    if (!c->nccl) {  // world == 1: no communicator
        *ranks_out = c->world;
        if (rank_out) *rank_out = c->rank;
        return 0;
    }
    RcclApi &r = RcclApi::get();
    ncclResult_t rc = r.CommCount(c->nccl, ranks_out);
    if (rc != ncclSuccess) return fail("ncclCommCount failed: %s", r.GetErrorString(rc));
    if (rank_out) {
        rc = r.CommUserRank(c->nccl, rank_out);
        if (rc != ncclSuccess) return fail("ncclCommUserRank failed: %s", r.GetErrorString(rc));
    }
    return 0;
}

int c3_comm_abort(c3_comm *c) {
    if (!c) return 0;
    if (c->nccl) {
        RcclApi &r = RcclApi::get();
        const ncclResult_t rc = r.CommAbort(c->nccl);
        c->nccl = nullptr;
        c->world = 1;  // whatever is asked of this handle from now on is local
        if (rc != ncclSuccess) return fail("ncclCommAbort failed: %s", r.GetErrorString(rc));
    }
    return 0;
}

int c3_stream_wait(void *stream, int device, int timeout_ms) {
    HIP_TRY(hipSetDevice(device));
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t e = hipStreamQuery((hipStream_t)stream);
        if (e == hipSuccess) return 0;
        if (e != hipErrorNotReady) return fail("hipStreamQuery: %s", hipGetErrorString(e));
        if (timeout_ms >= 0 &&
            std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() >= timeout_ms) {
            (void)hipGetLastError();
            g_err = "timeout";
            return 1;
        }
        struct timespec ts = {0, 50000};  // 50 us
        nanosleep(&ts, nullptr);
    }
}

int c3_host_register(void *p, size_t bytes) {
    if (!p || !bytes) return fail("null buffer");
    HIP_TRY(hipHostRegister(p, bytes, hipHostRegisterDefault));
    std::lock_guard<std::mutex> lk(g_registered_mu);
    g_registered.push_back({(const char *)p, bytes});
    return 0;
}

int c3_host_unregister(void *p) {
    {
        std::lock_guard<std::mutex> lk(g_registered_mu);
        bool found = false;
        for (size_t i = 0; i < g_registered.size(); ++i)
            if (g_registered[i].p == (const char *)p) {
                g_registered.erase(g_registered.begin() + i);
                found = true;
                break;
            }
        if (!found) return fail("buffer was not registered with c3_host_register");
    }
    HIP_TRY(hipHostUnregister(p));
    return 0;
}

int c3_model_describe(c3_model *m, char *buf, int n) {
    if (!m || !buf || n <= 0) return fail("null argument");
    if (m->kind == C3_KIND_PILEUP)
        snprintf(buf, (size_t)n, "other_handles_active=%d lstm1=%s proj2=%s lstm2=%s on_fp32_fallback=%d", (int)m->concurrent,
                 m->choice_lstm1, m->choice_proj2, m->choice_lstm2, (int)!m->f16_ok);
    else
        snprintf(buf, (size_t)n, "other_handles_active=%d conv_stack=%s on_fp32_fallback=%d", (int)m->concurrent, m->choice_fa,
                 (int)!m->f16_ok);
    return 0;
}

int c3_model_synchronize(c3_model *m) {
    if (!m) return fail("null model");
    HIP_TRY(hipStreamSynchronize(m->stream));
    return 0;
}

int c3_model_destroy(c3_model *m) {
    if (!m) return 0;
    forget_activity(m);
    (void)hipSetDevice(m->device);
    (void)hipDeviceSynchronize();
    free_workspace(m);
    float *ws[] = {m->proj_w[0], m->proj_w[1], m->proj_b[0], m->proj_b[1], m->whh[0], m->whh[1], m->whh16[0], m->whh16[1],
                   m->l4_w, m->l4_b, m->w5t, m->b5, m->wh, m->bh, m->zeros, m->l1_wih, m->l1_wih16, m->l1_bias,
                   m->conv1_wfrag, m->conv1_wfrag16, m->w5f, m->whf, m->bh48, m->proj2_frag, m->l4_w3, m->proj2_w3, m->proj2_pw, m->proj2_pw32, m->proj2_pwr};
    for (float *p : ws)
        if (p) (void)hipFree(p);
    if (m->decode_dev) (void)hipFree(m->decode_dev);
    if (m->range_flag) (void)hipFree(m->range_flag);
    if (m->lstm_trace_dev) (void)hipFree(m->lstm_trace_dev);
    if (m->pin_flag) (void)hipHostFree(m->pin_flag);
    for (int l = 0; l < 9; ++l) {
        if (m->conv_w[l]) (void)hipFree(m->conv_w[l]);
        if (m->conv_b[l]) (void)hipFree(m->conv_b[l]);
        if (m->wino_v[l]) (void)hipFree(m->wino_v[l]);
        if (m->conv_w3[l]) (void)hipFree(m->conv_w3[l]);
        if (m->pconv_w[l]) (void)hipFree(m->pconv_w[l]);
        if (m->wino_v16[l]) (void)hipFree(m->wino_v16[l]);
    }
    for (auto &sl : m->slot) {
        if (sl.pin_x) (void)hipHostFree(sl.pin_x);
        if (sl.pin_y) (void)hipHostFree(sl.pin_y);
        if (sl.pin_flag) (void)hipHostFree(sl.pin_flag);
        if (sl.dev_x) (void)hipFree(sl.dev_x);
        if (sl.dev_y) (void)hipFree(sl.dev_y);
        if (sl.ev_h2d) (void)hipEventDestroy(sl.ev_h2d);
        if (sl.ev_compute) (void)hipEventDestroy(sl.ev_compute);
        if (sl.ev_out) (void)hipEventDestroy(sl.ev_out);
    }
    for (auto &r : m->recs) {
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    if (m->stream) (void)hipStreamDestroy(m->stream);
    if (m->h2d_stream) (void)hipStreamDestroy(m->h2d_stream);
    if (m->d2h_stream) (void)hipStreamDestroy(m->d2h_stream);
    delete m;
    return 0;
}

int c3_debug_keep_activations(c3_model *m, int enable) {
    if (!m) return fail("null model");
    HIP_TRY(hipSetDevice(m->device));
    if (m->keep != (enable != 0)) {
        HIP_TRY(hipStreamSynchronize(m->stream));
        free_workspace(m);
        m->keep = enable != 0;
    }
    return 0;
}

int c3_debug_fetch(c3_model *m, const char *name, float *host_out, int64_t n_floats) {
    if (!m || !name || !host_out) return fail("null argument");
    HIP_TRY(hipSetDevice(m->device));
    if (m->last_n <= 0) return fail("nothing has been predicted yet");
    const std::string s = name;
    const float *src = nullptr;
    int64_t n = 0;
    if (m->kind == C3_KIND_PILEUP) {
        if (s == "lstm1_out") src = m->h1, n = m->last_n * m->positions * 256;
        else if (s == "lstm2_out") src = m->h2, n = m->last_n * m->positions * 320;
        else if (s == "gx1") src = m->gx1, n = m->last_n * m->positions * 1024;
        else if (s == "gx2") src = m->gx2, n = m->last_n * m->positions * 1280;
    } else {
        int hh[10], ww[10];
        fa_geometry(m, hh, ww);
        if (s.size() == 4 && s.compare(0, 3, "act") == 0 && s[3] >= '0' && s[3] <= '8') {
            if (!m->keep) return fail("activations are recycled: enable c3_debug_keep_activations first");
            const int l = s[3] - '0';
            src = m->act[l], n = m->last_n * hh[l + 1] * ww[l + 1] * kConvCout[l];
        } else if (s == "spp") src = m->spp, n = m->last_n * m->K4;
    }
    if (s == "l4_out") {
        if (!m->keep) return fail("l4_out is only written with c3_debug_keep_activations enabled");
        src = m->l4dbg, n = m->last_n * m->FC;
    }
    if (!src) return fail("unknown debug tensor \"%s\"", name);
    if (n != n_floats) return fail("debug tensor %s has %lld floats, caller expects %lld", name, (long long)n, (long long)n_floats);
    HIP_TRY(hipDeviceSynchronize());
    if (m->last_planes && ((m->kind == C3_KIND_FULL_ALIGNMENT && s.compare(0, 3, "act") == 0) || (m->kind == C3_KIND_PILEUP && s == "lstm1_out"))) {
        // the layer holds plane activations (c3_conv3.h): hand the caller the fp32 values they stand for
        const int C = m->kind == C3_KIND_PILEUP ? 256 : kConvCout[s[3] - '0'];
        float *tmp = nullptr;
        HIP_TRY(hipMalloc((void **)&tmp, (size_t)n * sizeof(float)));
        hipLaunchKernelGGL(planes_to_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (const void *)src, tmp, n / C, C);
        hipError_t e = hipMemcpy(host_out, tmp, (size_t)n * sizeof(float), hipMemcpyDeviceToHost);
        (void)hipFree(tmp);
        if (e != hipSuccess) return fail("debug fetch copy failed: %s", hipGetErrorString(e));
        return 0;
    }
    HIP_TRY(hipMemcpy(host_out, src, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

int c3_profile_enable(c3_model *m, int enable) {
    if (!m) return fail("null model");
    m->prof = enable != 0;
    return 0;
}

int c3_profile_reset(c3_model *m) {
    if (!m) return fail("null model");
    HIP_TRY(hipSetDevice(m->device));
    HIP_TRY(hipDeviceSynchronize());
    for (auto &r : m->recs) {
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    m->recs.clear();
    return 0;
}

int c3_profile_read(c3_model *m, c3_kernel_stat *out, int max_entries) {
    if (!m || (!out && max_entries > 0)) {
        fail("null argument");
        return -1;
    }
    if (hipSetDevice(m->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
        fail("device synchronize failed");
        return -1;
    }
    std::vector<std::string> order;
    std::map<std::string, c3_kernel_stat> agg;
    for (auto &r : m->recs) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) continue;
        auto it = agg.find(r.name);
        if (it == agg.end()) {
            c3_kernel_stat st;
            memset(&st, 0, sizeof(st));
            snprintf(st.name, sizeof(st.name), "%s", r.name.c_str());
            it = agg.insert({r.name, st}).first;
            order.push_back(r.name);
        }
        it->second.launches += 1;
        it->second.total_ms += ms;
        it->second.flops += r.flops;
        it->second.bytes += r.bytes;
        it->second.mfma_flops += r.mfma_flops;
        it->second.mfma_peak_tflops = std::max(it->second.mfma_peak_tflops, r.mfma_peak);
    }
    int n = 0;
    for (auto &k : order) {
        if (n >= max_entries) break;
        out[n++] = agg[k];
    }
    return (int)order.size();
}

}  // extern "C"

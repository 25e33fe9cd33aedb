// c3_model.hip -- the C ABI of include/c3hip.h: model handles (create / geometry / load / destroy), the device-resident entry
// points and their range guard.  The units it is made of are listed in c3_model.h.
#include "c3_model.h"
#include "c3_pack.h"
#include "c3_forward.h"
#include "c3_hostring.h"
#include "c3_comm.h"
#include "c3_debug.h"
#include "c3_rows.h"

extern "C" {

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

int c3_device_pci_bus_id(int device, char *buf, int buf_bytes) {
    if (!buf || buf_bytes < 16) return fail("c3_device_pci_bus_id: buffer of at least 16 bytes needed");
    HIP_TRY(hipDeviceGetPCIBusId(buf, buf_bytes, device));
    for (char *c = buf; *c; ++c) *c = (char)tolower((unsigned char)*c);  // sysfs spells the address in lower case
    return 0;
}

c3_model *c3_model_create(int kind, int in_channels, int add_indel_length, int device) {
    if (kind != C3_KIND_PILEUP && kind != C3_KIND_FULL_ALIGNMENT) {
        fail("unknown model kind %d", kind);
        return nullptr;
    }
    if (kind == C3_KIND_PILEUP && (in_channels < 1 || in_channels > 4 * kFusedKS)) {
        fail("pileup input_channels must be in [1,%d], got %d", 4 * kFusedKS, in_channels);
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
    if (const char *e = getenv("C3HIP_STREAM_PRIORITY")) m->stream_priority = atoi(e);
    if (const char *e = getenv("C3HIP_LAZY_H2D_STREAM")) m->lazy_h2d = atoi(e) != 0;
    if (new_stream(m, &m->stream) != hipSuccess || (!m->lazy_h2d && new_stream(m, &m->h2d_stream) != hipSuccess)) {
        fail("hipStreamCreate failed");
        delete m;
        return nullptr;
    }
    if (getenv("C3HIP_KEEP_ACTIVATIONS")) m->keep = true;
    if (const char *e = getenv("C3HIP_FP32")) {  // an explicit choice: 1 = start on the fp32-MFMA forms (what the range guard falls back to), 0 = fp16x3, no automatism
        m->f16_ok = atoi(e) == 0, m->precision_forced = true;
        if (!m->f16_ok) m->precision = "fp32-forced";
    }
    if (const char *e = getenv("C3HIP_AUTO_FP32")) m->auto_fp32_at = (float)atof(e);
    if (const char *e = getenv("C3HIP_CONV1_FUSED")) m->conv1_fused = atoi(e) != 0;
    if (const char *e = getenv("C3HIP_WINO")) m->wino = atoi(e);
    if (const char *e = getenv("C3HIP_DUO")) m->duo = atoi(e);
    if (const char *e = getenv("C3HIP_SPP_FUSED")) m->spp_fused = atoi(e) != 0;
    m->tail_fused = kind == C3_KIND_PILEUP;
    if (const char *e = getenv("C3HIP_TAIL_FUSED")) m->tail_fused = atoi(e) != 0;
    if (const char *e = getenv("C3HIP_HALF_TILES")) m->half_tiles = atoi(e) != 0;
    // the ring's FC chain on its own stream: on for full alignment (same-box A/B, profiles/r06_h_ab_tail_stream.txt: ring 680 - 685 k -> 696 k windows/s at
    // B = 256, 687 -> 705 - 709 k over the driver's 100 steps, 768 -> 778 - 780 k at B = 1000), off for the pileup network (4.09 - 4.14 M -> 4.03 M: a chain
    // beside the next batch's latency-bound LSTM1 slows the recurrence more than it hides)
    // (round 6, later: OFF for both.  The +1.5 - 2 % of that A/B was one placement of the handle's streams on the runtime's four hardware queues;
    // another placement of the same streams loses 15 %, and a handle with FEWER streams is what holds everywhere: see lane_h2d in c3_model.h and
    // profiles/r06_o_ring_streams_and_hardware_queues.txt)
    m->tail_split = false;
    // two lanes for the ring (c3_model.h Lane): the kind's default follows the same-box A/B of profiles/r06_i_ab_ring_lanes.txt
    // (one MI355X, alternating: full alignment ring 728 - 732 k -> 768 - 775 k windows/s at B = 256 but 807 - 809 k -> 768 - 778 k at B = 1000; pileup
    // 4.24 M -> 4.32 - 4.33 M at B = 1024): more than one lane, for batches that do not fill the chip by themselves
    // ... and three against two lanes (run 11, another box): full alignment ring 703 - 704 k (one lane) -> 745 k (two) -> 769 - 770 k (three) at B = 256;
    // pileup 4.11 - 4.14 M -> 4.23 - 4.26 M (two) -> 4.07 - 4.08 M (three: three recurrences side by side starve each other)
    m->ring_lanes = kind == C3_KIND_FULL_ALIGNMENT ? 3 : 2;
    m->lane_max_batch = kind == C3_KIND_FULL_ALIGNMENT ? 512 : 1024;
    if (const char *e = getenv("C3HIP_RING_LANES")) m->ring_lanes = std::min(std::max(atoi(e), 1), (int)c3_model::kMaxLanes);
    if (const char *e = getenv("C3HIP_RING_LANES_MAX_BATCH")) m->lane_max_batch = atoll(e);
    if (const char *e = getenv("C3HIP_TAIL_STREAM")) m->tail_split = atoi(e) != 0;
    if (const char *e = getenv("C3HIP_LANE_SHARING")) m->lane_sharing_ok = atoi(e) != 0;
    if (const char *e = getenv("C3HIP_LANE_ORDER")) m->lane_by_slot = strcmp(e, "slot") == 0;
    if (const char *e = getenv("C3HIP_LANE_H2D")) m->lane_h2d = atoi(e) != 0;
    if (const char *e = getenv("C3HIP_HOST_COPY_KERNEL")) m->host_copy_kernel = atoi(e);
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
            m->wg_slots = 2 * prop.multiProcessorCount;
    }
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
    HIP_TRY(hipDeviceSynchronize());
    free_all_workspaces(m);
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
        m->lstm_wmax = m->lstm_hh_norm = 0.f;
        TRY(pack_lstm(m, tm, 0, 128, m->C, 32));
        TRY(pack_lstm(m, tm, 1, 160, 256, 256));
        expected = 16;
        // precision escalation without a user switch (c3_model.h): decided once per load, from the weights alone
        if (!m->precision_forced) {
            const bool up = m->auto_fp32_at > 0.f && m->lstm_wmax >= m->auto_fp32_at;
            if (up)
                fprintf(stderr, "libc3hip: LSTM weights reach |w| = %.3g (>= %.3g): this pileup handle runs on the fp32 matrix instructions "
                                "(C3HIP_FP32=0 keeps the fp16x3 kernels)\n", (double)m->lstm_wmax, (double)m->auto_fp32_at);
            // (new weights, new decision: a reload also ends what the range guard decided for the weights before)
            m->f16_ok = !up, m->precision = up ? "fp32-auto" : "fp16x3";
        }
    } else {
        int cin = m->C;
        if (3 * cin > 32) return fail("full-alignment input_channels %d not supported (3*C must be <= 32)", cin);
        FaChannelExps ex;
        TRY(fa_channel_exps(tm, ex));
        for (int l = 0; l < 9; ++l) {
            TRY(pack_conv(m, tm, l, cin, ex));
            m->act_exp[l] = *ex.out_of(l);
            cin = kConvCout[l];
        }
        expected = 54;
        TRY(pack_tail(m, tm, &ex.stage[2]));
    }
    if (m->kind == C3_KIND_PILEUP) TRY(pack_tail(m, tm));
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
    TRY(use_lane(m, 0));  // (calls on one handle must not overlap: the device-resident entries always work in the first lane)
    return forward_device(m, (hipStream_t)stream, x_dev, x_dtype, batch, y_dev);
}

int c3_predict_device_checked(c3_model *m, const void *x_dev, int x_dtype, int64_t batch, float *y_dev, void *stream) {
    if (!m) return fail("null model");
    if (batch > 0 && (!x_dev || !y_dev)) return fail("null buffer");
    HIP_TRY(hipSetDevice(m->device));
    hipStream_t s = (hipStream_t)stream;
    const bool f16 = m->f16_ok;
    TRY(use_lane(m, 0));
    TRY(forward_device(m, s, x_dev, x_dtype, batch, y_dev));
    if (!f16 || batch == 0) return 0;
    if (!m->pin_flag) {
        HIP_TRY(hipHostMalloc((void **)&m->pin_flag, 4096, hipHostMallocDefault));  // (a whole page: MADV_DONTFORK works on pages)
        keep_out_of_children(m->pin_flag, 4096);
    }
    const int64_t nf = batch * m->row;
    hipLaunchKernelGGL(rows_finite_kernel, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, s, y_dev, nf, m->range_flag);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(m->pin_flag, m->range_flag, 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (*m->pin_flag) {
        fprintf(stderr, "libc3hip: activations beyond the range of the fp16x3 kernels; this handle continues on fp32 matrix instructions\n");
        m->f16_ok = false, m->precision = "fp32-range-guard";
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
    TRY(d2h_staged(&f, m->range_flag, 4));
    if (flag_out) *flag_out = (int)f;
    if (on_fp32_out) *on_fp32_out = m->f16_ok ? 0 : 1;
    return 0;
}

int c3_model_set_sharing(c3_model *m, int handles) {
    if (!m) return fail("null model");
    if (handles < 1) return fail("handles must be >= 1");
    m->sharing = handles;
    return 0;
}

int c3_model_describe(c3_model *m, char *buf, int n) {
    if (!m || !buf || n <= 0) return fail("null argument");
    if (m->kind == C3_KIND_PILEUP)
        snprintf(buf, (size_t)n, "sharing=%d duo=%d lstm1=%s proj2=%s lstm2=%s on_fp32=%d precision=%s lstm_wmax=%.4g lstm_hh_norm=%.4g auto_fp32_at=%.4g "
                 "ring_lanes=%d lane_max_batch=%lld tail_stream=%d", m->sharing,
                 m->duo, m->choice_lstm1, m->choice_proj2, m->choice_lstm2, (int)!m->f16_ok, m->precision, (double)m->lstm_wmax, (double)m->lstm_hh_norm,
                 (double)(m->precision_forced ? 0.f : m->auto_fp32_at), m->ring_lanes, (long long)m->lane_max_batch, (int)m->tail_split);
    else
        snprintf(buf, (size_t)n, "sharing=%d duo=%d conv_stack=%s stride1=%s conv3=%s conv5=%s on_fp32=%d ring_lanes=%d lane_max_batch=%lld tail_stream=%d", m->sharing,
                 m->duo, m->choice_fa, m->choice_s1, m->choice_s2[0], m->choice_s2[1], (int)!m->f16_ok, m->ring_lanes, (long long)m->lane_max_batch, (int)m->tail_split);
    return 0;
}

int c3_model_synchronize(c3_model *m) {
    if (!m) return fail("null model");
    HIP_TRY(hipStreamSynchronize(m->stream));
    if (m->tail_stream) HIP_TRY(hipStreamSynchronize(m->tail_stream));
    for (int k = 0; k < c3_model::kMaxLanes; ++k) {
        if (k == m->lane_cur) continue;
        if (m->parked[k].stream) HIP_TRY(hipStreamSynchronize(m->parked[k].stream));
        if (m->parked[k].tail_stream) HIP_TRY(hipStreamSynchronize(m->parked[k].tail_stream));
    }
    return 0;
}

int c3_model_destroy(c3_model *m) {
    if (!m) return 0;
    (void)hipSetDevice(m->device);
    (void)hipDeviceSynchronize();
    free_all_workspaces(m);
    for (int k = 0; k < c3_model::kMaxLanes; ++k) {
        if (k == m->lane_cur) continue;
        if (m->parked[k].stream) (void)hipStreamDestroy(m->parked[k].stream);
        if (m->parked[k].tail_stream) (void)hipStreamDestroy(m->parked[k].tail_stream);
        if (m->parked[k].ev_body_done) (void)hipEventDestroy(m->parked[k].ev_body_done);
        if (m->parked[k].ev_tail_done) (void)hipEventDestroy(m->parked[k].ev_tail_done);
    }
    float *ws[] = {m->proj_w[0], m->proj_w[1], m->proj_b[0], m->proj_b[1], m->whh[0], m->whh[1], m->whh16[0], m->whh16[1],
                   m->l4_w, m->l4_b, m->l4_wf, m->b5, m->zeros, m->l1_wih, m->l1_wih16, m->l1_bias, m->conv1_w16,
                   m->conv1_wfrag16, m->w5f, m->whf, m->bh48, m->proj2_pw, m->proj2_pwr, m->proj2_post, m->conv1_post,
                   m->conv1_w16_post, m->l4_pre, m->l4_post};
    for (float *p : ws)
        if (p) (void)hipFree(p);
    if (m->decode_dev) (void)hipFree(m->decode_dev);
    if (m->range_flag) (void)hipFree(m->range_flag);
    if (m->pin_flag) (void)hipHostFree(m->pin_flag);
    for (int l = 0; l < 9; ++l) {
        if (m->conv_w[l]) (void)hipFree(m->conv_w[l]);
        if (m->conv_b[l]) (void)hipFree(m->conv_b[l]);
        if (m->pconv_w[l]) (void)hipFree(m->pconv_w[l]);
        if (m->pconv_pre[l]) (void)hipFree(m->pconv_pre[l]);
        if (m->pconv_post[l]) (void)hipFree(m->pconv_post[l]);
        if (m->wconv_w[l]) (void)hipFree(m->wconv_w[l]);
        if (m->wconv_post[l]) (void)hipFree(m->wconv_post[l]);
    }
    for (auto &sl : m->slot) {
        if (sl.pin_x) (void)hipHostFree(sl.pin_x);
        if (sl.pin_y) (void)hipHostFree(sl.pin_y);
        if (sl.pin_flag) (void)hipHostFree(sl.pin_flag);
        if (sl.dev_x) (void)hipFree(sl.dev_x);
        if (sl.dev_y) (void)hipFree(sl.dev_y);
        if (sl.ev_h2d) (void)hipEventDestroy(sl.ev_h2d);
        if (sl.ev_out) (void)hipEventDestroy(sl.ev_out);
    }
    for (auto &r : m->recs) {
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    if (m->stream) (void)hipStreamDestroy(m->stream);
    if (m->h2d_stream) (void)hipStreamDestroy(m->h2d_stream);
    if (m->duo_stream) (void)hipStreamDestroy(m->duo_stream);
    if (m->tail_stream) (void)hipStreamDestroy(m->tail_stream);
    if (m->ev_body_done) (void)hipEventDestroy(m->ev_body_done);
    if (m->ev_tail_done) (void)hipEventDestroy(m->ev_tail_done);
    if (m->duo_fork) (void)hipEventDestroy(m->duo_fork);
    if (m->duo_join) (void)hipEventDestroy(m->duo_join);
    delete m;
    return 0;
}

}  // extern "C"

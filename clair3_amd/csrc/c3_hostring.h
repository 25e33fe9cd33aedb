// c3_hostring.h -- the host <-> device side of the model call: c3_predict (the reference's blocking _torch_predict,
// clair3/CallVariantsFromCffi.py:48-52), its asynchronous pair c3_predict_submit / c3_predict_wait (a ring of C3_HOST_SLOTS
// batches in flight: staging copy, H2D, kernels, D2H, range guard), the region form of the pileup call, and the decoder entry
// points that take host rows.  The library never page-locks CALLER memory (see stage_h2d).
#pragma once
#include "c3_forward.h"

// ------------------------------------------------------------------------------------------ host staging
// The caller's windows are pageable numpy memory (clair3/CallVariantsFromCffi.py:112-133: np.load slices); they go
// through a pinned buffer, cut into pieces: the H2D transfer of a piece is queued as soon as it is staged, so the DMA of
// piece i runs under the memcpy of piece i + 1, and every piece's memcpy is split over the staging pool (c3_host.h).
// There is NO zero-copy source path: until round 5 the ABI could page-lock caller memory (hipHostRegister: c3_host_register,
// c3_model_set_lock_sources).  On ROCm 7.2 a process that registers / unregisters host ranges and also lets another HIP user (PyTorch)
// copy from pageable memory takes "Memory access fault by GPU" sooner or later (tests/diag/register_vs_torch_probe.py reproduces it
// with hipHostRegister and torch alone), and a C ABI cannot know who else lives in its process: the entry points are gone, every
// caller buffer -- numpy, a memory-mapped tensor file, libclair3's fa_data.matrix -- is staged through the library's own pinned,
// MADV_DONTFORK memory.  Price: 0.86 instead of 0.88 of the device-resident rate on a blocking call of 1000 full-alignment windows.

// Rows always leave through a copy kernel on the COMPUTE stream (host_copy_kernel writes the pinned, device-mapped result
// buffer): handing them to a transfer stream -- event, cross-queue wait, DMA copies, event -- cost the compute queue ~75 us per
// batch (profiles/r03_e_d2h_by_kernel.txt).  Windows come in on the transfer stream (DMA engine; the compute stream's wait for
// that event is free, the transfer finished batches ago) -- except a small batch with nothing else in flight (the blocking call
// of one chunk): there a copy kernel on the compute stream reading the pinned staging buffer is the shorter way, up to
// kKernelCopyMax bytes (4000 pileup windows = 2.4 MB, ~50 us at PCIe Gen5 rates).
constexpr size_t kKernelCopyMax = (size_t)4 << 20;
__global__ __launch_bounds__(256) void host_copy_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16,
                                                       const uint32_t *flag_src, uint32_t *flag_dst) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
    if (flag_dst && blockIdx.x == 0 && threadIdx.x == 0) *flag_dst = *flag_src;
}

// one thread per output float: raises the handle's range flag when a probability is not finite
__global__ void rows_finite_kernel(const float *y, int64_t n, uint32_t *flag) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n && (__float_as_uint(y[i]) & 0x7f800000u) == 0x7f800000u) atomicOr(flag, 2u);
}

// workgroups of the copy kernel that writes the rows into the pinned result buffer: 32 for a batch (96 KB), up to 256 for a group
// of 2000 full-alignment rows with decoder columns (968 KB): more stores in flight across PCIe
static unsigned rows_out_grid(size_t bytes) { return (unsigned)std::min<size_t>(256, std::max<size_t>(32, bytes / 4096)); }

// stage [src, src + bytes) through `pin` into `dev` on stream s, piecewise
static int stage_h2d(void *dev, void *pin, const void *src, size_t bytes, hipStream_t s) {
    // >= 4 MiB and at most four pieces: every queued transfer costs ~15 us of host time (2 MiB x 8 was slower again)
    const size_t piece = std::max<size_t>((size_t)4 << 20, ((bytes / 4) + 4095) & ~(size_t)4095);
    for (size_t off = 0; off < bytes; off += piece) {
        const size_t n = std::min(piece, bytes - off);
        StagePool::get().copy((char *)pin + off, (const char *)src + off, n);
        HIP_TRY(hipMemcpyAsync((char *)dev + off, (char *)pin + off, n, hipMemcpyHostToDevice, s));
    }
    return 0;
}

// The forward pass of a batch of the ring on the handle's stream, its FC chain on tail_stream (c3_forward.h tail_begin): *outs is the stream the
// rows are complete on -- where the copy-out kernel and the slot's event go.  The next batch's layers are queued on m->stream right behind this
// batch's LAST LAYER, not behind its chain.
static int ring_forward(c3_model *m, const void *x_dev, int x_dtype, int64_t batch, float *y_dev, hipStream_t *outs) {
    m->tail_now = m->tail_split && !m->keep && m->duo == 0 && !m->prof;
    const int rc = forward_device(m, m->stream, x_dev, x_dtype, batch, y_dev);
    *outs = (m->tail_now && m->tail_stream && batch > 0) ? m->tail_stream : m->stream;
    m->tail_now = false;
    return rc;
}

extern "C" {

// Pinned staging memory stays out of forked children (keep_out_of_children, c3_model.h: the staging copies of the 40 groups after
// eight forks 155 -> 97 ms, profiles/r05_k_host_loop_feeder_not_kept.txt).
static int ensure_slot(c3_model *m, HostSlot &sl, size_t xb, size_t yb) {
    // host_copy_kernel moves whole 16-byte pieces ((bytes + 15) / 16 of them): every buffer it touches is sized to a multiple of
    // 256 bytes here, for every path (90-column rows of an odd batch, 121-float decoder rows: yb % 16 != 0) -- and to whole pages, so that
    // the pinned halves can be kept out of forked children page by page (keep_out_of_children)
    xb = (xb + 4095) & ~(size_t)4095, yb = (yb + 4095) & ~(size_t)4095;
    if (!sl.ev_h2d) {
        HIP_TRY(hipEventCreateWithFlags(&sl.ev_h2d, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&sl.ev_out, hipEventDisableTiming));
    }
    if (!sl.pin_flag) {
        HIP_TRY(hipHostMalloc((void **)&sl.pin_flag, 4096, hipHostMallocDefault));  // (a whole page: MADV_DONTFORK works on pages)
        keep_out_of_children(sl.pin_flag, 4096);
    }
    if (xb > sl.cap_x) {
        if (sl.pin_x) (void)hipHostFree(sl.pin_x);
        if (sl.dev_x) (void)hipFree(sl.dev_x);
        sl.pin_x = sl.dev_x = nullptr, sl.cap_x = 0;
        HIP_TRY(hipHostMalloc(&sl.pin_x, xb, hipHostMallocDefault));
        keep_out_of_children(sl.pin_x, xb);
        HIP_TRY(hipMalloc(&sl.dev_x, xb));
        sl.cap_x = xb;
    }
    if (yb > sl.cap_y) {
        if (sl.pin_y) (void)hipHostFree(sl.pin_y);
        if (sl.dev_y) (void)hipFree(sl.dev_y);
        sl.pin_y = nullptr, sl.dev_y = nullptr, sl.cap_y = 0;
        HIP_TRY(hipHostMalloc((void **)&sl.pin_y, yb, hipHostMallocDefault));
        keep_out_of_children(sl.pin_y, yb);
        HIP_TRY(hipMalloc((void **)&sl.dev_y, yb));
        sl.cap_y = yb;
    }
    (void)m;
    return 0;
}

static int predict_submit(c3_model *m, const void *x_host, int x_dtype, int64_t batch, float *y_host, int slot, float *y_dev_out = nullptr);
int c3_predict_submit(c3_model *m, const void *x_host, int x_dtype, int64_t batch, float *y_host, int slot) {
    return predict_submit(m, x_host, x_dtype, batch, y_host, slot);
}
// the ring with the rows LEFT ON THE DEVICE (a rank of a sharded job: its rows go to the RCCL gather, not to this host): the
// forward pass writes them straight into the caller's device buffer, only the range flag crosses PCIe
int c3_predict_submit_dev(c3_model *m, const void *x_host, int x_dtype, int64_t batch, float *y_dev, int slot) {
    if (batch > 0 && !y_dev) return fail("null device buffer");
    return predict_submit(m, x_host, x_dtype, batch, nullptr, slot, y_dev);
}
static int predict_submit(c3_model *m, const void *x_host, int x_dtype, int64_t batch, float *y_host, int slot, float *y_dev_out) {
    if (!m) return fail("null model");
    if (slot < 0 || slot >= kHostSlots) return fail("slot must be in [0, %d)", kHostSlots);
    if (batch < 0) return fail("negative batch");
    if (batch > 0 && (!x_host || (!y_host && !y_dev_out))) return fail("null buffer");
    HostSlot &sl = m->slot[slot];
    if (sl.busy) return fail("slot %d still in flight: call c3_predict_wait first", slot);
    HIP_TRY(hipSetDevice(m->device));
    if (!m->loaded) return fail("model has no weights: call c3_model_load first");
    // the batch in slot k runs in lane k % lanes (c3_model.h Lane): its own workspace and kernel stream, so that it overlaps the batch of the
    // neighbouring slot on the chip; keep mode, the two-halves knob and profiling stay in one lane
    // -- and so does a batch that fills the chip by itself: two of those side by side only get in each other's way (same-box A/B,
    // profiles/r06_i_ab_ring_lanes.txt: full alignment ring +5.5 % at B = 256, -4 % at B = 1000)
    const bool in_lane = m->ring_lanes > 1 && batch <= m->lane_max_batch && !m->keep && m->duo == 0 && !m->prof;
    // lanes are dealt in the ORDER of the submits, not by slot number: three slots on two lanes (the pileup network) would put two of every
    // three batches behind each other in lane 0 (C3HIP_LANE_ORDER=slot: the slot's number, as before)
    TRY(use_lane(m, in_lane ? (m->lane_by_slot ? slot : (int)(m->lane_next++ % (unsigned)m->ring_lanes)) % m->ring_lanes : 0));
    const size_t xb = (size_t)(batch * c3_model_window_bytes(m, x_dtype));
    const size_t yb = (size_t)batch * m->row * sizeof(float);
    // C3HIP_HOST_COPY_KERNEL: 0 = never, 1 = up to kKernelCopyMax, n > 1 = up to n KB (A/B of the threshold)
    const size_t kcopy_max = m->host_copy_kernel > 1 ? (size_t)m->host_copy_kernel << 10 : kKernelCopyMax;
    // ... and only while no other batch of this handle is in flight: behind a running batch the transfer stream brings the windows
    // in under its kernels (pileup ring 4.22 M -> 4.37 M windows/s), alone the copy kernel is the shorter way (blocking call of one
    // chunk 3.87 M against 3.64 M)
    bool alone = true;
    for (int k = 0; k < kHostSlots; ++k) alone &= !m->slot[k].busy;
    // beside batches in the other lanes: the kernel forms for a shared chip (c3_model.h lane_sharing; rows bit-identical either way) -- when this
    // batch and the largest ones in flight in the other lanes would, on half tiles (two workgroups per 8 windows), ask for more workgroups than
    // the chip has CUs (ring of 1024-window batches: yes; the blocking call's 250 + 750 pieces: no, 252 half-tile workgroups fit side by side)
    int64_t beside_windows = 0;
    if (in_lane && !alone && m->lane_sharing_ok) {
        int64_t other[kHostSlots];
        int no = 0;
        for (int k = 0; k < kHostSlots; ++k)
            if (m->slot[k].busy && m->slot[k].batch <= m->lane_max_batch) other[no++] = m->slot[k].batch;
        std::sort(other, other + no, std::greater<int64_t>());
        for (int k = 0; k < no && k < m->ring_lanes - 1; ++k) beside_windows += other[k];
    }
    struct LaneSharing {
        c3_model *m;
        LaneSharing(c3_model *m_, int v) : m(m_) { m->lane_sharing = v; }
        ~LaneSharing() { m->lane_sharing = 1; }
    } lane_sharing(m, (beside_windows > 0 && 2 * ((batch + beside_windows + 7) / 8) > m->wg_slots / 2) ? m->ring_lanes : 1);
    if (batch > 0 && m->host_copy_kernel && alone && xb <= kcopy_max && yb <= kcopy_max) {
        TRY(ensure_slot(m, sl, (xb + 255) & ~(size_t)255, y_dev_out ? 0 : (yb + 255) & ~(size_t)255));  // (rows that stay on the device need no slot buffers)
        StagePool::get().copy(sl.pin_x, x_host, xb);  // (plain memcpy below 1 MB, split over the helpers above)
        hipLaunchKernelGGL(host_copy_kernel, dim3(128), dim3(256), 0, m->stream, (const uint4 *)sl.pin_x, (uint4 *)sl.dev_x, (xb + 15) / 16,
                           (const uint32_t *)nullptr, (uint32_t *)nullptr);
        HIP_TRY(hipGetLastError());
        const bool f16 = m->f16_ok;
        hipStream_t outs;
        TRY(ring_forward(m, sl.dev_x, x_dtype, batch, y_dev_out ? y_dev_out : sl.dev_y, &outs));
        if (y_dev_out && f16)
            hipLaunchKernelGGL(rows_finite_kernel, dim3((unsigned)((batch * m->row + 255) / 256)), dim3(256), 0, outs, y_dev_out, batch * m->row, m->range_flag);
        hipLaunchKernelGGL(host_copy_kernel, dim3(y_dev_out ? 1 : rows_out_grid(yb)), dim3(256), 0, outs, (const uint4 *)sl.dev_y, (uint4 *)sl.pin_y,
                           y_dev_out ? 0 : (yb + 15) / 16, (const uint32_t *)m->range_flag, sl.pin_flag);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(sl.ev_out, outs));
        sl.used_f16 = f16;
    } else
    if (batch > 0) {
        // the slot becomes busy only once everything has been queued: a failure on the way leaves it free
        TRY(ensure_slot(m, sl, xb, y_dev_out ? 0 : yb));  // (rows that stay on the device need no slot buffers)
        if (in_lane && m->lane_h2d) {
            // a small batch in a lane brings its windows in on the lane's OWN stream: copy, kernels and the copy-out in order in one queue, no
            // event between two streams -- the batches of the other lanes are what the copy runs under (c3_model.h lane_h2d)
            TRY(stage_h2d(sl.dev_x, sl.pin_x, x_host, xb, m->stream));
        } else {
            if (!m->h2d_stream) HIP_TRY(new_stream(m, &m->h2d_stream));
            TRY(stage_h2d(sl.dev_x, sl.pin_x, x_host, xb, m->h2d_stream));
            HIP_TRY(hipEventRecord(sl.ev_h2d, m->h2d_stream));
            HIP_TRY(hipStreamWaitEvent(m->stream, sl.ev_h2d, 0));
        }
        const bool f16 = m->f16_ok;
        hipStream_t outs;
        TRY(ring_forward(m, sl.dev_x, x_dtype, batch, y_dev_out ? y_dev_out : sl.dev_y, &outs));
        if (y_dev_out && f16)  // rows that stay on the device are scanned there (bit 1 of the flag: a non-finite row)
            hipLaunchKernelGGL(rows_finite_kernel, dim3((unsigned)((batch * m->row + 255) / 256)), dim3(256), 0, outs, y_dev_out, batch * m->row, m->range_flag);
        // the rows (96 - 484 B per window) and the range flag leave through a copy kernel on the COMPUTE stream, whatever the
        // batch: handing them to a transfer stream (event, cross-queue wait, two DMA copies, event) cost the compute queue
        // ~75 us per batch -- 538 k -> 647 k windows/s host to host at B = 256 (profiles/r03_e_d2h_by_kernel.txt)
        hipLaunchKernelGGL(host_copy_kernel, dim3(y_dev_out ? 1 : rows_out_grid(yb)), dim3(256), 0, outs, (const uint4 *)sl.dev_y, (uint4 *)sl.pin_y,
                           y_dev_out ? 0 : (yb + 15) / 16, (const uint32_t *)m->range_flag, sl.pin_flag);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(sl.ev_out, outs));
        sl.used_f16 = f16;
    }
    sl.y_dev_out = y_dev_out;
    sl.lane = m->lane_cur;
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
    if (sl.y_dev_out) {  // rows stayed on the device: the flag (range bit + the device-side scan for non-finite rows) is all there is to read
        if (sl.used_f16 && *sl.pin_flag != 0) {
            if (m->f16_ok)
                fprintf(stderr, "libc3hip: activations beyond the range of the fp16x3 kernels; this handle continues on fp32 matrix instructions\n");
            m->f16_ok = false, m->precision = "fp32-range-guard";
            TRY(use_lane(m, sl.lane));
            TRY(forward_device(m, m->stream, sl.dev_x, sl.x_dtype, sl.batch, sl.y_dev_out));
            HIP_TRY(hipStreamSynchronize(m->stream));
        }
        return 0;
    }
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
            m->f16_ok = false, m->precision = "fp32-range-guard";
            TRY(use_lane(m, sl.lane));
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
    // A blocking call cannot hide its FIRST staging copy behind a previous batch (pageable -> pinned, ~16 GB/s with the staging pool);
    // every later piece's copy and transfer run under the kernels of the piece before it.
    int64_t n_sub = 0, n_done = 0;  // chunks submitted / waited for
    int rc = 0;
    // piece sizes grow threefold from a quarter of the batch (between chunk / 2 and chunk): the kernels start after a SHORT first
    // transfer, every later transfer hides under the kernels of the piece before it (a full-alignment window takes 0.42 us to
    // cross PCIe and 1.3 us to compute), and few big pieces fill the chip better than many small ones -- 1000 windows as 250 +
    // 750 instead of 128 + 256 + 616: 0.79 -> 0.82 of device-resident; a tail shorter than half the next size joins the last piece
    int64_t next = std::min<int64_t>(std::max<int64_t>(chunk / 2, batch / 4), chunk);
    next = std::max<int64_t>(next, 1);
    // A batch that fits the ring's lanes as EQUAL pieces -- one lane-sized piece per lane: 334 + 333 + 333 full-alignment windows for the
    // reference's batch of 1000 -- is cut that way: every piece brings its windows in on its own lane's stream and runs beside the others
    // (profiles/r06_o_*: 656 - 667 k -> 700 - 721 k windows/s same-box; with the lanes' first form -- up to three streams each -- the same cut
    // LOST 4 %, profiles/r06_i_ab_ring_lanes.txt).  C3HIP_PREDICT_EQUAL=0: the growing pieces below for every batch.
    static const bool equal_ok = !(getenv("C3HIP_PREDICT_EQUAL") && atoi(getenv("C3HIP_PREDICT_EQUAL")) == 0);
    const bool lanes_on = m->ring_lanes > 1 && !m->keep && m->duo == 0 && !m->prof;
    const int64_t equal = (equal_ok && lanes_on && m->kind == C3_KIND_FULL_ALIGNMENT && batch <= (int64_t)m->ring_lanes * m->lane_max_batch)
                              ? (batch + m->ring_lanes - 1) / m->ring_lanes : 0;
    // C3HIP_PREDICT_PIECES=a,b,c: A/B knob -- the piece sizes themselves (the last one repeats)
    static const std::vector<int64_t> forced = [] {
        std::vector<int64_t> v;
        if (const char *e = getenv("C3HIP_PREDICT_PIECES"))
            for (const char *q = e; *q;) {
                char *end = nullptr;
                const long long n = strtoll(q, &end, 10);
                if (end == q) break;
                if (n > 0) v.push_back(n);
                q = *end ? end + 1 : end;
            }
        return v;
    }();
    for (int64_t off = 0; off < batch && rc == 0; ++n_sub) {
        int64_t take = std::min(next, batch - off);
        if (batch - off - take < next / 2 || batch - off - take < chunk / 2) take = batch - off;
        if (equal > 0) take = std::min(equal, batch - off);
        if (!forced.empty()) take = std::min<int64_t>(forced[std::min<size_t>((size_t)n_sub, forced.size() - 1)], batch - off);
        take = std::min(take, max_microbatch(m));
        if (n_sub - n_done == kRing) rc = c3_predict_wait(m, (int)(n_done++ % kRing));
        if (rc == 0)
            rc = predict_submit(m, (const char *)x_host + off * wbytes, x_dtype, take, y_host + off * m->row, (int)(n_sub % kRing));
        if (rc != 0) break;
        off += take;
        next = std::min(3 * next, 4 * chunk);
    }
    const std::string first_error = rc != 0 ? g_err : std::string();
    for (; n_done < n_sub; ++n_done) {  // drain, also after an error: no slot stays busy behind a failed call
        const int r = c3_predict_wait(m, (int)(n_done % kRing));
        if (rc == 0) rc = r;
    }
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
    TRY(use_lane(m, 0));
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
    TRY(h2d_staged(y, y_host, yb, m->stream));  // (pageable rows: through the bounce buffer, c3_model.h)
    TRY(h2d_staged(ref, ref21_host, (size_t)batch, m->stream));
    DecodeParams dp{y, m->nout, ref, maxp, arg, early, nullptr, (int)batch, m->nout == 90 ? 1 : 0};
    hipLaunchKernelGGL(outcome_maxima_kernel<false>, dim3((unsigned)((batch + 3) / 4)), dim3(256), 0, m->stream, dp);
    HIP_TRY(hipGetLastError());
    TRY(d2h_staged(maxp_host, maxp, mb, m->stream));
    TRY(d2h_staged(argmax_host, arg, mb, m->stream));
    TRY(d2h_staged(early_host, early, (size_t)batch, m->stream));
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
    {   // the rows widen on their way through the bounce buffer (the kernel fills the decoder columns behind each)
        BounceBuf &b = bounce_buf();
        std::lock_guard<std::mutex> lk(b.mu);
        TRY(bounce_ready(b));
        const int64_t per = (int64_t)(BounceBuf::kBytes / ((size_t)wide * sizeof(float)));
        for (int64_t r0 = 0; r0 < batch; r0 += per) {
            const int64_t nr = std::min(per, batch - r0);
            for (int64_t r = 0; r < nr; ++r)
                memcpy((float *)b.pin + r * wide, y_host + (r0 + r) * m->nout, (size_t)m->nout * sizeof(float));
            HIP_TRY(hipMemcpyAsync(rows + r0 * wide, b.pin, (size_t)nr * wide * sizeof(float), hipMemcpyHostToDevice, m->stream));
            HIP_TRY(hipStreamSynchronize(m->stream));
        }
    }
    DecodeParams dp{rows, wide, nullptr, nullptr, nullptr, nullptr, rows + m->nout, (int)batch, m->nout == 90 ? 1 : 0};
    hipLaunchKernelGGL(outcome_maxima_kernel<true>, dim3((unsigned)((batch + 3) / 4)), dim3(256), 0, m->stream, dp);
    HIP_TRY(hipGetLastError());
    TRY(d2h_staged(rows_host, rows, total, m->stream));
    return 0;
}

}  // extern "C"

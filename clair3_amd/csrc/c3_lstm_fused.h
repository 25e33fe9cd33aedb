// c3_lstm_fused.h -- first pileup LSTM layer with its input projection fused into the recurrence.
//
// LSTM1 reads only C = 18 counts per (window, position) (clair3/model.py:96-101, shared/param_p.py:32-36).  Hoisting
// its x-projection into a GEMM, as the generic path does, writes and re-reads a (B*33) x 1024 fp32 tensor -- 135 KB
// per window for 594 bytes of input, 276 MB of HBM traffic per 1024 windows and a 70 us write-bound launch.  Here the
// projection rides in the recurrent kernel instead: K = 18 pads to 5 MFMA k-steps of 4, i.e. 20 extra
// v_mfma_f32_16x16x4_f32 per step next to the 128 of the recurrence, W_ih fragments (20 VGPRs) and the bias stay in
// registers, and the 5 counts each lane needs for the next step are fetched (as bytes) during the current one.
// With `starts` the kernel gathers its windows straight out of a region-wide pileup matrix: the
// reference slices 33 overlapping columns per candidate on the host (preprocess/CreateTensorPileupFromCffi.py:
// 362-371), a 16x duplication that never has to exist on the device.
// Everything else is lstm_recurrent_kernel<128, true>: wave w owns hidden units [16w, 16w+16) as four gate
// accumulators, W_hh resident in 128 VGPRs, h_t exchanged through a double-buffered LDS tile, one LDS-only barrier
// per step, cell state in registers.
#pragma once
#include "c3_kernels.h"

namespace c3 {

template <typename TX>
struct LstmFusedParams {
    const TX *x;        // [B][T][C] window counts, or -- with `starts` -- one [n_cols][C] region matrix
    const int32_t *starts;  // optional [B]: first region column of each window (window b = columns starts[b] .. +T-1)
    const float *wih;   // [dir][wave][gate][ks = 5][lane]: W_ih[gate*H + wave*16 + (lane&15)][4*ks + (lane>>4)], 0 beyond C
    const float *bias;  // [dir][wave][gate][16]: b_ih + b_hh of row gate*H + wave*16 + u
    const float *whh;   // [dir][wave][gate][q = H/16][lane][4]   (same packing as lstm_recurrent_kernel)
    const uint32_t *wih16;  // F16 + int8 input: [dir][wave][gate][piece][lane][4 dwords = 8 fp16]: piece of
                            // 128 W_ih[gate*H + wave*16 + (lane&15)][8 (lane>>4) + j] (0 beyond C); the kernel feeds x / 128
    float *hout;        // [B][T][2H]; column = dir*H + unit
    int B, T, C;
    void *hplanes = nullptr;  // F16: write h as plane activations instead ([B*T][2H/64 slabs][hi 64 | lo 64] fp16, c3_conv3.h):
                              // the two pieces the recurrence forms anyway, and what dense_planes_kernel (c3_dense.h) reads
    unsigned long long *trace = nullptr;  // OPT bit 3 (C3HIP_LSTM_TRACE): shader-clock stamps of workgroup (0, 0), [wave][step][4]
};

constexpr int kFusedKS = 5;  // k-steps of 4 covering C <= 20 input channels

// F16: the recurrent product h_{t-1} W_hh^T on v_mfma_f32_16x16x32_f16, both operands as two fp16 pieces (fp16x3, see
// lstm_recurrent_kernel_v2): fragment slot q of a gate holds piece q & 1 of k-step q >> 1, h lives in two fp16 planes.
// The 18-channel input projection stays on its 20 fp32 matrix instructions.
// OPT bit 0: the counts of step t + 1 stay as loaded (raw 16-bit pairs) until the top of step t + 1 and are widened there.
// Widening them where they are requested -- what a single load_x16() does -- puts a wait for four just-issued loads between
// the projection and the recurrent matrix instructions of EVERY step (hipcc keeps the conversion next to the loads): one
// exposed L2 round trip per step on the critical path of a latency-bound kernel.
// OPT bit 1: h leaves as planes, known at compile time (see `planes` below).
// OPT bit 2: HALF tiles -- a workgroup takes 8 windows instead of 16, on rows {0,1, 4,5, 8,9, 12,13} of the 16-row matrix tile
//            (a lane's accumulator elements 0 and 1; rows 2,3 mod 4 stay zero and are never evaluated).  The matrix phase of a
//            step costs the same, but its cell phase -- vector-unit bound and as long as half the matrix phase -- halves, and a batch
//            of 1024 windows becomes 256 workgroups instead of 128: the recurrences are latency-bound chains and half of the
//            chip was idle.  The host picks it when the full tiles would leave CUs empty.
// OPT bit 3: phase trace (debug).
template <typename TX, bool F16 = false, int OPT = 0>
__global__ __launch_bounds__(512) void lstm1_fused_kernel(LstmFusedParams<TX> p) {
    constexpr int H = 128, NW = 8, NQ = 8, LDH = H + 4;
    constexpr int LDH16 = 2 * H + 16;  // bytes per row of an fp16 h plane: 16 rows fall on 16 different 16-byte bank slots
    __shared__ __attribute__((aligned(16))) float hbuf[2][16][LDH];
    static_assert(2 * 2 * 16 * LDH16 <= (int)sizeof(float) * 2 * 16 * (H + 4) + 2048 || !F16, "");
    __shared__ __attribute__((aligned(16))) char hbuf16[F16 ? 2 * 2 * 16 * LDH16 : 16];
    auto hb16 = [&](int buf, int piece, int row, int k) -> char * { return hbuf16 + ((buf * 2 + piece) * 16 + row) * LDH16 + 2 * k; };

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s = lane >> 4, col = lane & 15;
    const int dir = blockIdx.y;
    constexpr bool HALF = F16 && (OPT & 4);
    const int b0 = blockIdx.x * (HALF ? 8 : 16);
    // window of matrix row r: r itself, or (HALF) 2 (r >> 2) + (r & 1) for the rows with bit 1 clear
    auto row_window = [&](int r) { return HALF ? ((r & 2) ? p.B : b0 + 2 * (r >> 2) + (r & 1)) : b0 + r; };

    for (int i = tid; i < 16 * LDH; i += NW * 64) (&hbuf[0][0][0])[i] = 0.f;
    if constexpr (F16)
        for (int i = tid; i < 2 * 2 * 16 * LDH16 / 4; i += NW * 64) reinterpret_cast<uint32_t *>(hbuf16)[i] = 0u;

    // resident weights: W_hh fragments (128 VGPRs), W_ih fragments (20), bias (4)
    const float *wbase = p.whh + ((int64_t)(dir * NW + wave) * 4 * NQ * 64 + lane) * 4;
    f32x4v wres[4 * NQ];
#pragma unroll
    for (int i = 0; i < 4 * NQ; ++i) wres[i] = *reinterpret_cast<const f32x4v *>(wbase + (int64_t)i * 256);
    float wih[4][kFusedKS], bias[4];
    {
        const float *wi = p.wih + ((int64_t)(dir * NW + wave) * 4 * kFusedKS) * 64 + lane;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int ks = 0; ks < kFusedKS; ++ks) wih[g][ks] = wi[(g * kFusedKS + ks) * 64];
            bias[g] = p.bias[((dir * NW + wave) * 4 + g) * 16 + col];
        }
    }

    // A operand of the projection: lane (window = lane&15, s) supplies x[window][t][4*ks + s].  Buffer loads: one
    // loop-invariant vector offset per k-step (out of range beyond C channels: the load returns the 0 the padding needs),
    // the time step is the scalar offset -- no per-step address arithmetic.
    int xb = row_window(col);
    const bool xvalid = xb < p.B;  // (rows beyond the batch used to re-read the last window; their results were never stored)
    if (xb >= p.B) xb = p.B - 1;
    const int64_t xrow = p.starts ? (int64_t)p.starts[xb] * p.C : (int64_t)xb * p.T * p.C;
    // region mode: the host has checked starts[b] + T <= n_cols, the descriptor only has to end below the redirect offset
    const uint32_t xbytes = p.starts ? 0x7fffffffu : (uint32_t)((int64_t)p.B * p.T * p.C * sizeof(TX));
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<TX *>(p.x), 0, xbytes, 0x00020000);
    uint32_t xo[kFusedKS];
#pragma unroll
    for (int ks = 0; ks < kFusedKS; ++ks) {
        const int k = 4 * ks + s;
        xo[ks] = k < p.C ? (uint32_t)((xrow + k) * sizeof(TX)) : 0x80000000u;
    }
    auto load_x = [&](int t, float (&xa)[kFusedKS]) __attribute__((always_inline)) {
        const int so = t * p.C * (int)sizeof(TX);
#pragma unroll
        for (int ks = 0; ks < kFusedKS; ++ks) {
            if constexpr (sizeof(TX) == 1) xa[ks] = (float)(int8_t)__builtin_amdgcn_raw_buffer_load_b8(xrsrc, xo[ks], so, 0);
            else xa[ks] = (float)(int32_t)__builtin_amdgcn_raw_buffer_load_b32(xrsrc, xo[ks], so, 0);
        }
    };

    f32x4v biasv[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) biasv[g] = f32x4v{bias[g], bias[g], bias[g], bias[g]};
    // F16 + int8 windows: the projection too runs on v_mfma_f32_16x16x32_f16.  The counts are exact in fp16 (one piece);
    // lane (window, s) supplies channels 8 s .. 8 s + 7 of x[window][t] as b / 128 -- four 2-byte loads, each widened by
    // one v_perm_b32 + one v_pk_add_f16 (byte ^ 0x80 into the mantissa of 8.0, minus 9) -- against two fp16 pieces of
    // 128 W_ih: 8 matrix instructions of 16 cycles instead of 20 of 32.
    constexpr bool F16P = F16 && sizeof(TX) == 1;
    typedef uint32_t u32x4p __attribute__((ext_vector_type(4)));
    u32x4p wp16[4][2];
    uint32_t xo2[4];
    if constexpr (F16P) {
        const u32x4p *wi = reinterpret_cast<const u32x4p *>(p.wih16) + ((int64_t)(dir * NW + wave) * 4 * 2) * 64 + lane;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int q = 0; q < 2; ++q) wp16[g][q] = wi[(g * 2 + q) * 64];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = 8 * s + 2 * j;  // channels k, k + 1 (C is even: 18)
            xo2[j] = k < p.C && (xvalid || !HALF) ? (uint32_t)(xrow + k) : 0x80000000u;
        }
    }
    auto load_x16 = [&](int t, u32x4p &xa) __attribute__((always_inline)) {
        const int so = t * p.C;
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t u = (uint32_t)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(xrsrc, xo2[j], so, 0) ^ 0x8080u;
            const uint32_t w = __builtin_amdgcn_perm(0x48484848u, u, 0x04010400u);
            xa[j] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(h2, w) + h2{(_Float16)-9.0f, (_Float16)-9.0f});
        }
    };
    u32x4p xn16 = {0u, 0u, 0u, 0u};
    constexpr bool DEFER = F16P && (OPT & 1);
    uint32_t xraw[4] = {0u, 0u, 0u, 0u};
    auto load_raw = [&](int t) __attribute__((always_inline)) {
        const int so = t * p.C;
#pragma unroll
        for (int j = 0; j < 4; ++j) xraw[j] = (uint32_t)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(xrsrc, xo2[j], so, 0);
    };
    auto widen_raw = [&](u32x4p &xa) __attribute__((always_inline)) {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t w = __builtin_amdgcn_perm(0x48484848u, xraw[j] ^ 0x8080u, 0x04010400u);
            xa[j] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(h2, w) + h2{(_Float16)-9.0f, (_Float16)-9.0f});
        }
    };
    constexpr bool TRACE = (OPT & 8) != 0;
    auto stamp = [&](int step, int k) __attribute__((always_inline)) {
        if constexpr (TRACE) {
            __builtin_amdgcn_sched_barrier(0);
            if (p.trace && blockIdx.x == 0 && blockIdx.y == 0 && lane == 0) p.trace[(wave * 64 + step) * 4 + k] = __builtin_readcyclecounter();
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    const int h_col = dir * H + wave * 16 + col;
    float c[4] = {0.f, 0.f, 0.f, 0.f};
    const __amdgpu_buffer_rsrc_t hrsrc =
        __builtin_amdgcn_make_buffer_rsrc(p.hout, 0, (uint32_t)((int64_t)p.B * p.T * 2 * H * 4), 0x00020000);
    uint32_t ho[4];  // byte offset of hout[window 4s+v][0][h_col]; windows beyond B are out of range (store dropped)
    // OPT bit 1: the caller guarantees hplanes != nullptr.  With `planes` a run-time value the plane store sits in a branch, and
    // hipcc then waits for the step's four count loads with vmcnt(3..0) -- which, loads and stores sharing one in-order
    // counter, also waits for the 16-byte store issued just before them: a full HBM write round trip on every step.
    const bool planes = (F16 && (OPT & 2)) ? true : (F16 && p.hplanes != nullptr);  // same bytes per (window, step): 2H x 4
    const __amdgpu_buffer_rsrc_t prsrc = __builtin_amdgcn_make_buffer_rsrc(planes ? p.hplanes : (void *)p.hout, 0,
                                                                            (uint32_t)((int64_t)p.B * p.T * 2 * H * 4), 0x00020000);
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int b = row_window(4 * s + v);
        ho[v] = b < p.B ? (uint32_t)((((int64_t)b * p.T) * (2 * H) + h_col) * 4) : 0x80000000u;
    }
    // planes: the step's h tile (16 windows x H units, two pieces) already sits in LDS in plane order; one step later -- behind
    // the barrier that publishes it -- every thread copies ONE 16-byte piece of it to global memory: thread (piece, window,
    // 8-unit group) -> 8 lanes per 128-byte plane row, instead of eight scattered 2-byte stores per thread and step
    static_assert(!F16 || H == 128, "the plane copy assumes 512 threads = 2 pieces x 16 windows x 16 groups of 8 units");
    const int cp_p = tid >> 8, cp_r = (tid >> 4) & 15, cp_c = tid & 15;
    const uint32_t cp_off = planes && row_window(cp_r) < p.B
                                ? (uint32_t)(((int64_t)row_window(cp_r) * p.T) * (2 * H) * 4 + (dir * 2 + (cp_c >> 3)) * 256 + cp_p * 128 + (cp_c & 7) * 16)
                                : 0x80000000u;
    auto copy_planes = [&](int buf, int t) __attribute__((always_inline)) {
        typedef uint32_t u32x4c __attribute__((ext_vector_type(4)));
        const u32x4c d = *reinterpret_cast<const u32x4c *>(hb16(buf, cp_p, cp_r, 8 * cp_c));
        __builtin_amdgcn_raw_buffer_store_b128(d, prsrc, cp_off + (uint32_t)(t * 2 * H * 4), 0, 0);
    };
    float xn[kFusedKS];
    // OPT bit 0 for the fp32 projection (int32 windows): the same deferral -- raw integers until the top of their step
    constexpr bool DEFER32 = !F16P && (OPT & 1);
    int xr32[kFusedKS];
    auto load_raw32 = [&](int t) __attribute__((always_inline)) {
        const int so = t * p.C * (int)sizeof(TX);
#pragma unroll
        for (int ks = 0; ks < kFusedKS; ++ks) {
            if constexpr (sizeof(TX) == 1) xr32[ks] = (int)(int8_t)__builtin_amdgcn_raw_buffer_load_b8(xrsrc, xo[ks], so, 0);
            else xr32[ks] = (int)__builtin_amdgcn_raw_buffer_load_b32(xrsrc, xo[ks], so, 0);
        }
    };
    if constexpr (DEFER) load_raw(dir ? p.T - 1 : 0);
    else if constexpr (F16P) load_x16(dir ? p.T - 1 : 0, xn16);
    else if constexpr (DEFER32) load_raw32(dir ? p.T - 1 : 0);
    else load_x(dir ? p.T - 1 : 0, xn);
    __syncthreads();

    for (int step = 0; step < p.T; ++step) {
        const int t = dir ? p.T - 1 - step : step;
        const int cur = step & 1;
        float xa[kFusedKS];
#pragma unroll
        for (int ks = 0; ks < kFusedKS; ++ks) xa[ks] = DEFER32 ? (float)xr32[ks] : xn[ks];
        f32x4v acc[4];
        // gates = bias + x_t W_ih^T   (clair3/model.py:131-132: x.float() then LSTM1); the bias is the C operand of the
        // first MFMA (a resident 4-register vector per gate) instead of 16 register moves per step
        stamp(step, 0);
        if constexpr (F16P) {
            if constexpr (DEFER) widen_raw(xn16);
            const f16x8v xh = __builtin_bit_cast(f16x8v, xn16);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, __builtin_bit_cast(f16x8v, wp16[g][1]), biasv[g], 0, 0, 0);
                acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, __builtin_bit_cast(f16x8v, wp16[g][0]), acc[g], 0, 0, 0);
            }
            if (step + 1 < p.T) {
                if constexpr (DEFER) load_raw(dir ? t - 1 : t + 1);
                else load_x16(dir ? t - 1 : t + 1, xn16);
            }
        } else {
#pragma unroll
        for (int ks = 0; ks < kFusedKS; ++ks)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[ks], wih[g][ks], ks == 0 ? biasv[g] : acc[g], 0, 0, 0);
        // next step's counts: requested from inside the MFMA stream (a load costs ~60 cycles of issue outside it)
        if (step + 1 < p.T) {
            if constexpr (DEFER32) load_raw32(dir ? t - 1 : t + 1);
            else load_x(dir ? t - 1 : t + 1, xn);
        }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (F16) {
            if (step > 0) {
#pragma unroll
                for (int ks = 0; ks < H / 32; ++ks) {
                    const f16x8v a0 = *reinterpret_cast<const f16x8v *>(hb16(cur, 0, col, 32 * ks + 8 * s));
                    const f16x8v a1 = *reinterpret_cast<const f16x8v *>(hb16(cur, 1, col, 32 * ks + 8 * s));
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f16x8v v0 = __builtin_bit_cast(f16x8v, wres[g * NQ + 2 * ks]), v1 = __builtin_bit_cast(f16x8v, wres[g * NQ + 2 * ks + 1]);
                        acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, v0, acc[g], 0, 0, 0);
                        acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, v1, acc[g], 0, 0, 0);
                        acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, v0, acc[g], 0, 0, 0);
                    }
                }
            }
        } else if (step > 0) {  // gates += h_{t-1} W_hh^T, h_{-1} = 0
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const f32x4v a = *reinterpret_cast<const f32x4v *>(&hbuf[cur][col][16 * q + 4 * s]);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], wres[g * NQ + q][e], acc[g], 0, 0, 0);
            }
        }
        stamp(step, 1);
#pragma unroll
        for (int v = 0; v < (HALF ? 2 : 4); v += 2) {  // two cells per packed instruction
            f32x2g cc = {c[v], c[v + 1]};
            const f32x2g h = pk_lstm_cell(f32x2g{acc[0][v], acc[0][v + 1]}, f32x2g{acc[1][v], acc[1][v + 1]},
                                          f32x2g{acc[2][v], acc[2][v + 1]}, f32x2g{acc[3][v], acc[3][v + 1]}, cc);
            c[v] = cc[0], c[v + 1] = cc[1];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                if constexpr (F16) {
                    const _Float16 h0 = (_Float16)h[e];
                    const _Float16 h1 = (_Float16)(h[e] - (float)h0);
                    *reinterpret_cast<_Float16 *>(hb16(cur ^ 1, 0, 4 * s + v + e, wave * 16 + col)) = h0;
                    *reinterpret_cast<_Float16 *>(hb16(cur ^ 1, 1, 4 * s + v + e, wave * 16 + col)) = h1;
                    if (planes) continue;  // copied out of LDS one step later (copy_planes)
                } else {
                    hbuf[cur ^ 1][4 * s + v + e][wave * 16 + col] = h[e];
                }
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(h[e]), hrsrc, ho[v + e] + (uint32_t)(t * 2 * H * 4), 0, 0);
            }
        }
        stamp(step, 2);
        lds_barrier();
        stamp(step, 3);
        if constexpr (F16)
            if (planes) copy_planes(cur ^ 1, t);  // the tile this step wrote; the next step writes the other buffer
    }
}

}  // namespace c3

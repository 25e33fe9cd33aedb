// c3_kernels.h -- the non-GEMM kernels: persistent bidirectional LSTM recurrence, spatial-pyramid
// max-pool, and the fused FC tail (split-K reduce + SELU + branches + heads + SELU + soft-max).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "c3_gemm.h"

namespace c3 {

// nn.SELU constants (torch/nn/functional.py selu; same values as TF)
__device__ __forceinline__ float selu_f(float x) {
    const float kScale = 1.0507009873554804934193349852946f;
    const float kAlpha = 1.6732632423543772848170429916717f;
    return x > 0.f ? kScale * x : kScale * kAlpha * expm1f(x);
}
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }

// ------------------------------------------------------------------------------- LSTM recurrence
// One workgroup = 16 windows x one direction, all T steps (clair3/model.py:132-133, torch.nn.LSTM:
// gates = W_ih x_t + b_ih + W_hh h_{t-1} + b_hh, rows i,f,g,o; c' = s(f)c + s(i)tanh(g); h' = s(o)tanh(c')).
// The x-projection (+ both biases) of every step is hoisted into one matrix product (gx2, c3_dense.h); the kernel below adds
// the recurrent term on the matrix cores and applies the cell; the cell state c stays in registers for all T steps.

typedef float f32x4v __attribute__((ext_vector_type(4)));

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding global
// store (vmcnt(0)); in the recurrence the h_t stores to HBM are consumed by a later kernel, so waiting for
// their ~1-2 us write latency at every step is pure serialisation.  LDS writes are complete at lgkmcnt(0).
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}


// Gate non-linearities on the hardware exp2 / rcp units (v_exp_f32, v_rcp_f32: ~1 ulp each).  Absolute error
// of a gate value <= ~2e-7, two orders of magnitude inside the 1e-4 parity budget even after 66 steps; the
// libm forms (expf / tanhf) cost ~10x the instructions and sit on the per-step critical path.
// The reciprocal is the bare v_rcp_f32 (1 ulp): __frcp_rn is a correctly rounded 1/x, i.e. the ten-instruction
// v_div_scale / v_div_fmas / v_div_fixup sequence -- 200 of the 330 vector instructions of an LSTM step, and under an
// fp32 MFMA stream every vector instruction costs matrix-pipe time.
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
// tanh(x) = 1 - 2 / (1 + e^2x) loses what makes small arguments small: its absolute error is an ulp of 1 (6e-8) whatever x is,
// i.e. a RELATIVE error of 6e-8 / |x|.  One cell does not care; a recurrence whose weights make it sensitive does -- on LSTM
// weights with a few +-8 entries the rows of some windows moved by 2e-4 against the reference's 1e-5, and an emulation with exact
// sums pinned it on this formula alone (per-row error of LSTM2: median 1.2e-6 / worst 6e-4 with it, 1.5e-7 / 5e-5 without;
// tests/diag/fuzz_parity.py found the window, tests/test_parity_gpu.py::test_sensitive_recurrence keeps it).  Below |x| = 1/4 the
// odd series x + x^3 (c0 + c1 x^2 + c2 x^4 + c3 x^6) is good to 1.7e-7 relative (the x^11 term is 8e-9 there); above, the
// formula's 6e-8 absolute is < 5e-7 relative.
constexpr float kTanhSmall = 0.25f, kTanhC0 = -1.f / 3.f, kTanhC1 = 2.f / 15.f, kTanhC2 = -17.f / 315.f, kTanhC3 = 62.f / 2835.f;
__device__ __forceinline__ float fast_tanh(float x) {
    const float s = x * x;
    const float p = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(kTanhC3, s, kTanhC2), s, kTanhC1), s, kTanhC0);
    const float small = __builtin_fmaf(x * s, p, x);
    const float big = 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __expf(2.f * x));
    return __builtin_fabsf(x) < kTanhSmall ? small : big;
}

// Two cells at a time on the packed fp32 instructions (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 cost the same 8
// matrix-pipe cycles as their scalar forms); only the transcendentals stay scalar.
typedef float f32x2g __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2g pk_sigmoid(f32x2g x) {
    const f32x2g e = x * -1.4426950408889634f;
    const f32x2g d = f32x2g{__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])} + 1.f;
    return f32x2g{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
}
__device__ __forceinline__ f32x2g pk_tanh(f32x2g x) {  // fast_tanh, two at a time
    const f32x2g e = x * 2.8853900817779268f;
    const f32x2g d = f32x2g{__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])} + 1.f;
    const f32x2g r = f32x2g{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    const f32x2g big = __builtin_elementwise_fma(r, f32x2g{-2.f, -2.f}, f32x2g{1.f, 1.f});
    const f32x2g s = x * x;
    f32x2g p = __builtin_elementwise_fma(f32x2g{kTanhC3, kTanhC3}, s, f32x2g{kTanhC2, kTanhC2});
    p = __builtin_elementwise_fma(p, s, f32x2g{kTanhC1, kTanhC1});
    p = __builtin_elementwise_fma(p, s, f32x2g{kTanhC0, kTanhC0});
    const f32x2g small = __builtin_elementwise_fma(x * s, p, x);
    return f32x2g{__builtin_fabsf(x[0]) < kTanhSmall ? small[0] : big[0], __builtin_fabsf(x[1]) < kTanhSmall ? small[1] : big[1]};
}
// c' = s(f) c + s(i) tanh(g), returns h' = s(o) tanh(c')   (torch.nn.LSTM cell, gate order i, f, g, o)
__device__ __forceinline__ f32x2g pk_lstm_cell(f32x2g gi, f32x2g gf, f32x2g gg, f32x2g go, f32x2g &c) {
    const f32x2g ig = pk_sigmoid(gi), fg = pk_sigmoid(gf), gt = pk_tanh(gg), og = pk_sigmoid(go);
    c = __builtin_elementwise_fma(fg, c, ig * gt);
    return og * pk_tanh(c);
}

// H = 160 gives 10 unit blocks, which do not divide evenly over the SIMDs.
// The 4H/16 gate-column blocks (natural PyTorch order n = gate*H + unit) are dealt evenly to 8 waves
// (H/32 blocks each), which balances the matrix pipes exactly and lets EVERY wave keep its W_hh fragments
// resident (H/32 * H/16 * 4 = 200 VGPRs at H = 160: the whole 400 KiB matrix lives in the CU's register file
// instead of being re-streamed from L2 every step, which was bandwidth-bound at ~15 B/clk/CU).
// The price is that i/f/g/o of one unit now sit in different waves: the gate pre-activations go through LDS
// once per step (16 x 4H floats), then thread (window, units) applies the cell.  The x-projection of step t+1 has
// exactly the accumulators' layout: it is requested between the MFMAs of step t into spare registers and enters step
// t+1 as the C operand of the first MFMA of each block.
struct Lstm2Params {
    const float *gx;   // [B*T][ld_gx]; column = dir*4H + gate*H + unit  (PyTorch gate-row order)
    const float *whh;  // [dir][gate block 4H/16][q = H/16][lane][4]
    float *hout;       // [B][T][2H]; column = dir*H + unit
    int B, T;
    int64_t ld_gx;
    unsigned long long *trace = nullptr;  // OPT bit 3 (C3HIP_LSTM_TRACE): shader-clock stamps of workgroup (0, 0), [wave][step][4]
};

// F16: h_{t-1} W_hh^T on v_mfma_f32_16x16x32_f16 with both operands as two fp16 pieces (fp16x3, c3_gemm.h SPLIT mode 2):
// W_hh fragment slot q holds piece q & 1 of k-step q >> 1 (32 k per step; same bytes, same addressing as the fp32
// fragments), the cell phase stores h as two fp16 planes, and a block's 40 fp32 matrix instructions of 32 cycles become
// 15 of 16.
typedef _Float16 f16x8v __attribute__((ext_vector_type(8)));
template <int H, bool F16 = false, int OPT = 0>
__global__ __launch_bounds__(512, 2) void lstm_recurrent_kernel_v2(Lstm2Params p) {
    constexpr int NB = H / 32;        // gate-column blocks per wave (8 waves)
    constexpr int NQ = H / 16;        // k groups of 16
    constexpr int NQL = 2;            // k groups of W_hh kept in LDS; the other NQ-2 (160 VGPRs at H=160) in registers
    constexpr int NQR = NQ - NQL;
    constexpr int LDH = H + 4;        // h tile row stride (floats)
    constexpr int LDG = 4 * H + 4;    // gate tile row stride (floats, 16-byte multiple)
    constexpr int NP = 16 * H / 512;  // (window, unit) pairs per thread in the cell phase
    constexpr int WCOLS = NB * 16;    // gate columns owned by one wave
    static_assert(NP * 32 == H, "cell phase: thread (row = tid >> 5) covers units (tid & 31) + 32 r");
    // OPT bit 2: HALF tiles -- 8 windows per workgroup on rows {0,1,4,5,8,9,12,13} of the 16-row matrix tile (elements v = 0, 1 of
    // every lane's accumulators), the other rows stay zero.  The matrix phase is unchanged, the gate exchange and the x loads are
    // half as long and the cell phase is a packed pair + one single (lanes 0-31) per thread instead of two pairs + a single: for
    // batches whose full tiles leave half the CUs without a workgroup (1024 windows = 128 of them).
    constexpr bool HALF = F16 && (OPT & 4) != 0;
    constexpr int NW = HALF ? 8 : 16;   // windows per workgroup
    constexpr int NV = HALF ? 2 : 4;    // accumulator elements (= windows) per lane in use
    static_assert(!HALF || H == 160, "half tiles: 8 windows x 160 units over 512 threads as unit, unit + 64, unit + 128 (lanes 0-31)");
    // ONE LDS object, carved up by hand
    constexpr int OFF_H = 0;                                 // float hbuf[2][16][LDH]
    constexpr int LDH16 = 2 * H + 16;                        // F16: bytes per row of an fp16 h plane (see hb16)
    constexpr int H_BYTES = F16 ? 2 * 2 * 16 * LDH16 : 2 * 16 * LDH * 4;
    constexpr int OFF_G = OFF_H + H_BYTES;                   // float gbuf[16][LDG]: gate pre-activations
    constexpr int OFF_W = OFF_G + 16 * LDG * 4;              // float wlds[NQL][8][NB][64][4]
    constexpr int LDS_BYTES = OFF_W + NQL * 8 * NB * 64 * 16;
    __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
    auto hb = [&](int buf, int row, int k) -> float * { return reinterpret_cast<float *>(smem + OFF_H) + (buf * 16 + row) * LDH + k; };
    auto gb = [&](int row, int n) -> float * { return reinterpret_cast<float *>(smem + OFF_G) + row * LDG + n; };
    // F16: hbuf[2 buffers][2 pieces][16 rows][LDH16 bytes] inside the same region; rows 2 H + 16 bytes apart, so the 16
    // rows a ds_read_b128 touches fall on 16 different 16-byte slots of the 256-byte bank row
    static_assert(!F16 || H % 32 == 0, "F16 needs whole 32-wide k-steps");
    auto hb16 = [&](int buf, int piece, int row, int k) -> char * { return smem + OFF_H + ((buf * 2 + piece) * 16 + row) * LDH16 + 2 * k; };

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s = lane >> 4, col = lane & 15;
    const int dir = blockIdx.y;
    const int b0 = blockIdx.x * NW;

    for (int i = tid; i < (F16 ? H_BYTES / 4 : 16 * LDH); i += 512) reinterpret_cast<float *>(smem + OFF_H)[i] = 0.f;

    // W_hh fragments of this wave's NB blocks: k-groups 0..NQR-1 resident in VGPRs, the last NQL in LDS
    f32x4v wres[NB][NQR];
    float *wl = reinterpret_cast<float *>(smem + OFF_W) + ((wave * NB) * 64 + lane) * 4;  // + (ql*8*NB + b)*256
    {
        const float *wb = p.whh + ((int64_t)(dir * (4 * H / 16) + wave * NB) * NQ * 64 + lane) * 4;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
#pragma unroll
            for (int q = 0; q < NQR; ++q) wres[b][q] = *reinterpret_cast<const f32x4v *>(wb + (int64_t)(b * NQ + q) * 256);
#pragma unroll
            for (int ql = 0; ql < NQL; ++ql)
                *reinterpret_cast<f32x4v *>(wl + (ql * 8 * NB + b) * 256) =
                    *reinterpret_cast<const f32x4v *>(wb + (int64_t)(b * NQ + NQR + ql) * 256);
        }
    }

    // x-projection: lane (col, s) of this wave needs gx[window 4s+v][t][its NB column blocks] -- exactly the layout of
    // its accumulators, so the values are loaded STRAIGHT INTO the accumulators (buffer loads: one loop-invariant
    // vector offset per v, everything else scalar) right after the previous step's pre-activations have left them, and
    // fly during the cell phase.  No staging, no register moves, no read-modify-write in LDS.
    const int swave = __builtin_amdgcn_readfirstlane(wave);
    const __amdgpu_buffer_rsrc_t xrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.gx), 0, (uint32_t)((int64_t)p.B * p.T * p.ld_gx * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t hrsrc =
        __builtin_amdgcn_make_buffer_rsrc(p.hout, 0, (uint32_t)((int64_t)p.B * p.T * 2 * H * 4), 0x00020000);
    uint32_t xo[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        int b = b0 + (HALF ? 2 : 4) * s + v;
        if (b >= p.B) b = p.B - 1;
        xo[v] = (uint32_t)(((int64_t)b * p.T * p.ld_gx + col) * 4);
    }
    // xn holds the x-projection of the NEXT step: requested from inside the current step's MFMA stream (a load costs
    // ~60 cycles of issue outside it, next to nothing between two MFMAs) and handed to the next step as the C operand of
    // its first MFMA per block -- no register move.
    f32x4v acc[NB], xn[NB];
    auto load_x = [&](int t) __attribute__((always_inline)) {
        const int so = (int)((t * p.ld_gx + dir * 4 * H + swave * WCOLS) * 4);
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int v = 0; v < NV; ++v)
                xn[b][v] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrsrc, xo[v], so + b * 64, 0));
    };
    if constexpr (HALF) {
#pragma unroll
        for (int b = 0; b < NB; ++b) xn[b] = f32x4v{0.f, 0.f, 0.f, 0.f};  // elements 2, 3: rows without a window
    }

    // cell phase: thread (row = tid >> 5, units (tid & 31) + 32 r): every LDS / output address is one base + immediates
    // (HALF: window slot j = tid >> 6 on tile row 4 (j >> 1) + (j & 1), units (tid & 63), + 64 and, lanes 0-31, + 128)
    const int cwin = HALF ? tid >> 6 : tid >> 5;
    const int crow = HALF ? 4 * (cwin >> 1) + (cwin & 1) : cwin, cu0 = HALF ? tid & 63 : tid & 31;
    float *gbase = gb(crow, cu0);
    uint32_t hobase = b0 + cwin < p.B ? (uint32_t)((((int64_t)(b0 + cwin) * p.T) * 2 * H + dir * H + cu0) * 4) : 0x80000000u;
    float *gwr = gb(4 * s, (wave * NB) * 16 + col);  // + v * LDG + b * 16

    float c[NP];
#pragma unroll
    for (int r = 0; r < NP; ++r) c[r] = 0.f;
    load_x(dir ? p.T - 1 : 0);
    __syncthreads();

    auto put_h = [&](int buf, int row, int k, float h) __attribute__((always_inline)) {
        if constexpr (F16) {
            const _Float16 h0 = (_Float16)h;
            const _Float16 h1 = (_Float16)(h - (float)h0);
            *reinterpret_cast<_Float16 *>(hb16(buf, 0, row, k)) = h0;
            *reinterpret_cast<_Float16 *>(hb16(buf, 1, row, k)) = h1;
        } else {
            *hb(buf, row, k) = h;
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
    for (int step = 0; step < p.T; ++step) {
        const int t = dir ? p.T - 1 - step : step;
        const int cur = step & 1;
        stamp(step, 0);
        if (step == 0) {  // h_{-1} = 0: the pre-activations are the x-projection alone
#pragma unroll
            for (int b = 0; b < NB; ++b) acc[b] = xn[b];
            if (p.T > 1) load_x(dir ? t - 1 : t + 1);
        } else if constexpr (F16) {
            // k-step ks covers h units 32 ks .. 32 ks + 31; lane (window = col, group s) holds units 32 ks + 8 s .. + 7 of
            // both pieces; products h1 w0 + h0 w1 + h0 w0 per gate block (fragment slot 2 ks + piece)
            constexpr int NKS = H / 32;
            f16x8v a0[2], a1[2];
            a0[0] = *reinterpret_cast<const f16x8v *>(hb16(cur, 0, col, 8 * s));
            a1[0] = *reinterpret_cast<const f16x8v *>(hb16(cur, 1, col, 8 * s));
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                if (ks + 1 < NKS) {
                    a0[(ks + 1) & 1] = *reinterpret_cast<const f16x8v *>(hb16(cur, 0, col, 32 * (ks + 1) + 8 * s));
                    a1[(ks + 1) & 1] = *reinterpret_cast<const f16x8v *>(hb16(cur, 1, col, 32 * (ks + 1) + 8 * s));
                }
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    f32x4v w0, w1;
                    if (2 * ks < NQR) {
                        w0 = wres[b][2 * ks < NQR ? 2 * ks : 0], w1 = wres[b][2 * ks + 1 < NQR ? 2 * ks + 1 : 0];
                    } else {
                        w0 = *reinterpret_cast<const f32x4v *>(wl + ((2 * ks - NQR) * 8 * NB + b) * 256);
                        w1 = *reinterpret_cast<const f32x4v *>(wl + ((2 * ks + 1 - NQR) * 8 * NB + b) * 256);
                    }
                    const f16x8v v0 = __builtin_bit_cast(f16x8v, w0), v1 = __builtin_bit_cast(f16x8v, w1);
                    f32x4v t = ks == 0 ? xn[b] : acc[b];
                    t = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[ks & 1], v0, t, 0, 0, 0);
                    t = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0[ks & 1], v1, t, 0, 0, 0);
                    acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0[ks & 1], v0, t, 0, 0, 0);
                }
                if (ks == 0 && step + 1 < p.T) load_x(dir ? t - 1 : t + 1);  // xn was consumed by the MFMAs just issued
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            f32x4v a[2];
            a[0] = *reinterpret_cast<const f32x4v *>(hb(cur, col, 4 * s));
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                if (q + 1 < NQ) a[(q + 1) & 1] = *reinterpret_cast<const f32x4v *>(hb(cur, col, 16 * (q + 1) + 4 * s));
                if (q < NQR) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int b = 0; b < NB; ++b)
                            acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q & 1][e], wres[b][q][e], q == 0 && e == 0 ? xn[b] : acc[b], 0, 0, 0);
                    if (q == 0 && step + 1 < p.T) load_x(dir ? t - 1 : t + 1);  // xn was consumed by the MFMAs just issued
                } else {
#pragma unroll
                    for (int b = 0; b < NB; ++b) {
                        const f32x4v w = *reinterpret_cast<const f32x4v *>(wl + ((q - NQR) * 8 * NB + b) * 256);
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q & 1][e], w[e], acc[b], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        stamp(step, 1);
        // gate pre-activations (x-projection + recurrent part) -> LDS, then the next step's x-projection into the
        // freed accumulators
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int v = 0; v < NV; ++v) gwr[v * LDG + b * 16] = acc[b][v];
        __syncthreads();
        stamp(step, 2);
        // cell: c' = s(f) c + s(i) tanh(g), h' = s(o) tanh(c')   (rows i, f, g, o)
        const uint32_t ho = hobase + (uint32_t)(t * 2 * H * 4);
        if constexpr (HALF) {
            {
                const float *g0 = gbase, *g1 = g0 + 64;
                f32x2g cc = {c[0], c[1]};
                const f32x2g h = pk_lstm_cell(f32x2g{g0[0], g1[0]}, f32x2g{g0[H], g1[H]}, f32x2g{g0[2 * H], g1[2 * H]},
                                              f32x2g{g0[3 * H], g1[3 * H]}, cc);
                c[0] = cc[0], c[1] = cc[1];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    put_h(cur ^ 1, crow, cu0 + 64 * e, h[e]);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(h[e]), hrsrc, ho + 256 * e, 0, 0);
                }
            }
            if (cu0 < 32) {
                const float *g = gbase + 128;
                const float ig = fast_sigmoid(g[0]);
                const float fg = fast_sigmoid(g[H]);
                const float gg = fast_tanh(g[2 * H]);
                const float og = fast_sigmoid(g[3 * H]);
                c[2] = __builtin_fmaf(fg, c[2], ig * gg);  // the pair form's order, spelled out: left to contraction the two tile shapes compiled differently
                const float h = og * fast_tanh(c[2]);
                put_h(cur ^ 1, crow, cu0 + 128, h);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(h), hrsrc, ho + 512, 0, 0);
            }
        } else {
#pragma unroll
        for (int r = 0; r + 1 < NP; r += 2) {  // units cu0 + 32 r and cu0 + 32 (r + 1) as one packed pair
            const float *g0 = gbase + 32 * r, *g1 = g0 + 32;
            f32x2g cc = {c[r], c[r + 1]};
            const f32x2g h = pk_lstm_cell(f32x2g{g0[0], g1[0]}, f32x2g{g0[H], g1[H]}, f32x2g{g0[2 * H], g1[2 * H]},
                                          f32x2g{g0[3 * H], g1[3 * H]}, cc);
            c[r] = cc[0], c[r + 1] = cc[1];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                put_h(cur ^ 1, crow, cu0 + 32 * (r + e), h[e]);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(h[e]), hrsrc, ho + 128 * (r + e), 0, 0);
            }
        }
        if (NP & 1) {
            constexpr int r = NP - 1;
            const float *g = gbase + 32 * r;
            const float ig = fast_sigmoid(g[0]);
            const float fg = fast_sigmoid(g[H]);
            const float gg = fast_tanh(g[2 * H]);
            const float og = fast_sigmoid(g[3 * H]);
            c[r] = __builtin_fmaf(fg, c[r], ig * gg);
            const float h = og * fast_tanh(c[r]);
            put_h(cur ^ 1, crow, cu0 + 32 * r, h);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(h), hrsrc, ho + 128 * r, 0, 0);
        }
        }
        stamp(step, 3);
        lds_barrier();
    }
}

// ------------------------------------------------------------------------------- spatial pyramid pooling
// clair3/model.py:250-279.  in: NHWC fp32 (B, H, W, C) ; out: (B, nbins*C) in (bin, c) order where bins run
// pool 3 (h, w), pool 2 (h, w), pool 1 -- the reference's permute(0,2,3,1)+flatten+cat order.
struct SppParams {
    const float *in;
    float *out;
    int B, H, W, C, nbins;
    // window of every bin clipped to the image; pad != 0 when the un-clipped window reached into the zero
    // padding added by F.pad (model.py:267-268), in which case 0 takes part in the max.
    short h0[16], h1[16], w0[16], w1[16], pad[16];
};
// bins described at run time (spp_bins in the host code); the fp32-activation form of the range-guard fallback
__global__ __launch_bounds__(256) void spp_kernel(SppParams p) {
    const int64_t total = (int64_t)p.B * p.nbins * p.C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % p.C);
        const int bin = (int)((i / p.C) % p.nbins);
        const int64_t b = i / ((int64_t)p.C * p.nbins);
        const float *src = p.in + b * p.H * p.W * p.C + c;
        float m = p.pad[bin] ? 0.f : -INFINITY;
        for (int h = p.h0[bin]; h < p.h1[bin]; ++h)
            for (int w = p.w0[bin]; w < p.w1[bin]; ++w) m = fmaxf(m, src[((int64_t)h * p.W + w) * p.C]);
        p.out[i] = m;
    }
}

}  // namespace c3

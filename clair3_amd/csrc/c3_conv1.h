// c3_conv1.h -- the first full-alignment convolution (3x3, stride 2, pad 1, 8 int8 channels -> 64, BN, ReLU;
// clair3/model.py:316-317,391) as a barrier-free, LDS-free MFMA kernel of its own.  In the product conv1 is computed INSIDE the
// first residual block (c3_conv3.h SRC8) and this launch does not exist; it runs when every layer keeps its own output
// (c3_debug_keep_activations, the layer-by-layer parity tests) and behind C3HIP_CONV1_FUSED=0.
//
// K = 72 is three k-steps of a tiled implicit GEMM -- a workgroup's prologue and epilogue would outweigh its main loop.  Here
// nothing is shared between waves, so nothing is synchronised: a wave keeps the WHOLE weight matrix as matrix-instruction
// fragments in registers and walks over groups of 32 output pixels; padding taps are redirected out of range of the buffer
// descriptor (the load returns 0 = the zero padding); the next group's taps are requested before the matrix instructions of
// the current one.

#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "c3_gemm.h"

namespace c3 {

// The int8 window is EXACT in fp16, so only the weights need two pieces: x * w = x h0 + x h1 on
// v_mfma_f32_32x32x16_f16 (fp32 accumulation) -- two matrix instructions of 32 cycles per 16 k and column block instead
// of eight of 64.  K order = (tap, channel): k-step t holds taps 2t (lanes 0-31) and 2t + 1 (lanes 32-63), a lane loads
// the 8 channels of its tap as one 8-byte piece; tap 9 does not exist (out-of-range offset -> zeros, zero weights).
// int8 -> fp16 without a conversion instruction: byte u = b ^ 0x80 or-ed into 0x4800 is the fp16 number 8 + u / 128, so
// v_perm_b32 builds two such halves per dword and one v_pk_add_f16 of -9 yields b / 128 twice, exactly.  The 1/128 is
// there for the WEIGHTS: with x.float()/100 folded in they are ~1e-3 and the low fp16 piece of such a number is
// subnormal (6 bits left: conv1 was 20x less accurate that way); the host folds 128/100 instead, which keeps both
// pieces normal and the product unchanged.
typedef _Float16 c1_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 c1_f16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t c1_u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t c1_u32x4 __attribute__((ext_vector_type(4)));

struct Conv1F16Params {
    const int8_t *x;        // [B][H][W][8]
    const uint32_t *wfrag;  // [5 k-steps][2 column blocks][2 pieces][64 lanes][4 dwords = 8 fp16]: piece of
                            // 1.28 W'[cout = 32 cb + (lane & 31)][tap = 2 t + (lane >> 5)][channel j]  (W' has BN and /100 folded)
    const float *bias;      // [64]
    const float *post;      // [64] 2^-k of the output channel: the fragments are packed times 2^k (c3_pack.h row_scales)
    float *out;             // [B][OH][OW][64]
    uint32_t *range_flag;   // set to 1 when an output reaches kF16Range (c3_gemm.h)
    int B, H, W, OH, OW, M, groups;
};

// PLANES: the output leaves as plane activations (c3_conv3.h).  The two matrix operands swap places, so the accumulators
// hold the block transposed -- lane = pixel, four consecutive channels per v >> 2.
template <bool PLANES = true>
__global__ __launch_bounds__(256, 2) void conv1_i8_f16_kernel(Conv1F16Params p) {
    static_assert(PLANES, "the product writes plane activations (the fp32-activation form of conv1 is the tiled contraction)");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 31, kh = lane >> 5;
    const int gw = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
    __shared__ __attribute__((aligned(16))) float bias_lds[PLANES ? 128 : 4];
    float *const post_lds = bias_lds + 64;
    if constexpr (PLANES) {  // before any wave leaves
        if (tid < 64) bias_lds[tid] = p.bias[tid], post_lds[tid] = p.post[tid];
        __syncthreads();
    }
    if (gw >= p.groups) return;

    const int rowB = p.W * 8;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(p.x)) - (rowB + 8), 0, p.B * p.H * rowB + rowB + 8, 0x00020000);
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.M * 256, 0x00020000);

    c1_u32x4 wf[5][2][2];
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int q = 0; q < 2; ++q) wf[t][cb][q] = *reinterpret_cast<const c1_u32x4 *>(p.wfrag + (((t * 2 + cb) * 2 + q) * 64 + lane) * 4);
    // !PLANES: the bias is the C operand of a group's first matrix instructions (one resident value per column block).
    // PLANES: the transposed accumulators would need 16 distinct bias values per column block = 32 resident registers, and
    // with them the kernel spilled 12 dwords -- whose reloads sit in the request code, each behind an s_waitcnt vmcnt(0)
    // that (one in-order counter for loads, stores and scratch) also waits for the tap loads just issued and for the
    // previous group's stores.  The groups start from the constant 0 instead and store_group() adds the bias from LDS.
    f32x16 biasv[PLANES ? 1 : 2];
    if constexpr (!PLANES) {
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const float b = p.bias[32 * cb + m];
#pragma unroll
            for (int v = 0; v < 16; ++v) biasv[cb][v] = b;
        }
    }
    // this lane's tap of every k-step: (ky, kx) and its byte offset from the (shifted) pixel base
    int ky[5], kx[5];
    uint32_t toff[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        const int tap = 2 * t + kh;
        ky[t] = tap / 3, kx[t] = tap - 3 * ky[t];
        toff[t] = (uint32_t)(ky[t] * rowB + kx[t] * 8);
    }
    const int ohw = p.OH * p.OW;
    auto request = [&](int g, c1_u32x2 (&d)[5]) __attribute__((always_inline)) {
        const int pix = g * 32 + m;
        const int b = pix / ohw, r = pix - b * ohw;
        const int oy = r / p.OW, ox = r - oy * p.OW;
        const uint32_t base = (uint32_t)(((b * p.H + 2 * oy) * p.W + 2 * ox) * 8);
        const bool valid = pix < p.M;
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            const int iy = 2 * oy - 1 + ky[t], ix = 2 * ox - 1 + kx[t];
            const bool ok = valid && (2 * t + kh < 9) && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            d[t] = __builtin_bit_cast(c1_u32x2, __builtin_amdgcn_raw_buffer_load_b64(xrsrc, ok ? base + toff[t] : 0x80000000u, 0, 0));
        }
    };
    // 8 int8 -> 8 exact fp16 values b / 128 (see the header comment); out-of-range loads returned 0 = the zero padding
    auto widen = [&](c1_u32x2 d) __attribute__((always_inline)) {
        c1_u32x4 o;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t x = d[h] ^ 0x80808080u;
            const uint32_t p01 = __builtin_amdgcn_perm(0x48484848u, x, 0x04010400u);  // [x.b0, 0x48, x.b1, 0x48]
            const uint32_t p23 = __builtin_amdgcn_perm(0x48484848u, x, 0x04030402u);  // [x.b2, 0x48, x.b3, 0x48]
            const c1_f16x2 bias2 = {(_Float16)-9.0f, (_Float16)-9.0f};
            o[2 * h] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(c1_f16x2, p01) + bias2);
            o[2 * h + 1] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(c1_f16x2, p23) + bias2);
        }
        return o;
    };
    int omax_i = 0;
    // PLANES: the 32 x 64 block of the previous group goes through a wave-private LDS tile in its final byte order
    // ([pixel][hi 64 x fp16 | lo 64 x fp16], rows 272 B apart), written as 8-byte pieces by the lanes that own them
    // (slots 0, 2, .., 14: group 4 cb + q = channels 32 cb + 8 q + 4 kh .. + 3 of pixel m) and read back as 16-byte pieces
    // that leave in ONE contiguous kilobyte per store instruction (slots 24..31) -- strided 8-byte stores cost 9 us per
    // 50 MB here.  Same wave, program order: no barrier, the compiler's lgkmcnt waits order write -> read -> next write.
    __shared__ __attribute__((aligned(16))) char st_all[PLANES ? 4 * 32 * 272 : 16];
    char *const st = st_all + (PLANES ? wave * 32 * 272 : 0);
    auto store_group = [&](const f32x16 (&r)[2], int idx) __attribute__((always_inline)) {
        const int cb = idx >> 2, q = idx & 3;
        const f32x4 b4 = *reinterpret_cast<const f32x4 *>(bias_lds + 32 * cb + 8 * q + 4 * kh);
        const f32x4 s4 = *reinterpret_cast<const f32x4 *>(post_lds + 32 * cb + 8 * q + 4 * kh);
        f32x4 val;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int relu = max(__float_as_int(__builtin_fmaf(r[cb][4 * q + e], s4[e], b4[e])), 0);
            omax_i = max(omax_i, relu);
            val[e] = __int_as_float(relu);
        }
        u32x2 pc[2];
        split2_f16(val, pc);
        char *dst = st + m * 272 + (32 * cb + 8 * q + 4 * kh) * 2;
        *reinterpret_cast<u32x2 *>(dst) = pc[0];
        *reinterpret_cast<u32x2 *>(dst + 128) = pc[1];
    };
    auto store_piece = [&](uint32_t o0, int j) __attribute__((always_inline)) {
        const int idx = lane + 64 * j;  // piece idx & 15 of pixel idx >> 4
        const c1_u32x4 d = *reinterpret_cast<const c1_u32x4 *>(st + (idx >> 4) * 272 + (idx & 15) * 16);
        __builtin_amdgcn_raw_buffer_store_b128(d, orsrc, o0 + (uint32_t)(idx * 16), 0, 0);
    };
    auto store_one = [&](const f32x16 (&r)[2], uint32_t o0, int idx) __attribute__((always_inline)) {
        if constexpr (PLANES) {
            if (idx < 16 && (idx & 1) == 0) store_group(r, idx >> 1);
            if (idx >= 24) store_piece(o0, idx - 24);
            return;
        }
        const int cb = idx >> 4, v = idx & 15;
        const uint32_t off = o0 + (uint32_t)(((v & 3) + 8 * (v >> 2)) * 256 + cb * 128);
        const float val = r[cb][v];
        const int relu = max(__float_as_int(val), 0);
        omax_i = max(omax_i, relu);  // non-negative floats order like their bit patterns
        __builtin_amdgcn_raw_buffer_store_b32((uint32_t)relu, orsrc, off, 0, 0);
    };
    c1_u32x2 d[5];
    request(gw, d);
    auto body = [&](int g, f32x16 (&acc)[2], const f32x16 (&prev)[2], uint32_t prev_o0) __attribute__((always_inline)) {
        c1_u32x4 a[5];
#pragma unroll
        for (int t = 0; t < 5; ++t) a[t] = widen(d[t]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 5; ++t) {
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                f32x16 c;
                if constexpr (PLANES) {
                    if (t == 0) {
#pragma unroll
                        for (int v = 0; v < 16; ++v) c[v] = 0.f;
                    } else {
                        c = acc[cb];
                    }
                } else {
                    c = t == 0 ? biasv[cb] : acc[cb];
                }
                if constexpr (PLANES) {
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(c1_f16x8, wf[t][cb][1]), __builtin_bit_cast(c1_f16x8, a[t]), c, 0, 0, 0);
                    acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(c1_f16x8, wf[t][cb][0]), __builtin_bit_cast(c1_f16x8, a[t]), c, 0, 0, 0);
                } else {
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(c1_f16x8, a[t]), __builtin_bit_cast(c1_f16x8, wf[t][cb][1]), c, 0, 0, 0);
                    acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(c1_f16x8, a[t]), __builtin_bit_cast(c1_f16x8, wf[t][cb][0]), c, 0, 0, 0);
                }
                // the previous group's 32 stores, spread over this group's 20 matrix instructions
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int idx = (t * 2 + cb) * 4 + k;
                    if (idx < 32) store_one(prev, prev_o0, idx);
                }
            }
            if (t == 0 && g + nw < p.groups) request(g + nw, d);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto out_base = [&](int g) { return PLANES ? (uint32_t)(g * 32 * 256) : (uint32_t)((g * 32 + 4 * kh) * 256 + m * 4); };
    f32x16 accA[2], accB[2] = {};  // accB is the first group's (dropped) "previous" result: defined, so that the range check sees zeros
    uint32_t prev_o0 = 0x80000000u;
    int g = gw;
    for (; g + nw < p.groups; g += 2 * nw) {
        body(g, accA, accB, prev_o0);
        body(g + nw, accB, accA, out_base(g));
        prev_o0 = out_base(g + nw);
    }
    if (g < p.groups) {
        body(g, accA, accB, prev_o0);
#pragma unroll
        for (int i = 0; i < 32; ++i) store_one(accA, out_base(g), i);
    } else {
#pragma unroll
        for (int i = 0; i < 32; ++i) store_one(accB, prev_o0, i);
    }
    if (p.range_flag && omax_i >= __float_as_int(kF16Range)) atomicOr(p.range_flag, 1u);  // +NaN patterns are larger still
}

}  // namespace c3

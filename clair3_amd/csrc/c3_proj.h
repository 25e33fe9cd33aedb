// c3_proj.h -- the LSTM2 input projection gx = h1 W_ih^T + (b_ih + b_hh) (clair3/model.py:133; rows = B*33
// (window, position) pairs, K = 256 = both directions of LSTM1, N = 1280 = 2 directions x 4 gates x 160) as an
// A-stationary streaming GEMM.
//
// With K = 256 the tiled kernel spends 8 k-steps per 128x128 tile between a prologue and a 64 KB epilogue: 103 TFLOP/s
// even at B = 16384, and 44 % of the pileup step.  Here a wave keeps a 32-row block of A as MFMA fragments in
// 128 registers (lane (row, kk) holds A[row][128 kk .. 128 kk + 127]: the k-step order k = 128 kk + j is chosen so
// that the rows load as plain 16-byte pieces), and streams the weights past it one 32-column block at a time -- fragment
// order on the host, 32 x 16-byte loads per lane and block, two quarters of 8 in flight.  No LDS, no barrier, no
// address arithmetic in the loop (scalar offsets); 128 back-to-back v_mfma_f32_32x32x2_f32 per (row block, column
// block) unit, then bias and 16 dword stores per lane (128-byte lines).  The 42240 units of the B = 1024 batch are dealt
// to the waves as contiguous ranges (+-1 unit), so a wave reloads A once or twice in its life and the chip is balanced
// to 2 % whatever the batch size.  The bias add and the 16 stores of a unit are issued between the MFMAs of the NEXT
// unit (two accumulator sets swapping roles).  Measured: 244 -> 178 us at B = 1024, 103 -> 127 TFLOP/s at B = 16384.
// Shader-clock trace (C3HIP_PROJ_TRACE=1): with the two waves of a SIMD interleaving, 64 MFMAs take ~4200 cycles
// (65.6 per MFMA; the pipe's limit is 64) and a unit boundary ~1400: the loop runs at 91 % of the matrix pipe; the rest
// of the kernel's 78 % is its start (2048 waves fetching their 32 KB of A at once), launch ramp and tail.  What did
// NOT matter (tools/stream_probe.hip, ablations): one vs two accumulation chains, 192 distinct operand registers,
// the fragment loads (-3 % when removed), L2 traffic (same time with every load hitting one KiB).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "c3_gemm.h"

namespace c3 {

struct ProjParams {
    const float *a;      // [M][256]
    const float *bfrag;  // [N/32][32][64 lanes][4]: W[n = 32 cb + (lane&31)][k = 128 (lane>>5) + 4 i + e]
    const float *bias;   // [N]
    float *out;          // [M][N]
    int M, N;
    int row_blocks, col_blocks;  // ceil(M/32), N/32
    unsigned long long *trace;   // debug (C3HIP_PROJ_TRACE): shader-clock stamps of workgroup 0, wave 0; nullptr in production
};

__global__ __launch_bounds__(256, 2) void proj_stream_kernel(ProjParams p) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 31, kk = lane >> 5;
    const int64_t units = (int64_t)p.row_blocks * p.col_blocks;
    const int gw = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
    const int u0 = (int)(units * gw / nw), u1 = (int)(units * (gw + 1) / nw);
    if (u0 >= u1) return;

    const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.a), 0, (uint32_t)((int64_t)p.M * 1024), 0x00020000);
    const __amdgpu_buffer_rsrc_t brsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.bfrag), 0, p.N * 1024, 0x00020000);
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (uint32_t)((int64_t)p.M * p.N * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t biasrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.bias), 0, p.N * 4, 0x00020000);
    const uint32_t b_lane = (uint32_t)lane * 16u;

    f32x4 bq[2][8];
    auto load_b = [&](f32x4 (&dst)[8], int cb, int qq) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            dst[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(brsrc, b_lane, (cb * 32 + qq * 8 + i) * 1024, 0));
    };

    f32x4 a[32];
    int cur_rb = -1;
    int rb = u0 / p.col_blocks, cb = u0 - rb * p.col_blocks;
    load_b(bq[0], cb, 0);
    // One unit: 128 MFMAs into `acc`, and BETWEEN them the bias add + 16 stores of the PREVIOUS unit (`prev`): a store
    // costs ~60 cycles of issue after the stream has ended and next to nothing inside it.  The two
    // accumulator sets swap roles by name; `prs` is a zero-length descriptor while there is no previous unit.
    auto unit = [&](int u, f32x16 &acc, const f32x16 &prev, uint32_t prev_o0, float prev_bias, __amdgpu_buffer_rsrc_t prs,
                    uint32_t &o0, float &bias) __attribute__((always_inline)) {
        if (rb != cur_rb) {  // new row block: its 32 rows x 256 k as fragments (rows beyond M read as zeros)
            const uint32_t aoff = (uint32_t)(rb * 32 + m) * 1024u + (uint32_t)kk * 512u;
#pragma unroll
            for (int i = 0; i < 32; ++i) a[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(arsrc, aoff, i * 16, 0));
            cur_rb = rb;
        }
        if (p.trace && gw == 0 && u - u0 < 24) p.trace[(u - u0) * 8] = __builtin_readcyclecounter();
        int nrb = rb, ncb = cb + 1;
        if (ncb == p.col_blocks) ncb = 0, nrb = rb + 1;
        // requested before this unit's fragment loads: its wait must not drain the prefetches issued after it
        bias = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(biasrsrc, (uint32_t)(cb * 32 + m) * 4u, 0, 0));
        // acc[v] = row 8 (v >> 2) + 4 kk + (v & 3) of the block, column 32 cb + m
        o0 = (uint32_t)(((int64_t)(rb * 32 + 4 * kk) * p.N + cb * 32 + m) * 4);
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            if (qq < 3) load_b(bq[(qq + 1) & 1], cb, qq + 1);  // the next quarter flies under this one's 32 MFMAs
            else if (u + 1 < u1) load_b(bq[0], ncb, 0);
            __builtin_amdgcn_sched_barrier(0);  // or the compiler sinks the loads to just before their use
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (qq == 0 && i == 0 && e == 0) {
                        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0][0], bq[0][0][0], zero, 0, 0, 0);
                    } else {
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[8 * qq + i][e], bq[qq & 1][i][e], acc, 0, 0, 0);
                    }
                }
                if (i & 1) {
                    const int v = 4 * qq + (i >> 1);
                    const float val = prev[v] + prev_bias;
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(val), prs, prev_o0 + (uint32_t)((8 * (v >> 2) + (v & 3)) * p.N * 4), 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (p.trace && gw == 0 && u - u0 < 24) p.trace[(u - u0) * 8 + 1 + qq] = __builtin_readcyclecounter();
        }
        rb = nrb, cb = ncb;
    };
    auto flush = [&](const f32x16 &prev, uint32_t prev_o0, float prev_bias) __attribute__((always_inline)) {
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const float val = prev[v] + prev_bias;
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(val), orsrc, prev_o0 + (uint32_t)((8 * (v >> 2) + (v & 3)) * p.N * 4), 0, 0);
        }
    };
    const __amdgpu_buffer_rsrc_t nullrsrc = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, 0, 0x00020000);
    f32x16 accA, accB = {};
    uint32_t oA = 0, oB = 0;
    float bA = 0.f, bB = 0.f;
    int u = u0;
    unit(u, accA, accB, oB, bB, nullrsrc, oA, bA);
    ++u;
    for (; u + 1 < u1; u += 2) {
        unit(u, accB, accA, oA, bA, orsrc, oB, bB);
        unit(u + 1, accA, accB, oB, bB, orsrc, oA, bA);
    }
    if (u < u1) {
        unit(u, accB, accA, oA, bA, orsrc, oB, bB);
        flush(accB, oB, bB);
    } else {
        flush(accA, oA, bA);
    }
}

}  // namespace c3

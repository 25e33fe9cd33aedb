// c3_dense.h -- C = A (plane activations, c3_conv3.h) x W^T + bias on v_mfma_f32_32x32x16_f16, fp16x3 products.
//
//   dense_planes_wres_kernel   the LSTM2 input projection of Clair3_P (clair3/model.py:96-107,132-133: gx2 = h1 W_ih2^T + b_ih +
//                              b_hh for both directions at once; M = 33 B rows of 256 features -> 1280 gate pre-activations)
//                              with its weights resident in registers (K = 256);
//   dense_planes_pipe_kernel   the same projection for batches too small for the kernel above (both operands through
//                              registers and ds_write); bit-identical rows (same chunk order, same instruction sequence per
//                              accumulator).
//
// Common ground: LSTM1 / the convolutions write their outputs as planes (hi / lo fp16 pieces, 64-channel slabs of 256 B), so
// both operands reach LDS as plain 16-byte copies; one workgroup (512 threads = 8 waves) per CU, persistent; fragment reads
// run one k-step ahead of the matrix instructions (pinned order, as in conv3x3_planes_kernel).
// tools/dense_probe.hip times both on their shape (and the stride-2 convolutions of c3_conv3s2.h, which ran on a third kernel of this
// family -- both operands through LDS-DMA -- until round 4) with parts switched off (the ABL template bits).
#pragma once
#include "c3_conv3.h"

namespace c3 {

constexpr int kDnBM = 128, kDnBN = 128, kDnThreads = 512;
constexpr int kDnABytes = kDnBM * kPlRowB, kDnBBytes = kDnBN * kPlRowB;  // one 64-channel chunk of each operand
constexpr int kDnStage = kDnABytes + kDnBBytes;                           // 69 632 B; two stages

struct DensePlanesParams {
    const void *a;      // plane activations [M][K/64][hi 64 | lo 64] fp16
    const void *w;      // [N/128][K/64][128 rows][16 pieces of 16 B]: pieces 0-7 = hi of k 8g..8g+7 of the chunk, 8-15 = lo; times 2^s
    const float *bias;  // [N]
    float *c;           // [M][N] fp32
    const float *post;  // [N] 2^-k of every output row / channel: its weights are packed times 2^k (c3_pack.h row_scales)
    int M, N, K;
    int tiles_n, tiles;  // N / 128, ceil(M / 128) * tiles_n
    uint32_t *range_flag = nullptr;
};

// ------------------------------------------------------------------------------------------------------------------------
// dense_planes_pipe_kernel -- both operands through registers into LDS (rows 272 B apart: conflict-free ds_read_b128 of 16
// consecutive rows, immediate (piece, k-step) offsets), the chunk stream spread over the matrix stream.
//
// A phase trace of the first form of this kernel (all loads, then all LDS writes, then the matrix instructions; shader
// cycles per 64-channel chunk) read: stage 800 -> request 1400 -> 24 matrix instructions + fragment reads 1400 -> barrier 900 =
// 4500 for 1536 cycles of matrix work on the two waves of a SIMD.  Both operands stream, so a chunk is 64 KB through the vector memory pipe of a CU
// (64 B / clk: 1024 cycles) and 64 KB through the LDS write path; issued back to back at the top of a chunk, the eight
// loads of a thread fill the memory pipe's queue and the wave stalls in the ISSUE of its loads, then in its eight LDS
// writes, before its first matrix instruction -- every wave at the same time.  Here piece j of the chunk held in registers
// goes to LDS in k-step j of the current chunk and its register is re-requested right away (one LDS write and one load
// per operand per six matrix instructions): the three pipes run side by side.
// The tile's results leave at the top of the NEXT tile (stores of the first tile's predecessor are out-of-range offsets),
// and a tile's first chunk is its own copy of the code: hipcc derives an s_waitcnt vmcnt(N) per static instruction from the
// worst path into it, and loads and stores of a wave share one in-order counter -- one copy of the chunk code for all
// chunks made the first wait after a tile's stores a wait for those stores as well.
// ABL (tools/dense_probe.hip only; 0 in the product): 1 no operand loads, 2 no LDS staging writes, 4 no matrix instructions,
// 8 no fragment reads, 16 no result stores, 32 no barriers.
template <int ABL = 0>
__global__ __launch_bounds__(kDnThreads, 2) void dense_planes_pipe_kernel(DensePlanesParams p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * kDnStage + 16384];
    float *bias_lds = reinterpret_cast<float *>(smem + 2 * kDnStage);  // the whole bias vector (N <= 2048): no global load in an epilogue
    float *post_lds = bias_lds + 2048;                                 // and the per-channel 2^-k next to it
    for (int i = threadIdx.x; i < p.N && i < 2048; i += kDnThreads) bias_lds[i] = p.bias[i], post_lds[i] = p.post[i];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;  // 2 x 4 waves
    const int frow = lane & 31, kh = lane >> 5;
    const int NK = p.K / 64;
    const int G = gridDim.x;
    const int rowb = (p.K / 64) * 256;  // bytes per row of A

    const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void *>(p.a), 0, (uint32_t)((int64_t)p.M * rowb), 0x00020000);
    const __amdgpu_buffer_rsrc_t crsrc =
        __builtin_amdgcn_make_buffer_rsrc(p.c, 0, (uint32_t)((int64_t)p.M * p.N * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.w), 0, (uint32_t)((int64_t)p.N * p.K * 4), 0x00020000);
    auto tile_mn = [&](int v, int &m0, int &tn) __attribute__((always_inline)) {
        const int tile = xcd_tile_index(v, p.tiles);
        const int tm = tile / p.tiles_n;
        tn = tile - tm * p.tiles_n;
        m0 = tm * kDnBM;
    };
    // piece j of a chunk: row (tid >> 4) + 32 j of each operand, 16-byte position tid & 15
    pl_u32x4 ra[4], rb[4];
    // `on` = false: both loads still issue, at out-of-range offsets (they return zeros).  A conditional request would give
    // hipcc paths with fewer loads in flight, and it sizes every s_waitcnt vmcnt(N) for the emptiest path into it: the wait
    // for piece j then also waits for the pieces requested a few instructions earlier.
    auto issue1 = [&](int j, int m0, int tn, int kc, bool on) __attribute__((always_inline)) {
        const int idx = tid + kDnThreads * j;
        const int m = m0 + (idx >> 4);
        const uint32_t off = on && m < p.M ? (uint32_t)m * (uint32_t)rowb + (uint32_t)(kc * 256 + (idx & 15) * 16) : kPlOob;
        const uint32_t woff = on ? (uint32_t)((tn * NK + kc) * (kDnBN * 256) + idx * 16) : kPlOob;
        if constexpr (ABL & 1) {
            ra[j] = pl_u32x4{off, woff, 0x3c003c00u, 0x3c003c00u}, rb[j] = pl_u32x4{woff, off, 0x3c003c00u, 0x3c003c00u};  // keeps the address arithmetic alive
        } else {
            ra[j] = __builtin_bit_cast(pl_u32x4, __builtin_amdgcn_raw_buffer_load_b128(arsrc, off, 0, 0));
            rb[j] = __builtin_bit_cast(pl_u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, woff, 0, 0));
        }
    };
    const int st_off = (tid >> 4) * kPlRowB + (tid & 15) * 16;  // 32 rows further per j
    auto stage1 = [&](int j, int buf) __attribute__((always_inline)) {
        char *dst = smem + buf * kDnStage + st_off + j * 32 * kPlRowB;
        if constexpr (ABL & 2) {
            if (ra[j][1] == 0x12345u && rb[j][2] == 0x54321u) *reinterpret_cast<pl_u32x4 *>(dst) = ra[j];  // never true: the registers stay live
        } else {
            *reinterpret_cast<pl_u32x4 *>(dst) = ra[j];
            *reinterpret_cast<pl_u32x4 *>(dst + kDnABytes) = rb[j];
        }
    };
    auto mma = [](f32x16 c, pl_u32x4 w, pl_u32x4 x) __attribute__((always_inline)) {
        if constexpr (ABL & 4) {
            c[0] += __uint_as_float(w[0] ^ x[0]), c[5] += __uint_as_float(w[3] ^ x[3]);
            return c;
        } else {
            return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, x), c, 0, 0, 0);
        }
    };
    const int a_rd[2] = {(wm * 64 + frow) * kPlRowB + kh * 16, (wm * 64 + 32 + frow) * kPlRowB + kh * 16};
    const int b_rd = kDnABytes + (wn * 32 + frow) * kPlRowB + kh * 16;

    int v = blockIdx.x;
    if (v >= p.tiles) return;
    int m0, tn;
    tile_mn(v, m0, tn);
#pragma unroll
    for (int j = 0; j < 4; ++j) issue1(j, m0, tn, 0, true);
#pragma unroll
    for (int j = 0; j < 4; ++j) stage1(j, 0);
    int vq = v, m0q = m0, tnq = tn, kq = 1;  // (tile, chunk) of the chunk held in registers
    if (kq == NK) {
        kq = 0, vq = v + G;
        if (vq < p.tiles) tile_mn(vq, m0q, tnq);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) issue1(j, m0q, tnq, kq, vq < p.tiles);
    lds_barrier();

    int g = 0;  // chunks done: the current chunk sits in LDS stage g & 1
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    // FIRST: the tile's first chunk -- the accumulators start from the constant 0 of its first matrix instructions
    auto chunk = [&](auto first_tag) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_tag)::value;
        const char *cur = smem + (g & 1) * kDnStage;
        const bool have_next = vq < p.tiles;  // the registers hold a chunk: it goes to the other stage during this one
        bool req = false;                     // ... and its registers are re-requested for the chunk after it
        if (have_next) {
            if (++kq == NK) {
                kq = 0, vq += G;
                if (vq < p.tiles) tile_mn(vq, m0q, tnq);
            }
            req = vq < p.tiles;
        }
        pl_u32x4 xh[2][2], xl[2][2], wh[2], wl[2];
        auto frags = [&](int ks, int st) __attribute__((always_inline)) {
            if constexpr (ABL & 8) {
                const pl_u32x4 f = {(uint32_t)(ks + g), (uint32_t)lane, 0x3c003c00u, 0x3c003c00u};
                wh[st] = wl[st] = xh[st][0] = xh[st][1] = xl[st][0] = xl[st][1] = f;
                return;
            }
            wh[st] = *reinterpret_cast<const pl_u32x4 *>(cur + b_rd + ks * 32);
            wl[st] = *reinterpret_cast<const pl_u32x4 *>(cur + b_rd + 128 + ks * 32);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                xh[st][i] = *reinterpret_cast<const pl_u32x4 *>(cur + a_rd[i] + ks * 32);
                xl[st][i] = *reinterpret_cast<const pl_u32x4 *>(cur + a_rd[i] + 128 + ks * 32);
            }
        };
        frags(0, 0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int st = ks & 1;
            if (ks < 3) frags(ks + 1, st ^ 1);
            stage1(ks, (g + 1) & 1);
            issue1(ks, m0q, tnq, kq, req);
            __builtin_amdgcn_sched_barrier(0);
            if (FIRST && ks == 0) {
                f32x16 zero;
#pragma unroll
                for (int e = 0; e < 16; ++e) zero[e] = 0.f;
                acc[0] = mma(zero, wh[st], xl[st][0]);
                acc[1] = mma(zero, wh[st], xl[st][1]);
            } else {
                acc[0] = mma(acc[0], wh[st], xl[st][0]);
                acc[1] = mma(acc[1], wh[st], xl[st][1]);
            }
            acc[0] = mma(acc[0], wl[st], xh[st][0]);
            acc[1] = mma(acc[1], wl[st], xh[st][1]);
            acc[0] = mma(acc[0], wh[st], xh[st][0]);
            acc[1] = mma(acc[1], wh[st], xh[st][1]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (!(ABL & 32)) lds_barrier();
        ++g;
    };
    // the finished tile (pm0, ptn): out-of-range offsets when there is none yet -- the same stores on every path
    auto epilogue = [&](int pm0, int ptn, bool valid) __attribute__((always_inline)) {
        const int cb0 = wn * 32 + 4 * kh;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = pm0 + wm * 64 + i * 32 + frow;
            const uint32_t rowoff = valid && m < p.M ? (uint32_t)(((int64_t)m * p.N + ptn * kDnBN + cb0) * 4) : kPlOob;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 bv = *reinterpret_cast<const f32x4 *>(bias_lds + ptn * kDnBN + cb0 + 8 * q);
                const f32x4 sv = *reinterpret_cast<const f32x4 *>(post_lds + ptn * kDnBN + cb0 + 8 * q);
                f32x4 val = {acc[i][4 * q], acc[i][4 * q + 1], acc[i][4 * q + 2], acc[i][4 * q + 3]};
#pragma unroll
                for (int e = 0; e < 4; ++e) val[e] = __builtin_fmaf(val[e], sv[e], bv[e]);
                if constexpr (ABL & 16) {
                    if (val[0] == 1234.5f) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pl_u32x4, val), crsrc, rowoff + 32 * q, 0, 0);
                } else {
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pl_u32x4, val), crsrc, rowoff + 32 * q, 0, 0);
                }
            }
        }
    };
    int pm0 = 0, ptn = 0;
    bool have_prev = false;
    for (;;) {
        epilogue(pm0, ptn, have_prev);
        if (v >= p.tiles) break;
        chunk(std::true_type{});  // the copy behind the stores
        for (int kc = 1; kc < NK; ++kc) chunk(std::false_type{});
        pm0 = m0, ptn = tn, have_prev = true;
        v += G;
        if (v < p.tiles) tile_mn(v, m0, tn);
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// dense_planes_wres_kernel -- C[M][N] = A W^T + b with the WEIGHTS RESIDENT IN REGISTERS, for the LSTM2 projection (K = 256).
//
// The ablations say the tile kernels above pay for traffic: both operands of every chunk come from L2, cross the LDS write
// path and are read back as fragments.  With K = 256 a wave that owns 32 output columns can keep ALL its weights -- 32 columns
// x 256 k x two fp16 pieces = 32 KB = 128 registers per lane, in fragment order, loaded once per launch -- and the eight waves
// of a workgroup cover a 256-column tile.  What streams is the activation operand alone: 64 rows x 64 channels per chunk
// (16 KB through L2 and the LDS write path per 1 M products instead of 64 KB), and a k-step reads 4 fragments for 6 matrix
// instructions instead of 6.  A workgroup keeps its column tile for the whole launch and walks row tiles; the five workgroups
// that share a row tile sit on one XCD (the activation rows are fetched into that L2 once).  Same chunk order, same matrix
// instructions per output as dense_planes_pipe_kernel: bit-identical rows.
// Grid: 8 XCDs x lanes_per_xcd x tiles_n workgroups (240 on MI355X for N = 1280: 48 row-tile lanes x 5 column tiles; 1024
// windows = 528 row tiles = 11 per lane exactly).
constexpr int kWrBM = 64, kWrBN = 256, kWrK = 256, kWrKS = kWrK / 16;
constexpr int kWrStage = kWrBM * kPlRowB;  // 17 408 B: one 64-channel chunk of 64 activation rows

struct DenseWresParams {
    const void *a;      // plane activations [M][4 slabs][hi 64 | lo 64] fp16 (K = 256)
    const void *w;      // [N/256][8 waves][16 k-steps][2 pieces][64 lanes][16 B]: lane (column n = lane & 31 of the wave's 32, half kh = lane >> 5)
                        // holds k = 16 ks + 8 kh .. + 7 of piece 0 (hi) / 1 (lo) of W[256 tn + 32 wave + n][.], times 2^s
    const float *bias;  // [N]
    float *c;           // [M][N] fp32
    float post_scale;   // 2^-k: the projection weights are packed times ONE power of two (the kernel has no register to spare for a vector; LSTM
                        // weights carry no folded BatchNorm, so their rows do not differ by orders of magnitude the way convolution channels can)
    int M, N;
    int tiles_m, tiles_n, lanes_per_xcd;  // ceil(M / 64), N / 256, row-tile lanes per XCD
};

// ABL (tools/dense_probe.hip only; 0 in the product): 1 no activation loads, 2 no LDS staging writes, 4 no matrix instructions,
// 8 no fragment reads, 16 no result stores, 32 no barriers; 64 (product A/B knob C3HIP_GX2_NT): non-temporal result stores; 128 / 256
// (probe only) the results stored in two / three bytes per value.
template <int ABL = 0>
__global__ __launch_bounds__(kDnThreads, 2) void dense_planes_wres_kernel(DenseWresParams p) {
    __shared__ __attribute__((aligned(16))) char smem[4 * kWrStage + 8192];
    float *bias_lds = reinterpret_cast<float *>(smem + 4 * kWrStage);
    for (int i = threadIdx.x; i < p.N && i < 2048; i += kDnThreads) bias_lds[i] = p.bias[i];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 31, kh = lane >> 5;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int tn = slot % p.tiles_n;
    const int ml = xcd * p.lanes_per_xcd + slot / p.tiles_n, nl = 8 * p.lanes_per_xcd;  // this workgroup's row-tile lane, lanes in all
    if (ml >= p.tiles_m) return;
    constexpr int rowb = 1024;  // bytes per row of A

    const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.a), 0, (uint32_t)((int64_t)p.M * rowb), 0x00020000);
    const __amdgpu_buffer_rsrc_t crsrc = __builtin_amdgcn_make_buffer_rsrc(p.c, 0, (uint32_t)((int64_t)p.M * p.N * 4), 0x00020000);

    // the wave's weights: 16 k-steps x 2 pieces, 16 bytes per lane each
    pl_u32x4 wr[kWrKS][2];
    {
        const char *wb = reinterpret_cast<const char *>(p.w) + ((size_t)(tn * 8 + wave) * kWrKS * 2 * 64 + lane) * 16;
#pragma unroll
        for (int ks = 0; ks < kWrKS; ++ks)
#pragma unroll
            for (int pc = 0; pc < 2; ++pc) wr[ks][pc] = *reinterpret_cast<const pl_u32x4 *>(wb + (ks * 2 + pc) * 1024);
    }
    // piece j (0, 1) of a chunk: row (tid >> 4) + 32 j, 16-byte position tid & 15; lim = 0 turns the request into an out-of-range one
    pl_u32x4 ra[2];
    auto issue1 = [&](int j, int m0, int kc, uint32_t lim) __attribute__((always_inline)) {
        const int m = m0 + (tid >> 4) + 32 * j;
        const uint32_t off = (uint32_t)m < lim ? (uint32_t)m * (uint32_t)rowb + (uint32_t)(kc * 256 + (tid & 15) * 16) : kPlOob;
        if constexpr (ABL & 1) ra[j] = pl_u32x4{off, off ^ 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};  // keeps the address arithmetic alive
        else ra[j] = __builtin_bit_cast(pl_u32x4, __builtin_amdgcn_raw_buffer_load_b128(arsrc, off, 0, 0));
    };
    const int st_off = (tid >> 4) * kPlRowB + (tid & 15) * 16;
    auto stage1 = [&](int j, int buf) __attribute__((always_inline)) {
        char *dst = smem + buf * kWrStage + st_off + j * 32 * kPlRowB;
        if constexpr (ABL & 2) {
            if (ra[j][1] == 0x12345u && ra[j][2] == 0x54321u) *reinterpret_cast<pl_u32x4 *>(dst) = ra[j];  // never true: the registers stay live
        } else {
            *reinterpret_cast<pl_u32x4 *>(dst) = ra[j];
        }
    };
    auto mma = [](f32x16 c, pl_u32x4 w, pl_u32x4 x) __attribute__((always_inline)) {
        if constexpr (ABL & 4) {
            c[0] += __uint_as_float(w[0] ^ x[0]), c[5] += __uint_as_float(w[3] ^ x[3]);
            return c;
        } else {
            return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, x), c, 0, 0, 0);
        }
    };
    const int a_rd[2] = {frow * kPlRowB + kh * 16, (32 + frow) * kPlRowB + kh * 16};

    // TWO accumulator sets: a tile's results leave while the NEXT tile is multiplied into the other set -- one 16-byte store per
    // k-step, between two matrix instructions, where a store costs next to nothing to issue (alone in front of the tile the eight
    // stores of a wave queued up behind each other: "result stores" were 15 of the kernel's 67 us in tools/dense_probe.hip)
    f32x16 accA[2], accB[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) accA[i][e] = 0.f, accB[i][e] = 0.f;
    const int n0 = tn * kWrBN + wave * 32 + 4 * kh;
    // piece (i, q) of the finished tile at pm0: out-of-range offsets when there is none -- the same stores on every path
    auto store_piece = [&](const f32x16 (&acc)[2], int i, int q, int pm0, bool valid) __attribute__((always_inline)) {
        const int m = pm0 + i * 32 + frow;
        const uint32_t rowoff = valid && m < p.M ? (uint32_t)(((int64_t)m * p.N + n0) * 4) : kPlOob;
        const f32x4 bv = *reinterpret_cast<const f32x4 *>(bias_lds + n0 + 8 * q);
        f32x4 val = {acc[i][4 * q], acc[i][4 * q + 1], acc[i][4 * q + 2], acc[i][4 * q + 3]};
#pragma unroll
        for (int e = 0; e < 4; ++e) val[e] = __builtin_fmaf(val[e], p.post_scale, bv[e]);
        if constexpr (ABL & 16) {
            if (val[0] == 1234.5f) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pl_u32x4, val), crsrc, rowoff + 32 * q, 0, 0);
        } else if constexpr (ABL & 128) {
            // tools/dense_probe.hip `gx2` only (round 6): what the launch takes when the pre-activations leave in TWO
            // bytes per value -- the four values of a piece as fp16, one 8-byte store at half the offset.  Timing only: an upper bound of
            // what any narrower gx2 format can save in this kernel (a 3-byte format lies between this and the fp32 stores)
            const f16x2 a = __builtin_convertvector(f32x2{val[0], val[1]}, f16x2), b = __builtin_convertvector(f32x2{val[2], val[3]}, f16x2);
            const u32x2 hv = {__builtin_bit_cast(uint32_t, a), __builtin_bit_cast(uint32_t, b)};
            __builtin_amdgcn_raw_buffer_store_b64(hv, crsrc, rowoff == kPlOob ? kPlOob : (rowoff + 32 * q) / 2, 0, 0);
        } else if constexpr (ABL & 256) {
            // ... and in THREE bytes per value: 12 of the piece's 16 bytes, one 12-byte store at three quarters of the offset
            const pl_u32x4 w = __builtin_bit_cast(pl_u32x4, val);
            typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
            __builtin_amdgcn_raw_buffer_store_b96(u32x3{w[0], w[1], w[2]}, crsrc, rowoff == kPlOob ? kPlOob : (rowoff + 32 * q) / 4 * 3, 0, 0);
        } else if constexpr (ABL & 64) {  // A/B knob (C3HIP_GX2_NT=1): the 173 MB of pre-activations leave with the non-temporal hint
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pl_u32x4, val), crsrc, rowoff + 32 * q, 0, 2);
        } else {
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pl_u32x4, val), crsrc, rowoff + 32 * q, 0, 0);
        }
    };

    // Chunk pipeline (the one of conv3x3_planes_kernel): FOUR LDS stages, stage = chunk index inside the tile.  Chunk g is requested
    // in the middle of chunk g - 3, written to LDS in the middle of chunk g - 2 -- right behind that chunk's only barrier, a bare
    // s_barrier: every wave has left chunk g - 3, the stage's last reader -- and first read at the end of chunk g - 1, when its
    // first fragments are prefetched: the fragment reads run one k-step ahead of the matrix instructions across chunk and tile
    // boundaries, and no wave ever drains its LDS queue at a barrier.
    int mt = ml;
    int m0 = mt * kWrBM;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
#pragma unroll
        for (int j = 0; j < 2; ++j) issue1(j, m0, c, (uint32_t)p.M);
#pragma unroll
        for (int j = 0; j < 2; ++j) stage1(j, c);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) issue1(j, m0, 2, (uint32_t)p.M);
    lds_barrier();

    pl_u32x4 xh[2][2], xl[2][2];
    auto frags = [&](int stage, int ks, int st) __attribute__((always_inline)) {
        const char *cur = smem + stage * kWrStage;
        if constexpr (ABL & 8) {
            const pl_u32x4 f = {(uint32_t)(ks + stage), (uint32_t)lane, 0x3c003c00u, 0x3c003c00u};
            xh[st][0] = xh[st][1] = xl[st][0] = xl[st][1] = f;
            return;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            xh[st][i] = *reinterpret_cast<const pl_u32x4 *>(cur + a_rd[i] + ks * 32);
            xl[st][i] = *reinterpret_cast<const pl_u32x4 *>(cur + a_rd[i] + 128 + ks * 32);
        }
    };
    frags(0, 0, 0);
    int pm0 = 0;
    bool have_prev = false;
    bool more = true;
    // one tile into `acc`, the previous tile's results leaving from `prev`
    auto tile = [&](f32x16 (&acc)[2], const f32x16 (&prev)[2]) __attribute__((always_inline)) {
        const int mtn = mt + nl;
        const int m0n = mtn * kWrBM;
        const uint32_t limn = mtn < p.tiles_m ? (uint32_t)p.M : 0u;  // the next tile of this lane, if any
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int st = ks & 1, kk = c * 4 + ks;
                if (ks < 3) frags(c, ks + 1, st ^ 1);
                else frags((c + 1) & 3, 0, 0);  // (after the last tile: a stage of zeros)
                if (kk < 8) store_piece(prev, kk >> 2, kk & 3, pm0, have_prev);
                __builtin_amdgcn_sched_barrier(0);
                if (kk == 0) {  // the accumulators start from the constant 0 of the tile's first matrix instructions
                    f32x16 zero;
#pragma unroll
                    for (int e = 0; e < 16; ++e) zero[e] = 0.f;
                    acc[0] = mma(zero, wr[kk][0], xl[st][0]);
                    acc[1] = mma(zero, wr[kk][0], xl[st][1]);
                } else {
                    acc[0] = mma(acc[0], wr[kk][0], xl[st][0]);
                    acc[1] = mma(acc[1], wr[kk][0], xl[st][1]);
                }
                acc[0] = mma(acc[0], wr[kk][1], xh[st][0]);
                acc[1] = mma(acc[1], wr[kk][1], xh[st][1]);
                acc[0] = mma(acc[0], wr[kk][0], xh[st][0]);
                acc[1] = mma(acc[1], wr[kk][0], xh[st][1]);
                __builtin_amdgcn_sched_barrier(0);
                if (ks == 1) {
                    // the chunk's one barrier, bare: the reads in flight are this chunk's own prefetch, the wave's last LDS writes a chunk old
                    asm volatile("" ::: "memory");
                    if constexpr (!(ABL & 32)) __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int j = 0; j < 2; ++j) stage1(j, (c + 2) & 3);  // chunk c + 2 (c >= 2: chunk c - 2 of the next tile)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {  // ... and chunk c + 3 is requested into the registers just freed
                        if (c < 1) issue1(j, m0, c + 3, (uint32_t)p.M);
                        else issue1(j, m0n, c - 1, limn);
                    }
                }
            }
        }
        pm0 = m0, have_prev = true;
        mt = mtn, m0 = m0n;
        more = mt < p.tiles_m;
    };
    for (;;) {
        tile(accA, accB);
        if (!more) {
#pragma unroll
            for (int k = 0; k < 8; ++k) store_piece(accA, k >> 2, k & 3, pm0, true);
            break;
        }
        tile(accB, accA);
        if (!more) {
#pragma unroll
            for (int k = 0; k < 8; ++k) store_piece(accB, k >> 2, k & 3, pm0, true);
            break;
        }
    }
}

}  // namespace c3

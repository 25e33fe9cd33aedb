// c3_conv3w.h -- the stride-1 3x3 convolutions of Clair3_F's residual blocks (clair3/model.py:200-235) with FEWER matrix
// instructions: Winograd F(2,3) along the image's ROWS only, on the machinery of c3_conv3.h (plane activations, fp16x3 piece
// products on v_mfma_f32_32x32x16_f16, weights in fragment order straight into registers, two workgroups per CU).
//
// Why one dimension.  The direct kernels sit at the power-limited matrix rate: only fewer matrix instructions
// make them faster.  The two-dimensional F(2x2,3x3) needs 16 accumulators per 2x2 outputs (4 per output: a wave cannot hold a
// tile that amortises its weight stream, or the 16 products of a tile must meet through LDS) and a transform of 16 values per
// 4 inputs on the vector unit, whose rate is 1/16 of the matrix pipe's; round 1's kernel of that shape lost on exactly those
// two counts (424 MB of weight re-reads per launch, 9.7 vector instructions per matrix instruction).  F(2,3) along H keeps
// everything the direct kernel has -- the three column taps stay a row offset into one LDS tile, a wave owns 64 x 32 (tile-pixel,
// cout) blocks with the direct kernel's operand traffic per matrix instruction -- and needs 12 products per output PAIR instead
// of 18: 1.5x fewer matrix instructions, 2 accumulators per output, a transform of 4 values per 2 new inputs, and the output
// transform is register arithmetic of the lane that owns the accumulators (no exchange).  Rows pair up well: H = 45 / 23 / 12 give
// 23 / 12 / 6 pairs (2 / 4 / 0 % padding) where the columns (17 / 9 / 5) would give 6 / 11 / 20 %.
//
//   output rows (2j, 2j+1) of a window from input rows d0..d3 = 2j-1 .. 2j+2 (zero outside the window), per column and channel:
//     V0 = d0 - d2   V1 = d1 + d2   V2 = d2 - d1   V3 = d1 - d3                      (input transform, exact in fp32)
//     U0 = g0        U1 = (g0+g1+g2)/2   U2 = (g0-g1+g2)/2   U3 = g2                 (weights g[kh], per kw: packed once, c3_pack.h)
//     M_xi = sum over kw, ci of V_xi[w + kw - 1][ci] * U_xi[kw][ci][co]              (12 "taps" = 4 xi x 3 kw)
//     y[2j] = M0 + M1 + M2      y[2j+1] = M1 - M2 - M3
//
// Tile-pixels.  m' = (b * Hj + j) * W + w flattens (window, row pair, column); one workgroup (256 threads = 4 waves as 2 x 2) owns
// 126 consecutive m' x 64 output channels = 252 output pixels plus one halo tile-pixel on either side = 128 V rows (so that the
// transform is exactly two items per thread); a wave computes 64 rows x 32 couts x 4 xi = eight 32 x 32 accumulators (rows 0 and 127
// of a tile are computed on zeros and dropped).  The column taps of tile-pixel m' are m' - 1, m', m' + 1 (masked at the window's edge
// by a redirect to a zero row, as in c3_conv3.h).
// LDS holds V of ONE 32-channel slab: 4 xi planes x 129 rows x 144 B (hi 64 B | lo 64 B | 16 B pad: the ds_read_b128 of the 16 rows
// of a lane group covers all 64 banks once) = 75 KB -- two workgroups per CU -- and the slab's transform (8 buffer loads, ~100
// vector instructions, 8 ds_write_b128 per (row, 8-channel) item; two items per thread) runs between two barriers while the
// OTHER workgroup of the CU is in its tap loop.  Per slab and wave: 144 matrix instructions (direct: 2 x 108 for the same outputs).
#pragma once
#include "c3_conv3.h"

namespace c3 {

constexpr int kWTM = 126;                          // output tile-pixels per workgroup tile
constexpr int kWRows = 128;                        // V rows of a tile: its outputs and the column-tap halo row on either side
constexpr int kWRowB = 144;                        // LDS row stride of one xi plane (32 channels x 2 pieces x 2 B + 16)
constexpr int kWPlaneB = (kWRows + 1) * kWRowB;    // + the zero row (row kWRows) a masked column tap reads: 18 576 B
constexpr int kWLdsV = 4 * kWPlaneB;               // 74 304 B; the epilogue stages 256 output rows x 272 B = 69 632 B in it
static_assert(4 * kWRows == 2 * kPlThreads, "two (row, 8-channel group) transform items per thread");
static_assert(2 * kWRows * kPlRowB <= kWLdsV, "the staged output tile must fit the V planes");

struct WinoConvParams {
    const void *x;        // plane activations [M][C/64][2][64] fp16 (c3_conv3.h)
    const void *wf;       // U in fragment order: [Cout/64][C/32][12 taps = xi * 3 + kw][2 cout halves][2 k-steps][hi | lo][64 lanes] x 16 B
    const float *bias;    // [Cout]
    const float *post;    // [Cout] 2^-k of the packing (c3_pack.h row_scales)
    const void *res;      // residual, plane layout of the output (RES)
    void *out;            // plane activations [M][Cout/64][2][64]
    uint32_t *range_flag;
    int M;                // pixels = B * H * W
    int Mp;               // tile-pixels = B * Hj * W
    int H, W, Hj;         // Hj = ceil(H / 2) row pairs
    int tiles;            // ceil(Mp / 126) * (C / 64)
    uint32_t mg_hjw = 0, mg_w = 0;  // fast_div magics of Hj * W and W
    long long *trace = nullptr;     // ABL bit 32 (tools/wino_probe.hip): [2 workgroups][256] x {tag, shader clock}
};

// One mixed-precision instruction instead of two conversions and an add: (float)h16 + (float)l16 -- exact (the two fp16 pieces of an
// fp32 value) -- and v - (float)h16, the remainder the low piece is rounded from.  HALF picks the 16-bit half of the register.
template <int HALF> __device__ __forceinline__ float mix_add(uint32_t h, uint32_t l) {
    float r;
    if constexpr (HALF == 0) asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(h), "v"(l));
    else asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(h), "v"(l));
    return r;
}
template <int HALF> __device__ __forceinline__ float mix_rem(float v, uint32_t h) {
    float r;
    if constexpr (HALF == 0) asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(v));
    else asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(v));
    return r;
}
// split2_f16 (c3_gemm.h) with the remainder formed by mix_rem: the same pieces bit for bit
__device__ __forceinline__ void split2_f16_mix(const f32x4 x, u32x2 (&piece)[2]) {
#pragma unroll
    for (int i = 0; i < 4; i += 2) {
        const uint32_t h = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{x[i], x[i + 1]}, f16x2));
        const f32x2 r = {mix_rem<0>(x[i], h), mix_rem<1>(x[i + 1], h)};
        piece[0][i >> 1] = h;
        piece[1][i >> 1] = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, f16x2));
    }
}

// ABL (tools/wino_probe.hip only; 0 in the product): 1 no weight loads, 2 no transform after the first slab of the first tile,
// 4 no epilogue, 8 no matrix instructions, 32 shader-clock trace of workgroups 0 and 256 (wave 0) at phase boundaries, 16 the transform's loads NOT requested ahead (round-5 first cut: each item's loads issued
// and waited for at the slab switch), 64 the UPPER BOUND of a split-K launch (round 6): p.tiles is twice the real
// count, workgroups 2k and 2k + 1 run the same tile on half of its slabs each and both store their (partial) result -- the time of
// a split-K pair WITHOUT its exchange; 128 start / end stamps of every workgroup (device-wide counter).  MIX = false: the transform on plain conversions (the probe checks the two forms agree bit for bit).
//
// Where the transform's memory latency goes.  A thread owns two items (row r = (tid + 256 it) >> 2 of the 128 V rows, 8-channel group
// g = tid & 3) per slab.  The 16 buffer loads of BOTH items of the next slab go out right behind the slab's last matrix instruction,
// into registers the pixel fragments no longer need (requesting item 0 from the middle of the tap loop does not fit: 128 accumulator
// + 32 ring + 32 fragment + 32 item registers and the addresses spill): one memory latency per slab switch, under the other waves'
// tails and the OTHER workgroup's tap loop.  The loads of the next TILE's first slab go out as soon as the accumulators are staged:
// none at a tile switch -- the store loop of the epilogue runs in between.
template <int C, bool RES, int ABL = 0, bool MIX = true>
__global__ __launch_bounds__(kPlThreads, 2) void conv3x3_wino_planes_kernel(WinoConvParams p) {
    constexpr int NS = C / 64;     // output column tiles
    constexpr int NS32 = C / 32;   // input slabs
    constexpr int PIXB = 4 * C;    // bytes per pixel
    constexpr int NCH = 12 * NS32; // 8 KB weight chunks per tile
    constexpr int T = kWRows;      // index of the zero row of every plane
    constexpr bool AHEAD = !(ABL & 16);
    __shared__ __attribute__((aligned(16))) char smem[kWLdsV + 512 + 2 * kWRows * 8];
    char *const vlds = smem;
    float *const bias_lds = reinterpret_cast<float *>(smem + kWLdsV);
    float *const post_lds = bias_lds + 64;
    int2 *const rowinfo0 = reinterpret_cast<int2 *>(smem + kWLdsV + 512);  // two row tables: this tile's and the next one's

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int frow = lane & 31, kh = lane >> 5;
    const int W = p.W, H = p.H;
    const int G = gridDim.x;
    const uint32_t rowB = (uint32_t)W * (uint32_t)PIXB;  // bytes between two image rows
    // the zero rows of the four planes (the staged output tile of the epilogue runs over three of them: rewritten per tile)
    auto zero_rows = [&]() __attribute__((always_inline)) {
        uint32_t z = 0;
        int t = tid;
        asm volatile("" : "+v"(z), "+v"(t));  // (a plain constant or address is hoisted out of the tile loop and, with no register left for it, spilled)
        if (t < 4 * 8) *reinterpret_cast<pl_u32x4 *>(vlds + (t >> 3) * kWPlaneB + T * kWRowB + (t & 7) * 16) = pl_u32x4{z, z, z, z};
    };

    constexpr int SPLIT = (ABL & 64) ? 1 : 0;  // probe only: two workgroups per tile, half the slabs each
    constexpr int NSL = SPLIT ? NS32 / 2 : NS32;
    int v = blockIdx.x;
    int tile = xcd_tile_index(v, p.tiles) >> SPLIT;
    const int tn = tile % NS;
    int m0 = (tile / NS) * kWTM;

    const __amdgpu_buffer_rsrc_t xrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.x), 0, (uint32_t)((int64_t)p.M * PIXB), 0x00020000);
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (uint32_t)((int64_t)p.M * PIXB), 0x00020000);
    const __amdgpu_buffer_rsrc_t rrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(RES ? p.res : p.out), 0, (uint32_t)((int64_t)p.M * PIXB), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(p.wf)) + (size_t)tn * NCH * 8192, 0, (uint32_t)(NCH * 8192), 0x00020000);
    const uint32_t w_voff = (uint32_t)(wn * 4096 + lane * 16);

    // ---- the weight ring: chunk cc (= slab * 12 + tap) sits in slot cc & 1 = tap & 1 as wq[slot][k-step][hi | lo]; a k-step's
    // registers are refilled with chunk cc + 2 right behind the matrix instructions that read them
    pl_u32x4 wq[2][2][2];
    auto w_issue = [&](int slot, int ks, int cc) __attribute__((always_inline)) {
        if constexpr (ABL & 1) return;
        const uint32_t so = (uint32_t)(cc * 8192 + ks * 2048);
        wq[slot][ks][0] = __builtin_bit_cast(pl_u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, w_voff, so, 0));
        wq[slot][ks][1] = __builtin_bit_cast(pl_u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, w_voff, so + 1024, 0));
    };
    auto mma = [](f32x16 c, pl_u32x4 w, pl_u32x4 x) __attribute__((always_inline)) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, x), c, 0, 0, 0);
    };

    // ---- row table of a tile: V row r is tile-pixel mbase - 1 + r = (b, j, w); rows 1 .. 126 are the tile's outputs
    //   .x  byte offset of input pixel (b, 2j - 1, w)  (may lie in front of the tensor: only used together with its row bit)
    //   .y  bits 0-3 input rows 2j-1 .. 2j+2 inside the window; bits 4-6 column taps kw = 0..2 inside the window AND the tile; bit 7
    //       output row 2j + 1 exists; bit 8 the tile-pixel exists and is an output of this tile
    auto make_rowinfo = [&](int2 *tab, int mbase, bool on) __attribute__((always_inline)) {
        if (tid < kWRows) {
            const int mp = mbase - 1 + tid;
            int2 ri = make_int2(0, 0);
            if (on && (unsigned)mp < (unsigned)p.Mp) {
                const int hjw = p.Hj * W;
                const int b = fast_div(mp, p.mg_hjw), rem = mp - b * hjw;
                const int j = fast_div(rem, p.mg_w), w = rem - j * W;
                ri.x = (int)((uint32_t)((b * H + 2 * j - 1) * W + w) * (uint32_t)PIXB);
                int bits = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) bits |= (unsigned)(2 * j - 1 + k) < (unsigned)H ? 1 << k : 0;
                if (tid >= 1 && tid <= kWTM) {  // rows 0 and 127 are halo only: their own outputs belong to the neighbouring tiles
                    bits |= 0x100;
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) bits |= (unsigned)(w + kw - 1) < (unsigned)W ? 16 << kw : 0;
                    bits |= 2 * j + 1 < H ? 0x80 : 0;
                }
                ri.y = bits;
            }
            tab[tid] = ri;
        }
    };

    // ---- input transform of one 32-channel slab, in two halves: the loads of item `it` of slab s32 of the tile whose row table is
    // `tab`; and (later) V0..V3 of that item as fp16 pieces into the four planes
    auto titem_issue = [&](const int2 *tab, int s32, int it, pl_u32x4 (&rh)[4], pl_u32x4 (&rl)[4]) __attribute__((always_inline)) {
        const uint32_t soff = (uint32_t)((s32 >> 1) * 256 + (s32 & 1) * 64);
        int tid_ = tid;
        asm volatile("" : "+v"(tid_));
        const int idx = tid_ + kPlThreads * it;
        const int r = idx >> 2, g = idx & 3;
        const int2 ri = tab[r];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t off = ((ri.y >> k) & 1) ? (uint32_t)ri.x + (uint32_t)k * rowB + soff + (uint32_t)(g * 16) : kPlOob;
            rh[k] = __builtin_bit_cast(pl_u32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, off, 0, 0));
            rl[k] = __builtin_bit_cast(pl_u32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, off, 128, 0));
        }
    };
    auto titem_finish = [&](int it, const pl_u32x4 (&rh)[4], const pl_u32x4 (&rl)[4]) __attribute__((always_inline)) {
        int tid_ = tid;
        asm volatile("" : "+v"(tid_));
        const int idx = tid_ + kPlThreads * it;
        const int r = idx >> 2, g = idx & 3;
        char *dst = vlds + r * kWRowB + g * 16;
#pragma unroll
        for (int half = 0; half < 2; ++half) {  // four channels at a time (registers)
            f32x4 d[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if constexpr (MIX) {
                    d[k][0] = mix_add<0>(rh[k][2 * half], rl[k][2 * half]), d[k][1] = mix_add<1>(rh[k][2 * half], rl[k][2 * half]);
                    d[k][2] = mix_add<0>(rh[k][2 * half + 1], rl[k][2 * half + 1]), d[k][3] = mix_add<1>(rh[k][2 * half + 1], rl[k][2 * half + 1]);
                } else {
                    const f16x8 h8 = __builtin_bit_cast(f16x8, rh[k]), l8 = __builtin_bit_cast(f16x8, rl[k]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) d[k][e] = (float)h8[4 * half + e] + (float)l8[4 * half + e];
                }
            }
            const f32x4 vv[4] = {d[0] - d[2], d[1] + d[2], d[2] - d[1], d[1] - d[3]};
#pragma unroll
            for (int xi = 0; xi < 4; ++xi) {
                u32x2 pc[2];
                if constexpr (MIX) split2_f16_mix(vv[xi], pc);
                else split2_f16(vv[xi], pc);
                *reinterpret_cast<u32x2 *>(dst + xi * kWPlaneB + half * 8) = pc[0];
                *reinterpret_cast<u32x2 *>(dst + xi * kWPlaneB + 64 + half * 8) = pc[1];
            }
        }
    };

    int tr_n = 0;
    auto trace = [&](int tag) __attribute__((always_inline)) {
        if constexpr (ABL & 32) {
            if ((blockIdx.x == 0 || blockIdx.x == 256) && tid == 0 && tr_n < 250) {
                long long *tb = p.trace + ((blockIdx.x ? 1 : 0) * 256 + tr_n) * 2;
                tb[0] = tag, tb[1] = (long long)__builtin_readcyclecounter();
                ++tr_n;
            }
        }
    };
    trace(1);
    // ABL 128 (probe only, round 6): every workgroup stamps its start and its end with the device-wide 100 MHz counter -- when do the
    // tiles of a launch finish relative to each other, i.e. how early could a consumer tile of the NEXT layer start
    if constexpr (ABL & 128) {
        if (tid == 0) p.trace[2 * blockIdx.x] = (long long)__builtin_amdgcn_s_memrealtime();
    }
    const int lrow[2] = {wm * 64 + frow, wm * 64 + 32 + frow};  // this lane's V rows (tile-pixel m0 - 1 + lrow; its column taps: lrow + kw - 1)
    const int cb0 = wn * 32 + 4 * kh;                            // first of this lane's output channels inside the column tile
    float omax = 0.f;
    pl_u32x4 ta_h[4], ta_l[4], tb_h[4], tb_l[4];                 // the two transform items in flight

    // ---- prologue
    int cur = 0;
    make_rowinfo(rowinfo0, m0, true);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) w_issue(0, ks, 0);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) w_issue(1, ks, 1);
    if (tid < 64) bias_lds[tid] = p.bias[tn * 64 + tid], post_lds[tid] = p.post[tn * 64 + tid];
    zero_rows();
    lds_barrier();  // the row table is there
    titem_issue(rowinfo0, 0, 0, ta_h, ta_l);
    titem_issue(rowinfo0, 0, 1, tb_h, tb_l);
    titem_finish(0, ta_h, ta_l);
    titem_finish(1, tb_h, tb_l);
    trace(2);
    lds_barrier();
    trace(3);

    for (;;) {
        int2 *const rinfo = rowinfo0 + cur * kWRows, *const rnext = rowinfo0 + (cur ^ 1) * kWRows;
        // where this lane's two tile-pixels read their three column taps: V row lrow + kw - 1, or the zero row where the tap falls off
        // the window or the tile (plane and k-step are immediate offsets of the reads)
        const char *asrc[2][3];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const uint32_t mk = (uint32_t)rinfo[lrow[i]].y >> 4;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) asrc[i][kw] = vlds + (((mk >> kw) & 1u) ? lrow[i] + kw - 1 : T) * kWRowB + kh * 16;
        }
        const int vn = v + G;
        const bool more = vn < p.tiles;
        const int m0n = more ? ((xcd_tile_index(vn, p.tiles) >> SPLIT) / NS) * kWTM : 0;
        make_rowinfo(rnext, m0n, more);  // (read from the first slab switch on; a tile has at least two slabs)

        f32x16 acc[4][2];
#pragma unroll
        for (int xi = 0; xi < 4; ++xi)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[xi][i][e] = 0.f;

        // operand registers of the tile-pixels: two stages (k-step 0 / 1 of a tap)
        pl_u32x4 xh[2][2], xl[2][2];
        auto frags = [&](int tap, int ks, int st) __attribute__((always_inline)) {
            const int xi = tap / 3, kw = tap % 3;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const char *src = asrc[i][kw] + xi * kWPlaneB + ks * 32;
                xh[st][i] = *reinterpret_cast<const pl_u32x4 *>(src);
                xl[st][i] = *reinterpret_cast<const pl_u32x4 *>(src + 64);
            }
        };

#pragma unroll 1
        for (int s32 = 0; s32 < NSL; ++s32) {
            const bool last_slab = s32 + 1 == NSL;
            frags(0, 0, 0);
#pragma unroll
            for (int tap = 0; tap < 12; ++tap) {
                const int xi = tap / 3, slot = tap & 1;
                const int cc = s32 * 12 + tap;
                int ccn = cc + 2;  // the ring refills with the chunk two ahead of this workgroup's cyclic stream
                if (ccn >= NCH) ccn -= NCH;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    if (ks == 0) frags(tap, 1, 1);
                    else if (tap != 11) frags(tap + 1, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (!(ABL & 8)) {
                        acc[xi][0] = mma(acc[xi][0], wq[slot][ks][0], xl[ks][0]);
                        acc[xi][1] = mma(acc[xi][1], wq[slot][ks][0], xl[ks][1]);
                        acc[xi][0] = mma(acc[xi][0], wq[slot][ks][1], xh[ks][0]);
                        acc[xi][1] = mma(acc[xi][1], wq[slot][ks][1], xh[ks][1]);
                        acc[xi][0] = mma(acc[xi][0], wq[slot][ks][0], xh[ks][0]);
                        acc[xi][1] = mma(acc[xi][1], wq[slot][ks][0], xh[ks][1]);
                    } else {
                        acc[xi][0][ks] += __uint_as_float(wq[slot][ks][0][0] ^ xl[ks][0][1] ^ xh[ks][1][2] ^ wq[slot][ks][1][3]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    w_issue(slot, ks, ccn);
                }
            }
            // both items of the next slab's transform go out right behind the slab's last matrix instruction, into registers the pixel
            // fragments no longer need (the next TILE's: as soon as the accumulators are staged, below)
            trace(10 + s32);
            if (!last_slab) {  // slab switch inside the tile
                if constexpr (AHEAD && !(ABL & 2)) {
                    titem_issue(rinfo, s32 + 1, 0, ta_h, ta_l);
                    titem_issue(rinfo, s32 + 1, 1, tb_h, tb_l);
                }
                if constexpr (!(ABL & 2)) {
                    if constexpr (!AHEAD) {
                        lds_barrier();
                        titem_issue(rinfo, s32 + 1, 0, ta_h, ta_l);
                        titem_finish(0, ta_h, ta_l);
                        titem_issue(rinfo, s32 + 1, 1, tb_h, tb_l);
                        titem_finish(1, tb_h, tb_l);
                    } else {
                        trace(20);
                        lds_barrier();  // every wave has finished reading the old slab
                        trace(21);
                        titem_finish(0, ta_h, ta_l);
                        titem_finish(1, tb_h, tb_l);
                        trace(22);
                    }
                } else {
                    lds_barrier();
                }
                lds_barrier();
                trace(23);
            }
        }

        // ---- epilogue: output transform in registers, bias, the two output rows of every tile-pixel through LDS, then
        // (output pixel, 8-channel) items: residual, ReLU, split, two 16-byte stores
        lds_barrier();  // all waves are done with the V planes
        trace(30);
        if constexpr (ABL & 4) {
            float sacc = 0.f;
#pragma unroll
            for (int xi = 0; xi < 4; ++xi) sacc += acc[xi][0][0] + acc[xi][1][3];
            if (sacc == 12345.f) p.range_flag[1] = 1u;
            if constexpr (AHEAD && !(ABL & 2)) {
                titem_issue(rnext, 0, 0, ta_h, ta_l);
                titem_issue(rnext, 0, 1, tb_h, tb_l);
            }
        } else {
            // (output pixel, 8-channel) items of the store loop: idx = tid + 256 k, k = 0..7 -> staged row pr = idx >> 3 (output row of the
            // pair = pr >> 7, V row = pr & 127), channel group g = idx & 7.  Their residual pixels are requested EARLY -- the first four
            // items' before the accumulators are staged, the last four's right behind -- so that the store loop waits for nothing
            int tid_e = tid;
            asm volatile("" : "+v"(tid_e));  // (addresses of the eight items are recomputed per tile instead of living through the tap loop)
            uint32_t ioff[8];
            pl_u32x4 rh[8], rl[8];
            auto item_request = [&](int k) __attribute__((always_inline)) {
                const int idx = tid_e + kPlThreads * k;
                const int pr = idx >> 3, g = idx & 7;
                const int2 ri = rinfo[pr & (kWRows - 1)];
                const bool ok = (ri.y >> ((pr >> 7) ? 7 : 8)) & 1;
                ioff[k] = ok ? (uint32_t)ri.x + (uint32_t)(1 + (pr >> 7)) * rowB + (uint32_t)(tn * 256 + g * 16) : kPlOob;
                if constexpr (RES) {
                    rh[k] = __builtin_bit_cast(pl_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rrsrc, ioff[k], 0, 0));
                    rl[k] = __builtin_bit_cast(pl_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rrsrc, ioff[k], 128, 0));
                }
            };
#pragma unroll
            for (int k = 0; k < 4; ++k) item_request(k);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 bv = *reinterpret_cast<const f32x4 *>(bias_lds + cb0 + 8 * q);
                    const f32x4 sv = *reinterpret_cast<const f32x4 *>(post_lds + cb0 + 8 * q);
                    f32x4 y0, y1;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float a0 = acc[0][i][4 * q + e], a1 = acc[1][i][4 * q + e], a2 = acc[2][i][4 * q + e], a3 = acc[3][i][4 * q + e];
                        y0[e] = __builtin_fmaf((a0 + a1) + a2, sv[e], bv[e]);
                        y1[e] = __builtin_fmaf((a1 - a2) - a3, sv[e], bv[e]);
                    }
                    // (staged row = 128 * output row of the pair + V row: consecutive lanes 272 B apart, conflict-free 16-byte writes)
                    *reinterpret_cast<f32x4 *>(vlds + lrow[i] * kPlRowB + (cb0 + 8 * q) * 4) = y0;
                    *reinterpret_cast<f32x4 *>(vlds + (kWRows + lrow[i]) * kPlRowB + (cb0 + 8 * q) * 4) = y1;
                }
#pragma unroll
            for (int k = 4; k < 8; ++k) item_request(k);
            // the accumulators are staged: the transform loads of the next tile's first slab go out under the store loop
            if constexpr (AHEAD && !(ABL & 2)) {
                titem_issue(rnext, 0, 0, ta_h, ta_l);
                titem_issue(rnext, 0, 1, tb_h, tb_l);
            }
            trace(31);
            lds_barrier();
            trace(32);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int idx = tid_e + kPlThreads * k;
                const int pr = idx >> 3, g = idx & 7;
                f32x4 a = *reinterpret_cast<const f32x4 *>(vlds + pr * kPlRowB + g * 32);
                f32x4 b = *reinterpret_cast<const f32x4 *>(vlds + pr * kPlRowB + g * 32 + 16);
                if constexpr (RES) {
                    const f16x8 h8 = __builtin_bit_cast(f16x8, rh[k]), l8 = __builtin_bit_cast(f16x8, rl[k]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        a[e] += (float)h8[e] + (float)l8[e];
                        b[e] += (float)h8[4 + e] + (float)l8[4 + e];
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    a[e] = __int_as_float(max(__float_as_int(a[e]), 0));
                    b[e] = __int_as_float(max(__float_as_int(b[e]), 0));
                }
                if (ioff[k] != kPlOob)  // (rows 0 and 127 of the tile and rows beyond the tensor carry no output)
                    omax = fmaxf(omax, fmaxf(fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3])), fmaxf(fmaxf(b[0], b[1]), fmaxf(b[2], b[3]))));
                u32x2 pa[2], pb[2];
                split2_f16(a, pa);
                split2_f16(b, pb);
                const pl_u32x4 hi = {pa[0][0], pa[0][1], pb[0][0], pb[0][1]}, lo = {pa[1][0], pa[1][1], pb[1][0], pb[1][1]};
                __builtin_amdgcn_raw_buffer_store_b128(hi, orsrc, ioff[k], 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(lo, orsrc, ioff[k], 128, 0);
            }
        }
        trace(33);
        if (!more) break;
        lds_barrier();  // the staged tile has been read back
        trace(34);
        zero_rows();
        if constexpr (!(ABL & 2)) {
            if constexpr (!AHEAD) {
                titem_issue(rnext, 0, 0, ta_h, ta_l);
                titem_finish(0, ta_h, ta_l);
                titem_issue(rnext, 0, 1, tb_h, tb_l);
                titem_finish(1, tb_h, tb_l);
            } else {
                titem_finish(0, ta_h, ta_l);
                titem_finish(1, tb_h, tb_l);
            }
        }
        trace(35);
        lds_barrier();
        trace(36);
        v = vn, m0 = m0n, cur ^= 1;
    }
    if constexpr (ABL & 128) {
        __builtin_amdgcn_s_waitcnt(0);  // (the tile's stores have been issued and acknowledged)
        if (tid == 0) p.trace[2 * blockIdx.x + 1] = (long long)__builtin_amdgcn_s_memrealtime();
    }
    if (p.range_flag && !(omax < kF16Range)) atomicOr(p.range_flag, 1u);
}

}  // namespace c3

// c3_conv3w16.h -- the F(2,3)-along-H convolution of c3_conv3w.h with its INPUT ROWS ON AN LDS-DMA STREAM.
//
// What the phase trace of conv3x3_wino_planes_kernel says (profiles/r05_d_wino_probe_trace.txt): of a tile's ~70 k cycles per
// workgroup only 28 k are tap loops; every slab switch spends 1 - 1.5 k cycles REQUESTING the 16 loads of the transform (a
// vector-memory instruction outside a matrix phase costs 60 - 100 cycles of issue: the path is busy with the other workgroup's weight
// fragments) and 2.4 - 4 k waiting for them and transforming, and the epilogue requests 16 more plus the residual pixels.  The loads
// cannot leave earlier through registers (128 accumulators + ring + fragments: DESIGN.md 3.8).  Here they do not use registers:
//   * a slab is 16 channels (ONE k-step); the four input rows of every V row land in a 32 KB LDS buffer by
//     `buffer_load ... lds` (lane L's 16 bytes go to M0 + 16 L: 64 (row, 16-byte unit) slots per instruction), requested one per tap
//     from INSIDE the tap loop of the slab before -- between matrix instructions, where a request costs next to nothing -- and for
//     a tile's first slab from inside the last tap loop of the tile before;
//   * the slab switch is: s_waitcnt for the DMA (the eight weight loads behind it stay in flight), barrier, ONE transform item per
//     thread out of LDS (8 ds_read_b128, ~125 vector instructions, 8 LDS writes), barrier;
//   * the 16 units of a landed row (4 input rows x hi | lo x two 8-channel halves) sit at slot u ^ (row & 15): the DMA cannot pad
//     rows (256 B apart), the XOR keeps the transform's reads of 8 consecutive rows on distinct banks (c3_conv3s2.h uses the same trick);
//   * V of a 16-channel slab is 4 planes x 129 rows x 80 B = 41 KB; the epilogue stages the tile's output rows in it in TWO halves
//     (row 2j of every pair, then row 2j+1: 35 KB each), so the landing buffer -- which already holds the next tile's first slab --
//     stays untouched.  LDS: 32 + 41 + 2.5 = 75 KB, two workgroups per CU as before.
// Same arithmetic as c3_conv3w.h in another summation order over the channels (16 instead of 32 per slab changes nothing: the
// k-steps of a tap still accumulate in channel order); weights: [Cout/64][C/16][12 taps][cout half][hi | lo][lane] x 16 B (c3_pack.h).
#pragma once
#include "c3_conv3w.h"

namespace c3 {

constexpr int kW16RowB = 80;  // LDS row stride of one xi plane: 16 channels x 2 pieces x 2 B + 16
// RT = row tiles (32 V rows) per wave: 2 -> 128-row tiles (126 output pairs), 128 accumulator registers, two workgroups per CU;
// 1 -> 64-row tiles (62 output pairs), 64 accumulator registers, <= 128 registers a lane and 40 KB of LDS: FOUR workgroups per CU, so
// that one workgroup's slab switches and epilogue run under three others' tap loops (a weight fragment then feeds three matrix
// instructions instead of six: twice the vector-memory traffic per matrix instruction)
template <int RT> struct W16Geom {
    static constexpr int Rows = 64 * RT, TM = Rows - 2;
    static constexpr int PlaneB = (Rows + 1) * kW16RowB;  // (row Rows: the zero row)
    static constexpr int LdsV = 4 * PlaneB;                // RT 2: 41 280 B, RT 1: 20 800 B; the epilogue stages Rows x 272 B in it, twice
    static constexpr int Raw = Rows * 256;                 // Rows V rows x 16 units of 16 B (4 input rows x hi | lo x 2 halves)
    static constexpr int Smem = Raw + LdsV + 512 + 2 * Rows * 8 + 2 * Rows * 16;
    static_assert(Rows * kPlRowB <= LdsV, "half of the staged output tile must fit the V planes");
};

// ABL (tools/wino_probe.hip only; 0 in the product): 1 no weight loads, 2 no DMA / transform after the first slab of the first tile,
// 4 no epilogue, 8 no matrix instructions, 32 shader-clock trace (as c3_conv3w.h).
template <int C, bool RES, int ABL = 0, int RT = 2, int WD = 4>
__global__ __launch_bounds__(kPlThreads, RT == 1 ? 4 : 2) void conv3x3_wino16_planes_kernel(WinoConvParams p) {
    using Geo = W16Geom<RT>;
    constexpr int kWRows = Geo::Rows, kWTM = Geo::TM, kW16PlaneB = Geo::PlaneB, kW16LdsV = Geo::LdsV, kW16Raw = Geo::Raw;  // (shadow c3_conv3w.h's)
    constexpr int QW = 4 * RT;  // DMA requests per wave and slab
    static_assert(12 % WD == 0 && WD >= 2, "ring slots are named statically");
    constexpr int NS = C / 64;      // output column tiles
    constexpr int NS16 = C / 16;    // input slabs
    constexpr int PIXB = 4 * C;     // bytes per pixel
    constexpr int NCH = 12 * NS16;  // 4 KB weight chunks per tile
    constexpr int T = kWRows;       // index of the zero row of every plane
    __shared__ __attribute__((aligned(1024))) char smem[Geo::Smem];
    char *const raw = smem;                         // the DMA landing buffer (1 KB per instruction: 4 rows)
    char *const vlds = smem + kW16Raw;
    float *const bias_lds = reinterpret_cast<float *>(smem + kW16Raw + kW16LdsV);
    float *const post_lds = bias_lds + 64;
    int2 *const rowinfo0 = reinterpret_cast<int2 *>(smem + kW16Raw + kW16LdsV + 512);  // two row tables: this tile's and the next one's
    // ... and per V row the byte offsets of its four input rows, out-of-range (bit 31) where the row lies outside the window: what a DMA lane adds its unit's constant to
    uint32_t *const rowoff0 = reinterpret_cast<uint32_t *>(smem + kW16Raw + kW16LdsV + 512 + 2 * kWRows * 8);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int frow = lane & 31, kh = lane >> 5;
    const int W = p.W, H = p.H;
    const int G = gridDim.x;
    const uint32_t rowB = (uint32_t)W * (uint32_t)PIXB;  // bytes between two image rows

    int v = blockIdx.x;
    int tile = xcd_tile_index(v, p.tiles);
    const int tn = tile % NS;
    int m0 = (tile / NS) * kWTM;

    const __amdgpu_buffer_rsrc_t xrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.x), 0, (uint32_t)((int64_t)p.M * PIXB), 0x00020000);
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (uint32_t)((int64_t)p.M * PIXB), 0x00020000);
    const __amdgpu_buffer_rsrc_t rrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(RES ? p.res : p.out), 0, (uint32_t)((int64_t)p.M * PIXB), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(p.wf)) + (size_t)tn * NCH * 4096, 0, (uint32_t)(NCH * 4096), 0x00020000);
    const uint32_t w_voff = (uint32_t)(wn * 2048 + lane * 16);

    // ---- the weight ring: chunk cc (= slab * 12 + tap) sits in slot tap & 3 as (hi, lo); refilled with chunk cc + 4 right behind the
    // matrix instructions that read it
    pl_u32x4 wq[WD][2];
    auto w_issue = [&](int slot, int cc) __attribute__((always_inline)) {
        if constexpr (ABL & 1) return;
        const uint32_t so = (uint32_t)(cc * 4096);
        wq[slot][0] = __builtin_bit_cast(pl_u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, w_voff, so, 0));
        wq[slot][1] = __builtin_bit_cast(pl_u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, w_voff, so + 1024, 0));
    };
    auto mma = [](f32x16 c, pl_u32x4 w, pl_u32x4 x) __attribute__((always_inline)) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, x), c, 0, 0, 0);
    };

    // ---- row table of a tile (c3_conv3w.h): V row r is tile-pixel mbase - 1 + r = (b, j, w); rows 1 .. 126 are the tile's outputs
    auto make_rowinfo = [&](int2 *tab, int mbase, bool on) __attribute__((always_inline)) {
        if (tid < kWRows) {
            const int mp = mbase - 1 + tid;
            uint32_t *const offs = rowoff0 + (tab - rowinfo0) * 4 + tid * 4;
            int2 ri = make_int2(0, 0);
            if (on && (unsigned)mp < (unsigned)p.Mp) {
                const int hjw = p.Hj * W;
                const int b = fast_div(mp, p.mg_hjw), rem = mp - b * hjw;
                const int j = fast_div(rem, p.mg_w), w = rem - j * W;
                ri.x = (int)((uint32_t)((b * H + 2 * j - 1) * W + w) * (uint32_t)PIXB);
                int bits = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) bits |= (unsigned)(2 * j - 1 + k) < (unsigned)H ? 1 << k : 0;
                if (tid >= 1 && tid <= kWTM) {
                    bits |= 0x100;
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) bits |= (unsigned)(w + kw - 1) < (unsigned)W ? 16 << kw : 0;
                    bits |= 2 * j + 1 < H ? 0x80 : 0;
                }
                ri.y = bits;
            }
            tab[tid] = ri;
#pragma unroll
            for (int k = 0; k < 4; ++k) offs[k] = ((ri.y >> k) & 1) ? (uint32_t)ri.x + (uint32_t)k * rowB : 0x80000000u;
        }
    };

    // ---- DMA request q (0 .. 7) of this wave for slab s16 of the tile whose row table is `tab`: rows R0 = 4 (8 wave + q) .. + 3 of the
    // landing buffer; lane L is slot L & 15 of row R0 + (L >> 4) and fetches the unit that belongs there: u = slot ^ (row & 15),
    // u = 4 k + 2 piece + half (input row k of the V row, hi | lo, channels 8 half .. + 7 of the slab)
    typedef void __attribute__((address_space(3))) *lds_ptr;
    auto dma = [&](const int2 *tab, int s16, int q) __attribute__((always_inline)) {
        int lane_ = lane;
        asm volatile("" : "+v"(lane_));  // (per-request lane arithmetic is recomputed instead of living through the tap loop)
        const int r = 4 * (QW * wave + q) + (lane_ >> 4);
        const int u = (lane_ & 15) ^ ((4 * q + (lane_ >> 4)) & 15);  // (row & 15: 4 QW wave is a multiple of 16)
        const uint32_t base = rowoff0[(tab - rowinfo0) * 4 + r * 4 + (u >> 2)];  // bit 31 set: outside (stays out of range under the adds below)
        const uint32_t off = base + (uint32_t)((s16 >> 2) * 256 + (s16 & 3) * 32) + (uint32_t)(((u >> 1) & 1) * 128 + (u & 1) * 16);
        char *dst = raw + (QW * wave + q) * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lds_ptr)dst, 16, off, 0, 0, 0);
    };

    // ---- input transform of the landed slab: thread = (V row r, 8-channel half g) -> V0..V3 as fp16 pieces
    auto transform = [&]() __attribute__((always_inline)) {
        int tid_ = tid;
        asm volatile("" : "+v"(tid_));
        if constexpr (RT == 2) {
            const int r = (tid_ & 15) + 16 * (tid_ >> 5), g = (tid_ >> 4) & 1;  // a lane group of 16 = 16 consecutive rows, one half: its 16 slots differ
            const char *src = raw + r * 256;
            const int x = r & 15;
            f32x4 d[2][4];  // [4-channel half][input row]
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const pl_u32x4 h = *reinterpret_cast<const pl_u32x4 *>(src + (((4 * k + g) ^ x) << 4));
                const pl_u32x4 l = *reinterpret_cast<const pl_u32x4 *>(src + (((4 * k + 2 + g) ^ x) << 4));
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    d[half][k][0] = mix_add<0>(h[2 * half], l[2 * half]), d[half][k][1] = mix_add<1>(h[2 * half], l[2 * half]);
                    d[half][k][2] = mix_add<0>(h[2 * half + 1], l[2 * half + 1]), d[half][k][3] = mix_add<1>(h[2 * half + 1], l[2 * half + 1]);
                }
            }
            char *dst = vlds + r * kW16RowB + g * 16;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const f32x4 vv[4] = {d[half][0] - d[half][2], d[half][1] + d[half][2], d[half][2] - d[half][1], d[half][1] - d[half][3]};
#pragma unroll
                for (int xi = 0; xi < 4; ++xi) {
                    u32x2 pc[2];
                    split2_f16_mix(vv[xi], pc);
                    *reinterpret_cast<u32x2 *>(dst + xi * kW16PlaneB + half * 8) = pc[0];
                    *reinterpret_cast<u32x2 *>(dst + xi * kW16PlaneB + 32 + half * 8) = pc[1];
                }
            }
        } else {
            // 64 rows: thread = (row, 8-channel half g, 4-channel half): 8-byte reads; lanes 0..15 = 16 consecutive rows (16 distinct slots),
            // lanes 16..31 the other eight bytes of the same slots: a half wave covers every bank once
            const int r = (tid_ & 15) + 16 * (tid_ >> 6), half = (tid_ >> 4) & 1, g = (tid_ >> 5) & 1;
            const char *src = raw + r * 256 + half * 8;
            const int x = r & 15;
            f32x4 d[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const u32x2 h = *reinterpret_cast<const u32x2 *>(src + (((4 * k + g) ^ x) << 4));
                const u32x2 l = *reinterpret_cast<const u32x2 *>(src + (((4 * k + 2 + g) ^ x) << 4));
                d[k][0] = mix_add<0>(h[0], l[0]), d[k][1] = mix_add<1>(h[0], l[0]);
                d[k][2] = mix_add<0>(h[1], l[1]), d[k][3] = mix_add<1>(h[1], l[1]);
            }
            char *dst = vlds + r * kW16RowB + g * 16 + half * 8;
            const f32x4 vv[4] = {d[0] - d[2], d[1] + d[2], d[2] - d[1], d[1] - d[3]};
#pragma unroll
            for (int xi = 0; xi < 4; ++xi) {
                u32x2 pc[2];
                split2_f16_mix(vv[xi], pc);
                *reinterpret_cast<u32x2 *>(dst + xi * kW16PlaneB) = pc[0];
                *reinterpret_cast<u32x2 *>(dst + xi * kW16PlaneB + 32) = pc[1];
            }
        }
    };
    auto zero_rows = [&]() __attribute__((always_inline)) {
        uint32_t z = 0;
        int t = tid;
        asm volatile("" : "+v"(z), "+v"(t));
        if (t < 4 * 4) *reinterpret_cast<pl_u32x4 *>(vlds + (t >> 2) * kW16PlaneB + T * kW16RowB + (t & 3) * 16) = pl_u32x4{z, z, z, z};
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
    int lrow[RT];  // this lane's V rows
#pragma unroll
    for (int i = 0; i < RT; ++i) lrow[i] = wm * 32 * RT + 32 * i + frow;
    const int cb0 = wn * 32 + 4 * kh;                            // first of this lane's output channels inside the column tile
    float omax = 0.f;

    // ---- prologue
    int cur = 0;
    make_rowinfo(rowinfo0, m0, true);
#pragma unroll
    for (int t = 0; t < WD; ++t) w_issue(t, t);
    if (tid < 64) bias_lds[tid] = p.bias[tn * 64 + tid], post_lds[tid] = p.post[tn * 64 + tid];
    zero_rows();
    lds_barrier();  // the row table is there
#pragma unroll
    for (int q = 0; q < QW; ++q) dma(rowinfo0, 0, q);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    transform();
    trace(2);
    lds_barrier();
    trace(3);

    for (;;) {
        int2 *const rinfo = rowinfo0 + cur * kWRows, *const rnext = rowinfo0 + (cur ^ 1) * kWRows;
        const char *asrc[RT][3];
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            const uint32_t mk = (uint32_t)rinfo[lrow[i]].y >> 4;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) asrc[i][kw] = vlds + (((mk >> kw) & 1u) ? lrow[i] + kw - 1 : T) * kW16RowB + kh * 16;
        }
        const int vn = v + G;
        const bool more = vn < p.tiles;
        const int m0n = more ? (xcd_tile_index(vn, p.tiles) / NS) * kWTM : 0;
        make_rowinfo(rnext, m0n, more);  // (read from the first slab switch on; a tile has at least four slabs)

        f32x16 acc[4][RT];
#pragma unroll
        for (int xi = 0; xi < 4; ++xi)
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[xi][i][e] = 0.f;

        pl_u32x4 xh[2][RT], xl[2][RT];  // operand registers of the tile-pixels: two stages (even / odd taps)
        auto frags = [&](int tap, int st) __attribute__((always_inline)) {
            const int xi = tap / 3, kw = tap % 3;
#pragma unroll
            for (int i = 0; i < RT; ++i) {
                const char *src = asrc[i][kw] + xi * kW16PlaneB;
                xh[st][i] = *reinterpret_cast<const pl_u32x4 *>(src);
                xl[st][i] = *reinterpret_cast<const pl_u32x4 *>(src + 32);
            }
        };

#pragma unroll 1
        for (int s16 = 0; s16 < NS16; ++s16) {
            const bool last_slab = s16 + 1 == NS16;
            const int2 *const ntab = last_slab ? rnext : rinfo;  // whose rows the next transform works on
            const int ns16 = last_slab ? 0 : s16 + 1;
            frags(0, 0);
#pragma unroll
            for (int tap = 0; tap < 12; ++tap) {
                const int xi = tap / 3, slot = tap % WD, st = tap & 1;
                const int cc = s16 * 12 + tap;
                int ccn = cc + WD;  // the ring refills with the chunk WD ahead of this workgroup's cyclic stream
                if (ccn >= NCH) ccn -= NCH;
                if (tap != 11) frags(tap + 1, st ^ 1);
                // the next slab's input rows: the eight DMA requests at the HEAD of the slab (four in front of tap 0, four in front of tap 1).
                // Loads retire through one in-order counter: the wait for a weight fragment requested behind a DMA request waits for that
                // request too, so spread one per tap (round 5's first cut) every tap from the fifth on stood behind one; at the head only tap
                // 4 / 5's fragments do, four taps after the requests left
                if constexpr (!(ABL & 2))
                    if (tap < RT) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) dma(ntab, ns16, 4 * tap + q);
                    }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (!(ABL & 8)) {
#pragma unroll
                    for (int i = 0; i < RT; ++i) acc[xi][i] = mma(acc[xi][i], wq[slot][0], xl[st][i]);
#pragma unroll
                    for (int i = 0; i < RT; ++i) acc[xi][i] = mma(acc[xi][i], wq[slot][1], xh[st][i]);
#pragma unroll
                    for (int i = 0; i < RT; ++i) acc[xi][i] = mma(acc[xi][i], wq[slot][0], xh[st][i]);
                } else {
                    acc[xi][0][tap & 3] += __uint_as_float(wq[slot][0][0] ^ xl[st][0][1] ^ xh[st][RT - 1][2] ^ wq[slot][1][3]);
                }
                __builtin_amdgcn_sched_barrier(0);
                // 64-row tiles: the ring does not run ahead into the next tile (its registers are the epilogue's: residual pixels)
                if (RT == 2 || tap + WD < 12 || !last_slab) w_issue(slot, ccn);
            }
            trace(10 + (s16 & 7));
            // the DMA requests are older than the weight loads of taps 2 .. 11: with the eight of taps 8 .. 11 still in flight they have landed
            if constexpr (ABL & 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            lds_barrier();  // every wave has finished reading the old slab; every wave's part of the next one is in the landing buffer
            trace(21);
            if (!last_slab) {
                if constexpr (!(ABL & 2)) transform();
                trace(22);
                lds_barrier();
                trace(23);
            }
        }

        // ---- epilogue: output transform in registers, bias; the tile's output rows through LDS in two halves (row 2j of every pair,
        // then row 2j + 1), each: (output pixel, 8-channel) items -- residual, ReLU, split, two 16-byte stores.  The landing buffer
        // (the next tile's first slab) is not touched.
        trace(30);
        if constexpr (ABL & 4) {
            float sacc = 0.f;
#pragma unroll
            for (int xi = 0; xi < 4; ++xi) sacc += acc[xi][0][0] + acc[xi][RT - 1][3];
            if (sacc == 12345.f) p.range_flag[1] = 1u;
        } else {
            int tid_e = tid;
            asm volatile("" : "+v"(tid_e));
            constexpr int IP = 2 * RT;  // items per thread and half
            uint32_t ioff[2 * IP];
            pl_u32x4 rh[2 * IP], rl[2 * IP];
            auto item_request = [&](int k) __attribute__((always_inline)) {  // item k: half k / IP, staged row (tid + 256 (k % IP)) >> 3
                const int idx = tid_e + kPlThreads * (k % IP);
                const int pr = idx >> 3, g = idx & 7;
                const int2 ri = rinfo[pr];
                const bool ok = (ri.y >> ((k / IP) ? 7 : 8)) & 1;
                ioff[k] = ok ? (uint32_t)ri.x + (uint32_t)(1 + k / IP) * rowB + (uint32_t)(tn * 256 + g * 16) : kPlOob;
                if constexpr (RES) {
                    rh[k] = __builtin_bit_cast(pl_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rrsrc, ioff[k], 0, 0));
                    rl[k] = __builtin_bit_cast(pl_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rrsrc, ioff[k], 128, 0));
                }
            };
#pragma unroll
            for (int k = 0; k < IP; ++k) item_request(k);
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) {
#pragma unroll
                for (int i = 0; i < RT; ++i)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 bv = *reinterpret_cast<const f32x4 *>(bias_lds + cb0 + 8 * q);
                        const f32x4 sv = *reinterpret_cast<const f32x4 *>(post_lds + cb0 + 8 * q);
                        f32x4 y;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float a0 = acc[0][i][4 * q + e], a1 = acc[1][i][4 * q + e], a2 = acc[2][i][4 * q + e], a3 = acc[3][i][4 * q + e];
                            y[e] = ph == 0 ? __builtin_fmaf((a0 + a1) + a2, sv[e], bv[e]) : __builtin_fmaf((a1 - a2) - a3, sv[e], bv[e]);
                        }
                        *reinterpret_cast<f32x4 *>(vlds + lrow[i] * kPlRowB + (cb0 + 8 * q) * 4) = y;
                    }
                if (ph == 0) {
#pragma unroll
                    for (int k = IP; k < 2 * IP; ++k) item_request(k);
                }
                trace(31);
                lds_barrier();
                trace(32);
#pragma unroll
                for (int kk = 0; kk < IP; ++kk) {
                    const int k = IP * ph + kk;
                    const int idx = tid_e + kPlThreads * kk;
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
                    if (ioff[k] != kPlOob)
                        omax = fmaxf(omax, fmaxf(fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3])), fmaxf(fmaxf(b[0], b[1]), fmaxf(b[2], b[3]))));
                    u32x2 pa[2], pb[2];
                    split2_f16(a, pa);
                    split2_f16(b, pb);
                    const pl_u32x4 hi = {pa[0][0], pa[0][1], pb[0][0], pb[0][1]}, lo = {pa[1][0], pa[1][1], pb[1][0], pb[1][1]};
                    __builtin_amdgcn_raw_buffer_store_b128(hi, orsrc, ioff[k], 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(lo, orsrc, ioff[k], 128, 0);
                }
                trace(33);
                lds_barrier();  // the staged half has been read back
            }
        }
        if (!more) break;
        if constexpr (RT == 1) {
#pragma unroll
            for (int t = 0; t < WD; ++t) w_issue(t, t);
        }
        if constexpr (ABL & 4) lds_barrier();
        zero_rows();
        if constexpr (!(ABL & 2)) transform();  // the next tile's first slab landed during this tile's last tap loop
        trace(35);
        lds_barrier();
        trace(36);
        v = vn, m0 = m0n, cur ^= 1;
    }
    if (p.range_flag && !(omax < kF16Range)) atomicOr(p.range_flag, 1u);
}

}  // namespace c3

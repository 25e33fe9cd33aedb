// c3_conv3s2.h -- the two stride-2 3x3 convolutions of Clair3_F (conv3: 64 -> 128 channels, conv5: 128 -> 256; clair3/model.py:
// 317-342 BasicConv2D with stride 2, :183-197) as implicit GEMMs on plane activations (c3_conv3.h).
//
// A stride-2 tile cannot keep its input halo in LDS the way the stride-1 layers do (128 outputs of conv3 read ~520 input pixels:
// 141 KB per 64-channel slab), so the pixel operand is gathered per (tap, slab) chunk: output row m of the tile needs the 256 bytes
// of input pixel (2 oh - 1 + kh, 2 ow - 1 + kw), slab s.  Until round 4 these layers ran on dense_planes_glds_kernel (c3_dense.h):
// BOTH operands through LDS-DMA into two 64 KB stages -- bound by the L2 -> LDS stream, which is as long as the tile's matrix work
// and shares the LDS with the fragment reads.  Same workgroup here -- 512 threads = 8 waves as 2 (pixels) x 4 (couts), 128 output
// pixels x 128 output channels, a wave owns 64 x 32 outputs = two 32 x 32 accumulators, one workgroup per CU -- but, as in c3_conv3.h,
//  * only the PIXEL operand goes through LDS: `buffer_load_dwordx4 ... lds` (no staging registers, no ds_write) into two stages of
//    128 rows x 256 B; lane L's 16 bytes land at M0 + 16 L whatever its source address (tools/glds_probe.hip), so rows are 256 B
//    apart and piece q of row r sits at slot q ^ (r & 15) -- the involution applied to the source address by the loading lane
//    and to the read address by the reading lane; a lane whose tap falls off the window asks an out-of-range offset and gets
//    zeros (the padding);
//  * the WEIGHTS never touch LDS: packed in fragment order (c3_pack.h: [N/64][chunk][cout half][k-step][piece][lane] x 16 B), one
//    contiguous kilobyte per piece and k-step, fetched straight into registers one chunk ahead -- half the DMA volume, half the LDS
//    write traffic and two thirds of the fragment reads of the form before; the two waves of a cout block (wm = 0, 1) read the same
//    kilobyte (the second read is an L1 hit);
//  * one barrier per chunk: the DMA pieces of the next chunk have landed for every wave (s_waitcnt vmcnt(6): the six weight loads
//    issued behind the last DMA request may stay in flight), then s_barrier;
//  * the finished tile crosses LDS in two halves of 64 channels (through the stage its last chunk occupied) into 16-byte plane
//    stores: bias, ReLU, split.
// Chunk order kc = tap * (Cin / 64) + slab, products per accumulator lo x hi, hi x lo, hi x hi: the same sums as before (rows bit-identical
// to the LDS-DMA form's).
// Measured (tools/dense_probe.hip, profiles/r04_m_*; conv3 / conv5): B = 256: both operands through LDS-DMA 32.1 / 29.5 us, this kernel
// 27.3 (two workgroups per CU) / 27.5 us (one); B = 1000: 128 / 108 -> 101.5 / 99.7 us.  In the step (bench.py, same box): 35.4 / 32.5 ->
// 30.7 / 31.5 us.  By ablation at B = 256 the matrix instructions alone (with epilogue and launch) take 21 / 20 us, the pixel requests
// alone 18 / 16 us, the weight loads alone 17.5 / 16 us: a 128 x 128 tile still moves 64 KB out of L2 per 6.3 MFLOP chunk (96 FLOP per
// byte) -- what changed is that half of it no longer crosses the LDS.
#pragma once
#include "c3_conv3.h"

namespace c3 {

constexpr int kS2BM = 128, kS2BN = 128, kS2Threads = 512;
constexpr int kS2Row = 256, kS2Stage = kS2BM * kS2Row;  // 32 768 B per stage (one bit of an LDS address: stages toggle by XOR)

struct S2ConvParams {
    const void *a;      // input plane activations [B][Hin][Win][Cin/64][hi 64 | lo 64] fp16
    const void *wf;     // [N/64][NK chunks][2 cout halves][4 k-steps][hi | lo][64 lanes] x 16 B, chunk kc = tap * (Cin/64) + slab; times 2^k per cout
    const float *bias;  // [N]
    const float *post;  // [N] 2^-k
    void *c;            // output plane activations [M][N/64][hi | lo]
    uint32_t *range_flag;
    int M, N, NK;        // output pixels, output channels, 9 * Cin / 64
    int tiles_n, tiles;  // N / 128, ceil(M / 128) * tiles_n
    int Hin, Win, Cin, Ho, Wo;
    uint32_t mg_hw, mg_w;  // fast_div magics of Ho * Wo and Wo
};

// ABL (tools/dense_probe.hip only; 0 in the product): 1 no DMA requests inside the chunk loop, 2 no weight loads, 4 no matrix
// instructions, 8 no fragment reads, 16 no epilogue
// PAIR: two workgroups per CU (128 registers a lane: ONE set of fragment registers, the reads of k-step ks + 1 issued behind the matrix
// instructions of k-step ks and landing under the other waves' matrix work) -- for a layer with more tiles than CUs (conv3: 414),
// whose second round would leave 40 % of the chip idle and whose epilogues then hide under the neighbour's matrix work; otherwise
// one workgroup per CU with two sets of fragment registers (conv5: 240 tiles)
template <int ABL = 0, bool PAIR = false>
__global__ __launch_bounds__(kS2Threads, PAIR ? 4 : 2) void conv3x3_s2_planes_kernel(S2ConvParams p) {
    __shared__ __attribute__((aligned(1024))) char smem[2 * kS2Stage + 1024];
    float *bias_lds = reinterpret_cast<float *>(smem + 2 * kS2Stage);
    float *post_lds = bias_lds + 128;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;  // 2 x 4 waves, 64 x 32 outputs each
    const int frow = lane & 31, kh = lane >> 5;
    const int NK = p.NK, G = gridDim.x;
    const int rowb = p.Cin * 4;  // bytes per input pixel
    const int nsin = p.Cin / 64;

    int v = blockIdx.x;
    if (v >= p.tiles) return;
    const int tile0 = xcd_tile_index(v, p.tiles);
    const int tn = tile0 % p.tiles_n;  // the grid is a multiple of 8 tiles_n (or the tile count): every tile of this workgroup has this tn
    int m0 = (tile0 / p.tiles_n) * kS2BM;

    const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void *>(p.a), 0, (uint32_t)((int64_t)(p.M / (p.Ho * p.Wo)) * p.Hin * p.Win * rowb), 0x00020000);
    const __amdgpu_buffer_rsrc_t crsrc = __builtin_amdgcn_make_buffer_rsrc(p.c, 0, (uint32_t)((int64_t)p.M * p.N * 4), 0x00020000);
    // this wave's weight stream: half (wn & 1) of the 64-channel column tile 2 tn + (wn >> 1), NK chunks of 16 KB
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(p.wf)) + (size_t)(tn * 2 + (wn >> 1)) * NK * 16384, 0, (uint32_t)(NK * 16384), 0x00020000);
    const uint32_t w_voff = (uint32_t)((wn & 1) * 8192 + lane * 16);

    // DMA geometry: instruction j of this wave fills rows 16 wave + 4 j .. + 3 of a stage (1 KiB); lane L is slot L & 15 of row
    // 16 wave + 4 j + (L >> 4) and fetches the piece that belongs there: (L & 15) ^ (row & 15)
    const int drow = lane >> 4;
    int rbase[4];
    uint32_t rmask[4];
    auto row_info = [&](int mt) __attribute__((always_inline)) {
        const int hw = p.Ho * p.Wo;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = mt + 16 * wave + 4 * j + drow;
            uint32_t mk = 0;
            int base = 0;
            if (m < p.M) {
                const int b = fast_div(m, p.mg_hw), rem = m - b * hw;
                const int oh = fast_div(rem, p.mg_w), ow = rem - oh * p.Wo;
                const int ih0 = oh * 2 - 1, iw0 = ow * 2 - 1;
                base = ((b * p.Hin + ih0) * p.Win + iw0) * rowb;
                mk = tap_mask9(ih0, iw0, p.Hin, p.Win);
            }
            rbase[j] = base, rmask[j] = mk;
        }
    };
    typedef void __attribute__((address_space(3))) *lds_ptr;
    auto dma1 = [&](int j, int kc, bool on, int stage) __attribute__((always_inline)) {
        const int r16 = 4 * j + drow;  // row & 15 (16 wave is a multiple of 16)
        const uint32_t piece = (uint32_t)((lane & 15) ^ r16) * 16u;
        const int tap = kc / nsin, slab = kc - tap * nsin;
        const int kh3 = tap / 3, kw3 = tap - 3 * kh3;
        const uint32_t aoff = (on && ((rmask[j] >> tap) & 1u)) ? (uint32_t)(rbase[j] + (kh3 * p.Win + kw3) * rowb + slab * 256) + piece : kPlOob;
        char *dst = smem + stage * kS2Stage + (16 * wave + 4 * j) * kS2Row;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(arsrc, (lds_ptr)dst, 16, aoff, 0, 0, 0);
    };
    pl_u32x4 wq[4][2];
    auto w_issue = [&](int ks, int cc) __attribute__((always_inline)) {
        if constexpr (ABL & 2) return;
        const uint32_t so = (uint32_t)(cc * 16384 + ks * 2048);
        wq[ks][0] = __builtin_bit_cast(pl_u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, w_voff, so, 0));
        wq[ks][1] = __builtin_bit_cast(pl_u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, w_voff, so + 1024, 0));
    };
    auto mma = [](f32x16 c, pl_u32x4 w, pl_u32x4 x) __attribute__((always_inline)) {
        if constexpr (ABL & 4) {
            c[0] += __uint_as_float(w[0] ^ x[0]), c[5] += __uint_as_float(w[3] ^ x[3]);
            return c;
        } else {
            return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, x), c, 0, 0, 0);
        }
    };
    // fragment addresses: logical piece q (hi: 2 ks + kh, lo: 8 + 2 ks + kh) of row R sits at slot q ^ (R & 15); R & 15 = frow & 15 for
    // both rows this lane reads (R = 64 wm + 32 i + frow).  q ^ x = (kh ^ x) ^ (8 h + 2 ks): ONE register holds the address of piece
    // kh (stage 0; the stage bit is flipped in place at chunk ends), the others are that XOR a constant below 256
    uint32_t va0 = (uint32_t)((wm * 64 + frow) * kS2Row) + (uint32_t)((kh ^ (frow & 15)) * 16);
    // ---- prologue: the tile's rows, chunk 0 -> stage 0, the first chunk of the weight ring, this column tile's bias / scale
    row_info(m0);
#pragma unroll
    for (int j = 0; j < 4; ++j) dma1(j, 0, true, 0);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) w_issue(ks, 0);
    if (tid < 128) bias_lds[tid] = p.bias[tn * 128 + tid], post_lds[tid] = p.post[tn * 128 + tid];
    int vq = v, kq = 0;  // (tile, chunk) requested last
    auto advance = [&]() __attribute__((always_inline)) {  // -> the chunk after (vq, kq); returns whether it exists
        if (++kq == NK) {
            kq = 0, vq += G;
            if (vq < p.tiles) row_info((xcd_tile_index(vq, p.tiles) / p.tiles_n) * kS2BM);
        }
        return vq < p.tiles;
    };
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    int g = 0;  // chunks done: the current chunk sits in stage g & 1
    float omax = 0.f;
    f32x16 acc[2];
    auto chunk = [&](int kc, bool first) __attribute__((always_inline)) {
        const bool req = advance();  // the next chunk is requested during this one, into the other stage
        const int nstage = (g + 1) & 1;
        const int ccn = kc + 1 == NK ? 0 : kc + 1;  // the weight stream of this wave is cyclic: the same column tile for every tile
        constexpr int NF = PAIR ? 1 : 2;
        pl_u32x4 xh[NF][2], xl[NF][2];
        auto frags = [&](int ks, int st) __attribute__((always_inline)) {
            if constexpr (ABL & 8) {
#pragma unroll
                for (int i = 0; i < 2; ++i) xh[st][i] = pl_u32x4{va0, (uint32_t)(i + ks), 0u, 0u}, xl[st][i] = pl_u32x4{va0, (uint32_t)i, 1u, 0u};
                return;
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                xh[st][i] = *reinterpret_cast<const pl_u32x4 *>(smem + (va0 ^ (uint32_t)(32 * ks)) + i * 32 * kS2Row);
                xl[st][i] = *reinterpret_cast<const pl_u32x4 *>(smem + (va0 ^ (uint32_t)(128 + 32 * ks)) + i * 32 * kS2Row);
            }
        };
        frags(0, 0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int st = PAIR ? 0 : ks & 1;
            if constexpr (!PAIR)
                if (ks < 3) frags(ks + 1, st ^ 1);
            // the next chunk's four requests in the first two k-steps: two k-steps of matrix work for them to land in
            if constexpr (!(ABL & 1))
                if (ks < 2) dma1(2 * ks, kq, req, nstage), dma1(2 * ks + 1, kq, req, nstage);
            __builtin_amdgcn_sched_barrier(0);
            if (first && ks == 0) {
                f32x16 zero;
#pragma unroll
                for (int e = 0; e < 16; ++e) zero[e] = 0.f;
                acc[0] = mma(zero, wq[ks][0], xl[st][0]);
                acc[1] = mma(zero, wq[ks][0], xl[st][1]);
            } else {
                acc[0] = mma(acc[0], wq[ks][0], xl[st][0]);
                acc[1] = mma(acc[1], wq[ks][0], xl[st][1]);
            }
            acc[0] = mma(acc[0], wq[ks][1], xh[st][0]);
            acc[1] = mma(acc[1], wq[ks][1], xh[st][1]);
            acc[0] = mma(acc[0], wq[ks][0], xh[st][0]);
            acc[1] = mma(acc[1], wq[ks][0], xh[st][1]);
            __builtin_amdgcn_sched_barrier(0);
            w_issue(ks, ccn);
            if constexpr (PAIR)
                if (ks < 3) frags(ks + 1, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // this wave's reads of the stage are done and its DMA pieces of the next chunk have landed (everything up to the last
        // request; the six weight loads issued behind it -- k-steps 1, 2, 3 -- may stay in flight); behind the barrier everybody's have
        if constexpr (ABL & 2) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        va0 ^= (uint32_t)kS2Stage;
        ++g;
    };
    // the finished tile leaves through the stage its last chunk occupied ((g - 1) & 1: its readers are behind the barrier, and the
    // chunk in flight lands in the other stage), 64 channels at a time: 128 rows x 256 B of fp32, the 16-byte unit u of row r at u ^ (r & 15)
    auto epilogue = [&](int pm0) __attribute__((always_inline)) {
        // the lane's epilogue indices are derived afresh from a laundered lane id: hoisted out of the tile loop they would sit in
        // registers through the chunk loop (PAIR has none to spare: they spilled, and a kernel with scratch pays for it at every launch)
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int frow = tid & 31, kh = (tid >> 5) & 1;
        const int cb0 = (wn & 1) * 32 + 4 * kh;  // first of this lane's channels inside its 64-channel half
        char *stg = smem + ((g - 1) & 1) * kS2Stage;
#pragma unroll
        for (int half = 0; half < 2; ++half) {  // channels 64 half .. + 63 of the tile: the waves of cout blocks 2 half, 2 half + 1 stage them
            if ((wn >> 1) == half) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 bv = *reinterpret_cast<const f32x4 *>(bias_lds + 64 * half + cb0 + 8 * q);
                        const f32x4 sv = *reinterpret_cast<const f32x4 *>(post_lds + 64 * half + cb0 + 8 * q);
                        f32x4 val = {acc[i][4 * q], acc[i][4 * q + 1], acc[i][4 * q + 2], acc[i][4 * q + 3]};
#pragma unroll
                        for (int e = 0; e < 4; ++e) val[e] = __builtin_fmaf(val[e], sv[e], bv[e]);
                        const int r = wm * 64 + i * 32 + frow, u = (cb0 + 8 * q) >> 2;
                        *reinterpret_cast<f32x4 *>(stg + r * kS2Row + ((u ^ (frow & 15)) << 4)) = val;
                    }
            }
            lds_barrier();
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int idx = tid + kS2Threads * j;
                const int r = idx >> 3, c8 = idx & 7;  // row of the tile, group of 8 channels of this half
                const int m = pm0 + r;
                f32x4 a = *reinterpret_cast<const f32x4 *>(stg + r * kS2Row + (((2 * c8) ^ (r & 15)) << 4));
                f32x4 b = *reinterpret_cast<const f32x4 *>(stg + r * kS2Row + (((2 * c8 + 1) ^ (r & 15)) << 4));
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    a[e] = __int_as_float(max(__float_as_int(a[e]), 0));  // ReLU on the bit pattern
                    b[e] = __int_as_float(max(__float_as_int(b[e]), 0));
                }
                omax = fmaxf(omax, fmaxf(fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3])), fmaxf(fmaxf(b[0], b[1]), fmaxf(b[2], b[3]))));
                u32x2 pa[2], pb[2];
                split2_f16(a, pa);
                split2_f16(b, pb);
                const pl_u32x4 hi = {pa[0][0], pa[0][1], pb[0][0], pb[0][1]}, lo = {pa[1][0], pa[1][1], pb[1][0], pb[1][1]};
                const uint32_t off = m < p.M ? (uint32_t)m * (uint32_t)(p.N * 4) + (uint32_t)((tn * 2 + half) * 256 + c8 * 16) : kPlOob;
                __builtin_amdgcn_raw_buffer_store_b128(hi, crsrc, off, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(lo, crsrc, off + 128, 0, 0);
            }
            lds_barrier();  // the staged half has been read: the next half / the DMA of the chunk after next may overwrite the stage
        }
    };
    for (;;) {
        chunk(0, true);
        for (int kc = 1; kc < NK; ++kc) chunk(kc, false);
        if constexpr (ABL & 16) {
            if (acc[0][0] == 12345.f && acc[1][3] == 1.f) omax = 1e30f;  // keep the accumulators alive
        } else {
            epilogue(m0);
        }
        v += G;
        if (v >= p.tiles) break;
        m0 = (xcd_tile_index(v, p.tiles) / p.tiles_n) * kS2BM;
    }
    if (p.range_flag && !(omax < kF16Range)) atomicOr(p.range_flag, 1u);  // also taken for NaN
}

}  // namespace c3

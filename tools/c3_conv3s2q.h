// c3_conv3s2q.h -- (probe only, tools/dense_probe.hip; NOT in the library) the stride-2 convolution tile of c3_conv3s2.h on four waves.
// Measured (profiles/r05_j_s2_four_waves_probe.txt): rows bit-identical; B = 256 conv3 29.8 - 31.4 us against 28.0 (two eight-wave
// workgroups per CU), conv5 33.5 - 34.3 against 29.5 - 34.7; B = 1000 conv3 97.5 - 105.6 against 108.7, conv5 101 - 108 against 107.
// Halving the weight loads of a workgroup does cut the loop (no epilogue: 17.6 / 21.3 us against 24.7 / 27.4), but a thread now
// stages and stores twice as many output rows and at B = 256 every workgroup's ONE epilogue is exposed (414 / 240 tiles on 512
// slots: the whole chip stores 27 MB at the same moment).  Not kept.
#pragma once
#include "c3_conv3s2.h"

namespace c3 {

// ---- the same tile on FOUR waves (256 threads) as 1 (pixels) x 4 (couts): a wave owns 128 x 32 outputs = four 32 x 32 accumulators.
// The form above is bound by the vector-memory path (per chunk and wave 4 LDS-DMA requests + 8 weight loads per 24
// matrix instructions; the two waves of a cout block fetch the same weight kilobytes, and an L1 hit still crosses the path).  Here
// every weight fragment is fetched once per workgroup and feeds twelve matrix instructions instead of six: per chunk and workgroup
// 32 requests + 32 weight loads instead of 32 + 64.  The fragment reads per matrix instruction are unchanged (each 32-row block of
// pixels is read by the four cout waves, as before), LDS 66 KB: two workgroups per CU.  Same chunk order, same three products per
// accumulator in the same order: rows bit-identical to the eight-wave form's.
constexpr int kS2QThreads = 256;
template <int ABL = 0>
__global__ __launch_bounds__(kS2QThreads, 2) void conv3x3_s2q_planes_kernel(S2ConvParams p) {
    __shared__ __attribute__((aligned(1024))) char smem[2 * kS2Stage + 1024];
    float *bias_lds = reinterpret_cast<float *>(smem + 2 * kS2Stage);
    float *post_lds = bias_lds + 128;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);  // this wave's 32 couts
    const int frow = lane & 31, kh = lane >> 5;
    const int NK = p.NK, G = gridDim.x;
    const int rowb = p.Cin * 4;  // bytes per input pixel
    const int nsin = p.Cin / 64;

    int v = blockIdx.x;
    if (v >= p.tiles) return;
    const int tile0 = xcd_tile_index(v, p.tiles);
    const int tn = tile0 % p.tiles_n;
    int m0 = (tile0 / p.tiles_n) * kS2BM;

    const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void *>(p.a), 0, (uint32_t)((int64_t)(p.M / (p.Ho * p.Wo)) * p.Hin * p.Win * rowb), 0x00020000);
    const __amdgpu_buffer_rsrc_t crsrc = __builtin_amdgcn_make_buffer_rsrc(p.c, 0, (uint32_t)((int64_t)p.M * p.N * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(p.wf)) + (size_t)(tn * 2 + (wn >> 1)) * NK * 16384, 0, (uint32_t)(NK * 16384), 0x00020000);
    const uint32_t w_voff = (uint32_t)((wn & 1) * 8192 + lane * 16);

    // DMA geometry: instruction j (0 .. 7) of this wave fills rows 32 wave + 4 j .. + 3 of a stage; lane L is slot L & 15 of row
    // 32 wave + 4 j + (L >> 4) and fetches the piece that belongs there: (L & 15) ^ (row & 15)
    const int drow = lane >> 4;
    int rbase[8];
    uint32_t rmask[8];
    auto row_info = [&](int mt) __attribute__((always_inline)) {
        const int hw = p.Ho * p.Wo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int m = mt + 32 * wn + 4 * j + drow;
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
        const int r16 = (4 * j + drow) & 15;  // row & 15 (32 wave is a multiple of 16)
        const uint32_t piece = (uint32_t)((lane & 15) ^ r16) * 16u;
        const int tap = kc / nsin, slab = kc - tap * nsin;
        const int kh3 = tap / 3, kw3 = tap - 3 * kh3;
        const uint32_t aoff = (on && ((rmask[j] >> tap) & 1u)) ? (uint32_t)(rbase[j] + (kh3 * p.Win + kw3) * rowb + slab * 256) + piece : kPlOob;
        char *dst = smem + stage * kS2Stage + (32 * wn + 4 * j) * kS2Row;
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
    // fragment address of piece kh of row frow in stage 0 (the other rows: + 32 i rows; the other pieces: XOR a constant, as above)
    uint32_t va0 = (uint32_t)(frow * kS2Row) + (uint32_t)((kh ^ (frow & 15)) * 16);
    row_info(m0);
#pragma unroll
    for (int j = 0; j < 8; ++j) dma1(j, 0, true, 0);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) w_issue(ks, 0);
    if (tid < 128) bias_lds[tid] = p.bias[tn * 128 + tid], post_lds[tid] = p.post[tn * 128 + tid];
    int vq = v, kq = 0;  // (tile, chunk) requested last
    auto advance = [&]() __attribute__((always_inline)) {
        if (++kq == NK) {
            kq = 0, vq += G;
            if (vq < p.tiles) row_info((xcd_tile_index(vq, p.tiles) / p.tiles_n) * kS2BM);
        }
        return vq < p.tiles;
    };
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    int g = 0;
    float omax = 0.f;
    f32x16 acc[4];
    auto chunk = [&](int kc, bool first) __attribute__((always_inline)) {
        const bool req = advance();
        const int nstage = (g + 1) & 1;
        const int ccn = kc + 1 == NK ? 0 : kc + 1;
        pl_u32x4 xh[2][4], xl[2][4];
        auto frags = [&](int ks, int st) __attribute__((always_inline)) {
            if constexpr (ABL & 8) {
#pragma unroll
                for (int i = 0; i < 4; ++i) xh[st][i] = pl_u32x4{va0, (uint32_t)(i + ks), 0u, 0u}, xl[st][i] = pl_u32x4{va0, (uint32_t)i, 1u, 0u};
                return;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                xh[st][i] = *reinterpret_cast<const pl_u32x4 *>(smem + (va0 ^ (uint32_t)(32 * ks)) + i * 32 * kS2Row);
                xl[st][i] = *reinterpret_cast<const pl_u32x4 *>(smem + (va0 ^ (uint32_t)(128 + 32 * ks)) + i * 32 * kS2Row);
            }
        };
        frags(0, 0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int st = ks & 1;
            if (ks < 3) frags(ks + 1, st ^ 1);
            // the next chunk's eight requests in the first two k-steps
            if constexpr (!(ABL & 1))
                if (ks < 2) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) dma1(4 * ks + j, kq, req, nstage);
                }
            __builtin_amdgcn_sched_barrier(0);
            if (first && ks == 0) {
                f32x16 zero;
#pragma unroll
                for (int e = 0; e < 16; ++e) zero[e] = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = mma(zero, wq[ks][0], xl[st][i]);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = mma(acc[i], wq[ks][0], xl[st][i]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = mma(acc[i], wq[ks][1], xh[st][i]);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = mma(acc[i], wq[ks][0], xh[st][i]);
            __builtin_amdgcn_sched_barrier(0);
            w_issue(ks, ccn);
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (ABL & 2) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        va0 ^= (uint32_t)kS2Stage;
        ++g;
    };
    auto epilogue = [&](int pm0) __attribute__((always_inline)) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int frow = tid & 31, kh = (tid >> 5) & 1;
        const int cb0 = (wn & 1) * 32 + 4 * kh;
        char *stg = smem + ((g - 1) & 1) * kS2Stage;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if ((wn >> 1) == half) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 bv = *reinterpret_cast<const f32x4 *>(bias_lds + 64 * half + cb0 + 8 * q);
                        const f32x4 sv = *reinterpret_cast<const f32x4 *>(post_lds + 64 * half + cb0 + 8 * q);
                        f32x4 val = {acc[i][4 * q], acc[i][4 * q + 1], acc[i][4 * q + 2], acc[i][4 * q + 3]};
#pragma unroll
                        for (int e = 0; e < 4; ++e) val[e] = __builtin_fmaf(val[e], sv[e], bv[e]);
                        const int r = i * 32 + frow, u = (cb0 + 8 * q) >> 2;
                        *reinterpret_cast<f32x4 *>(stg + r * kS2Row + ((u ^ (frow & 15)) << 4)) = val;
                    }
            }
            lds_barrier();
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int idx = tid + kS2QThreads * j;
                const int r = idx >> 3, c8 = idx & 7;
                const int m = pm0 + r;
                f32x4 a = *reinterpret_cast<const f32x4 *>(stg + r * kS2Row + (((2 * c8) ^ (r & 15)) << 4));
                f32x4 b = *reinterpret_cast<const f32x4 *>(stg + r * kS2Row + (((2 * c8 + 1) ^ (r & 15)) << 4));
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    a[e] = __int_as_float(max(__float_as_int(a[e]), 0));
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
            lds_barrier();
        }
    };
    for (;;) {
        chunk(0, true);
        for (int kc = 1; kc < NK; ++kc) chunk(kc, false);
        if constexpr (ABL & 16) {
            if (acc[0][0] == 12345.f && acc[3][3] == 1.f) omax = 1e30f;
        } else {
            epilogue(m0);
        }
        v += G;
        if (v >= p.tiles) break;
        m0 = (xcd_tile_index(v, p.tiles) / p.tiles_n) * kS2BM;
    }
    if (p.range_flag && !(omax < kF16Range)) atomicOr(p.range_flag, 1u);
}

}  // namespace c3

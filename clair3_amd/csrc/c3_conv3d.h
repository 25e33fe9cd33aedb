// c3_conv3d.h -- the six stride-1 3x3 convolutions of Clair3_F's residual blocks (clair3/model.py:200-235) as TWO workgroups
// per CU ("duo"): the direct convolution of c3_conv3.h (plane activations, halo tile in LDS, weights as first matrix operand,
// fp16x3 products on v_mfma_f32_32x32x16_f16) re-cut so that a tile's non-matrix phases -- halo load, slab switch, epilogue --
// run under the matrix phase of the OTHER workgroup of the CU instead of leaving the matrix pipe idle.
//
// What the phase trace of the one-workgroup form showed (tools/conv_probe.hip, profiles/r04_a_conv_probe.txt; B = 256):
// a 256-pixel tile of res1 takes 28.5 k cycles of which 13.8 k are matrix instructions -- 8.4 k are the tail of the tile
// (wait for the slowest wave, tile through LDS, stores, next halo in), and inside the tap loop a chunk of 1536 matrix cycles
// takes 2100 (one workgroup-wide barrier per weight chunk: eight waves in lock-step).  One 512-thread workgroup with 132 KB
// of LDS owns the CU, so nothing runs under those gaps.
//
// The re-cut:
//  * workgroup = 256 threads = 4 waves as 2 (pixels) x 2 (couts), tile = 128 consecutive output pixels x 64 output channels;
//    a wave's work is what it was (64 x 32 outputs = two 32 x 32 accumulators);
//  * LDS = the halo tile only: (128 + 2 W + 2) pixel rows of one 64-channel slab (+ conv1's fragments in the first block):
//    <= 45 KB (70 KB with conv1) -- two workgroups per CU (80 KB each), one wave of each on every SIMD;
//  * WEIGHTS NEVER TOUCH LDS: they are packed in fragment order (c3_pack.h: [tn][slab][tap][wn][k-step][piece][lane] x 16 B),
//    so a wave's operand of one k-step is ONE contiguous 1 KB buffer load per piece, fetched straight into registers four
//    k-steps (= one chunk, >= 1500 cycles) ahead.  The tap loop therefore has NO barrier at all: the four waves of a workgroup
//    drift freely, the only workgroup-wide synchronisation left is the slab switch and the epilogue.  The two waves that
//    share a cout half (wm = 0, 1) read the same kilobyte -- the second read is an L1 hit;
//  * the two workgroups of a CU start half a tile apart (the later half of the grid sleeps first), so one is in its tap
//    loop while the other is in its tile tail.
// Everything else is c3_conv3.h's: tap masks, the zero row, conv1 computed in here for the first residual block (SRC8),
// the pyramid pooling as the last convolution's epilogue (SPPF; two 12 x 5 windows per tile), per-channel powers of two.
// Products per accumulator are issued in the same order as there (slab, tap, k-step; lo x hi, hi x lo, hi x hi): the rows are
// bit-identical to the one-workgroup form's.
#pragma once
#include "c3_conv3.h"

namespace c3 {

constexpr int kDuBM = 128, kDuThreads = 256;
constexpr int kDuHaloRows = kDuBM + 2 * kPlMaxW + 2;                              // 164
constexpr int kDuHaloBytes = (kDuHaloRows + 1) * kPlRowB;                         // + the zero row: 44 880 B
constexpr int kDuHaloLoads = (kDuHaloRows * 16 + kDuThreads - 1) / kDuThreads;    // 16-byte pieces per thread: 11

template <int C, bool RES, int ABL = 0, int SRC8 = 0, bool SPPF = false, int C1 = 8>
__global__ __launch_bounds__(kDuThreads, 2) void conv3x3_duo_kernel(PlaneConvParams p) {
    static_assert(C1 == 8 || C1 == 9, "conv1 inside this kernel: 8- or 9-channel windows");
    constexpr int NT1 = C1 == 8 ? 5 : 6;  // conv1 k-steps of 16
    static_assert(SRC8 == 0 || C == 64, "conv1 feeds the 64-channel block only");
    static_assert(SRC8 != 2 || RES, "SRC8 = 2 replaces the residual read");
    constexpr int NS = C / 64;   // input slabs = output column tiles
    constexpr int PIXB = 4 * C;  // bytes per pixel
    constexpr int NCH = 9 * NS;  // weight chunks per tile
    constexpr int kC1WBytes = NT1 * 2 * 2 * 64 * 16;
    __shared__ __attribute__((aligned(16))) char smem[kDuHaloBytes + 768 + (SRC8 ? kC1WBytes + 512 : 0)];
    char *const halo = smem;
    float *const bias_lds = reinterpret_cast<float *>(smem + kDuHaloBytes);
    float *const post_lds = bias_lds + 64, *const pre_lds = bias_lds + 128;
    char *const c1w_lds = smem + kDuHaloBytes + 768;
    float *const c1b_lds = reinterpret_cast<float *>(c1w_lds + kC1WBytes);
    float *const c1post_lds = c1b_lds + 64;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int frow = lane & 31, kh = lane >> 5;
    const int W = p.W, HW = p.H * p.W;
    const int T = kDuBM + 2 * W + 2;  // halo rows in use; row T is the zero row
    const int G = gridDim.x;

    // PERSISTENT: workgroup w walks the tiles of virtual blocks w, w + G, ... (XCD-aware order); G is a multiple of 8 NS (or
    // the tile count), so a workgroup keeps its column tile tn -- and with it its weight stream -- for every tile it takes.
    const int tile_stride = SPPF ? (kDuBM / HW) * HW : kDuBM;  // SPPF: whole windows per tile (2 x 60 pixels)
    int v = blockIdx.x;
    int tile = xcd_tile_index(v, p.tiles);
    const int tn = tile % NS;
    int m0 = (tile / NS) * tile_stride;

    const __amdgpu_buffer_rsrc_t xrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.x), 0, (uint32_t)((int64_t)p.M * PIXB), 0x00020000);
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (uint32_t)((int64_t)p.M * PIXB), 0x00020000);
    const __amdgpu_buffer_rsrc_t rrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(RES ? p.res : p.out), 0, (uint32_t)((int64_t)p.M * PIXB), 0x00020000);
    // this workgroup's weight stream: NCH chunks of 16 KB in fragment order
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(p.wf)) + (size_t)tn * NCH * 16384, 0, (uint32_t)(NCH * 16384), 0x00020000);
    const uint32_t w_voff = (uint32_t)(wn * 8192 + lane * 16);

    auto halo_issue = [&](pl_u32x4 (&h)[kDuHaloLoads], int mbase, int slab, bool on = true) __attribute__((always_inline)) {
        const int m_lo = mbase - W - 1;
        const uint32_t lim = on ? (uint32_t)p.M : 0u;
        int tid_ = tid;
        asm volatile("" : "+v"(tid_));
#pragma unroll
        for (int j = 0; j < kDuHaloLoads; ++j) {
            const int idx = tid_ + kDuThreads * j;
            const int row = idx >> 4, pos = idx & 15;
            const int pix = m_lo + row;
            const bool ok = row < T && (unsigned)pix < lim;
            const uint32_t off = ok ? (uint32_t)pix * (uint32_t)PIXB + (uint32_t)(slab * 256 + pos * 16) : kPlOob;
            h[j] = __builtin_bit_cast(pl_u32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, off, 0, 0));
        }
    };
    auto halo_write = [&](const pl_u32x4 (&h)[kDuHaloLoads]) __attribute__((always_inline)) {
        int tid_ = tid;
        asm volatile("" : "+v"(tid_));
#pragma unroll
        for (int j = 0; j < kDuHaloLoads; ++j) {
            const int idx = tid_ + kDuThreads * j;
            const int row = idx >> 4, pos = idx & 15;
            if (row < T) *reinterpret_cast<pl_u32x4 *>(halo + row * kPlRowB + pos * 16) = h[j];
        }
    };
    // the weight ring: k-step ks of the chunk in flight sits in wq[ks] (hi piece, lo piece); it is refilled with the same
    // k-step of the NEXT chunk right behind the matrix instructions that read it
    pl_u32x4 wq[4][2];
    auto w_issue = [&](int ks, int cc) __attribute__((always_inline)) {
        if constexpr (ABL & 1) return;
        const uint32_t so = (uint32_t)(cc * 16384 + ks * 2048);
        wq[ks][0] = __builtin_bit_cast(pl_u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, w_voff, so, 0));
        wq[ks][1] = __builtin_bit_cast(pl_u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, w_voff, so + 1024, 0));
    };
    auto mma = [](f32x16 c, pl_u32x4 w, pl_u32x4 x) __attribute__((always_inline)) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, x), c, 0, 0, 0);
    };

    const int lrow[2] = {wm * 64 + frow, wm * 64 + 32 + frow};
    const int cb0 = wn * 32 + 4 * kh;  // first of this lane's output channels inside the 64-channel slab tn

    // ---- conv1 inside this kernel (SRC8): see c3_conv3.h -- the same arithmetic, lane for lane
    typedef uint32_t c1u2 __attribute__((ext_vector_type(2)));
    const int c1_rowB = p.Win * C1;
    const __amdgpu_buffer_rsrc_t x8rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(p.x8)) - (SRC8 ? c1_rowB + C1 : 0), 0,
        SRC8 ? (uint32_t)((int64_t)(p.M / (p.H * p.W)) * p.Hin * c1_rowB + c1_rowB + C1) : 0u, 0x00020000);
    auto c1_request = [&](int pix, c1u2 (&d)[NT1]) __attribute__((always_inline)) {
        int kh_ = kh;
        asm volatile("" : "+v"(kh_));
        const bool valid = (unsigned)pix < (unsigned)p.M;
        const int b = fast_div(pix, p.mg_hw), r = pix - b * HW;
        const int oy = fast_div(r, p.mg_w), ox = r - oy * W;
        const uint32_t base = (uint32_t)(((b * p.Hin + 2 * oy) * p.Win + 2 * ox) * C1);
        if constexpr (C1 == 9) {
            const bool left_out = ox == 0, right_out = 2 * ox + 1 >= p.Win;
#pragma unroll
            for (int t = 0; t < 6; ++t) {
                const int ky = t >> 1, u = t & 1;
                const int iy = 2 * oy - 1 + ky;
                bool ok = valid && (unsigned)iy < (unsigned)p.Hin;
                uint32_t off = base + (uint32_t)(ky * c1_rowB + 16 * u) + (uint32_t)(8 * kh_);
                int shl = 0, shr = 0;
                if (u == 0) {
                    if (left_out) {
                        if (kh_) off += 1, shl = 8;
                        else ok = false;
                    }
                } else if (kh_) {
                    if (right_out) ok = false;
                    else off -= 5, shr = 40;
                } else {
                    if (right_out) off -= 6, shr = 48;
                }
                const c1u2 raw = __builtin_bit_cast(c1u2, __builtin_amdgcn_raw_buffer_load_b64(x8rsrc, ok ? off : 0x80000000u, 0, 0));
                uint64_t vv = (uint64_t)raw[0] | ((uint64_t)raw[1] << 32);
                vv = (vv << shl) >> shr;
                d[t] = c1u2{(uint32_t)vv, (uint32_t)(vv >> 32)};
            }
            return;
        }
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            const int tap = 2 * t + kh_;
            const int ky = tap / 3, kx = tap - 3 * ky;
            const int iy = 2 * oy - 1 + ky, ix = 2 * ox - 1 + kx;
            const bool ok = valid && tap < 9 && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
            d[t] = __builtin_bit_cast(c1u2, __builtin_amdgcn_raw_buffer_load_b64(x8rsrc, ok ? base + (uint32_t)(ky * c1_rowB + kx * 8) : 0x80000000u, 0, 0));
        }
    };
    auto c1_widen = [&](c1u2 d) __attribute__((always_inline)) {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        pl_u32x4 o;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t x = d[h] ^ 0x80808080u;
            const uint32_t p01 = __builtin_amdgcn_perm(0x48484848u, x, 0x04010400u);
            const uint32_t p23 = __builtin_amdgcn_perm(0x48484848u, x, 0x04030402u);
            const h2 nine = {(_Float16)-9.0f, (_Float16)-9.0f};
            o[2 * h] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(h2, p01) + nine);
            o[2 * h + 1] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(h2, p23) + nine);
        }
        return o;
    };
    auto c1_block = [&](const c1u2 (&d)[NT1], int cb) __attribute__((always_inline)) {
        f32x16 c;
#pragma unroll
        for (int e = 0; e < 16; ++e) c[e] = 0.f;
#pragma unroll
        for (int t = 0; t < NT1; ++t) {
            const pl_u32x4 a = c1_widen(d[t]);
            const pl_u32x4 w1 = *reinterpret_cast<const pl_u32x4 *>(c1w_lds + (((t * 2 + cb) * 2 + 1) * 64 + lane) * 16);
            const pl_u32x4 w0 = *reinterpret_cast<const pl_u32x4 *>(c1w_lds + (((t * 2 + cb) * 2 + 0) * 64 + lane) * 16);
            c = mma(c, w1, a);
            c = mma(c, w0, a);
        }
        return c;
    };
    float omax = 0.f;
    // SRC8 = 1: halo rows 32 g .. 32 g + 31 (pixel m_lo + row) for g = wave and wave + 4 (six groups cover the 165 rows)
    c1u2 c1d[2][NT1];
    auto c1_halo_request = [&](int mbase) __attribute__((always_inline)) {
        const int m_lo = mbase - W - 1;
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) c1_request(32 * (wave + 4 * gi) < T ? m_lo + 32 * (wave + 4 * gi) + frow : -0x40000000, c1d[gi]);
    };
    auto c1_halo_write = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
            const int row = 32 * (wave + 4 * gi) + frow;
            if (32 * (wave + 4 * gi) >= T) continue;  // wave-uniform
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                const f32x16 c = c1_block(c1d[gi], cb);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 b4 = *reinterpret_cast<const f32x4 *>(c1b_lds + 32 * cb + 8 * q + 4 * kh);
                    const f32x4 s4 = *reinterpret_cast<const f32x4 *>(c1post_lds + 32 * cb + 8 * q + 4 * kh);
                    f32x4 val;
#pragma unroll
                    for (int e = 0; e < 4; ++e) val[e] = __int_as_float(max(__float_as_int(__builtin_fmaf(c[4 * q + e], s4[e], b4[e])), 0));
                    omax = fmaxf(omax, fmaxf(fmaxf(val[0], val[1]), fmaxf(val[2], val[3])));
                    u32x2 pc[2];
                    split2_f16(val, pc);
                    if (row < T) {
                        char *dst = halo + row * kPlRowB + (32 * cb + 8 * q + 4 * kh) * 2;
                        *reinterpret_cast<u32x2 *>(dst) = pc[0];
                        *reinterpret_cast<u32x2 *>(dst + 128) = pc[1];
                    }
                }
            }
        }
    };
    auto c1_res_request = [&](int mbase) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 2; ++i) c1_request(mbase + lrow[i], c1d[i]);
    };

    int tr_n = 0;
    auto trace = [&](int tag) __attribute__((always_inline)) {
        if constexpr (ABL & 64) {
            if ((blockIdx.x == 0 || blockIdx.x == 256) && tid == 0 && tr_n < 250) {
                long long *tb = reinterpret_cast<long long *>(const_cast<void *>(p.res)) + ((blockIdx.x ? 1 : 0) * 256 + tr_n) * 2;
                tb[0] = tag, tb[1] = (long long)__builtin_readcyclecounter();
                ++tr_n;
            }
        }
    };
    trace(1);
    // ---- prologue: zero row, first halo slab, the first chunk of the weight ring
    pl_u32x4 hreg[SRC8 == 1 ? 1 : kDuHaloLoads];
    if (tid < 64) {
        bias_lds[tid] = p.bias[tn * 64 + tid], post_lds[tid] = p.post[tn * 64 + tid];
        if constexpr (SRC8 == 2) pre_lds[tid] = p.pre[tn * 64 + tid];
    }
    if constexpr (SRC8) {
        for (int i = tid; i < kC1WBytes / 16; i += kDuThreads)
            *reinterpret_cast<pl_u32x4 *>(c1w_lds + i * 16) = *reinterpret_cast<const pl_u32x4 *>(reinterpret_cast<const char *>(p.c1w) + i * 16);
        if (tid < 64) c1b_lds[tid] = p.c1b[tid], c1post_lds[tid] = p.c1post[tid];
    }
    if constexpr (SRC8 == 1) c1_halo_request(m0);
    else halo_issue(hreg, m0, 0);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) w_issue(ks, 0);
    if (tid < 16) *reinterpret_cast<pl_u32x4 *>(halo + T * kPlRowB + tid * 16) = pl_u32x4{0u, 0u, 0u, 0u};
    // the two workgroups of a CU run half a tile apart: the later half of the grid (the second workgroup every CU receives)
    // waits here, its loads in flight, while the first half is already in its tap loop
    if (p.skew > 0 && blockIdx.x >= (unsigned)(G >> 1))
        for (int i = 0; i < p.skew; ++i) __builtin_amdgcn_s_sleep(16);
    trace(2);
    if constexpr (SRC8 == 1) {
        lds_barrier();  // conv1's weight fragments and bias are in LDS
        c1_halo_write();
    } else {
        halo_write(hreg);
    }
    lds_barrier();
    trace(3);

    // operand registers of the pixels, two stages; stage 0 holds k-step 0 / 2, stage 1 k-step 1 / 3 of the current chunk
    pl_u32x4 xh[2][2], xl[2][2];
    const char *asrc[2];
    uint32_t mask[2] = {0u, 0u};
    auto set_asrc = [&](int tap) __attribute__((always_inline)) {
        const int toff = (W + 1) + (tap / 3 - 1) * W + (tap % 3 - 1);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = ((mask[i] >> tap) & 1u) ? lrow[i] + toff : T;
            asrc[i] = halo + r * kPlRowB + kh * 16;
        }
    };
    auto frags = [&](int ks, int st) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            xh[st][i] = *reinterpret_cast<const pl_u32x4 *>(asrc[i] + ks * 32);
            xl[st][i] = *reinterpret_cast<const pl_u32x4 *>(asrc[i] + 128 + ks * 32);
        }
    };

    for (;;) {
        if constexpr (SRC8 == 2) c1_res_request(m0);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m0 + lrow[i];
            uint32_t mk = 0;
            if (m < p.M) {
                const int b = fast_div(m, p.mg_hw), rem = m - b * HW;
                const int oh = fast_div(rem, p.mg_w), ow = rem - oh * W;
                mk = tap_mask9(oh - 1, ow - 1, p.H, W);
            }
            mask[i] = mk;
        }
        const int vn = v + G;
        const bool more = vn < p.tiles;
        const int m0n = more ? (xcd_tile_index(vn, p.tiles) / NS) * tile_stride : 0;

        f32x16 acc[2];
        if constexpr (SRC8 == 2) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const f32x16 c = c1_block(c1d[i], wn);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 b4 = *reinterpret_cast<const f32x4 *>(c1b_lds + 32 * wn + 8 * q + 4 * kh);
                    const f32x4 s4 = *reinterpret_cast<const f32x4 *>(c1post_lds + 32 * wn + 8 * q + 4 * kh);
                    const f32x4 k4 = *reinterpret_cast<const f32x4 *>(pre_lds + 32 * wn + 8 * q + 4 * kh);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        acc[i][4 * q + e] = __int_as_float(max(__float_as_int(__builtin_fmaf(c[4 * q + e], s4[e], b4[e])), 0)) * k4[e];
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        }
        set_asrc(0);
        frags(0, 0);

#pragma unroll 1
        for (int slab = 0; slab < NS; ++slab)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int cc = slab * 9 + tap;
            const bool last = cc == NCH - 1;
            const int ccn = last ? 0 : cc + 1;  // the ring refills with the next chunk of this workgroup's (cyclic) stream
            constexpr int kHaloTap = 8;
            if (tap == kHaloTap) {
                if constexpr (NS > 1 && !(ABL & 2)) {
                    const bool lastslab = slab == NS - 1;
                    halo_issue(hreg, lastslab ? m0n : m0, lastslab ? 0 : slab + 1, !lastslab || more);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int st = ks & 1;
                if (ks < 3) {
                    frags(ks + 1, st ^ 1);
                } else if (tap != 8) {
                    set_asrc(tap + 1);
                    frags(0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (!(ABL & 8)) {
                    acc[0] = mma(acc[0], wq[ks][0], xl[st][0]);
                    acc[1] = mma(acc[1], wq[ks][0], xl[st][1]);
                    acc[0] = mma(acc[0], wq[ks][1], xh[st][0]);
                    acc[1] = mma(acc[1], wq[ks][1], xh[st][1]);
                    acc[0] = mma(acc[0], wq[ks][0], xh[st][0]);
                    acc[1] = mma(acc[1], wq[ks][0], xh[st][1]);
                } else {
                    acc[0][ks] += __uint_as_float(wq[ks][0][0] ^ xl[st][0][1] ^ xh[st][1][2] ^ wq[ks][1][3]);
                }
                __builtin_amdgcn_sched_barrier(0);
                w_issue(ks, ccn);
            }
            if constexpr (NS > 1)
            if (tap == 8 && !last) {  // slab switch inside the tile
                lds_barrier();        // every wave has finished reading the old slab
                if constexpr (!(ABL & 2)) halo_write(hreg);
                lds_barrier();
                set_asrc(0);
                frags(0, 0);
            }
            trace(10 + tap);
        }

        // ---- epilogue (c3_conv3.h): the tile crosses LDS once, (pixel, 8-channel) items, residual, ReLU, split, two stores
        if constexpr (NS == 1) {
            if constexpr (SRC8 == 1) {
                c1_halo_request(more ? m0n : -0x40000000);
            } else if constexpr (!(ABL & 2)) {
                halo_issue(hreg, m0n, 0, more);
            }
        }
        lds_barrier();  // all waves are done with the halo rows
        trace(30);
        if constexpr (ABL & 4) {
            if (acc[0][0] == 12345.f && acc[1][3] == 1.f) p.range_flag[1] = 1u;
        } else {
        uint32_t ioff[4];
        pl_u32x4 rh[4], rl[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int idx = tid + kDuThreads * j;
            const int m = m0 + (idx >> 3);
            ioff[j] = m < p.M && (idx >> 3) < tile_stride ? (uint32_t)m * (uint32_t)PIXB + (uint32_t)(tn * 256 + (idx & 7) * 16) : kPlOob;
            if constexpr (RES && SRC8 != 2) {
                rh[j] = __builtin_bit_cast(pl_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rrsrc, ioff[j], 0, 0));
                rl[j] = __builtin_bit_cast(pl_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rrsrc, ioff[j] + 128, 0, 0));
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 bv = *reinterpret_cast<const f32x4 *>(bias_lds + cb0 + 8 * q);
                const f32x4 sv = *reinterpret_cast<const f32x4 *>(post_lds + cb0 + 8 * q);
                f32x4 val = {acc[i][4 * q], acc[i][4 * q + 1], acc[i][4 * q + 2], acc[i][4 * q + 3]};
#pragma unroll
                for (int e = 0; e < 4; ++e) val[e] = __builtin_fmaf(val[e], sv[e], bv[e]);
                *reinterpret_cast<f32x4 *>(halo + lrow[i] * kPlRowB + (cb0 + 8 * q) * 4) = val;
            }
        lds_barrier();
        trace(31);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int idx = tid + kDuThreads * j;
            const int pr = idx >> 3, g = idx & 7;
            const uint32_t off = ioff[j];
            f32x4 a = *reinterpret_cast<const f32x4 *>(halo + pr * kPlRowB + g * 32);
            f32x4 b = *reinterpret_cast<const f32x4 *>(halo + pr * kPlRowB + g * 32 + 16);
            if constexpr (RES && SRC8 != 2) {
                const f16x8 h8 = __builtin_bit_cast(f16x8, rh[j]), l8 = __builtin_bit_cast(f16x8, rl[j]);
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
            omax = fmaxf(omax, fmaxf(fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3])), fmaxf(fmaxf(b[0], b[1]), fmaxf(b[2], b[3]))));
            if constexpr (SPPF) {
                *reinterpret_cast<f32x4 *>(halo + pr * kPlRowB + g * 32) = a;
                *reinterpret_cast<f32x4 *>(halo + pr * kPlRowB + g * 32 + 16) = b;
                continue;
            }
            u32x2 pa[2], pb[2];
            split2_f16(a, pa);
            split2_f16(b, pb);
            const pl_u32x4 hi = {pa[0][0], pa[0][1], pb[0][0], pb[0][1]}, lo = {pa[1][0], pa[1][1], pb[1][0], pb[1][1]};
            __builtin_amdgcn_raw_buffer_store_b128(hi, orsrc, off, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(lo, orsrc, off + 128, 0, 0);
        }
        if constexpr (SPPF) {
            // thread (channel c = tid & 63, window w = (tid >> 6) & 1 of the tile, level half = tid >> 7): half 0 takes the nine
            // 3x3 bins, half 1 the four 2x2 bins and the 1x1 bin (c3_conv3.h)
            lds_barrier();
            const int sc = tid & 63, sw = (tid >> 6) & 1, half = tid >> 7;
            const int wb = m0 / HW + sw;
            const float *src = reinterpret_cast<const float *>(halo + (sw * HW) * kPlRowB) + sc;
            constexpr int RS = kPlRowB / 4;
            float mx[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) mx[k] = 0.f;
            if (half == 0) {
#pragma unroll
                for (int h = 0; h < 12; ++h)
#pragma unroll
                    for (int w = 0; w < 5; ++w) mx[(h / 4) * 3 + w / 2] = fmaxf(mx[(h / 4) * 3 + w / 2], src[(h * 5 + w) * RS]);
            } else {
#pragma unroll
                for (int h = 0; h < 12; ++h)
#pragma unroll
                    for (int w = 0; w < 5; ++w) mx[(h / 6) * 2 + w / 3] = fmaxf(mx[(h / 6) * 2 + w / 3], src[(h * 5 + w) * RS]);
                mx[4] = fmaxf(fmaxf(mx[0], mx[1]), fmaxf(mx[2], mx[3]));
            }
            if (wb * HW < p.M) {
                float *dst = p.spp + ((int64_t)wb * 14 + (half ? 9 : 0)) * C + tn * 64 + sc;
#pragma unroll
                for (int k = 0; k < 9; ++k)
                    if (k < (half ? 5 : 9)) dst[(int64_t)k * C] = mx[k];
            }
        }
        }
        trace(32);
        if (!more) break;
        lds_barrier();  // the staged tile has been read back
        if constexpr (SRC8 == 1) c1_halo_write();
        else if constexpr (!(ABL & 2)) halo_write(hreg);
        lds_barrier();
        trace(33);
        v = vn, m0 = m0n;
    }
    if (p.range_flag && !(omax < kF16Range)) atomicOr(p.range_flag, 1u);
}

}  // namespace c3

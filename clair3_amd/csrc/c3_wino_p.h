// c3_wino_p.h -- persistent form of the 32-tile x 64-cout Winograd F(2x2,3x3) workgroup (c3_wino.h).
//
// wino_conv_kernel_n64 spends about half of a workgroup's life outside the MFMA stream when Cin is small: the 64- and
// 128-channel blocks have only 4 / 8 K-chunks per workgroup, so the fixed costs -- index arithmetic, the first patch
// load (a full HBM/MALL round trip), the first V fragments, the residual loads of the epilogue -- are paid once per
// 4 / 8 chunks.  Here a workgroup keeps its column block and walks over row groups tm0, tm0 + stride, ...:
//   * the index arithmetic and the first patch of group g+1 are issued during the LAST chunk of group g, so they fly
//     under its MFMAs and its epilogue;
//   * V fragments rotate: xi's fragments for chunk c+1 are requested right after xi's MFMAs of chunk c were issued
//     (same registers), the ones for the next group's chunk 0 from inside the epilogue as soon as the accumulators they
//     replace have been written to LDS;
//   * residual pixels are requested before the LDS exchange they are added after.
// V comes through a buffer descriptor (scalar chunk offset + one lane offset register) instead of 64-bit pointers.
// Same arithmetic, same summation order and same memory layouts as wino_conv_kernel_n64.
#pragma once
#include <type_traits>

#include "c3_wino.h"

namespace c3 {

// ABL (tools/wino_probe only; 0 in the product): bit0 no patch loads, bit1 no transform+LDS writes, bit2 no V loads,
// bit3 no epilogue exchange/stores, bit4 no MFMAs.
// OPT bit1 (tools/wino_probe.hip only): workgroup 0 records the shader clock at its phase boundaries.
// a - b on both halves in one v_pk_add_f32 (hipcc emits two v_sub_f32 for the vector expression)
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ f32x4 pk_sub4(f32x4 a, f32x4 b) {
    const f32x2 lo = pk_sub(f32x2{a[0], a[1]}, f32x2{b[0], b[1]}), hi = pk_sub(f32x2{a[2], a[3]}, f32x2{b[2], b[3]});
    return f32x4{lo[0], lo[1], hi[0], hi[1]};
}

// F16: the Winograd-domain products on v_mfma_f32_32x32x16_f16 with both operands split into two fp16 pieces
// (fp16x3, c3_gemm.h SPLIT mode 2): the transform threads write U as two fp16 planes per xi (same 32 KiB), V arrives as
// two fp16 fragments per (xi, chunk, column block) instead of two k-halves of fp32 (same bytes, same addressing), and a
// chunk of 16 channels is ONE k-step: three matrix instructions of 32 cycles per xi and column block instead of eight of 64.
template <bool RES, int ABL = 0, int OPT = 0, bool F16 = false>
__global__ __launch_bounds__(256, 2) void wino_conv_kernel_p(WinoParams p) {
    constexpr int PT = 32, NT = 64;
    // one LDS object: U [16][32][16] floats (32 KiB) / epilogue exchange [16][32][32] (64 KiB), then tcoord[2][PT]
    __shared__ __attribute__((aligned(16))) char smem[65536 + 2 * PT * 8];
    char *ubuf = smem;
    int2 *tcoord = reinterpret_cast<int2 *>(smem + 65536);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ngm = (p.P + PT - 1) / PT;
    const int tn = blockIdx.x % p.tiles_n, tm0 = blockIdx.x / p.tiles_n, tstride = gridDim.x / p.tiles_n;
    const int n0 = tn * NT;

    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(p.x), 0, p.B * p.H * p.W * p.Cin * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t vrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(p.v), 0, 16 * p.Cin * p.Cout * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.B * p.H * p.W * p.Cout * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(RES ? p.res : p.out), 0, p.B * p.H * p.W * p.Cout * 4, 0x00020000);

    // ---- transform role: thread (tile, cin quad, channel pair)
    const int tl = tid >> 3, q = (tid >> 1) & 3, half = tid & 1;
    const int tpw = p.th * p.tw;
    auto setup = [&](int tm, int buf, uint32_t &base, uint32_t &okmask) __attribute__((always_inline)) {
        int pp = tm * PT + tl;
        const bool valid = pp < p.P;
        if (!valid) pp = p.P - 1;
        const int b = pp / tpw, r = pp - b * tpw;
        const int ty = r / p.tw, tx = r - ty * p.tw;
        const int iy0 = 2 * ty - 1, ix0 = 2 * tx - 1;
        base = (uint32_t)((((b * p.H + iy0) * p.W + ix0) * p.Cin + q * 4 + half * 2) * 4);
        uint32_t rows = 0, cols = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            rows |= (iy0 + k >= 0 && iy0 + k < p.H) ? 0xfu << (4 * k) : 0u;
            cols |= (ix0 + k >= 0 && ix0 + k < p.W) ? 0x1111u << k : 0u;
        }
        okmask = rows & cols;
        if ((tid & 7) == 0)
            tcoord[buf * PT + tl] = make_int2(((b * p.H + 2 * ty) * p.W + 2 * tx) * p.Cout * 4,
                                              (valid ? 1 : 0) | (2 * ty + 1 < p.H ? 2 : 0) | (2 * tx + 1 < p.W ? 4 : 0));
    };
    constexpr int kPlane = PT * 64;  // bytes of one xi plane of U (F16: two piece planes of PT * 32 bytes)
    const int u_wr = F16 ? tl * 32 + q * 8 + half * 4 : tl * 64 + ((q ^ ((tl >> 2) & 3)) << 4) + 8 * half;
    auto put_u = [&](char *dst, f32x2 v) __attribute__((always_inline)) {
        if constexpr (F16) {
            const f16x2 h0 = __builtin_convertvector(v, f16x2);
            const f32x2 hf = __builtin_convertvector(h0, f32x2);
            const f32x2 r = pk_sub(v, hf);
            *reinterpret_cast<uint32_t *>(dst) = __builtin_bit_cast(uint32_t, h0);
            *reinterpret_cast<uint32_t *>(dst + PT * 32) = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, f16x2));
        } else {
            *reinterpret_cast<f32x2 *>(dst) = v;
        }
    };

    // ---- MFMA role: wave owns xi = 4*wave .. 4*wave+3, one row block (32 tiles) x two column blocks (64 couts)
    const int frow = lane & 31, fhi = lane >> 5;
    const int a_rd = frow * 64, a_sw = (frow >> 2) & 3;
    const int nchunks = p.Cin / kWinoBK;
    const uint32_t v_lane = (uint32_t)lane * 16u;
    // byte offset of fragment (cb, xi = 4*wave + i, chunk c, g): ((((2tn+cb)*16 + 4wave+i)*nchunks + c)*2 + g) * 1024
    const int v_cb0 = ((tn * 2) * 16 + wave * 4) * nchunks * 2048;
    const int v_cbs = 16 * nchunks * 2048, v_is = nchunks * 2048;

    auto load_patch_rows = [&](f32x2 (&d)[4][4], uint32_t base, uint32_t okmask, int c, int dy0, int dy1) __attribute__((always_inline)) {
        const uint32_t choff = base + (uint32_t)(c * kWinoBK) * 4u;
#pragma unroll
        for (int dy = 0; dy < 4; ++dy) {
            if (dy < dy0 || dy >= dy1) continue;
#pragma unroll
            for (int dx = 0; dx < 4; ++dx) {
                const bool ok = (okmask >> (dy * 4 + dx)) & 1u;
                const uint32_t off = ok ? choff + (uint32_t)((dy * p.W + dx) * p.Cin) * 4u : 0x80000000u;
                if constexpr (ABL & 1) d[dy][dx] = f32x2{(float)off, 1.f};
                else d[dy][dx] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(xrsrc, off, 0, 0));
            }
        }
    };
    auto load_patch = [&](f32x2 (&d)[4][4], uint32_t base, uint32_t okmask, int c) __attribute__((always_inline)) {
        const uint32_t choff = base + (uint32_t)(c * kWinoBK) * 4u;
#pragma unroll
        for (int dy = 0; dy < 4; ++dy)
#pragma unroll
            for (int dx = 0; dx < 4; ++dx) {
                const bool ok = (okmask >> (dy * 4 + dx)) & 1u;
                const uint32_t off = ok ? choff + (uint32_t)((dy * p.W + dx) * p.Cin) * 4u : 0x80000000u;
                if constexpr (ABL & 1) d[dy][dx] = f32x2{(float)off, 1.f};
                else d[dy][dx] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(xrsrc, off, 0, 0));
            }
    };
    auto load_v = [&](f32x4 (&bf)[2][2], int i, int c) __attribute__((always_inline)) {  // [cb][g] of xi 4*wave + i
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                if constexpr (ABL & 4) bf[cb][g] = f32x4{1.f, 2.f, 3.f, (float)(c + cb)};
                else
                    bf[cb][g] = __builtin_bit_cast(
                        f32x4, __builtin_amdgcn_raw_buffer_load_b128(vrsrc, v_lane, v_cb0 + cb * v_cbs + i * v_is + (c * 2 + g) * 1024, 0));
            }
    };

    // OPT bit1 (probe only): workgroup 0 records the shader clock at phase boundaries into p.zeros (unused otherwise)
    int tr_n = 0;
    auto trace = [&](int tag) __attribute__((always_inline)) {
        if constexpr (OPT & 2) {
            if (blockIdx.x == 0 && lane == 0 && tr_n < 250) {
                long long *tb = reinterpret_cast<long long *>(const_cast<float *>(p.zeros)) + (wave * 256 + tr_n) * 2;
                tb[0] = tag, tb[1] = (long long)__builtin_readcyclecounter();
                ++tr_n;
            }
        }
    };
    if (tm0 >= ngm) return;
    uint32_t base, okmask;
    setup(tm0, 0, base, okmask);
    f32x2 d[4][4];
    load_patch(d, base, okmask, 0);
    f32x4 bf[4][2][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) load_v(bf[i], i, 0);

    int buf = 0;
    for (int tm = tm0; tm < ngm; tm += tstride, buf ^= 1) {
        f32x16 acc[4][2];
        float omax = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int v = 0; v < 16; ++v)
                    if (!F16) acc[i][cb][v] = 0.f;

        // one K chunk; LAST: the prefetches target the next group (patch here, V fragments from inside the epilogue)
        auto chunk = [&](int c, auto last_tag, auto first_tag) __attribute__((always_inline)) {
            constexpr bool LAST = decltype(last_tag)::value;
            constexpr bool FIRST = decltype(first_tag)::value;  // F16: the group's first products take C = 0 (no 128 v_mov)
#pragma unroll
            for (int dx = 0; dx < 4; ++dx) {  // t = B^T d
                const f32x2 d0 = d[0][dx], d1 = d[1][dx], d2 = d[2][dx], d3 = d[3][dx];
                d[0][dx] = pk_sub(d0, d2), d[1][dx] = d1 + d2, d[2][dx] = pk_sub(d2, d1), d[3][dx] = pk_sub(d1, d3);
            }
            trace(1);  // patch arrived, column transform done
            __syncthreads();  // every wave finished reading the previous chunk's U (or the previous group's exchange)
            trace(2);
            if constexpr (ABL & 2) asm volatile("" ::"v"(d[0][0]), "v"(d[1][1]), "v"(d[2][2]), "v"(d[3][3]));
#pragma unroll
            for (int i = 0; i < (ABL & 2 ? 0 : 4); ++i) {  // U = t B, xi = 4i + j
                const f32x2 t0 = d[i][0], t1 = d[i][1], t2 = d[i][2], t3 = d[i][3];
                char *dst = ubuf + u_wr;
                put_u(dst + (4 * i + 0) * kPlane, pk_sub(t0, t2));
                put_u(dst + (4 * i + 1) * kPlane, t1 + t2);
                put_u(dst + (4 * i + 2) * kPlane, pk_sub(t2, t1));
                put_u(dst + (4 * i + 3) * kPlane, pk_sub(t1, t3));
            }
            trace(3);  // U written (issue)
            __syncthreads();
            trace(4);
            if constexpr (LAST) {
                // coordinates of the next row group (at the end of the walk: this one again -- its patch is never used);
                // its first patch flies under these MFMAs and the epilogue
                setup(tm + tstride < ngm ? tm + tstride : tm, buf ^ 1, base, okmask);
            }
            __builtin_amdgcn_sched_barrier(0);
            trace(5);  // prefetches issued
            if constexpr (ABL & 16) {
                asm volatile("" ::"v"(bf[0][0][0]), "v"(bf[1][1][1]), "v"(bf[2][0][0]), "v"(bf[3][1][1]));
                if constexpr (!LAST) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) load_v(bf[i], i, c + 1);
                }
                load_patch(d, base, okmask, LAST ? 0 : c + 1);
                return;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const char *plane = ubuf + (wave * 4 + i) * kPlane;
                if constexpr (F16) {
                    // lane (tile, k half) holds channels 8 fhi .. 8 fhi + 7 of its tile: one 16-byte read per piece
                    const f16x8 a0 = *reinterpret_cast<const f16x8 *>(plane + frow * 32 + fhi * 16);
                    const f16x8 a1 = *reinterpret_cast<const f16x8 *>(plane + PT * 32 + frow * 32 + fhi * 16);
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb) {
                        const f16x8 v0 = __builtin_bit_cast(f16x8, bf[i][cb][0]), v1 = __builtin_bit_cast(f16x8, bf[i][cb][1]);
                        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        acc[i][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, v0, FIRST ? zero : acc[i][cb], 0, 0, 0);
                        acc[i][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, v1, acc[i][cb], 0, 0, 0);
                        acc[i][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, v0, acc[i][cb], 0, 0, 0);
                    }
                } else {
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const f32x4 a = *reinterpret_cast<const f32x4 *>(plane + a_rd + (((2 * g + fhi) ^ a_sw) << 4));
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], bf[i][0][g][j], acc[i][0], 0, 0, 0);
                        acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], bf[i][1][g][j], acc[i][1], 0, 0, 0);
                    }
                }
                }
                if constexpr (!LAST) load_v(bf[i], i, c + 1);  // same registers, a whole chunk ahead of their use
                // The next patch (chunk c+1, or the next group's chunk 0) is requested from INSIDE the MFMA stream, two
                // patch rows after xi 0 and two after xi 1: a buffer_load costs ~60 cycles of issue when the wave has
                // nothing else to do and next to nothing between two MFMAs (shader-clock trace in tools/wino_probe.hip:
                // the 16 loads took 1000 of a chunk's 6450 cycles when issued before the MFMAs).
                if (i < 2) load_patch_rows(d, base, okmask, LAST ? 0 : c + 1, 2 * i, 2 * i + 2);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if constexpr (F16) {  // nchunks >= 2 (Cin >= 64 on every Winograd layer; checked on the host)
            chunk(0, std::false_type{}, std::true_type{});
            for (int c = 1; c + 1 < nchunks; ++c) chunk(c, std::false_type{}, std::false_type{});
        } else {
            for (int c = 0; c + 1 < nchunks; ++c) chunk(c, std::false_type{}, std::false_type{});
        }
        chunk(nchunks - 1, std::true_type{}, std::false_type{});
        trace(6);  // all MFMAs of the group issued

        // ---- epilogue: one pass per column block (32 couts): exchange M_xi through LDS, A^T M A, bias (+res), ReLU, store
        float *mbuf = reinterpret_cast<float *>(ubuf);  // [16 xi][32 tiles][32 couts]
        if constexpr (ABL & 8) {
            float sacc = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int v = 0; v < 16; ++v) sacc += acc[i][0][v] + acc[i][1][v];
            if (sacc == 12345.678f) p.out[tid] = sacc;
#pragma unroll
            for (int i = 0; i < 4; ++i) load_v(bf[i], i, 0);
            continue;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int cq = tid & 7, t = tid >> 3;
            const int2 tc = tcoord[buf * PT + t];
            const int n = n0 + 32 * h + 4 * cq;
            const uint32_t o00 = (uint32_t)tc.x + (uint32_t)n * 4u;
            const uint32_t row_bytes = p.W * p.Cout * 4, px_bytes = p.Cout * 4, kOut = 0x80000000u;
            const bool v = tc.y & 1, vr = (tc.y & 3) == 3, vc = (tc.y & 5) == 5, vrc = (tc.y & 7) == 7;
            const uint32_t off[2][2] = {{v ? o00 : kOut, vc ? o00 + px_bytes : kOut},
                                        {vr ? o00 + row_bytes : kOut, vrc ? o00 + row_bytes + px_bytes : kOut}};
            f32x4 r[2][2];
            if constexpr (RES) {  // residual pixels: requested before the exchange, added after it
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        r[i][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rrsrc, off[i][j], 0, 0));
            }
            const f32x4 bias = *reinterpret_cast<const f32x4 *>(p.bias + n);
            __syncthreads();
            trace(7);
            {
                const int c32 = lane & 31;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int vv = 0; vv < 16; ++vv) {
                        const int tt = (vv & 3) + 8 * (vv >> 2) + 4 * fhi;
                        mbuf[((wave * 4 + i) * PT + tt) * 32 + c32] = acc[i][h][vv];
                    }
            }
            trace(8);  // exchange written (issue)
            __syncthreads();
            trace(9);
            // the accumulators just written free the registers of two xi's fragments for the next group's chunk 0
            load_v(bf[2 * h], 2 * h, 0);
            load_v(bf[2 * h + 1], 2 * h + 1, 0);
            f32x4 y[2][2];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 m[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) m[i] = *reinterpret_cast<const f32x4 *>(&mbuf[((4 * i + j) * PT + t) * 32 + 4 * cq]);
                const f32x4 s0 = m[0] + m[1] + m[2], s1 = pk_sub4(pk_sub4(m[1], m[2]), m[3]);
                if (j == 0) y[0][0] = s0, y[1][0] = s1;
                if (j == 1) y[0][0] += s0, y[1][0] += s1, y[0][1] = s0, y[1][1] = s1;
                if (j == 2) y[0][0] += s0, y[1][0] += s1, y[0][1] = pk_sub4(y[0][1], s0), y[1][1] = pk_sub4(y[1][1], s1);
                if (j == 3) y[0][1] = pk_sub4(y[0][1], s0), y[1][1] = pk_sub4(y[1][1], s1);
            }
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = __builtin_fmaf(y[i][j][e], p.post_scale, bias[e]);  // post_scale = 1: the plain add
                    if constexpr (RES) o += r[i][j];
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f);
                    if constexpr (F16) omax = fmaxf(omax, fmaxf(fmaxf(o[0], o[1]), fmaxf(o[2], o[3])));
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), orsrc, off[i][j], 0, 0);
                }
        }
        if constexpr (F16)
            if (p.range_flag && !(omax < kF16Range)) atomicOr(p.range_flag, 1u);
    }
}

}  // namespace c3

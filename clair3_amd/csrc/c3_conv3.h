// c3_conv3.h -- the six stride-1 3x3 convolutions of Clair3_F's residual blocks (clair3/model.py:200-235, 83 % of the
// network's FLOPs) as DIRECT convolutions on v_mfma_f32_32x32x16_f16, reading and writing "plane" activations, TWO workgroups
// per CU.
//
// Plane activations.  Every fp32 activation x is kept as the two fp16 pieces the fp16x3 products need (x = hi + lo)
// -- the same 4 bytes per value as fp32, but split ONCE by the producing epilogue instead of by every consumer tile.  Layout, C
// channels per pixel, NHWC order:
//     pixel row = C/64 slabs of 256 B;  slab s = [hi of channels 64s..64s+63 : 128 B][lo of the same channels : 128 B]
// so a lane's MFMA operand (8 consecutive channels of one pixel) is one 16-byte piece per plane.
//
// One workgroup (256 threads = 4 waves as 2 (pixels) x 2 (couts), persistent over tiles) = 128 consecutive output pixels
// (flattened over batch, rows, columns) x 64 output channels; a wave owns 64 x 32 outputs = two 32 x 32 accumulators:
//  * the input pixels all nine taps of those 128 outputs touch are the flat range [m0 - W - 1, m0 + 127 + W + 1] -- ONE contiguous
//    run of pixel rows.  For each 64-channel slab it is loaded into LDS once (164 rows x 256 B for W = 17) and every tap reads it
//    at a row offset dh*W + dw; taps that fall off the window are redirected to an all-zero row by a per-lane 9-bit mask -- no
//    im2col; LDS rows are 272 B apart (the ds_read_b128 of 16 consecutive rows covers all 64 banks once) and the (piece, k-step)
//    position inside a row is an immediate offset;
//  * LDS holds the halo tile and nothing else (45 KB; 70 KB with conv1's fragments in the first block), <= 214 registers: TWO
//    workgroups per CU, one wave of each on every SIMD, so a tile's non-matrix phases -- halo load, slab switch, the tile through
//    LDS, stores -- run under the matrix phase of the other workgroup;
//  * WEIGHTS NEVER TOUCH LDS: packed in fragment order (c3_pack.h: [tn][slab][tap][wn][k-step][piece][lane] x 16 B), a wave's
//    operand of one k-step is ONE contiguous 1 KB buffer load per piece, fetched straight into registers four k-steps (= one
//    chunk) ahead.  The tap loop has NO barrier: the four waves drift freely, the only workgroup-wide synchronisation left is the
//    slab switch and the epilogue.  The two waves of a cout half read the same kilobyte (the second read is an L1 hit);
//  * weights are the FIRST matrix operand, so a lane ends up with 4 consecutive output channels of one pixel; the epilogue adds
//    the bias, sends the tile through LDS and leaves as (pixel, 8-channel) items: residual, ReLU, split into the two fp16
//    pieces, two 16-byte stores; every barrier is LDS-only (lds_barrier).
// History: until round 4 these layers ran as ONE 512-thread workgroup per CU on 256-pixel tiles with the
// weights streaming through three LDS buffers and a barrier per chunk.  Its phase trace (profiles/r04_a_conv_probe.txt): a
// chunk of 1536 matrix cycles took 2100, and 8.4 k of a tile's 28.5 k cycles were its tail, with nothing to run under it.  The
// form below gives bit-identical rows (same products per accumulator, same order) and is 4 % faster per step on the same box
// (profiles/r04_b_ab_duo.txt); both forms sit at the clock the chip sustains under the matrix stream.
#pragma once
#include "c3_gemm.h"
#include "c3_kernels.h"

#ifndef C3_HALO_TAP
#define C3_HALO_TAP 8
#endif

namespace c3 {

typedef uint32_t pl_u32x4 __attribute__((ext_vector_type(4)));

constexpr int kPlBM = 128, kPlBN = 64;
constexpr int kPlThreads = 256;                       // 4 waves: 2 (pixels) x 2 (couts), 64 x 32 outputs each
constexpr int kPlRowB = 272;                           // LDS row stride
constexpr int kPlMaxW = 17;                            // widest image the halo tile is sized for (45x17 stage of the ONT window)
constexpr int kPlHaloRows = kPlBM + 2 * kPlMaxW + 2;   // 164
constexpr int kPlHaloBytes = (kPlHaloRows + 1) * kPlRowB;  // + the zero row
constexpr int kPlHaloLoads = (kPlHaloRows * 16 + kPlThreads - 1) / kPlThreads;  // 16-byte pieces per thread
constexpr uint32_t kPlOob = 0xffffff00u;               // buffer offset beyond every activation tensor (loads return 0, stores vanish)

struct PlaneConvParams {
    const void *x;        // plane activations [M][C/64][2][64] fp16
    const void *wf;       // weights in fragment order: [Cout/64][C/64][9 taps][2 cout halves][4 k-steps][hi | lo][64 lanes] x 16 B
    const float *bias;    // [Cout]
    const void *res;      // residual, plane layout of the output (RES)
    void *out;            // plane activations [M][Cout/64][2][64]
    uint32_t *range_flag;
    const float *post;    // [Cout] 2^-k: every output channel's weights are packed times its own power of two 2^k (c3_pack.h
                          // row_scales); undone here, exactly, inside the bias FMA
    const float *pre;     // [Cout] 2^k (SRC8 = 2: the residual enters the accumulators times it)
    int M, H, W;
    int tiles;            // ceil(M / 256) * (C / 64)
    // SRC8 (first residual block of Clair3_F, 8-channel windows): conv1 (clair3/model.py:316-317,391) is computed in here
    // from the int8 windows instead of being read as plane activations
    float *spp = nullptr;            // SPPF: PyramidPolling output [B][14 bins][C] (clair3/model.py:250-279), written instead of `out`
    const int8_t *x8 = nullptr;      // [B][Hin][Win][8] windows
    const uint32_t *c1w = nullptr;   // conv1_i8_f16_kernel's weight fragments (c3_conv1.h): [5 k-steps][2 column blocks][2 pieces][64 lanes][16 B]
    const float *c1b = nullptr;      // conv1 bias [64] (BatchNorm folded)
    const float *c1post = nullptr;   // conv1's [64] per-channel 2^-k (its fragments are packed times 2^k)
    int Hin = 0, Win = 0;
    uint32_t mg_hw = 0, mg_w = 0;    // fast_div magics of H * W and W (c3_gemm.h)
    int skew = 0;                    // units of 1024 cycles the second workgroup of a CU (the later half of the grid) starts later (probe knob)
};

// split four fp32 values into their fp16 pieces and store them behind `off` (hi plane) / `off + 128` (lo plane)
__device__ __forceinline__ void store_planes4(const __amdgpu_buffer_rsrc_t rsrc, uint32_t off, const f32x4 v) {
    u32x2 pc[2];
    split2_f16(v, pc);
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(st_u32x2, pc[0]), rsrc, off, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(st_u32x2, pc[1]), rsrc, off + 128, 0, 0);
}

// four consecutive channels of a plane activation back as fp32 (hi + lo is exact in fp32)
__device__ __forceinline__ f32x4 load_planes4(const __amdgpu_buffer_rsrc_t rsrc, uint32_t off) {
    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
    const f16x4 h = __builtin_bit_cast(f16x4, __builtin_amdgcn_raw_buffer_load_b64(rsrc, off, 0, 0));
    const f16x4 l = __builtin_bit_cast(f16x4, __builtin_amdgcn_raw_buffer_load_b64(rsrc, off + 128, 0, 0));
    return __builtin_convertvector(h, f32x4) + __builtin_convertvector(l, f32x4);
}

// ABL: ablation switches of tools/conv_probe.hip (0 in the product): 1 no weight loads, 2 no halo loads after the first tile,
// 4 no epilogue, 8 no matrix instructions, 64 shader-clock trace of two
// workgroups (wave 0) at phase boundaries into p.res ([2][256] x {tag, clock}).
// SRC8 (C = 64 only; the first residual block behind the 8-channel conv1): conv1's output never exists in HBM.
//   1  the INPUT halo rows are computed here from the int8 windows (conv1 + BatchNorm + ReLU, split into the two fp16 pieces,
//      written straight into the LDS halo tile): 165 rows x 64 channels = 6 groups of 32 pixels over the 4 waves, 20 matrix
//      instructions of 32 cycles per group against conv1_i8_f16_kernel's weight fragments (kept in LDS), taps requested
//      during the previous tile's last chunk;
//   2  the RESIDUAL (= conv1's output at the tile's own pixels) is computed in the accumulators' own layout -- conv1 with the
//      weights as first operand leaves a lane the same 4 consecutive channels of the same pixel as this kernel -- and is the
//      value the accumulators start from (times the channel's 2^k of this layer), 10 matrix instructions per 32 x 32 block.
// conv1 costs 7 MFLOP per window against 56 for each of these layers; what it saves is its own launch (16 us, store-bound) and
// 150 MB of HBM traffic per 256 windows (its output written once and read twice).
// SPPF (last convolution of the network, 12 x 5 windows): PyramidPolling (clair3/model.py:245-279: 3x3, 2x2 and 1x1 max-pooling
// bins over the 12 x 5 window, W padded on the right) is this kernel's epilogue.  Tiles are aligned to WINDOWS for that -- a tile
// starts every 2 windows = 120 pixels and its last 8 rows are computed and dropped (256 windows = 128 x 4 tiles = two per CU) -- so every bin of a window lies inside one tile: the ReLU'd
// fp32 tile goes back into LDS, thread (window, channel, level) takes the maxima of its bins over 60 LDS values and stores
// them.  No plane output (its only reader was the pooling kernel), no pooling launch, no atomics.
// C1 (with SRC8): channels of the windows, 8 or 9 (--enable_dwell_time adds the dwell channel).  Nine-byte pixels have no aligned
// 8-byte pieces, so the 9-channel form takes conv1's K as the three ROWS of the 3 x 3 patch: the three pixels of a row are 27
// consecutive bytes of the window tensor, padded to 32 = two k-steps (weights of the pad bytes are zero), fetched as four
// unaligned 8-byte pieces per row (gfx950 serves those at any byte offset -- tools/unaligned_probe.hip -- but zeroes every dword
// that straddles the end of the buffer, and bytes in front of the tensor must not be touched at all: at the left / right edge
// of the window the piece that mixes an outside pixel with inside ones is fetched one byte later / six bytes earlier and
// shifted, which also clears the outside bytes).  6 k-steps instead of 5, otherwise the 8-channel scheme.
template <int C, bool RES, int ABL = 0, int SRC8 = 0, bool SPPF = false, int C1 = 8>
__global__ __launch_bounds__(kPlThreads, 2) void conv3x3_planes_kernel(PlaneConvParams p) {
    static_assert(C1 == 8 || C1 == 9, "conv1 inside this kernel: 8- or 9-channel windows");
    constexpr int NT1 = C1 == 8 ? 5 : 6;  // conv1 k-steps of 16
    static_assert(SRC8 == 0 || C == 64, "conv1 feeds the 64-channel block only");
    static_assert(SRC8 != 2 || RES, "SRC8 = 2 replaces the residual read");
    constexpr int NS = C / 64;   // input slabs = output column tiles
    constexpr int PIXB = 4 * C;  // bytes per pixel
    constexpr int NCH = 9 * NS;  // weight chunks per tile
    constexpr int kC1WBytes = NT1 * 2 * 2 * 64 * 16;
    __shared__ __attribute__((aligned(16))) char smem[kPlHaloBytes + 768 + (SRC8 ? kC1WBytes + 512 : 0)];
    char *const halo = smem;
    float *const bias_lds = reinterpret_cast<float *>(smem + kPlHaloBytes);
    float *const post_lds = bias_lds + 64, *const pre_lds = bias_lds + 128;
    char *const c1w_lds = smem + kPlHaloBytes + 768;
    float *const c1b_lds = reinterpret_cast<float *>(c1w_lds + kC1WBytes);
    float *const c1post_lds = c1b_lds + 64;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int frow = lane & 31, kh = lane >> 5;
    const int W = p.W, HW = p.H * p.W;
    const int T = kPlBM + 2 * W + 2;  // halo rows in use; row T is the zero row
    const int G = gridDim.x;

    // PERSISTENT: workgroup w walks the tiles of virtual blocks w, w + G, ... (XCD-aware order); G is a multiple of 8 NS (or
    // the tile count), so a workgroup keeps its column tile tn -- and with it its weight stream -- for every tile it takes.
    const int tile_stride = SPPF ? (kPlBM / HW) * HW : kPlBM;  // SPPF: whole windows per tile (2 x 60 pixels)
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

    auto halo_issue = [&](pl_u32x4 (&h)[kPlHaloLoads], int mbase, int slab, bool on = true) __attribute__((always_inline)) {
        const int m_lo = mbase - W - 1;
        const uint32_t lim = on ? (uint32_t)p.M : 0u;
        int tid_ = tid;
        asm volatile("" : "+v"(tid_));
#pragma unroll
        for (int j = 0; j < kPlHaloLoads; ++j) {
            const int idx = tid_ + kPlThreads * j;
            const int row = idx >> 4, pos = idx & 15;
            const int pix = m_lo + row;
            const bool ok = row < T && (unsigned)pix < lim;
            const uint32_t off = ok ? (uint32_t)pix * (uint32_t)PIXB + (uint32_t)(slab * 256 + pos * 16) : kPlOob;
            h[j] = __builtin_bit_cast(pl_u32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, off, 0, 0));
        }
    };
    auto halo_write = [&](const pl_u32x4 (&h)[kPlHaloLoads]) __attribute__((always_inline)) {
        int tid_ = tid;
        asm volatile("" : "+v"(tid_));
#pragma unroll
        for (int j = 0; j < kPlHaloLoads; ++j) {
            const int idx = tid_ + kPlThreads * j;
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
    // (the long-latency requests go out first -- halo rows / window taps and the weight ring -- and only then the small tables that
    // are copied into LDS: a copy waits for its load, and every request behind that wait would start a memory latency late)
    pl_u32x4 hreg[SRC8 == 1 ? 1 : kPlHaloLoads];
    if constexpr (SRC8 == 1) c1_halo_request(m0);
    else halo_issue(hreg, m0, 0);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) w_issue(ks, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (tid < 64) {
        bias_lds[tid] = p.bias[tn * 64 + tid], post_lds[tid] = p.post[tn * 64 + tid];
        if constexpr (SRC8 == 2) pre_lds[tid] = p.pre[tn * 64 + tid];
    }
    if constexpr (SRC8) {
        for (int i = tid; i < kC1WBytes / 16; i += kPlThreads)
            *reinterpret_cast<pl_u32x4 *>(c1w_lds + i * 16) = *reinterpret_cast<const pl_u32x4 *>(reinterpret_cast<const char *>(p.c1w) + i * 16);
        if (tid < 64) c1b_lds[tid] = p.c1b[tid], c1post_lds[tid] = p.c1post[tid];
    }
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
            const int idx = tid + kPlThreads * j;
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
            const int idx = tid + kPlThreads * j;
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


// ------------------------------------------------------------------------------------------ plane utilities
// plane activations -> fp32 NHWC (parity tests: c3_debug_fetch)
__global__ void planes_to_f32_kernel(const void *x, float *out, int64_t M, int C) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // one thread per (pixel, channel)
    if (i >= M * C) return;
    const int64_t m = i / C;
    const int c = (int)(i - m * C);
    const _Float16 *row = reinterpret_cast<const _Float16 *>(x) + m * 2 * C + (c >> 6) * 128;
    out[i] = (float)row[c & 63] + (float)row[64 + (c & 63)];
}

// channel c of pixel `pix` of a plane activation tensor with C channels
__device__ __forceinline__ float plane_value(const _Float16 *x, int64_t pix, int C, int c) {
    const _Float16 *row = x + pix * 2 * C + (c >> 6) * 128 + (c & 63);
    return (float)row[0] + (float)row[64];
}

// PyramidPolling (clair3/model.py:245-279) over plane activations: the two kernels of c3_kernels.h with the loads replaced
template <int H, int W>
__global__ __launch_bounds__(256) void spp_planes_fixed_kernel(const void *__restrict__ in, float *__restrict__ out, int B, int C) {
    constexpr int P[3] = {3, 2, 1};
    constexpr int NB0 = ((H + (H + 2) / 3 - 1) / ((H + 2) / 3)) * ((W + (W + 2) / 3 - 1) / ((W + 2) / 3));
    constexpr int NB1 = ((H + (H + 1) / 2 - 1) / ((H + 1) / 2)) * ((W + (W + 1) / 2 - 1) / ((W + 1) / 2));
    constexpr int NBINS = NB0 + NB1 + 1;
    const _Float16 *x = reinterpret_cast<const _Float16 *>(in);
    const int64_t total = (int64_t)B * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        const int64_t b = i / C;
        float v[H * W];
#pragma unroll
        for (int k = 0; k < H * W; ++k) v[k] = plane_value(x, b * H * W + k, C, c);
        float m[NBINS];
        int base = 0;
#pragma unroll
        for (int pi = 0; pi < 3; ++pi) {
            const int p = P[pi];
            const int wh = (H + p - 1) / p, ww = (W + p - 1) / p;
            const int ohn = (H + wh - 1) / wh, own = (W + ww - 1) / ww;
            const int pad_h = ohn * wh - H > 0 ? ohn * wh - H : 0, pad_w = own * ww - W > 0 ? own * ww - W : 0;
            const int pt = pad_h / 2, pl = pad_w / 2;
#pragma unroll
            for (int oh = 0; oh < ohn; ++oh)
#pragma unroll
                for (int ow = 0; ow < own; ++ow) {
                    const int a0 = oh * wh - pt, a1 = a0 + wh, c0 = ow * ww - pl, c1 = c0 + ww;
                    const bool padded = a0 < 0 || a1 > H || c0 < 0 || c1 > W;  // F.pad zeros take part in the max
                    float mm = padded ? 0.f : -INFINITY;
#pragma unroll
                    for (int h = (a0 < 0 ? 0 : a0); h < (a1 > H ? H : a1); ++h)
#pragma unroll
                        for (int w = (c0 < 0 ? 0 : c0); w < (c1 > W ? W : c1); ++w) mm = fmaxf(mm, v[h * W + w]);
                    m[base + oh * own + ow] = mm;
                }
            base += ohn * own;
        }
        float *dst = out + b * NBINS * C + c;
#pragma unroll
        for (int k = 0; k < NBINS; ++k) dst[(int64_t)k * C] = m[k];
    }
}

__global__ __launch_bounds__(256) void spp_planes_kernel(SppParams p) {  // p.in points at plane activations
    const _Float16 *x = reinterpret_cast<const _Float16 *>(p.in);
    const int64_t total = (int64_t)p.B * p.nbins * p.C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % p.C);
        const int bin = (int)((i / p.C) % p.nbins);
        const int64_t b = i / ((int64_t)p.C * p.nbins);
        float m = p.pad[bin] ? 0.f : -INFINITY;
        for (int h = p.h0[bin]; h < p.h1[bin]; ++h)
            for (int w = p.w0[bin]; w < p.w1[bin]; ++w) m = fmaxf(m, plane_value(x, (b * p.H + h) * p.W + w, p.C, c));
        p.out[i] = m;
    }
}

}  // namespace c3

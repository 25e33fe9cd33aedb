// c3_gemm.h -- the one tiled fp32 MFMA contraction every dense layer of both networks runs on.
//
//   C[m][n] = epilogue( sum_k A[m][k] * Bt[n][k] )
//
// * A rows come from a pluggable loader: im2col gather of a 3x3 convolution over NHWC fp32 activations
//   (clair3/model.py:195,228-231), the int8 full-alignment window itself for conv1 (model.py:378-382,
//   x.float()/100 folded into the packed weights), the int8/int32 pileup window (model.py:131-132), or a
//   plain row-major fp32 matrix (nn.Linear L4, LSTM input projections).
// * Bt is always "[N][K], K contiguous" -- PyTorch's native nn.Linear layout; conv weights are packed once
//   on the host as [Cout][kh][kw][Cin] with the BatchNorm scale folded in.
// * Matrix core: v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD).  One wave owns a (BM/2)x(BN/2)
//   sub-tile = RBxCB accumulators of 32x32.  A and B fragments are read from LDS with ds_read_b128: lane l
//   takes 4 consecutive k of row (l&31) starting at k = 8g + 4*(l>>5); MFMA j of group g then contracts
//   k = 8g+j (lanes 0-31) and k = 8g+4+j (lanes 32-63) -- the same k permutation on both operands.
// * LDS tile rows are 128 B (BK = 32 floats); the 16-B chunk c of row r is stored at chunk c ^ ((r>>1)&7),
//   which makes both the staging ds_write_b128 (8 consecutive lanes = one row) and the fragment
//   ds_read_b128 (16 rows x one chunk column per lane group) bank-conflict free.
// * Global -> register -> LDS staging, double buffered: chunk k+1 is fetched into registers before the
//   MFMAs of chunk k are issued and written to the other LDS buffer after them; one barrier per chunk.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace c3 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int st_u32x2 __attribute__((__vector_size__(8)));  // the data operand type of __builtin_amdgcn_raw_buffer_store_b64

// SPLIT (fp16x3): x = h0 + h1 with two fp16 pieces (11 mantissa bits each, round to nearest, subnormals kept --
// v_mfma_f32_32x32x16_f16 honours subnormal inputs, tools/f16_denorm_probe.hip); x*w from x0w0 + x0w1 + x1w0, dropping
// x1w1 < 2^-22 of the product.  Same operand bytes as fp32 (two 16-bit pieces), 2.5 vector instructions per split
// value; rows as close to the reference as the fp32 path's (tests/diag/bf16x_study.py).  Needs |x| < 65504, which every activation and weight of these networks satisfies by orders of
// magnitude (int8/100 inputs, BatchNorm-scaled stages).
__device__ __forceinline__ void split2_f16(const f32x4 x, u32x2 (&piece)[2]) {
#pragma unroll
    for (int i = 0; i < 4; i += 2) {
        const f16x2 h = __builtin_convertvector(f32x2{x[i], x[i + 1]}, f16x2);
        const f32x2 r = f32x2{x[i], x[i + 1]} - __builtin_convertvector(h, f32x2);
        piece[0][i >> 1] = __builtin_bit_cast(uint32_t, h);
        piece[1][i >> 1] = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, f16x2));
    }
}

// Outputs at or above this magnitude make the host re-run the batch on fp32 matrix instructions: a consumer may add up
// to four of them (Winograd input transform) before splitting the sum into fp16 pieces (max 65504).
constexpr float kF16Range = 16000.f;

constexpr int kBK = 32;        // floats per K chunk
constexpr int kThreads = 256;  // 4 waves, 2 (M) x 2 (N)

__device__ __forceinline__ int lds_chunk_off(int row, int chunk) {
    return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}

// XCD-aware, bijective block -> tile map: block b runs on XCD b%8 (observed; speed only), so give every
// XCD a contiguous range of tiles (n fastest): the N-tiles of one M-tile and neighbouring M-tiles, which
// share A rows / convolution halos, then hit the same 4 MiB L2.
// floor(n / d) as ONE multiply-high instead of the ~25 instructions of a 32-bit division (they sit in the per-tile address code
// of the convolution kernels, whose vector instructions cost matrix-pipe time): magic = floor(2^32 / d) + 1, exact for
// 0 <= n with n * d < 2^32 (the host checks that: div_magic in c3_model.hip); magic 0 stands for d = 1.
__device__ __forceinline__ int fast_div(int n, uint32_t magic) { return magic ? (int)__umulhi((uint32_t)n, magic) : n; }

// 9-bit validity mask of a 3 x 3 patch whose top-left tap is input pixel (ih0, iw0) of an H x W image: bit 3 r + c set when
// (ih0 + r, iw0 + c) lies inside.  Three row tests and three column tests instead of nine (row, column) pairs.
__device__ __forceinline__ uint32_t tap_mask9(int ih0, int iw0, int H, int W) {
    uint32_t cb = 0, mk = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) cb |= (unsigned)(iw0 + c) < (unsigned)W ? 1u << c : 0u;
#pragma unroll
    for (int r = 0; r < 3; ++r) mk |= (unsigned)(ih0 + r) < (unsigned)H ? cb << (3 * r) : 0u;
    return mk;
}

__device__ __forceinline__ int xcd_tile_index(int block, int n_tiles) {
    const int xcd = block & 7, slot = block >> 3;
    const int q = n_tiles >> 3, r = n_tiles & 7;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + slot;
}

// ------------------------------------------------------------------------------------------ A loaders
// Every loader serves R = BM/32 rows per thread: rows lr + 32*i of the tile, 16-B chunk lc of the K chunk.
// A fetch is split in two so that the global-load latency hides behind the MFMAs of the current chunk:
//   issue(raw)        -- address arithmetic + the global loads, nothing that consumes the loaded values;
//   finish(raw, out)  -- whatever turns the raw registers into the fp32 chunk (int -> float conversion),
//                        called after the MFMA block, right before the LDS write.
// Padding taps / out-of-range elements are not branched around and not masked afterwards: their ADDRESS is
// redirected to a 256-byte zero page, so the loaded value already is the padding value.  Rows beyond M are
// clamped to a valid row -- their accumulators are never stored.

// 3x3 / pad 1 convolution over NHWC fp32, Cin % 32 == 0.  K order = (kh, kw, cin).
struct ConvLoaderParams {
    const float *x;
    const float *zeros;  // >= 16 readable zero bytes
    int Hin, Win, Cin, Ho, Wo, stride;
    int chunks_per_tap;  // Cin / 32
};
template <int R>
struct ConvLoader {
    typedef ConvLoaderParams Params;
    typedef f32x4 Raw;
    const float *x, *zeros;
    int64_t off[R];
    uint32_t mask[R];
    int Win, Cin, cpt_shift, cpt_mask;
    __device__ __forceinline__ void init(const Params &p, int m0, int lr, int lc, int M) {
        x = p.x, zeros = p.zeros, Win = p.Win, Cin = p.Cin;
        cpt_mask = p.chunks_per_tap - 1, cpt_shift = 31 - __builtin_clz(p.chunks_per_tap);  // power of two
        const int hw = p.Ho * p.Wo;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            int m = m0 + lr + 32 * i;
            if (m >= M) m = M - 1;
            const int b = m / hw, rem = m - b * hw;
            const int oh = rem / p.Wo, ow = rem - oh * p.Wo;
            const int ih0 = oh * p.stride - 1, iw0 = ow * p.stride - 1;
            off[i] = (((int64_t)b * p.Hin + ih0) * p.Win + iw0) * p.Cin + lc * 4;
            uint32_t mk = 0;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int ih = ih0 + t / 3, iw = iw0 + t % 3;
                if (ih >= 0 && ih < p.Hin && iw >= 0 && iw < p.Win) mk |= 1u << t;
            }
            mask[i] = mk;
        }
    }
    // loads of K chunk kc (stateless: any chunk, any order)
    __device__ __forceinline__ void issue(Raw (&raw)[R], int kc) const {
        const int tap = kc >> cpt_shift, cc = kc & cpt_mask;
        const int kh = tap / 3, kw = tap - kh * 3;
        const int64_t koff = (int64_t)(kh * Win + kw) * Cin + cc * kBK;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const bool ok = (mask[i] >> tap) & 1u;
            raw[i] = *reinterpret_cast<const f32x4 *>(ok ? x + off[i] + koff : zeros);
        }
    }
    __device__ __forceinline__ void finish(const Raw (&raw)[R], f32x4 (&out)[R]) const {
#pragma unroll
        for (int i = 0; i < R; ++i) out[i] = raw[i];
    }
};

// conv1 of Clair3_F straight from the int8 window (B, H, W, C), stride 2, pad 1, 3*C <= 32.
// K is padded to 3 chunks (one per kh) of 32 slots: slot j = kw*C + c for j < 3C (weights are zero beyond);
// the three input pixels (kw = 0..2) of one kh are 3C consecutive bytes starting at pixel (ih, iw0).
struct Conv1LoaderParams {
    const int8_t *x;
    const int8_t *zeros;
    int Hin, Win, C, Ho, Wo;
};
template <int R>
struct Conv1Loader {
    typedef Conv1LoaderParams Params;
    struct Raw {
        int v[4];
    };
    const int8_t *x, *zeros;
    int64_t off[R];     // byte offset of (b, ih0, iw0, 0) + 4*lc
    int ih0[R];
    uint32_t colok[R];  // bit e: slot 4*lc+e is a real (kw, c) whose column iw0+kw is inside the image
    int Hin, rowbytes;
    __device__ __forceinline__ void init(const Params &p, int m0, int lr, int lc, int M) {
        x = p.x, zeros = p.zeros, Hin = p.Hin, rowbytes = p.Win * p.C;
        const int hw = p.Ho * p.Wo;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            int m = m0 + lr + 32 * i;
            if (m >= M) m = M - 1;
            const int b = m / hw, rem = m - b * hw;
            const int oh = rem / p.Wo, ow = rem - oh * p.Wo;
            const int iw0 = ow * 2 - 1;
            ih0[i] = oh * 2 - 1;
            off[i] = (((int64_t)b * p.Hin + ih0[i]) * p.Win + iw0) * p.C + lc * 4;
            uint32_t ck = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j = lc * 4 + e;
                const int kw = (j >= p.C) + (j >= 2 * p.C);
                const int iw = iw0 + kw;
                if (j < 3 * p.C && iw >= 0 && iw < p.Win) ck |= 1u << e;
            }
            colok[i] = ck;
        }
    }
    __device__ __forceinline__ void issue(Raw (&raw)[R], int kh) const {  // chunk index = kh
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int ih = ih0[i] + kh;
            const bool row_ok = ih >= 0 && ih < Hin;
            const int8_t *p = x + off[i] + (int64_t)kh * rowbytes;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool ok = row_ok && ((colok[i] >> e) & 1u);
                raw[i].v[e] = *(ok ? p + e : zeros);
            }
        }
    }
    __device__ __forceinline__ void finish(const Raw (&raw)[R], f32x4 (&out)[R]) const {
#pragma unroll
        for (int i = 0; i < R; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) out[i][e] = (float)raw[i].v[e];
    }
};

// Row-major fp32 matrix A[M][lda]; the K range [k0, k0 + 32*nk) is selected by the kernel (split-K).
struct DenseLoaderParams {
    const float *a;
    int64_t lda;
};
template <int R>
struct DenseLoader {
    typedef DenseLoaderParams Params;
    typedef f32x4 Raw;
    const float *row[R];
    int k0;
    __device__ __forceinline__ void init(const Params &p, int m0, int lr, int lc, int M) {
        k0 = 0;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            int m = m0 + lr + 32 * i;
            if (m >= M) m = M - 1;
            row[i] = p.a + (int64_t)m * p.lda + lc * 4;
        }
    }
    __device__ __forceinline__ void seek(int k) { k0 = k; }
    __device__ __forceinline__ void issue(Raw (&raw)[R], int kc) const {
#pragma unroll
        for (int i = 0; i < R; ++i) raw[i] = *reinterpret_cast<const f32x4 *>(row[i] + k0 + kc * kBK);
    }
    __device__ __forceinline__ void finish(const Raw (&raw)[R], f32x4 (&out)[R]) const {
#pragma unroll
        for (int i = 0; i < R; ++i) out[i] = raw[i];
    }
};

// ------------------------------------------------------------------------------------------ epilogues
enum { EPI_BIAS = 0, EPI_BIAS_RELU = 1, EPI_BIAS_RES_RELU = 2, EPI_PARTIAL = 3,
       EPI_BIAS_RELU_PLANES = 4 };  // bias + ReLU, output written as plane activations (c3_conv3.h) with ldc = N channels

struct EpilogueParams {
    float *c;            // [M][ldc]   (EPI_PARTIAL: [split][M][ldc])
    const float *bias;   // [N]
    const float *res;    // residual, same layout as c (EPI_BIAS_RES_RELU)
    int64_t ldc;
    int64_t split_stride;  // M*ldc for EPI_PARTIAL
    uint32_t *range_flag = nullptr;  // SPLIT: set to 1 when an output reaches kF16Range (the consumers split it into fp16 pieces)
    const float *post = nullptr;  // SPLIT: [N] 2^-k -- every weight row is packed times its own power of two 2^k (so that its low
                                  // fp16 pieces stay normal numbers, c3_pack.h row_scales) and the sum is scaled back here,
                                  // exactly, inside the bias FMA; nullptr = 1
};

struct GemmParams {
    const float *bt;  // [N][ldb]
    const uint16_t *bt3;  // SPLIT: the same weights as two fp16 pieces, [2][N][ldb], times the per-tensor power of two
    int64_t ldb;
    int M, N;
    int nk;          // K chunks per block (per split)
    int tiles_n;     // N / BN
    int tiles;       // tiles_m * tiles_n
};

// SPLIT = 0: fp32 operands on v_mfma_f32_32x32x2_f32 (the forms of the range-guard fallback); SPLIT = 2: fp16x3.
template <class Loader, int EPI, int BM, int BN, int SPLIT = 0>
__global__ __launch_bounds__(kThreads) void gemm_mfma_kernel(typename Loader::Params lp, GemmParams gp,
                                                              EpilogueParams ep) {
    constexpr int RA = BM / 32, RBt = BN / 32;  // staged rows per thread
    constexpr int RB = BM / 64, CB = BN / 64;   // 32x32 accumulators per wave (rows x cols)
    // fp32: rows of 32 floats (128 B).  SPLIT: per operand NP piece planes with rows of 32 16-bit values (64 B);
    // (two fp16 pieces, three products).
    static_assert(SPLIT == 0 || SPLIT == 2, "fp32 or fp16x3");
    constexpr int NP = 2;
    constexpr int kRowB = SPLIT ? 64 : 128;
    constexpr int kPlaneA = BM * 64, kPlaneB = BN * 64;          // SPLIT: bytes of one piece plane
    constexpr int kStage = SPLIT ? NP * (kPlaneA + kPlaneB) : (BM + BN) * 128;  // bytes per LDS stage
    constexpr int kBase_b = SPLIT ? NP * kPlaneA : BM * 128;
    __shared__ __attribute__((aligned(16))) char smem[2 * kStage];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lc = tid & 7, lr = tid >> 3;

    const int tile = xcd_tile_index(blockIdx.x, gp.tiles);
    const int tm = tile / gp.tiles_n, tn = tile - tm * gp.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int split = blockIdx.y;
    const int k0 = split * gp.nk * kBK;

    Loader loader;
    loader.init(lp, m0, lr, lc, gp.M);
    if constexpr (requires { loader.seek(0); }) loader.seek(k0);

    struct BPieces {
        u32x2 p[NP];
    };
    using BReg = std::conditional_t<(SPLIT != 0), BPieces, f32x4>;  // one thread's 4 k of one weight row: fp32, or NP x 4 16-bit pieces
    const float *bptr[RBt];
    const uint16_t *bptr3[RBt];
    const int64_t piece_stride = (int64_t)gp.N * gp.ldb;
#pragma unroll
    for (int i = 0; i < RBt; ++i) {
        bptr[i] = gp.bt + (int64_t)(n0 + lr + 32 * i) * gp.ldb + k0 + lc * 4;
        bptr3[i] = gp.bt3 + (int64_t)(n0 + lr + 32 * i) * gp.ldb + k0 + lc * 4;
    }
    auto load_b = [&](BReg (&dst)[RBt], int kchunk) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < RBt; ++i) {
            if constexpr (SPLIT) {
#pragma unroll
                for (int q = 0; q < NP; ++q)
                    dst[i].p[q] = *reinterpret_cast<const u32x2 *>(bptr3[i] + q * piece_stride + (int64_t)kchunk * kBK);
            } else {
                dst[i] = *reinterpret_cast<const f32x4 *>(bptr[i] + (int64_t)kchunk * kBK);
            }
        }
    };

    // staging offsets.  SPLIT: 8-byte slot lc of the 64-byte row = half (lc & 1) of 16-byte chunk lc >> 1, chunks
    // swizzled by (row >> 2) & 3 so that the ds_read_b128 of 16 consecutive rows covers all 64 banks once
    int st_off_a[RA], st_off_b[RBt];
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        const int row = lr + 32 * i;
        st_off_a[i] = SPLIT ? row * 64 + (((lc >> 1) ^ ((row >> 2) & 3)) << 4) + (lc & 1) * 8 : lds_chunk_off(row, lc);
    }
#pragma unroll
    for (int i = 0; i < RBt; ++i) {
        const int row = lr + 32 * i;
        st_off_b[i] = kBase_b + (SPLIT ? row * 64 + (((lc >> 1) ^ ((row >> 2) & 3)) << 4) + (lc & 1) * 8 : lds_chunk_off(row, lc));
    }
    auto stage = [&](char *dst, const f32x4 (&av)[RA], const BReg (&bv)[RBt]) __attribute__((always_inline)) {
        if constexpr (SPLIT) {
#pragma unroll
            for (int i = 0; i < RA; ++i) {
                u32x2 pc[NP];
                split2_f16(av[i], pc);
#pragma unroll
                for (int q = 0; q < NP; ++q) *reinterpret_cast<u32x2 *>(dst + q * kPlaneA + st_off_a[i]) = pc[q];
            }
#pragma unroll
            for (int i = 0; i < RBt; ++i)
#pragma unroll
                for (int q = 0; q < NP; ++q) *reinterpret_cast<u32x2 *>(dst + q * kPlaneB + st_off_b[i]) = bv[i].p[q];
        } else {
#pragma unroll
            for (int i = 0; i < RA; ++i) *reinterpret_cast<f32x4 *>(dst + st_off_a[i]) = av[i];
#pragma unroll
            for (int i = 0; i < RBt; ++i) *reinterpret_cast<f32x4 *>(dst + st_off_b[i]) = bv[i];
        }
    };

    // fragment read offsets: row (lane&31) of each 32-row block, chunk 2g + (lane>>5)
    const int frow = lane & 31, fhi = lane >> 5, fsw = SPLIT ? (frow >> 2) & 3 : (frow >> 1) & 7;
    int rd_a[RB], rd_b[CB];
#pragma unroll
    for (int i = 0; i < RB; ++i) rd_a[i] = (wm * (BM / 2) + i * 32 + frow) * kRowB;
#pragma unroll
    for (int i = 0; i < CB; ++i) rd_b[i] = kBase_b + (wn * (BN / 2) + i * 32 + frow) * kRowB;

    f32x16 acc[RB][CB];
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int j = 0; j < CB; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

    // Software pipeline over K chunks, two register stages + two LDS stages, branch-free body:
    //   iteration kc:  issue global loads of chunk kc+2          (-> rawB / rbB, land during this iteration)
    //                  write chunk kc+1 (loaded LAST iteration)   (rawA / rbA -> LDS[nxt], no wait needed)
    //                  MFMAs on LDS[cur]
    //                  barrier
    // Chunk indices past the end are clamped to the last chunk (redundant loads / writes nobody reads), so
    // the body has no conditionals and the scheduler can interleave loader VALU with the MFMAs.
    typename Loader::Raw rawA[RA], rawB[RA];
    f32x4 ra[RA];
    BReg rbA[RBt], rbB[RBt];
    const int last = gp.nk - 1;
    loader.issue(rawA, 0);
    load_b(rbA, 0);
    loader.finish(rawA, ra);
    stage(smem, ra, rbA);
    {
        const int k1 = last < 1 ? last : 1;
        loader.issue(rawA, k1);
        load_b(rbA, k1);
    }
    __syncthreads();

    auto body = [&](int kc, typename Loader::Raw (&rCur)[RA], BReg (&bCur)[RBt], typename Loader::Raw (&rNext)[RA],
                    BReg (&bNext)[RBt]) __attribute__((always_inline)) {
        const char *cur = smem + (kc & 1) * kStage;
        char *nxt = smem + ((kc + 1) & 1) * kStage;
        const int k2 = kc + 2 < last ? kc + 2 : last;
        loader.issue(rNext, k2);
        load_b(bNext, k2);
        __builtin_amdgcn_sched_barrier(0);  // keep the global loads at the top: they must fly during the MFMAs
        loader.finish(rCur, ra);
        stage(nxt, ra, bCur);
        if constexpr (SPLIT) {
            // two k-groups of 16; per group and 32x32 block the three piece products, smallest first.  Weights are the
            // FIRST operand, as in the fp32 path: the accumulators hold the block transposed (see the epilogue).
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            u32x4 af[2][RB][NP], bf[2][CB][NP];
            auto frags = [&](int g, u32x4 (&ao)[RB][NP], u32x4 (&bo)[CB][NP]) __attribute__((always_inline)) {
                const int coff = ((2 * g + fhi) ^ fsw) << 4;
#pragma unroll
                for (int q = 0; q < NP; ++q) {
#pragma unroll
                    for (int i = 0; i < RB; ++i) ao[i][q] = *reinterpret_cast<const u32x4 *>(cur + q * kPlaneA + rd_a[i] + coff);
#pragma unroll
                    for (int i = 0; i < CB; ++i) bo[i][q] = *reinterpret_cast<const u32x4 *>(cur + q * kPlaneB + rd_b[i] + coff);
                }
            };
            auto mma = [&](f32x16 t, u32x4 w, u32x4 x) __attribute__((always_inline)) {
                return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, x), t, 0, 0, 0);
            };
            frags(0, af[0], bf[0]);
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                if (g == 0) frags(1, af[1], bf[1]);
#pragma unroll
                for (int i = 0; i < RB; ++i)
#pragma unroll
                    for (int c = 0; c < CB; ++c) {
                        f32x16 t = acc[i][c];
                        t = mma(t, bf[g][c][0], af[g][i][1]);
                        t = mma(t, bf[g][c][1], af[g][i][0]);
                        t = mma(t, bf[g][c][0], af[g][i][0]);
                        acc[i][c] = t;
                    }
            }
            __syncthreads();
            return;
        }
        // fragments of k-group g+1 are read from LDS while the MFMAs of group g run
        f32x4 a[2][RB], b[2][CB];
        {
            const int coff = (fhi ^ fsw) << 4;
#pragma unroll
            for (int i = 0; i < RB; ++i) a[0][i] = *reinterpret_cast<const f32x4 *>(cur + rd_a[i] + coff);
#pragma unroll
            for (int i = 0; i < CB; ++i) b[0][i] = *reinterpret_cast<const f32x4 *>(cur + rd_b[i] + coff);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g < 3) {
                const int coff = ((2 * (g + 1) + fhi) ^ fsw) << 4;
#pragma unroll
                for (int i = 0; i < RB; ++i) a[(g + 1) & 1][i] = *reinterpret_cast<const f32x4 *>(cur + rd_a[i] + coff);
#pragma unroll
                for (int i = 0; i < CB; ++i) b[(g + 1) & 1][i] = *reinterpret_cast<const f32x4 *>(cur + rd_b[i] + coff);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < RB; ++i)
#pragma unroll
                    for (int c = 0; c < CB; ++c)
                        acc[i][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[g & 1][c][j], a[g & 1][i][j], acc[i][c], 0, 0, 0);
        }
        __syncthreads();
    };
    // unrolled by two so the register stages swap roles by name (no copies: a copy would be a use of the
    // in-flight loads and stall on them)
    int kc = 0;
    for (; kc + 1 <= last; kc += 2) {
        body(kc, rawA, rbA, rawB, rbB);
        body(kc + 1, rawB, rbB, rawA, rbA);
    }
    if (kc <= last) body(kc, rawA, rbA, rawB, rbB);

    // epilogue.  The MFMAs above take the weight fragment as the FIRST operand, so each accumulator holds its 32x32
    // block transposed: C/D map of 32x32x2 = col lane&31 -> output ROW m, row (v&3) + 8*(v>>2) + 4*(lane>>5) -> output
    // COLUMN n.  A lane therefore owns 4 consecutive n for one m per v>>2: 16-byte stores (4 per block instead of 16
    // dword stores), f32x4 bias / residual loads, no per-element branch (rows beyond M get an out-of-range buffer
    // offset), 32-bit offsets.  The tile epilogue was 18 of a conv3 tile's 110 kcycles (shader-clock trace).
    float *cbase = ep.c + (EPI == EPI_PARTIAL ? (int64_t)split * ep.split_stride : 0);
    const __amdgpu_buffer_rsrc_t crsrc = __builtin_amdgcn_make_buffer_rsrc(cbase, 0, (uint32_t)((int64_t)gp.M * ep.ldc * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(EPI == EPI_BIAS_RES_RELU ? ep.res : ep.c), 0, (uint32_t)((int64_t)gp.M * ep.ldc * 4), 0x00020000);
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    float omax = 0.f;
    if constexpr (SPLIT == 2 && BM == 128 && (EPI == EPI_BIAS || EPI == EPI_BIAS_RELU_PLANES)) {
        // Coalesced epilogue.  A lane owns 4 consecutive columns of ONE row per store, so a wave's store instruction above
        // touches 32 different rows with 16 bytes each -- partial lines, one request per row.  These two epilogues write the
        // largest tensors of their networks (173 MB of LSTM2 pre-activations per 1024 windows; the stride-2 convolutions'
        // plane activations), so the tile crosses LDS (free now; rows 272 B apart, 64 columns per pass) and leaves as whole
        // 16-byte pieces of 8 consecutive columns, 8 lanes per row segment of 256 B (fp32) / 128 B per plane.
        constexpr int kRowE = 272;
        static_assert(2 * kStage >= 128 * kRowE, "the staged tile must fit the operand buffers");
#pragma unroll
        for (int sl = 0; sl < BN / 64; ++sl) {  // 64-column slices of the tile
            __syncthreads();  // operand reads of the last chunk / item reads of the previous slice are done
            if (BN == 64 || wn == sl) {
#pragma unroll
                for (int i = 0; i < RB; ++i)
#pragma unroll
                    for (int c = 0; c < CB; ++c)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int col = (BN == 64 ? wn * 32 : 0) + c * 32 + 4 * (lane >> 5) + 8 * q;  // inside the slice
                            const f32x4 bv = *reinterpret_cast<const f32x4 *>(ep.bias + n0 + sl * 64 + col);
                            const f32x4 sv = ep.post ? *reinterpret_cast<const f32x4 *>(ep.post + n0 + sl * 64 + col) : f32x4{1.f, 1.f, 1.f, 1.f};
                            f32x4 val = {acc[i][c][4 * q], acc[i][c][4 * q + 1], acc[i][c][4 * q + 2], acc[i][c][4 * q + 3]};
#pragma unroll
                            for (int e = 0; e < 4; ++e) val[e] = __builtin_fmaf(val[e], sv[e], bv[e]);
                            *reinterpret_cast<f32x4 *>(smem + (wm * 64 + i * 32 + (lane & 31)) * kRowE + col * 4) = val;
                        }
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int idx = tid + 256 * j;
                const int r = idx >> 3, g = idx & 7;  // row of the tile, group of 8 columns of the slice
                const int m = m0 + r, n = n0 + sl * 64 + g * 8;
                f32x4 a = *reinterpret_cast<const f32x4 *>(smem + r * kRowE + g * 32);
                f32x4 b = *reinterpret_cast<const f32x4 *>(smem + r * kRowE + g * 32 + 16);
                if constexpr (EPI == EPI_BIAS) {
                    const uint32_t off = m < gp.M ? (uint32_t)(((int64_t)m * ep.ldc + n) * 4) : 0xffffff00u;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, a), crsrc, off, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, b), crsrc, off + 16, 0, 0);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        a[e] = __int_as_float(max(__float_as_int(a[e]), 0));
                        b[e] = __int_as_float(max(__float_as_int(b[e]), 0));
                    }
                    omax = fmaxf(omax, fmaxf(fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3])), fmaxf(fmaxf(b[0], b[1]), fmaxf(b[2], b[3]))));
                    u32x2 pa[2], pb[2];
                    split2_f16(a, pa);
                    split2_f16(b, pb);
                    const u32x4 hi = {pa[0][0], pa[0][1], pb[0][0], pb[0][1]}, lo = {pa[1][0], pa[1][1], pb[1][0], pb[1][1]};
                    const uint32_t po = m < gp.M ? (uint32_t)((int64_t)m * ep.ldc * 4) + (uint32_t)((n >> 6) * 256 + (n & 63) * 2) : 0xffffff00u;
                    __builtin_amdgcn_raw_buffer_store_b128(hi, crsrc, po, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(lo, crsrc, po + 128, 0, 0);
                }
            }
        }
        if constexpr (EPI == EPI_BIAS_RELU_PLANES)
            if (ep.range_flag && !(omax < kF16Range)) atomicOr(ep.range_flag, 1u);
        return;
    }
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        const int m = m0 + wm * (BM / 2) + i * 32 + (lane & 31);
#pragma unroll
        for (int c = 0; c < CB; ++c) {
            const int nb = n0 + wn * (BN / 2) + c * 32 + 4 * (lane >> 5);
            // padding rows: an offset beyond every buffer this kernel writes (the LSTM2 projection's gx2 passes 2 GiB at 12710
            // windows, so 0x80000000 would be IN range there), with room for the + 32 q below
            const uint32_t off = m < gp.M ? (uint32_t)(((int64_t)m * ep.ldc + nb) * 4) : 0xffffff00u;
#pragma unroll
            for (int q = 0; q < 4; ++q) {  // columns nb + 8q .. nb + 8q + 3
                f32x4 val = {acc[i][c][4 * q], acc[i][c][4 * q + 1], acc[i][c][4 * q + 2], acc[i][c][4 * q + 3]};
                if (EPI != EPI_PARTIAL) {
                    const f32x4 bv = *reinterpret_cast<const f32x4 *>(ep.bias + nb + 8 * q);
                    const f32x4 sv = (SPLIT && ep.post) ? *reinterpret_cast<const f32x4 *>(ep.post + nb + 8 * q) : f32x4{1.f, 1.f, 1.f, 1.f};
#pragma unroll
                    for (int e = 0; e < 4; ++e) val[e] = __builtin_fmaf(val[e], sv[e], bv[e]);  // 1: the plain add
                }
                if (EPI == EPI_BIAS_RES_RELU) val += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rrsrc, off + 32 * q, 0, 0));
                if (EPI == EPI_BIAS_RELU || EPI == EPI_BIAS_RES_RELU || EPI == EPI_BIAS_RELU_PLANES) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) val[e] = __int_as_float(max(__float_as_int(val[e]), 0));
                }
                if constexpr (SPLIT != 0 && EPI != EPI_PARTIAL) omax = fmaxf(omax, fmaxf(fmaxf(fabsf(val[0]), fabsf(val[1])), fmaxf(fabsf(val[2]), fabsf(val[3]))));
                if constexpr (EPI == EPI_BIAS_RELU_PLANES) {
                    // channel n of pixel m: slab n >> 6, hi piece at 2 (n & 63), lo piece 128 bytes further; `off` is the pixel
                    // row (4 ldc bytes) + 4 nb, so the plane offset of column nb + 8 q is (off - 4 nb) + ...
                    const int n = nb + 8 * q;
                    u32x2 pc[2];
                    split2_f16(val, pc);
                    const uint32_t po = m < gp.M ? (uint32_t)((int64_t)m * ep.ldc * 4) + (uint32_t)((n >> 6) * 256 + (n & 63) * 2) : 0xffffff00u;
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(st_u32x2, pc[0]), crsrc, po, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(st_u32x2, pc[1]), crsrc, po + 128, 0, 0);
                } else
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, val), crsrc, off + 32 * q, 0, 0);
            }
        }
    }
    if constexpr (SPLIT != 0 && EPI != EPI_PARTIAL)
        if (ep.range_flag && !(omax < kF16Range)) atomicOr(ep.range_flag, 1u);  // also taken for NaN
}

}  // namespace c3

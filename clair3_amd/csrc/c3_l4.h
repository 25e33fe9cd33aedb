// c3_l4.h -- L4 of both networks (clair3/model.py:137-139 Clair3_P, :392-394 Clair3_F: selu(L4 x + b) on the flattened recurrence / pyramid
// output) as a split-K contraction that STREAMS: partial[s][window][feature] = sum over the K slice s of x[window][k] W4[feature][k], fp16x3.
//
// The shapes are all K: 1024 windows x 128 features x 10 560 inputs (pileup; 43 MB of fp32 activations read once) and 256 x 256 x 3 584
// (full alignment).  Until round 4 they ran on the tiled GEMM (c3_gemm.h: 128 x 64 tiles, 32-float chunks, both operands through registers
// and ds_write, one chunk of each in flight): 19.2 / 8.1 us.  Here
//  * a workgroup (512 threads = 8 waves as 2 x 4) owns 64 windows x 128 features x one K slice; the grid is (window tiles) x (feature
//    tiles) x S, dealt to the XCDs in contiguous runs of slices so that an XCD's L2 holds the ~2 weight slices its workgroups share;
//  * the ACTIVATIONS (fp32 rows) are requested kL4Depth chunks of 64 inputs ahead: thread t holds 8 consecutive inputs of row t >> 3 per
//    chunk in flight, splits them into fp16 pieces once (not once per reading wave) and writes them as a plane row -- hi 64 | lo 64, piece q
//    of row r at slot q ^ (r & 15), the layout of c3_conv3s2.h -- into one of two 16 KB stages;
//  * the WEIGHTS, packed in fragment order (c3_pack.h: [FC/64][K4/64][feature half][k-step][piece][lane] x 16 B, the packing of the
//    convolutions), come by LDS-DMA two chunks ahead into three 32 KB stages -- lane-linear landing IS fragment order -- and every wave
//    reads its fragments from there: straight into registers (the first form of this kernel) the two waves of a feature block each
//    fetched the same kilobytes through the vector memory pipe, 64 KB per chunk and CU next to 16 KB of activations;
//  * one barrier per chunk, behind an explicit s_waitcnt vmcnt(8) (the requests of the NEXT chunk's weights are older than the eight
//    operations issued since); per chunk and wave 12 matrix instructions (one 32 x 32 accumulator: lo x hi, hi x lo, hi x hi per k-step).
// Measured (tools/l4_probe.hip, profiles/r04_n_l4_probe.txt, r04_s_*): full alignment 8.1 -> 6.3 us alone (10.2 -> 8.9 us in the step); pileup
// 19.2 -> 15.4 us alone, 20.6 -> 17.7 us in the step.  Why not more: a plain read of the 43 MB takes 7.4 us, but a workgroup walks its 11
// chunks in lock step -- without any load the kernel still takes 9.4 us (3 of launch, 0.55 us per chunk: 768 matrix cycles per SIMD, the
// fragment reads and a barrier), and the two load streams add to that instead of hiding under it; more slices make it worse (S = 33: two
// rounds of workgroups), fewer leave CUs idle; one accumulator per k-step, a step-major activation layout ([step][window][320]: perfectly
// linear reads) changed nothing.
// The partials carry the features' powers of two (c3_pack.h row_scales) exactly as before; splitk_reduce_selu_kernel / the tail kernel
// add them in the fixed order s = 0..S-1, so a window's bits do not depend on the batch it travels in.
#pragma once
#include "c3_conv3.h"

namespace c3 {

constexpr int kL4BM = 64, kL4BN = 128, kL4Threads = 512;
constexpr int kL4Row = 256, kL4Stage = kL4BM * kL4Row;  // 16 KB per stage
constexpr int kL4Depth = 3;                             // activation chunks in flight ahead of the one being split; also the weight stages (slots named
                                                        // statically: chunk j lives in slot j % kL4Depth of both rings)

struct L4Params {
    const float *a;  // [n][lda] fp32 activations
    int64_t lda;
    const void *wf;  // [FC/64][nk][2 feature halves][4 k-steps][hi | lo][64 lanes] x 16 B, every feature times its power of two
    float *part;     // [S][n][FC]
    int n, FC, nk, S;  // nk = K4 / 64 chunks in all; S divides it
    int m_tiles, n_tiles;  // ceil(n / 64), FC / 128
};

// ABL (probes only; 0 in the product): 1 no activation loads, 2 no weight requests, 4 no matrix instructions
constexpr int kL4WStage = 32768;  // one 64-input chunk of the 128 features of a tile: two fragment-ordered 16 KB chunks (c3_pack.h)
template <int ABL = 0>
__global__ __launch_bounds__(kL4Threads, 2) void l4_stream_kernel(L4Params p) {
    __shared__ __attribute__((aligned(1024))) char smem[2 * kL4Stage + 3 * kL4WStage];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;  // 2 x 4 waves, 32 windows x 32 features each
    const int frow = lane & 31, kh = lane >> 5;

    const int L = xcd_tile_index(blockIdx.x, gridDim.x);  // an XCD takes a contiguous run of (slice, feature tile, window tile)
    const int per_s = p.m_tiles * p.n_tiles;
    const int s = L / per_s, rest = L - s * per_s;
    const int nt = rest / p.m_tiles, mt = rest - nt * p.m_tiles;
    const int m0 = mt * kL4BM;
    const int cps = p.nk / p.S, c0 = s * cps;  // this workgroup's chunks: c0 .. c0 + cps - 1

    const __amdgpu_buffer_rsrc_t arsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.a), 0, (uint32_t)((int64_t)p.n * p.lda * 4), 0x00020000);
    // activations: thread t -> inputs 8 (t & 7) .. + 7 of row t >> 3 of every chunk
    const int ar = tid >> 3, ao = tid & 7;
    const uint32_t a_row = m0 + ar < p.n ? (uint32_t)(((int64_t)(m0 + ar) * p.lda + (int64_t)c0 * 64 + ao * 8) * 4) : kPlOob;
    pl_u32x4 ra[kL4Depth][2];
    auto a_issue = [&](int d, int c) __attribute__((always_inline)) {  // chunk c of this workgroup -> ring slot d
        if constexpr (ABL & 1) {
            ra[d][0] = pl_u32x4{(uint32_t)c, 0u, 0u, 0u}, ra[d][1] = pl_u32x4{0u, 0u, 0u, 0u};
            return;
        }
        const uint32_t off = (c < cps && a_row != kPlOob) ? a_row + (uint32_t)c * 256u : kPlOob;
        ra[d][0] = __builtin_bit_cast(pl_u32x4, __builtin_amdgcn_raw_buffer_load_b128(arsrc, off, 0, 0));
        ra[d][1] = __builtin_bit_cast(pl_u32x4, __builtin_amdgcn_raw_buffer_load_b128(arsrc, off, 16, 0));
    };
    // ring slot d (chunk j lives in slot j % kL4Depth: the slots are named statically, a register move would wait for the loads) ->
    // plane row in `stage`
    const uint32_t a_dst = (uint32_t)(ar * kL4Row + ((ao ^ (ar & 15)) << 4));
    auto a_split = [&](int d, int stage) __attribute__((always_inline)) {
        u32x2 pa[2], pb[2];
        split2_f16(__builtin_bit_cast(f32x4, ra[d][0]), pa);
        split2_f16(__builtin_bit_cast(f32x4, ra[d][1]), pb);
        const pl_u32x4 hi = {pa[0][0], pa[0][1], pb[0][0], pb[0][1]}, lo = {pa[1][0], pa[1][1], pb[1][0], pb[1][1]};
        char *dst = smem + stage * kL4Stage;
        *reinterpret_cast<pl_u32x4 *>(dst + a_dst) = hi;
        *reinterpret_cast<pl_u32x4 *>(dst + (a_dst ^ 128u)) = lo;  // piece 8 + ao: slot (ao ^ x) ^ 8
    };
    const __amdgpu_buffer_rsrc_t wall = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.wf), 0, (uint32_t)((int64_t)p.FC * p.nk * 256), 0x00020000);
    typedef void __attribute__((address_space(3))) *lds_ptr;
    // weight chunk c of both 64-feature tiles -> stage d, by LDS-DMA: lane L's 16 bytes land at M0 + 16 L, so a fragment-ordered kilobyte
    // arrives in fragment order; wave w brings kilobytes 4 w .. 4 w + 3 of the 32; a chunk beyond the slice asks an out-of-range offset
    auto w_dma = [&](int d, int c) __attribute__((always_inline)) {
        if constexpr (ABL & 2) return;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int j = wave * 4 + i, t = j >> 4;
            const uint32_t off = c < cps ? (uint32_t)(((nt * 2 + t) * p.nk + c0 + c) * 16384 + (j & 15) * 1024 + lane * 16) : kPlOob;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wall, (lds_ptr)(smem + 2 * kL4Stage + d * kL4WStage + j * 1024), 16, off, 0, 0, 0);
        }
    };
    const uint32_t w_lds = (uint32_t)(2 * kL4Stage + (wn >> 1) * 16384 + (wn & 1) * 8192 + lane * 16);
    auto w_frag = [&](int d, int ks, int piece) __attribute__((always_inline)) {
        return *reinterpret_cast<const pl_u32x4 *>(smem + w_lds + d * kL4WStage + ks * 2048 + piece * 1024);
    };
    auto mma = [](f32x16 c, pl_u32x4 w, pl_u32x4 x) __attribute__((always_inline)) {
        if constexpr (ABL & 4) {
            c[0] += __uint_as_float(w[0] ^ x[0]), c[5] += __uint_as_float(w[3] ^ x[3]);
            return c;
        } else {
            return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, x), c, 0, 0, 0);
        }
    };

    // ---- prologue: kL4Depth chunks of both operands requested (chunk by chunk: the loads come back in order), chunk 0 split into stage 0
    // the requests of weight chunks 0 and 1 FIRST: the loads come back in order, so the wait for activation chunk 0 below covers them
    w_dma(0, 0);
    w_dma(1, 1);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
#pragma unroll
    for (int d = 0; d < kL4Depth; ++d) a_issue(d, d);
    a_split(0, 0);
    a_issue(0, kL4Depth);
    lds_barrier();

    // fragment address of piece kh of row 32 wm + frow in stage 0 (c3_conv3s2.h: the other pieces are that XOR a constant)
    uint32_t va0 = (uint32_t)((wm * 32 + frow) * kL4Row) + (uint32_t)((kh ^ (frow & 15)) * 16);
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    auto chunk = [&](int c, int slot, int next_slot) __attribute__((always_inline)) {  // slot = c % kL4Depth, next_slot = (c + 1) % kL4Depth: constants at every call
        pl_u32x4 xh[4], xl[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            xh[ks] = *reinterpret_cast<const pl_u32x4 *>(smem + (va0 ^ (uint32_t)(32 * ks)));
            xl[ks] = *reinterpret_cast<const pl_u32x4 *>(smem + (va0 ^ (uint32_t)(128 + 32 * ks)));
        }
        // stage (c + 2) % 3 = (c - 1) % 3 was read in the chunk before this one: behind that chunk's barrier it is free
        w_dma(next_slot == 0 ? 1 : next_slot == 1 ? 2 : 0, c + 2);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const pl_u32x4 w0 = w_frag(slot, ks, 0), w1 = w_frag(slot, ks, 1);
            acc = mma(acc, w0, xl[ks]);
            acc = mma(acc, w1, xh[ks]);
            acc = mma(acc, w0, xh[ks]);
        }
        // the next chunk's activations (requested kL4Depth chunks ago) -> the other stage; its slot is requested again
        a_split(next_slot, (c + 1) & 1);
        a_issue(next_slot, c + 1 + kL4Depth);
        // the weight requests of chunk c + 1 (made one chunk ago) have landed for this wave once only the eight younger operations -- the
        // two activation loads of the chunk before, this chunk's four requests and two loads -- are outstanding; behind the barrier
        // everybody's have
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        lds_barrier();
        va0 ^= (uint32_t)kL4Stage;
    };
    // a loop body WITHOUT exits, the remainder behind it: hipcc sizes every s_waitcnt vmcnt(N) for the emptiest path into it, and an early
    // exit is routed through the loop's one latch -- a path on which a chunk's weights look like the youngest loads in flight
    static_assert(kL4Depth == 3, "the loop below names the ring slots");
    int c = 0;
    for (; c + 3 <= cps; c += 3) {
        chunk(c, 0, 1);
        chunk(c + 1, 1, 2);
        chunk(c + 2, 2, 0);
    }
    if (c < cps) {
        chunk(c, 0, 1);
        if (c + 1 < cps) chunk(c + 1, 1, 2);
    }

    // partial tile: lane (window 32 wm + frow; features 32 wn + 8 q + 4 kh .. + 3)
    const int row = m0 + wm * 32 + frow;
    const __amdgpu_buffer_rsrc_t prsrc =
        __builtin_amdgcn_make_buffer_rsrc(p.part + (int64_t)s * p.n * p.FC, 0, (uint32_t)((int64_t)p.n * p.FC * 4), 0x00020000);
    const uint32_t poff = row < p.n ? (uint32_t)((row * p.FC + nt * kL4BN + wn * 32 + 4 * kh) * 4) : kPlOob;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const pl_u32x4 v = {__float_as_uint(acc[4 * q]), __float_as_uint(acc[4 * q + 1]), __float_as_uint(acc[4 * q + 2]), __float_as_uint(acc[4 * q + 3])};
        __builtin_amdgcn_raw_buffer_store_b128(v, prsrc, poff, 32 * q, 0);
    }
}

}  // namespace c3

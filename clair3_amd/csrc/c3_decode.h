// c3_decode.h -- SURVEY 8f N1, first slice: the arithmetic of the reference decoder on the device.
//
// For every probability row, clair3/CallVariants.py:510-659 (possible_outcome_probabilites_from) enumerates the joint
// probabilities of ten outcome classes -- with the indel-length heads ~800 float32 products per row, in pure Python
// (3.8 k rows/s/core measured in SURVEY 8a A7) -- and output_from (:676-741) only ever looks at the maximum of each
// class list, at which class lists contain the overall maximum, and at the position of that maximum in its list.
// This kernel produces exactly those: per row and class the maximum and the index of its FIRST occurrence in the
// reference's enumeration order, plus the homo-reference early-exit test (:532-534 / :573-576).  Products are formed
// in the reference's order with one float32 rounding each (numpy float32 scalars): bit-identical values, so
// `maximum in class_list` in the reference is `class_max == overall_max` here.  Allele strings, alt_info handling and
// the retry loop of output_from stay in Python.
//
// Classes, in the order of the reference's max(...) call (:722-733):
//   0 homo_Ref (1)          1 homo_SNP (4: AA CC GG TT)      2 hetero_SNP (6: AC AG AT CG CT GT)
//   3 homo_Ins (16|1)       4 homo_Del (16|1)                5 hetero_ACGT_Ins (64|4: length-major, base A C G T)
//   6 hetero_InsIns (136|1: i <= j, i outer)                 7 hetero_ACGT_Del (64|4)
//   8 hetero_DelDel (241|1: i, j in 1..16, i == j only for 16)   9 hetero_InsDel (256|1: deletion length outer)
// (list lengths with | without --add_indel_length).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace c3 {

struct DecodeParams {
    const float *y;         // [B][ldy]: 24 | 90 probabilities at the start of every row
    int64_t ldy;            // row stride in floats
    const uint8_t *ref21;   // [B] gt21 index of (ref, ref): 0 AA, 4 CC, 7 GG, 9 TT          (COLS = false)
    float *maxp;            // [B][10]                                                        (COLS = false)
    int32_t *argmax;        // [B][10]                                                        (COLS = false)
    uint8_t *early;         // [B] 1 = the reference returns [homo_Ref_probability] without enumerating (COLS = false)
    float *cols;            // [B][ldy] the row's kDecodeCols decoder columns                 (COLS = true)
    int B, indel;
};

constexpr int kDecodeClasses = 10;
// Decoder columns (c3_model_set_decode_columns): the same results appended to the probability row itself, for a caller
// that does not know the reference base when it asks for the prediction (_torch_predict sees only X,
// clair3/CallVariantsFromCffi.py:48-52).  Only class 0 and the early exit depend on that base, so they are given for
// all four:
//   [0..8]   maxima of classes 1..9          [9..12]  homo_Ref probability for reference base A, C, G, T
//   [13..21] positions of those maxima       [22]     early-exit bits (bit b: base b takes the early exit)
//   [23..26] the class output_from (:722-751) settles on FIRST for reference base A, C, G, T: 0 when the overall maximum is
//            the homo_Ref probability (or the base takes the early exit), else the first class in the order of its
//            if / elif chain (:753-978: homo_SNP, hetero_SNP, homo_Ins, hetero_ACGT_Ins, hetero_InsIns, homo_Del,
//            hetero_ACGT_Del, hetero_DelDel, hetero_InsDel) whose list contains that maximum
//   [27..30] 100 x QUAL of that first decision, quality_score_from (:375-381) of the overall maximum, as an integer:
//            QUAL = col / 100.0 is the double round(tmp, 2) returns
// positions, classes, bits and 100 x QUAL as float values (all < 2^24: exact).
constexpr int kDecodeCols = 31;

// quality_score_from (clair3/CallVariants.py:375-381) of a float32 probability, times 100 and rounded to an integer.  The
// reference evaluates `((1.0 - p) + 1e-10) / (p + 1e-10)` on a numpy float32 scalar -- float32 arithmetic under numpy >= 2
// (python floats are weak scalars), which is what the goldens were generated with -- and takes math.log of the result in
// double precision; Phred_Trans = -10 * log(e, 10) (:27).
__device__ __forceinline__ float qual_x100(float p) {
    const float num = __fadd_rn(__fsub_rn(1.0f, p), 1e-10f), den = __fadd_rn(p, 1e-10f);
    const float q = __fdiv_rn(num, den);
    const double phred_trans = -10.0 * 0.43429448190325176;  // -10 * log10(e)
    double tmp = phred_trans * log((double)q) + 10.0;
    tmp = tmp > 0.0 ? tmp : 0.0;
    return (float)rint(tmp * 100.0);
}

// one wave per row; candidates of a class are dealt to the lanes in enumeration order, then a 64-lane max with
// smallest-index tie-break (key = value bits << 32 | ~index: probabilities are >= 0, so float order == bit order)
template <bool COLS>
__global__ __launch_bounds__(256) void outcome_maxima_kernel(DecodeParams p) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.B) return;
    const float *y = p.y + (int64_t)row * p.ldy;
    const float *g = y, *z = y + 21, *p1 = y + 24, *p2 = y + 57;
    const int ref = COLS ? 0 : p.ref21[row];
    const float hr = z[0], hv = z[1], ht = z[2];
    float *cols = COLS ? p.cols + (int64_t)row * p.ldy : nullptr;

    // COLS: the overall maximum of classes 1..9 and the first class of the reference's if / elif chain that holds it
    float best19 = -1.f;
    int best_rank = 99, best_cls = 0;
    auto reduce_store = [&](int cls, unsigned long long key) {
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) {
            const unsigned long long o = __shfl_xor(key, m);
            key = o > key ? o : key;
        }
        if (lane == 0) {
            if constexpr (COLS) {
                if (cls) {
                    const float v = __uint_as_float((unsigned)(key >> 32));
                    cols[cls - 1] = v, cols[12 + cls] = (float)(int32_t)(~(unsigned)key);
                    const int rank_of[10] = {0, 1, 2, 3, 6, 4, 5, 7, 8, 9};  // position of class cls in the if / elif chain
                    const int rk = rank_of[cls];
                    if (v > best19 || (v == best19 && rk < best_rank)) best19 = v, best_rank = rk, best_cls = cls;
                }
            } else {
                p.maxp[(int64_t)row * kDecodeClasses + cls] = __uint_as_float((unsigned)(key >> 32));
                p.argmax[(int64_t)row * kDecodeClasses + cls] = (int32_t)(~(unsigned)key);
            }
        }
    };
    float ref_b = 0.f;   // lanes 0..3: homo_Ref probability / early exit for reference base A, C, G, T
    bool early_b = false;
    // class 0 and the early exit for each of the four possible reference bases (lanes 0..3)
    auto all_bases = [&](float scale, bool lengths_ok) {
        const int hs[4] = {0, 4, 7, 9};
        const float gb = g[hs[lane & 3]];
        const unsigned long long bits = __ballot(lane < 4 && lengths_ok && hr >= 0.5f && gb >= 0.5f);
        ref_b = p.indel ? __fmul_rn(scale, gb) : __fmul_rn(hr, gb);
        early_b = (bits >> (lane & 3)) & 1u;
        if (lane < 4) cols[9 + lane] = ref_b;
        if (lane == 0) cols[22] = (float)(unsigned)(bits & 15u);
    };
    // per reference base (lanes 0..3 hold its homo_Ref probability and early bit after all_bases): winner class + 100 x QUAL
    auto first_decision = [&]() {
        const float b19 = __shfl(best19, 0);
        const int c19 = __shfl(best_cls, 0);
        if (lane < 4) {
            const bool is_ref = early_b || !(b19 > ref_b);  // is_reference is tested first: a tie goes to homo_Ref
            cols[23 + lane] = is_ref ? 0.f : (float)c19;
            cols[27 + lane] = qual_x100(is_ref ? ref_b : b19);
        }
    };
    auto mk = [](float v, int idx) { return ((unsigned long long)__float_as_uint(v) << 32) | (unsigned)~idx; };
    const unsigned long long none = 0ull;  // below every real candidate (idx 0xffffffff never occurs)

    if (!p.indel) {
        if constexpr (COLS) all_bases(0.f, true);
        else if (lane == 0) p.early[row] = hr >= 0.5f && g[ref] >= 0.5f;
        const int hsnp[4] = {0, 4, 7, 9}, tsnp[6] = {1, 2, 3, 5, 6, 8};
        reduce_store(0, lane == 0 ? mk(__fmul_rn(hr, g[ref]), 0) : none);
        reduce_store(1, lane < 4 ? mk(__fmul_rn(hv, g[hsnp[lane & 3]]), lane) : none);
        reduce_store(2, lane < 6 ? mk(__fmul_rn(ht, g[tsnp[lane < 6 ? lane : 0]]), lane) : none);
        reduce_store(3, lane == 0 ? mk(__fmul_rn(hv, g[15]), 0) : none);
        reduce_store(4, lane == 0 ? mk(__fmul_rn(hv, g[10]), 0) : none);
        reduce_store(5, lane < 4 ? mk(__fmul_rn(g[16 + (lane & 3)], ht), lane) : none);
        reduce_store(6, lane == 0 ? mk(__fmul_rn(ht, g[15]), 0) : none);
        reduce_store(7, lane < 4 ? mk(__fmul_rn(g[11 + (lane & 3)], ht), lane) : none);
        reduce_store(8, lane == 0 ? mk(__fmul_rn(ht, g[10]), 0) : none);
        reduce_store(9, lane == 0 ? mk(__fmul_rn(ht, g[20]), 0) : none);
        if constexpr (COLS) first_decision();
        return;
    }

    const float v0 = __fmul_rn(p1[16], p2[16]);
    if constexpr (COLS) all_bases(__fmul_rn(v0, hr), p1[16] >= 0.5f && p2[16] >= 0.5f);
    else if (lane == 0) p.early[row] = p1[16] >= 0.5f && p2[16] >= 0.5f && hr >= 0.5f && g[ref] >= 0.5f;
    {
        const int hsnp[4] = {0, 4, 7, 9}, tsnp[6] = {1, 2, 3, 5, 6, 8};
        reduce_store(0, lane == 0 ? mk(__fmul_rn(__fmul_rn(v0, hr), g[ref]), 0) : none);
        reduce_store(1, lane < 4 ? mk(__fmul_rn(__fmul_rn(v0, hv), g[hsnp[lane & 3]]), lane) : none);
        reduce_store(2, lane < 6 ? mk(__fmul_rn(__fmul_rn(v0, ht), g[tsnp[lane < 6 ? lane : 0]]), lane) : none);
    }
    {   // homo_Ins / homo_Del: i = 1..16
        const int i = (lane & 15) + 1;
        const float xi = __fmul_rn(hv, g[15]), xd = __fmul_rn(hv, g[10]);
        reduce_store(3, lane < 16 ? mk(__fmul_rn(__fmul_rn(p1[16 + i], p2[16 + i]), xi), lane) : none);
        reduce_store(4, lane < 16 ? mk(__fmul_rn(__fmul_rn(p1[16 - i], p2[16 - i]), xd), lane) : none);
    }
    {   // hetero_ACGT_Ins / _Del: index = (i-1)*4 + base, 64 candidates = one per lane
        const int i = (lane >> 2) + 1, b = lane & 3;
        reduce_store(5, mk(__fmul_rn(__fmul_rn(__fmul_rn(p1[16], p2[16 + i]), g[16 + b]), ht), lane));
        reduce_store(7, mk(__fmul_rn(__fmul_rn(__fmul_rn(p1[16 - i], p2[16]), g[11 + b]), ht), lane));
    }
    {   // hetero_InsIns: (i, j), j >= i, i outer: 136 candidates
        const float x = __fmul_rn(ht, g[15]);
        unsigned long long key = none;
        for (int c = lane; c < 136; c += 64) {
            int i = 1, rem = c;  // row i of the triangle has 17 - i entries
            while (rem >= 17 - i) rem -= 17 - i, ++i;
            const int j = i + rem;
            const unsigned long long k2 = mk(__fmul_rn(__fmul_rn(p1[16 + i], p2[16 + j]), x), c);
            key = k2 > key ? k2 : key;
        }
        reduce_store(6, key);
    }
    {   // hetero_DelDel: i outer, j inner, (i == j) skipped unless i == 16: 241 candidates
        const float x = __fmul_rn(ht, g[10]);
        unsigned long long key = none;
        for (int c = lane; c < 256; c += 64) {
            const int i = (c >> 4) + 1, j = (c & 15) + 1;
            if (i == j && i != 16) continue;
            // list position = c minus the diagonal entries skipped before (i, j): one per earlier row, plus this row's if j > i
            const int skipped = (i - 1) + (j > i && i != 16 ? 1 : 0);
            const unsigned long long k2 = mk(__fmul_rn(__fmul_rn(p1[16 - i], p2[16 - j]), x), c - skipped);
            key = k2 > key ? k2 : key;
        }
        reduce_store(8, key);
    }
    {   // hetero_InsDel: deletion length i outer, insertion length j inner: 256 candidates
        const float x = __fmul_rn(ht, g[20]);
        unsigned long long key = none;
        for (int c = lane; c < 256; c += 64) {
            const int i = (c >> 4) + 1, j = (c & 15) + 1;
            const unsigned long long k2 = mk(__fmul_rn(__fmul_rn(p1[16 - i], p2[16 + j]), x), c);
            key = k2 > key ? k2 : key;
        }
        reduce_store(9, key);
    }
    if constexpr (COLS) first_decision();
}

}  // namespace c3

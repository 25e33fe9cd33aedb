// c3_rows.h -- SURVEY 8f N1, the last per-row Python of the decoder: the VCF text of the rows of a batch whose FIRST decision
// stands, in one pass over the batch on the host (plain C++, no HIP: the reference's decode workers are forked children that never
// touch the device, clair3/CallVariantsFromCffi.py:302-353).
//
// What it restates, block by block (clair3/CallVariants.py):
//   :1127-1154  batch_output / output_with: "chr:pos:seq" and "depth-KEY n KEY n ..." into contig, position, reference base, depth, dict
//   :117-201    insertion_bases_using_alt_info_from / deletion_bases_using_alt_info_from (proposals, the general range, return_multi)
//   :662-673    find_alt_base (stable sort by falling count, the depth-gap rule)
//   :748-1008   output_from's first pass per class: the alleles of the winning entry, or "the reads do not offer it"
//   :1176-1394  output_with's tail: genotype string, read counts per allele (AD), AF, QUAL (quality_score_from :375-381), FILTER, the row
//   :721-1016   the loop's later passes (a rejected candidate is zeroed, the next best is tried): the walk over the nine class lists
// It is an ACCELERATOR of clair3_amd/vcf_rows.py (RowPrinter.rows), not a second decoder: whatever is not plainly its business -- a
// maximum shared by two classes (first pass or walk), a reference base outside the IUPAC table, bytes outside printable ASCII, numbers
// that are not plain digits, a depth of zero, more than kMaxKeys alleles -- is handed back (status 1) and takes the Python path it
// took before.  tests/test_rows_c.py holds every row it does print to the text of that
// Python path, and tests/test_decode_dropin.py holds both to the unpatched reference decoder, character for character.
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>

#include "../../include/c3hip.h"

namespace c3rows {

constexpr int kMaxKeys = 192;  // distinct alleles of one alt_info string (a real candidate has a handful)
constexpr int kMaxLen = 16;    // VariantLength.max (clair3/task/variant_length.py:6-12), as clair3_amd/decode.py MAX_LEN

struct Str {  // a slice of the batch's text
    const char *p;
    int n;
    bool eq(const Str &o) const { return n == o.n && (n == 0 || memcmp(p, o.p, (size_t)n) == 0); }
};

struct Entry {  // one item of the row's alt_info dictionary, in first-insertion order (a repeated key keeps its place, the LAST count wins)
    Str key;
    long long count;
};

struct Row {
    Entry e[kMaxKeys];
    int n = 0;
    long long depth = 0;
    // answers of the allele lookups for THIS dictionary, each distinct question asked once (vcf_rows._Lookups does the same): a row that
    // rejects hundreds of candidates asks the same seventeen proposals again and again.  [kind I / D][proposed length 0 = None .. 16]
    mutable signed char one_n[2][kMaxLen + 1], two_n[2];
    mutable Str one_s[2][kMaxLen + 1], two_s[2][2];
    void forget() const { memset(one_n, -1, sizeof one_n), two_n[0] = two_n[1] = -1; }
};

static inline bool plain_ascii(const char *p, int n) {
    for (int i = 0; i < n; ++i)
        if ((unsigned char)p[i] < 0x20 || (unsigned char)p[i] > 0x7e) return false;
    return true;
}
// str.rstrip(): trailing white space goes (only the ASCII kinds can be there: everything else was refused by plain_ascii)
static inline int rstrip(const char *p, int n) {
    while (n > 0 && (p[n - 1] == ' ' || p[n - 1] == '\n' || p[n - 1] == '\t' || p[n - 1] == '\r' || p[n - 1] == '\v' || p[n - 1] == '\f')) --n;
    return n;
}
// int(text) for the texts the pipeline writes: plain decimal digits (anything else Python may or may not accept: handed back)
static inline bool plain_int(const char *p, int n, long long *out) {
    if (n <= 0 || n > 17) return false;
    long long v = 0;
    for (int i = 0; i < n; ++i) {
        if (p[i] < '0' || p[i] > '9') return false;
        v = v * 10 + (p[i] - '0');
    }
    *out = v;
    return true;
}

// ---- "depth-KEY n KEY n ..." (:1150-1154 and the same lines of vcf_rows.rows): a = text.rstrip().split("-"); depth = int(a[0]);
// seqs = a[1].split(" "); dict(zip(seqs[::2], ints(seqs[1::2])))
static bool parse_alt(const char *p, int n, Row &r) {
    n = rstrip(p, n);
    int d0 = 0;
    while (d0 < n && p[d0] != '-') ++d0;
    if (!plain_int(p, d0, &r.depth)) return false;
    r.n = 0;
    r.forget();
    if (d0 >= n) return true;  // no "-": an empty dictionary
    int s = d0 + 1, e = s;
    while (e < n && p[e] != '-') ++e;  // a[1] ends at the next "-" (what lies behind it is dropped by the reference too)
    if (e == s) return true;  // "": split(" ") gives [""], zip with no counts gives nothing
    // tokens of a[1].split(" "): every single space separates (two spaces in a row make an empty token: handed back)
    Str tok[2 * kMaxKeys + 2];
    int nt = 0;
    for (int i = s; i <= e;) {
        int j = i;
        while (j < e && p[j] != ' ') ++j;
        if (j == i) return false;  // an empty token (k[0] would raise in the reference)
        if (nt >= 2 * kMaxKeys + 2) return false;
        tok[nt++] = Str{p + i, j - i};
        i = j + 1;
    }
    for (int t = 0; t + 1 < nt; t += 2) {  // (a key without a count is dropped by zip)
        long long c;
        if (!plain_int(tok[t + 1].p, tok[t + 1].n, &c)) return false;
        int at = -1;
        for (int k = 0; k < r.n; ++k)
            if (r.e[k].key.eq(tok[t])) { at = k; break; }
        if (at >= 0) r.e[at].count = c;
        else {
            if (r.n >= kMaxKeys) return false;
            r.e[r.n++] = Entry{tok[t], c};
        }
    }
    // (int() is applied to EVERY odd token before zip cuts the lists: a trailing count without a partner cannot exist, a trailing key can)
    for (int t = 1; t < nt; t += 2) {
        long long c;
        if (!plain_int(tok[t].p, tok[t].n, &c)) return false;
    }
    return true;
}

// ---- find_alt_base (:662-673).  out: the alleles' bases by falling count (stable), n_out of them (<= cap); *alt: the base it settles
// on (0 = None).  false: a key "X" without a base (alt_base[1] raises in the reference)
static bool find_alt_base(const Row &r, char want, char *bases, int cap, int *n_out, char *alt) {
    char b[kMaxKeys];
    long long c[kMaxKeys];
    int n = 0;
    for (int k = 0; k < r.n; ++k)
        if (r.e[k].key.p[0] == 'X') {
            if (r.e[k].key.n < 2) return false;
            b[n] = r.e[k].key.p[1], c[n] = r.e[k].count, ++n;
        }
    // sorted(..., key=count, reverse=True): stable, equal counts keep their order
    for (int i = 1; i < n; ++i) {
        const char bi = b[i];
        const long long ci = c[i];
        int j = i - 1;
        while (j >= 0 && c[j] < ci) b[j + 1] = b[j], c[j + 1] = c[j], --j;
        b[j + 1] = bi, c[j + 1] = ci;
    }
    *n_out = n < cap ? n : cap;
    for (int i = 0; i < *n_out; ++i) bases[i] = b[i];
    if (n == 0) { *alt = 0; return true; }
    long long have = -1;
    bool found = false;
    if (want)
        for (int i = 0; i < n; ++i)
            if (b[i] == want) { have = c[i], found = true; break; }
    *alt = (!found || c[0] - have >= 9) ? b[0] : want;  // max_depth_gap = 9
    return true;
}

// ---- insertion_bases_using_alt_info_from / deletion_bases_using_alt_info_from (:117-201).  kind 'I' / 'D'; propose 0 = None; ignore:
// bases to leave out; multi: return_multi.  Returns the number of alleles in out[] (0 = "" / nothing), at most two.
static int indel_bases(const Row &r, char kind, int propose, int infer, const Str *ignore, bool multi, Str out[2]) {
    if (propose && kind == 'I') propose += 1;  // include reference base
    if (r.n == 0) return 0;
    Str pk[kMaxKeys], gk[kMaxKeys];
    long long pc[kMaxKeys], gc[kMaxKeys];
    int np = 0, ng = 0;
    for (int k = 0; k < r.n; ++k) {
        if (r.e[k].key.p[0] != kind) continue;
        const Str key{r.e[k].key.p + 1, r.e[k].key.n - 1};
        const bool ign = ignore && key.eq(*ignore);
        if (propose && key.n == propose && !ign) pk[np] = key, pc[np] = r.e[k].count, ++np;
        else if (1 <= key.n && key.n <= infer && !ign) gk[ng] = key, gc[ng] = r.e[k].count, ++ng;
    }
    auto first_max = [](const Str *k, const long long *c, int n) {  // max(dict, key=dict.get): the first of the largest
        int at = 0;
        for (int i = 1; i < n; ++i)
            if (c[i] > c[at]) at = i;
        return k[at];
    };
    if (propose && np) { out[0] = first_max(pk, pc, np); return 1; }
    if (multi) {
        // sorted(items, key=count)[::-1]: ascending and stable, then reversed -- of equal counts the LATER one comes first
        if (kind == 'D' && ng <= 1) return 0;
        if (ng == 0) return 0;
        int a = 0;  // the largest, the last of them
        for (int i = 1; i < ng; ++i)
            if (gc[i] >= gc[a]) a = i;
        if (ng == 1) { out[0] = gk[a]; return 1; }
        int b = -1;  // the next in that order
        for (int i = 0; i < ng; ++i) {
            if (i == a) continue;
            if (b < 0 || gc[i] >= gc[b]) b = i;
        }
        if (kind == 'D') {  // [longer, shorter]: the first stays in front only when it is strictly longer
            if (gk[a].n > gk[b].n) out[0] = gk[a], out[1] = gk[b];
            else out[0] = gk[b], out[1] = gk[a];
        } else {
            out[0] = gk[a], out[1] = gk[b];
        }
        return 2;
    }
    if (ng) { out[0] = first_max(gk, gc, ng); return 1; }
    return 0;
}

static int lookup(const Row &r, char kind, int propose, int infer, Str out[2]) {  // no bases to ignore, one allele
    const int k = kind == 'I' ? 0 : 1;
    if (r.one_n[k][propose] < 0) {
        Str o[2] = {{nullptr, 0}, {nullptr, 0}};
        r.one_n[k][propose] = (signed char)indel_bases(r, kind, propose, infer, nullptr, false, o);
        r.one_s[k][propose] = o[0];
    }
    out[0] = r.one_s[k][propose];
    return r.one_n[k][propose];
}
static int lookup_two(const Row &r, char kind, int infer, Str out[2]) {  // return_multi
    const int k = kind == 'I' ? 0 : 1;
    if (r.two_n[k] < 0) {
        Str o[2] = {{nullptr, 0}, {nullptr, 0}};
        r.two_n[k] = (signed char)indel_bases(r, kind, 0, infer, nullptr, true, o);
        r.two_s[k][0] = o[0], r.two_s[k][1] = o[1];
    }
    out[0] = r.two_s[k][0], out[1] = r.two_s[k][1];
    return r.two_n[k];
}

// the entries beside the probability lists (clair3_amd/decode.py class_entry; clair3/CallVariants.py:318-371)
struct Tables {
    short insins[136][2], deldel[241][2], insdel[256][2];
    Tables() {
        int n = 0;
        for (int i = 1; i <= kMaxLen; ++i)
            for (int j = i; j <= kMaxLen; ++j) insins[n][0] = (short)i, insins[n][1] = (short)j, ++n;
        n = 0;
        for (int i = 1; i <= kMaxLen; ++i)
            for (int j = 1; j <= kMaxLen; ++j) {
                if (i == j && i != kMaxLen) continue;
                deldel[n][0] = (short)(i < j ? i : j), deldel[n][1] = (short)(i < j ? j : i), ++n;
            }
        n = 0;
        for (int i = 1; i <= kMaxLen; ++i)
            for (int j = 1; j <= kMaxLen; ++j) insdel[n][0] = (short)i, insdel[n][1] = (short)j, ++n;
    }
};
static const Tables &tables() {
    static const Tables t;
    return t;
}
static const char kHomoSnp[4][3] = {"AA", "CC", "GG", "TT"};                     // HOMO_SNP_LABELS (clair3/task/gt21.py:111-112)
static const char kHeteroSnp[6][3] = {"AC", "AG", "AT", "CG", "CT", "GT"};       // HETERO_SNP_LABELS (:114-115)
static const signed char kGenotypeOfClass[10] = {0, 1, 2, 1, 1, 2, 2, 2, 2, 3};  // homo_reference / homo_variant / hetero_variant (:1204-1209); class 9 is always multi

// ---- the alleles of the first candidate (vcf_rows.RowPrinter._alleles = output_from :748-1008).  1: (ref, alt) filled; 0: the reads do not
// offer it (the walk: Python); -1: hand the row back
static int alleles(const c3_rows_config &cf, const Row &r, int cls, int pos, char refc, std::string &ref, std::string &alt) {
    const bool indel = cf.width == 90;
    const int cap = cf.max_len, infer = cf.infer;
    auto S = [](const Str &s) { return std::string(s.p, (size_t)s.n); };
    auto prop = [&](int len) { return (len && len < cap) ? len : 0; };
    char bases[kMaxKeys], a = 0;
    int nb = 0;
    Str out[2];
    ref.assign(1, refc);
    if (cls == 1) {
        if (pos < 0 || pos >= 4) return -1;
        const char *lab = kHomoSnp[pos];
        if (!find_alt_base(r, lab[0] != refc ? lab[0] : lab[1], bases, kMaxKeys, &nb, &a)) return -1;
        if (!a) return 0;
        alt.assign(1, a);
        return 1;
    }
    if (cls == 2) {
        if (pos < 0 || pos >= 6) return -1;
        const char *lab = kHeteroSnp[pos];
        if (lab[0] != refc && lab[1] != refc) {
            if (!find_alt_base(r, 0, bases, kMaxKeys, &nb, &a)) return -1;
            if (nb < 2) return 0;
            alt.assign(1, bases[0]), alt += ',', alt += bases[1];
            return 1;
        }
        if (!find_alt_base(r, lab[0] != refc ? lab[0] : lab[1], bases, kMaxKeys, &nb, &a)) return -1;
        if (!a) return 0;
        alt.assign(1, a);
        return 1;
    }
    const Tables &T = tables();
    if (cls == 3) {
        if (indel && (pos < 0 || pos >= kMaxLen)) return -1;
        if (!lookup(r, 'I', indel ? prop(pos + 1) : 0, infer, out)) return 0;
        alt = S(out[0]);
        return 1;
    }
    if (cls == 5) {
        if (pos < 0 || pos >= (indel ? 4 * kMaxLen : 4)) return -1;
        const char base = "ACGT"[indel ? pos % 4 : pos];
        if (!lookup(r, 'I', indel ? prop(pos / 4 + 1) : 0, infer, out)) return 0;
        if (base != refc) {
            if (!find_alt_base(r, 0, bases, kMaxKeys, &nb, &a)) return -1;
            if (nb == 0) { alt = S(out[0]); return 1; }  // :822-825: the loop ends with both alleles assigned
            alt.assign(1, bases[0]), alt += ',', alt += S(out[0]);
            return 1;
        }
        alt = S(out[0]);
        return 1;
    }
    if (cls == 6) {
        Str pair[2];
        int np = 0;
        if (indel) {
            if (pos < 0 || pos >= 136) return -1;
            Str b1[2], b2[2];
            if (lookup(r, 'I', prop(T.insins[pos][0]), infer, b1)) {
                if (indel_bases(r, 'I', prop(T.insins[pos][1]), infer, &b1[0], false, b2)) pair[0] = b1[0], pair[1] = b2[0], np = 2;
            }
        }
        if (np < 2) np = lookup_two(r, 'I', infer, pair);  // (ignore = "": an insertion key always holds its reference base)
        if (np < 2) return 0;
        if (pair[1].eq(pair[0])) alt = S(pair[0]);
        else alt = S(pair[1]) + "," + S(pair[0]);  // :869-877
        return 1;
    }
    if (cls == 4) {
        if (indel && (pos < 0 || pos >= kMaxLen)) return -1;
        if (!lookup(r, 'D', indel ? prop(pos + 1) : 0, infer, out)) return 0;
        ref += S(out[0]);
        alt.assign(1, ref[0]);
        return 1;
    }
    if (cls == 7) {
        if (pos < 0 || pos >= (indel ? 4 * kMaxLen : 4)) return -1;
        const char base = "ACGT"[indel ? pos % 4 : pos];
        if (!lookup(r, 'D', indel ? prop(pos / 4 + 1) : 0, infer, out)) return 0;
        ref += S(out[0]);
        alt.assign(1, ref[0]);
        if (base != ref[0]) alt += ',', alt += base, alt.append(ref, 1, std::string::npos);
        return 1;
    }
    if (cls == 8) {
        Str pair[2];
        int np = 0;
        if (indel) {
            if (pos < 0 || pos >= 241) return -1;
            const int l1 = T.deldel[pos][1], l2 = T.deldel[pos][0];  // sorted(entry, reverse=True)
            Str b1[2], b2[2];
            if (lookup(r, 'D', prop(l1), infer, b1)) {
                if (indel_bases(r, 'D', prop(l2), infer, &b1[0], false, b2)) {
                    if (b1[0].n > b2[0].n) pair[0] = b1[0], pair[1] = b2[0];
                    else pair[0] = b2[0], pair[1] = b1[0];
                    np = 2;
                }
            }
        }
        if (np < 2) np = lookup_two(r, 'D', infer, pair);
        if (np < 2) return 0;
        ref += S(pair[0]);  // the longer one
        const std::string a1(1, ref[0]);
        const std::string a2 = a1 + ref.substr((size_t)pair[1].n + 1 <= ref.size() ? (size_t)pair[1].n + 1 : ref.size());
        if (a1 != a2 && ref != a1 && ref != a2) alt = a1 + "," + a2;
        else alt = a1;  // :973-975
        return 1;
    }
    if (cls == 9) {
        int l1 = 0, l2 = 0;
        if (indel) {
            if (pos < 0 || pos >= 256) return -1;
            l1 = T.insdel[pos][0], l2 = T.insdel[pos][1];
        }
        Str ib[2], db[2];
        const int ni = lookup(r, 'I', prop(l2), infer, ib);
        const int nd = lookup(r, 'D', prop(l1), infer, db);
        if (!ni || !nd) return 0;
        ref += S(db[0]);
        alt.assign(1, ref[0]), alt += ',', alt += S(ib[0]), alt.append(ref, 1, std::string::npos);
        return 1;
    }
    return -1;
}

// quality_score_from (:375-381) of the float32 maximum as the reference's interpreter evaluates it: on a numpy float32 scalar the
// quotient is float32 arithmetic under numpy >= 2 (python floats are weak scalars) and double arithmetic before -- the caller says
// which rule ITS numpy follows -- then math.log in double, round(tmp, 2): the correctly rounded two-decimal value, which is exactly
// what "%.2f" of the C library prints.  Fills the printed text and the integer part (the GQ field is "%d" of the same float).
static bool qual_text(const c3_rows_config &cf, float p, char *buf, size_t cap, long long *gq, double *qual) {
    double q;
    if (cf.f32_arith) {
        const volatile float num = (1.0f - p) + (float)1e-10, den = p + (float)1e-10;
        q = (double)(float)(num / den);
    } else {
        q = ((1.0 - (double)p) + 1e-10) / ((double)p + 1e-10);
    }
    if (!(q > 0.0) || !(q < 1e300)) return false;  // (a "probability" above one: math.log raises in the reference -- the Python path's business)
    double tmp = cf.phred_trans * log(q) + 10.0;
    if (!(tmp > 0.0)) tmp = 0.0;  // max(tmp, 0)
    snprintf(buf, cap, "%.2f", tmp);
    *qual = strtod(buf, nullptr);  // float(round(tmp, 2))
    *gq = (long long)*qual;
    return true;
}

// ---- compute_PL (:1397-1454; --gvcf): the phred-scaled likelihoods of the genotypes 00 01 11 (02 12 22) from the row's 21-genotype and
// zygosity probabilities.  `ref` / `alt` are the strings output_with prints (after convert_iupac_to_n; "." for a reference call).  The
// arithmetic is the reference's: the products are float32 x float32; the sum, the division and the + 1e-8 are float32 under numpy >= 2 and
// double before (Python's sum() starts from the int 0; a numpy scalar next to a Python scalar), the logarithm is math.log of a double.
// false = hand the row back (the reference would raise: a base outside shared/utils.py:42-45, more than two alleles, an empty allele,
// probabilities that sum to zero).
static bool pl_text(const c3_rows_config &cf, const float *y, std::string ref, const std::string &alt, std::string &out) {
    if (!y) return false;
    int cut[3], ncut = 0;
    cut[ncut++] = 0;
    for (size_t i = 0; i < alt.size(); ++i)
        if (alt[i] == ',') {
            if (ncut >= 2) return false;  // genotypes[3]: KeyError
            cut[ncut++] = (int)i + 1;
        }
    if (ref.size() == 1) {  // BASE2ACGT[reference_base]
        static const char *from = "ACGTURYSWKMBDHVN", *to = "ACGTTACCAGACAAAA";
        const char *f = ref[0] ? strchr(from, ref[0]) : nullptr;
        if (!f) return false;
        ref[0] = to[f - from];
    }
    struct Label { int kind; char c; };  // 0: one base, 1: "Ins", 2: "Del"   (partial_label_from, clair3/task/gt21.py:66-71)
    Label lab[3];
    bool ok = true;
    auto label = [&](const char *p, int n) -> Label {
        if ((int)ref.size() > n) return Label{2, 0};
        if ((int)ref.size() < n) return Label{1, 0};
        if (n < 1) { ok = false; return Label{0, 0}; }
        return Label{0, p[0]};
    };
    lab[0] = label(ref.data(), (int)ref.size());
    for (int k = 0; k < ncut; ++k) lab[1 + k] = label(alt.data() + cut[k], (k + 1 < ncut ? cut[k + 1] - 1 : (int)alt.size()) - cut[k]);
    if (!ok) return false;
    auto acgt = [](char c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : -1; };
    auto gt21 = [&](const Label &a, const Label &b) -> int {  // gt21_enum_from_label(mix_two_partial_labels(a, b)) (:74-94); -1: KeyError
        if (a.kind == 0 && b.kind == 0) {
            const int i = acgt(a.c <= b.c ? a.c : b.c), j = acgt(a.c <= b.c ? b.c : a.c);
            if (i < 0 || j < 0) return -1;
            static const int first[4] = {0, 4, 7, 9};  // AA AC AG AT | CC CG CT | GG GT | TT
            return first[i] + (j - i);
        }
        if (a.kind == 0 || b.kind == 0) {
            const Label &one = a.kind == 0 ? a : b, &many = a.kind == 0 ? b : a;
            const int i = acgt(one.c);
            if (i < 0) return -1;
            return (many.kind == 2 ? 11 : 16) + i;  // ADel .. TDel = 11 .. 14, AIns .. TIns = 16 .. 19
        }
        return a.kind == b.kind ? (a.kind == 2 ? 10 : 15) : 20;  // DelDel, InsIns, InsDel
    };
    static const int G[6][2] = {{0, 0}, {0, 1}, {1, 1}, {0, 2}, {1, 2}, {2, 2}};
    const int ng = ncut == 1 ? 3 : 6;
    float L[6];
    for (int g = 0; g < ng; ++g) {
        const int idx = gt21(lab[G[g][0]], lab[G[g][1]]);
        if (idx < 0) {  // "skip N positions" (:1419-1424)
            out = "990";
            if (alt != ".")
                for (int q = 1; q < ng; ++q) out += ",990";
            return true;
        }
        const int zyg = (G[g][0] == 0 && G[g][1] == 0) ? 0 : G[g][0] == G[g][1] ? 1 : 2;  // genotype_enum_for_task(genotype_enum_from(..))
        const volatile float p = y[idx] * y[21 + zyg];
        L[g] = p;
    }
    double x[6];
    if (cf.f32_arith) {
        volatile float sum = L[0];
        for (int g = 1; g < ng; ++g) sum = sum + L[g];
        if (!(sum > 0.f) || !(sum < 3e38f)) return false;
        for (int g = 0; g < ng; ++g) {
            const volatile float q = L[g] / sum;
            const volatile float e = q + (float)1e-8;
            x[g] = (double)e;
        }
    } else {
        double sum = 0.0;
        for (int g = 0; g < ng; ++g) sum += (double)L[g];
        if (!(sum > 0.0) || !(sum < 1e300)) return false;
        for (int g = 0; g < ng; ++g) x[g] = (double)L[g] / sum + 1e-8;
    }
    const double log10_ = log(10.0);
    double pl[6], lo = 0.0;
    for (int g = 0; g < ng; ++g) {
        if (!(x[g] > 0.0)) return false;  // (math.log raises)
        pl[g] = -10.0 * (log(x[g]) / log10_);
        if (g == 0 || pl[g] < lo) lo = pl[g];
    }
    out.clear();
    char num[32];
    for (int g = 0; g < ng; ++g) {
        const double c = ceil(pl[g] - lo);
        if (!(c == c) || c > 1e15) return false;
        snprintf(num, sizeof num, g ? ",%lld" : "%lld", (long long)c);
        out += num;
    }
    return true;
}

// ---- the tail of output_with (:1176-1394 = vcf_rows.RowPrinter._row).  Appends the row to `out`; false = hand the row back
static bool row_text(const c3_rows_config &cf, const Row &r, int cls, std::string ref, std::string alt, float prob, const Str &chrom,
                     long long position, const float *y, std::string &out) {
    const bool is_ref = cls == 0;
    if ((!cf.show_reference && is_ref) || (!is_ref && ref == alt)) return true;  // prints nothing (:1176-1180)
    const bool multi = alt.find(',') != std::string::npos;
    if ((cf.haploid & 1) && (cls == 2 || cls >= 5)) return true;  // haploid, precise mode: no heterozygous call is printed (:1191-1196)
    else if ((cf.haploid & 2) && multi) return true;             // sensitive mode: none with two alternative alleles (:1197-1199)
    const char *gt = cf.gt[multi ? 3 : kGenotypeOfClass[cls]];
    if (cf.haploid) gt = strchr(gt, '1') ? "1" : "0";  // :1327-1329
    // decode_alt_info (:1215-1230)
    long long snp[128];
    bool has_snp[128] = {false};
    Str ik[kMaxKeys], dk[kMaxKeys];
    long long ic[kMaxKeys], dc[kMaxKeys];
    int ni = 0, nd = 0;
    long long ref_count = 0;
    for (int k = 0; k < r.n; ++k) {
        const Str &key = r.e[k].key;
        const char t = key.p[0];
        if (t == 'X') {
            if (key.n < 2) return false;
            snp[(int)key.p[1]] = r.e[k].count, has_snp[(int)key.p[1]] = true;
        } else if (t == 'I') {
            ik[ni] = Str{key.p + 1, key.n - 1}, ic[ni] = r.e[k].count, ++ni;
        } else if (t == 'D') {
            dk[nd] = Str{key.p + 1, key.n - 1}, dc[nd] = r.e[k].count, ++nd;
        } else if (t == 'R') {
            ref_count = r.e[k].count;
        }
    }
    if (ref_count < 0) ref_count = 0;
    auto ins_of = [&](const char *p, int n) -> long long {
        const Str s{p, n};
        for (int i = 0; i < ni; ++i)
            if (ik[i].eq(s)) return ic[i];
        return 0;
    };
    auto del_of = [&](const char *p, int n) -> long long {
        const Str s{p, n};
        for (int i = 0; i < nd; ++i)
            if (dk[i].eq(s)) return dc[i];
        return 0;
    };
    auto del_of_len = [&](int len) -> long long {  // [deld[k] for k in deld if len(k) == n_del][0]
        for (int i = 0; i < nd; ++i)
            if (dk[i].n == len) return dc[i];
        return 0;
    };
    // --enable_long_indel: the reads of the OTHER insertion alleles whose length lies within long_prop of a long proposed allele's count with
    // it (get_long_indel_read_count, :383-402; lengths include the reference base, the proposal's own does not: :393).  The reference calls
    // the same function for deletions WITHOUT is_del (:1273-1275, :1282-1283, :1294-1297, :1313-1315): the proposal's length is then
    // len("") - 1, the window [50, -1.1] holds nothing, and a deletion's count stays what it is -- restated as it runs, not as it reads.
    auto long_ins = [&](const char *p, int n) -> long long {
        if (!cf.long_indel || !(n > cf.long_infer)) return 0;
        const double len = (double)(n - 1);
        const double lo = std::max(len * (1.0 - cf.long_prop), (double)cf.long_infer), hi = len * (1.0 + cf.long_prop);
        const Str s{p, n};
        long long sum = 0;
        for (int i = 0; i < ni; ++i)
            if (!ik[i].eq(s) && (double)ik[i].n >= lo && (double)ik[i].n <= hi) sum += ic[i];
        return sum;
    };
    long long supported = 0, counts[8];
    int nc = 0;
    // alt.split(","): the pieces
    int cut[4], ncut = 0;
    cut[ncut++] = 0;
    for (size_t i = 0; i < alt.size(); ++i)
        if (alt[i] == ',') {
            if (ncut >= 3) return false;  // more than two alleles: not a row of this path
            cut[ncut++] = (int)i + 1;
        }
    auto piece = [&](int k, const char **p, int *n) {
        *p = alt.data() + cut[k];
        *n = (k + 1 < ncut ? cut[k + 1] - 1 : (int)alt.size()) - cut[k];
    };
    const char *pp;
    int pn;
    if (is_ref) {
        supported = ref_count, alt = ".";
    } else if (cls <= 2) {
        for (size_t i = 0; i < alt.size(); ++i) {
            if (alt[i] == ',') continue;
            const long long n = has_snp[(int)alt[i]] ? snp[(int)alt[i]] : 0;
            supported += n, counts[nc++] = n;
        }
    } else if (cls == 3 || cls == 6) {
        for (int k = 0; k < ncut; ++k) {
            piece(k, &pp, &pn);
            const long long n = ins_of(pp, pn) + long_ins(pp, pn);
            supported += n, counts[nc++] = n;
        }
    } else if (cls == 5) {
        long long n_snp = 0;
        if (multi) {
            piece(0, &pp, &pn);
            if (pn < 1) return false;
            n_snp = has_snp[(int)pp[0]] ? snp[(int)pp[0]] : 0;
            piece(1, &pp, &pn);
        } else {
            piece(0, &pp, &pn);
        }
        const long long n_ins = ins_of(pp, pn) + long_ins(pp, pn);
        supported = n_ins + n_snp;
        if (multi) counts[nc++] = n_snp;
        counts[nc++] = n_ins;
    } else if (cls == 4 || cls == 8) {
        if (nd > 0) {
            if (cls == 4) {
                if (cf.long_indel && ref.size() <= 1) return false;  // (len(None): the reference raises)
                supported = ref.size() > 1 ? del_of(ref.data() + 1, (int)ref.size() - 1) : 0;
                counts[nc++] = supported;
            } else if (nd > 1) {
                for (int k = 0; k < ncut; ++k) {
                    piece(k, &pp, &pn);
                    const long long n = del_of_len((int)ref.size() - pn);
                    counts[nc++] = n, supported += n;
                }
            }
        }
    } else if (cls == 7) {
        long long n_snp = 0;
        bool has_base = false;
        if (multi && ncut > 1) {
            piece(1, &pp, &pn);
            if (pn < 1) return false;
            has_base = true, n_snp = has_snp[(int)pp[0]] ? snp[(int)pp[0]] : 0;
        }
        if (cf.long_indel && ref.size() <= 1) return false;  // (len(None): the reference raises)
        const long long n_del = ref.size() > 1 ? del_of(ref.data() + 1, (int)ref.size() - 1) : 0;
        supported = n_del + n_snp;
        if (has_base) counts[nc++] = n_snp;
        counts[nc++] = n_del;
    } else {  // insertion and deletion (:1306-1322)
        for (int k = 0; k < ncut; ++k) {
            piece(k, &pp, &pn);
            const int n_del = (int)ref.size() - pn;
            long long n;
            if (n_del < 0) {
                const int in = ref.size() > 1 ? pn - ((int)ref.size() - 1) : pn;
                n = ins_of(pp, in) + long_ins(pp, in);
            }
            else n = del_of_len(n_del);
            counts[nc++] = n, supported += n;
        }
    }
    double af = (double)supported / (double)r.depth;  // (depth != 0: checked by the caller)
    if (af > 1) af = 1;
    char qbuf[64];
    long long gq;
    double qual;
    if (!qual_text(cf, prob, qbuf, sizeof qbuf, &gq, &qual)) return false;
    const char *filt = is_ref ? "RefCall" : (!cf.has_qs_pass || qual >= cf.qs_pass) ? "PASS" : "LowQual";  // filtration_value_from (:100-114)
    if (!cf.keep_iupac) {  // convert_iupac_to_n (shared/utils.py:27-40): what is not A C G T N , . (either case) becomes N; "." stays
        auto conv = [](std::string &s) {
            if (s == ".") return;
            for (char &c : s) {
                const char u = (c >= 'a' && c <= 'z') ? (char)(c - 32) : c;
                if (!(u == 'A' || u == 'C' || u == 'G' || u == 'T' || u == 'N' || u == ',' || u == '.')) c = 'N';
            }
        };
        conv(ref), conv(alt);
    }
    std::string pls;
    if (cf.gvcf && !pl_text(cf, y, ref, alt, pls)) return false;  // :1360-1363
    char num[96];
    out.append(chrom.p, (size_t)chrom.n);
    snprintf(num, sizeof num, "\t%lld\t.\t", position);
    out += num;
    out += ref, out += '\t', out += alt, out += '\t', out += qbuf, out += '\t', out += filt, out += '\t';
    out += cf.pileup ? 'P' : 'F';
    out += cf.gvcf ? "\tGT:GQ:DP:AD:AF:PL\t" : "\tGT:GQ:DP:AD:AF\t";
    out += gt;
    snprintf(num, sizeof num, ":%lld:%lld:%lld", gq, r.depth, ref_count);
    out += num;
    for (int i = 0; i < nc; ++i) snprintf(num, sizeof num, ",%lld", counts[i]), out += num;
    out += ':';
    if (nc <= 1) {
        snprintf(num, sizeof num, "%.4f", af), out += num;
    } else {
        for (int i = 0; i < nc; ++i) {
            double f = 1.0 * (double)counts[i] / (double)r.depth;
            if (f > 1.0) f = 1.0;
            snprintf(num, sizeof num, i ? ",%.4f" : "%.4f", f), out += num;
        }
    }
    if (cf.gvcf) out += ':', out += pls;
    out += '\n';
    return true;
}

// ---- the nine probability lists of a row laid end to end IN THE ORDER THE LOOP'S if / elif CHAIN AND .index() BREAK TIES (classes 1, 2,
// 3, 5, 6, 4, 7, 8, 9; entries by index): vcf_rows.class_lists_of_rows()[_CHAIN] = clair3/CallVariants.py:526-659.  The same float32
// products in the same order and association (only multiplications: nothing for the compiler to contract), so the values are the
// reference's bit for bit (tests/test_rows_c.py compares them with the Python form).
struct Cand {
    float v;
    short cls, idx;
};
static int class_lists_chain(const float *y, bool indel, Cand *out) {
    const float *g = y, *z = y + 21;
    const float hv = z[1], ht = z[2];
    static const int HS[4] = {0, 4, 7, 9}, TS[6] = {1, 2, 3, 5, 6, 8};
    int n = 0;
    auto put = [&](float v, int cls, int idx) { out[n].v = v, out[n].cls = (short)cls, out[n].idx = (short)idx, ++n; };
    if (!indel) {  // :526-566: every entry is ONE product zygosity x gt21
        for (int i = 0; i < 4; ++i) put(hv * g[HS[i]], 1, i);
        for (int i = 0; i < 6; ++i) put(ht * g[TS[i]], 2, i);
        put(hv * g[15], 3, 0);
        for (int b = 0; b < 4; ++b) put(ht * g[16 + b], 5, b);
        put(ht * g[15], 6, 0);
        put(hv * g[10], 4, 0);
        for (int b = 0; b < 4; ++b) put(ht * g[11 + b], 7, b);
        put(ht * g[10], 8, 0);
        put(ht * g[20], 9, 0);
        return n;
    }
    const float *p1 = y + 24, *p2 = y + 57;
    constexpr int o = 16;  // VariantLength.index_offset
    const Tables &T = tables();
    const float v0 = p1[o] * p2[o];
    const float v0hv = v0 * hv, v0ht = v0 * ht;
    for (int i = 0; i < 4; ++i) put(v0hv * g[HS[i]], 1, i);                                                   // :579-581
    for (int i = 0; i < 6; ++i) put(v0ht * g[TS[i]], 2, i);                                                   // :582-584
    const float hv15 = hv * g[15], hv10 = hv * g[10], ht15 = ht * g[15], ht10 = ht * g[10], ht20 = ht * g[20];
    for (int L = 1; L <= kMaxLen; ++L) put((p1[o + L] * p2[o + L]) * hv15, 3, L - 1);                          // :303-308, :587-590
    for (int L = 1; L <= kMaxLen; ++L)
        for (int b = 0; b < 4; ++b) put(((p1[o] * p2[o + L]) * g[16 + b]) * ht, 5, 4 * (L - 1) + b);          // :311-316, :600-606
    for (int k = 0; k < 136; ++k) put((p1[o + T.insins[k][0]] * p2[o + T.insins[k][1]]) * ht15, 6, k);        // :318-328
    for (int L = 1; L <= kMaxLen; ++L) put((p1[o - L] * p2[o - L]) * hv10, 4, L - 1);                          // :331-336, :613-616
    for (int L = 1; L <= kMaxLen; ++L)
        for (int b = 0; b < 4; ++b) put(((p1[o - L] * p2[o]) * g[11 + b]) * ht, 7, 4 * (L - 1) + b);          // :339-345, :627-633
    {
        int k = 0;  // enumeration order (i outer), the pair BEFORE the (min, max) swap of the entries table
        for (int i = 1; i <= kMaxLen; ++i)
            for (int j = 1; j <= kMaxLen; ++j) {
                if (i == j && i != kMaxLen) continue;
                put((p1[o - i] * p2[o - j]) * ht10, 8, k++);                                                  // :348-359
            }
    }
    for (int k = 0; k < 256; ++k) put((p1[o - T.insdel[k][0]] * p2[o + T.insdel[k][1]]) * ht20, 9, k);        // :362-371
    return n;
}

// ---- the passes of output_from's loop after its first candidate was rejected (vcf_rows.RowPrinter._next_candidate = :721-1016): the
// entries above the homo-reference probability by falling probability, ties in chain order, each looked up until the reads offer one.
// 1: a row was printed (or the reference prints nothing); 0: hand the row back (a shared maximum, something odd)
static int walk(const c3_rows_config &cf, const Row &r, const float *y, const float *cols, int bi, int cls0, int pos0, char refc, const Str &chrom,
                long long position, std::string &ref, std::string &alt, std::string &text) {
    static thread_local Cand cand_tls[804];
    static thread_local short keep_tls[804];
    Cand *const cand = cand_tls;
    short *const keep = keep_tls;
    const bool indel = cf.width == 90;
    const int n = class_lists_chain(y, indel, cand);
    const float homo = cols[9 + bi];
    const char acgt = "ACGT"[bi];
    // the head of the walk -- the first of the largest entries above the homo-reference probability -- must be the device's first decision
    int head = -1;
    for (int i = 0; i < n; ++i) {
        if (!(cand[i].v == cand[i].v)) return 0;
        if (cand[i].v > homo && (head < 0 || cand[i].v > cand[head].v)) head = i;
    }
    if (head < 0 || cand[head].cls != cls0 || cand[head].idx != pos0) return 0;
    // Entries no lookup can satisfy are stepped over without asking (vcf_rows.RowPrinter._dead, from the answers themselves here): a
    // homo insertion / deletion and an ACGT + insertion / deletion are rejected exactly when the lookup for their length comes back
    // empty, an insertion-and-deletion when either does, two insertions / two deletions whose FIRST proposal comes back empty exactly
    // when return_multi offers fewer than two, a SNP when the reads hold no (or, for two new bases, fewer than two) SNP alleles.
    int n_x = 0;
    for (int k = 0; k < r.n; ++k)
        if (r.e[k].key.p[0] == 'X') {
            if (r.e[k].key.n < 2) return 0;  // (find_alt_base would raise)
            ++n_x;
        }
    bool ins_dead[kMaxLen + 1], del_dead[kMaxLen + 1];
    Str tmp[2];
    for (int L = 0; L <= (indel ? kMaxLen - 1 : 0); ++L) ins_dead[L] = lookup(r, 'I', L, cf.infer, tmp) == 0, del_dead[L] = lookup(r, 'D', L, cf.infer, tmp) == 0;
    const bool two_ins_short = lookup_two(r, 'I', cf.infer, tmp) < 2, two_del_short = lookup_two(r, 'D', cf.infer, tmp) < 2;
    const Tables &T = tables();
    auto prop = [&](int len) { return (indel && len && len < cf.max_len) ? len : 0; };
    auto dead = [&](const Cand &c) -> bool {
        switch (c.cls) {
            case 1: return n_x == 0;
            case 2: return (kHeteroSnp[c.idx][0] != refc && kHeteroSnp[c.idx][1] != refc) ? n_x < 2 : n_x == 0;
            case 3: return ins_dead[prop(c.idx + 1)];
            case 4: return del_dead[prop(c.idx + 1)];
            case 5: return ins_dead[prop(c.idx / 4 + 1)];
            case 7: return del_dead[prop(c.idx / 4 + 1)];
            case 9: return indel ? (ins_dead[prop(T.insdel[c.idx][1])] || del_dead[prop(T.insdel[c.idx][0])]) : (ins_dead[0] || del_dead[0]);
            case 6: return (!indel || ins_dead[prop(T.insins[c.idx][0])]) && two_ins_short;
            case 8: return (!indel || del_dead[prop(T.deldel[c.idx][1])]) && two_del_short;
        }
        return false;
    };
    int m = 0;
    for (int i = 0; i < n; ++i)
        if (i != head && cand[i].v > homo && !dead(cand[i])) keep[m++] = (short)i;
    // The walk of the loop = the chain-ordered entries above the homo-reference probability by falling probability, ties in chain order
    // (a total order: probability falling, then chain position), produced in growing tiers -- most walks end within a few candidates.
    const auto before = [&](short a, short b) { return cand[a].v > cand[b].v || (cand[a].v == cand[b].v && a < b); };
    int done = 0;
    for (int tier = 0; tier < 3 && done < m; ++tier) {
        const int hi = tier == 0 ? std::min(m, 16) : tier == 1 ? std::min(m, 80) : m;
        if (hi == m) std::sort(keep + done, keep + m, before);
        else std::partial_sort(keep + done, keep + hi, keep + m, before);
        for (int j = done; j < hi; ++j) {
            const int at = keep[j];
            const int got = alleles(cf, r, cand[at].cls, cand[at].idx, refc, ref, alt);
            if (got < 0) return 0;
            if (got == 0) continue;
            const Cand &c = cand[at];  // accepted; LATER entries of the walk with the same probability: flags of other classes (:742-750)
            for (int i = at + 1; i < n; ++i)
                if (cand[i].v == c.v && cand[i].cls != c.cls) return 0;
            return row_text(cf, r, c.cls, ref, alt, c.v, chrom, position, y, text) ? 1 : 0;
        }
        done = hi;
    }
    // nothing above the homo-reference probability is offered by the reads (:735-740)
    return row_text(cf, r, 0, std::string(1, acgt), std::string(1, acgt), homo, chrom, position, y, text) ? 1 : 0;
}

// reference base -> A/C/G/T index as output_from resolves it (clair3_amd/decode.py _REF_BASE_INDEX = shared/utils.py:42-45)
static int ref_base_index(char c) {
    static const char *from = "ACGTURYSWKMBDHVN", *to = "ACGTTACCAGACAAAA";
    for (int i = 0; from[i]; ++i)
        if (from[i] == c) return (int)(strchr("ACGT", to[i]) - "ACGT");
    return -1;
}

}  // namespace c3rows

extern "C" int c3_vcf_rows(const c3_rows_config *cfg, int64_t n, const char *pos_text, int64_t pos_bytes, const char *alt_text, int64_t alt_bytes,
                           const float *rows, int64_t row_stride_floats, char *out, int64_t out_cap, int64_t *out_off, uint8_t *status) {
    using namespace c3rows;
    if (!cfg || n < 0 || !pos_text || !alt_text || !rows || !out || !out_off || !status) return fail("c3_vcf_rows: null argument");
    if (cfg->width != 24 && cfg->width != 90) return fail("c3_vcf_rows: width must be 24 or 90");
    if (cfg->max_len != kMaxLen) return fail("c3_vcf_rows: VariantLength.max is %d here, the caller's is %d", kMaxLen, cfg->max_len);
    if (row_stride_floats < cfg->width + C3_DECODE_COLS) return fail("c3_vcf_rows: rows carry no decoder columns");
    const c3_rows_config &cf = *cfg;
    std::string text, ref, alt;
    text.reserve((size_t)n * 96);
    static thread_local Row r;
    const char *pp = pos_text, *pe = pos_text + pos_bytes, *ap = alt_text, *ae = alt_text + alt_bytes;
    for (int64_t i = 0; i < n; ++i) {
        // the i-th NUL-separated text of either list
        if (pp > pe || ap > ae) return fail("c3_vcf_rows: fewer texts than rows");
        const char *pz = (const char *)memchr(pp, 0, (size_t)(pe - pp));
        const char *az = (const char *)memchr(ap, 0, (size_t)(ae - ap));
        const int pn = (int)((pz ? pz : pe) - pp), an = (int)((az ? az : ae) - ap);
        const char *ps = pp, *as = ap;
        pp = (pz ? pz : pe) + 1, ap = (az ? az : ae) + 1;
        out_off[i] = (int64_t)text.size();
        status[i] = 1;
        // "chr:pos:seq" (:1127-1143): the last two fields are position and sequence, a contig name may hold colons itself
        const int pl = rstrip(ps, pn);
        if (!plain_ascii(ps, pl) || !plain_ascii(as, rstrip(as, an))) continue;  // (trailing white space is stripped by the reference too)
        int c2 = pl - 1;
        while (c2 >= 0 && ps[c2] != ':') --c2;
        int c1 = c2 - 1;
        while (c1 >= 0 && ps[c1] != ':') --c1;
        if (c2 < 0 || c1 < 0) continue;
        const Str chrom{ps, c1}, seq{ps + c2 + 1, pl - c2 - 1};
        long long position;
        if (!plain_int(ps + c1 + 1, c2 - c1 - 1, &position)) continue;
        const int at = seq.n > 1 ? cf.flank : 0;
        if (at >= seq.n) continue;
        const char refc = seq.p[at];
        const int bi = ref_base_index(refc);
        if (bi < 0) continue;
        const float *y = rows + i * row_stride_floats, *cols = y + cf.width;
        const int cls = (int)cols[23 + bi];
        if (cls < 0 || cls > 9) continue;
        const int k = (cls > 0 ? cls : 1) - 1;
        const int pos = (int)cols[13 + k];
        const float prob = cls > 0 ? cols[k] : cols[9 + bi];
        if (!(prob == prob)) continue;
        if (cls > 0) {  // a maximum two classes share: output_with's flag chains are not output_from's
            int same = 0;
            for (int q = 0; q < 9; ++q) same += cols[q] == cols[k];
            if (same > 1) continue;
        }
        if (!parse_alt(as, an, r) || r.depth == 0) continue;
        bool bad_key = false;
        for (int q = 0; q < r.n; ++q) bad_key |= r.e[q].key.n < 1;
        if (bad_key) continue;
        const char acgt = "ACGT"[bi];
        const size_t mark = text.size();
        if (cls == 0) {
            if (!row_text(cf, r, 0, std::string(1, acgt), std::string(1, acgt), prob, chrom, position, y, text)) { text.resize(mark); continue; }
            status[i] = 0;
            continue;
        }
        const int got = alleles(cf, r, cls, pos, refc, ref, alt);
        if (got < 0) continue;  // something odd: Python
        if (got == 0) {  // the reads do not offer the first candidate: the walk over the class lists
            if (!cf.walk || walk(cf, r, y, cols, bi, cls, pos, refc, chrom, position, ref, alt, text) != 1) { text.resize(mark); continue; }
            status[i] = 2;
            continue;
        }
        if (!row_text(cf, r, cls, ref, alt, prob, chrom, position, y, text)) { text.resize(mark); continue; }
        status[i] = 0;
    }
    out_off[n] = (int64_t)text.size();
    if ((int64_t)text.size() > out_cap) return fail("c3_vcf_rows: %zu bytes of text for a buffer of %lld", text.size(), (long long)out_cap);
    memcpy(out, text.data(), text.size());
    return 0;
}

// tools/wino_probe.hip -- conv3x3_wino_planes_kernel (Winograd F(2,3) along H, c3_conv3w.h) against conv3x3_planes_kernel (direct,
// c3_conv3.h) on the residual-block shapes of a full-alignment batch (run on the GPU box):
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -fno-gpu-flush-denormals-to-zero -I clair3_amd/csrc tools/wino_probe.hip -o /tmp/wino_probe && /tmp/wino_probe
// 1. correctness: both kernels on the same random planes / weights / residual; max |difference| between them and of each against
//    an fp64 host evaluation of sampled outputs (small batches with ragged tails, then B = 256);
// 2. time: direct whole, Winograd whole and with parts switched off (ABL bits: 1 no weight loads, 2 no transform after the first,
//    4 no epilogue, 8 no matrix instructions, 16 transform loads not requested ahead), one and two workgroups per CU.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include "c3_conv3w16.h"
using namespace c3;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <class K>
static float time_it(K launch, int reps = 20) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    launch(); launch(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < reps; ++i) launch(); hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b); return ms / reps * 1e3f;
}
static uint32_t magic(int d) { return d == 1 ? 0u : (uint32_t)((1ull << 32) / (uint64_t)d + 1); }
static void scales(const std::vector<float> &w, int rows, std::vector<float> &sc, std::vector<float> &post) {  // c3_pack.h row_scales
    const size_t cols = w.size() / rows;
    sc.assign(rows, 1.f), post.assign(rows, 1.f);
    for (int r = 0; r < rows; ++r) {
        float mx = 0.f;
        for (size_t i = 0; i < cols; ++i) mx = std::max(mx, std::fabs(w[r * cols + i]));
        if (!(mx > 0.f)) continue;
        int e; (void)std::frexp(mx, &e);
        const int k = 13 - e;
        sc[r] = std::ldexp(1.f, k), post[r] = std::ldexp(1.f, -k);
    }
}
static void put16(uint16_t *q, size_t dst, float v) {
    const _Float16 h0 = (_Float16)v, h1 = (_Float16)(v - (float)h0);
    memcpy(&q[dst], &h0, 2), memcpy(&q[dst + 64 * 8], &h1, 2);
}

template <int C, bool RES, int ABL> static void launch_w(const WinoConvParams &wp, int grid) {
    hipLaunchKernelGGL((conv3x3_wino_planes_kernel<C, RES, ABL>), dim3(grid), dim3(kPlThreads), 0, 0, wp);
}

template <int C, bool RES, int ABL, int RT = 2, int WD = 4> static void launch_w16(const WinoConvParams &wp, int grid) {
    hipLaunchKernelGGL((conv3x3_wino16_planes_kernel<C, RES, ABL, RT, WD>), dim3(grid), dim3(kPlThreads), 0, 0, wp);
}

template <int C> static int shape(const char *name, int B, int H, int W, int cus, bool timing, bool split_only = false) {
    constexpr int NS = C / 64, NS32 = C / 32;
    const int M = B * H * W, Hj = (H + 1) / 2, Mp = B * Hj * W;
    const size_t bytes = (size_t)M * C * 4;
    std::mt19937 rng(1234 + C + B);
    std::uniform_real_distribution<float> U(-1.f, 1.f);
    // activations and residual: post-ReLU-like values as genuine hi / lo pairs
    std::vector<float> xf((size_t)M * C), rf((size_t)M * C);
    std::vector<_Float16> xp(bytes / 2), rp(bytes / 2);
    for (size_t m = 0; m < (size_t)M; ++m)
        for (int c = 0; c < C; ++c) {
            float xv = U(rng), rv = U(rng);
            xv = xv < -0.3f ? 0.f : (xv + 0.3f) * 2.5f, rv = rv < -0.3f ? 0.f : (rv + 0.3f) * 2.5f;
            const _Float16 xh = (_Float16)xv, xl = (_Float16)(xv - (float)xh), rh = (_Float16)rv, rl = (_Float16)(rv - (float)rh);
            xf[m * C + c] = (float)xh + (float)xl, rf[m * C + c] = (float)rh + (float)rl;
            const size_t o = m * 2 * C + (c >> 6) * 128 + (c & 63);
            xp[o] = xh, xp[o + 64] = xl, rp[o] = rh, rp[o + 64] = rl;
        }
    // weights g[co][kh*3+kw][ci], bias
    std::vector<float> g((size_t)C * 9 * C), bias(C);
    const float wscale = 1.f / std::sqrt(9.f * C);
    for (auto &v : g) v = U(rng) * wscale * (U(rng) > 0.9f ? 4.f : 1.f);
    for (auto &v : bias) v = U(rng) * 0.5f;
    // direct fragments (c3_pack.h)
    std::vector<float> sc, post;
    scales(g, C, sc, post);
    std::vector<float> pk((size_t)NS * NS * 9 * 64 * 64);
    uint16_t *q16 = reinterpret_cast<uint16_t *>(pk.data());
    for (int tn = 0; tn < NS; ++tn) for (int slab = 0; slab < NS; ++slab) for (int tap = 0; tap < 9; ++tap) for (int wn = 0; wn < 2; ++wn)
        for (int ks = 0; ks < 4; ++ks) for (int lane = 0; lane < 64; ++lane) for (int j = 0; j < 8; ++j) {
            const int co = tn * 64 + 32 * wn + (lane & 31), ci = slab * 64 + 8 * (2 * ks + (lane >> 5)) + j;
            put16(q16, (((((((size_t)tn * NS + slab) * 9 + tap) * 2 + wn) * 4 + ks) * 2) * 64 + lane) * 8 + j, g[((size_t)co * 9 + tap) * C + ci] * sc[co]);
        }
    // Winograd weights U[co][xi*3+kw][ci] (double, rounded once), their scales, fragments
    std::vector<float> u((size_t)C * 12 * C);
    for (int co = 0; co < C; ++co) for (int kw = 0; kw < 3; ++kw) for (int ci = 0; ci < C; ++ci) {
        const double g0 = g[((size_t)co * 9 + 0 + kw) * C + ci], g1 = g[((size_t)co * 9 + 3 + kw) * C + ci], g2 = g[((size_t)co * 9 + 6 + kw) * C + ci];
        const double uu[4] = {g0, 0.5 * (g0 + g1 + g2), 0.5 * (g0 - g1 + g2), g2};
        for (int xi = 0; xi < 4; ++xi) u[((size_t)co * 12 + xi * 3 + kw) * C + ci] = (float)uu[xi];
    }
    std::vector<float> scw, postw;
    scales(u, C, scw, postw);
    std::vector<float> pw((size_t)NS * NS32 * 12 * 2048);  // 8 KB per chunk
    uint16_t *w16 = reinterpret_cast<uint16_t *>(pw.data());
    for (int tn = 0; tn < NS; ++tn) for (int s32 = 0; s32 < NS32; ++s32) for (int tap = 0; tap < 12; ++tap) for (int wn = 0; wn < 2; ++wn)
        for (int ks = 0; ks < 2; ++ks) for (int lane = 0; lane < 64; ++lane) for (int j = 0; j < 8; ++j) {
            const int co = tn * 64 + 32 * wn + (lane & 31), ci = s32 * 32 + 8 * (2 * ks + (lane >> 5)) + j;
            put16(w16, (((((((size_t)tn * NS32 + s32) * 12 + tap) * 2 + wn) * 2 + ks) * 2) * 64 + lane) * 8 + j, u[((size_t)co * 12 + tap) * C + ci] * scw[co]);
        }

    constexpr int NS16 = C / 16;
    std::vector<float> pw16v((size_t)NS * NS16 * 12 * 1024);  // 4 KB per chunk: [tn][s16][tap][wn][piece][lane] x 16 B
    uint16_t *w16b = reinterpret_cast<uint16_t *>(pw16v.data());
    for (int tn = 0; tn < NS; ++tn) for (int s16 = 0; s16 < NS16; ++s16) for (int tap = 0; tap < 12; ++tap) for (int wn = 0; wn < 2; ++wn)
        for (int lane = 0; lane < 64; ++lane) for (int j = 0; j < 8; ++j) {
            const int co = tn * 64 + 32 * wn + (lane & 31), ci = s16 * 16 + 8 * (lane >> 5) + j;
            put16(w16b, ((((((size_t)tn * NS16 + s16) * 12 + tap) * 2 + wn) * 2) * 64 + lane) * 8 + j, u[((size_t)co * 12 + tap) * C + ci] * scw[co]);
        }
    void *wfw16; CK(hipMalloc(&wfw16, pw16v.size() * 4)); CK(hipMemcpy(wfw16, pw16v.data(), pw16v.size() * 4, hipMemcpyHostToDevice));
    void *x, *yd, *yw, *r, *wfd, *wfw; float *dbias, *dpost, *dpostw, *fd, *fw; uint32_t *flag;
    CK(hipMalloc(&x, bytes)); CK(hipMalloc(&yd, bytes)); CK(hipMalloc(&yw, bytes)); CK(hipMalloc(&r, bytes));
    CK(hipMalloc(&wfd, pk.size() * 4)); CK(hipMalloc(&wfw, pw.size() * 4));
    CK(hipMalloc(&dbias, C * 4)); CK(hipMalloc(&dpost, C * 4)); CK(hipMalloc(&dpostw, C * 4)); CK(hipMalloc(&flag, 256));
    CK(hipMalloc(&fd, bytes)); CK(hipMalloc(&fw, bytes));
    CK(hipMemcpy(x, xp.data(), bytes, hipMemcpyHostToDevice)); CK(hipMemcpy(r, rp.data(), bytes, hipMemcpyHostToDevice));
    CK(hipMemcpy(wfd, pk.data(), pk.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(wfw, pw.data(), pw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dbias, bias.data(), C * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dpost, post.data(), C * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dpostw, postw.data(), C * 4, hipMemcpyHostToDevice)); CK(hipMemset(flag, 0, 256));
    CK(hipMemset(yd, 0xff, bytes)); CK(hipMemset(yw, 0xff, bytes));

    PlaneConvParams cp;
    cp.x = x, cp.wf = wfd, cp.bias = dbias, cp.res = r, cp.out = yd, cp.range_flag = flag, cp.post = dpost, cp.pre = dpost;
    cp.M = M, cp.H = H, cp.W = W, cp.mg_hw = magic(H * W), cp.mg_w = magic(W);
    const int tiles_m = (M + kPlBM - 1) / kPlBM;
    cp.tiles = tiles_m * NS;
    WinoConvParams wp;
    wp.x = x, wp.wf = wfw, wp.bias = dbias, wp.post = dpostw, wp.res = r, wp.out = yw, wp.range_flag = flag;
    wp.M = M, wp.Mp = Mp, wp.H = H, wp.W = W, wp.Hj = Hj, wp.mg_hjw = magic(Hj * W), wp.mg_w = magic(W);
    const int tiles_w = (Mp + kWTM - 1) / kWTM;
    wp.tiles = tiles_w * NS;
    const int unit = 8 * NS;
    auto grid_for = [&](int tiles, int s) { return tiles <= s ? tiles : s / unit * unit; };
    const int gd = grid_for(cp.tiles, 2 * cus), gw = grid_for(wp.tiles, 2 * cus);
    hipLaunchKernelGGL((conv3x3_planes_kernel<C, true>), dim3(gd), dim3(kPlThreads), 0, 0, cp);
    CK(hipGetLastError());
    launch_w<C, true, 0>(wp, gw);
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(planes_to_f32_kernel, dim3((unsigned)(((size_t)M * C + 255) / 256)), dim3(256), 0, 0, yd, fd, (int64_t)M, C);
    hipLaunchKernelGGL(planes_to_f32_kernel, dim3((unsigned)(((size_t)M * C + 255) / 256)), dim3(256), 0, 0, yw, fw, (int64_t)M, C);
    CK(hipDeviceSynchronize());
    std::vector<float> hd((size_t)M * C), hw((size_t)M * C);
    CK(hipMemcpy(hd.data(), fd, bytes, hipMemcpyDeviceToHost)); CK(hipMemcpy(hw.data(), fw, bytes, hipMemcpyDeviceToHost));
    double dmax = 0, ymax = 0; size_t worst = 0, nan_w = 0;
    for (size_t i = 0; i < hd.size(); ++i) {
        if (!(hw[i] == hw[i])) { ++nan_w; continue; }
        const double d = std::fabs((double)hd[i] - hw[i]);
        if (d > dmax) dmax = d, worst = i;
        ymax = std::max(ymax, (double)std::fabs(hd[i]));
    }
    // fp64 host evaluation of sampled outputs
    double ed = 0, ew = 0;
    std::uniform_int_distribution<int> pm(0, M - 1), pc(0, C - 1);
    const int samples = 3000;
    for (int s = 0; s < samples + 4; ++s) {
        int m = pm(rng), co = pc(rng);
        if (s == samples) m = 0; if (s == samples + 1) m = M - 1; if (s == samples + 2) m = W - 1; if (s == samples + 3) m = (H - 1) * W;
        const int b = m / (H * W), oh = (m / W) % H, ow = m % W;
        double a = bias[co];
        for (int kh = 0; kh < 3; ++kh) for (int kw = 0; kw < 3; ++kw) {
            const int ih = oh + kh - 1, iw = ow + kw - 1;
            if (ih < 0 || ih >= H || iw < 0 || iw >= W) continue;
            const float *xr = &xf[((size_t)(b * H + ih) * W + iw) * C], *gr = &g[((size_t)co * 9 + kh * 3 + kw) * C];
            for (int ci = 0; ci < C; ++ci) a += (double)xr[ci] * gr[ci];
        }
        a += rf[(size_t)m * C + co];
        a = a > 0 ? a : 0;
        ed = std::max(ed, std::fabs(a - hd[(size_t)m * C + co])), ew = std::max(ew, std::fabs(a - hw[(size_t)m * C + co]));
    }
    size_t mixdiff = 0;
    {   // the transform on plain conversions instead of v_fma_mix_f32: the same planes bit for bit
        void *yw2; CK(hipMalloc(&yw2, bytes)); CK(hipMemset(yw2, 0xff, bytes));
        WinoConvParams w2 = wp; w2.out = yw2;
        hipLaunchKernelGGL((conv3x3_wino_planes_kernel<C, true, 0, false>), dim3(gw), dim3(kPlThreads), 0, 0, w2);
        CK(hipGetLastError()); CK(hipDeviceSynchronize());
        std::vector<uint32_t> a(bytes / 4), b2(bytes / 4);
        CK(hipMemcpy(a.data(), yw, bytes, hipMemcpyDeviceToHost)); CK(hipMemcpy(b2.data(), yw2, bytes, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < a.size(); ++i) mixdiff += a[i] != b2[i];
        hipFree(yw2);
    }
    size_t w16diff = 0, s16diff = 0; double w16max = 0, s16max = 0;
    {   // the LDS-DMA / 16-channel-slab form: the same V values, the same products per accumulator in the same channel order
        void *yw3; CK(hipMalloc(&yw3, bytes)); CK(hipMemset(yw3, 0xff, bytes));
        WinoConvParams w3 = wp; w3.out = yw3; w3.wf = wfw16;
        launch_w16<C, true, 0>(w3, gw);
        CK(hipGetLastError()); CK(hipDeviceSynchronize());
        hipLaunchKernelGGL(planes_to_f32_kernel, dim3((unsigned)(((size_t)M * C + 255) / 256)), dim3(256), 0, 0, yw3, fd, (int64_t)M, C);
        CK(hipDeviceSynchronize());
        std::vector<float> h3((size_t)M * C);
        CK(hipMemcpy(h3.data(), fd, bytes, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < h3.size(); ++i) {
            if (!(h3[i] == h3[i])) { ++w16diff; continue; }
            if (h3[i] != hw[i]) ++w16diff;
            w16max = std::max(w16max, (double)std::fabs(h3[i] - hw[i]));
        }
        // ... and its 64-row tiles (four workgroups per CU)
        CK(hipMemset(yw3, 0xff, bytes));
        WinoConvParams w4 = w3;
        w4.tiles = (Mp + W16Geom<1>::TM - 1) / W16Geom<1>::TM * NS;
        launch_w16<C, true, 0, 1, 3>(w4, grid_for(w4.tiles, 4 * cus));
        CK(hipGetLastError()); CK(hipDeviceSynchronize());
        hipLaunchKernelGGL(planes_to_f32_kernel, dim3((unsigned)(((size_t)M * C + 255) / 256)), dim3(256), 0, 0, yw3, fd, (int64_t)M, C);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h3.data(), fd, bytes, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < h3.size(); ++i) {
            if (!(h3[i] == h3[i])) { ++s16diff; continue; }
            if (h3[i] != hw[i]) ++s16diff;
            s16max = std::max(s16max, (double)std::fabs(h3[i] - hw[i]));
        }
        hipFree(yw3);
    }
    printf("== %s C=%d B=%d %dx%d: M=%d Mp=%d; direct %d tiles grid %d, wino %d tiles grid %d\n", name, C, B, H, W, M, Mp, cp.tiles, gd, wp.tiles, gw);
    printf("  max|y| %.3f  max|wino - direct| %.3e (at pixel %zu ch %zu: direct %.6f wino %.6f)  NaN/unwritten in wino %zu\n", ymax, dmax, worst / C, worst % C,
           hd[worst], hw[worst], nan_w);
    printf("  vs fp64 host on %d sampled outputs: direct %.3e  wino %.3e;  words differing between the two transform forms: %zu\n", samples + 4, ed, ew, mixdiff);
    printf("  wino16 (LDS-DMA, 16-channel slabs) against wino: %zu values differ, max |difference| %.3e\n", w16diff, w16max);
    printf("  wino16 with 64-row tiles against wino: %zu values differ, max |difference| %.3e\n", s16diff, s16max);
    if (timing && split_only) {
        {   // when do the workgroups of ONE launch start and finish (100 MHz device-wide counter: 10 ns ticks)?  A
            // consumer tile of the next layer needs its three producer tiles: it cannot start before they have finished.
            long long *tb; CK(hipMalloc(&tb, (size_t)2 * 1024 * 8)); CK(hipMemset(tb, 0, (size_t)2 * 1024 * 8));
            WinoConvParams wt = wp; wt.trace = tb;
            launch_w<C, true, 128>(wt, gw); CK(hipDeviceSynchronize());
            launch_w<C, true, 128>(wt, gw); CK(hipDeviceSynchronize());
            std::vector<long long> ht(2 * 1024);
            CK(hipMemcpy(ht.data(), tb, ht.size() * 8, hipMemcpyDeviceToHost));
            std::vector<double> st, en;
            long long t0 = ht[0];
            for (int i = 0; i < gw; ++i) t0 = std::min(t0, ht[2 * i]);
            for (int i = 0; i < gw; ++i) st.push_back((ht[2 * i] - t0) * 0.01), en.push_back((ht[2 * i + 1] - t0) * 0.01);
            std::sort(st.begin(), st.end()); std::sort(en.begin(), en.end());
            auto q = [&](const std::vector<double> &v, double f) { return v[(size_t)(f * (v.size() - 1))]; };
            printf("  one launch, %d workgroups of one tile each (us after the first workgroup's start): starts min %.2f p50 %.2f max %.2f; ends min %.2f p10 %.2f p50 %.2f p90 %.2f max %.2f\n",
                   gw, q(st, 0), q(st, 0.5), q(st, 1), q(en, 0), q(en, 0.1), q(en, 0.5), q(en, 0.9), q(en, 1));
            // how early could tile t of the NEXT layer start?  It reads the pixels of tiles t - 1, t, t + 1 (same column tile order): when the last of the three has finished
            std::vector<double> ready, end_of_tile(wp.tiles, 0.0);
            auto tile_of = [&](int block) {  // xcd_tile_index (c3_gemm.h): block b runs on XCD b % 8, an XCD owns a contiguous run of tiles
                const int xcd = block & 7, slot = block >> 3, qq = wp.tiles >> 3, r = wp.tiles & 7;
                return (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + slot;
            };
            for (int i = 0; i < gw; ++i) end_of_tile[tile_of(i)] = (ht[2 * i + 1] - t0) * 0.01;  // (grid == tiles here: one tile per workgroup)
            for (int t = 0; t < wp.tiles; ++t) {
                double r = 0;
                for (int d = -NS; d <= NS; d += NS) { const int j = t + d; if (j >= 0 && j < wp.tiles) r = std::max(r, end_of_tile[j]); }
                ready.push_back(r);
            }
            std::sort(ready.begin(), ready.end());
            printf("  a consumer tile is runnable when its (up to) three producer workgroups have ended: min %.2f p10 %.2f p50 %.2f p90 %.2f max %.2f us -- a launch boundary releases all of them at %.2f us\n",
                   q(ready, 0), q(ready, 0.1), q(ready, 0.5), q(ready, 0.9), q(ready, 1), q(en, 1));
            hipFree(tb);
        }
        // split-K over the input channels for the launches that leave workgroup slots empty at B = 256.  ABL 64 runs every
        // tile as TWO workgroups on half of the slabs each, with no exchange of the partial sums: an upper bound of what the real thing
        // (partial tile through L2 / memory, second half adds and runs the epilogue) could gain.
        WinoConvParams w2 = wp;
        w2.tiles = 2 * wp.tiles;
        const int g2 = grid_for(w2.tiles, 2 * cus);
        for (int rep = 0; rep < 3; ++rep) {
            printf("  wino full                    %6.1f us  (%d tiles, grid %d)\n", time_it([&] { launch_w<C, true, 0>(wp, gw); }), wp.tiles, gw);
            printf("  wino split-K bound           %6.1f us  (%d half tiles, grid %d; no exchange)\n", time_it([&] { launch_w<C, true, 64>(w2, g2); }), w2.tiles, g2);
            printf("  wino split-K, no epilogue    %6.1f us\n", time_it([&] { launch_w<C, true, 64 | 4>(w2, g2); }));
            printf("  direct full                  %6.1f us  (%d tiles, grid %d)\n", time_it([&] { hipLaunchKernelGGL((conv3x3_planes_kernel<C, true>), dim3(gd), dim3(kPlThreads), 0, 0, cp); }), cp.tiles, gd);
        }
    } else if (timing) {
        const double mfd = 2.0 * tiles_m * kPlBM * (double)C * 9.0 * C * 3, mfw = 2.0 * tiles_w * kWRows * (double)C * 12.0 * C * 3;
        printf("  executed matrix work: direct %.1f GFLOP (%.1f us at 2500 TF), wino %.1f GFLOP (%.1f us)\n", mfd / 1e9, mfd / 2500e6, mfw / 1e9, mfw / 2500e6);
        printf("  direct full                  %6.1f us\n", time_it([&] { hipLaunchKernelGGL((conv3x3_planes_kernel<C, true>), dim3(gd), dim3(kPlThreads), 0, 0, cp); }));
        printf("  wino full                    %6.1f us\n", time_it([&] { launch_w<C, true, 0>(wp, gw); }));
        printf("  wino no weight loads         %6.1f us\n", time_it([&] { launch_w<C, true, 1>(wp, gw); }));
        printf("  wino no transform            %6.1f us\n", time_it([&] { launch_w<C, true, 2>(wp, gw); }));
        printf("  wino no epilogue             %6.1f us\n", time_it([&] { launch_w<C, true, 4>(wp, gw); }));
        printf("  wino no loads/transform/epi  %6.1f us\n", time_it([&] { launch_w<C, true, 7>(wp, gw); }));
        printf("  wino no MFMA                 %6.1f us\n", time_it([&] { launch_w<C, true, 8>(wp, gw); }));
        printf("  wino loads not ahead         %6.1f us\n", time_it([&] { launch_w<C, true, 16>(wp, gw); }));
        WinoConvParams w6 = wp; w6.wf = wfw16;
        printf("  wino16 full                  %6.1f us\n", time_it([&] { launch_w16<C, true, 0>(w6, gw); }));
        printf("  wino16 no weight loads       %6.1f us\n", time_it([&] { launch_w16<C, true, 1>(w6, gw); }));
        printf("  wino16 no DMA / transform    %6.1f us\n", time_it([&] { launch_w16<C, true, 2>(w6, gw); }));
        printf("  wino16 no epilogue           %6.1f us\n", time_it([&] { launch_w16<C, true, 4>(w6, gw); }));
        printf("  wino16 matrix + LDS only     %6.1f us\n", time_it([&] { launch_w16<C, true, 7>(w6, gw); }));
        printf("  wino16 full again            %6.1f us\n", time_it([&] { launch_w16<C, true, 0>(w6, gw); }));
        {
            WinoConvParams w7 = w6;
            w7.tiles = (Mp + W16Geom<1>::TM - 1) / W16Geom<1>::TM * NS;
            const int g4 = grid_for(w7.tiles, 4 * cus), g3 = grid_for(w7.tiles, 3 * cus), g2 = grid_for(w7.tiles, 2 * cus);
            printf("  wino16 64-row tiles: %d tiles, grid %d\n", w7.tiles, g4);
            printf("  wino16/64 full (ring 4, scratch) %6.1f us\n", time_it([&] { launch_w16<C, true, 0, 1, 4>(w7, g4); }));
            printf("  wino16/64 full (ring 3)      %6.1f us\n", time_it([&] { launch_w16<C, true, 0, 1, 3>(w7, g4); }));
            printf("  wino16/64 full (ring 2)      %6.1f us\n", time_it([&] { launch_w16<C, true, 0, 1, 2>(w7, g4); }));
            printf("  wino16/64 grid 3 per CU      %6.1f us (grid %d)\n", time_it([&] { launch_w16<C, true, 0, 1, 3>(w7, g3); }), g3);
            printf("  wino16/64 grid 2 per CU      %6.1f us (grid %d)\n", time_it([&] { launch_w16<C, true, 0, 1, 3>(w7, g2); }), g2);
            printf("  wino16/64 no weight loads    %6.1f us\n", time_it([&] { launch_w16<C, true, 1, 1, 3>(w7, g4); }));
            printf("  wino16/64 no DMA / transform %6.1f us\n", time_it([&] { launch_w16<C, true, 2, 1, 3>(w7, g4); }));
            printf("  wino16/64 no epilogue        %6.1f us\n", time_it([&] { launch_w16<C, true, 4, 1, 3>(w7, g4); }));
            printf("  wino16/64 matrix + LDS only  %6.1f us\n", time_it([&] { launch_w16<C, true, 7, 1, 3>(w7, g4); }));
            printf("  wino16/64 full again (ring 3)%6.1f us\n", time_it([&] { launch_w16<C, true, 0, 1, 3>(w7, g4); }));
            long long *tb; CK(hipMalloc(&tb, 2 * 256 * 16)); CK(hipMemset(tb, 0, 2 * 256 * 16));
            WinoConvParams wt = w7; wt.trace = tb;
            launch_w16<C, true, 32, 1, 3>(wt, g4); launch_w16<C, true, 32, 1, 3>(wt, g4);
            CK(hipDeviceSynchronize());
            std::vector<long long> ht(2 * 256 * 2);
            CK(hipMemcpy(ht.data(), tb, ht.size() * 8, hipMemcpyDeviceToHost));
            printf("  wino16/64 trace workgroup 0:\n   ");
            for (int i = 1; i < 250 && ht[i * 2] != 0; ++i) printf(" %lld:+%lld", ht[i * 2], ht[i * 2 + 1] - ht[(i - 1) * 2 + 1]);
            printf("\n");
            hipFree(tb);
        }
        {   // phase trace of the wino16 form
            long long *tb; CK(hipMalloc(&tb, 2 * 256 * 16)); CK(hipMemset(tb, 0, 2 * 256 * 16));
            WinoConvParams wt = w6; wt.trace = tb;
            launch_w16<C, true, 32>(wt, gw); launch_w16<C, true, 32>(wt, gw);
            CK(hipDeviceSynchronize());
            std::vector<long long> ht(2 * 256 * 2);
            CK(hipMemcpy(ht.data(), tb, ht.size() * 8, hipMemcpyDeviceToHost));
            printf("  wino16 trace workgroup 0 (10+s taps of slab s done, 21 DMA landed + barrier, 22 transformed, 23 barrier; 30 epilogue 31 half staged 32 barrier 33 half stored 35 next tile transformed 36 barrier):\n   ");
            for (int i = 1; i < 250 && ht[i * 2] != 0; ++i) printf(" %lld:+%lld", ht[i * 2], ht[i * 2 + 1] - ht[(i - 1) * 2 + 1]);
            printf("\n");
            hipFree(tb);
        }
        const int g1 = grid_for(wp.tiles, cus);
        printf("  wino full, one wg per CU     %6.1f us (grid %d)\n", time_it([&] { launch_w<C, true, 0>(wp, g1); }), g1);
        {   // shader-clock trace of workgroups 0 and 256, wave 0
            long long *tb; CK(hipMalloc(&tb, 2 * 256 * 16)); CK(hipMemset(tb, 0, 2 * 256 * 16));
            WinoConvParams wt = wp; wt.trace = tb;
            launch_w<C, true, 32>(wt, gw); launch_w<C, true, 32>(wt, gw);
            CK(hipDeviceSynchronize());
            std::vector<long long> ht(2 * 256 * 2);
            CK(hipMemcpy(ht.data(), tb, ht.size() * 8, hipMemcpyDeviceToHost));
            for (int wg = 0; wg < 2; ++wg) {
                printf("  trace workgroup %d (tag:+cycles; 2 first transform done 3 barrier; 10+s taps of slab s done, 20 next items requested, 21 barrier, 22 transformed, 23 barrier; 30 V free 31 staged 32 barrier 33 stored 34 barrier 35 next tile transformed 36 barrier):\n   ", wg ? 256 : 0);
                for (int i = 1; i < 250 && ht[(wg * 256 + i) * 2] != 0; ++i)
                    printf(" %lld:+%lld", ht[(wg * 256 + i) * 2], ht[(wg * 256 + i) * 2 + 1] - ht[(wg * 256 + i - 1) * 2 + 1]);
                printf("\n");
            }
            hipFree(tb);
        }
        printf("  direct full again            %6.1f us\n", time_it([&] { hipLaunchKernelGGL((conv3x3_planes_kernel<C, true>), dim3(gd), dim3(kPlThreads), 0, 0, cp); }));
        printf("  wino full again              %6.1f us\n", time_it([&] { launch_w<C, true, 0>(wp, gw); }));
    }
    hipFree(wfw16); hipFree(x); hipFree(yd); hipFree(yw); hipFree(r); hipFree(wfd); hipFree(wfw); hipFree(dbias); hipFree(dpost); hipFree(dpostw); hipFree(flag); hipFree(fd); hipFree(fw);
    return 0;
}
int main(int argc, char **argv) {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const bool quick = argc > 1 && !strcmp(argv[1], "quick");
    if (argc > 1 && !strcmp(argv[1], "split")) {  // round 6: what a split-K launch of the under-filled layers could take at most (no exchange)
        if (shape<256>("res3 split-K bound", 256, 12, 5, cus, true, true)) return 1;
        if (shape<128>("res2 split-K bound", 256, 23, 9, cus, true, true)) return 1;
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "big")) {  // the counter passes of tools/wino_pmc.sh: the two res2 batches only
        if (shape<128>("res2", 256, 23, 9, cus, true)) return 1;
        if (shape<128>("res2 B=1000", 1000, 23, 9, cus, true)) return 1;
        return 0;
    }
    // ragged small batches first (tile tails, odd and even H)
    if (shape<64>("res1-small", 3, 45, 17, cus, false)) return 1;
    if (shape<128>("res2-small", 5, 23, 9, cus, false)) return 1;
    if (shape<256>("res3-small", 7, 12, 5, cus, false)) return 1;
    if (shape<128>("res2-odd", 1, 7, 3, cus, false)) return 1;
    if (quick) return 0;
    if (shape<64>("res1", 256, 45, 17, cus, true)) return 1;
    if (shape<128>("res2", 256, 23, 9, cus, true)) return 1;
    if (shape<256>("res3", 256, 12, 5, cus, true)) return 1;
    if (shape<128>("res2 B=1000", 1000, 23, 9, cus, true)) return 1;
    return 0;
}

// c3_debug.h -- introspection for the parity tests and bench.py: per-layer activations of the last forward pass, HIP-event
// timing of every launch on the launch stream.  Not needed by a pipeline.
#pragma once
#include "c3_forward.h"

extern "C" {

int c3_debug_keep_activations(c3_model *m, int enable) {
    if (!m) return fail("null model");
    HIP_TRY(hipSetDevice(m->device));
    if (m->keep != (enable != 0)) {
        HIP_TRY(hipStreamSynchronize(m->stream));
        free_workspace(m);
        m->keep = enable != 0;
    }
    return 0;
}

int c3_debug_fetch(c3_model *m, const char *name, float *host_out, int64_t n_floats) {
    if (!m || !name || !host_out) return fail("null argument");
    HIP_TRY(hipSetDevice(m->device));
    if (m->last_n <= 0) return fail("nothing has been predicted yet");
    const std::string s = name;
    const float *src = nullptr;
    int64_t n = 0;
    if (m->kind == C3_KIND_PILEUP) {
        if (s == "lstm1_out") src = m->h1, n = m->last_n * m->positions * 256;
        else if (s == "lstm2_out") src = m->h2, n = m->last_n * m->positions * 320;
        else if (s == "gx2") src = m->gx2, n = m->last_n * m->positions * 1280;
    } else {
        int hh[10], ww[10];
        fa_geometry(m, hh, ww);
        if (s.size() == 4 && s.compare(0, 3, "act") == 0 && s[3] >= '0' && s[3] <= '8') {
            if (!m->keep) return fail("activations are recycled: enable c3_debug_keep_activations first");
            const int l = s[3] - '0';
            src = m->act[l], n = m->last_n * hh[l + 1] * ww[l + 1] * kConvCout[l];
        } else if (s == "spp") src = m->spp, n = m->last_n * m->K4;
    }
    if (s == "l4_out") {
        if (!m->keep) return fail("l4_out is only written with c3_debug_keep_activations enabled");
        src = m->l4dbg, n = m->last_n * m->FC;
    }
    if (!src) return fail("unknown debug tensor \"%s\"", name);
    if (n != n_floats) return fail("debug tensor %s has %lld floats, caller expects %lld", name, (long long)n, (long long)n_floats);
    HIP_TRY(hipDeviceSynchronize());
    // activations of the full-alignment network live on the device with every channel times its power of two (channel
    // equalisation, c3_pack.h): hand the caller the values of the checkpoint as given
    const std::vector<int> *exps = nullptr;
    if (m->kind == C3_KIND_FULL_ALIGNMENT && s.compare(0, 3, "act") == 0) exps = &m->act_exp[s[3] - '0'];
    if (m->kind == C3_KIND_FULL_ALIGNMENT && s == "spp") exps = &m->act_exp[8];
    auto unscale = [&]() {
        if (!exps || exps->empty()) return;
        const size_t C = exps->size();
        for (int64_t i = 0; i < n; ++i) host_out[i] = std::ldexp(host_out[i], -(*exps)[(size_t)i % C]);
    };
    if (m->last_planes && ((m->kind == C3_KIND_FULL_ALIGNMENT && s.compare(0, 3, "act") == 0) || (m->kind == C3_KIND_PILEUP && s == "lstm1_out"))) {
        // the layer holds plane activations (c3_conv3.h): hand the caller the fp32 values they stand for
        const int C = m->kind == C3_KIND_PILEUP ? 256 : kConvCout[s[3] - '0'];
        float *tmp = nullptr;
        HIP_TRY(hipMalloc((void **)&tmp, (size_t)n * sizeof(float)));
        hipLaunchKernelGGL(planes_to_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (const void *)src, tmp, n / C, C);
        const int rc = d2h_staged(host_out, tmp, (size_t)n * sizeof(float));
        (void)hipFree(tmp);
        if (rc != 0) return rc;
        unscale();
        return 0;
    }
    TRY(d2h_staged(host_out, src, (size_t)n * sizeof(float)));
    unscale();
    return 0;
}

int c3_profile_enable(c3_model *m, int enable) {
    if (!m) return fail("null model");
    m->prof = enable != 0;
    return 0;
}

int c3_profile_reset(c3_model *m) {
    if (!m) return fail("null model");
    HIP_TRY(hipSetDevice(m->device));
    HIP_TRY(hipDeviceSynchronize());
    for (auto &r : m->recs) {
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    m->recs.clear();
    return 0;
}

int c3_profile_read(c3_model *m, c3_kernel_stat *out, int max_entries) {
    if (!m || (!out && max_entries > 0)) {
        fail("null argument");
        return -1;
    }
    if (hipSetDevice(m->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
        fail("device synchronize failed");
        return -1;
    }
    std::vector<std::string> order;
    std::map<std::string, c3_kernel_stat> agg;
    for (auto &r : m->recs) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) continue;
        auto it = agg.find(r.name);
        if (it == agg.end()) {
            c3_kernel_stat st;
            memset(&st, 0, sizeof(st));
            snprintf(st.name, sizeof(st.name), "%s", r.name.c_str());
            it = agg.insert({r.name, st}).first;
            order.push_back(r.name);
        }
        it->second.launches += 1;
        it->second.total_ms += ms;
        it->second.flops += r.flops;
        it->second.bytes += r.bytes;
        it->second.mfma_flops += r.mfma_flops;
        it->second.mfma_peak_tflops = std::max(it->second.mfma_peak_tflops, r.mfma_peak);
    }
    int n = 0;
    for (auto &k : order) {
        if (n >= max_entries) break;
        out[n++] = agg[k];
    }
    return (int)order.size();
}

}  // extern "C"

// fork_stall_probe.hip -- how long the device stops answering after the process fork()s (what the reference's stage-B loop does to
// its decode pool right after the first model call: profiles/r05_k_host_loop_feeder_not_kept.txt).  A stream of short kernels with a
// hipEventSynchronize after each; the longest gap between two completions around 8 fork()s, for a bare process, with pinned host
// memory (plain and MADV_DONTFORK) and with a big touched heap.
//   hipcc --offload-arch=gfx950 -O2 tools/fork_stall_probe.hip -o /tmp/fork_stall_probe && /tmp/fork_stall_probe
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void spin(float *p, int n) {
    float v = p[threadIdx.x];
    for (int i = 0; i < n; ++i) v = v * 1.0001f + 0.5f;
    p[threadIdx.x] = v;
}
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static size_t g_vram_mb = 0, g_pageable_mb = 0;
static int g_streams = 1;
static int run(const char *name, size_t pinned_mb, bool dontfork, size_t heap_mb, int forks) {
    float *d;
    CK(hipMalloc(&d, 1 << 20));
    std::vector<void *> vram;
    for (size_t i = 0; i < g_vram_mb / 256; ++i) {  // a model's workspace: many 256 MB blocks, touched
        void *p;
        CK(hipMalloc(&p, (size_t)256 << 20));
        CK(hipMemset(p, 0, (size_t)256 << 20));
        vram.push_back(p);
    }
    if (g_pageable_mb) {  // weights uploaded from pageable memory (the runtime pins the source for the copy)
        char *w = (char *)malloc(g_pageable_mb << 20);
        memset(w, 3, g_pageable_mb << 20);
        void *p;
        CK(hipMalloc(&p, g_pageable_mb << 20));
        CK(hipMemcpy(p, w, g_pageable_mb << 20, hipMemcpyHostToDevice));
        vram.push_back(p);
        if (getenv("FREE_PAGEABLE")) free(w);
    }
    std::vector<hipStream_t> extra;
    for (int i = 1; i < g_streams; ++i) {
        hipStream_t xs;
        CK(hipStreamCreate(&xs));
        hipLaunchKernelGGL(spin, dim3(1), dim3(256), 0, xs, d, 10);
        extra.push_back(xs);
    }
    CK(hipDeviceSynchronize());
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t ev;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    void *pin = nullptr;
    if (pinned_mb) {
        CK(hipHostMalloc(&pin, pinned_mb << 20, hipHostMallocDefault));
        memset(pin, 1, pinned_mb << 20);
        if (dontfork) madvise(pin, pinned_mb << 20, MADV_DONTFORK);
    }
    char *heap = nullptr;
    if (heap_mb) {
        heap = (char *)malloc(heap_mb << 20);
        memset(heap, 2, heap_mb << 20);
    }
    auto one = [&]() {
        if (pin) (void)hipMemcpyAsync(d, pin, 1 << 20, hipMemcpyHostToDevice, s);
        hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s, d, 2000);
        (void)hipEventRecord(ev, s);
        (void)hipEventSynchronize(ev);
    };
    for (int i = 0; i < 200; ++i) one();
    double worst_before = 0, t = now_ms();
    for (int i = 0; i < 500; ++i) {
        one();
        const double n = now_ms();
        worst_before = std::max(worst_before, n - t), t = n;
    }
    std::vector<pid_t> kids;
    const double tf0 = now_ms();
    for (int k = 0; k < forks; ++k) {
        pid_t p = fork();
        if (p == 0) {
            sleep(20);
            _exit(0);
        }
        kids.push_back(p);
    }
    const double tf1 = now_ms();
    double worst = 0, first = -1;
    t = now_ms();
    const double t0 = t;
    int slow = 0;
    for (int i = 0; i < 3000; ++i) {
        if (pin) ((char *)pin)[(size_t)i * 4096 % (pinned_mb << 20)] = (char)i;  // the caller keeps writing its staging memory
        one();
        const double n = now_ms();
        if (first < 0) first = n - t0;
        if (n - t > 1.0) ++slow;
        worst = std::max(worst, n - t), t = n;
    }
    printf("%-58s forks %d in %6.1f ms; first completion after %7.1f ms, longest gap %7.1f ms (before the forks %.2f ms), %d gaps > 1 ms, 3000 launches in %.0f ms\n",
           name, forks, tf1 - tf0, first, worst, worst_before, slow, t - t0);
    for (pid_t p : kids) kill(p, 9), waitpid(p, nullptr, 0);
    if (pin) (void)hipHostFree(pin);
    free(heap);
    (void)hipFree(d);
    for (void *p : vram) (void)hipFree(p);
    for (hipStream_t xs : extra) (void)hipStreamDestroy(xs);
    (void)hipStreamDestroy(s);
    return 0;
}
int main(int argc, char **argv) {
    if (argc > 1) {  // vram_mb pageable_mb streams
        g_vram_mb = atoi(argv[1]), g_pageable_mb = argc > 2 ? atoi(argv[2]) : 0, g_streams = argc > 3 ? atoi(argv[3]) : 1;
        char nm[128];
        snprintf(nm, sizeof nm, "%zu MB of VRAM in 256 MB blocks, %zu MB pageable upload, %d streams", g_vram_mb, g_pageable_mb, g_streams);
        return run(nm, 64, true, 0, 8);
    }
    if (run("bare process", 0, false, 0, 8)) return 1;
    if (run("bare process, one fork", 0, false, 0, 1)) return 1;
    if (run("150 MB pinned (hipHostMalloc)", 150, false, 0, 8)) return 1;
    if (run("150 MB pinned, MADV_DONTFORK", 150, true, 0, 8)) return 1;
    if (run("2 GB of touched heap", 0, false, 2048, 8)) return 1;
    if (run("2 GB of touched heap, one fork", 0, false, 2048, 1)) return 1;
    if (run("150 MB pinned DONTFORK + 2 GB heap", 150, true, 2048, 8)) return 1;
    return 0;
}

// fork_stall_lib.cpp -- the first model call behind a fork(), from a plain C++ process on libc3hip (no Python, no numpy, no torch):
// does the ~0.3 s stall of tests/diag/fork_stall.py belong to the library or to the interpreter's process?
//   python tools/fork_stall_lib.py /tmp/sd      (writes /tmp/sd.bin + /tmp/sd.txt: a synthetic pileup state dict)
//   g++ -O2 -I include tools/fork_stall_lib.cpp -o /tmp/fork_stall_lib -ldl && /tmp/fork_stall_lib clair3_amd/lib/libc3hip.so /tmp/sd
#include <dlfcn.h>
#include <sys/wait.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "c3hip.h"
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    if (argc < 3) return 2;
    void *h = dlopen(argv[1], RTLD_NOW);
    if (!h) { printf("dlopen: %s\n", dlerror()); return 1; }
    auto create = (c3_model * (*)(int, int, int, int)) dlsym(h, "c3_model_create");
    auto load = (int (*)(c3_model *, const c3_tensor_desc *, int))dlsym(h, "c3_model_load");
    auto predict = (int (*)(c3_model *, const void *, int, int64_t, float *))dlsym(h, "c3_predict");
    auto err = (const char *(*)())dlsym(h, "c3_last_error");
    std::string base = argv[2];
    FILE *fb = fopen((base + ".bin").c_str(), "rb");
    fseek(fb, 0, SEEK_END);
    const long nb = ftell(fb);
    fseek(fb, 0, SEEK_SET);
    std::vector<char> blob(nb);
    if (fread(blob.data(), 1, nb, fb) != (size_t)nb) return 1;
    fclose(fb);
    FILE *ft = fopen((base + ".txt").c_str(), "r");
    std::vector<c3_tensor_desc> descs;
    std::vector<std::string> names(512);
    char name[256];
    int nd;
    long off;
    long long sh[4];
    int k = 0;
    while (fscanf(ft, "%255s %d %lld %lld %lld %lld %ld", name, &nd, &sh[0], &sh[1], &sh[2], &sh[3], &off) == 7) {
        names[k] = name;
        c3_tensor_desc d;
        d.name = names[k].c_str(), d.dtype = C3_DTYPE_F32, d.ndim = nd, d.data = blob.data() + off;
        for (int i = 0; i < 4; ++i) d.shape[i] = sh[i];
        descs.push_back(d);
        ++k;
    }
    fclose(ft);
    c3_model *m = create(C3_KIND_PILEUP, 18, 0, 0);
    if (!m || load(m, descs.data(), (int)descs.size())) { printf("load: %s\n", err()); return 1; }
    const int64_t n = 1000;
    std::vector<int8_t> x((size_t)n * 33 * 18, 1);
    std::vector<float> y((size_t)n * 24);
    auto call = [&]() {
        const double t0 = now_ms();
        if (predict(m, x.data(), C3_DTYPE_I8, n, y.data())) printf("predict: %s\n", err());
        return now_ms() - t0;
    };
    for (int i = 0; i < 5; ++i) call();
    double steady = 1e9;
    for (int i = 0; i < 5; ++i) steady = std::min(steady, call());
    const int forks = argc > 3 ? atoi(argv[3]) : 1;
    std::vector<pid_t> kids;
    for (int i = 0; i < forks; ++i) {
        pid_t p = fork();
        if (p == 0) { sleep(15); _exit(0); }
        kids.push_back(p);
    }
    printf("C++ process on libc3hip, pileup 1000 windows: steady %.2f ms, after %d fork(s):", steady, forks);
    for (int i = 0; i < 5; ++i) printf(" %.1f", call());
    printf(" ms\n");
    for (pid_t p : kids) kill(p, 9), waitpid(p, nullptr, 0);
    return 0;
}

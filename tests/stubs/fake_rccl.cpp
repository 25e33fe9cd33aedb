// fake_rccl.cpp -- TEST INFRASTRUCTURE: a stand-in for librccl that lets the grouped ncclSend / ncclRecv path of
// clair3_amd/csrc/c3_comm.h (c3_gather_rows at world > 1) execute with two processes on ONE GPU (real RCCL refuses two ranks
// on one device, and a lease has one device).  Selected with C3HIP_RCCL_LIB=<this .so>.  Messages travel through files in a
// rendezvous directory named by the unique id: device -> host (hipMemcpy) -> file -> host -> device.  Group semantics as
// RCCL's: operations issued between ncclGroupStart / ncclGroupEnd start together at the end -- sends never block, receives
// poll for their message.  Signatures come from rccl.h itself.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

struct ncclComm {
    std::string dir;
    int rank = 0, nranks = 1;
    std::vector<long> sent, received;  // per-peer message counters
};

namespace {
struct Op {
    bool send;
    void *buf;
    size_t bytes;
    int peer;
    ncclComm *comm;
    hipStream_t stream;
};
thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;
const double kTimeoutS = 30.0;

double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
bool exists(const std::string &p) {
    struct stat st;
    return stat(p.c_str(), &st) == 0;
}
size_t width(ncclDataType_t t) { return t == ncclFloat32 || t == ncclInt32 || t == ncclUint32 ? 4 : t == ncclFloat64 || t == ncclInt64 || t == ncclUint64 ? 8 : t == ncclFloat16 || t == ncclBfloat16 ? 2 : 1; }

ncclResult_t run(const Op &op) {
    char name[64];
    if (op.send) {
        std::vector<char> host(op.bytes);
        if (hipStreamSynchronize(op.stream) != hipSuccess) return ncclUnhandledCudaError;
        if (op.bytes && hipMemcpy(host.data(), op.buf, op.bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
        snprintf(name, sizeof name, "/msg_%d_%d_%ld", op.comm->rank, op.peer, op.comm->sent[op.peer]++);
        const std::string tmp = op.comm->dir + name + ".tmp", fin = op.comm->dir + name;
        FILE *f = fopen(tmp.c_str(), "wb");
        if (!f) return ncclSystemError;
        const bool ok = fwrite(host.data(), 1, op.bytes, f) == op.bytes;
        fclose(f);
        if (!ok || rename(tmp.c_str(), fin.c_str()) != 0) return ncclSystemError;
        return ncclSuccess;
    }
    snprintf(name, sizeof name, "/msg_%d_%d_%ld", op.peer, op.comm->rank, op.comm->received[op.peer]++);
    const std::string fin = op.comm->dir + name;
    const double t0 = now();
    while (!exists(fin)) {
        if (now() - t0 > kTimeoutS) return ncclSystemError;
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
    std::vector<char> host(op.bytes);
    FILE *f = fopen(fin.c_str(), "rb");
    if (!f) return ncclSystemError;
    const bool ok = fread(host.data(), 1, op.bytes, f) == op.bytes && fgetc(f) == EOF;  // size mismatch = a protocol error
    fclose(f);
    unlink(fin.c_str());
    if (!ok) return ncclInvalidArgument;
    if (hipStreamSynchronize(op.stream) != hipSuccess) return ncclUnhandledCudaError;
    if (op.bytes && hipMemcpy(op.buf, host.data(), op.bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}

ncclResult_t issue(const Op &op) {
    if (op.peer < 0 || op.peer >= op.comm->nranks || op.peer == op.comm->rank) return ncclInvalidArgument;
    if (g_depth > 0) {
        g_ops.push_back(op);
        return ncclSuccess;
    }
    return run(op);
}
}  // namespace

extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    memset(id, 0, sizeof *id);
    snprintf(id->internal, sizeof id->internal, "/tmp/fake_rccl_%d_%lld", (int)getpid(), (long long)(now() * 1e6));
    return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank) {
    if (nranks < 1 || rank < 0 || rank >= nranks || id.internal[0] != '/') return ncclInvalidArgument;
    ncclComm *c = new ncclComm;
    c->dir = std::string(id.internal, strnlen(id.internal, sizeof id.internal));
    c->rank = rank, c->nranks = nranks;
    c->sent.assign(nranks, 0), c->received.assign(nranks, 0);
    mkdir(c->dir.c_str(), 0700);
    FILE *f = fopen((c->dir + "/joined_" + std::to_string(rank)).c_str(), "w");
    if (!f) return ncclSystemError;
    fclose(f);
    const double t0 = now();
    for (int r = 0; r < nranks; ++r)
        while (!exists(c->dir + "/joined_" + std::to_string(r))) {
            if (now() - t0 > kTimeoutS) return ncclSystemError;
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
    *comm = c;
    return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    delete comm;
    return ncclSuccess;
}
ncclResult_t ncclCommAbort(ncclComm_t comm) {
    delete comm;
    return ncclSuccess;
}
ncclResult_t ncclCommCount(const ncclComm_t comm, int *count) {
    *count = comm->nranks;
    return ncclSuccess;
}
ncclResult_t ncclCommUserRank(const ncclComm_t comm, int *rank) {
    *rank = comm->rank;
    return ncclSuccess;
}
ncclResult_t ncclCommGetAsyncError(ncclComm_t, ncclResult_t *e) {
    *e = ncclSuccess;
    return ncclSuccess;
}
ncclResult_t ncclGroupStart() {
    ++g_depth;
    return ncclSuccess;
}
ncclResult_t ncclGroupEnd() {
    if (g_depth <= 0) return ncclInvalidUsage;
    if (--g_depth > 0) return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(g_ops);
    ncclResult_t rc = ncclSuccess;
    for (const Op &op : ops)  // sends first: they never block, so no order of ranks can deadlock
        if (op.send && rc == ncclSuccess) rc = run(op);
    for (const Op &op : ops)
        if (!op.send && rc == ncclSuccess) rc = run(op);
    return rc;
}
ncclResult_t ncclSend(const void *sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
    return issue(Op{true, const_cast<void *>(sendbuff), count * width(datatype), peer, comm, stream});
}
ncclResult_t ncclRecv(void *recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
    return issue(Op{false, recvbuff, count * width(datatype), peer, comm, stream});
}
const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "fake_rccl error"; }
}

// c3_comm.h -- the one collective of the sharded job: a gather(v) of probability rows to one rank, issued
// directly on RCCL (grouped ncclSend / ncclRecv over xGMI) on the caller's HIP stream -- no framework between the forward
// pass and the wire.  librccl is bound at run time with dlopen: a single-GPU worker never loads it, and a process that has
// PyTorch loaded shares the copy PyTorch ships instead of bringing a second one.
//
// Rendezvous is the caller's business (128-byte unique id from rank 0 to everybody -- the Python layer sends it through
// whatever control plane launched the ranks, e.g. torch.distributed's store); the data path is RCCL only.
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

namespace c3 {

struct RcclUniqueId {  // ncclUniqueId: 128 opaque bytes, passed BY VALUE to ncclCommInitRank
    char b[128];
};

struct RcclApi {
    typedef int (*GetUniqueId_t)(RcclUniqueId *id);
    typedef int (*CommInitRank_t)(void **comm, int nranks, RcclUniqueId id, int rank);
    typedef int (*CommDestroy_t)(void *comm);
    typedef int (*Group_t)(void);
    typedef int (*SendRecv_t)(void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t stream);
    typedef const char *(*ErrorString_t)(int);
    void *handle = nullptr;
    GetUniqueId_t GetUniqueId = nullptr;
    CommInitRank_t CommInitRank = nullptr;
    CommDestroy_t CommDestroy = nullptr;
    Group_t GroupStart = nullptr, GroupEnd = nullptr;
    SendRecv_t Send = nullptr, Recv = nullptr;
    ErrorString_t GetErrorString = nullptr;
    std::string error;

    static RcclApi &get() {
        static RcclApi api;
        return api;
    }
    bool load() {
        if (handle) return true;
        const char *names[] = {getenv("C3HIP_RCCL_LIB"), "librccl.so.1", "librccl.so"};
        // first: a copy that is already in the process (PyTorch's), then the system one
        for (int pass = 0; pass < 2 && !handle; ++pass)
            for (const char *n : names) {
                if (!n) continue;
                handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
                if (handle) break;
            }
        if (!handle) {
            error = std::string("cannot load librccl: ") + (dlerror() ? dlerror() : "not found");
            return false;
        }
        auto sym = [&](const char *s) -> void * {
            void *p = dlsym(handle, s);
            if (!p) error = std::string("librccl has no symbol ") + s;
            return p;
        };
        GetUniqueId = (GetUniqueId_t)sym("ncclGetUniqueId");
        CommInitRank = (CommInitRank_t)sym("ncclCommInitRank");
        CommDestroy = (CommDestroy_t)sym("ncclCommDestroy");
        GroupStart = (Group_t)sym("ncclGroupStart");
        GroupEnd = (Group_t)sym("ncclGroupEnd");
        Send = (SendRecv_t)sym("ncclSend");
        Recv = (SendRecv_t)sym("ncclRecv");
        GetErrorString = (ErrorString_t)sym("ncclGetErrorString");
        if (!GetUniqueId || !CommInitRank || !CommDestroy || !GroupStart || !GroupEnd || !Send || !Recv || !GetErrorString) {
            handle = nullptr;
            return false;
        }
        return true;
    }
};

constexpr int kNcclFloat = 7;  // ncclFloat32 (rccl.h)

}  // namespace c3

struct c3_comm {
    void *nccl = nullptr;  // ncclComm_t, null when world == 1
    int rank = 0, world = 1, device = 0;
};

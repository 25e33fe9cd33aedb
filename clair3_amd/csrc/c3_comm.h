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
#include <rccl/rccl.h>  // types, enums and prototypes only: the symbols are bound with dlopen below, nothing links librccl
#include <stdint.h>

#include <string>
#include <vector>

namespace c3 {

// Function-pointer types are taken from rccl.h's own prototypes (decltype), so the by-value 128-byte ncclUniqueId, the
// ncclDataType_t numbering and every argument list are the header's, not a transcription of it.
struct RcclApi {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclCommUserRank) CommUserRank = nullptr;
    decltype(&ncclCommGetAsyncError) CommGetAsyncError = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;

    static RcclApi &get() {
        static RcclApi api;
        return api;
    }
    bool load() {
        if (handle) return true;
        const char *names[] = {getenv("C3HIP_RCCL_LIB"), "librccl.so.1", "librccl.so"};
        // first: a copy that is already in the process (PyTorch's), then the system one
        const char *why = nullptr;
        for (int pass = 0; pass < 2 && !handle; ++pass)
            for (const char *n : names) {
                if (!n) continue;
                handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
                if (handle) break;
                if (pass == 1) {
                    const char *e = dlerror();  // reading it clears it: once per failure
                    if (e && !why) why = e;
                }
            }
        if (!handle) {
            error = std::string("cannot load librccl: ") + (why ? why : "not found");
            return false;
        }
        bool ok = true;
        auto sym = [&](const char *s) -> void * {
            void *p = dlsym(handle, s);
            if (!p) error = std::string("librccl has no symbol ") + s, ok = false;
            return p;
        };
        GetUniqueId = (decltype(GetUniqueId))sym("ncclGetUniqueId");
        CommInitRank = (decltype(CommInitRank))sym("ncclCommInitRank");
        CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
        CommAbort = (decltype(CommAbort))sym("ncclCommAbort");
        CommCount = (decltype(CommCount))sym("ncclCommCount");
        CommUserRank = (decltype(CommUserRank))sym("ncclCommUserRank");
        CommGetAsyncError = (decltype(CommGetAsyncError))sym("ncclCommGetAsyncError");
        GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
        Send = (decltype(Send))sym("ncclSend");
        Recv = (decltype(Recv))sym("ncclRecv");
        GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
        if (!ok) {
            handle = nullptr;
            return false;
        }
        return true;
    }
};

}  // namespace c3

struct c3_comm {
    ncclComm_t nccl = nullptr;  // null when world == 1
    int rank = 0, world = 1, device = 0;
};

// c3_comm.h -- the one collective of the sharded job: a gather(v) of probability rows to one rank, issued
// directly on RCCL (grouped ncclSend / ncclRecv over xGMI) on the caller's HIP stream -- no framework between the forward
// pass and the wire.  librccl is bound at run time with dlopen: a single-GPU worker never loads it, and a process that has
// PyTorch loaded shares the copy PyTorch ships instead of bringing a second one.
//
// Rendezvous is the caller's business (128-byte unique id from rank 0 to everybody -- the Python layer sends it through
// whatever control plane launched the ranks, e.g. torch.distributed's store); the data path is RCCL only.
#pragma once
#include "c3_model.h"
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
        const char *why = nullptr;
        // an explicit C3HIP_RCCL_LIB wins over any copy already in the process (RTLD_LOCAL: its ncclSend must not interpose
        // PyTorch's) -- a site-specific build, or the two-ranks-on-one-GPU stand-in of tests/stubs/fake_rccl.cpp
        if (names[0] && *names[0]) {
            handle = dlopen(names[0], RTLD_NOW | RTLD_LOCAL);
            if (!handle) why = dlerror();
        }
        // otherwise first a copy that is already in the process (PyTorch's), then the system one
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
    ncclComm_t nccl = nullptr;  // null when world == 1 (unless C3HIP_FORCE_RCCL made a one-rank communicator)
    int rank = 0, world = 1, device = 0;
    int orig_rank = 0;     // the rank the handle was created with: the index of this process in the caller's counts[] for good
    bool aborted = false;  // c3_comm_abort: from then on the gather is the local copy of this rank's own rows
};

extern "C" {

// ---- the gather of the sharded job on RCCL (c3_comm.h) ----
int c3_comm_unique_id(void *id128) {
    if (!id128) return fail("null buffer");
    RcclApi &r = RcclApi::get();
    if (!r.load()) return fail("%s", r.error.c_str());
    static_assert(sizeof(ncclUniqueId) == 128, "c3_comm_unique_id hands out 128 bytes");
    ncclUniqueId id;
    const ncclResult_t rc = r.GetUniqueId(&id);
    if (rc != ncclSuccess) return fail("ncclGetUniqueId failed: %s", r.GetErrorString(rc));
    memcpy(id128, &id, 128);
    return 0;
}

c3_comm *c3_comm_create(const void *id128, int rank, int world, int device) {
    if (world < 1 || rank < 0 || rank >= world) {
        fail("bad rank %d of %d", rank, world);
        return nullptr;
    }
    if (hipSetDevice(device) != hipSuccess) {
        fail("hipSetDevice(%d) failed", device);
        return nullptr;
    }
    c3_comm *c = new c3_comm();
    c->rank = c->orig_rank = rank, c->world = world, c->device = device;
    // C3HIP_FORCE_RCCL=1 (test knob): a world of ONE still goes through librccl -- dlopen + symbol binding, ncclGetUniqueId,
    // ncclCommInitRank(nranks = 1), ncclCommCount, a grouped self ncclSend / ncclRecv in c3_gather_rows, ncclCommDestroy -- so
    // that every call this file makes has met the REAL library on a one-GPU box before the day an 8-GPU node exists
    const char *force = getenv("C3HIP_FORCE_RCCL");
    const bool forced = world == 1 && force && atoi(force) > 0;
    if (world == 1 && !forced) return c;  // nothing to talk to: c3_gather_rows is a device copy
    if (!id128 && !forced) {
        fail("null unique id");
        delete c;
        return nullptr;
    }
    RcclApi &r = RcclApi::get();
    if (!r.load()) {
        fail("%s", r.error.c_str());
        delete c;
        return nullptr;
    }
    ncclUniqueId id;
    bool have_id = false;
    if (id128) {
        memcpy(&id, id128, 128);
        for (int i = 0; i < 128 && !have_id; ++i) have_id = reinterpret_cast<const char *>(id128)[i] != 0;
    }
    if (!have_id) {  // forced one-rank communicator without an id from the caller: make one here
        if (!forced) {
            fail("all-zero unique id");
            delete c;
            return nullptr;
        }
        const ncclResult_t rid = r.GetUniqueId(&id);
        if (rid != ncclSuccess) {
            fail("ncclGetUniqueId failed: %s", r.GetErrorString(rid));
            delete c;
            return nullptr;
        }
    }
    const ncclResult_t rc = r.CommInitRank(&c->nccl, world, id, rank);
    if (rc != ncclSuccess) {
        fail("ncclCommInitRank failed: %s", r.GetErrorString(rc));
        delete c;
        return nullptr;
    }
    return c;
}

int c3_comm_destroy(c3_comm *c) {
    if (!c) return 0;
    if (c->nccl) (void)RcclApi::get().CommDestroy(c->nccl);
    delete c;
    return 0;
}

int c3_gather_rows(c3_comm *c, const float *rows_dev, int row_floats, const int64_t *counts, float *all_dev, int dst, void *stream) {
    if (!c || !counts) return fail("null argument");
    if (row_floats <= 0) return fail("bad arguments (%d floats per row)", row_floats);
    hipStream_t s = (hipStream_t)stream;
    if (c->aborted) {
        // the communicator is gone: what is left of the gather is the local copy of THIS rank's rows.  counts[] is still the
        // caller's array over the original ranks, so this rank's count sits at orig_rank (not at 0)
        if (counts[c->orig_rank] < 0) return fail("negative row count for rank %d", c->orig_rank);
        const size_t own = (size_t)counts[c->orig_rank] * row_floats;
        if (own && !rows_dev) return fail("null rows");
        if (!all_dev) return fail("communicator aborted: the gather is the local copy of rank %d's rows and needs a destination buffer", c->orig_rank);
        HIP_TRY(hipSetDevice(c->device));
        if (own && all_dev != rows_dev) HIP_TRY(hipMemcpyAsync(all_dev, rows_dev, own * sizeof(float), hipMemcpyDeviceToDevice, s));
        return 0;
    }
    if (dst < 0 || dst >= c->world) return fail("bad arguments (dst %d of %d ranks)", dst, c->world);
    for (int r = 0; r < c->world; ++r)
        if (counts[r] < 0) return fail("negative row count for rank %d", r);
    HIP_TRY(hipSetDevice(c->device));
    const size_t mine = (size_t)counts[c->rank] * row_floats;
    if (mine && !rows_dev) return fail("null rows");
    if (c->rank == dst && !all_dev) return fail("the destination rank needs the gathered buffer");
    if (c->world == 1 && !c->nccl) {
        if (mine && all_dev != rows_dev) HIP_TRY(hipMemcpyAsync(all_dev, rows_dev, mine * sizeof(float), hipMemcpyDeviceToDevice, s));
        return 0;
    }
    RcclApi &r = RcclApi::get();
    if (c->world == 1) {  // C3HIP_FORCE_RCCL: the one rank sends its rows to itself through the grouped send / receive pair
        if (!mine || all_dev == rows_dev) return 0;
        ncclResult_t g = r.GroupStart();
        if (g != ncclSuccess) return fail("ncclGroupStart failed: %s", r.GetErrorString(g));
        const ncclResult_t a = r.Send(rows_dev, mine, ncclFloat32, 0, c->nccl, s);
        const ncclResult_t b = r.Recv(all_dev, mine, ncclFloat32, 0, c->nccl, s);
        g = r.GroupEnd();
        if (a != ncclSuccess) return fail("ncclSend (to self) failed: %s", r.GetErrorString(a));
        if (b != ncclSuccess) return fail("ncclRecv (from self) failed: %s", r.GetErrorString(b));
        if (g != ncclSuccess) return fail("ncclGroupEnd failed: %s", r.GetErrorString(g));
        return 0;
    }
    int rc = (int)r.GroupStart();
    if (rc) return fail("ncclGroupStart failed: %s", r.GetErrorString((ncclResult_t)rc));
    if (c->rank == dst) {
        size_t off = 0;
        for (int src = 0; src < c->world && !rc; ++src) {
            const size_t n = (size_t)counts[src] * row_floats;
            if (src == dst) {
                if (n && all_dev + off != rows_dev) {
                    hipError_t e = hipMemcpyAsync(all_dev + off, rows_dev, n * sizeof(float), hipMemcpyDeviceToDevice, s);
                    if (e != hipSuccess) rc = -1;
                }
            } else if (n) {
                rc = (int)r.Recv(all_dev + off, n, ncclFloat32, src, c->nccl, s);
            }
            off += n;
        }
    } else if (mine) {
        rc = (int)r.Send(rows_dev, mine, ncclFloat32, dst, c->nccl, s);
    }
    const ncclResult_t rc2 = r.GroupEnd();
    if (rc > 0) return fail("ncclSend/ncclRecv failed: %s", r.GetErrorString((ncclResult_t)rc));
    if (rc < 0) return fail("device copy inside the gather failed");
    if (rc2 != ncclSuccess) return fail("ncclGroupEnd failed: %s", r.GetErrorString(rc2));
    return 0;
}

int c3_comm_count(c3_comm *c, int *ranks_out, int *rank_out) {
    if (!c || !ranks_out) return fail("null argument");
    if (!c->nccl) {  // world == 1: no communicator
        *ranks_out = c->world;
        if (rank_out) *rank_out = c->rank;
        return 0;
    }
    RcclApi &r = RcclApi::get();
    ncclResult_t rc = r.CommCount(c->nccl, ranks_out);
    if (rc != ncclSuccess) return fail("ncclCommCount failed: %s", r.GetErrorString(rc));
    if (rank_out) {
        rc = r.CommUserRank(c->nccl, rank_out);
        if (rc != ncclSuccess) return fail("ncclCommUserRank failed: %s", r.GetErrorString(rc));
    }
    return 0;
}

int c3_comm_abort(c3_comm *c) {
    if (!c) return 0;
    if (c->nccl) {
        RcclApi &r = RcclApi::get();
        const ncclResult_t rc = r.CommAbort(c->nccl);
        c->nccl = nullptr;
        c->world = 1, c->rank = 0, c->aborted = true;  // whatever is asked of this handle from now on is local (orig_rank keeps its place in counts[])
        if (rc != ncclSuccess) return fail("ncclCommAbort failed: %s", r.GetErrorString(rc));
    }
    return 0;
}

int c3_stream_wait(void *stream, int device, int timeout_ms) {
    HIP_TRY(hipSetDevice(device));
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t e = hipStreamQuery((hipStream_t)stream);
        if (e == hipSuccess) return 0;
        if (e != hipErrorNotReady) return fail("hipStreamQuery: %s", hipGetErrorString(e));
        if (timeout_ms >= 0 &&
            std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() >= timeout_ms) {
            (void)hipGetLastError();
            g_err = "timeout";
            return 1;
        }
        struct timespec ts = {0, 50000};  // 50 us
        nanosleep(&ts, nullptr);
    }
}

}  // extern "C"

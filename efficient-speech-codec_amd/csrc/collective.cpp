// Multi-GPU exchange step of the sharded path (BASELINE configs[3]): ONE all-gather of the emitted codes over
// RCCL / xGMI.  The reference has no inference-time collective; clips are independent end to end, so this is the only exchange.
//
// libescx.so does not link RCCL: the library is resolved at run time (dlopen) so that the communicator the caller created and the
// ncclAllGather we call come from the SAME RCCL instance - in a PyTorch process that is torch's bundled librccl (already mapped,
// found by SONAME), in a plain C/C++ host it is the system one.  escx_set_rccl_library() overrides the name.
#include <dlfcn.h>
#include <stdlib.h>
#include <hip/hip_runtime.h>

#include <mutex>
#include <string>

#include "escx_internal.h"
#include "launchers.h"

using namespace escx;

namespace {
typedef int (*allgather_fn)(const void*, void*, size_t, int, void*, hipStream_t);
typedef const char* (*errstr_fn)(int);
std::mutex g_mu;
std::string g_name;                 // empty = default search order
void* g_lib = nullptr;
allgather_fn g_allgather = nullptr;
errstr_fn g_errstr = nullptr;

int resolve() {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_allgather) return 0;
    const char* defaults[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    if (!g_name.empty()) g_lib = dlopen(g_name.c_str(), RTLD_NOW | RTLD_LOCAL);
    else for (const char* n : defaults) if ((g_lib = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    if (!g_lib) ESCX_FAIL(ESCX_ERR_STATE, "RCCL library not found (%s): %s", g_name.empty() ? "librccl.so.1" : g_name.c_str(), dlerror());
    g_allgather = (allgather_fn)dlsym(g_lib, "ncclAllGather");
    g_errstr = (errstr_fn)dlsym(g_lib, "ncclGetErrorString");
    if (!g_allgather) ESCX_FAIL(ESCX_ERR_STATE, "ncclAllGather not exported by the RCCL library");
    return 0;
}
}  // namespace

extern "C" int escx_set_rccl_library(const char* path) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_allgather) {              // name the bound instance so that a caller can verify its communicator comes from the same one
        Dl_info info{};
        char real[4096];
        const char* bound = (dladdr((void*)g_allgather, &info) && info.dli_fname) ? info.dli_fname : "?";
        if (bound[0] == '/' && realpath(bound, real)) bound = real;
        ESCX_FAIL(ESCX_ERR_STATE, "RCCL is already resolved (bound: %s); call escx_set_rccl_library before the first collective", bound);
    }
    g_name = path ? path : "";
    return ESCX_OK;
}

// codes_local: (n_local_codes) int64 on this rank; codes_all: (world_size * n_local_codes) int64, rank order.  Codes are < 2^15
// (codebook_size <= 32768), so they cross the links as int16: 4x fewer bytes than the int64 the API carries.  RCCL moves bytes
// (it has no 16-bit integer type): ncclInt8 x 2n.  The payload at 36 clips x 6 x 3 x 150 codes is 194 KB per rank - latency
// bound, far below the ~153 GB/s of one xGMI link - so one direct collective, no pipelining.
extern "C" int escx_allgather_codes(escx_handle h, const int64_t* codes_local, int64_t n_local, int64_t* codes_all, int world_size,
                                    void* nccl_comm, void* stream) {
    if (!h || !codes_local || !codes_all || !nccl_comm) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "null argument");
    if (n_local < 1 || world_size < 1) ESCX_FAIL(ESCX_ERR_INVALID_ARG, "n_local_codes and world_size must be positive");
    int rc = resolve(); if (rc) return rc;
    ESCX_HIP(hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream;
    const size_t need = (size_t)(world_size + 1) * (size_t)n_local * sizeof(short);
    if (need > h->coll_cap) {                              // staging grows only; synchronous (like escx_reserve)
        ESCX_HIP(hipDeviceSynchronize());
        if (h->coll_buf) ESCX_HIP(hipFree(h->coll_buf));
        h->coll_buf = nullptr; h->coll_cap = 0;
        ESCX_HIP(hipMalloc(&h->coll_buf, need));
        h->coll_cap = need;
    }
    short* send = (short*)h->coll_buf;
    short* recv = send + n_local;
    codes_narrow((const long long*)codes_local, send, n_local, st);
    const int res = g_allgather(send, recv, (size_t)n_local * sizeof(short), /*ncclInt8*/ 0, nccl_comm, st);
    if (res != 0) ESCX_FAIL(ESCX_ERR_HIP, "ncclAllGather failed: %s", g_errstr ? g_errstr(res) : "unknown");
    codes_widen(recv, (long long*)codes_all, (long long)world_size * n_local, st);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) ESCX_FAIL(ESCX_ERR_HIP, "allgather_codes: kernel launch failed: %s", hipGetErrorString(e));
    return ESCX_OK;
}

// C ABI, part 4 (include/viprs_hip.h, "multi-GPU"): the RCCL communicator behind the scalar reductions of
// the EM iteration.  LD blocks shard over the GPUs of a node without any data-path exchange; per EM
// iteration the ranks combine one small float64 vector (M-step / ELBO partial sums, max |eta_diff|:
// VIPRS.py:426-484, :497-581, :997) -- ONE ncclAllGather over xGMI plus a rank-ordered reduction kernel,
// enqueued on the plan's stream right behind the local reduction kernels.
//
// librccl is opened with dlopen on first use: single-GPU processes never load it.
#include <dlfcn.h>
#include <rccl/rccl.h>          // types and prototypes only; the symbols are resolved at run time

#include <map>
#include "internal.h"

using namespace viprs;

namespace {

struct RcclApi {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclCommUserRank) CommUserRank = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;
};

RcclApi* rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {getenv("VIPRS_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            if (!n || !*n) continue;
            api.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (api.handle) break;
            api.error = dlerror();
        }
        if (!api.handle) return;
#define SYM(NAME)                                                                      \
        api.NAME = reinterpret_cast<decltype(api.NAME)>(dlsym(api.handle, "nccl" #NAME)); \
        if (!api.NAME) { api.error = "librccl: missing symbol nccl" #NAME; dlclose(api.handle); api.handle = nullptr; return; }
        SYM(GetUniqueId) SYM(CommInitRank) SYM(CommDestroy) SYM(AllGather) SYM(GetErrorString) SYM(CommCount) SYM(CommUserRank)
#undef SYM
    });
    return &api;
}

#define NCCL_TRY(expr)                                                                                   \
    do {                                                                                                 \
        ncclResult_t _r = (expr);                                                                        \
        if (_r != ncclSuccess)                                                                           \
            return ::viprs::fail(VIPRS_EDEVICE, std::string(#expr) + ": " + rccl()->GetErrorString(_r)); \
    } while (0)

// out[i] = reduction over ranks r = 0 .. world-1, IN RANK ORDER, of gathered[r * n + i]: a sum, except for the
// last element of every `group` (and for every element when group == -1), which is a maximum
__global__ void comm_reduce_kernel(const double* __restrict__ gathered, int world, int n, int group,
                                   double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bool is_max = group == -1 || (group > 0 && (i % group) == group - 1);
    double a = gathered[i];
    for (int r = 1; r < world; ++r) {
        const double v = gathered[(int64_t)r * n + i];
        a = is_max ? fmax(a, v) : a + v;
    }
    out[i] = a;
}

}  // namespace

struct viprs_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    hipStream_t stream = nullptr;                  // host-vector collectives (viprs_comm_allreduce / barrier)
    DevBuf<double> d_vec;
    // one all-gather landing buffer PER STREAM that reduces through this communicator (the plan streams of the states
    // it is attached to, and `stream` above): reductions on different streams are not ordered against each other
    std::map<hipStream_t, DevBuf<double>> d_gather;
    double* h_pin = nullptr;
    size_t h_cap = 0;
    ~viprs_comm() {
        if (h_pin) (void)hipHostFree(h_pin);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

namespace viprs {

// all-gather + ordered reduction of the device vector `d_vec` (n doubles, in place) on `stream`
int comm_reduce_on_stream(viprs_comm* C, double* d_vec, int n, int group, hipStream_t stream) {
    if (!C || n <= 0) return VIPRS_OK;
    const size_t need = (size_t)C->world * (size_t)n;
    DevBuf<double>& gather = C->d_gather[stream];
    if (gather.n < need) {
        HIP_TRY(hipStreamSynchronize(stream));     // (growth only: an earlier reduction on this stream may still read it)
        HIP_TRY(gather.alloc(std::max<size_t>(need, 4096)));
    }
    NCCL_TRY(rccl()->AllGather(d_vec, gather.p, (size_t)n, ncclDouble, C->comm, stream));
    comm_reduce_kernel<<<(n + 255) / 256, 256, 0, stream>>>(gather.p, C->world, n, group, d_vec);
    HIP_TRY(hipGetLastError());
    return VIPRS_OK;
}

}  // namespace viprs

extern "C" {

int viprs_comm_unique_id(void* id) {
    if (!id) return fail(VIPRS_EINVAL, "null id buffer");
    RcclApi* api = rccl();
    if (!api->handle) return fail(VIPRS_EUNSUPPORTED, "librccl could not be loaded: " + api->error);
    static_assert(sizeof(ncclUniqueId) == VIPRS_COMM_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId uid;
    NCCL_TRY(api->GetUniqueId(&uid));
    std::memcpy(id, &uid, sizeof(uid));
    return VIPRS_OK;
}

int viprs_comm_create(viprs_comm** out, const void* id, int rank, int world_size, int device) {
    if (!out || !id) return fail(VIPRS_EINVAL, "null argument");
    *out = nullptr;
    if (world_size < 1 || rank < 0 || rank >= world_size) return fail(VIPRS_EINVAL, "bad rank / world size");
    RcclApi* api = rccl();
    if (!api->handle) return fail(VIPRS_EUNSUPPORTED, "librccl could not be loaded: " + api->error);
    HIP_TRY(hipSetDevice(device));
    std::unique_ptr<viprs_comm> C(new viprs_comm());
    C->rank = rank;
    C->world = world_size;
    C->device = device;
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof(uid));
    NCCL_TRY(api->CommInitRank(&C->comm, world_size, uid, rank));
    HIP_TRY(hipStreamCreateWithFlags(&C->stream, hipStreamNonBlocking));
    *out = C.release();
    return VIPRS_OK;
}

int viprs_comm_destroy(viprs_comm* C) {
    if (!C) return VIPRS_OK;
    (void)hipSetDevice(C->device);
    (void)hipDeviceSynchronize();
    if (C->comm) (void)rccl()->CommDestroy(C->comm);
    delete C;
    return VIPRS_OK;
}

int viprs_comm_rank(const viprs_comm* C, int* rank, int* world_size) {
    // what RCCL itself says about the communicator (not what the caller passed to viprs_comm_create): a launcher
    // that started fewer processes than it claims shows up here
    if (!C) return fail(VIPRS_EINVAL, "null communicator");
    int r = C->rank, w = C->world;
    if (C->comm) {
        NCCL_TRY(rccl()->CommUserRank(C->comm, &r));
        NCCL_TRY(rccl()->CommCount(C->comm, &w));
    }
    if (rank) *rank = r;
    if (world_size) *world_size = w;
    return VIPRS_OK;
}

int viprs_comm_allreduce(viprs_comm* C, double* vec, int n, int group) {
    if (!C || (!vec && n > 0)) return fail(VIPRS_EINVAL, "null argument");
    if (n <= 0) return VIPRS_OK;
    HIP_TRY(hipSetDevice(C->device));
    if (C->h_cap < (size_t)n) {
        if (C->h_pin) HIP_TRY(hipHostFree(C->h_pin));
        C->h_pin = nullptr;
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&C->h_pin), (size_t)n * sizeof(double), hipHostMallocDefault));
        C->h_cap = (size_t)n;
    }
    if (C->d_vec.n < (size_t)n) HIP_TRY(C->d_vec.alloc(std::max<size_t>((size_t)n, 512)));
    std::memcpy(C->h_pin, vec, (size_t)n * sizeof(double));
    HIP_TRY(hipMemcpyAsync(C->d_vec.p, C->h_pin, (size_t)n * sizeof(double), hipMemcpyHostToDevice, C->stream));
    int rc = comm_reduce_on_stream(C, C->d_vec.p, n, group, C->stream);
    if (rc != VIPRS_OK) return rc;
    HIP_TRY(hipMemcpyAsync(C->h_pin, C->d_vec.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, C->stream));
    HIP_TRY(hipStreamSynchronize(C->stream));
    std::memcpy(vec, C->h_pin, (size_t)n * sizeof(double));
    return VIPRS_OK;
}

int viprs_comm_allgather(viprs_comm* C, const double* send, int64_t n, double* recv) {
    // bulk exchange of per-SNP vectors (the posterior of every rank's SNPs when fit() returns): ONE ncclAllGather of
    // n doubles per rank; `recv` holds world_size x n doubles in rank order.  Not an EM-iteration collective.
    if (!C || ((!send || !recv) && n > 0)) return fail(VIPRS_EINVAL, "null argument");
    if (n <= 0) return VIPRS_OK;
    HIP_TRY(hipSetDevice(C->device));
    DevBuf<double> d_send, d_recv;
    HIP_TRY(d_send.alloc((size_t)n));
    HIP_TRY(d_recv.alloc((size_t)n * (size_t)C->world));
    HIP_TRY(hipMemcpyAsync(d_send.p, send, (size_t)n * sizeof(double), hipMemcpyHostToDevice, C->stream));
    NCCL_TRY(rccl()->AllGather(d_send.p, d_recv.p, (size_t)n, ncclDouble, C->comm, C->stream));
    HIP_TRY(hipMemcpyAsync(recv, d_recv.p, (size_t)n * (size_t)C->world * sizeof(double), hipMemcpyDeviceToHost, C->stream));
    HIP_TRY(hipStreamSynchronize(C->stream));
    return VIPRS_OK;
}

int viprs_comm_barrier(viprs_comm* C) {
    if (!C) return fail(VIPRS_EINVAL, "null communicator");
    HIP_TRY(hipSetDevice(C->device));
    HIP_TRY(hipDeviceSynchronize());
    double one = 1.0;
    return viprs_comm_allreduce(C, &one, 1, 0);
}

int viprs_state_set_comm(viprs_state* S, viprs_comm* C) {
    if (!S) return fail(VIPRS_EINVAL, "null state");
    if (C && C->device != S->plan->device) return fail(VIPRS_EINVAL, "communicator and plan live on different devices");
    S->comm = C;
    return VIPRS_OK;
}

int viprs_device_synchronize(int device) {
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipDeviceSynchronize());
    return VIPRS_OK;
}

}  // extern "C"

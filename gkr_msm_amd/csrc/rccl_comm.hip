// Native multi-GPU exchange: RCCL over xGMI behind the gm_comm seam.
//
// The path shards by MSM window: rank g owns the bucket rows of its windows (pushforward.rs:401); what crosses GPUs is small
// and latency-bound -- the window points once per MSM (27 KB at config B), the 2-3 partial round sums per sumcheck round
// (<= 96 B per rank), the bucket sums once per proof (0.75 MiB) -- plus one large transfer, the replication of the operands
// (ncclBroadcast, link-bound: 1.5 GiB at x_logsize = 24).  All of it is ncclAllGather / ncclBroadcast on the caller's HIP
// stream; nothing is reduced by RCCL (field and curve additions are not RCCL ops: every rank adds the gathered parts itself,
// which is exact and order-independent).
//
// RCCL is bound at run time (dlopen) so that single-GPU users and the CPU-side tests never load it; when the process already
// has a copy (PyTorch bundles one) that copy is used, which also keeps one RCCL per process.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <memory>

#include "internal.hpp"

namespace gm {
namespace {

struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
    char why[256] = {0};        // what failed, captured at the failing dlopen / dlsym
};

RcclApi& rccl_api() {
    static RcclApi a = [] {
        RcclApi r;
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names)  // a copy already in the process (PyTorch's) first
            if ((r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
        if (!r.lib)
            for (const char* n : names)
                if ((r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!r.lib) {
            const char* e = dlerror();
            snprintf(r.why, sizeof(r.why), "dlopen librccl.so.1 failed: %s", e ? e : "(no dlerror)");
            return r;
        }
        r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.lib, "ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.lib, "ncclCommInitRank");
        r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
        r.AllGather = (decltype(r.AllGather))dlsym(r.lib, "ncclAllGather");
        r.Broadcast = (decltype(r.Broadcast))dlsym(r.lib, "ncclBroadcast");
        r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather && r.Broadcast && r.GetErrorString;
        if (!r.ok) snprintf(r.why, sizeof(r.why), "librccl.so.1 loaded but a symbol is missing (ncclGetUniqueId/CommInitRank/CommDestroy/AllGather/Broadcast/GetErrorString)");
        return r;
    }();
    return a;
}

#define GM_NCCL(call)                                                                                              \
    do {                                                                                                           \
        ncclResult_t r__ = (call);                                                                                 \
        if (r__ != ncclSuccess)                                                                                    \
            return gm::set_err(GM_ERR_HIP, "%s:%d %s -> %s", __FILE__, __LINE__, #call, rccl_api().GetErrorString(r__)); \
    } while (0)

int32_t need_rccl() {
    if (!rccl_api().ok) return set_err(GM_ERR_STATE, "RCCL is not available in this process (%s)", rccl_api().why);
    return GM_OK;
}

}  // namespace
}  // namespace gm

using namespace gm;

struct gm_rccl {
    ncclComm_t comm = nullptr;
    uint32_t rank = 0, world = 1;
    int dev = 0;
    hipStream_t stream = nullptr;
    DevBuf send, recv;          // device staging of the host-buffer all-gather (grown on demand)
    char* pinned = nullptr;     // pinned host staging for small payloads (world * PIN_BYTES)
    static constexpr size_t PIN_BYTES = 64 << 10;
    uint64_t calls = 0, bytes = 0, dev_calls = 0;
    ~gm_rccl() {
        if (pinned) (void)hipHostFree(pinned);
        if (comm && rccl_api().ok) (void)rccl_api().CommDestroy(comm);
    }
};

// gm_comm::all_gather over RCCL: buf = world * nbytes host bytes, this rank's part already in place
static int32_t rccl_all_gather_host(void* ctx, void* buf, uint64_t nbytes) {
    gm_rccl* r = static_cast<gm_rccl*>(ctx);
    if (!r || !buf) return 1;
    if (nbytes == 0) return 0;
    const size_t total = (size_t)r->world * nbytes;
    {
        // the staging outlives any prover arena: a gather issued inside a layer's ArenaScope must not carve it from there
        // (the arena is reset after the layer and would hand the same bytes to the next layer's columns)
        ArenaScope none(nullptr);
        if (r->send.bytes < nbytes && r->send.alloc(nbytes < 4096 ? 4096 : nbytes)) return 2;
        if (r->recv.bytes < total && r->recv.alloc(total < 65536 ? 65536 : total)) return 2;
    }
    char* hb = static_cast<char*>(buf);
    const bool small = nbytes <= gm_rccl::PIN_BYTES && r->pinned;
    char* stage = small ? r->pinned : hb;   // pinned staging keeps the two copies asynchronous and short
    if (small) memcpy(stage + (size_t)r->rank * nbytes, hb + (size_t)r->rank * nbytes, nbytes);
    if (hipMemcpyAsync(r->send.p, stage + (size_t)r->rank * nbytes, nbytes, hipMemcpyHostToDevice, r->stream) != hipSuccess) return 3;
    if (rccl_api().AllGather(r->send.p, r->recv.p, nbytes, ncclUint8, r->comm, r->stream) != ncclSuccess) return 4;
    if (hipMemcpyAsync(stage, r->recv.p, total, hipMemcpyDeviceToHost, r->stream) != hipSuccess) return 5;
    if (hipStreamSynchronize(r->stream) != hipSuccess) return 6;
    if (small) memcpy(hb, stage, total);
    r->calls++;
    r->bytes += nbytes;
    return 0;
}

// gm_comm::all_gather_dev over RCCL: asynchronous on the caller's stream
static int32_t rccl_all_gather_dev_cb(void* ctx, const void* d_send, void* d_recv, uint64_t nbytes, void* stream) {
    gm_rccl* r = static_cast<gm_rccl*>(ctx);
    if (!r || !d_send || !d_recv) return 1;
    if (nbytes == 0) return 0;
    if (rccl_api().AllGather(d_send, d_recv, nbytes, ncclUint8, r->comm, as_stream(stream)) != ncclSuccess) return 4;
    r->dev_calls++;
    return 0;
}

extern "C" {

// rank 0 makes the id; the caller hands the 128 bytes to every rank over whatever side channel it has (the reference's
// process launcher, torch.distributed's store, MPI ...)
int32_t gm_comm_rccl_unique_id(uint8_t* out_id128) {
    GM_REQUIRE(out_id128, "null argument");
    int32_t rc = need_rccl();
    if (rc) return rc;
    ncclUniqueId id;
    GM_NCCL(rccl_api().GetUniqueId(&id));
    static_assert(sizeof(id) == GM_RCCL_UNIQUE_ID_BYTES, "ncclUniqueId size");
    memcpy(out_id128, &id, sizeof(id));
    return GM_OK;
}

// one communicator per process on the CURRENT device (gm_set_device first); collective: every rank calls it
int32_t gm_comm_rccl_create(const uint8_t* id128, uint32_t rank, uint32_t world, gm_rccl** out, void* stream) {
    GM_REQUIRE(id128 && out && world >= 1 && rank < world, "bad argument");
    int32_t rc = need_rccl();
    if (rc) return rc;
    std::unique_ptr<gm_rccl> r(new gm_rccl());
    r->rank = rank; r->world = world; r->stream = as_stream(stream);
    GM_HIP(hipGetDevice(&r->dev));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    GM_NCCL(rccl_api().CommInitRank(&r->comm, (int)world, id, (int)rank));
    GM_HIP(hipHostMalloc((void**)&r->pinned, (size_t)world * gm_rccl::PIN_BYTES, hipHostMallocDefault));
    *out = r.release();
    return GM_OK;
}

int32_t gm_comm_rccl_destroy(gm_rccl* r) {
    delete r;
    return GM_OK;
}

// the gm_comm the sharded prover takes (gm_pip_witness_create_sharded): its all_gather runs ncclAllGather on a device
// staging buffer on the communicator's stream -- no callback into the caller's language
int32_t gm_comm_rccl_as_comm(gm_rccl* r, gm_comm* out) {
    GM_REQUIRE(r && out, "null argument");
    out->ctx = r;
    out->rank = r->rank;
    out->world = r->world;
    out->all_gather = rccl_all_gather_host;
    out->all_gather_dev = rccl_all_gather_dev_cb;
    out->pull_dev = nullptr;   // (pairwise ncclSend / ncclRecv would serve; this box cannot rehearse them with more than one rank)
    return GM_OK;
}

// device-to-device collectives on the caller's stream (asynchronous, as every RCCL call)
int32_t gm_comm_rccl_all_gather_dev(gm_rccl* r, const void* d_send, void* d_recv, uint64_t bytes_per_rank, void* stream) {
    GM_REQUIRE(r && d_send && d_recv, "null argument");
    if (bytes_per_rank == 0) return GM_OK;
    GM_NCCL(rccl_api().AllGather(d_send, d_recv, bytes_per_rank, ncclUint8, r->comm, as_stream(stream)));
    return GM_OK;
}

// operand replication: `root`'s buffer to every rank (points and scalars; the one link-bound transfer of the path)
int32_t gm_comm_rccl_broadcast_dev(gm_rccl* r, void* d_buf, uint64_t bytes, uint32_t root, void* stream) {
    GM_REQUIRE(r && d_buf && root < r->world, "bad argument");
    if (bytes == 0) return GM_OK;
    GM_NCCL(rccl_api().Broadcast(d_buf, d_buf, bytes, ncclUint8, (int)root, r->comm, as_stream(stream)));
    return GM_OK;
}

int32_t gm_comm_rccl_stats(const gm_rccl* r, uint64_t* host_all_gathers, uint64_t* bytes_per_rank_total) {
    GM_REQUIRE(r, "null argument");
    if (host_all_gathers) *host_all_gathers = r->calls + r->dev_calls;   // exchanges of either form
    if (bytes_per_rank_total) *bytes_per_rank_total = r->bytes;
    return GM_OK;
}

}  // extern "C"

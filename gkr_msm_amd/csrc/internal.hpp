// Cross-translation-unit declarations inside libgkrmsm_hip.so (not part of the ABI).
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

#include "common.hpp"
#include "segfn.hip.h"

namespace gm {

int32_t to_gmfn(const gm_fn* f, GmFn* g);
int32_t launch_dense_map(const SegPlan& sp, const Fr* const* in, Fr* const* out, uint64_t n, hipStream_t s);
int32_t launch_dense_map_split(const SegPlan& sp, const Fr* const* in, Fr* const* out, uint64_t n, uint32_t lo_bit,
                               uint32_t bundle, hipStream_t s);
int32_t launch_dense_fold(const Fr* const* in, Fr* const* out, int k, uint64_t n_out, const Fr& t, hipStream_t s);
// extra: up to 16 field elements the same launch stores at extra_dst (a layer's gamma powers: one launch less)
int32_t launch_eq_sequence(const Fr& mult, const Fr* pt, uint32_t nvars, Fr* const* levels, hipStream_t s, const Fr* extra = nullptr,
                           uint32_t n_extra = 0, Fr* extra_dst = nullptr);
// two small sequences + a few scalars in one launch (poly.hip); false = does not fit, use launch_eq_sequence
bool launch_eq_pair(const Fr& mult0, const Fr* pt0, uint32_t nvars0, Fr* const* levels0, const Fr& mult1, const Fr* pt1, uint32_t nvars1,
                    Fr* const* levels1, const Fr* scal, uint32_t n_scal, Fr* scal_dst, hipStream_t s, const Fr* extra = nullptr,
                    uint32_t n_extra = 0, Fr* extra_dst = nullptr);

int32_t launch_offsets_next(const uint32_t* off_in, uint32_t* off_out, uint32_t nrows, hipStream_t s);
int32_t launch_offsets_all_from_off(const uint32_t* off0, uint32_t* off_all, uint32_t nrows, uint32_t nlevels, hipStream_t s);
int32_t launch_offsets_from_len(const uint32_t* len, uint32_t* off, uint32_t nrows, hipStream_t s);

// ---- host-side univariate helpers (liblasso UniPoly::from_evals = interpolation on 0..D; un-vendored
// dependency, git 925a7a74; call sites sumcheck.rs:220,327, vecvec_eq.rs:209)
std::vector<Fr> unipoly_from_evals(const std::vector<Fr>& evals);
Fr evaluate_univar(const std::vector<Fr>& coeffs, const Fr& x);                 // sumcheck.rs:33-44
std::vector<Fr> from12(const Fr& p1, const Fr& p2, const Fr& eq1, const Fr& prev_claim);  // vecvec_eq.rs:197-216
Fr eq_bind_factor(const Fr& q, const Fr& t);                                    // 1 - q - t + 2qt
Fr eq_sum_host(const Fr* pt, uint32_t n, uint64_t k);                           // utils.rs:265-291

// ---- sharding context: which slice of the bucket rows this process owns and how to reach the other ranks.
// Sumcheck objects capture the current one at creation (the drivers set it around the sharded layers).
// bound of every wait on the other side (kernels waiting for a challenge, host loops waiting for results or for another rank):
// gm_set_wait_timeout_ms, default 20 s.  Device side: wall_clock64() ticks (100 MHz).
inline std::atomic<uint32_t>& wait_timeout_ms() {
    static std::atomic<uint32_t> v{20000};
    return v;
}
inline uint64_t wait_timeout_ticks() { return (uint64_t)wait_timeout_ms().load() * 100000ull; }
inline std::chrono::milliseconds wait_timeout_host() { return std::chrono::milliseconds(wait_timeout_ms().load()); }

// Where the wall time of a sharded call goes on this rank's host thread (thread-local; gm_shard_clock reads / resets it):
// `small` = host all-gathers of <= 4 KiB (the per-round sums, agreement words, group elements: mostly WAITING for the slowest rank),
// `bulk` = larger host all-gathers (bucket sums, host-staged redistributions), `pull` = gm_comm::pull_dev calls (device to device).
struct ShardClock {
    double small_us = 0, bulk_us = 0, pull_us = 0;
    uint64_t small_n = 0, bulk_n = 0, pull_n = 0, bulk_bytes = 0, pull_bytes = 0;
    static ShardClock& get() { static thread_local ShardClock c; return c; }
    static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    void gather(double t0, uint64_t bytes) {
        const double dt = now() - t0;
        if (bytes <= 4096) { small_us += dt; small_n++; } else { bulk_us += dt; bulk_n++; bulk_bytes += bytes; }
    }
    void pulled(double t0, uint64_t bytes) { pull_us += now() - t0; pull_n++; pull_bytes += bytes; }
};
// every call of a gm_comm's collectives inside the library goes through these two
inline int32_t comm_all_gather(const gm_comm* c, void* h_buf, uint64_t bytes_per_rank) {
    const double t0 = ShardClock::now();
    const int32_t rc = c->all_gather(c->ctx, h_buf, bytes_per_rank);
    ShardClock::get().gather(t0, bytes_per_rank);
    return rc;
}
inline int32_t comm_pull_dev(const gm_comm* c, const void* d_src, uint64_t src_bytes, uint32_t n, const gm_pull* pieces, void* stream) {
    const double t0 = ShardClock::now();
    const int32_t rc = c->pull_dev(c->ctx, d_src, src_bytes, n, pieces, stream);
    uint64_t b = 0;
    for (uint32_t i = 0; i < n; i++) b += pieces[i].bytes;
    ShardClock::get().pulled(t0, b);
    return rc;
}

struct Shard {
    const gm_comm* comm = nullptr;  // nullptr: unsharded
    uint32_t rank = 0, world = 1, lg = 0;
};
inline Shard& current_shard() {
    static thread_local Shard s;
    return s;
}
struct ShardScope {
    Shard prev;
    explicit ShardScope(const Shard& s) : prev(current_shard()) { current_shard() = s; }
    ~ShardScope() { current_shard() = prev; }
};
// all ranks' copies of `bytes` bytes, rank-major
inline int32_t shard_all_gather(const Shard& sh, const void* mine, size_t bytes, std::vector<char>* all) {
    all->assign((size_t)sh.world * bytes, 0);
    memcpy(all->data() + (size_t)sh.rank * bytes, mine, bytes);
    const int32_t rc = comm_all_gather(sh.comm, all->data(), bytes);
    if (rc) return set_err(GM_ERR_STATE, "gm_comm all_gather failed with %d", rc);
    return GM_OK;
}
// vals[i] <- sum over ranks of vals[i]   (field addition is exact: the order of the ranks does not matter)
inline int32_t shard_sum_fr(const Shard& sh, Fr* vals, int n) {
    std::vector<char> all;
    const int32_t rc = shard_all_gather(sh, vals, (size_t)n * sizeof(Fr), &all);
    if (rc) return rc;
    const Fr* a = reinterpret_cast<const Fr*>(all.data());
    for (int i = 0; i < n; i++) {
        Fr s = fr_zero();
        for (uint32_t r = 0; r < sh.world; r++) s = fr_add(s, a[(size_t)r * n + i]);
        vals[i] = s;
    }
    return GM_OK;
}

// ---- a rank's view of the KZG key (the key has 2^(x + clm + 1) - 1 points -- 51 GB at x_logsize 24, clm 4 --
// and no rank needs it whole): the segments of kzg_basis() that are resident on this rank's device.  Every sharded G1 step asks for
// the range it needs and fails by name when it is not there.
inline const uint64_t* key_range(const gm_key_view* kv, uint64_t first, uint64_t count) {
    if (!kv) return nullptr;
    for (uint32_t i = 0; i < kv->n_segments; i++)
        if (first >= kv->first[i] && first + count <= kv->first[i] + kv->count[i]) return kv->d_segment[i] + 12 * (first - kv->first[i]);
    return nullptr;
}
#define GM_KEY_RANGE(var, kv, first, count, what)                                                                            \
    const uint64_t* var = gm::key_range(kv, first, count);                                                                   \
    if (!var && (count))                                                                                                     \
        return gm::set_err(GM_ERR_INVALID, "%s: key points [%llu, %llu) are not resident on this rank (gm_key_view)", what, \
                           (unsigned long long)(first), (unsigned long long)((first) + (count)))

// The bulk moves of the sharded provers, one collective call: every rank exposes `src_elems` field elements at d_src and fetches
// `pieces` (gm_pull: peer, byte offset into that peer's source, bytes, destination) -- device to device through gm_comm::pull_dev
// when the communicator has it; through the host all-gather (every rank's whole source) otherwise, or once pull_dev has answered
// "unavailable" (100: the same answer on every rank; *host_staged is sticky so that later calls do not ask again).
inline int32_t shard_pull(const Shard& sh, const Fr* d_src, uint64_t src_elems, const std::vector<gm_pull>& pieces, bool* host_staged,
                          hipStream_t s) {
    if (sh.comm->pull_dev && !*host_staged) {
        const int32_t rc = comm_pull_dev(sh.comm, d_src, src_elems * sizeof(Fr), (uint32_t)pieces.size(), pieces.data(), reinterpret_cast<void*>(s));
        if (rc == 0) return GM_OK;
        if (rc != 100) return set_err(GM_ERR_STATE, "gm_comm pull_dev failed with %d", rc);
    }
    *host_staged = true;
    std::vector<Fr> all((size_t)sh.world * src_elems);
    GM_HIP(hipMemcpyAsync(all.data() + (size_t)sh.rank * src_elems, d_src, src_elems * sizeof(Fr), hipMemcpyDeviceToHost, s));
    GM_HIP(hipStreamSynchronize(s));
    const int32_t rc = comm_all_gather(sh.comm, all.data(), src_elems * sizeof(Fr));
    if (rc) return set_err(GM_ERR_STATE, "gm_comm all_gather failed with %d", rc);
    for (const gm_pull& p : pieces)
        GM_HIP(hipMemcpyAsync(p.d_dst, all.data() + (size_t)p.peer * src_elems + p.src_offset / sizeof(Fr), p.bytes, hipMemcpyHostToDevice, s));
    GM_HIP(hipStreamSynchronize(s));   // `all` goes out of scope
    return GM_OK;
}

// Read elements [lo, lo + count) of an array that is distributed over the ranks in contiguous slices of S elements (rank q holds
// [q S, (q + 1) S)) into d_dst; indices outside [0, world * S) read as zero (lo may be negative).  Collective: every rank calls it
// with its own slice as d_src and its own (lo, count) -- count = 0 included.
inline int32_t dist_read(const Shard& sh, const Fr* d_src, uint64_t S, int64_t lo, uint64_t count, Fr* d_dst, bool* host_staged, hipStream_t s) {
    if (count) GM_HIP(hipMemsetAsync(d_dst, 0, count * sizeof(Fr), s));
    std::vector<gm_pull> pc;
    for (uint32_t q = 0; q < sh.world; q++) {
        const int64_t a = std::max<int64_t>(lo, (int64_t)q * (int64_t)S), b = std::min<int64_t>(lo + (int64_t)count, ((int64_t)q + 1) * (int64_t)S);
        if (a >= b) continue;
        pc.push_back(gm_pull{q, 0u, (uint64_t)(a - (int64_t)q * (int64_t)S) * sizeof(Fr), (uint64_t)(b - a) * sizeof(Fr), d_dst + (a - lo)});
    }
    return shard_pull(sh, d_src, S, pc, host_staged, s);
}

// Bump allocator over one device allocation.  The image-part prover creates ~65 short-lived sumcheck objects;
// hipMalloc/hipFree per buffer (hipFree synchronises the device) dominated the per-round cost, so the driver opens
// an ArenaScope around each layer and resets the arena afterwards.  DevBuf::alloc carves from the current arena
// when one is active on the calling thread, otherwise it owns a hipMalloc.
struct Arena {
    char* base = nullptr;
    size_t cap = 0, used = 0, high = 0;
    int32_t init(size_t bytes) {
        hipError_t e = dev_alloc((void**)&base, bytes);
        if (e != hipSuccess) return set_err(GM_ERR_HIP, "hipMalloc(arena %zu): %s", bytes, hipGetErrorString(e));
        cap = bytes;
        return GM_OK;
    }
    void* carve(size_t b) {
        const size_t a = (used + 255) & ~(size_t)255;
        if (a + b > cap) return nullptr;
        used = a + b;
        if (used > high) high = used;
        return base + a;
    }
    void reset() { used = 0; }
    ~Arena() { if (base) dev_free(base); }
};
inline Arena*& current_arena() {
    static thread_local Arena* a = nullptr;
    return a;
}
inline Fr*& shared_pinned() {  // pinned host staging (>= 8 Fr) shared by the round objects of one driver
    static thread_local Fr* p = nullptr;
    return p;
}
// true while at most one sumcheck object at a time uses the shared pinned staging (the drivers' normal case); objects then may
// pre-enqueue kernels that report through it.  The combined sumcheck of the pushforward argument runs two objects in
// lock-step on one stream and clears it for that stretch.
inline bool& pinned_exclusive() {
    static thread_local bool v = true;
    return v;
}
struct PinnedSharedScope {
    bool prev;
    PinnedSharedScope() : prev(pinned_exclusive()) { pinned_exclusive() = false; }
    ~PinnedSharedScope() { pinned_exclusive() = prev; }
};
// The round objects of a driver report through 16 field elements of pinned, device-visible host memory.  One buffer per (host thread,
// device), allocated on first use and kept: hipHostMalloc / hipHostFree per call cost ~100 us each, and a free may wait for the whole
// device -- including another rank-thread's kernel that is itself waiting for this thread (ranks as threads of one process).
inline int32_t thread_pinned_staging(Fr** out) {
    static thread_local Fr* per_dev[GM_MAX_DEVICES] = {nullptr};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= GM_MAX_DEVICES) return set_err(GM_ERR_INVALID, "device id %d: the per-device tables of this library hold %d devices", dev, GM_MAX_DEVICES);
    if (!per_dev[dev]) {
        hipError_t e = hipHostMalloc((void**)&per_dev[dev], 16 * sizeof(Fr), hipHostMallocCoherent | hipHostMallocMapped);
        if (e != hipSuccess) return set_err(GM_ERR_HIP, "hipHostMalloc(pinned staging): %s", hipGetErrorString(e));
        memset(per_dev[dev], 0, 16 * sizeof(Fr));
    }
    *out = per_dev[dev];
    return GM_OK;
}
struct SharedPinnedScope {   // shared_pinned() = the thread's staging for the life of the scope
    Fr* prev;
    explicit SharedPinnedScope(Fr* p) : prev(shared_pinned()) { shared_pinned() = p; }
    ~SharedPinnedScope() { shared_pinned() = prev; }
};
struct ArenaScope {
    Arena* prev;
    explicit ArenaScope(Arena* a) : prev(current_arena()) { current_arena() = a; }
    ~ArenaScope() { current_arena() = prev; }
};

// device buffer owned by a handle (or borrowed from the current arena)
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    bool owned = true;
    int32_t alloc(size_t b) {
        free();
        if (b == 0) b = 32;
        if (Arena* a = current_arena()) {
            p = a->carve(b);
            if (!p) return set_err(GM_ERR_HIP, "workspace arena exhausted (%zu + %zu > %zu)", a->used, b, a->cap);
            owned = false;
            bytes = b;
            return GM_OK;
        }
        hipError_t e = dev_alloc(&p, b);
        if (e != hipSuccess) return set_err(GM_ERR_HIP, "hipMalloc(%zu): %s", b, hipGetErrorString(e));
        owned = true;
        bytes = b;
        return GM_OK;
    }
    void free() {
        if (p && owned) dev_free(p);
        p = nullptr;
        bytes = 0;
    }
    ~DevBuf() { free(); }
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    Fr* fr() const { return reinterpret_cast<Fr*>(p); }
};

// Host-side combination loops (a few thousand G1 operations per call, ~0.5 us each) spread over a handful of threads.
template <class F>
inline void host_parallel_for(uint32_t n, F&& body, uint32_t grain = 16) {
    unsigned hw = std::thread::hardware_concurrency();
    uint32_t nt = hw ? (hw > 12 ? 12 : hw) : 4;
    if (n < 4 * grain || nt < 2) {
        for (uint32_t i = 0; i < n; i++) body(i);
        return;
    }
    if (nt > n / grain) nt = n / grain ? n / grain : 1;
    std::vector<std::thread> th;
    const uint32_t per = (n + nt - 1) / nt;
    for (uint32_t t = 1; t < nt; t++)
        th.emplace_back([&, t] {
            const uint32_t hi = (t + 1) * per < n ? (t + 1) * per : n;
            for (uint32_t i = t * per; i < hi; i++) body(i);
        });
    for (uint32_t i = 0; i < per && i < n; i++) body(i);
    for (auto& x : th) x.join();
}


}  // namespace gm

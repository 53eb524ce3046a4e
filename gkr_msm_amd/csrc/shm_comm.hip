// One-node backing of the gm_comm seam over POSIX shared memory.
//
// What the sharded provers exchange every sumcheck round is 2-3 field elements per rank (<= 96 bytes), and it is the HOST that needs
// them: the round's sums go into the Fiat-Shamir transcript, which lives on the caller's side of the ABI, and the challenge comes
// back from there.  A device-side collective (ncclAllGather: a kernel launch + a ring over xGMI, ~20-30 us for 96 bytes) puts that
// latency into each of the ~1500 rounds of a proof and keeps the rounds from being pre-enqueued or run inside the persistent stage
// kernel.  All ranks of the path live on one node (one process per GPU, as the reference's rayon pool lives in one process), so the
// small exchanges go where the data is needed anyway: every rank's host thread posts its sums in a shared-memory slot and reads
// the others' -- a cache-line transfer between cores, well under a microsecond -- and the device side of a sharded round is exactly
// the unsharded one (same kernels, same pre-enqueued folds, same k_stage).  The transfers that carry volume (operand replication,
// window points) stay on RCCL (rccl_comm.hip); the once-per-proof bucket sums (0.75 MiB at config B) fit through here in chunks.
//
// Layout of the object: a header line, then per rank one sequence word (a cache line of its own) and two data slots of SLOT bytes.
// all_gather number n (chunk by chunk for payloads above SLOT): write slot n & 1, store-release seq[rank] = n, then for every other
// rank wait for seq >= n (load-acquire) and copy its slot n & 1.  A rank can be at most one call ahead of the slowest one (call n + 1
// needs everybody's n), so slot n & 1 is rewritten only after every rank has finished reading call n.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <mutex>
#include <cerrno>
#include <cstring>
#include <chrono>
#include <memory>
#include <string>
#include <vector>
#include <thread>

#include "internal.hpp"

using namespace gm;

namespace {
constexpr size_t SHM_SLOT = 256 << 10;       // bytes per rank and slot
constexpr uint64_t SHM_MAGIC = 0x676d73686d303031ull;   // "gmshm001"
struct ShmHeader {
    std::atomic<uint64_t> magic;   // set by rank 0 once the object has its size
    uint32_t world;
    uint32_t pad;
    std::atomic<uint32_t> attached;   // ranks that have mapped the object
};
static_assert(sizeof(ShmHeader) <= 64, "header fits its line");
size_t shm_bytes(uint32_t world) { return 64 + (size_t)world * 64 + (size_t)world * 2 * SHM_SLOT; }
}  // namespace

// same-device copies of pull_dev (see ShmIpcMsg::bus): 16 bytes per lane and step, grid-stride
__global__ void __launch_bounds__(256) k_shm_copy16(const uint4* __restrict__ src, uint4* __restrict__ dst, uint64_t n16) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
// the same with SYSTEM-coherent loads (sc0 sc1: past this device's caches): GM_SHM_COPY_SC=1, an experiment for DESIGN section 6
__global__ void __launch_bounds__(256) k_shm_copy16_sc(const uint4* __restrict__ src, uint4* __restrict__ dst, uint64_t n16) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) {
        uint4 v;
        asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(src + i) : "memory");
        dst[i] = v;
    }
}

// HIP IPC mappings of peer processes' allocations, kept for the life of the PROCESS and shared by all its communicators.
// The same few pool blocks / arenas come back pull after pull and proof after proof, and opening one costs ~a millisecond.  The key of
// a mapping is the peer's ALLOCATION -- the exporting process, the allocation's base address and size in its address space, published
// with the handle -- and the importing device; not the handle's bytes (their encoding is the runtime's business).
// A mapping is never closed and opened again for an allocation that is still the same: round 4 found that a communicator which closed
// its mappings on destruction, followed by a second communicator re-opening the SAME allocations, read other bytes through some of the
// fresh mappings when the device was time-slicing the ranks' processes ("An oversubscribed device": one importer
// wrong, another importer of the same allocation in the same call right).  What is closed: everything imported from a peer whose pool
// gave memory back to the driver (it publishes an epoch with every handle; compared at the head of a pull, before any address of
// that call has been resolved), and -- beyond GM_SHM_MAX_OPENED entries (default 1024: evicting an entry invites exactly the close-and-import-again this cache exists to avoid) -- the least recently used entries no pull of
// any thread is working with.
struct IpcCache {
    struct Opened { uint64_t pid, base_va, alloc_bytes; int dev; void* ptr; uint64_t stamp; uint32_t in_use; bool dead; };
    std::mutex mu;
    std::vector<Opened> opened;
    std::vector<std::pair<uint64_t, uint64_t>> epoch_of;   // (pid, last epoch seen)
    uint64_t clock = 0;
    size_t max_opened = [] { const char* e = getenv("GM_SHM_MAX_OPENED"); long v = e ? atol(e) : 0; return (size_t)(v > 0 ? v : 1024); }();
    static IpcCache& get() { static IpcCache c; return c; }
    // a new epoch of `pid`: its allocations may have gone back to the driver -- drop what nobody is using; an entry another thread's
    // pull is working with (resolved under the epoch THAT pull was told) is marked dead: never handed out again, closed when let go
    uint64_t sync_epoch(uint64_t pid, uint64_t epoch) {
        std::lock_guard<std::mutex> g(mu);
        uint64_t closed = 0;
        for (auto& pe : epoch_of)
            if (pe.first == pid) {
                if (pe.second == epoch) return 0;
                pe.second = epoch;
                for (size_t i = 0; i < opened.size();) {
                    if (opened[i].pid == pid && opened[i].in_use == 0) { (void)hipIpcCloseMemHandle(opened[i].ptr); closed++; opened.erase(opened.begin() + i); continue; }
                    if (opened[i].pid == pid) opened[i].dead = true;
                    i++;
                }
                return closed;
            }
        epoch_of.emplace_back(pid, epoch);
        return 0;
    }
    // the mapping of (pid, base_va, alloc_bytes) on the current device, opened if need be; pinned (in_use) until release()
    void* acquire(uint64_t pid, const hipIpcMemHandle_t& h, uint64_t base_va, uint64_t alloc_bytes, bool* was_new) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::lock_guard<std::mutex> g(mu);
        for (Opened& o : opened)
            if (!o.dead && o.pid == pid && o.base_va == base_va && o.alloc_bytes == alloc_bytes && o.dev == dev) {
                o.stamp = ++clock;
                o.in_use++;
                *was_new = false;
                return o.ptr;
            }
        void* p = nullptr;
        if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        opened.push_back(Opened{pid, base_va, alloc_bytes, dev, p, ++clock, 1u, false});
        *was_new = true;
        return p;
    }
    void release(void* ptr) {
        std::lock_guard<std::mutex> g(mu);
        for (size_t i = 0; i < opened.size(); i++)
            if (opened[i].ptr == ptr && opened[i].in_use) {
                if (--opened[i].in_use == 0 && opened[i].dead) { (void)hipIpcCloseMemHandle(opened[i].ptr); opened.erase(opened.begin() + i); }
                return;
            }
    }
    uint64_t trim() {   // least recently used first, never an entry a pull is working with
        std::lock_guard<std::mutex> g(mu);
        uint64_t closed = 0;
        while (opened.size() > max_opened) {
            size_t victim = opened.size();
            for (size_t i = 0; i < opened.size(); i++)
                if (opened[i].in_use == 0 && (victim == opened.size() || opened[i].stamp < opened[victim].stamp)) victim = i;
            if (victim == opened.size()) break;
            (void)hipIpcCloseMemHandle(opened[victim].ptr);
            opened.erase(opened.begin() + victim);
            closed++;
        }
        return closed;
    }
    size_t held() { std::lock_guard<std::mutex> g(mu); return opened.size(); }
};

struct gm_shm {
    std::string name;
    uint32_t rank = 0, world = 1;
    char* base = nullptr;
    size_t bytes = 0;
    uint64_t seq = 0;       // all_gather chunks done by this rank
    uint64_t calls = 0, payload = 0;
    // peers' allocations opened through HIP IPC (pull_dev): a PROCESS-WIDE cache (IpcCache below), shared by every communicator of the process
    uint64_t pull_no = 0;          // (diagnostics) pull_dev calls of this communicator so far
    uint64_t ipc_opens = 0, ipc_closes = 0;
    bool last_open_was_new = false;   // (GM_SHM_DEBUG)
    ShmHeader* hdr() const { return reinterpret_cast<ShmHeader*>(base); }
    std::atomic<uint64_t>* seq_of(uint32_t r) const { return reinterpret_cast<std::atomic<uint64_t>*>(base + 64 + (size_t)r * 64); }
    char* slot(uint32_t r, uint64_t n) const { return base + 64 + (size_t)world * 64 + ((size_t)r * 2 + (n & 1)) * SHM_SLOT; }
    ~gm_shm() {
        if (base) munmap(base, bytes);   // (the IPC mappings stay with the process: IpcCache)
    }
};

// wait until *w >= want; false on time-out (gm_set_wait_timeout_ms)
static bool shm_wait(const std::atomic<uint64_t>* w, uint64_t want) {
    for (int spin = 0; spin < 20000; spin++) {
        if (w->load(std::memory_order_acquire) >= want) return true;
        __builtin_ia32_pause();
    }
    const auto t0 = std::chrono::steady_clock::now();
    const auto limit = wait_timeout_host() + std::chrono::milliseconds(200);
    for (uint32_t it = 0;; it++) {
        if (w->load(std::memory_order_acquire) >= want) return true;
        if ((it & 63u) == 63u) {
            if (std::chrono::steady_clock::now() - t0 > limit) return false;
            std::this_thread::yield();   // more ranks than cores (the shared-GPU rehearsals): let the others run
        } else {
            __builtin_ia32_pause();
        }
    }
}

static int32_t shm_all_gather(void* ctx, void* buf, uint64_t nbytes) {
    gm_shm* c = static_cast<gm_shm*>(ctx);
    if (!c || !buf) return 1;
    char* hb = static_cast<char*>(buf);
    for (uint64_t done = 0; done < nbytes; done += SHM_SLOT) {
        const size_t len = (size_t)(nbytes - done < SHM_SLOT ? nbytes - done : SHM_SLOT);
        const uint64_t n = ++c->seq;
        memcpy(c->slot(c->rank, n), hb + (size_t)c->rank * nbytes + done, len);
        c->seq_of(c->rank)->store(n, std::memory_order_release);
        for (uint32_t d = 1; d < c->world; d++) {
            const uint32_t r = (c->rank + d) % c->world;
            if (!shm_wait(c->seq_of(r), n)) return 2;
            memcpy(hb + (size_t)r * nbytes + done, c->slot(r, n), len);
        }
    }
    c->calls++;
    c->payload += nbytes;
    return 0;
}

// gm_comm::pull_dev: the bulk redistributions of the sharded pushforward argument, device to device.  Every rank publishes a HIP IPC
// handle of the allocation its source lies in (+ the offset inside it) through the all-gather above, opens the handles of the ranks it
// pulls from and copies its pieces with hipMemcpyAsync -- between two GPUs that is a peer copy over xGMI, between ranks that share
// a device (the rehearsals on this box) a local copy.  Two host barriers: sources complete before anybody reads, everybody done
// reading before a source may be reused.
struct ShmIpcMsg {
    hipIpcMemHandle_t handle;
    uint64_t offset, bytes, epoch;
    uint64_t base_va, alloc_bytes;   // the allocation the handle stands for, in the exporter's address space: the key of the peers' mapping caches
    uint32_t failed, pad;     // the rank's source could not be completed (stream error): every rank returns an error, nobody waits
    char bus[16];             // PCI bus id of the rank's device: a peer on the SAME device (ranks sharing a GPU: the one-box rehearsals) is
                              // copied from by a kernel -- hipMemcpyAsync from an IPC mapping takes the SDMA path, ~55 GB/s, where
                              // the device copies itself at TB/s; between two GPUs the copy stays a peer hipMemcpyAsync over xGMI
    uint64_t pid, raw;        // ranks that are THREADS of one process (one process driving several GPUs, or the shared-GPU rehearsals
                              // of more ranks than the box allows processes) share an address space: the source pointer itself serves,
                              // no IPC mapping (a process cannot open its own export)
};
// Return value GM_PULL_UNAVAILABLE (100): some rank could not export or open a mapping (devices hidden from each other, IPC
// switched off): every rank learns it in the same exchange, nothing was copied, and the caller stages this redistribution through
// the host instead -- on every rank alike.
#define GM_PULL_UNAVAILABLE 100
static int32_t shm_pull_dev(void* ctx, const void* d_src, uint64_t src_bytes, uint32_t n, const gm_pull* pieces, void* stream) {
    gm_shm* c = static_cast<gm_shm*>(ctx);
    if (!c || (n && !pieces)) return 1;
    hipStream_t s = as_stream(stream);
    c->pull_no++;
    std::vector<ShmIpcMsg> msgs(c->world);
    ShmIpcMsg& mine = msgs[c->rank];
    memset(&mine, 0, sizeof(mine));
    static const bool no_ipc = [] { const char* e = getenv("GM_SHM_NO_IPC"); return e && e[0] == '1'; }();   // tests: this rank cannot export
    bool exported = !no_ipc;
    if (exported && d_src && src_bytes) {
        void* base = nullptr;
        size_t size = 0;
        if (hipMemGetAddressRange(reinterpret_cast<hipDeviceptr_t*>(&base), &size, const_cast<void*>(d_src)) != hipSuccess ||
            hipIpcGetMemHandle(&mine.handle, base) != hipSuccess) {
            (void)hipGetLastError();
            exported = false;
        } else {
            mine.offset = (uint64_t)(static_cast<const char*>(d_src) - static_cast<const char*>(base));
            mine.bytes = src_bytes;
            mine.base_va = (uint64_t)(uintptr_t)base;
            mine.alloc_bytes = (uint64_t)size;
        }
    }
    mine.epoch = dev_pool().release_epoch.load();
    {
        int dev_ = 0;
        (void)hipGetDevice(&dev_);
        char busid[64] = {0};
        if (hipDeviceGetPCIBusId(busid, sizeof(busid), dev_) != hipSuccess) { (void)hipGetLastError(); busid[0] = 0; }
        strncpy(mine.bus, busid, sizeof(mine.bus) - 1);
    }
    mine.pid = (uint64_t)getpid();
    mine.raw = exported ? (uint64_t)(uintptr_t)d_src : 0;
    // the source is complete before its handle goes out; a rank that cannot complete it says so IN the exchange (its peers would
    // otherwise sit in the all-gather until the time-out)
    static const bool dev_sync = [] { const char* e = getenv("GM_SHM_DEVICE_SYNC"); return e && e[0] == '1'; }();   // experiment: the whole device, not the stream
    if ((dev_sync ? hipDeviceSynchronize() : hipStreamSynchronize(s)) != hipSuccess) { (void)hipGetLastError(); mine.failed = 1; mine.bytes = 0; }
    if (int32_t rc = shm_all_gather(c, msgs.data(), sizeof(ShmIpcMsg))) return rc;
    for (uint32_t r = 0; r < c->world; r++)
        if (msgs[r].failed) return r == c->rank ? 5 : 10;
    // mappings of a peer whose pool gave memory back are dropped now, before any address of this call is resolved
    for (uint32_t r = 0; r < c->world; r++)
        if (r != c->rank && msgs[r].pid != mine.pid) c->ipc_closes += IpcCache::get().sync_epoch(msgs[r].pid, msgs[r].epoch);
    // open what this rank pulls from; then agree that everybody could
    std::vector<const char*> src(n, nullptr);
    std::vector<void*> pinned;   // mappings this call resolved addresses into: released after the "done reading" barrier (or on the way out)
    struct Unpin {
        std::vector<void*>& v;
        ~Unpin() { for (void* p_ : v) IpcCache::get().release(p_); }
    } unpin{pinned};
    int32_t err = 0;
    bool usable = exported;
    for (uint32_t k = 0; k < n && usable; k++) {
        const gm_pull& p = pieces[k];
        if (p.peer >= c->world || !p.d_dst) { err = 6; break; }
        if (p.src_offset + p.bytes > msgs[p.peer].bytes) { usable = false; break; }   // (a peer that could not export announces 0 bytes)
        if (p.peer == c->rank) src[k] = static_cast<const char*>(d_src) + p.src_offset;
        else if (msgs[p.peer].pid == mine.pid) {
            if (!msgs[p.peer].raw) { usable = false; break; }
            src[k] = reinterpret_cast<const char*>((uintptr_t)msgs[p.peer].raw) + p.src_offset;
        } else {
            void* peer_base = IpcCache::get().acquire(msgs[p.peer].pid, msgs[p.peer].handle, msgs[p.peer].base_va, msgs[p.peer].alloc_bytes,
                                                      &c->last_open_was_new);
            if (!peer_base) { usable = false; break; }
            pinned.push_back(peer_base);
            if (c->last_open_was_new) c->ipc_opens++;
            static const bool dbg = [] { const char* e = getenv("GM_SHM_DEBUG"); return e && e[0] == '1'; }();
            if (dbg)
                fprintf(stderr, "[gm shm pull %llu] rank %u piece %u <- rank %u: %s mapping %p of allocation %llx + %llu MiB, source offset %llu + %llu, %llu bytes\n",
                        (unsigned long long)c->pull_no, c->rank, k, p.peer, c->last_open_was_new ? "NEW" : "cached", peer_base,
                        (unsigned long long)msgs[p.peer].base_va, (unsigned long long)(msgs[p.peer].alloc_bytes >> 20),
                        (unsigned long long)msgs[p.peer].offset, (unsigned long long)p.src_offset, (unsigned long long)p.bytes);
            src[k] = static_cast<const char*>(peer_base) + msgs[p.peer].offset + p.src_offset;
        }
    }
    std::vector<uint32_t> st(c->world, 0);
    st[c->rank] = err ? 3u : usable ? 1u : 2u;
    if (int32_t rc = shm_all_gather(c, st.data(), sizeof(uint32_t))) return rc;
    if (err) return err;
    for (uint32_t r = 0; r < c->world; r++) {
        if (st[r] == 3u) return 10;                     // a peer failed outright
        if (st[r] != 1u) return GM_PULL_UNAVAILABLE;    // same answer on every rank; nothing has been copied
    }
    static const bool no_kernel_copy = [] { const char* e = getenv("GM_SHM_NO_KERNEL_COPY"); return e && e[0] == '1'; }();   // A/B
    for (uint32_t k = 0; k < n && !err; k++) {
        const gm_pull& p = pieces[k];
        const bool same_dev = mine.bus[0] && memcmp(mine.bus, msgs[p.peer].bus, sizeof(mine.bus)) == 0;
        if (same_dev && !no_kernel_copy && p.bytes >= 4096 && p.bytes % 16 == 0 && ((uintptr_t)p.d_dst | (uintptr_t)src[k]) % 16 == 0) {
            const uint64_t n16 = p.bytes / 16;
            uint64_t blocks = (n16 + 1023) / 1024;
            if (blocks > 4096) blocks = 4096;
            static const bool copy_sc = [] { const char* e = getenv("GM_SHM_COPY_SC"); return e && e[0] == '1'; }();
            if (copy_sc)
                hipLaunchKernelGGL(k_shm_copy16_sc, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<const uint4*>(src[k]),
                                   reinterpret_cast<uint4*>(p.d_dst), n16);
            else
            hipLaunchKernelGGL(k_shm_copy16, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<const uint4*>(src[k]),
                               reinterpret_cast<uint4*>(p.d_dst), n16);
            if (hipGetLastError() != hipSuccess) err = 8;
        } else if (hipMemcpyAsync(p.d_dst, src[k], p.bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) err = 8;
    }
    if ((dev_sync ? hipDeviceSynchronize() : hipStreamSynchronize(s)) != hipSuccess && !err) err = 9;
    if (err) (void)hipGetLastError();
    // everybody is done reading (also after an error on this rank: the others must not hang in their barrier)
    std::vector<uint32_t> done(c->world, 0);
    done[c->rank] = err ? 2u : 1u;
    if (int32_t rc = shm_all_gather(c, done.data(), sizeof(uint32_t))) return rc;
    for (void* p_ : pinned) IpcCache::get().release(p_);   // this rank's copies have completed
    pinned.clear();
    c->ipc_closes += IpcCache::get().trim();
    if (err) return err;
    for (uint32_t r = 0; r < c->world; r++)
        if (done[r] != 1u) return 10;   // a peer failed
    return 0;
}

extern "C" {

// Collective: every rank of the job calls it with the same name ("/gm-<job id>": unique per job, e.g. the launcher's pid and port)
// and world.  Rank 0 creates the object (an existing one of that name is an error: a stale or duplicated job id), the others wait
// for it to appear; all return once every rank has mapped it, and rank 0 removes the name again at that point.
int32_t gm_comm_shm_create(const char* name, uint32_t rank, uint32_t world, gm_shm** out) {
    GM_REQUIRE(name && name[0] == '/' && out && world >= 1 && rank < world, "bad argument (the name starts with '/')");
    std::unique_ptr<gm_shm> c(new gm_shm());
    c->name = name; c->rank = rank; c->world = world;
    c->bytes = shm_bytes(world);
    const auto t0 = std::chrono::steady_clock::now();
    const auto limit = wait_timeout_host() + std::chrono::milliseconds(200);
    int fd = -1;
    // rank 0 owns the NAME from the moment it created it: whatever way this function is left, the name is gone again (a job whose
    // rank did not start must not leave /dev/shm/<name> behind -- 0.5 MiB per rank, and O_EXCL would refuse the next job of that name)
    struct NameGuard {
        const char* name = nullptr;
        ~NameGuard() { if (name) shm_unlink(name); }
    } name_guard;
    if (rank == 0) {
        fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0) return set_err(GM_ERR_STATE, "shm_open(%s, create) failed: %s (a stale or duplicated job name?)", name, strerror(errno));
        name_guard.name = name;
        if (ftruncate(fd, (off_t)c->bytes) != 0) {
            const int e = errno;
            close(fd);
            return set_err(GM_ERR_STATE, "ftruncate(%s, %zu) failed: %s", name, c->bytes, strerror(e));
        }
    } else {
        for (;;) {
            fd = shm_open(name, O_RDWR, 0600);
            if (fd >= 0) {
                struct stat st;
                if (fstat(fd, &st) == 0 && (size_t)st.st_size >= c->bytes) break;   // rank 0 has sized it
                close(fd);
                fd = -1;
            }
            if (std::chrono::steady_clock::now() - t0 > limit) return set_err(GM_ERR_STATE, "shared-memory object %s did not appear (rank 0 creates it)", name);
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
    }
    void* p = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) {
        return set_err(GM_ERR_STATE, "mmap(%s, %zu) failed: %s", name, c->bytes, strerror(errno));
    }
    c->base = static_cast<char*>(p);
    ShmHeader* h = c->hdr();
    if (rank == 0) {
        h->world = world;
        h->magic.store(SHM_MAGIC, std::memory_order_release);
    } else {
        while (h->magic.load(std::memory_order_acquire) != SHM_MAGIC) {
            if (std::chrono::steady_clock::now() - t0 > limit) return set_err(GM_ERR_STATE, "shared-memory object %s was never initialised", name);
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
        if (h->world != world) return set_err(GM_ERR_STATE, "shared-memory object %s belongs to a job of %u ranks, not %u", name, h->world, world);
    }
    h->attached.fetch_add(1, std::memory_order_acq_rel);
    while (h->attached.load(std::memory_order_acquire) < world) {
        if (std::chrono::steady_clock::now() - t0 > limit)
            return set_err(GM_ERR_STATE, "only %u of %u ranks attached to %s", h->attached.load(), world, name);
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
    // everybody holds a mapping: the name is no longer needed (and a rank that dies later leaves nothing behind in /dev/shm);
    // name_guard removes it here as on every failure path above
    *out = c.release();
    return GM_OK;
}

int32_t gm_comm_shm_destroy(gm_shm* c) {
    delete c;
    return GM_OK;
}

// the gm_comm the sharded prover takes: host all_gather only (no all_gather_dev: the per-round sums are wanted on the host)
int32_t gm_comm_shm_as_comm(gm_shm* c, gm_comm* out) {
    GM_REQUIRE(c && out, "null argument");
    out->ctx = c;
    out->rank = c->rank;
    out->world = c->world;
    out->all_gather = shm_all_gather;
    out->all_gather_dev = nullptr;
    out->pull_dev = shm_pull_dev;
    return GM_OK;
}

int32_t gm_comm_shm_stats(const gm_shm* c, uint64_t* all_gathers, uint64_t* bytes_per_rank_total) {
    GM_REQUIRE(c, "null argument");
    if (all_gathers) *all_gathers = c->calls;
    if (bytes_per_rank_total) *bytes_per_rank_total = c->payload;
    return GM_OK;
}

// HIP IPC mappings of the peers' allocations: opened / closed so far and held now (tests of the cache's bound and its eviction order)
int32_t gm_comm_shm_ipc_stats(const gm_shm* c, uint64_t* opens, uint64_t* closes, uint64_t* held) {
    GM_REQUIRE(c, "null argument");
    if (opens) *opens = c->ipc_opens;
    if (closes) *closes = c->ipc_closes;
    if (held) *held = IpcCache::get().held();
    return GM_OK;
}

}  // extern "C"

// Shared host-side plumbing for libgkrmsm_hip.so: error reporting and launch helpers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <atomic>
#include <map>
#include <tuple>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "../../include/gkrmsm.h"

namespace gm {

// per-device tables inside the library (stage-kernel budgets, pinned staging, persistent counters) hold this many devices;
// gm_set_device refuses ids beyond it instead of aliasing slot 0
#define GM_MAX_DEVICES 16

inline char* err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

inline int32_t set_err(int32_t code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

#define GM_HIP(call)                                                                                   \
    do {                                                                                               \
        hipError_t e__ = (call);                                                                       \
        if (e__ != hipSuccess)                                                                         \
            return gm::set_err(GM_ERR_HIP, "%s:%d %s -> %s", __FILE__, __LINE__, #call,                \
                               hipGetErrorString(e__));                                                \
    } while (0)

#define GM_REQUIRE(cond, ...)                                                                          \
    do {                                                                                               \
        if (!(cond)) return gm::set_err(GM_ERR_INVALID, __VA_ARGS__);                                  \
    } while (0)

#define GM_LAUNCH_CHECK() GM_HIP(hipGetLastError())

// Device memory for handles and workspaces.  Large blocks are cached instead of returned to the driver: a proof allocates
// tens of GiB of witness trace, and on MI355X re-allocating memory the process has just freed costs ~40 ms per GiB
// (measured: 43 GiB of gen-1 trace took 25 ms of hipMalloc on first use and 1.6 s after a free), while hipFree also
// synchronises the device.  Blocks >= 1 MiB are rounded to a size class (eight per octave) and kept on an idle list;
// gm_release_cached_memory() hands them back.  Same-stream reuse is safe by stream order, as with any caching allocator;
// callers that share buffers across streams must synchronise before destroying handles (they already must for hipFree).  Blocks below 1 MiB are cached in power-of-two classes.
// Idle blocks are keyed by the device that was current when they were allocated (gm_set_device): a block is only ever reused on its own GPU.
// They are also keyed by the host thread that freed them: two host threads that prove concurrently (each on its own stream) never
// hand each other a block whose last kernels may still be running on the other's stream.
struct DevPool {
    std::mutex mu;
    struct Block { size_t bytes; int dev; };
    std::unordered_map<void*, Block> live;
    typedef std::tuple<int, int, size_t> Key;   // (device, host thread, size class)
    std::multimap<Key, void*> idle;
    static int thread_key() {
        static std::atomic<int> next{0};
        static thread_local int k = next.fetch_add(1);
        return k;
    }
    size_t idle_bytes = 0;
    uint64_t n_driver_allocs = 0, driver_alloc_bytes = 0;  // pool misses (diagnostics)
    static constexpr size_t SMALL = (size_t)1 << 20;
    // ---- reserved slabs (gm_reserve): memory taken from the driver ONCE, at set-up time, that the pool cuts its blocks from.  The
    // driver hands out memory at ~25-40 GiB/s (a gen-1 proof at 2^20 x 2^8 allocates 120 GiB: 5 s on its first call against 0.33 s
    // in steady state; the whole gen-2 proof 0.2 s against 0.18), so a prover that proves ONCE pays several times its proving time
    // for allocation unless the cost moves to where the SRS load already is.  First fit over a list of free ranges per slab; blocks
    // go back to their slab only in release() (behind a device synchronisation, as hipFree implies one) -- between releases a freed
    // block idles under its (device, thread, class) key exactly like a driver block.
    struct Slab {
        char* base = nullptr;
        size_t bytes = 0;
        int dev = 0;
        std::map<size_t, size_t> free_;   // offset -> length
    };
    std::vector<Slab> slabs;
    std::unordered_map<void*, int> slab_of;   // blocks cut from a slab (live or idle) -> slab index
    size_t slab_bytes = 0, slab_used = 0;
    uint64_t n_slab_cuts = 0;
    static bool& no_slab() { static thread_local bool v = false; return v; }   // ExportableScope: allocations that must be driver blocks
    hipError_t reserve(size_t bytes) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        Slab sl;
        bytes = (bytes + 255) & ~(size_t)255;
        hipError_t e = hipMalloc((void**)&sl.base, bytes);
        if (e != hipSuccess) return e;
        sl.bytes = bytes; sl.dev = dev; sl.free_[0] = bytes;
        std::lock_guard<std::mutex> g(mu);
        slabs.push_back(sl);
        slab_bytes += bytes;
        return hipSuccess;
    }
    void* slab_cut(int dev, size_t b) {   // mu held; b is a multiple of 256
        for (size_t i = 0; i < slabs.size(); i++) {
            Slab& sl = slabs[i];
            if (sl.dev != dev) continue;
            for (auto it = sl.free_.begin(); it != sl.free_.end(); ++it) {
                if (it->second < b) continue;
                const size_t off = it->first, len = it->second;
                sl.free_.erase(it);
                if (len > b) sl.free_[off + b] = len - b;
                void* p = sl.base + off;
                slab_of[p] = (int)i;
                slab_used += b;
                n_slab_cuts++;
                return p;
            }
        }
        return nullptr;
    }
    void slab_return(void* p, size_t b) {   // mu held
        auto so = slab_of.find(p);
        Slab& sl = slabs[so->second];
        size_t off = (size_t)(static_cast<char*>(p) - sl.base), len = b;
        auto nx = sl.free_.lower_bound(off);
        if (nx != sl.free_.end() && off + len == nx->first) { len += nx->second; nx = sl.free_.erase(nx); }
        if (nx != sl.free_.begin()) {
            auto pv = std::prev(nx);
            if (pv->first + pv->second == off) { off = pv->first; len += pv->second; sl.free_.erase(pv); }
        }
        sl.free_[off] = len;
        slab_used -= b;
        slab_of.erase(so);
    }
    hipError_t alloc(void** out, size_t b) {
        // small blocks: power-of-two classes from 256 bytes (the provers create and drop hundreds of few-KiB buffers per
        // proof; hipMalloc costs tens of microseconds and hipFree synchronises the device); large ones: 2 MiB granules
        if (b < SMALL) {
            size_t c = 256;
            while (c < b) c <<= 1;
            b = c;
        } else {
            // eight classes per octave (<= 12.5 % slack) and exact-class reuse: a repeated request sequence (proof after
            // proof) settles on the same blocks instead of stealing slightly larger ones from later requests
            int lg = 63 - __builtin_clzll((unsigned long long)b);
            const size_t g = (size_t)1 << (lg - 3);
            b = (b + g - 1) & ~(g - 1);
        }
        int dev = 0;
        (void)hipGetDevice(&dev);
        {
            std::lock_guard<std::mutex> g(mu);
            auto it = idle.find(Key{dev, thread_key(), b});
            if (it != idle.end()) {
                *out = it->second;
                live[it->second] = Block{b, dev};
                idle_bytes -= b;
                idle.erase(it);
                return hipSuccess;
            }
            if (!slabs.empty() && !no_slab()) {
                if (void* p = slab_cut(dev, b)) {
                    *out = p;
                    live[p] = Block{b, dev};
                    return hipSuccess;
                }
            }
        }
        hipError_t e = hipMalloc(out, b);
        if (e != hipSuccess) {  // out of memory with blocks idling: give them back and retry once
            (void)hipGetLastError();
            release();
            e = hipMalloc(out, b);
            if (e != hipSuccess) return e;
        }
        std::lock_guard<std::mutex> g(mu);
        live[*out] = Block{b, dev};
        n_driver_allocs++;
        driver_alloc_bytes += b;
        return hipSuccess;
    }
    void free(void* p) {
        if (!p) return;
        {
            std::lock_guard<std::mutex> g(mu);
            auto it = live.find(p);
            if (it != live.end()) {
                idle.insert({Key{it->second.dev, thread_key(), it->second.bytes}, p});
                idle_bytes += it->second.bytes;
                live.erase(it);
                return;
            }
        }
        (void)hipFree(p);
    }
    std::atomic<uint64_t> release_epoch{0};   // bumps whenever blocks went back to the driver: peers holding HIP IPC mappings of this
                                              // process's allocations (shm_comm.hip) drop them when they see a new value
    void release() {
        release_epoch++;
        std::multimap<Key, void*> take;
        {
            std::lock_guard<std::mutex> g(mu);
            take.swap(idle);
            idle_bytes = 0;
        }
        std::vector<std::pair<void*, size_t>> to_slab;
        {
            std::lock_guard<std::mutex> g(mu);
            for (auto it = take.begin(); it != take.end();) {
                if (slab_of.count(it->second)) { to_slab.emplace_back(it->second, std::get<2>(it->first)); it = take.erase(it); }
                else ++it;
            }
        }
        if (!to_slab.empty()) {
            (void)hipDeviceSynchronize();   // back into a slab: the next cut may hand the range to another thread / stream
            std::lock_guard<std::mutex> g(mu);
            for (auto& pb : to_slab) slab_return(pb.first, pb.second);
        }
        for (auto& kv : take) (void)hipFree(kv.second);
    }
    // give the slabs back to the driver; false when blocks cut from them are still live
    bool unreserve() {
        release();
        std::lock_guard<std::mutex> g(mu);
        if (!slab_of.empty()) return false;
        for (Slab& sl : slabs) (void)hipFree(sl.base);
        slabs.clear();
        slab_bytes = slab_used = 0;
        return true;
    }
};
// allocations made inside the scope are driver blocks of their own even when slabs are reserved: buffers a rank EXPORTS to its
// peers (gm_comm::pull_dev opens the allocation a source lies in: a slab would export everything in it)
struct ExportableScope {
    bool prev;
    ExportableScope() : prev(DevPool::no_slab()) { DevPool::no_slab() = true; }
    ~ExportableScope() { DevPool::no_slab() = prev; }
};
inline DevPool& dev_pool() {
    static DevPool p;
    return p;
}
inline hipError_t dev_alloc(void** out, size_t bytes) { return dev_pool().alloc(out, bytes); }
inline void dev_free(void* p) { dev_pool().free(p); }

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

inline unsigned ceil_div(uint64_t a, uint64_t b) { return (unsigned)((a + b - 1) / b); }

}  // namespace gm

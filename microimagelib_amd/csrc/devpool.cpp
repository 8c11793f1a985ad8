// Device-memory pool behind DeviceBuffer.  The reference allocates and frees every device buffer inside each libapi.h
// call (src/api_decon.cpp:198-253: five cudaMalloc / cudaFree of ~1.9 GB per deconvolution); a time-lapse batch makes the
// same calls with the same sizes thousands of times.  Freed blocks are kept (per device, by size) and handed out again.
//   * MIL_POOL_MB caps what is kept PER DEVICE (default: 6144 MB or 1/16 of the device, whichever is smaller; 0 = no
//     pooling).  Beyond the cap the blocks freed longest ago go back to the driver (least recently freed first).
//   * mil_release_cached_memory() gives everything back (a host that shares the device with torch's or rocFFT's own
//     allocators calls it between phases).
//   * A block is synchronised on ITS device before it is reused, whichever device is current in the calling thread.
#include <dlfcn.h>

#include <list>
#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "mil_common.h"
#include "trace.h"

namespace mil {

namespace {

struct Block {
    size_t bytes;
    int device;
};

struct Cached {
    void *p;
    size_t bytes;
    int device;
};

// Runs f with `device` current, restoring the caller's device afterwards.
template <class F>
void on_device(int device, F &&f) noexcept {
    int cur = -1;
    (void)hipGetDevice(&cur);
    if (cur != device) (void)hipSetDevice(device);
    f();
    if (cur != device && cur >= 0) (void)hipSetDevice(cur);
}

// Blocks leaving the pool: collected under the lock, given to the driver AFTER it is released -- hipFree synchronises the
// device, and every other thread's pool_alloc / pool_free would wait behind it (lanes, several workers on one device).
using Doomed = std::vector<void *>;
void free_doomed(const Doomed &d) noexcept {
    for (void *p : d) (void)hipFree(p);
}

struct Pool {
    std::mutex mu;
    std::list<Cached> lru;  // front = freed longest ago
    std::map<std::pair<int, size_t>, std::vector<std::list<Cached>::iterator>> free_blocks;  // (device, bytes) -> blocks
    std::unordered_map<void *, Block> live;                                                   // every block handed out
    std::map<int, size_t> cached_bytes, cap_bytes;                                            // per device
    long long cap_env_mb = -1;
    Pool() {
        if (const char *v = getenv("MIL_POOL_MB")) cap_env_mb = atoll(v);
    }
    size_t cap_of(int device) {
        auto it = cap_bytes.find(device);
        if (it != cap_bytes.end()) return it->second;
        size_t cap;
        if (cap_env_mb >= 0) {
            cap = (size_t)cap_env_mb << 20;
        } else {
            cap = (size_t)6144 << 20;
            size_t free_b = 0, total_b = 0;
            bool ok = false;  // the size of THAT device, whichever one the calling thread has current
            on_device(device, [&] { ok = hipMemGetInfo(&free_b, &total_b) == hipSuccess; });
            if (!ok) return cap;  // not cached: asked again next time
            if (total_b / 16 < cap) cap = total_b / 16;
        }
        cap_bytes[device] = cap;
        return cap;
    }
    void drop_locked(std::list<Cached>::iterator it, Doomed &doomed) {
        auto &vec = free_blocks[{it->device, it->bytes}];
        for (size_t i = 0; i < vec.size(); ++i)
            if (vec[i] == it) {
                vec.erase(vec.begin() + (long)i);
                break;
            }
        cached_bytes[it->device] -= it->bytes;
        doomed.push_back(it->p);
        lru.erase(it);
    }
    void release_all_locked(Doomed &doomed) {
        for (auto &c : lru) doomed.push_back(c.p);
        lru.clear();
        free_blocks.clear();
        cached_bytes.clear();
    }
};

Pool &pool() {
    static Pool *p = new Pool();  // never destroyed: device memory goes with the process, hipFree at exit races the runtime
    return *p;
}

size_t round_up(size_t bytes) {
    const size_t g = bytes < (1u << 20) ? 4096 : (2u << 20);
    return (bytes + g - 1) / g * g;
}

int device_of(void *p) {
    Pool &pl = pool();
    std::lock_guard<std::mutex> lock(pl.mu);
    auto it = pl.live.find(p);
    return it == pl.live.end() ? -1 : it->second.device;
}

void give_back(void *p) noexcept {
    Pool &pl = pool();
    Doomed doomed;
    {
        std::lock_guard<std::mutex> lock(pl.mu);
        auto it = pl.live.find(p);
        if (it == pl.live.end()) {
            doomed.push_back(p);
        } else {
            const Block b = it->second;
            pl.live.erase(it);
            const size_t cap = pl.cap_of(b.device);
            if (b.bytes > cap) {
                doomed.push_back(p);
            } else {
                // make room: least recently freed blocks of this device first
                for (auto c = pl.lru.begin(); c != pl.lru.end() && pl.cached_bytes[b.device] + b.bytes > cap;) {
                    auto next = std::next(c);
                    if (c->device == b.device) pl.drop_locked(c, doomed);
                    c = next;
                }
                pl.lru.push_back(Cached{p, b.bytes, b.device});
                pl.free_blocks[{b.device, b.bytes}].push_back(std::prev(pl.lru.end()));
                pl.cached_bytes[b.device] += b.bytes;
            }
        }
    }
    free_doomed(doomed);
}

}  // namespace

void *pool_alloc(size_t bytes) {
    if (bytes == 0) return nullptr;
    Pool &pl = pool();
    const size_t sz = round_up(bytes);
    int dev = 0;
    MIL_HIP(hipGetDevice(&dev));
    {
        std::lock_guard<std::mutex> lock(pl.mu);
        auto it = pl.free_blocks.find({dev, sz});
        if (it != pl.free_blocks.end() && !it->second.empty()) {
            auto c = it->second.back();
            it->second.pop_back();
            void *p = c->p;
            pl.cached_bytes[dev] -= sz;
            pl.lru.erase(c);
            pl.live[p] = Block{sz, dev};
            return p;
        }
    }
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, sz);  // outside the lock: the driver call may take milliseconds
    if (e != hipSuccess) {  // give the cache back and try once more
        (void)hipGetLastError();
        Doomed doomed;
        {
            std::lock_guard<std::mutex> lock(pl.mu);
            pl.release_all_locked(doomed);
        }
        free_doomed(doomed);
        e = hipMalloc(&p, sz);
    }
    check_hip(e, "hipMalloc", __FILE__, __LINE__);
    std::lock_guard<std::mutex> lock(pl.mu);
    pl.live[p] = Block{sz, dev};
    return p;
}

void pool_free(void *p) noexcept {
    if (!p) return;
    // hipFree waits for the device; callers rely on that (a buffer may go out of scope behind work still queued).  The
    // device to wait for is the BLOCK's, not whichever one the calling thread has current.
    const int dev = device_of(p);
    if (dev >= 0) on_device(dev, [] { (void)hipDeviceSynchronize(); });
    else (void)hipDeviceSynchronize();
    give_back(p);
}

void pool_free_after(void *p, hipStream_t s) noexcept {
    if (!p) return;
    (void)hipStreamSynchronize(s);
    give_back(p);
}

size_t pool_release_all() {
    Pool &pl = pool();
    Doomed doomed;
    size_t n = 0;
    {
        std::lock_guard<std::mutex> lock(pl.mu);
        for (auto &kv : pl.cached_bytes) n += kv.second;
        pl.release_all_locked(doomed);
    }
    free_doomed(doomed);
    return n;
}

size_t pool_cached_bytes() {
    Pool &pl = pool();
    std::lock_guard<std::mutex> lock(pl.mu);
    size_t n = 0;
    for (auto &kv : pl.cached_bytes) n += kv.second;
    return n;
}

size_t pool_cached_bytes_on(int device) {
    Pool &pl = pool();
    std::lock_guard<std::mutex> lock(pl.mu);
    auto it = pl.cached_bytes.find(device);
    return it == pl.cached_bytes.end() ? 0 : it->second;
}

// ---------------------------------------------------------------------------------------------------------------------
// roctx hooks of trace.h
// ---------------------------------------------------------------------------------------------------------------------
const TraceHooks &trace_hooks() {
    static const TraceHooks hooks = [] {
        TraceHooks h;
        const char *v = getenv("MIL_TRACE");
        if (v && v[0] == '0') return h;
        void *lib = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_LOCAL);
        if (!lib) lib = dlopen("librocprofiler-sdk-roctx.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!lib) return h;  // a ROCm install without rocprofiler-sdk: ranges are no-ops
        h.push = reinterpret_cast<int (*)(const char *)>(dlsym(lib, "roctxRangePushA"));
        h.pop = reinterpret_cast<int (*)()>(dlsym(lib, "roctxRangePop"));
        if (!h.push || !h.pop) h.push = nullptr, h.pop = nullptr;
        return h;
    }();
    return hooks;
}

}  // namespace mil

// Device-memory pool behind DeviceBuffer.  The reference allocates and frees every device buffer inside each libapi.h
// call (src/api_decon.cpp:198-253: five cudaMalloc / cudaFree of ~1.9 GB per deconvolution); a time-lapse batch makes the
// same calls with the same sizes thousands of times.  Freed blocks are kept (per device, by size) and handed out again;
// MIL_POOL_MB caps what is kept (default 16384 MB, 0 = no pooling), mil_release_cached_memory() gives it back.
#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "mil_common.h"

namespace mil {

namespace {

struct Block {
    size_t bytes;
    int device;
};

struct Pool {
    std::mutex mu;
    std::map<std::pair<int, size_t>, std::vector<void *>> free_blocks;  // (device, bytes) -> blocks
    std::unordered_map<void *, Block> live;                             // every block handed out
    size_t cached_bytes = 0;
    size_t cap_bytes;
    Pool() {
        const char *v = getenv("MIL_POOL_MB");
        cap_bytes = (size_t)(v ? atoll(v) : 16384) << 20;
    }
    void release_all_locked() {
        for (auto &kv : free_blocks)
            for (void *p : kv.second) (void)hipFree(p);
        free_blocks.clear();
        cached_bytes = 0;
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

}  // namespace

void *pool_alloc(size_t bytes) {
    if (bytes == 0) return nullptr;
    Pool &pl = pool();
    const size_t sz = round_up(bytes);
    int dev = 0;
    MIL_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(pl.mu);
    auto it = pl.free_blocks.find({dev, sz});
    if (it != pl.free_blocks.end() && !it->second.empty()) {
        void *p = it->second.back();
        it->second.pop_back();
        pl.cached_bytes -= sz;
        pl.live[p] = Block{sz, dev};
        return p;
    }
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, sz);
    if (e != hipSuccess) {  // give the cache back and try once more
        (void)hipGetLastError();
        pl.release_all_locked();
        e = hipMalloc(&p, sz);
    }
    check_hip(e, "hipMalloc", __FILE__, __LINE__);
    pl.live[p] = Block{sz, dev};
    return p;
}

namespace {
void give_back(void *p) noexcept;
}

void pool_free(void *p) noexcept {
    if (!p) return;
    // hipFree waits for the device; callers rely on that (a buffer may go out of scope behind work still queued)
    (void)hipDeviceSynchronize();
    give_back(p);
}

void pool_free_after(void *p, hipStream_t s) noexcept {
    if (!p) return;
    (void)hipStreamSynchronize(s);
    give_back(p);
}

namespace {
void give_back(void *p) noexcept {
    Pool &pl = pool();
    std::lock_guard<std::mutex> lock(pl.mu);
    auto it = pl.live.find(p);
    if (it == pl.live.end()) {
        (void)hipFree(p);
        return;
    }
    const Block b = it->second;
    pl.live.erase(it);
    if (pl.cached_bytes + b.bytes > pl.cap_bytes) {
        (void)hipFree(p);
        return;
    }
    pl.free_blocks[{b.device, b.bytes}].push_back(p);
    pl.cached_bytes += b.bytes;
}
}  // namespace

size_t pool_release_all() {
    Pool &pl = pool();
    std::lock_guard<std::mutex> lock(pl.mu);
    const size_t n = pl.cached_bytes;
    pl.release_all_locked();
    return n;
}

}  // namespace mil

// Shared host-side helpers for the MI355X (gfx950) implementation of the
// microImageLib hot path.  Not a public header.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>

namespace mil {

constexpr float kSmallValue = 0.01f;  // SMALLVALUE, reference src/api_subfunc.cu:24
constexpr int kNdim = 12;             // NDIM, reference src/api_subfunc.cu:25

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &what) : std::runtime_error(what), code(c) {}
};

inline void check_hip(hipError_t e, const char *what, const char *file, int line) {
    if (e != hipSuccess) {
        char buf[512];
        snprintf(buf, sizeof buf, "%s failed: %s (%s:%d)", what, hipGetErrorString(e), file, line);
        throw Error((int)e, buf);
    }
}

#define MIL_HIP(expr) ::mil::check_hip((expr), #expr, __FILE__, __LINE__)

// Dimensions of a volume stored x-fastest: idx = x + nx*(y + ny*z).
struct Dims {
    long long nx, ny, nz;
    long long count() const { return nx * ny * nz; }
    bool operator==(const Dims &o) const { return nx == o.nx && ny == o.ny && nz == o.nz; }
    bool operator!=(const Dims &o) const { return !(*this == o); }
};

inline long long div_up(long long a, long long b) { return (a + b - 1) / b; }

// Device allocations go through a pool that keeps freed blocks for the next call of the same size (devpool.cpp).
void *pool_alloc(size_t bytes);
void pool_free(void *p) noexcept;               // waits for the whole device first, like hipFree
void pool_free_after(void *p, hipStream_t s) noexcept;  // waits for `s` only: the caller knows no other stream uses p
size_t pool_release_all();  // hipFree everything cached; returns the bytes given back
size_t pool_cached_bytes();  // bytes currently kept for reuse (all devices)
size_t pool_cached_bytes_on(int device);

// RAII device buffer.
template <class T>
class DeviceBuffer {
public:
    DeviceBuffer() = default;
    explicit DeviceBuffer(size_t n) { alloc(n); }
    ~DeviceBuffer() { release(); }
    DeviceBuffer(const DeviceBuffer &) = delete;
    DeviceBuffer &operator=(const DeviceBuffer &) = delete;
    DeviceBuffer(DeviceBuffer &&o) noexcept : p_(o.p_), n_(o.n_) { o.p_ = nullptr; o.n_ = 0; }
    DeviceBuffer &operator=(DeviceBuffer &&o) noexcept {
        if (this != &o) { release(); p_ = o.p_; n_ = o.n_; o.p_ = nullptr; o.n_ = 0; }
        return *this;
    }
    void alloc(size_t n) {
        release();
        if (n) p_ = static_cast<T *>(pool_alloc(n * sizeof(T)));
        n_ = n;
    }
    void release() {
        if (p_) pool_free(p_);
        p_ = nullptr;
        n_ = 0;
    }
    // for a buffer only ever used on stream `s`: does not stall on other streams' work (a time-lapse host overlaps the
    // deconvolution of one time point with the registration of the next)
    void release_after(hipStream_t s) {
        if (p_) pool_free_after(p_, s);
        p_ = nullptr;
        n_ = 0;
    }
    T *get() const { return p_; }
    size_t size() const { return n_; }
    operator T *() const { return p_; }

private:
    T *p_ = nullptr;
    size_t n_ = 0;
};

// Translate exceptions into C return codes at the ABI boundary.
template <class F>
inline int guarded(const char *where, F &&f) noexcept {
    try {
        return f();
    } catch (const Error &e) {
        fprintf(stderr, "[microimagelib_amd] %s: %s\n", where, e.what());
        return e.code ? e.code : -1;
    } catch (const std::exception &e) {
        fprintf(stderr, "[microimagelib_amd] %s: %s\n", where, e.what());
        return -1;
    }
}

}  // namespace mil

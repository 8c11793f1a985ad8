// mil_device.h -- the thin device-pointer C ABI.  No C++ types cross this boundary.
#include "../../include/mil_device.h"

#include <cstring>
#include <map>
#include <memory>

#include "cpu_fft.h"
#include "decon.h"
#include "powell.h"
#include "reg.h"

using namespace mil;

struct mil_decon {
    DeconEngine engine;
    mil_decon(Dims box, hipStream_t s, int backend) : engine(box, s, backend) {}
};

namespace {
inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }
inline Dims dims3(int x, int y, int z) { return Dims{x, y, z}; }

// The registration entry points poll a pinned result slot between launches, so they do not run on the null stream
// (every launch there would also serialise against the caller's other streams).  A caller that passes stream 0 --
// torch's default stream -- gets a private non-blocking stream that is ORDERED after everything already queued on the
// null stream (pending H2D copies, the kernels that produced the inputs), and the null stream is ordered after the
// private one on the way out, so the call behaves as if it had run on stream 0.
class WorkStream {
public:
    explicit WorkStream(hipStream_t given) : s_(given) {
        if (s_) return;
        MIL_HIP(hipStreamCreateWithFlags(&s_, hipStreamNonBlocking));
        own_ = true;
        if (hipEventCreateWithFlags(&ev_, hipEventDisableTiming) != hipSuccess ||
            hipEventRecord(ev_, nullptr) != hipSuccess || hipStreamWaitEvent(s_, ev_, 0) != hipSuccess) {
            release();
            throw Error(-1, "WorkStream: cannot order the private stream after the null stream");
        }
    }
    WorkStream(const WorkStream &) = delete;
    WorkStream &operator=(const WorkStream &) = delete;
    ~WorkStream() { release(); }
    hipStream_t get() const { return s_; }
    void finish() {
        if (!own_) return;
        MIL_HIP(hipEventRecord(ev_, s_));
        MIL_HIP(hipStreamWaitEvent(nullptr, ev_, 0));
    }

private:
    void release() {
        if (ev_) (void)hipEventDestroy(ev_);
        if (own_ && s_) {
            (void)hipStreamSynchronize(s_);
            (void)hipStreamDestroy(s_);
        }
        ev_ = nullptr;
        s_ = nullptr;
        own_ = false;
    }
    hipStream_t s_ = nullptr;
    hipEvent_t ev_ = nullptr;
    bool own_ = false;
};
}  // namespace

extern "C" {

const char *mil_version(void) { return "microimagelib_amd 0.1 (gfx950)"; }

int mil_device_count(void) {
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

int mil_set_device(int device) {
    return guarded("mil_set_device", [&] { MIL_HIP(hipSetDevice(device)); return 0; });
}

int mil_dev_mem_info(int device, size_t *free_bytes, size_t *total_bytes) {
    if (free_bytes) *free_bytes = 0;
    if (total_bytes) *total_bytes = 0;
    return guarded("mil_dev_mem_info", [&] {
        int before = 0;
        MIL_HIP(hipGetDevice(&before));
        MIL_HIP(hipSetDevice(device));
        size_t f = 0, t = 0;
        const hipError_t e = hipMemGetInfo(&f, &t);
        MIL_HIP(hipSetDevice(before));
        MIL_HIP(e);
        if (free_bytes) *free_bytes = f + pool_cached_bytes_on(device);
        if (total_bytes) *total_bytes = t;
        return 0;
    });
}

int mil_snap_transform_size(int n) { return snap_transform_size(n); }

int mil_set_texture_weight_bits(int bits) {
    k::set_texture_weight_bits(bits);
    return k::texture_weight_bits();
}

size_t mil_release_cached_memory(void) {
    decon_cache_drop();
    return pool_release_all();
}
size_t mil_cached_memory_bytes(void) { return pool_cached_bytes(); }

int mil_rl_ratio(float *d_T, const float *d_A, size_t n, void *stream) {
    return guarded("mil_rl_ratio", [&] { k::rl_ratio(d_T, d_A, n, as_stream(stream)); return 0; });
}
int mil_rl_update(float *d_E, const float *d_T, size_t n, float eps, void *stream) {
    return guarded("mil_rl_update", [&] { k::rl_update(d_E, d_T, n, eps, as_stream(stream)); return 0; });
}
int mil_cmul(void *d_S, const void *d_OTF, size_t n_complex, void *stream) {
    return guarded("mil_cmul", [&] {
        k::cmul_inplace((float2 *)d_S, (const float2 *)d_OTF, n_complex, as_stream(stream));
        return 0;
    });
}
int mil_sum_f64(const float *d_in, size_t n, double *h_out, void *stream) {
    return guarded("mil_sum_f64", [&] {
        k::ReduceWorkspace ws;
        *h_out = k::reduce_sync(k::kSum, d_in, n, 0.f, ws, as_stream(stream));
        return 0;
    });
}

int mil_pad_psf(float *d_out, const float *d_psf, int fx, int fy, int fz, int px, int py, int pz, float scale, int flip,
                void *stream) {
    return guarded("mil_pad_psf", [&] {
        k::pad_psf(d_out, d_psf, dims3(fx, fy, fz), dims3(px, py, pz), scale, flip != 0, as_stream(stream));
        return 0;
    });
}
int mil_pad_stack(float *d_out, const float *d_in, int fx, int fy, int fz, int sx, int sy, int sz, void *stream) {
    return guarded("mil_pad_stack", [&] {
        k::pad_stack(d_out, d_in, dims3(fx, fy, fz), dims3(sx, sy, sz), as_stream(stream));
        return 0;
    });
}
int mil_crop_stack(float *d_out, const float *d_in, int sx, int sy, int sz, int fx, int fy, int fz, void *stream) {
    return guarded("mil_crop_stack", [&] {
        k::crop_stack(d_out, d_in, dims3(sx, sy, sz), dims3(fx, fy, fz), as_stream(stream));
        return 0;
    });
}
int mil_align_size(float *d_out, const float *d_in, int ox, int oy, int oz, int ix, int iy, int iz, void *stream) {
    return guarded("mil_align_size", [&] {
        k::align_size(d_out, d_in, dims3(ox, oy, oz), dims3(ix, iy, iz), as_stream(stream));
        return 0;
    });
}
int mil_flip3(float *d_out, const float *d_in, int nx, int ny, int nz, void *stream) {
    return guarded("mil_flip3", [&] { k::flip3(d_out, d_in, dims3(nx, ny, nz), as_stream(stream)); return 0; });
}
int mil_rot_y90(float *d_out, const float *d_in, int nx, int ny, int nz, int direction, void *stream) {
    return guarded("mil_rot_y90", [&] {
        k::rot_y90(d_out, d_in, dims3(nx, ny, nz), direction, as_stream(stream));
        return 0;
    });
}
int mil_max_projection(float *d_out, const float *d_in, int nx, int ny, int nz, int direction, void *stream) {
    return guarded("mil_max_projection", [&] {
        k::max_projection(d_out, d_in, dims3(nx, ny, nz), direction, as_stream(stream));
        return 0;
    });
}

int mil_resample_mip_z(float *d_plane, const float *d_src, int bx, int by, int bz, int sx, int sy, int sz, const float *m12,
                       void *stream) {
    return guarded("mil_resample_mip_z", [&] {
        k::Affine12 m;
        std::memcpy(m.m, m12, sizeof m.m);
        k::resample_mip_z(d_plane, d_src, dims3(bx, by, bz), dims3(sx, sy, sz), m, as_stream(stream));
        return 0;
    });
}

// ---- device memory, streams and transfers for hosts built without the HIP headers (cli/*.cpp compile with g++ alone) -----
int mil_dev_alloc(void **d_ptr, size_t bytes) {
    *d_ptr = nullptr;
    return guarded("mil_dev_alloc", [&] { *d_ptr = pool_alloc(bytes ? bytes : 1); return 0; });
}
void mil_dev_free(void *d_ptr) {
    if (d_ptr) (void)guarded("mil_dev_free", [&] { pool_free(d_ptr); return 0; });
}
int mil_host_alloc(void **h_ptr, size_t bytes) {
    *h_ptr = nullptr;
    return guarded("mil_host_alloc", [&] { MIL_HIP(hipHostMalloc(h_ptr, bytes ? bytes : 1, hipHostMallocDefault)); return 0; });
}
void mil_host_free(void *h_ptr) {
    if (h_ptr) (void)hipHostFree(h_ptr);
}
int mil_stream_create(void **stream) {
    *stream = nullptr;
    return guarded("mil_stream_create", [&] {
        hipStream_t s = nullptr;
        MIL_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        *stream = s;
        return 0;
    });
}
void mil_stream_destroy(void *stream) {
    if (!stream) return;
    // the stream's resample counters go to the next stream that asks: not before the last launch that uses them has finished
    // (hipStreamDestroy alone returns while work is still queued)
    (void)hipStreamSynchronize(as_stream(stream));
    k::release_march_ticket(as_stream(stream));
    (void)hipStreamDestroy(as_stream(stream));
}
int mil_stream_sync(void *stream) {
    return guarded("mil_stream_sync", [&] { MIL_HIP(hipStreamSynchronize(as_stream(stream))); return 0; });
}
int mil_upload(void *d_dst, const void *h_src, size_t bytes, void *stream) {
    return guarded("mil_upload", [&] { MIL_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, as_stream(stream))); return 0; });
}
int mil_download(void *h_dst, const void *d_src, size_t bytes, void *stream) {
    return guarded("mil_download", [&] { MIL_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, as_stream(stream))); return 0; });
}
int mil_u16_to_f32(float *d_out, const unsigned short *d_in, size_t n, void *stream) {
    return guarded("mil_u16_to_f32", [&] { k::u16_to_f32(d_out, d_in, n, as_stream(stream)); return 0; });
}
int mil_f32_to_u16(unsigned short *d_out, const float *d_in, size_t n, void *stream) {
    return guarded("mil_f32_to_u16", [&] { k::f32_to_u16(d_out, d_in, n, as_stream(stream)); return 0; });
}
// imresize3d (src/apifunc.cpp:429-446) on device pointers: the same diagonal matrix through the same resampler
int mil_imresize3d(float *d_out, const float *d_in, long long sx, long long sy, long long sz, long long sx2, long long sy2,
                   long long sz2, void *stream) {
    return guarded("mil_imresize3d", [&] {
        k::Affine12 m;
        for (float &v : m.m) v = 0.0f;
        m.m[0] = float(sx2) / float(sx);
        m.m[5] = float(sy2) / float(sy);
        m.m[10] = float(sz2) / float(sz);
        k::affine_resample(d_out, d_in, Dims{sx, sy, sz}, Dims{sx2, sy2, sz2}, m, as_stream(stream));
        return 0;
    });
}

int mil_decon_create(mil_decon **out, int nx, int ny, int nz, void *stream) {
    return mil_decon_create_backend(out, nx, ny, nz, stream, MIL_BACKEND_AUTO);
}
int mil_decon_create_backend(mil_decon **out, int nx, int ny, int nz, void *stream, int backend) {
    *out = nullptr;
    return guarded("mil_decon_create", [&] {
        *out = new mil_decon(dims3(nx, ny, nz), as_stream(stream), backend);
        return 0;
    });
}
int mil_decon_backend(const mil_decon *e) { return e->engine.backend(); }
int mil_decon_two_pass(const mil_decon *e) { return e->engine.two_pass(); }
int mil_fx_two_pass_box(int nx, int ny, int nz, int *otf_cols) {
    const mil::Dims b = dims3(nx, ny, nz);
    const bool ok = nx > 0 && ny > 0 && nz > 0 && mil::fx::ConvPlan::supported(b) && mil::fx::ConvPlan::two_pass_supported(b);
    if (otf_cols) *otf_cols = ok ? mil::fx::ConvPlan::yz_tile_cols(b) : 16;
    return ok ? 1 : 0;
}
int mil_decon_convolve(mil_decon *e, float *d_out, const float *d_in, const void *d_otf) {
    return guarded("mil_decon_convolve", [&] { e->engine.convolve(d_out, d_in, (const float2 *)d_otf); return 0; });
}
int mil_decon_otf_to_natural(mil_decon *e, void *d_natural, const void *d_otf) {
    return guarded("mil_decon_otf_to_natural", [&] {
        e->engine.otf_to_natural((float2 *)d_natural, (const float2 *)d_otf);
        return 0;
    });
}
int mil_fft_digit_reversal(int n, unsigned short *freq_of_pos) {
    if (n < 1 || n > 65535) return -1;
    int rest = n;
    for (int r : {2, 3, 5, 7, 11, 13})
        while (rest % r == 0) rest /= r;
    if (rest != 1) return -1;  // prime factor the transforms have no radix for
    fx::digit_reversal_freq(n, freq_of_pos);
    return 0;
}
void mil_decon_destroy(mil_decon *e) { delete e; }
size_t mil_decon_real_count(const mil_decon *e) { return e->engine.real_count(); }
size_t mil_decon_spectrum_count(const mil_decon *e) { return e->engine.spectrum_count(); }

int mil_decon_gen_otf(mil_decon *e, void *d_otf, const float *d_psf, int px, int py, int pz, int flip) {
    return guarded("mil_decon_gen_otf", [&] {
        e->engine.gen_otf((float2 *)d_otf, d_psf, dims3(px, py, pz), flip != 0);
        return 0;
    });
}
int mil_decon_init_single(mil_decon *e, float *d_A, float *d_E, int const_init) {
    return guarded("mil_decon_init_single", [&] { e->engine.init_single(d_A, d_E, const_init != 0); return 0; });
}
int mil_decon_init_dual(mil_decon *e, float *d_A, float *d_B, float *d_E, int const_init) {
    return guarded("mil_decon_init_dual", [&] { e->engine.init_dual(d_A, d_B, d_E, const_init != 0); return 0; });
}
int mil_decon_run_single(mil_decon *e, float *d_E, const float *d_A, const void *d_otf, const void *d_otf_bp,
                         int iters) {
    return guarded("mil_decon_run_single", [&] {
        e->engine.run_single(d_E, d_A, (const float2 *)d_otf, (const float2 *)d_otf_bp, iters);
        return 0;
    });
}
int mil_decon_run_dual(mil_decon *e, float *d_E, const float *d_A, const float *d_B, const void *d_otf1,
                       const void *d_otf2, const void *d_otf_bp1, const void *d_otf_bp2, int iters) {
    return guarded("mil_decon_run_dual", [&] {
        e->engine.run_dual(d_E, d_A, d_B, (const float2 *)d_otf1, (const float2 *)d_otf2, (const float2 *)d_otf_bp1,
                           (const float2 *)d_otf_bp2, iters);
        return 0;
    });
}
int mil_decon_sync(mil_decon *e) {
    return guarded("mil_decon_sync", [&] { e->engine.synchronize(); return 0; });
}
int mil_decon_timer_enable(mil_decon *e, int which, int enable) {
    if (which < 0 || which >= DeconEngine::kTimedCount) return -1;
    e->engine.timers[which].enabled = enable != 0;
    e->engine.timers[which].every = enable > 1 ? enable : 1;  // enable = n > 1: bracket every n-th launch only
    e->engine.timers[which].reset();
    return 0;
}
int mil_decon_timer_read(mil_decon *e, int which, double *total_ms, size_t *launches) {
    if (which < 0 || which >= DeconEngine::kTimedCount) return -1;
    return guarded("mil_decon_timer_read", [&] {
        e->engine.synchronize();
        *launches = e->engine.timers[which].collect(total_ms);
        e->engine.timers[which].reset();
        return 0;
    });
}

int mil_affine_resample(float *d_out, const float *d_src, int ox, int oy, int oz, int sx, int sy, int sz,
                        const float *m12, void *stream) {
    return guarded("mil_affine_resample", [&] {
        k::Affine12 m;
        std::memcpy(m.m, m12, sizeof m.m);
        k::affine_resample(d_out, d_src, dims3(ox, oy, oz), dims3(sx, sy, sz), m, as_stream(stream));
        return 0;
    });
}
int mil_affine_resample_path(int ox, int oy, int oz, int sx, int sy, int sz) {
    if (k::resample_march_eligible(dims3(ox, oy, oz), dims3(sx, sy, sz))) return 2;
    return k::small_source(dims3(sx, sy, sz)) ? 1 : 0;
}

int mil_corr_sums(const float *d_target, const float *d_src, int tx, int ty, int tz, int sx, int sy, int sz,
                  const float *m12, double *h_out, void *stream) {
    return guarded("mil_corr_sums", [&] {
        k::Affine12 m;
        std::memcpy(m.m, m12, sizeof m.m);
        // one workspace per (thread, device): the partials buffer and the pinned result slot are reused
        static thread_local std::map<int, std::unique_ptr<k::CorrWorkspace>> cache;
        int dev = 0;
        MIL_HIP(hipGetDevice(&dev));
        auto &ws = cache[dev];
        if (!ws) ws.reset(new k::CorrWorkspace());
        k::corr_sums_sync(d_target, d_src, dims3(tx, ty, tz), dims3(sx, sy, sz), m, *ws, as_stream(stream), &h_out[0],
                          &h_out[1]);
        return 0;
    });
}
struct mil_padded_source {
    k::PaddedSource src;
};
int mil_padded_source_eligible(int tx, int ty, int tz, int sx, int sy, int sz) {
    return k::PaddedSource::eligible(dims3(tx, ty, tz), dims3(sx, sy, sz)) ? 1 : 0;
}
int mil_padded_source_create(mil_padded_source **out, const float *d_src, int sx, int sy, int sz, void *stream) {
    return guarded("mil_padded_source_create", [&] {
        if (!out || !d_src) throw Error(-1, "mil_padded_source_create: null argument");
        *out = nullptr;
        if (!k::PaddedSource::eligible(dims3(sx, sy, sz), dims3(sx, sy, sz)))
            throw Error(-1, "mil_padded_source_create: volume beyond the padded kernel's 32-bit offset range");
        std::unique_ptr<mil_padded_source> p(new mil_padded_source());
        p->src.build(d_src, dims3(sx, sy, sz), as_stream(stream));
        *out = p.release();
        return 0;
    });
}
void mil_padded_source_destroy(mil_padded_source *p) { delete p; }
int mil_corr_padded_pick_kernel(const float *m12) {
    k::Affine12 m;
    std::memcpy(m.m, m12, sizeof m.m);
    return k::corr_padded_pick_kernel(m);
}
int mil_corr_padded_pick_kernel2(const float *m12, int wide) {
    k::Affine12 m;
    std::memcpy(m.m, m12, sizeof m.m);
    return k::corr_padded_pick_kernel(m, wide != 0);
}
int mil_corr_set_fences(int on) {
    k::corr_set_fences(on);
    return 0;
}
int mil_corr_set_chunk_wgs_per_cu(int n) {
    k::corr_set_chunk_wgs_per_cu(n);
    return 0;
}
int mil_corr_padded_force_kernel(int kernel) {
    k::corr_padded_force_kernel(kernel);
    return 0;
}
int mil_corr_sums_padded(const float *d_target, const mil_padded_source *src, int tx, int ty, int tz,
                         const float *m12, double *h_out, void *stream) {
    return guarded("mil_corr_sums_padded", [&] {
        if (!src) throw Error(-1, "mil_corr_sums_padded: null source");
        k::Affine12 m;
        std::memcpy(m.m, m12, sizeof m.m);
        static thread_local std::map<int, std::unique_ptr<k::CorrWorkspace>> cache;
        int dev = 0;
        MIL_HIP(hipGetDevice(&dev));
        auto &ws = cache[dev];
        if (!ws) ws.reset(new k::CorrWorkspace());
        k::corr_sums_padded_sync(d_target, src->src, dims3(tx, ty, tz), m, *ws, as_stream(stream), &h_out[0], &h_out[1]);
        return 0;
    });
}
int mil_reg3d_last_cost_launches(void) { return reg3d_last_cost_launches(); }
int mil_reg3d_trace_begin(int capacity) {
    reg3d_trace_begin(capacity);
    return 0;
}
int mil_reg3d_trace_read(float *out13, int max_entries) { return reg3d_trace_read(out13, max_entries); }
int mil_reg3d_affine(float *d_reg, float *iTmx, const float *d_img1, const float *d_img2, int nx, int ny, int nz,
                     int affMethod, int flagTmx, float FTOL, int itLimit, int verbose, float *records, void *stream) {
    return guarded("mil_reg3d_affine", [&] {
        WorkStream ws(as_stream(stream));
        const int rc = reg3d_affine_device(d_reg, iTmx, d_img1, d_img2, dims3(nx, ny, nz), affMethod, flagTmx != 0,
                                           FTOL, itLimit, verbose != 0, records, ws.get());
        ws.finish();
        return rc;
    });
}

int mil_reg3d_mip_shift(float *shift3, const float *d_img1, const float *d_img2, int nx, int ny, int nz, void *stream) {
    return guarded("mil_reg3d_mip_shift", [&] {
        WorkStream ws(as_stream(stream));
        mip_shift_search(shift3, d_img1, d_img2, dims3(nx, ny, nz), ws.get());
        ws.finish();
        return 0;
    });
}

int mil_reg3d_phasor_shift(long long *shift3, const float *d_img1, const float *d_img2, int nx, int ny, int nz,
                           void *stream) {
    return guarded("mil_reg3d_phasor_shift", [&] {
        WorkStream ws(as_stream(stream));
        phasor_shift(shift3, d_img1, d_img2, dims3(nx, ny, nz), ws.get());
        ws.finish();
        return 0;
    });
}
int mil_shift_volume(float *d_out, const float *d_in, int nx, int ny, int nz, long long dx, long long dy, long long dz,
                     int wrap, void *stream) {
    return guarded("mil_shift_volume", [&] {
        k::shift_volume(d_out, d_in, dims3(nx, ny, nz), dx, dy, dz, wrap != 0, as_stream(stream));
        return 0;
    });
}

int mil_powell(float *p, float *xi, int ld, int n, float ftol, int *iter, float *fret, mil_cost_fn f, void *user,
               int *totalIt, int itLimit) {
    if (n < 1 || ld < n || !f || !totalIt) return -1;
    return guarded("mil_powell", [&] {
        CostFn cost = [&](const float *x) -> float { return f(x, user); };
        PowellResult r = powell_minimize(p, xi, ld, n, ftol, cost, totalIt, itLimit);
        if (iter) *iter = r.iterations;
        if (fret) *fret = r.fret;
        return 0;
    });
}
void mil_p2matrix(float *m12, const float *p12) { p2matrix(m12, p12); }
void mil_matrix2p(float *p12, const float *m12) { matrix2p(p12, m12); }
void mil_dof9tomatrix(float *m12, const float *p9, int dofNum) { dof9tomatrix(m12, p9, dofNum); }
void mil_matrixmultiply(float *m, const float *m1, const float *m2) { matrixmultiply(m, m1, m2); }
void mil_rot2matrix(float *m12, float theta, long long sx, long long sy, long long sz, int rotAxis) {
    rot2matrix(m12, theta, sx, sy, sz, rotAxis);
}

int mil_cpu_rfft3d(float *h_real, void *h_spec, int nx, int ny, int nz, int inverse, int threads) {
    return guarded("mil_cpu_rfft3d", [&] {
        if (!h_real || !h_spec) throw Error(-1, "mil_cpu_rfft3d: null argument");
        cpu::Rfft3D plan(dims3(nx, ny, nz), threads > 0 ? threads : cpu::default_threads());
        if (inverse) plan.inverse((float2 *)h_spec, h_real);
        else plan.forward(h_real, (float2 *)h_spec);
        return 0;
    });
}

}  // extern "C"

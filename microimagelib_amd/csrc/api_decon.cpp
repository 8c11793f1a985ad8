// libapi.h deconvolution entry points: host pointers in, host pointers out.
// Mirrors the argument meaning, return codes and records[] layout of the reference's
// decon_singleview / decon_dualview (src/api_decon.cpp:53-331, 333-704).
#include <chrono>
#include <cstdarg>
#include <cstring>
#include <memory>
#include <mutex>
#include <vector>

#include "../../include/libapi.h"
#include "cpu_decon.h"
#include "cpu_fft.h"
#include "decon.h"
#include "trace.h"

using namespace mil;

namespace {

float free_mb() {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return 0.f;
    return (float)free_b / 1048576.0f;
}

double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct Geometry {
    Dims im, psf, box;
    size_t n_im, n_psf, n_box, n_spec;
    bool padded;
};

Geometry make_geometry(const unsigned int *imSize, const unsigned int *psfSize) {
    Geometry g;
    g.im = Dims{(long long)imSize[0], (long long)imSize[1], (long long)imSize[2]};
    g.psf = Dims{(long long)psfSize[0], (long long)psfSize[1], (long long)psfSize[2]};
    if (g.im.count() <= 0 || g.psf.count() <= 0) throw Error(-1, "decon: empty image or PSF");
    // FFT box = snapTransformSize of each image extent (reference src/api_decon.cpp:77-79)
    g.box = Dims{snap_transform_size((int)g.im.nx), snap_transform_size((int)g.im.ny),
                 snap_transform_size((int)g.im.nz)};
    g.n_im = (size_t)g.im.count();
    g.n_psf = (size_t)g.psf.count();
    g.n_box = (size_t)g.box.count();
    // memory estimate only; buffers are sized by DeconEngine::spectrum_count()
    g.n_spec = (size_t)(g.box.nx / 2 + 16) * g.box.ny * g.box.nz;
    g.padded = g.box != g.im;
    return g;
}

// Resolve the requested memory mode.  288 GB of HBM: the reference's memory-saving mode 2 is run as mode 1.  In auto
// mode (-1) the reference's ladder is 1 -> 2 -> 0 by free device memory (src/api_decon.cpp:116-134, :398-416); here
// it is 1 -> 0: a volume that does not fit the device is deconvolved on the host.  A missing or failing device is
// NOT a reason to fall back -- that stays an error (hipMemGetInfo throws).  Returns the mode used.
int resolve_mode(int requested, size_t bytes_needed, bool verbose) {
    if (requested == 1 || requested == 2) return 1;
    size_t free_b = 0, total_b = 0;
    MIL_HIP(hipMemGetInfo(&free_b, &total_b));
    if (const char *cap = getenv("MIL_DECON_FREE_MB_CAP")) {  // test aid: pretend the device has less free memory
        const size_t cap_b = (size_t)atoll(cap) << 20;
        if (cap_b < free_b) free_b = cap_b;
    }
    if (free_b <= bytes_needed && !getenv("MIL_DECON_FREE_MB_CAP")) {
        // what earlier calls left cached (device pool, engine, OTFs) is ours to give back before giving up on the GPU
        decon_cache_drop();
        pool_release_all();
        MIL_HIP(hipMemGetInfo(&free_b, &total_b));
    }
    if (free_b > bytes_needed) {
        if (verbose) printf("\t... GPU memory is sufficient, processing in efficient mode !!!\n");
        return 1;
    }
    fprintf(stderr, "[microimagelib_amd] decon: %.0f MB needed, %.0f MB free on the device: CPU mode (gpuMemMode 0)\n",
            bytes_needed / 1048576.0, free_b / 1048576.0);
    return 0;
}

// gpuMemMode 0 (reference src/api_decon.cpp:151-197, :430-520): no device call at all.
int run_cpu(float *h_decon, const float *h_img1, const float *h_img2, const float *h_psf1, const float *h_psf2,
            const float *h_psf_bp1, const float *h_psf_bp2, const Geometry &g, bool dual, bool constInit, int iters,
            bool verbose, bool unmatch, float *deconRecords, double t_start) {
    cpu::DeconCpuArgs a;
    a.out = h_decon;
    a.img[0] = h_img1; a.img[1] = h_img2;
    a.psf[0] = h_psf1; a.psf[1] = h_psf2;
    a.psf_bp[0] = h_psf_bp1; a.psf_bp[1] = h_psf_bp2;
    a.im = g.im; a.psf_dims = g.psf; a.box = g.box;
    a.dual = dual; a.const_init = constInit; a.unmatch = unmatch; a.verbose = verbose;
    a.iterations = iters;
    double sec[3] = {0, 0, 0};
    a.seconds = sec;
    if (verbose) printf("...CPU deconvolution (gpuMemMode 0), %d thread(s)\n", cpu::default_threads());
    const int rc = cpu::decon(a);
    deconRecords[0] = 0.f;
    for (int i = 1; i <= 5; ++i) deconRecords[i] = 0.f;  // no device memory snapshots in CPU mode
    deconRecords[6] = (float)(now_s() - t_start - sec[2]);
    deconRecords[7] = (float)sec[0];
    deconRecords[8] = (float)sec[1];
    deconRecords[9] = (float)(now_s() - t_start);
    return rc;
}

void upload_volume(float *d_dst, float *d_stage, const float *h_src, const Geometry &g, hipStream_t s) {
    TraceRange trace("mil::upload_volume");
    if (g.padded) {  // reference src/api_decon.cpp:225-231
        MIL_HIP(hipMemcpyAsync(d_stage, h_src, g.n_im * sizeof(float), hipMemcpyHostToDevice, s));
        k::pad_stack(d_dst, d_stage, g.box, g.im, s);
    } else {
        MIL_HIP(hipMemcpyAsync(d_dst, h_src, g.n_im * sizeof(float), hipMemcpyHostToDevice, s));
    }
}

void download_volume(float *h_dst, float *d_src, float *d_stage, const Geometry &g, hipStream_t s) {
    TraceRange trace("mil::download_volume");
    if (g.padded) {  // reference src/api_decon.cpp:237-243
        k::crop_stack(d_stage, d_src, g.im, g.box, s);
        MIL_HIP(hipMemcpyAsync(h_dst, d_stage, g.n_im * sizeof(float), hipMemcpyDeviceToHost, s));
    } else {
        MIL_HIP(hipMemcpyAsync(h_dst, d_src, g.n_im * sizeof(float), hipMemcpyDeviceToHost, s));
    }
    MIL_HIP(hipStreamSynchronize(s));
}

// What a libapi.h deconvolution call keeps for the next one (the reference rebuilds all of it per call,
// src/api_decon.cpp:198-223: plans, work areas and two to four OTFs): engines with their tables and scratch, keyed by
// (device, FFT box), and the OTFs each made, found again by comparing the PSF bytes.  A time-lapse batch calls with the
// same box and the same PSFs for every time point.  Results are those of an uncached call, bit for bit (same kernels on
// the same data); MIL_DECON_CACHE=0 disables it, mil_release_cached_memory() drops it.
//
// Several host threads may deconvolve at once -- one per GPU of a node (the reference's deviceNum), or several on one GPU:
// an entry is CHECKED OUT for the duration of a call (the registry lock is held for the look-up only), a second caller
// with the same key gets a second entry.  At most kEntriesPerDevice idle-or-busy entries are kept per device (least
// recently used idle one evicted); a caller that finds them all busy with other keys works on a private entry that is
// destroyed when it returns.
struct DeconCache {
    struct Otf {
        std::vector<float> psf;  // host copy: a hit is a full comparison, not a hash match
        Dims dims;
        bool flip;
        DeviceBuffer<float2> buf;
        unsigned long long stamp;
    };
    struct Entry {
        int device = -1;
        Dims box{0, 0, 0};
        std::unique_ptr<DeconEngine> eng;
        std::vector<std::unique_ptr<Otf>> otfs;
        unsigned long long clock = 0;  // OTF recency
        unsigned long long used = 0;   // entry recency (registry clock)
        bool busy = false;
        bool cached = true;            // false: private to one call
        ~Entry() {                     // engine, stream and buffers belong to `device`
            if (!eng && otfs.empty()) return;
            int cur = -1;
            (void)hipGetDevice(&cur);
            if (cur != device && device >= 0) (void)hipSetDevice(device);
            otfs.clear();
            eng.reset();
            if (cur != device && cur >= 0) (void)hipSetDevice(cur);
        }
    };
    static constexpr size_t kEntriesPerDevice = 2;
    std::mutex mu;  // registry only: never held while an engine works
    std::vector<std::unique_ptr<Entry>> entries;
    unsigned long long clock = 0;

    // The calling thread has `device` current.
    Entry *checkout(int device, Dims box, bool enabled) {
        std::unique_ptr<Entry> doomed;  // destroyed after the lock is released (its destructor waits for the device)
        Entry *e = nullptr;
        {
            std::lock_guard<std::mutex> lock(mu);
            if (enabled) {
                size_t on_device = 0;
                for (auto &c : entries) {
                    if (c->device != device) continue;
                    ++on_device;
                    if (!e && !c->busy && c->box == box) e = c.get();
                }
                if (!e && on_device >= kEntriesPerDevice) {  // evict the least recently used idle entry of this device
                    size_t lru = entries.size();
                    for (size_t i = 0; i < entries.size(); ++i)
                        if (entries[i]->device == device && !entries[i]->busy &&
                            (lru == entries.size() || entries[i]->used < entries[lru]->used))
                            lru = i;
                    if (lru != entries.size()) {
                        doomed = std::move(entries[lru]);
                        entries.erase(entries.begin() + (long)lru);
                        --on_device;
                    }
                }
                if (!e && on_device < kEntriesPerDevice) {
                    entries.emplace_back(new Entry());
                    e = entries.back().get();
                    e->device = device;
                    e->box = box;
                }
            }
            if (e) {
                e->busy = true;
                e->used = ++clock;
            }
        }
        doomed.reset();
        if (!e) {  // cache off, or every entry of this device busy with another box: private
            e = new Entry();
            e->device = device;
            e->box = box;
            e->busy = true;
            e->cached = false;
        }
        try {
            if (!e->eng) e->eng.reset(new DeconEngine(box, nullptr));
        } catch (...) {
            checkin(e);
            throw;
        }
        return e;
    }
    void checkin(Entry *e) {
        if (!e->cached) {
            delete e;
            return;
        }
        std::lock_guard<std::mutex> lock(mu);
        e->busy = false;
    }
    // Destroys every idle entry; busy ones finish their call first and stay (their owners may drop them later).
    void drop_idle() {
        std::vector<std::unique_ptr<Entry>> doomed;
        {
            std::lock_guard<std::mutex> lock(mu);
            for (size_t i = 0; i < entries.size();)
                if (!entries[i]->busy) {
                    doomed.push_back(std::move(entries[i]));
                    entries.erase(entries.begin() + (long)i);
                } else {
                    ++i;
                }
        }
        doomed.clear();
    }
};

DeconCache &decon_cache() {
    static DeconCache *c = new DeconCache();  // never destroyed: see devpool.cpp
    return *c;
}

bool cache_enabled() {
    const char *v = getenv("MIL_DECON_CACHE");
    return !(v && v[0] == '0');
}

// RAII check-out of a cache entry for one libapi.h call.
struct CacheLease {
    DeconCache::Entry *e;
    CacheLease(int device, Dims box) : e(decon_cache().checkout(device, box, cache_enabled())) {}
    ~CacheLease() { decon_cache().checkin(e); }
    CacheLease(const CacheLease &) = delete;
    CacheLease &operator=(const CacheLease &) = delete;
    DeconEngine &engine() { return *e->eng; }
};

// OTF of (psf, flip) for the cached engine: reused when the same PSF bytes were seen before, made otherwise.
const float2 *cached_otf(DeconCache::Entry &c, DeconEngine &eng, const float *h_psf, bool flip, float *d_psf_stage,
                         const Geometry &g) {
    constexpr size_t kMaxOtfs = 8;  // two views x (OTF, back projector) x two PSF sets
    for (auto &o : c.otfs)
        if (o->flip == flip && o->dims == g.psf && std::memcmp(o->psf.data(), h_psf, g.n_psf * sizeof(float)) == 0) {
            o->stamp = ++c.clock;
            return o->buf.get();
        }
    if (c.otfs.size() >= kMaxOtfs) {  // evict the least recently used
        size_t lru = 0;
        for (size_t i = 1; i < c.otfs.size(); ++i)
            if (c.otfs[i]->stamp < c.otfs[lru]->stamp) lru = i;
        eng.synchronize();
        c.otfs.erase(c.otfs.begin() + (long)lru);
    }
    std::unique_ptr<DeconCache::Otf> o(new DeconCache::Otf());
    o->psf.assign(h_psf, h_psf + g.n_psf);
    o->dims = g.psf;
    o->flip = flip;
    o->buf.alloc(eng.spectrum_count());
    o->stamp = ++c.clock;
    MIL_HIP(hipMemcpyAsync(d_psf_stage, h_psf, g.n_psf * sizeof(float), hipMemcpyHostToDevice, eng.stream()));
    eng.gen_otf(o->buf.get(), d_psf_stage, g.psf, flip);
    eng.synchronize();  // d_psf_stage is reused by the next OTF
    c.otfs.push_back(std::move(o));
    return c.otfs.back()->buf.get();
}

// OTF + back-projector OTF for one view (reference src/api_decon.cpp:211-223): matched back projector = flipped PSF.
void view_otfs(DeconCache::Entry &c, DeconEngine &eng, const float2 *&otf, const float2 *&otf_bp, float *d_psf_stage,
               const float *h_psf, const float *h_psf_bp, bool unmatch, const Geometry &g) {
    otf = cached_otf(c, eng, h_psf, false, d_psf_stage, g);
    otf_bp = unmatch ? cached_otf(c, eng, h_psf_bp, false, d_psf_stage, g) : cached_otf(c, eng, h_psf, true, d_psf_stage, g);
}

}  // namespace

namespace mil {
void decon_cache_drop() { decon_cache().drop_idle(); }
}  // namespace mil

// Where the seconds of this thread's last decon_singleview / decon_dualview call went (mil_decon_last_call_split): the
// records[10] of the reference's interface have no room for it.  {engine + buffers, OTFs, upload + initial estimate, RL
// loop, download}; the stream is synchronised at each boundary (four waits of a few microseconds per call).
static thread_local double g_call_split[5] = {0, 0, 0, 0, 0};
extern "C" int mil_decon_last_call_split(double *out5) {
    for (int i = 0; i < 5; ++i) out5[i] = g_call_split[i];
    return 0;
}

extern "C" int decon_singleview(float *h_decon, float *h_img, unsigned int *imSize, float *h_psf,
                                unsigned int *psfSize, bool flagConstInitial, int itNumForDecon, int deviceNum,
                                int gpuMemMode, bool verbose, float *deconRecords, bool flagUnmatch,
                                float *h_psf_bp) {
    if (gpuMemMode < -1 || gpuMemMode > 2) {
        printf("\n****Wrong gpuMemMode setup, no deconvolution performed !!! ****\n");
        return 1;  // reference src/api_decon.cpp:316-318
    }
    return guarded("decon_singleview", [&]() -> int {
        TraceRange trace("libapi::decon_singleview");
        const double t_start = now_s();
        const Geometry g = make_geometry(imSize, psfSize);
        if (gpuMemMode == 0)
            return run_cpu(h_decon, h_img, nullptr, h_psf, nullptr, h_psf_bp, nullptr, g, false, flagConstInitial,
                           itNumForDecon, verbose, flagUnmatch, deconRecords, t_start);
        MIL_HIP(hipSetDevice(deviceNum));
        if (verbose) {
            printf("Image information:\n");
            printf("...Image size %lld x %lld x %lld\n", g.im.nx, g.im.ny, g.im.nz);
            printf("...PSF size %lld x %lld x %lld\n", g.psf.nx, g.psf.ny, g.psf.nz);
            printf("...FFT size %lld x %lld x %lld\n", g.box.nx, g.box.ny, g.box.nz);
        }
        deconRecords[1] = free_mb();
        // A, E, T (real) + OTF, OTF_bp, S (spectra) + PSF staging + FFT work area (~1 spectrum)
        const size_t need = (3 * g.n_box + g.n_psf + (g.padded ? g.n_im : 0)) * sizeof(float) +
                            4 * g.n_spec * sizeof(float2);
        const int mode = resolve_mode(gpuMemMode, need, verbose);
        if (mode == 0)
            return run_cpu(h_decon, h_img, nullptr, h_psf, nullptr, h_psf_bp, nullptr, g, false, flagConstInitial,
                           itNumForDecon, verbose, flagUnmatch, deconRecords, t_start);
        deconRecords[0] = (float)mode;
        const double t_init = now_s();

        CacheLease lease(deviceNum, g.box);  // this call's engine + OTFs; other threads / devices proceed in parallel
        DeconCache::Entry &cache = *lease.e;
        DeconEngine &eng = lease.engine();
        hipStream_t s = eng.stream();
        DeviceBuffer<float> dA(g.n_box), dE(g.n_box), dPsf(g.n_psf), dStage(g.padded ? g.n_im : 0);
        deconRecords[2] = free_mb();

        const double t_ready = now_s();
        const float2 *dOtf, *dOtfBp;
        view_otfs(cache, eng, dOtf, dOtfBp, dPsf, h_psf, h_psf_bp, flagUnmatch, g);
        MIL_HIP(hipStreamSynchronize(s));
        const double t_otf = now_s();
        upload_volume(dA, dStage, h_img, g, s);
        eng.init_single(dA, dE, flagConstInitial);
        MIL_HIP(hipStreamSynchronize(s));
        const double t_loop = now_s();
        eng.run_single(dE, dA, dOtf, dOtfBp, itNumForDecon);
        eng.synchronize();
        const double t_done = now_s();
        if (verbose) printf("...Time cost for decon is %2.3f s\n", t_done - t_loop);
        deconRecords[3] = free_mb();
        download_volume(h_decon, dE, dStage, g, s);
        deconRecords[4] = free_mb();
        const double t_end = now_s();
        deconRecords[5] = deconRecords[4];
        deconRecords[6] = (float)(t_init - t_start);
        deconRecords[7] = (float)(t_loop - t_init);
        deconRecords[8] = (float)(t_done - t_loop);
        deconRecords[9] = (float)(t_end - t_start);
        const double split[5] = {t_ready - t_init, t_otf - t_ready, t_loop - t_otf, t_done - t_loop, t_end - t_done};
        for (int i = 0; i < 5; ++i) g_call_split[i] = split[i];
        return 0;
    });
}

extern "C" int decon_dualview(float *h_decon, float *h_img1, float *h_img2, unsigned int *imSize, float *h_psf1,
                              float *h_psf2, unsigned int *psfSize, bool flagConstInitial, int itNumForDecon,
                              int deviceNum, int gpuMemMode, bool verbose, float *deconRecords, bool flagUnmatch,
                              float *h_psf_bp1, float *h_psf_bp2) {
    if (gpuMemMode < -1 || gpuMemMode > 2) {
        printf("\n****Wrong gpuMemMode setup, no deconvolution performed !!! ****\n");
        return -1;  // reference src/api_decon.cpp:685-687
    }
    return guarded("decon_dualview", [&]() -> int {
        TraceRange trace("libapi::decon_dualview");
        const double t_start = now_s();
        const Geometry g = make_geometry(imSize, psfSize);
        if (gpuMemMode == 0)
            return run_cpu(h_decon, h_img1, h_img2, h_psf1, h_psf2, h_psf_bp1, h_psf_bp2, g, true, flagConstInitial,
                           itNumForDecon, verbose, flagUnmatch, deconRecords, t_start);
        MIL_HIP(hipSetDevice(deviceNum));
        if (verbose) {
            printf("\tImage information:\n");
            printf("\t... Image size %lld x %lld x %lld\n", g.im.nx, g.im.ny, g.im.nz);
            printf("\t... PSF size %lld x %lld x %lld\n", g.psf.nx, g.psf.ny, g.psf.nz);
            printf("\t... FFT size %lld x %lld x %lld\n", g.box.nx, g.box.ny, g.box.nz);
        }
        deconRecords[1] = free_mb();
        const size_t need = (4 * g.n_box + g.n_psf + (g.padded ? g.n_im : 0)) * sizeof(float) +
                            6 * g.n_spec * sizeof(float2);
        const int mode = resolve_mode(gpuMemMode, need, verbose);
        if (mode == 0)
            return run_cpu(h_decon, h_img1, h_img2, h_psf1, h_psf2, h_psf_bp1, h_psf_bp2, g, true, flagConstInitial,
                           itNumForDecon, verbose, flagUnmatch, deconRecords, t_start);
        deconRecords[0] = (float)mode;
        const double t_init = now_s();

        CacheLease lease(deviceNum, g.box);  // this call's engine + OTFs; other threads / devices proceed in parallel
        DeconCache::Entry &cache = *lease.e;
        DeconEngine &eng = lease.engine();
        hipStream_t s = eng.stream();
        DeviceBuffer<float> dA(g.n_box), dB(g.n_box), dE(g.n_box), dPsf(g.n_psf), dStage(g.padded ? g.n_im : 0);
        deconRecords[2] = free_mb();

        const double t_ready = now_s();
        const float2 *dOtf1, *dOtfBp1, *dOtf2, *dOtfBp2;
        view_otfs(cache, eng, dOtf1, dOtfBp1, dPsf, h_psf1, h_psf_bp1, flagUnmatch, g);  // reference :522-534
        view_otfs(cache, eng, dOtf2, dOtfBp2, dPsf, h_psf2, h_psf_bp2, flagUnmatch, g);  // :535-547
        MIL_HIP(hipStreamSynchronize(s));
        const double t_otf = now_s();
        upload_volume(dA, dStage, h_img1, g, s);
        if (g.padded) MIL_HIP(hipStreamSynchronize(s));  // dStage is reused for the second view
        upload_volume(dB, dStage, h_img2, g, s);
        eng.init_dual(dA, dB, dE, flagConstInitial);
        MIL_HIP(hipStreamSynchronize(s));
        const double t_loop = now_s();
        eng.run_dual(dE, dA, dB, dOtf1, dOtf2, dOtfBp1, dOtfBp2, itNumForDecon);
        eng.synchronize();
        const double t_done = now_s();
        deconRecords[3] = free_mb();
        download_volume(h_decon, dE, dStage, g, s);
        deconRecords[4] = free_mb();
        const double t_end = now_s();
        deconRecords[5] = deconRecords[4];
        deconRecords[6] = (float)(t_init - t_start);
        deconRecords[7] = (float)(t_loop - t_init);
        deconRecords[8] = (float)(t_done - t_loop);
        deconRecords[9] = (float)(t_end - t_start);
        const double split[5] = {t_ready - t_init, t_otf - t_ready, t_loop - t_otf, t_done - t_loop, t_end - t_done};
        for (int i = 0; i < 5; ++i) g_call_split[i] = split[i];
        if (verbose) printf("\t... deconvolution done !!! \n");
        return 0;
    });
}

// ---- the same calls on device-resident volumes (include/mil_device.h) -------------------------------------------------------
// A host that keeps its stacks in HBM (the time-lapse batch tool) runs the stages of decon_singleview / decon_dualview without
// the two PCIe trips: same engine and OTF cache, same pad / initial estimate / RL loop / crop, hence the same bits.  The
// images are device pointers of imSize floats (read only: the engine works on its own padded copies), the PSFs stay host
// pointers (small, and the OTF cache finds them by content); d_decon receives imSize floats.  GPU mode only; records as in
// the host-pointer calls.  Runs on the cached engine's own stream, which first waits for `stream` (the producer of the
// images); the call returns when d_decon is complete.
namespace {

void device_in(float *d_dst, const float *d_src, const Geometry &g, hipStream_t s) {
    if (g.padded) k::pad_stack(d_dst, d_src, g.box, g.im, s);  // reference src/api_decon.cpp:225-231
    else MIL_HIP(hipMemcpyAsync(d_dst, d_src, g.n_im * sizeof(float), hipMemcpyDeviceToDevice, s));
}

void device_out(float *d_dst, const float *d_src, const Geometry &g, hipStream_t s) {
    if (g.padded) k::crop_stack(d_dst, d_src, g.im, g.box, s);  // :237-243
    else MIL_HIP(hipMemcpyAsync(d_dst, d_src, g.n_im * sizeof(float), hipMemcpyDeviceToDevice, s));
    MIL_HIP(hipStreamSynchronize(s));
}

void wait_for_producer(hipStream_t engine_stream, hipStream_t producer) {
    if (producer == engine_stream) return;
    hipEvent_t ev;
    MIL_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    const hipError_t e1 = hipEventRecord(ev, producer);
    const hipError_t e2 = e1 == hipSuccess ? hipStreamWaitEvent(engine_stream, ev, 0) : e1;
    (void)hipEventDestroy(ev);
    MIL_HIP(e2);
}

int decon_device(float *d_decon, const float *d_img1, const float *d_img2, const unsigned int *imSize, const float *h_psf1,
                 const float *h_psf2, const unsigned int *psfSize, bool dual, bool flagConstInitial, int iters, bool verbose,
                 float *deconRecords, bool flagUnmatch, const float *h_psf_bp1, const float *h_psf_bp2, hipStream_t producer) {
    const double t_start = now_s();
    const Geometry g = make_geometry(imSize, psfSize);
    int device = 0;
    MIL_HIP(hipGetDevice(&device));
    deconRecords[1] = free_mb();
    deconRecords[0] = 1.f;
    const double t_init = now_s();
    CacheLease lease(device, g.box);
    DeconCache::Entry &cache = *lease.e;
    DeconEngine &eng = lease.engine();
    hipStream_t s = eng.stream();
    DeviceBuffer<float> dA(g.n_box), dB(dual ? g.n_box : 0), dE(g.n_box), dPsf(g.n_psf);
    deconRecords[2] = free_mb();
    const float2 *dOtf1, *dOtfBp1, *dOtf2 = nullptr, *dOtfBp2 = nullptr;
    view_otfs(cache, eng, dOtf1, dOtfBp1, dPsf, h_psf1, h_psf_bp1, flagUnmatch, g);
    if (dual) view_otfs(cache, eng, dOtf2, dOtfBp2, dPsf, h_psf2, h_psf_bp2, flagUnmatch, g);
    wait_for_producer(s, producer);
    device_in(dA, d_img1, g, s);
    if (dual) {
        device_in(dB, d_img2, g, s);
        eng.init_dual(dA, dB, dE, flagConstInitial);
    } else {
        eng.init_single(dA, dE, flagConstInitial);
    }
    MIL_HIP(hipStreamSynchronize(s));
    const double t_loop = now_s();
    if (dual) eng.run_dual(dE, dA, dB, dOtf1, dOtf2, dOtfBp1, dOtfBp2, iters);
    else eng.run_single(dE, dA, dOtf1, dOtfBp1, iters);
    eng.synchronize();
    const double t_done = now_s();
    if (verbose && !dual) printf("...Time cost for decon is %2.3f s\n", t_done - t_loop);
    deconRecords[3] = free_mb();
    device_out(d_decon, dE, g, s);
    deconRecords[4] = free_mb();
    const double t_end = now_s();
    deconRecords[5] = deconRecords[4];
    deconRecords[6] = (float)(t_init - t_start);
    deconRecords[7] = (float)(t_loop - t_init);
    deconRecords[8] = (float)(t_done - t_loop);
    deconRecords[9] = (float)(t_end - t_start);
    if (verbose && dual) printf("\t... deconvolution done !!! \n");
    return 0;
}

}  // namespace

extern "C" int mil_decon_singleview(float *d_decon, const float *d_img, const unsigned int *imSize, const float *h_psf,
                                    const unsigned int *psfSize, int flagConstInitial, int itNumForDecon, int verbose,
                                    float *deconRecords, int flagUnmatch, const float *h_psf_bp, void *stream) {
    return guarded("mil_decon_singleview", [&]() -> int {
        TraceRange trace("mil_decon_singleview");
        return decon_device(d_decon, d_img, nullptr, imSize, h_psf, nullptr, psfSize, false, flagConstInitial != 0, itNumForDecon,
                            verbose != 0, deconRecords, flagUnmatch != 0, h_psf_bp, nullptr, (hipStream_t)stream);
    });
}

extern "C" int mil_decon_dualview(float *d_decon, const float *d_img1, const float *d_img2, const unsigned int *imSize,
                                  const float *h_psf1, const float *h_psf2, const unsigned int *psfSize, int flagConstInitial,
                                  int itNumForDecon, int verbose, float *deconRecords, int flagUnmatch, const float *h_psf_bp1,
                                  const float *h_psf_bp2, void *stream) {
    return guarded("mil_decon_dualview", [&]() -> int {
        TraceRange trace("mil_decon_dualview");
        return decon_device(d_decon, d_img1, d_img2, imSize, h_psf1, h_psf2, psfSize, true, flagConstInitial != 0, itNumForDecon,
                            verbose != 0, deconRecords, flagUnmatch != 0, h_psf_bp1, h_psf_bp2, (hipStream_t)stream);
    });
}

// Dead upstream (its guard at src/api_decon.cpp:1133 is always true, so it always returns 1
// before doing any work); kept so the symbol resolves.
extern "C" int fusion_dualview(float *, float *, float *, float *, float *, float *, float *, unsigned int *,
                               unsigned int *, float *, float *, int, bool, int, float, int, float *, float *,
                               unsigned int *, int, int, int, bool, float *, bool, float *, float *) {
    printf("\n****Wrong gpuMemMode setup, processing stopped !!! ****\n");
    return 1;
}

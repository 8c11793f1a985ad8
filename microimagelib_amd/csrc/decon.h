// Device-resident Richardson-Lucy engine (single- and dual-view) for MI355X.
//
// One engine = one FFT box on one GPU, one HIP stream, scratch allocated once.  Two
// transform backends behind the same interface:
//   kBackendFused   power-of-two boxes: our own LDS-tiled transforms with the OTF product,
//                   the RL ratio and the estimate update fused in (fftconv.h); 8 launches
//                   per view update; spectra / OTFs in a private layout.
//   kBackendRocfft  any box: rocFFT real 3-D plans + the element-wise kernels of kernels.hip;
//                   spectra in rocFFT's natural [nz][ny][nx/2+1] layout.
// Neither path synchronises with the host inside the RL loop (the reference issues
// cudaThreadSynchronize after every kernel, src/api_subfunc.cu:248 etc.).
#pragma once

#include <rocfft/rocfft.h>

#include <vector>

#include "fftconv.h"
#include "kernels.h"

namespace mil {

// snapTransformSize, reference src/api_subfunc.cu:57-87.
int snap_transform_size(int n);

// Drops what the libapi.h deconvolution entry points keep between calls (engine, OTFs): api_decon.cpp.
void decon_cache_drop();

class Fft3D {
public:
    Fft3D(Dims box, hipStream_t stream);
    ~Fft3D();
    Fft3D(const Fft3D &) = delete;
    Fft3D &operator=(const Fft3D &) = delete;
    void r2c(float *in, float2 *out);  // unnormalised, out-of-place
    void c2r(float2 *in, float *out);  // unnormalised, out-of-place (may clobber `in`)
    size_t work_bytes() const { return work_.size(); }

private:
    rocfft_plan fwd_ = nullptr, inv_ = nullptr;
    rocfft_execution_info info_ = nullptr;
    DeviceBuffer<char> work_;
};

struct KernelTimer {  // HIP-event bracket around one kernel family, on the engine's stream
    bool enabled = false;
    int every = 1;        // bracket every n-th launch of the family only (an event pair costs a few us of stream time)
    int seen = 0;         // launches seen since enable
    bool active = false;  // the current launch is bracketed
    std::vector<hipEvent_t> ev;  // start/stop pairs
    size_t used = 0;
    void begin(hipStream_t s);
    void end(hipStream_t s);
    // sums elapsed over recorded pairs; stream must be idle. Returns count.
    size_t collect(double *total_ms);
    void reset() { used = 0; seen = 0; active = false; }
    ~KernelTimer();
};

class DeconEngine {
public:
    // Timer families.  rocFFT backend: rl_update, rl_ratio, cmul, rocFFT executes.
    // Fused backend: x pass with update, x pass with ratio, z convolution, y passes.
    enum Timed { kTimeUpdate = 0, kTimeRatio = 1, kTimeCmul = 2, kTimeFft = 3, kTimedCount = 4 };
    enum Backend { kBackendAuto = 0, kBackendRocfft = 1, kBackendFused = 2 };

    // backend kBackendAuto: fused when the box allows it (env MIL_FFT_BACKEND=rocfft|fused overrides).
    DeconEngine(Dims fft_box, hipStream_t stream /* nullptr: engine-owned stream */, int backend = kBackendAuto);
    ~DeconEngine();
    DeconEngine(const DeconEngine &) = delete;
    DeconEngine &operator=(const DeconEngine &) = delete;

    Dims box() const { return box_; }
    size_t real_count() const { return (size_t)box_.count(); }
    int backend() const { return plan_ ? kBackendFused : kBackendRocfft; }
    // fused backend: 0 classic four-kernel half update, 3 two-pass (csrc/fft2p.hip), 1 / 2 mixed (tests)
    int two_pass() const { return two_pass_; }
    size_t natural_spectrum_count() const { return (size_t)(box_.nx / 2 + 1) * box_.ny * box_.nz; }
    // complex64 elements of a spectrum / OTF buffer of this engine (layout is backend-private)
    size_t spectrum_count() const { return plan_ ? plan_->spectrum_count() : natural_spectrum_count(); }
    hipStream_t stream() const { return stream_; }

    // genOTFgpu (+ flipgpu when `flip`), reference src/api_subfunc.cu:3269-3307.
    void gen_otf(float2 *d_otf, const float *d_psf, Dims psf_dims, bool flip, bool normalise = true);
    // estimate initialisation, reference :3380-3388 / :3606-3618.  A (and B) are clamped in place.
    void init_single(float *A, float *E, bool const_init);
    void init_dual(float *A, float *B, float *E, bool const_init);
    // one view update, reference :3406-3415.
    void view_update(float *E, const float *M, const float2 *otf, const float2 *otf_bp);
    void run_single(float *E, const float *A, const float2 *otf, const float2 *otf_bp, int iters);
    void run_dual(float *E, const float *A, const float *B, const float2 *otf1, const float2 *otf2,
                  const float2 *otf_bp1, const float2 *otf_bp2, int iters);
    // out = irfftn(rfftn(in) * otf), unnormalised (= N * circular convolution); test / utility entry.
    void convolve(float *out, const float *in, const float2 *otf);
    // OTF of this engine -> natural rocFFT/numpy order [nz][ny][nx/2+1]
    void otf_to_natural(float2 *natural, const float2 *otf);
    void synchronize();

    KernelTimer timers[kTimedCount];
    float *scratch_real() { return T_.get(); }

private:
    Dims box_;
    hipStream_t stream_ = nullptr;
    bool own_stream_ = false;
    Fft3D *fft_ = nullptr;          // rocFFT backend
    fx::ConvPlan *plan_ = nullptr;  // fused backend
    int two_pass_ = 0;              // fused backend, boxes of ConvPlan::two_pass_supported: bit 0 yz kernel, bit 1 xy kernel (csrc/fft2p*.hip)
    static constexpr int kTwoPassDefault = 3;
    void run_fused(float *E, const float *const *M, const float2 *const *otf, const float2 *const *otf_bp, int views,
                   int iters);
    void enqueue_iteration(float *E, const float *const *M, const float2 *const *otf, const float2 *const *otf_bp,
                           int views, bool last_iteration);

    DeviceBuffer<float> T_;
    DeviceBuffer<float2> S_;
    k::ReduceWorkspace red_;
};

}  // namespace mil

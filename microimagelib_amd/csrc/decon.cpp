#include "decon.h"
#include "trace.h"

#include <cstring>
#include <mutex>

namespace mil {

// ---------------------------------------------------------------------------
// FFT-friendly size rule, reference src/api_subfunc.cu:57-87: align up to 16; keep
// powers of two; below 128 go to the next power of two; otherwise align up to 64.
// ---------------------------------------------------------------------------
int snap_transform_size(int n) {
    auto align_up = [](int a, int b) { return (a % b != 0) ? (a - a % b + b) : a; };
    n = align_up(n, 16);
    int hi_bit = 31;
    while (hi_bit >= 0 && !((unsigned)n & (1u << hi_bit))) --hi_bit;
    if (hi_bit < 0) return n;
    const unsigned low_pot = 1u << hi_bit;
    if (low_pot == (unsigned)n) return n;
    const unsigned hi_pot = 1u << (hi_bit + 1);
    if (hi_pot <= 128u) return (int)hi_pot;
    return align_up(n, 64);
}

// ---------------------------------------------------------------------------
// rocFFT plumbing
// ---------------------------------------------------------------------------
namespace {

void check_fft(rocfft_status st, const char *what) {
    if (st != rocfft_status_success) {
        char buf[256];
        snprintf(buf, sizeof buf, "rocFFT: %s failed with status %d", what, (int)st);
        throw Error(-2, buf);
    }
}

void ensure_rocfft() {
    static std::once_flag once;
    // never paired with rocfft_cleanup(): the process may share librocfft with PyTorch.
    std::call_once(once, [] { check_fft(rocfft_setup(), "rocfft_setup"); });
}

}  // namespace

Fft3D::Fft3D(Dims box, hipStream_t stream) {
    ensure_rocfft();
    const size_t lengths[3] = {(size_t)box.nx, (size_t)box.ny, (size_t)box.nz};  // fastest first
    check_fft(rocfft_plan_create(&fwd_, rocfft_placement_notinplace, rocfft_transform_type_real_forward,
                                 rocfft_precision_single, 3, lengths, 1, nullptr),
              "plan_create(real_forward)");
    check_fft(rocfft_plan_create(&inv_, rocfft_placement_notinplace, rocfft_transform_type_real_inverse,
                                 rocfft_precision_single, 3, lengths, 1, nullptr),
              "plan_create(real_inverse)");
    size_t wf = 0, wi = 0;
    check_fft(rocfft_plan_get_work_buffer_size(fwd_, &wf), "work_buffer_size");
    check_fft(rocfft_plan_get_work_buffer_size(inv_, &wi), "work_buffer_size");
    const size_t w = wf > wi ? wf : wi;
    check_fft(rocfft_execution_info_create(&info_), "execution_info_create");
    if (w) {
        work_.alloc(w);
        check_fft(rocfft_execution_info_set_work_buffer(info_, work_.get(), w), "set_work_buffer");
    }
    check_fft(rocfft_execution_info_set_stream(info_, stream), "set_stream");
}

Fft3D::~Fft3D() {
    if (info_) rocfft_execution_info_destroy(info_);
    if (fwd_) rocfft_plan_destroy(fwd_);
    if (inv_) rocfft_plan_destroy(inv_);
}

void Fft3D::r2c(float *in, float2 *out) {
    void *ib[1] = {in}, *ob[1] = {out};
    check_fft(rocfft_execute(fwd_, ib, ob, info_), "execute(real_forward)");
}

void Fft3D::c2r(float2 *in, float *out) {
    void *ib[1] = {in}, *ob[1] = {out};
    check_fft(rocfft_execute(inv_, ib, ob, info_), "execute(real_inverse)");
}

// ---------------------------------------------------------------------------
// HIP-event kernel timers
// ---------------------------------------------------------------------------
void KernelTimer::begin(hipStream_t s) {
    if (!enabled) return;
    active = (seen++ % (every > 0 ? every : 1)) == 0;
    if (!active) return;
    if (used + 2 > ev.size()) {
        for (int i = 0; i < 2; ++i) {
            hipEvent_t e;
            MIL_HIP(hipEventCreate(&e));
            ev.push_back(e);
        }
    }
    MIL_HIP(hipEventRecord(ev[used], s));
}

void KernelTimer::end(hipStream_t s) {
    if (!enabled || !active) return;
    MIL_HIP(hipEventRecord(ev[used + 1], s));
    used += 2;
}

size_t KernelTimer::collect(double *total_ms) {
    double t = 0.0;
    for (size_t i = 0; i + 1 < used; i += 2) {
        float ms = 0.f;
        MIL_HIP(hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
        t += ms;
    }
    *total_ms = t;
    return used / 2;
}

KernelTimer::~KernelTimer() {
    for (hipEvent_t e : ev) (void)hipEventDestroy(e);
}

// ---------------------------------------------------------------------------
// engine
// ---------------------------------------------------------------------------
DeconEngine::DeconEngine(Dims fft_box, hipStream_t stream, int backend) : box_(fft_box), stream_(stream) {
    if (box_.nx <= 0 || box_.ny <= 0 || box_.nz <= 0) throw Error(-1, "DeconEngine: empty FFT box");
    if (box_.nx % 2) throw Error(-1, "DeconEngine: FFT box x extent must be even");
    if (backend == kBackendAuto) {
        const char *env = getenv("MIL_FFT_BACKEND");
        if (env && !strcmp(env, "rocfft")) backend = kBackendRocfft;
        else if (env && !strcmp(env, "fused")) backend = kBackendFused;
        else backend = fx::ConvPlan::supported(box_) ? kBackendFused : kBackendRocfft;
    }
    if (backend != kBackendRocfft && backend != kBackendFused) throw Error(-1, "DeconEngine: unknown backend");
    if (backend == kBackendFused && !fx::ConvPlan::supported(box_))
        throw Error(-1, "DeconEngine: the fused backend needs a power-of-two box (16..1024, x up to 2048)");
    if (!stream_) {
        MIL_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
        own_stream_ = true;
    }
    try {
        if (backend == kBackendFused) {
            plan_ = new fx::ConvPlan(box_);
            if (plan_->two_pass()) {  // MIL_FX_TWOPASS: 0 classic, 1 z kernel only, 2 x kernel only, 3 both
                const char *tp = getenv("MIL_FX_TWOPASS");
                two_pass_ = tp ? (atoi(tp) & 3) : kTwoPassDefault;
                plan_->set_yz_otf_layout(two_pass_ & 1);  // OTFs tiled per yz workgroup
            }
        }
        else fft_ = new Fft3D(box_, stream_);
        T_.alloc(real_count());
        S_.alloc(spectrum_count());
        // fused backend: the x pass never writes the padding columns of S; zero once, they stay zero
        MIL_HIP(hipMemsetAsync(S_.get(), 0, spectrum_count() * sizeof(float2), stream_));
    } catch (...) {
        delete plan_;
        delete fft_;
        if (own_stream_) (void)hipStreamDestroy(stream_);
        throw;
    }
}

DeconEngine::~DeconEngine() {
    delete fft_;
    delete plan_;
    if (own_stream_ && stream_) (void)hipStreamDestroy(stream_);
}

void DeconEngine::synchronize() { MIL_HIP(hipStreamSynchronize(stream_)); }

void DeconEngine::gen_otf(float2 *d_otf, const float *d_psf, Dims P, bool flip, bool normalise) {
    TraceRange trace("mil::gen_otf");
    float scale = 1.0f;
    if (normalise) {
        const double sum = k::reduce_sync(k::kSum, d_psf, (size_t)P.count(), 0.f, red_, stream_);  // :3283
        scale = (float)(1 / sum);                                                                 // :3284
    }
    const bool too_big = (box_.nx < P.nx) || (box_.ny < P.ny) || (box_.nz < P.nz);  // :3288
    if (!too_big) {
        k::pad_psf(T_.get(), d_psf, box_, P, scale, flip, stream_);  // normalise + flip + circular placement fused
    } else {
        // PSF exceeds the box on some axis: centre-crop / zero-pad to the box first (alignsize3Dgpu),
        // then place with the box-sized origin, exactly as :3289-3290.
        DeviceBuffer<float> tmp((size_t)P.count()), aligned(real_count());
        const float *srcp = d_psf;
        DeviceBuffer<float> flipped;
        if (flip) {
            flipped.alloc((size_t)P.count());
            k::flip3(flipped.get(), d_psf, P, stream_);
            srcp = flipped.get();
        }
        k::scale(tmp.get(), srcp, scale, (size_t)P.count(), stream_);
        k::align_size(aligned.get(), tmp.get(), box_, P, stream_);
        k::pad_psf(T_.get(), aligned.get(), box_, box_, 1.0f, false, stream_);
        MIL_HIP(hipStreamSynchronize(stream_));  // temporaries die at scope exit
    }
    if (plan_) {  // same forward transforms as the RL loop -> same private ordering
        plan_->x_pass(fx::kXForward, S_.get(), T_.get(), nullptr, nullptr, 0.f, stream_);
        plan_->y_pass(S_.get(), false, stream_);
        plan_->z_forward_to_otf(S_.get(), d_otf, stream_);
    } else {
        fft_->r2c(T_.get(), d_otf);  // :3298-3299
    }
}

void DeconEngine::convolve(float *out, const float *in, const float2 *otf) {
    if (plan_ && plan_->yz_otf_layout()) {  // the engine's OTFs are tiled for the yz kernel: y head | yz_conv | y head^-1
        plan_->x_pass(fx::kXForward, S_.get(), in, nullptr, nullptr, 0.f, stream_);
        plan_->y_part(S_.get(), 0, stream_);
        plan_->yz_conv(S_.get(), otf, stream_);
        plan_->y_part(S_.get(), 1, stream_);
        plan_->x_pass(fx::kXInverse, S_.get(), nullptr, out, nullptr, 0.f, stream_);
    } else if (plan_) {
        plan_->x_pass(fx::kXForward, S_.get(), in, nullptr, nullptr, 0.f, stream_);
        plan_->y_pass(S_.get(), false, stream_);
        plan_->z_conv(S_.get(), otf, stream_);
        plan_->y_pass(S_.get(), true, stream_);
        plan_->x_pass(fx::kXInverse, S_.get(), nullptr, out, nullptr, 0.f, stream_);
    } else {
        if (in != T_.get()) MIL_HIP(hipMemcpyAsync(T_.get(), in, real_count() * sizeof(float), hipMemcpyDeviceToDevice, stream_));
        fft_->r2c(T_.get(), S_.get());
        k::cmul_inplace(S_.get(), otf, spectrum_count(), stream_);
        fft_->c2r(S_.get(), out);
    }
}

void DeconEngine::otf_to_natural(float2 *natural, const float2 *otf) {
    if (plan_) plan_->otf_to_natural(natural, otf, stream_);
    else MIL_HIP(hipMemcpyAsync(natural, otf, spectrum_count() * sizeof(float2), hipMemcpyDeviceToDevice, stream_));
}

void DeconEngine::init_single(float *A, float *E, bool const_init) {
    const size_t n = real_count();
    if (const_init) {
        k::clamp_min(A, A, n, kSmallValue, stream_);                                             // :3380
        const float mean_value = (float)k::reduce_sync(k::kSum, A, n, 0.f, red_, stream_);       // the SUM, :3382
        k::fill(E, mean_value, n, stream_);                                                      // :3383-3384
    } else {
        k::init_single(A, E, n, kSmallValue, stream_);  // :3380 + :3387 fused
    }
}

void DeconEngine::init_dual(float *A, float *B, float *E, bool const_init) {
    const size_t n = real_count();
    if (const_init) {
        k::clamp_min(A, A, n, kSmallValue, stream_);
        k::clamp_min(B, B, n, kSmallValue, stream_);
        const float m1 = (float)k::reduce_sync(k::kSum, A, n, 0.f, red_, stream_);  // :3610
        const float m2 = (float)k::reduce_sync(k::kSum, B, n, 0.f, red_, stream_);  // :3611
        k::fill(E, (m1 + m2) / 2, n, stream_);                                      // :3613
    } else {
        k::init_dual(A, B, E, n, kSmallValue, stream_);  // :3606-3607 + :3616-3617 fused
    }
}

// Fused backend: the estimate's x-spectrum is carried from one view update to the next, so a
// view update is  y, z-conv(OTF), y^-1, x^-1 | ratio | x, y, z-conv(OTF_bp), y^-1, x^-1 | update | x.
//
// x and y kernels walk the volume in z order; each one runs opposite to its predecessor so that it
// starts on the slabs written last (Infinity Cache hits instead of HBM reads).  The direction flips
// four times per view update, so every iteration starts in the same state.
void DeconEngine::enqueue_iteration(float *E, const float *const *M, const float2 *const *otf,
                                    const float2 *const *otf_bp, int views, bool last_iteration) {
    static const bool pingpong = !(getenv("MIL_FX_PINGPONG") && !strcmp(getenv("MIL_FX_PINGPONG"), "0"));
    float2 *S = S_.get();
    if (two_pass_) {
        // Two-pass path (csrc/fft2p.hip): S carries the y HEAD (the stages longer than 8) between kernels; the y tail
        // rides in the z kernel (bit 0) and the head in the x kernel (bit 1).  With one bit only, classic partial y passes
        // fill in for the other kernel (test / bisect configurations).
        const bool yz = two_pass_ & 1, xy = two_pass_ & 2;
        for (int v = 0; v < views; ++v) {
            const bool last = last_iteration && (v == views - 1);
            for (int half = 0; half < 2; ++half) {
                const float2 *o = half == 0 ? otf[v] : otf_bp[v];
                if (!yz) {
                    timers[kTimeFft].begin(stream_);
                    plan_->y_part(S, 2, stream_);
                    timers[kTimeFft].end(stream_);
                }
                timers[kTimeCmul].begin(stream_);
                if (yz) plan_->yz_conv(S, o, stream_);
                else plan_->z_conv(S, o, stream_, false);
                timers[kTimeCmul].end(stream_);
                if (!yz) {
                    timers[kTimeFft].begin(stream_);
                    plan_->y_part(S, 3, stream_);
                    timers[kTimeFft].end(stream_);
                }
                const fx::XMode mode = half == 0 ? fx::kXRatio : (last ? fx::kXUpdateLast : fx::kXUpdate);
                const int t = half == 0 ? kTimeRatio : kTimeUpdate;
                if (xy) {
                    timers[t].begin(stream_);
                    plan_->xy_pass(mode, S, E, M[v], kSmallValue, stream_);
                    timers[t].end(stream_);
                } else {
                    timers[kTimeFft].begin(stream_);
                    plan_->y_part(S, 1, stream_);
                    timers[kTimeFft].end(stream_);
                    timers[t].begin(stream_);
                    plan_->x_pass(mode, S, nullptr, half == 0 ? nullptr : E, half == 0 ? M[v] : nullptr, kSmallValue, stream_, false);
                    timers[t].end(stream_);
                    if (mode != fx::kXUpdateLast) {
                        timers[kTimeFft].begin(stream_);
                        plan_->y_part(S, 0, stream_);
                        timers[kTimeFft].end(stream_);
                    }
                }
            }
        }
        return;
    }
    bool dir = false;  // direction of the most recent z-ordered kernel (the x pass that produced S)
    auto next_dir = [&] { dir = pingpong ? !dir : false; return dir; };
    for (int v = 0; v < views; ++v) {
        const bool last = last_iteration && (v == views - 1);
        for (int half = 0; half < 2; ++half) {
            timers[kTimeFft].begin(stream_);
            plan_->y_pass(S, false, stream_, next_dir());
            timers[kTimeFft].end(stream_);
            timers[kTimeCmul].begin(stream_);
            plan_->z_conv(S, half == 0 ? otf[v] : otf_bp[v], stream_, false);  // :3406-3408 / :3411-3413
            timers[kTimeCmul].end(stream_);
            timers[kTimeFft].begin(stream_);
            plan_->y_pass(S, true, stream_, dir);  // z_conv touched every slab: keep the direction
            timers[kTimeFft].end(stream_);
            if (half == 0) {
                timers[kTimeRatio].begin(stream_);
                plan_->x_pass(fx::kXRatio, S, nullptr, nullptr, M[v], 0.f, stream_, next_dir());  // :3409
                timers[kTimeRatio].end(stream_);
            } else {
                timers[kTimeUpdate].begin(stream_);
                plan_->x_pass(last ? fx::kXUpdateLast : fx::kXUpdate, S, nullptr, E, nullptr, kSmallValue, stream_,
                              next_dir());  // :3414-3415
                timers[kTimeUpdate].end(stream_);
            }
        }
    }
}

void DeconEngine::run_fused(float *E, const float *const *M, const float2 *const *otf, const float2 *const *otf_bp,
                            int views, int iters) {
    if (iters <= 0) return;
    plan_->x_pass(fx::kXForward, S_.get(), E, nullptr, nullptr, 0.f, stream_, false);
    if (two_pass_) plan_->y_part(S_.get(), 0, stream_);
    // Plain asynchronous launches.  Replaying one captured iteration as a HIP graph was measured and dropped:
    // 918 vs 915 it/s at 512x512x256 (kernels of 90-170 us hide the launches anyway) and SLOWER on small boxes
    // (64^3: 19.9k vs 27.1k it/s -- a replay costs more than eight queued launches).
    for (int it = 0; it < iters; ++it) enqueue_iteration(E, M, otf, otf_bp, views, it == iters - 1);
}

void DeconEngine::view_update(float *E, const float *M, const float2 *otf, const float2 *otf_bp) {
    if (plan_) {
        run_fused(E, &M, &otf, &otf_bp, 1, 1);
        return;
    }
    const size_t n = real_count(), ns = spectrum_count();
    float *T = T_.get();
    float2 *S = S_.get();
    timers[kTimeFft].begin(stream_);
    fft_->r2c(E, S);  // :3406
    timers[kTimeFft].end(stream_);
    timers[kTimeCmul].begin(stream_);
    k::cmul_inplace(S, otf, ns, stream_);  // :3407
    timers[kTimeCmul].end(stream_);
    timers[kTimeFft].begin(stream_);
    fft_->c2r(S, T);  // :3408
    timers[kTimeFft].end(stream_);
    timers[kTimeRatio].begin(stream_);
    k::rl_ratio(T, M, n, stream_);  // :3409
    timers[kTimeRatio].end(stream_);
    timers[kTimeFft].begin(stream_);
    fft_->r2c(T, S);  // :3411
    timers[kTimeFft].end(stream_);
    timers[kTimeCmul].begin(stream_);
    k::cmul_inplace(S, otf_bp, ns, stream_);  // :3412
    timers[kTimeCmul].end(stream_);
    timers[kTimeFft].begin(stream_);
    fft_->c2r(S, T);  // :3413
    timers[kTimeFft].end(stream_);
    timers[kTimeUpdate].begin(stream_);
    k::rl_update(E, T, n, kSmallValue, stream_);  // :3414-3415 fused
    timers[kTimeUpdate].end(stream_);
}

void DeconEngine::run_single(float *E, const float *A, const float2 *otf, const float2 *otf_bp, int iters) {
    TraceRange trace("mil::rl_single_view (enqueue)");
    if (plan_) {
        run_fused(E, &A, &otf, &otf_bp, 1, iters);
        return;
    }
    for (int it = 0; it < iters; ++it) view_update(E, A, otf, otf_bp);
}

void DeconEngine::run_dual(float *E, const float *A, const float *B, const float2 *otf1, const float2 *otf2,
                           const float2 *otf_bp1, const float2 *otf_bp2, int iters) {
    TraceRange trace("mil::rl_dual_view (enqueue)");
    if (plan_) {
        const float *M[2] = {A, B};
        const float2 *o[2] = {otf1, otf2}, *ob[2] = {otf_bp1, otf_bp2};
        run_fused(E, M, o, ob, 2, iters);
        return;
    }
    for (int it = 0; it < iters; ++it) {
        view_update(E, A, otf1, otf_bp1);  // :3636-3646
        view_update(E, B, otf2, otf_bp2);  // :3649-3659
    }
}

}  // namespace mil

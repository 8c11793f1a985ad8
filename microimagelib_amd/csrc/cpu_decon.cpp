// gpuMemMode 0: Richardson-Lucy deconvolution on the host (BASELINE.json configs[0]).
//
// Order of operations = the reference's CPU twin: decon_singleview_OTF0 (src/api_subfunc.cu:3310-3359),
// decon_dualview_OTF0 (:3519-3585), the mode-0 branches of decon_singleview / decon_dualview
// (src/api_decon.cpp:151-197, :430-520) and genOTFcpu (src/api_subfunc.cu:1650-1676), with two deliberate
// differences:
//   * the PSF is placed like the GPU path does (padPSFgpukernel, include/cukernel.cuh:680-697); padPSFcpu strides its
//     source with the wrong extents (src/api_subfunc.cu:1487-1497) and is only right when PSF box == FFT box;
//   * the transform is the bundled one of cpu_fft.h (FFTW is not in this image), planned before the data is filled.
// Element-wise loops are separate passes, one float operation per voxel each, exactly like sumcpu / maxvaluecpu /
// divcpu / multicpu / multicomplexcpu (:93-208) -- this file is compiled with -ffp-contract=off.
// One thread by default like the reference; MIL_CPU_THREADS=n (0 = all) runs every pass on n threads.
#include "cpu_decon.h"

#include <chrono>
#include <cstring>

#include "cpu_fft.h"

namespace mil {
namespace cpu {

namespace {

double seconds_now() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

double sum_f64(const float *a, size_t n, int threads) {  // sumcpu :93-99 (float64 accumulator)
    std::vector<double> part((size_t)std::max(1, threads), 0.0);
    parallel_for((long long)n, threads, [&](long long b, long long e, int t) {
        double s = 0;
        for (long long i = b; i < e; ++i) s += (double)a[i];
        part[(size_t)t] = s;
    });
    double s = 0;
    for (double p : part) s += p;  // fixed order: deterministic for a given thread count
    return s;
}

void clamp_small(float *a, size_t n, int threads) {  // maxvaluecpu :200-208
    parallel_for((long long)n, threads, [&](long long b, long long e, int) {
        for (long long i = b; i < e; ++i) a[i] = (a[i] > kSmallValue) ? a[i] : kSmallValue;
    });
}

void cmul_inplace(float2 *s, const float2 *otf, size_t n, int threads) {  // multicomplexcpu :160-169
    parallel_for((long long)n, threads, [&](long long b, long long e, int) {
        for (long long i = b; i < e; ++i) {
            const float2 a = s[i], o = otf[i];
            s[i] = make_float2(a.x * o.x - a.y * o.y, a.x * o.y + a.y * o.x);
        }
    });
}

void ratio(float *t, const float *a, size_t n, int threads) {  // divcpu :142-146: T = A / T, unguarded
    parallel_for((long long)n, threads, [&](long long b, long long e, int) {
        for (long long i = b; i < e; ++i) t[i] = a[i] / t[i];
    });
}

void multiply(float *e_, const float *t, size_t n, int threads) {  // multicpu :133-137
    parallel_for((long long)n, threads, [&](long long b, long long e, int) {
        for (long long i = b; i < e; ++i) e_[i] = e_[i] * t[i];
    });
}

// centred placement with edge replication (padstackcpu :1514-1556); dst box >= src on every axis
void pad_stack(float *dst, const float *src, Dims box, Dims im, int threads) {
    const long long ox = (box.nx - im.nx) / 2, oy = (box.ny - im.ny) / 2, oz = (box.nz - im.nz) / 2;
    parallel_for(box.nz, threads, [&](long long b, long long e, int) {
        for (long long z = b; z < e; ++z) {
            const long long sz = std::min(std::max(z - oz, 0LL), im.nz - 1);
            for (long long y = 0; y < box.ny; ++y) {
                const long long sy = std::min(std::max(y - oy, 0LL), im.ny - 1);
                const float *row = src + (sz * im.ny + sy) * im.nx;
                float *out = dst + (z * box.ny + y) * box.nx;
                for (long long x = 0; x < box.nx; ++x) out[x] = row[std::min(std::max(x - ox, 0LL), im.nx - 1)];
            }
        }
    });
}

void crop_stack(float *dst, const float *src, Dims im, Dims box, int threads) {  // cropcpu :1568-1586
    const long long ox = (box.nx - im.nx) / 2, oy = (box.ny - im.ny) / 2, oz = (box.nz - im.nz) / 2;
    parallel_for(im.nz, threads, [&](long long b, long long e, int) {
        for (long long z = b; z < e; ++z)
            for (long long y = 0; y < im.ny; ++y)
                std::memcpy(dst + (z * im.ny + y) * im.nx, src + ((z + oz) * box.ny + (y + oy)) * box.nx + ox,
                            (size_t)im.nx * sizeof(float));
    });
}

}  // namespace

// genOTFcpu (src/api_subfunc.cu:1650-1676) with the GPU path's placement: normalise by the float64 sum, centre-crop a
// PSF that is larger than the box on an axis (alignsize3D, offset (F - s) / 2 with C truncation), PSF voxel s/2 -> box
// origin, circular; `flip` reverses the PSF on all three axes first (flipcpu :1467-1477).
void gen_otf(const Rfft3D &fft, float2 *otf, const float *psf, Dims p, bool flip, float *work) {
    const Dims F = fft.box();
    const size_t n_psf = (size_t)p.count(), n_box = (size_t)F.count();
    const double sum = sum_f64(psf, n_psf, 1);
    const float scale = (float)(1.0 / sum);  // (float)(1 / sumValue)  :1659
    std::memset(work, 0, n_box * sizeof(float));
    // Per axis (PSF extent s, box extent S), for index i of the (flipped) PSF:
    //   no axis larger than the box:   destination (i - s/2) mod S                       (padPSF, origin s/2)
    //   some axis larger than the box: the PSF is first centred in the box -- alignsize3D: cropped at offset (s - S)/2
    //                                  where s > S, zero-padded at offset (S - s)/2 where s < S -- then shifted by S/2.
    const bool cropped = p.nx > F.nx || p.ny > F.ny || p.nz > F.nz;
    struct Axis {
        long long first, count, shift;  // PSF indices [first, first + count) survive; destination = (i + shift) mod S
    };
    auto axis = [&](long long s, long long S) {
        if (!cropped) return Axis{0, s, -(s / 2)};
        if (s > S) return Axis{(s - S) / 2, S, -((s - S) / 2) - S / 2};
        return Axis{0, s, (S - s) / 2 - S / 2};
    };
    const Axis ax = axis(p.nx, F.nx), ay = axis(p.ny, F.ny), az = axis(p.nz, F.nz);
    auto wrap = [](long long d, long long S) { return ((d % S) + S) % S; };
    for (long long z = az.first; z < az.first + az.count; ++z) {
        const long long fz = flip ? p.nz - 1 - z : z, dz = wrap(z + az.shift, F.nz);
        for (long long y = ay.first; y < ay.first + ay.count; ++y) {
            const long long fy = flip ? p.ny - 1 - y : y, dy = wrap(y + ay.shift, F.ny);
            for (long long x = ax.first; x < ax.first + ax.count; ++x) {
                const long long fx = flip ? p.nx - 1 - x : x, dx = wrap(x + ax.shift, F.nx);
                work[(dz * F.ny + dy) * F.nx + dx] = psf[(fz * p.ny + fy) * p.nx + fx] * scale;
            }
        }
    }
    fft.forward(work, otf);
}

int decon(const DeconCpuArgs &a) {
    const double t_start = seconds_now();
    const Dims im = a.im, F = a.box;
    const size_t n_im = (size_t)im.count(), n_box = (size_t)F.count();
    const bool padded = F != im;
    const int threads = a.threads > 0 ? a.threads : default_threads();
    Rfft3D fft(F, threads);
    const size_t n_spec = fft.spectrum_count();
    std::vector<float> A(n_box), B(a.dual ? n_box : 0), E(n_box), T(n_box);
    std::vector<float2> S(n_spec), otf[2], otf_bp[2];
    const int views = a.dual ? 2 : 1;
    for (int v = 0; v < views; ++v) {  // src/api_decon.cpp:163-173 (single), :446-470 (dual)
        otf[v].resize(n_spec);
        otf_bp[v].resize(n_spec);
        gen_otf(fft, otf[v].data(), a.psf[v], a.psf_dims, false, T.data());
        if (a.unmatch) gen_otf(fft, otf_bp[v].data(), a.psf_bp[v], a.psf_dims, false, T.data());
        else gen_otf(fft, otf_bp[v].data(), a.psf[v], a.psf_dims, true, T.data());
    }
    float *vols[2] = {A.data(), B.data()};
    for (int v = 0; v < views; ++v) {  // :176-181
        if (padded) pad_stack(vols[v], a.img[v], F, im, threads);
        else std::memcpy(vols[v], a.img[v], n_im * sizeof(float));
        clamp_small(vols[v], n_box, threads);  // decon_singleview_OTF0 :3323
    }
    // initial estimate  :3325-3332 (single), :3540-3553 (dual)
    if (a.const_init) {
        float mean = (float)sum_f64(A.data(), n_box, threads);  // the SUM, as upstream
        if (a.dual) mean = (mean + (float)sum_f64(B.data(), n_box, threads)) / 2;
        for (size_t i = 0; i < n_box; ++i) E[i] = 0.f + mean;  // memset 0 + addvaluecpu
    } else if (a.dual) {
        for (size_t i = 0; i < n_box; ++i) E[i] = (A[i] + B[i]) * 0.5f;  // addcpu + multivaluecpu
    } else {
        std::memcpy(E.data(), A.data(), n_box * sizeof(float));
    }
    const double t_loop = seconds_now();
    for (int it = 0; it < a.iterations; ++it) {
        for (int v = 0; v < views; ++v) {  // :3338-3350
            fft.forward(E.data(), S.data());
            cmul_inplace(S.data(), otf[v].data(), n_spec, threads);
            fft.inverse(S.data(), T.data());
            ratio(T.data(), vols[v], n_box, threads);
            fft.forward(T.data(), S.data());
            cmul_inplace(S.data(), otf_bp[v].data(), n_spec, threads);
            fft.inverse(S.data(), T.data());
            multiply(E.data(), T.data(), n_box, threads);
            clamp_small(E.data(), n_box, threads);
        }
    }
    const double t_done = seconds_now();
    if (a.verbose) printf("...Time cost for decon is %2.3f s\n", t_done - t_loop);
    if (padded) crop_stack(a.out, E.data(), im, F, threads);  // src/api_decon.cpp:186-191
    else std::memcpy(a.out, E.data(), n_im * sizeof(float));
    const double t_end = seconds_now();
    if (a.seconds) {
        a.seconds[0] = t_loop - t_start;  // preparation
        a.seconds[1] = t_done - t_loop;   // RL loop
        a.seconds[2] = t_end - t_start;   // whole call
    }
    return 0;
}

}  // namespace cpu
}  // namespace mil

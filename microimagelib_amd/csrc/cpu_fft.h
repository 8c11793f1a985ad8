// Host (CPU) 3-D real transforms for the gpuMemMode 0 path (reference: FFTW single precision,
// src/api_subfunc.cu:3334-3336 -- FFTW is a third-party library that is absent from this image, so the transform is
// bundled: mixed-radix Stockham passes, vectorised over a tile of 16 neighbouring columns).
// Unnormalised in both directions like fftwf_plan_dft_r2c_3d / c2r_3d.  Spectrum layout = FFTW's:
// [nz][ny][nx/2+1] interleaved complex for a volume stored x-fastest.
#pragma once

#include <functional>
#include <memory>
#include <vector>

#include "mil_common.h"

namespace mil {
namespace cpu {

// Runs fn(begin, end, tid) over [0, n) split into `threads` contiguous chunks; threads <= 1 runs inline.
void parallel_for(long long n, int threads, const std::function<void(long long, long long, int)> &fn);

// Threads for the CPU path: MIL_CPU_THREADS (0 = all hardware threads), default 1 -- the reference's CPU path is
// single-threaded.
int default_threads();

class Fft1D;

class Rfft3D {
public:
    Rfft3D(Dims box, int threads);
    ~Rfft3D();
    Rfft3D(const Rfft3D &) = delete;
    Rfft3D &operator=(const Rfft3D &) = delete;

    size_t spectrum_count() const { return (size_t)(box_.nx / 2 + 1) * box_.ny * box_.nz; }
    Dims box() const { return box_; }
    int threads() const { return threads_; }
    // real [nz][ny][nx] -> spec [nz][ny][nx/2+1]
    void forward(const float *real, float2 *spec) const;
    // spec is destroyed (used as the work area, like FFTW's c2r)
    void inverse(float2 *spec, float *real) const;

private:
    void columns(float2 *spec, bool along_z, bool inverse) const;
    Dims box_;
    int threads_;
    std::unique_ptr<Fft1D> fx_, fy_, fz_;  // fx_: length nx/2 (packed real rows)
    std::vector<float> unt_c_, unt_s_;      // e^{-2 pi i k / nx}, k = 0 .. nx/2
};

}  // namespace cpu
}  // namespace mil

// gpuMemMode 0: host Richardson-Lucy twin (reference decon_*_OTF0, src/api_subfunc.cu:3310-3359, 3519-3585).
#pragma once

#include "mil_common.h"

namespace mil {
namespace cpu {

struct DeconCpuArgs {
    float *out = nullptr;                        // im-sized
    const float *img[2] = {nullptr, nullptr};    // im-sized; [1] only when dual
    const float *psf[2] = {nullptr, nullptr};
    const float *psf_bp[2] = {nullptr, nullptr};  // only read when unmatch
    Dims im{0, 0, 0}, psf_dims{0, 0, 0}, box{0, 0, 0};
    bool dual = false, const_init = false, unmatch = false, verbose = false;
    int iterations = 0;
    int threads = 0;            // 0: MIL_CPU_THREADS or 1
    double *seconds = nullptr;  // optional [3]: preparation, RL loop, whole call
};

int decon(const DeconCpuArgs &args);

}  // namespace cpu
}  // namespace mil

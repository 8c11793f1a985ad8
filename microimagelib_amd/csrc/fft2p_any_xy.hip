// Two-pass path: the xy kernel for every nx = 192 .. 1024 (multiples of 64) and every ny = 128 .. 1024 (see fft2p_any.h).
// xy_kernel<MODE, 0, NH, STREAM> of fft2p_dev.h: the y extent is a run-time argument, one instantiation per NH = nx / 2.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "fft2p_any_xy.hip is written for gfx950 (v_permlane32_swap, 160 KB of LDS per workgroup): build with --offload-arch=gfx950"
#endif
#include "fft2p_any.h"

#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <utility>
#include <vector>

namespace mil {
namespace fx {

namespace {

#include "fft2p_dev.h"

template <int NH>
void launch_xy(int mode, bool stream, const XyArgs &a, unsigned blocks, hipStream_t s) {
    const dim3 grid(blocks), block(XyCfg<NH>::threads);
#define MIL_XY_ANY(STREAM)                                                                                      \
    switch (mode) {                                                                                             \
        case kXRatio: hipLaunchKernelGGL((xy_kernel<kXRatio, 0, NH, STREAM>), grid, block, 0, s, a); break;     \
        case kXUpdate: hipLaunchKernelGGL((xy_kernel<kXUpdate, 0, NH, STREAM>), grid, block, 0, s, a); break;   \
        default: hipLaunchKernelGGL((xy_kernel<kXUpdateLast, 0, NH, STREAM>), grid, block, 0, s, a); break;     \
    }
    if (stream) {
        MIL_XY_ANY(true)
    } else {
        MIL_XY_ANY(false)
    }
#undef MIL_XY_ANY
}

// NH = 96, 128, ... 512
template <int... I>
bool dispatch_xy(int nh, int mode, bool stream, const XyArgs &a, unsigned blocks, hipStream_t s, std::integer_sequence<int, I...>) {
    bool done = false;
    ((nh == 96 + 32 * I ? (launch_xy<96 + 32 * I>(mode, stream, a, blocks, s), done = true) : false), ...);
    return done;
}

}  // namespace

namespace any {

bool xy_covers(const Box &b) {
    return b.nx % 64 == 0 && b.nx >= 192 && b.nx <= 1024 && b.ny % 64 == 0 && b.ny >= 128 && b.ny <= 1024 &&
           b.xs == b.nx / 2 && b.pitch == b.xs + kW;
}

void xy_fill_tables(const float2 *tw_y_dev, int ny) { fill_tw_tables(tw_y_dev, ny, nullptr, 0); }

void xy_pass(const Box &b, int mode, float2 *S, float *E, const float *M, const float2 *tw_x, float eps, bool stream,
             hipStream_t s) {
    if (!xy_covers(b)) throw Error(-1, "two-pass path: xy kernel does not cover this box");
    XyArgs a;
    a.S = S; a.E = E; a.M = M; a.tw_x = tw_x; a.eps = eps;
    a.spec_bytes = (long long)b.pitch * b.ny * b.nz * 8;
    a.real_bytes = (long long)b.nx * b.ny * b.nz * 4;
    a.stamps = nullptr;
    a.t = b.ny / 64;
    if (!dispatch_xy(b.nx / 2, mode, stream, a, (unsigned)(b.nz * a.t), s, std::make_integer_sequence<int, 14>{}))
        throw Error(-1, "two-pass path: no xy kernel for this nx");
    MIL_HIP(hipGetLastError());
}

}  // namespace any
}  // namespace fx
}  // namespace mil

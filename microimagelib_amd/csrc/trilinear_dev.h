// Device-side pieces of the explicit trilinear "texture fetch" shared by affine.hip (one voxel per thread: resample, 2-D
// variants, the general cost kernel) and resample_march.hip (marching kernels with plane / row reuse).  Semantics restated
// from include/cukernel.cuh:501-556 -- see the header of affine.hip.  Everything here is float32 without contraction.
#pragma once

#include "kernels.h"

namespace mil {
namespace k {
namespace tri {

struct AffineArgs {
    float m[12];
    int tx, ty, tz;  // target / output dims
    int sx, sy, sz;  // source dims
    float wq;        // 0: exact float32 weights; 256: weights rounded to 8 fractional bits (texture_weight_bits)
};

__device__ __forceinline__ float coord(const float *r, float ix, float iy, float iz) {
    float t = r[0] * ix + r[1] * iy;  // d_aff[0]*ix + d_aff[1]*iy + d_aff[2]*iz + d_aff[3] + 0.5
    t = t + r[2] * iz;
    t = t + r[3];
    return t + 0.5f;
}

struct Tap {
    int i0, i1;
    float w0, w1;
};

// wq > 0: CUDA-texture emulation.  The CUDA programming guide (texture fetching, linear filtering) stores the weight
// "in 9-bit fixed point format with 8 bits of fractional value (so 1.0 is exactly represented)"; it does not say how it
// rounds -- round to nearest is ASSUMED here -- nor how the filter arithmetic is carried out (ours stays
// w0*q0 + w1*q1 in float32, x then y then z).  A comparison aid for volumes produced by the CUDA build, off by default.
// QUANT = false: the caller knows wq == 0 (the compiler otherwise evaluates the quantised weight -- an IEEE division --
// for every tap and selects: 3 divisions per voxel, a quarter of resample_kernel's instructions).
template <bool QUANT = true>
__device__ __forceinline__ Tap make_tap(float t, int n, float wq) {
    float b = t - 0.5f;
    float f = floorf(b);
    Tap p;
    p.w1 = b - f;
    if (QUANT && wq > 0.0f) p.w1 = floorf(p.w1 * wq + 0.5f) / wq;
    p.w0 = 1.0f - p.w1;
    // Only called for t inside the validity window, so i = floor(t - 0.5) lies in [-1, n-1]: the edge clamp of the
    // reference's texture fetch reduces to one max and one min.
    const int i = (int)f;
    p.i0 = i < 0 ? 0 : i;
    p.i1 = i + 1 > n - 1 ? n - 1 : i + 1;
    return p;
}

// The two x taps of a row are adjacent floats except where the clamp folds them onto one edge
// voxel, so one 8-byte gather at b = clamp(i, 0, sx-2) serves both: half the texture-address work of two
// 4-byte gathers (the cost kernel is bound by gather issue, not by HBM).  4-byte alignment only.
struct __attribute__((packed, aligned(4))) Pair {
    float lo, hi;
};

__device__ __forceinline__ float fetch_linear(const float *__restrict__ src, int sx, int sy, int sz, float tx, float ty,
                                              float tz, float wq) {
    Tap X = make_tap(tx, sx, wq), Y = make_tap(ty, sy, wq), Z = make_tap(tz, sz, wq);
    const size_t row = (size_t)sx, plane = (size_t)sx * sy;
    const float *p00 = src + Z.i0 * plane + Y.i0 * row;
    const float *p01 = src + Z.i0 * plane + Y.i1 * row;
    const float *p10 = src + Z.i1 * plane + Y.i0 * row;
    const float *p11 = src + Z.i1 * plane + Y.i1 * row;
    if (sx >= 2) {
        const int b = X.i0 < sx - 2 ? X.i0 : sx - 2;  // X.i0 >= 0 already
        const bool lo0 = X.i0 == b, lo1 = X.i1 == b;
        const Pair q00 = *reinterpret_cast<const Pair *>(p00 + b), q01 = *reinterpret_cast<const Pair *>(p01 + b);
        const Pair q10 = *reinterpret_cast<const Pair *>(p10 + b), q11 = *reinterpret_cast<const Pair *>(p11 + b);
        float c00 = X.w0 * (lo0 ? q00.lo : q00.hi) + X.w1 * (lo1 ? q00.lo : q00.hi);
        float c01 = X.w0 * (lo0 ? q01.lo : q01.hi) + X.w1 * (lo1 ? q01.lo : q01.hi);
        float c10 = X.w0 * (lo0 ? q10.lo : q10.hi) + X.w1 * (lo1 ? q10.lo : q10.hi);
        float c11 = X.w0 * (lo0 ? q11.lo : q11.hi) + X.w1 * (lo1 ? q11.lo : q11.hi);
        float c0 = Y.w0 * c00 + Y.w1 * c01;
        float c1 = Y.w0 * c10 + Y.w1 * c11;
        return Z.w0 * c0 + Z.w1 * c1;
    }
    float c00 = X.w0 * p00[X.i0] + X.w1 * p00[X.i1];
    float c01 = X.w0 * p01[X.i0] + X.w1 * p01[X.i1];
    float c10 = X.w0 * p10[X.i0] + X.w1 * p10[X.i1];
    float c11 = X.w0 * p11[X.i0] + X.w1 * p11[X.i1];
    float c0 = Y.w0 * c00 + Y.w1 * c01;
    float c1 = Y.w0 * c10 + Y.w1 * c11;
    return Z.w0 * c0 + Z.w1 * c1;
}

// The same fetch for volumes below 2 GiB whose planes hold fewer than 2^24 voxels (every volume of this pipeline): ONE 32-bit
// byte offset per tap row -- two UNSIGNED 24-bit multiplies (v_mul_u32_u24: every operand is a non-negative index or extent
// below 2^24; the signed form sign-extends from bit 23 and read planes of 2^23 .. 2^24 voxels as negative) and adds instead
// of 64-bit multiply-adds per pointer -- through a buffer descriptor.  Same taps, same float32 operations in the same order: the same bits.  (resample_kernel spent 195 vector
// instructions per voxel, most of them address arithmetic: it was bound by them, not by its four gathers.)
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float fetch_linear_small(__amdgpu_buffer_rsrc_t rsrc, int sx, int sy, int sz, float tx, float ty,
                                                    float tz, float wq) {
    const Tap X = make_tap<false>(tx, sx, wq), Y = make_tap<false>(ty, sy, wq), Z = make_tap<false>(tz, sz, wq);  // wq == 0 (host check)
    const int b = X.i0 < sx - 2 ? X.i0 : sx - 2;  // sx >= 2 (host check)
    const bool lo0 = X.i0 == b, lo1 = X.i1 == b;
    const unsigned plane = __umul24((unsigned)sx, (unsigned)sy);
    const unsigned z0 = __umul24((unsigned)Z.i0, plane), z1 = __umul24((unsigned)Z.i1, plane);
    const unsigned y0 = __umul24((unsigned)Y.i0, (unsigned)sx) + (unsigned)b, y1 = __umul24((unsigned)Y.i1, (unsigned)sx) + (unsigned)b;
    auto pair = [&](unsigned o) {
        const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)(o << 2), 0, 0);
        return Pair{__uint_as_float(v.x), __uint_as_float(v.y)};
    };
    const Pair q00 = pair(z0 + y0), q01 = pair(z0 + y1), q10 = pair(z1 + y0), q11 = pair(z1 + y1);
    const float c00 = X.w0 * (lo0 ? q00.lo : q00.hi) + X.w1 * (lo1 ? q00.lo : q00.hi);
    const float c01 = X.w0 * (lo0 ? q01.lo : q01.hi) + X.w1 * (lo1 ? q01.lo : q01.hi);
    const float c10 = X.w0 * (lo0 ? q10.lo : q10.hi) + X.w1 * (lo1 ? q10.lo : q10.hi);
    const float c11 = X.w0 * (lo0 ? q11.lo : q11.hi) + X.w1 * (lo1 ? q11.lo : q11.hi);
    const float c0 = Y.w0 * c00 + Y.w1 * c01;
    const float c1 = Y.w0 * c10 + Y.w1 * c11;
    return Z.w0 * c0 + Z.w1 * c1;
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t small_rsrc(const float *src, int sx, int sy, int sz) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0, sx * sy * sz * 4, 0x00020000);
}


}  // namespace tri
}  // namespace k
}  // namespace mil

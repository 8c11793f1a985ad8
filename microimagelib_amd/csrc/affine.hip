// Hand-written gfx950 kernels for the registration hot path: explicit trilinear
// resampling (gfx950 has no texture unit -- hipcc rejects tex3D for this target)
// and the fused interpolate + correlate + reduce cost kernel that Powell drives.
//
// Semantics restated from the reference CUDA kernels (include/cukernel.cuh:501-556):
//   t = M*(x,y,z,1) + 0.5 evaluated left to right in float32 (no FMA: this file is
//   built with -ffp-contract=off), validity window on t, then a linear "texture"
//   fetch at t with texel centres at +0.5 and neighbour indices clamped to the edge.
// Interpolation weights are exact float32 (CUDA's texture unit quantises them to 8
// fractional bits; see DESIGN.md "Registration numerics").
//
// Both kernels are gather/HBM-bound: per target voxel one coalesced target read and
// eight source gathers whose footprint, for the near-identity matrices registration
// visits, is a sheared box only slightly larger than the target tile -- so the gathers
// hit L1/L2 and HBM traffic stays ~ (target + source) bytes.
#include "kernels.h"

#include <cstdint>
#include <cstdlib>
#include <cstring>

namespace mil {
namespace k {

namespace {

constexpr int kTileX = 64;   // one wave spans 64 consecutive x -> coalesced target reads
constexpr int kTileY = 4;    // 4 waves per workgroup
constexpr int kCorrZ = 16;   // z-steps per thread in the cost kernel
constexpr int kBlockThreads = kTileX * kTileY;

struct AffineArgs {
    float m[12];
    int tx, ty, tz;  // target / output dims
    int sx, sy, sz;  // source dims
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

__device__ __forceinline__ Tap make_tap(float t, int n) {
    float b = t - 0.5f;
    float f = floorf(b);
    Tap p;
    p.w1 = b - f;
    p.w0 = 1.0f - p.w1;
    int i = (int)f;
    int j = i + 1;
    p.i0 = i < 0 ? 0 : (i > n - 1 ? n - 1 : i);
    p.i1 = j < 0 ? 0 : (j > n - 1 ? n - 1 : j);
    return p;
}

// The two x taps of a row are adjacent floats except where the clamp folds them onto one edge
// voxel, so one 8-byte gather at b = clamp(i, 0, sx-2) serves both: half the texture-address work of two
// 4-byte gathers (the cost kernel is bound by gather issue, not by HBM).  4-byte alignment only.
struct __attribute__((packed, aligned(4))) Pair {
    float lo, hi;
};

__device__ __forceinline__ float fetch_linear(const float *__restrict__ src, int sx, int sy, int sz, float tx, float ty,
                                              float tz) {
    Tap X = make_tap(tx, sx), Y = make_tap(ty, sy), Z = make_tap(tz, sz);
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

// affinetransformkernel, reference include/cukernel.cuh:501-524 (window 0 <= t < s2).
__global__ __launch_bounds__(kBlockThreads) void resample_kernel(float *__restrict__ out,
                                                               const float *__restrict__ src, AffineArgs a) {
    const int x = blockIdx.x * kTileX + (threadIdx.x & (kTileX - 1));
    const int y = blockIdx.y * kTileY + (threadIdx.x / kTileX);
    const int z = blockIdx.z;
    if (x >= a.tx || y >= a.ty) return;
    const float ix = (float)x, iy = (float)y, iz = (float)z;
    const float tx = coord(a.m + 0, ix, iy, iz);
    const float ty = coord(a.m + 4, ix, iy, iz);
    const float tz = coord(a.m + 8, ix, iy, iz);
    float v = 0.0f;
    if (tx >= 0 && tx < (float)a.sx && ty >= 0 && ty < (float)a.sy && tz >= 0 && tz < (float)a.sz)
        v = fetch_linear(src, a.sx, a.sy, a.sz, tx, ty, tz);
    out[(size_t)x + (size_t)a.tx * ((size_t)y + (size_t)a.ty * z)] = v;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// corrkernel, reference include/cukernel.cuh:526-556 (strict window 0 < t < s2), fused with
// the two reductions the reference does in separate launches (sumgpu1D, src/api_subfunc.cu:970-971).
// Each workgroup owns a 64 x 4 x kCorrZ target tile; float64 accumulators per lane, DPP/shuffle
// wave reduction, LDS across the 4 waves, one (ss, st) partial per workgroup.
__global__ __launch_bounds__(kBlockThreads) void corr_kernel(const float *__restrict__ tgt,
                                                           const float *__restrict__ src, AffineArgs a,
                                                           double *__restrict__ partials, int nblocks) {
    __shared__ double lds[2 * kTileY];
    const int lane = threadIdx.x & (kTileX - 1), wave = threadIdx.x / kTileX;
    const int x = blockIdx.x * kTileX + lane;
    const int y = blockIdx.y * kTileY + wave;
    const int z0 = blockIdx.z * kCorrZ;
    double ss = 0.0, st = 0.0;
    if (x < a.tx && y < a.ty) {
        const float ix = (float)x, iy = (float)y;
        const int z1 = (z0 + kCorrZ < a.tz) ? z0 + kCorrZ : a.tz;
        const float *tp = tgt + (size_t)x + (size_t)a.tx * ((size_t)y + (size_t)a.ty * z0);
        const size_t tstep = (size_t)a.tx * a.ty;
#pragma unroll 4
        for (int z = z0; z < z1; ++z, tp += tstep) {
            const float iz = (float)z;
            const float tx = coord(a.m + 0, ix, iy, iz);
            const float ty = coord(a.m + 4, ix, iy, iz);
            const float tz = coord(a.m + 8, ix, iy, iz);
            float sv = 0.0f;
            if (tx > 0 && tx < (float)a.sx && ty > 0 && ty < (float)a.sy && tz > 0 && tz < (float)a.sz)
                sv = fetch_linear(src, a.sx, a.sy, a.sz, tx, ty, tz);
            const float tv = *tp;
            ss += (double)sv * (double)sv;
            st += (double)sv * (double)tv;
        }
    }
    ss = wave_sum(ss);
    st = wave_sum(st);
    if (lane == 0) {
        lds[wave] = ss;
        lds[kTileY + wave] = st;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int w = 0; w < kTileY; ++w) {
            a0 += lds[w];
            a1 += lds[kTileY + w];
        }
        const int b = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        partials[b] = a0;
        partials[nblocks + b] = a1;
    }
}

// ---------------------------------------------------------------------------
// LDS-staged cost kernel.  The direct kernel above issues 8 global gathers per voxel and is bound by
// the texture-address path (16 cycles per wave-instruction), not by HBM.  Here a workgroup owns a
// 32 x 16 x 16 target tile, computes the exact source bounding box of the tile, stages that slab in LDS
// with coalesced 16-byte loads, and takes the 8 taps from LDS.
//
// Exactness of the bounding box: coord() is a fixed sequence of float32 multiplies and adds, each
// monotone in its operands, so t(x,y,z) is monotone along every axis and its extremes over the tile are
// attained at the 8 corners -- evaluated with the same expression they bound every voxel's t bit-for-bit;
// floor and the index clamp are monotone too.  A tile whose slab exceeds the LDS budget (large
// rotations / scales) falls back to global gathers for that workgroup.
// Per-voxel arithmetic is identical to corr_kernel, so (ss, st) differ only by float64 summation order.
// ---------------------------------------------------------------------------
constexpr int kLX = 32, kLY = 16, kLZ = 16;
constexpr int kSlabFloats = 20000;  // 80,000 B -> two workgroups per CU

struct SlabBox {
    int x0, y0, z0;  // slab origin in the source (x0 a multiple of 4)
    int bx, by, bz;  // slab extents (bx a multiple of 4)
};

__device__ __forceinline__ void axis_range(const float *r, const float *cx, const float *cy, const float *cz, int n,
                                           int &lo, int &hi) {
    float tmin = coord(r, cx[0], cy[0], cz[0]), tmax = tmin;
#pragma unroll
    for (int c = 1; c < 8; ++c) {
        const float t = coord(r, cx[c & 1], cy[(c >> 1) & 1], cz[(c >> 2) & 1]);
        tmin = fminf(tmin, t);
        tmax = fmaxf(tmax, t);
    }
    // taps of a voxel inside the window: i = floor(t - 0.5), i + 1, both clamped to [0, n-1]
    const float fn = (float)n;
    const float a = fminf(fmaxf(tmin, -2.0f), fn + 2.0f), b = fminf(fmaxf(tmax, -2.0f), fn + 2.0f);
    int i0 = (int)floorf(a - 0.5f), i1 = (int)floorf(b - 0.5f) + 1;
    lo = i0 < 0 ? 0 : (i0 > n - 1 ? n - 1 : i0);
    hi = i1 < 0 ? 0 : (i1 > n - 1 ? n - 1 : i1);
}

__device__ __forceinline__ float fetch_linear_slab(const float *__restrict__ slab, const SlabBox &bb, int sx, int sy,
                                                   int sz, float tx, float ty, float tz) {
    Tap X = make_tap(tx, sx), Y = make_tap(ty, sy), Z = make_tap(tz, sz);
    const int row = bb.bx, plane = bb.bx * bb.by;
    const float *p00 = slab + (Z.i0 - bb.z0) * plane + (Y.i0 - bb.y0) * row - bb.x0;
    const float *p01 = slab + (Z.i0 - bb.z0) * plane + (Y.i1 - bb.y0) * row - bb.x0;
    const float *p10 = slab + (Z.i1 - bb.z0) * plane + (Y.i0 - bb.y0) * row - bb.x0;
    const float *p11 = slab + (Z.i1 - bb.z0) * plane + (Y.i1 - bb.y0) * row - bb.x0;
    float c00 = X.w0 * p00[X.i0] + X.w1 * p00[X.i1];
    float c01 = X.w0 * p01[X.i0] + X.w1 * p01[X.i1];
    float c10 = X.w0 * p10[X.i0] + X.w1 * p10[X.i1];
    float c11 = X.w0 * p11[X.i0] + X.w1 * p11[X.i1];
    float c0 = Y.w0 * c00 + Y.w1 * c01;
    float c1 = Y.w0 * c10 + Y.w1 * c11;
    return Z.w0 * c0 + Z.w1 * c1;
}

constexpr int kSlabThreads = 512;  // 32 x-lanes x 16 y-rows; every thread walks the tile's 16 z

__global__ __launch_bounds__(kSlabThreads) void corr_slab_kernel(const float *__restrict__ tgt,
                                                               const float *__restrict__ src, AffineArgs a,
                                                               double *__restrict__ partials, int nblocks) {
    __shared__ __attribute__((aligned(16))) float slab[kSlabFloats];
    __shared__ double red[2 * (kSlabThreads / 64)];
    const int X0 = blockIdx.x * kLX, Y0 = blockIdx.y * kLY, Z0 = blockIdx.z * kLZ;
    const int X1 = min(X0 + kLX, a.tx) - 1, Y1 = min(Y0 + kLY, a.ty) - 1, Z1 = min(Z0 + kLZ, a.tz) - 1;
    // ---- source bounding box of this tile (wave-uniform)
    const float cx[2] = {(float)X0, (float)X1}, cy[2] = {(float)Y0, (float)Y1}, cz[2] = {(float)Z0, (float)Z1};
    int xl, xh, yl, yh, zl, zh;
    axis_range(a.m + 0, cx, cy, cz, a.sx, xl, xh);
    axis_range(a.m + 4, cx, cy, cz, a.sy, yl, yh);
    axis_range(a.m + 8, cx, cy, cz, a.sz, zl, zh);
    SlabBox bb;
    bb.x0 = xl & ~3; bb.y0 = yl; bb.z0 = zl;
    bb.bx = ((xh - bb.x0 + 1) + 3) & ~3; bb.by = yh - yl + 1; bb.bz = zh - zl + 1;
    const long long need = (long long)bb.bx * bb.by * bb.bz;
    const bool staged = need <= kSlabFloats;
    if (staged) {
        // float4 pieces of the slab, 8 loads in flight per lane before the first LDS write
        const int q4 = bb.bx >> 2, rows = bb.by * bb.bz, total = q4 * rows;
        const float rq4 = 1.0f / (float)q4, rby = 1.0f / (float)bb.by;
        constexpr int kB = 8;
        for (int it0 = 0; it0 < total; it0 += kSlabThreads * kB) {
            float4 v[kB];
            int dst[kB];
#pragma unroll
            for (int j = 0; j < kB; ++j) {
                const int idx = it0 + j * kSlabThreads + (int)threadIdx.x;
                dst[j] = -1;
                if (idx < total) {
                    // exact small-integer division through float (all operands < 2^15)
                    int r = (int)(((float)idx + 0.5f) * rq4);
                    const int q = idx - r * q4;
                    int czz = (int)(((float)r + 0.5f) * rby);
                    const int cyy = r - czz * bb.by;
                    v[j] = *reinterpret_cast<const float4 *>(
                        src + ((size_t)(bb.z0 + czz) * a.sy + (size_t)(bb.y0 + cyy)) * a.sx + bb.x0 + 4 * q);
                    dst[j] = r * bb.bx + 4 * q;
                }
            }
#pragma unroll
            for (int j = 0; j < kB; ++j)
                if (dst[j] >= 0) *reinterpret_cast<float4 *>(slab + dst[j]) = v[j];
        }
    }
    __syncthreads();
    const int lx = threadIdx.x & (kLX - 1), ly = threadIdx.x >> 5;
    const int x = X0 + lx, y = Y0 + ly;
    double ss = 0.0, st = 0.0;
    if (x < a.tx && y < a.ty) {
        const float ix = (float)x, iy = (float)y;
        const size_t tstep = (size_t)a.tx * a.ty;
        const float *tp = tgt + (size_t)x + (size_t)a.tx * ((size_t)y + (size_t)a.ty * Z0);
#pragma unroll 4
        for (int z = Z0; z <= Z1; ++z, tp += tstep) {
            const float iz = (float)z;
            const float tx = coord(a.m + 0, ix, iy, iz);
            const float ty = coord(a.m + 4, ix, iy, iz);
            const float tz = coord(a.m + 8, ix, iy, iz);
            float sv = 0.0f;
            if (tx > 0 && tx < (float)a.sx && ty > 0 && ty < (float)a.sy && tz > 0 && tz < (float)a.sz)
                sv = staged ? fetch_linear_slab(slab, bb, a.sx, a.sy, a.sz, tx, ty, tz)
                            : fetch_linear(src, a.sx, a.sy, a.sz, tx, ty, tz);
            const float tv = *tp;
            ss += (double)sv * (double)sv;
            st += (double)sv * (double)tv;
        }
    }
    ss = wave_sum(ss);
    st = wave_sum(st);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int NW = kSlabThreads / 64;
    if (lane == 0) {
        red[wave] = ss;
        red[NW + wave] = st;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            a0 += red[w];
            a1 += red[NW + w];
        }
        const int b = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        partials[b] = a0;
        partials[nblocks + b] = a1;
    }
}

// ---------------------------------------------------------------------------
// z-marching cost kernel: LDS ring of source planes.
//
// A workgroup owns a 32 x 8 column of the target and marches through kMarchZ target planes.  The source
// planes its current target plane needs (a window of 3-5 planes for near-axis-aligned matrices) live in a
// power-of-two ring of LDS slots, slot = plane & (RING-1), each holding the x/y bounding box of the whole
// column.  While step z is computed from LDS, the plane(s) step z+1 adds are already in flight from HBM
// (one float4 per lane), and are dropped into their ring slots before the single barrier of the step.
// Gathers thus cost LDS cycles (32 lanes/clk) instead of texture-address cycles (4 lanes/clk), staging is
// overlapped, and LDS use is small enough for 6 workgroups per CU.  Bounding boxes are exact (monotone-corner
// argument above); a column whose window does not fit the ring falls back to global pair gathers.
// ---------------------------------------------------------------------------
constexpr int kMX = 32, kMY = 16, kMarchZ = 64;  // target column per workgroup: 32 x 16 x 64
constexpr int kMarchThreads = 256;                 // 32 x-lanes x 8 y-rows, two y sweeps per plane
constexpr int kRingFloats = 7680;                  // 30 KB -> five workgroups per CU

struct RingBox {
    int x0, y0, bx, by, plane, n;  // plane = bx*by floats, n = ring slots
};

// slot of source plane p, given that plane `lo` sits in slot `slot_lo` and p - lo < n
__device__ __forceinline__ int ring_slot(int p, int lo, int slot_lo, int n) {
    int s = slot_lo + (p - lo);
    return s >= n ? s - n : s;
}

__device__ __forceinline__ float fetch_linear_ring(const float *__restrict__ ring, const RingBox &rb, int wlo, int slot_lo,
                                                   int sx, int sy, int sz, float tx, float ty, float tz) {
    Tap X = make_tap(tx, sx), Y = make_tap(ty, sy), Z = make_tap(tz, sz);
    const float *z0p = ring + ring_slot(Z.i0, wlo, slot_lo, rb.n) * rb.plane - rb.x0;
    const float *z1p = ring + ring_slot(Z.i1, wlo, slot_lo, rb.n) * rb.plane - rb.x0;
    const int r0 = (Y.i0 - rb.y0) * rb.bx, r1 = (Y.i1 - rb.y0) * rb.bx;
    float c00 = X.w0 * z0p[r0 + X.i0] + X.w1 * z0p[r0 + X.i1];
    float c01 = X.w0 * z0p[r1 + X.i0] + X.w1 * z0p[r1 + X.i1];
    float c10 = X.w0 * z1p[r0 + X.i0] + X.w1 * z1p[r0 + X.i1];
    float c11 = X.w0 * z1p[r1 + X.i0] + X.w1 * z1p[r1 + X.i1];
    float c0 = Y.w0 * c00 + Y.w1 * c01;
    float c1 = Y.w0 * c10 + Y.w1 * c11;
    return Z.w0 * c0 + Z.w1 * c1;
}

// clamped tap range [lo, hi] of the z axis for the x/y tile at target plane fz
__device__ __forceinline__ void z_window(const AffineArgs &a, const float *cx, const float *cy, float fz, int &lo, int &hi) {
    float tmin = coord(a.m + 8, cx[0], cy[0], fz), tmax = tmin;
#pragma unroll
    for (int c = 1; c < 4; ++c) {
        const float t = coord(a.m + 8, cx[c & 1], cy[c >> 1], fz);
        tmin = fminf(tmin, t);
        tmax = fmaxf(tmax, t);
    }
    const float fn = (float)a.sz;
    const float lo_t = fminf(fmaxf(tmin, -2.0f), fn + 2.0f), hi_t = fminf(fmaxf(tmax, -2.0f), fn + 2.0f);
    const int i0 = (int)floorf(lo_t - 0.5f), i1 = (int)floorf(hi_t - 0.5f) + 1;
    lo = i0 < 0 ? 0 : (i0 > a.sz - 1 ? a.sz - 1 : i0);
    hi = i1 < 0 ? 0 : (i1 > a.sz - 1 ? a.sz - 1 : i1);
}

__global__ __launch_bounds__(kMarchThreads) void corr_march_kernel(const float *__restrict__ tgt,
                                                                 const float *__restrict__ src, AffineArgs a,
                                                                 double *__restrict__ partials, int nblocks) {
    __shared__ __attribute__((aligned(16))) float ring[kRingFloats];
    __shared__ short win_lo[kMarchZ], win_hi[kMarchZ];
    __shared__ double red[2 * (kMarchThreads / 64)];
    const int X0 = blockIdx.x * kMX, Y0 = blockIdx.y * kMY, Z0 = blockIdx.z * kMarchZ;
    const int X1 = min(X0 + kMX, a.tx) - 1, Y1 = min(Y0 + kMY, a.ty) - 1, Z1 = min(Z0 + kMarchZ, a.tz) - 1;
    const int nzs = Z1 - Z0 + 1;
    const float cx[2] = {(float)X0, (float)X1}, cy[2] = {(float)Y0, (float)Y1}, cz[2] = {(float)Z0, (float)Z1};
    // ---- x / y bounding box of the whole column, ring geometry (wave-uniform)
    int xl, xh, yl, yh;
    axis_range(a.m + 0, cx, cy, cz, a.sx, xl, xh);
    axis_range(a.m + 4, cx, cy, cz, a.sy, yl, yh);
    RingBox rb;
    rb.x0 = xl & ~3; rb.y0 = yl;
    rb.bx = ((xh - rb.x0 + 1) + 3) & ~3; rb.by = yh - yl + 1;
    rb.plane = rb.bx * rb.by;
    rb.n = kRingFloats / rb.plane;
    const int q4 = rb.bx >> 2, plane4 = q4 * rb.by;
    const bool up = a.m[10] >= 0.0f;

    // ---- per-plane windows, one lane each; the column is staged only if every step's window, together
    // with the planes the next step adds, fits the ring and one float4 per lane covers a step's new planes
    bool fits = rb.n >= 2 && a.sz <= 32767 && plane4 <= kMarchThreads;
    if ((int)threadIdx.x < nzs) {
        int lo, hi;
        z_window(a, cx, cy, (float)(Z0 + (int)threadIdx.x), lo, hi);
        win_lo[threadIdx.x] = (short)lo;
        win_hi[threadIdx.x] = (short)hi;
    }
    __syncthreads();
    if ((int)threadIdx.x < nzs) {
        const int i = threadIdx.x, j = min(i + 1, nzs - 1);
        const int span = max(win_hi[i], win_hi[j]) - min(win_lo[i], win_lo[j]) + 1;
        const int added = up ? (int)win_hi[j] - max((int)win_hi[i] + 1, (int)win_lo[j]) + 1
                             : min((int)win_lo[i] - 1, (int)win_hi[j]) - (int)win_lo[j] + 1;
        if (span > rb.n || added * plane4 > kMarchThreads) fits = false;
    }
    const bool staged = !__syncthreads_or(!fits);

    // this lane's float4 of a staging sweep: plane offset, LDS offset, global offset (fixed per column)
    const int s_pl = (int)threadIdx.x / plane4, s_r = (int)threadIdx.x - s_pl * plane4;
    const int s_yy = s_r / q4, s_q = s_r - s_yy * q4;
    const int s_lds = s_yy * rb.bx + 4 * s_q;
    const float *s_src = src + (size_t)(rb.y0 + s_yy) * a.sx + rb.x0 + 4 * s_q;
    const size_t src_plane = (size_t)a.sx * a.sy;

    int wlo = win_lo[0], whi = win_hi[0], slot_lo = 0;  // plane wlo sits in slot slot_lo
    if (staged) {  // prologue: the first window
        const int per = kMarchThreads / plane4;  // whole planes per sweep (>= 1)
        for (int p = wlo; p <= whi; p += per)
            if (s_pl < min(per, whi - p + 1)) {
                const int zp = p + s_pl;
                *reinterpret_cast<float4 *>(ring + ring_slot(zp, wlo, 0, rb.n) * rb.plane + s_lds) =
                    *reinterpret_cast<const float4 *>(s_src + (size_t)zp * src_plane);
            }
    }
    __syncthreads();

    // Everything step k reads from global memory -- this lane's piece of the planes first needed at step k
    // ("A(k)") and its two target voxels -- is requested kDepth steps ahead into a register FIFO.  (vmcnt
    // retires in issue order: a load consumed in the step that issued it would drain the whole queue.)
    // A(k) is written to its ring slots at the end of step k-1, just before that step's barrier.
    constexpr int kDepth = 8;
    float4 fifo_v[kDepth];
    int fifo_p[kDepth];
    float fifo_t0[kDepth], fifo_t1[kDepth];
    const int lx = threadIdx.x & (kMX - 1), ly = threadIdx.x >> 5;
    const int x = X0 + lx, y0 = Y0 + ly, y1 = Y0 + ly + 8;
    const bool act0 = x < a.tx && y0 < a.ty, act1 = x < a.tx && y1 < a.ty;
    const size_t tstep = (size_t)a.tx * a.ty;
    const float *tp0 = tgt + (size_t)(act0 ? x : 0) + (size_t)a.tx * ((size_t)(act0 ? y0 : 0) + (size_t)a.ty * Z0);
    const float *tp1 = tgt + (size_t)(act1 ? x : 0) + (size_t)a.tx * ((size_t)(act1 ? y1 : 0) + (size_t)a.ty * Z0);
    auto request = [&](int k, float4 &v, int &zp_out, float &t0, float &t1) {
        zp_out = -1;
        if (k < nzs) {
            if (act0) t0 = tp0[(size_t)k * tstep];
            if (act1) t1 = tp1[(size_t)k * tstep];
            if (staged && k > 0) {
                const int plo = win_lo[k - 1], phi = win_hi[k - 1], nlo = win_lo[k], nhi = win_hi[k];
                const int add_lo = up ? max(phi + 1, nlo) : nlo;
                const int count = (up ? nhi : min(plo - 1, nhi)) - add_lo + 1;
                if (s_pl < count) {
                    zp_out = add_lo + s_pl;
                    v = *reinterpret_cast<const float4 *>(s_src + (size_t)zp_out * src_plane);
                }
            }
        }
    };
#pragma unroll
    for (int j = 0; j < kDepth; ++j) {
        fifo_v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        fifo_t0[j] = fifo_t1[j] = 0.f;
        request(j, fifo_v[j], fifo_p[j], fifo_t0[j], fifo_t1[j]);  // steps 0 .. kDepth-1
    }

    const float ix = (float)x;
    double ss = 0.0, st = 0.0;
    for (int base = 0; base < nzs; base += kDepth) {
#pragma unroll
        for (int j = 0; j < kDepth; ++j) {
            const int zi = base + j;
            if (zi >= nzs) break;
            const float iz = (float)(Z0 + zi);
            const float tv0 = fifo_t0[j], tv1 = fifo_t1[j];  // requested kDepth steps ago
            request(zi + kDepth, fifo_v[j], fifo_p[j], fifo_t0[j], fifo_t1[j]);  // slot j's planes are in the ring already
#pragma unroll
            for (int sweep = 0; sweep < 2; ++sweep) {
                if (sweep == 0 ? act0 : act1) {
                    const float iy = (float)(sweep == 0 ? y0 : y1);
                    const float tx = coord(a.m + 0, ix, iy, iz);
                    const float ty = coord(a.m + 4, ix, iy, iz);
                    const float tz = coord(a.m + 8, ix, iy, iz);
                    float sv = 0.0f;
                    if (tx > 0 && tx < (float)a.sx && ty > 0 && ty < (float)a.sy && tz > 0 && tz < (float)a.sz)
                        sv = staged ? fetch_linear_ring(ring, rb, wlo, slot_lo, a.sx, a.sy, a.sz, tx, ty, tz)
                                    : fetch_linear(src, a.sx, a.sy, a.sz, tx, ty, tz);
                    const float tv = sweep == 0 ? tv0 : tv1;
                    ss += (double)sv * (double)sv;
                    st += (double)sv * (double)tv;
                }
            }
            if (staged && zi + 1 < nzs) {
                // commit A(zi+1); offsets from wlo stay within (-n, n)
                const int jn = (j + 1) % kDepth;  // static after unrolling
                if (fifo_p[jn] >= 0) {
                    int sl = slot_lo + (fifo_p[jn] - wlo);
                    sl = sl >= rb.n ? sl - rb.n : (sl < 0 ? sl + rb.n : sl);
                    *reinterpret_cast<float4 *>(ring + sl * rb.plane + s_lds) = fifo_v[jn];
                }
                __syncthreads();
                const int nlo = win_lo[zi + 1];
                int sl = slot_lo + (nlo - wlo);
                slot_lo = sl >= rb.n ? sl - rb.n : (sl < 0 ? sl + rb.n : sl);
                wlo = nlo;
                whi = win_hi[zi + 1];
            }
        }
    }
    (void)whi;
    (void)tstep;
    ss = wave_sum(ss);
    st = wave_sum(st);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int NW = kMarchThreads / 64;
    if (lane == 0) {
        red[wave] = ss;
        red[NW + wave] = st;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            a0 += red[w];
            a1 += red[NW + w];
        }
        const int b = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        partials[b] = a0;
        partials[nblocks + b] = a1;
    }
}

constexpr int kFinalThreads = 1024;
__global__ __launch_bounds__(kFinalThreads) void corr_final_kernel(const double *__restrict__ partials, int count,
                                                                 double *__restrict__ out) {
    __shared__ double lds[2 * (kFinalThreads / 64)];
    double a0 = 0.0, a1 = 0.0;
    for (int i = threadIdx.x; i < count; i += kFinalThreads) {
        a0 += partials[i];
        a1 += partials[count + i];
    }
    a0 = wave_sum(a0);
    a1 = wave_sum(a1);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        lds[wave] = a0;
        lds[kFinalThreads / 64 + wave] = a1;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double r0 = 0.0, r1 = 0.0;
        for (int w = 0; w < kFinalThreads / 64; ++w) {
            r0 += lds[w];
            r1 += lds[kFinalThreads / 64 + w];
        }
        out[0] = r0;
        out[1] = r1;
    }
}

AffineArgs make_args(Dims t, Dims s, const Affine12 &m) {
    if (t.nx <= 0 || t.ny <= 0 || t.nz <= 0 || s.nx <= 0 || s.ny <= 0 || s.nz <= 0)
        throw Error(-1, "affine kernel: empty volume");
    if (t.nx > (1 << 24) || t.ny > (1 << 24) || t.nz > (1 << 24) || s.nx > (1 << 24) || s.ny > (1 << 24) ||
        s.nz > (1 << 24))
        throw Error(-1, "affine kernel: extent exceeds float32-exact index range");
    AffineArgs a;
    for (int i = 0; i < 12; ++i) a.m[i] = m.m[i];
    a.tx = (int)t.nx; a.ty = (int)t.ny; a.tz = (int)t.nz;
    a.sx = (int)s.nx; a.sy = (int)s.ny; a.sz = (int)s.nz;
    return a;
}

}  // namespace

void affine_resample(float *out, const float *src, Dims out_dims, Dims src_dims, const Affine12 &m, hipStream_t s) {
    AffineArgs a = make_args(out_dims, src_dims, m);
    dim3 grid((a.tx + kTileX - 1) / kTileX, (a.ty + kTileY - 1) / kTileY, a.tz);
    if (grid.y > 65535 || grid.z > 65535) throw Error(-1, "affine_resample: volume exceeds launch limits");
    hipLaunchKernelGGL(resample_kernel, grid, dim3(kBlockThreads), 0, s, out, src, a);
    MIL_HIP(hipGetLastError());
}

CorrWorkspace::CorrWorkspace() { MIL_HIP(hipHostMalloc((void **)&host, 2 * sizeof(double), hipHostMallocDefault)); }
CorrWorkspace::~CorrWorkspace() {
    if (host) (void)hipHostFree(host);
}

void corr_sums_async(const float *target, const float *src, Dims t_dims, Dims s_dims, const Affine12 &m,
                     CorrWorkspace &ws, hipStream_t s) {
    AffineArgs a = make_args(t_dims, s_dims, m);
    // LDS staging uses aligned float4 rows of the source; anything else takes the direct-gather kernel.
    // MIL_CORR_MODE = march (default) | direct | slab selects the kernel (A/B experiments).
    static const int mode = [] {
        const char *e = getenv("MIL_CORR_MODE");
        if (e && !strcmp(e, "direct")) return 0;
        if (e && !strcmp(e, "slab")) return 1;
        return 2;
    }();
    const bool lds_ok = (a.sx % 4 == 0) && ((reinterpret_cast<uintptr_t>(src) & 15u) == 0);
    const int kind = lds_ok ? mode : 0;
    dim3 grid;
    if (kind == 2) grid = dim3((a.tx + kMX - 1) / kMX, (a.ty + kMY - 1) / kMY, (a.tz + kMarchZ - 1) / kMarchZ);
    else if (kind == 1) grid = dim3((a.tx + kLX - 1) / kLX, (a.ty + kLY - 1) / kLY, (a.tz + kLZ - 1) / kLZ);
    else grid = dim3((a.tx + kTileX - 1) / kTileX, (a.ty + kTileY - 1) / kTileY, (a.tz + kCorrZ - 1) / kCorrZ);
    if (grid.y > 65535 || grid.z > 65535) throw Error(-1, "corr_sums: volume exceeds launch limits");
    const size_t nblocks = (size_t)grid.x * grid.y * grid.z;
    if (nblocks > (size_t)(1 << 30)) throw Error(-1, "corr_sums: too many workgroups");
    if (ws.partials.size() < 2 * nblocks + 2) ws.partials.alloc(2 * nblocks + 2);
    if (kind == 2)
        hipLaunchKernelGGL(corr_march_kernel, grid, dim3(kMarchThreads), 0, s, target, src, a, ws.partials.get(),
                           (int)nblocks);
    else if (kind == 1)
        hipLaunchKernelGGL(corr_slab_kernel, grid, dim3(kSlabThreads), 0, s, target, src, a, ws.partials.get(),
                           (int)nblocks);
    else
        hipLaunchKernelGGL(corr_kernel, grid, dim3(kBlockThreads), 0, s, target, src, a, ws.partials.get(),
                           (int)nblocks);
    MIL_HIP(hipGetLastError());
    // the final pass writes (ss, st) straight into pinned host memory: no copy node, one stream sync per evaluation
    hipLaunchKernelGGL(corr_final_kernel, dim3(1), dim3(kFinalThreads), 0, s, ws.partials.get(), (int)nblocks, ws.host);
    MIL_HIP(hipGetLastError());
}

void corr_sums_sync(const float *target, const float *src, Dims t_dims, Dims s_dims, const Affine12 &m,
                    CorrWorkspace &ws, hipStream_t s, double *ss, double *st) {
    corr_sums_async(target, src, t_dims, s_dims, m, ws, s);
    MIL_HIP(hipStreamSynchronize(s));
    *ss = ws.host[0];
    *st = ws.host[1];
}

}  // namespace k
}  // namespace mil

// Registration cost kernel on a border-padded source: the hottest kernel of a time-lapse batch (a 12-DOF Powell
// cascade evaluates it ~2 300 times per volume pair).
//
// The 3-D registration keeps its mean-subtracted source with a one-voxel REPLICATED border on every side,
// [sz+2][sy+2][sx+2].  The border is the texture unit's clamp addressing (reference include/cukernel.cuh:526-556 reads
// a clamped, linearly filtered 3-D texture) materialised in memory: a tap at index -1 or n reads the replicated edge
// voxel.  Inside the validity window floor(t - 0.5) lies in [-1, n-1] and indexes the padded volume directly, so
//   * there are no index clamps and no edge selects;
//   * the four 8-byte gathers of a voxel share ONE 32-bit byte offset: they go through a buffer descriptor whose
//     scalar offset carries the row / plane pitch;
//   * that offset is two 24-bit multiplies and two adds instead of 64-bit multiply-adds.
// The float32 operations and their order are those of fetch_linear (affine.hip), so the two sums are bit-identical to
// corr_kernel's on the unpadded volume.  The kernel is VALU-issue bound, so the arithmetic is written in explicit
// 2-wide vectors (v_pk_mul_f32 / v_pk_add_f32) and this file is compiled with the SLP vectoriser off -- left on, it
// turns the six window compares into a bit-mask reduction that costs more than it saves.
#include "corr_finish.h"
#include "kernels.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <type_traits>

namespace mil {
namespace k {

namespace {

constexpr int kTileX = 64;  // one wave = 64 consecutive x: coalesced target reads
constexpr int kTileY = 4;   // waves per workgroup
constexpr int kStepsZ = 16; // z steps per lane (8: 0.198 / 0.201 ms at the identity / 3 deg + 2 %, 32: 0.145 / 0.179, 16: 0.146 / 0.177 --
                            // profiles/r03/reg_cost_variants_z_steps.txt)
constexpr int kThreads = kTileX * kTileY;

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

struct PaddedArgs {
    float m[12];
    int tx, ty, tz;       // target extents
    float sxf, syf, szf;  // source extents: the validity window 0 < t < s
    int row4;             // padded row pitch in bytes, 4 * (sx + 2)
    int plane4;           // padded plane pitch in bytes
    int origin4;          // byte offset of source voxel (0, 0, 0) inside the padded volume
    int origin4m;         // the same with the constants of the two-voxel kernels' bit-pattern index arithmetic folded in
    unsigned bytes;       // size of the padded volume
    int gx, gy;           // tiles along x and y of the launched kernel (grid = gx * gy * gz workgroups, one-dimensional)
};

// Workgroup -> tile: one-dimensional grid, x fastest.  An XCD-aware order (each of the 8 XCDs, with its private L2, takes a
// contiguous eighth of the tiles instead of every 8th tile) was built and measured: it removes the duplicate fetches of the
// 128-byte lines on the rim of each XCD's column stripe -- FETCH_SIZE 1.26 x -> 1.04 x the algorithmic 2R
// (profiles/r02/reg_cost_experiment_corr_k0.json vs reg_cost_kernel.json) -- and changes nothing at the identity
// (0.155 vs 0.156 ms) while rotations lose 3-5 % (10 deg: 0.245 vs 0.233 ms): these kernels are bound by the
// address path and VALU issue, not by HBM, so the order stays plain.
__device__ __forceinline__ void corr_tile_of_block(int gx, int gy, int &bx, int &by, int &bz) {
    const unsigned tile = blockIdx.x;
    bx = (int)(tile % (unsigned)gx);
    const unsigned rest = tile / (unsigned)gx;
    by = (int)(rest % (unsigned)gy);
    bz = (int)(rest / (unsigned)gy);
}

// The target is read ONCE per evaluation, the source 2-4 times per voxel (and again by the next of the ~2 000 evaluations of
// a registration): the target's loads carry the non-temporal hint, so that the 256 MB Infinity Cache is left to the source
// (271 MB padded at 512 x 512 x 256) instead of being swept by 540 MB per evaluation: -8 .. -10 % per evaluation for every
// kernel and matrix class (profiles/r05/reg_cost_target_nontemporal.txt; the hint on the GATHERS instead: +20 .. +50 %).
__device__ __forceinline__ float target_load(const char *base, unsigned offset) {
    return __builtin_nontemporal_load(reinterpret_cast<const float *>(base + offset));
}

// (lo, hi) = two adjacent source voxels; address = base + voffset (per lane) + soffset (uniform)
__device__ __forceinline__ f32x2 gather_pair(__amdgpu_buffer_rsrc_t rsrc, unsigned voffset, unsigned soffset) {
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)voffset, (int)soffset, 0);
    return __builtin_bit_cast(f32x2, v);
}

__global__ __launch_bounds__(kThreads) void corr_padded_kernel(const float *__restrict__ tgt,
                                                             const float *__restrict__ srcp, PaddedArgs a,
                                                             CorrTail tail) {
    __shared__ double lds[2 * kTileY];
    __shared__ int last;
    const int lane = threadIdx.x & (kTileX - 1), wave = threadIdx.x / kTileX;
    int bx, by, bz;
    corr_tile_of_block(a.gx, a.gy, bx, by, bz);
    const int x = bx * kTileX + lane;
    const int y = by * kTileY + wave;
    const int z0 = bz * kStepsZ;
    double ss = 0.0, st = 0.0;
    if (x < a.tx && y < a.ty) {
        const float ix = (float)x, iy = (float)y;
        const int z1 = (z0 + kStepsZ < a.tz) ? z0 + kStepsZ : a.tz;
        // t = ((m0*x + m1*y) + m2*z) + m3, then + 0.5: the reference's left-to-right float32 evaluation; the x / y
        // part does not change along z
        const f32x2 mxy0 = {a.m[0], a.m[4]}, mxy1 = {a.m[1], a.m[5]}, mxy2 = {a.m[2], a.m[6]}, mxy3 = {a.m[3], a.m[7]};
        const f32x2 hxy = mxy0 * ix + mxy1 * iy;
        const float hz = a.m[8] * ix + a.m[9] * iy;
        const f32x2 half2 = {0.5f, 0.5f};
        // gfx950 raw buffer: dword 3 = 0x00020000 (32-bit data format), records = bytes
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(srcp), 0, (int)a.bytes, 0x00020000);
        const unsigned s01 = (unsigned)a.row4, s10 = (unsigned)a.plane4, s11 = (unsigned)(a.plane4 + a.row4);
        const char *tbase = reinterpret_cast<const char *>(tgt);
        unsigned toff = 4u * ((unsigned)x + (unsigned)a.tx * ((unsigned)y + (unsigned)a.ty * (unsigned)z0));
        const unsigned tstep = 4u * (unsigned)a.tx * (unsigned)a.ty;
        // One z step of one lane: weights, the byte offset of its lower-plane row pair, window flag, target value.
        struct Step {
            f32x2 wx, wy, wz;  // (1 - w, w) per axis
            unsigned off;
            bool in;
            float tv;
        };
        auto prepare = [&](int z, unsigned toff) {
            Step p;
            const float iz = (float)z;
            const f32x2 txy = ((hxy + mxy2 * iz) + mxy3) + half2;
            const float tz = ((hz + a.m[10] * iz) + a.m[11]) + 0.5f;
            const bool gx = txy.x > 0, lx = txy.x < a.sxf, gy = txy.y > 0, ly = txy.y < a.syf, gz = tz > 0, lz = tz < a.szf;
            p.in = gx && lx && gy && ly && gz && lz;
            const f32x2 bxy = txy - half2;
            const float bz = tz - 0.5f;
            const float fx = floorf(bxy.x), fy = floorf(bxy.y), fz = floorf(bz);
            p.wx.y = bxy.x - fx;
            p.wx.x = 1.0f - p.wx.y;
            p.wy.y = bxy.y - fy;
            p.wy.x = 1.0f - p.wy.y;
            p.wz.y = bz - fz;
            p.wz.x = 1.0f - p.wz.y;
            // in the window every index lies in [-1, n-1]; outside, the offset is replaced by a valid one and the
            // value discarded, so no lane ever leaves the padded volume
            const unsigned off = (unsigned)(__mul24((int)fz, a.plane4) + __mul24((int)fy, a.row4) + a.origin4) +
                                 ((unsigned)(int)fx << 2);
            p.off = p.in ? off : 0u;
            p.tv = target_load(tbase, toff);
            return p;
        };
        // lo = rows (y0, y1) of the lower source plane, up = of the upper one; the reference's interpolation order
        auto accumulate = [&](const Step &p, f32x2 lo0, f32x2 lo1, f32x2 up0, f32x2 up1) {
            const f32x2 p00 = p.wx * lo0, p01 = p.wx * lo1, p10 = p.wx * up0, p11 = p.wx * up1;
            const f32x2 lower = {p00.x + p00.y, p01.x + p01.y}, upper = {p10.x + p10.y, p11.x + p11.y};
            const f32x2 el = p.wy * lower, eu = p.wy * upper;
            const f32x2 e = {el.x + el.y, eu.x + eu.y};
            const f32x2 rz = p.wz * e;
            const float r = rz.x + rz.y;
            const float sv = __uint_as_float(p.in ? __float_as_uint(r) : 0u);
            // float x float is exact in double, so the fused form rounds exactly like multiply-then-add
            ss = __builtin_fma((double)sv, (double)sv, ss);
            st = __builtin_fma((double)sv, (double)p.tv, st);
        };
        // Software pipeline, one step deep: the gathers of step z are in flight while step z-1 is interpolated.
        // Plane reuse: when a lane moved exactly one source plane (same row pair, same x pair -- decided by offset
        // equality, so it can only skip a gather, never change a value) the upper plane of step z-1 is the lower
        // plane of step z; only the other lanes gather it.  ~2 instead of 4 gathers per voxel for the
        // near-axis-aligned matrices registration visits.
        Step cur = prepare(z0, toff);
        f32x2 lo0 = gather_pair(rsrc, cur.off, 0u), lo1 = gather_pair(rsrc, cur.off, s01);
        f32x2 up0 = gather_pair(rsrc, cur.off, s10), up1 = gather_pair(rsrc, cur.off, s11);
#pragma unroll 4
        for (int z = z0 + 1; z < z1; ++z) {
            toff += tstep;
            Step nxt = prepare(z, toff);
            // a lane outside the window needs no data: let it ride on the offset the plane reuse expects, so it never
            // triggers the (wave-wide) lower-plane gather; past the end of the buffer the gathers return 0
            nxt.off = nxt.in ? nxt.off : cur.off + (unsigned)a.plane4;
            const f32x2 n0 = gather_pair(rsrc, nxt.off, s10), n1 = gather_pair(rsrc, nxt.off, s11);
            accumulate(cur, lo0, lo1, up0, up1);
            if (nxt.off != cur.off + (unsigned)a.plane4) {
                lo0 = gather_pair(rsrc, nxt.off, 0u);
                lo1 = gather_pair(rsrc, nxt.off, s01);
            } else {
                lo0 = up0;
                lo1 = up1;
            }
            up0 = n0;
            up1 = n1;
            cur = nxt;
        }
        accumulate(cur, lo0, lo1, up0, up1);
    }
    ss = corr_wave_sum(ss);
    st = corr_wave_sum(st);
    if (lane == 0) {
        lds[wave] = ss;
        lds[kTileY + wave] = st;
    }
    __syncthreads();
    double a0 = 0.0, a1 = 0.0;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 0; w < kTileY; ++w) {
            a0 += lds[w];
            a1 += lds[kTileY + w];
        }
    }
    corr_publish_as<kThreads>(bx + a.gx * (by + a.gy * bz), a0, a1, tail, lds, &last);
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 gather_quad(__amdgpu_buffer_rsrc_t rsrc, unsigned voffset, unsigned soffset) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voffset, (int)soffset, 0);
    return __builtin_bit_cast(f32x4, v);
}

// Row-pair layout: element (z, y, x) of the padded volume holds the float2 (src[z][y][x], src[z][y+1][x]), so ONE
// 16-byte gather at (iz, iy, ix) returns the four taps of a source plane: (y0 x0, y1 x0, y0 x1, y1 x1).  Two gathers
// per voxel whatever the matrix (no reuse between z steps to lose when lanes drift across planes), at the price of a
// source stored -- and streamed from HBM -- twice.
template <bool CACHE>
__global__ __launch_bounds__(kThreads) void corr_rowpair_kernel(const float *__restrict__ tgt,
                                                              const float *__restrict__ srcp, PaddedArgs a,
                                                              CorrTail tail) {
    __shared__ double lds[2 * kTileY];
    __shared__ int last;
    const int lane = threadIdx.x & (kTileX - 1), wave = threadIdx.x / kTileX;
    int bx, by, bz;
    corr_tile_of_block(a.gx, a.gy, bx, by, bz);
    const int x = bx * kTileX + lane;
    const int y = by * kTileY + wave;
    const int z0 = bz * kStepsZ;
    double ss = 0.0, st = 0.0;
    if (x < a.tx && y < a.ty) {
        const float ix = (float)x, iy = (float)y;
        const int z1 = (z0 + kStepsZ < a.tz) ? z0 + kStepsZ : a.tz;
        const f32x2 mxy0 = {a.m[0], a.m[4]}, mxy1 = {a.m[1], a.m[5]}, mxy2 = {a.m[2], a.m[6]}, mxy3 = {a.m[3], a.m[7]};
        const f32x2 hxy = mxy0 * ix + mxy1 * iy;
        const float hz = a.m[8] * ix + a.m[9] * iy;
        const f32x2 half2 = {0.5f, 0.5f};
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(srcp), 0, (int)a.bytes, 0x00020000);
        const unsigned s1 = (unsigned)a.plane4;  // here: plane pitch of the float2 volume, in bytes
        const char *tbase = reinterpret_cast<const char *>(tgt);
        unsigned toff = 4u * ((unsigned)x + (unsigned)a.tx * ((unsigned)y + (unsigned)a.ty * (unsigned)z0));
        const unsigned tstep = 4u * (unsigned)a.tx * (unsigned)a.ty;
        struct Step {
            float wx0, wx1;
            f32x2 wy, wz;
            unsigned off;
            bool in;
            float tv;
        };
        auto prepare = [&](int z, unsigned toff) {
            Step p;
            const float iz = (float)z;
            const f32x2 txy = ((hxy + mxy2 * iz) + mxy3) + half2;
            const float tz = ((hz + a.m[10] * iz) + a.m[11]) + 0.5f;
            const bool gx = txy.x > 0, lx = txy.x < a.sxf, gy = txy.y > 0, ly = txy.y < a.syf, gz = tz > 0, lz = tz < a.szf;
            p.in = gx && lx && gy && ly && gz && lz;
            const f32x2 bxy = txy - half2;
            const float bz = tz - 0.5f;
            const float fx = floorf(bxy.x), fy = floorf(bxy.y), fz = floorf(bz);
            p.wx1 = bxy.x - fx;
            p.wx0 = 1.0f - p.wx1;
            p.wy.y = bxy.y - fy;
            p.wy.x = 1.0f - p.wy.y;
            p.wz.y = bz - fz;
            p.wz.x = 1.0f - p.wz.y;
            const unsigned off = (unsigned)(__mul24((int)fz, a.plane4) + __mul24((int)fy, a.row4) + a.origin4) +
                                 ((unsigned)(int)fx << 3);
            p.off = p.in ? off : 0u;
            p.tv = target_load(tbase, toff);
            return p;
        };
        auto accumulate = [&](const Step &p, f32x4 lo, f32x4 up) {
            const f32x2 lo_a = {lo.x, lo.y}, lo_b = {lo.z, lo.w}, up_a = {up.x, up.y}, up_b = {up.z, up.w};
            const f32x2 cl = lo_a * p.wx0 + lo_b * p.wx1;  // (row y0, row y1) of the lower plane: w0*s(x0) + w1*s(x1)
            const f32x2 cu = up_a * p.wx0 + up_b * p.wx1;
            const f32x2 el = p.wy * cl, eu = p.wy * cu;
            const f32x2 e = {el.x + el.y, eu.x + eu.y};
            const f32x2 rz = p.wz * e;
            const float r = rz.x + rz.y;
            const float sv = __uint_as_float(p.in ? __float_as_uint(r) : 0u);
            ss = __builtin_fma((double)sv, (double)sv, ss);
            st = __builtin_fma((double)sv, (double)p.tv, st);
        };
        Step cur = prepare(z0, toff);
        f32x4 lo = gather_quad(rsrc, cur.off, 0u), up = gather_quad(rsrc, cur.off, s1);
#pragma unroll 4
        for (int z = z0 + 1; z < z1; ++z) {
            toff += tstep;
            Step nxt = prepare(z, toff);
            if (CACHE) nxt.off = nxt.in ? nxt.off : cur.off + s1;  // out of the window: never trigger the lower gather
            const f32x4 nu = gather_quad(rsrc, nxt.off, s1);
            f32x4 nl;
            if (!CACHE) nl = gather_quad(rsrc, nxt.off, 0u);
            accumulate(cur, lo, up);
            if (CACHE) {
                if (nxt.off != cur.off + s1) lo = gather_quad(rsrc, nxt.off, 0u);
                else lo = up;
            } else {
                lo = nl;
            }
            up = nu;
            cur = nxt;
        }
        accumulate(cur, lo, up);
    }
    ss = corr_wave_sum(ss);
    st = corr_wave_sum(st);
    if (lane == 0) {
        lds[wave] = ss;
        lds[kTileY + wave] = st;
    }
    __syncthreads();
    double a0 = 0.0, a1 = 0.0;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 0; w < kTileY; ++w) {
            a0 += lds[w];
            a1 += lds[kTileY + w];
        }
    }
    corr_publish_as<kThreads>(bx + a.gx * (by + a.gy * bz), a0, a1, tail, lds, &last);
}

// ------------------------------------------------------------------------------------------------------------------
// Two voxels per lane, 64 apart in x.  The cost kernels above spend ~62 VALU instructions per voxel, 71 % of their run time
// in VALU issue (rocprofv3 PMC, profiles/r01_final/reg_cost_kernel.json), and one address-path pass per voxel on a
// 4-byte target load.  Here a lane owns target voxels (x, x + 64) -- the same lane of two neighbouring 64-wide workgroups,
// so every gather of a wave still covers 64 consecutive source columns:
//   * the coordinate chains of the two voxels run as ONE chain of 2-wide operations (v_pk_add_f32): each voxel's value
//     goes through the reference's own sequence of float32 operations, so the bits do not change;
//   * the validity window costs six compares per voxel -- but t is a non-decreasing (or non-increasing) function of x and
//     of z under float32 evaluation (every operation of ((m0*x + m1*y) + m2*z) + m3 + 0.5 is monotone), so a lane
//     whose two voxels lie in the window at the first AND the last z of its column segment lies in it throughout.  A wave
//     in which that holds for every lane (all waves but those at the rim of the overlap) runs a loop without
//     compares, selects or offset substitution.
// Each voxel keeps its own float64 accumulators, and the sums are combined in the order the one-voxel-per-lane kernels
// use -- a wave here covers two of their waves, a workgroup two of their workgroups, published separately
// (corr_publish_as) -- so the z-marching kernels (0-5) all return the same bits.  Kernel 6 (MARCHY below) adds the same
// float32 voxel values in another order: equal to ~1e-15 relative, and since the choice is made per matrix, ONE Powell
// trajectory mixes the two summation orders (the full-size trajectory test still sees the oracle's matrices bit for bit).
//
// MARCHY (plain layout with reuse only): the lane marches along y inside ONE target plane instead of along z -- tiles of
// 128 (x) x 16 (y steps) x 4 (z: the waves) -- and reuses source ROWS instead of planes.  Which axis to march along decides
// how often a wave needs its wave-wide re-gather: per step the two coordinates ACROSS the marching axis drift by
// (m2, m6) along z but by (m1, m9) along y, and the coordinate ALONG it skips an index when its scale differs from 1 -- for
// the lanes of a wave together if that coordinate does not depend on x.  A rotation about y with a 2 % scale (bench.py's
// matrix) re-gathers in 83 % of the wave-steps along z and in 2 % along y: 0.202 -> 0.161 ms per evaluation, past the
// row-pair kernels' 0.175 on 2/3 of their HBM traffic (profiles/r03/reg_cost_variants_march_y.txt).  The host picks the
// axis per matrix (pick_kernel).  Same voxels, same float32 values; the float64 partial sums are formed in another order
// (16 y steps per lane), so the two sums agree with the other kernels' to ~1e-15 relative, not bit for bit.
template <bool PAIRS, bool REUSE, bool MARCHY = false>
__device__ __forceinline__ void corr_pair_body(const float *__restrict__ tgt, const float *__restrict__ srcp,
                                               const PaddedArgs &a, const CorrTail &tail, int old_grid_x, double *lds,
                                               int *last_p) {
    const int lane = threadIdx.x & (kTileX - 1), wave = threadIdx.x / kTileX;
    int bx, by, bz;
    corr_tile_of_block(a.gx, a.gy, bx, by, bz);
    const int x = bx * 2 * kTileX + lane;  // voxel A; voxel B sits kTileX further on (the next 64-wide workgroup of the
                                           // one-voxel-per-lane decomposition); tx % 128 == 0 (host check)
    static_assert(!MARCHY || (!PAIRS && REUSE), "the y-marching form exists for the plain layout with reuse");
    // `fixed` = the coordinate of this wave across the marching axis (y, or z when marching along y); [z0, z1) = the steps
    const int fixed = by * kTileY + wave;
    const int z0 = bz * kStepsZ;
    const int n_steps = MARCHY ? a.ty : a.tz;
    const int z1 = (z0 + kStepsZ < n_steps) ? z0 + kStepsZ : n_steps;
    double ssA = 0.0, stA = 0.0, ssB = 0.0, stB = 0.0;
    const bool active = x < a.tx && fixed < (MARCHY ? a.tz : a.ty);
    const float ifix = (float)fixed;
    const f32x2 ixs = {(float)x, (float)(x + kTileX)};
    const f32x2 half2 = {0.5f, 0.5f};
    // t = ((m0*x + m1*y) + m2*z) + m3, then + 0.5: the reference's left-to-right float32 evaluation for both voxels
    // (voxel A, voxel B).  Marching along z the first sum is constant; marching along y the z product is.
    const f32x2 PX = ixs * a.m[0], PY = ixs * a.m[4], PZ = ixs * a.m[8];
    const float cx = (MARCHY ? a.m[2] : a.m[1]) * ifix, cy = (MARCHY ? a.m[6] : a.m[5]) * ifix, cz = (MARCHY ? a.m[10] : a.m[9]) * ifix;
    const f32x2 HX = PX + (f32x2){cx, cx}, HY = PY + (f32x2){cy, cy}, HZ = PZ + (f32x2){cz, cz};  // z-marching form only
    auto coords = [&](int step, f32x2 &TX, f32x2 &TY, f32x2 &TZ) {
        const float is = (float)step;
        if constexpr (MARCHY) {
            const float yx = a.m[1] * is, yy = a.m[5] * is, yz = a.m[9] * is;
            TX = (((PX + (f32x2){yx, yx}) + (f32x2){cx, cx}) + (f32x2){a.m[3], a.m[3]}) + half2;
            TY = (((PY + (f32x2){yy, yy}) + (f32x2){cy, cy}) + (f32x2){a.m[7], a.m[7]}) + half2;
            TZ = (((PZ + (f32x2){yz, yz}) + (f32x2){cz, cz}) + (f32x2){a.m[11], a.m[11]}) + half2;
        } else {
            const float zx = a.m[2] * is, zy = a.m[6] * is, zz = a.m[10] * is;
            TX = ((HX + (f32x2){zx, zx}) + (f32x2){a.m[3], a.m[3]}) + half2;
            TY = ((HY + (f32x2){zy, zy}) + (f32x2){a.m[7], a.m[7]}) + half2;
            TZ = ((HZ + (f32x2){zz, zz}) + (f32x2){a.m[11], a.m[11]}) + half2;
        }
    };
    auto inside = [&](float tx_, float ty_, float tz_) {
        return tx_ > 0 && tx_ < a.sxf && ty_ > 0 && ty_ < a.syf && tz_ > 0 && tz_ < a.szf;
    };
    bool whole = true;  // both voxels in the window at both ends of the z segment
    if (active) {
        f32x2 TX, TY, TZ;
        coords(z0, TX, TY, TZ);
        whole = inside(TX.x, TY.x, TZ.x) && inside(TX.y, TY.y, TZ.y);
        coords(z1 - 1, TX, TY, TZ);
        whole = whole && inside(TX.x, TY.x, TZ.x) && inside(TX.y, TY.y, TZ.y);
    }
    const bool fast = __all(whole) != 0;  // wave-uniform

    // One voxel of one z step between `prepare` and `accumulate`.  Only the upper weight of each axis is carried; 1 - w is
    // formed where it is used: 90 -> 82 registers.  (Compiled for SIX waves per SIMD -- 80 registers, nothing of the z loops
    // spilled -- the kernels are SLOWER: 0.152 / 0.188-0.200 against 0.146 / 0.175 ms, profiles/r03/
    // reg_cost_variants_six_waves_slim.txt; the five waves the compiler picks stay.)
    struct Vox {
        float wx1, wy1, wz1;
        unsigned off;
        bool in;
        float tv;
    };
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(srcp), 0, (int)a.bytes, 0x00020000);
    const unsigned sRow = (unsigned)a.row4, sPlane = (unsigned)a.plane4, sBoth = (unsigned)(a.plane4 + a.row4);
    const char *tbase = reinterpret_cast<const char *>(tgt);
    const unsigned tstep = MARCHY ? 4u * (unsigned)a.tx : 4u * (unsigned)a.tx * (unsigned)a.ty;
    // the source step the reuse expects (one row / one plane further) and the offsets of the four (x0, x1) gathers of a
    // voxel: (plane z0, plane z1) of one row when marching along y, (row y0, row y1) of one plane when marching along z
    const unsigned sStep = MARCHY ? sRow : sPlane, sSide = MARCHY ? sPlane : sRow;

    auto run = [&](auto check_tag) {
        constexpr bool CHECK = decltype(check_tag)::value;
        unsigned toff = MARCHY ? 4u * ((unsigned)x + (unsigned)a.tx * ((unsigned)z0 + (unsigned)a.ty * (unsigned)fixed))
                               : 4u * ((unsigned)x + (unsigned)a.tx * ((unsigned)fixed + (unsigned)a.ty * (unsigned)z0));
        auto prepare = [&](int z, unsigned toff, Vox &A, Vox &B) {
            f32x2 TX, TY, TZ;
            coords(z, TX, TY, TZ);
            if (CHECK) {
                A.in = inside(TX.x, TY.x, TZ.x);
                B.in = inside(TX.y, TY.y, TZ.y);
            } else {
                A.in = B.in = true;
            }
            const f32x2 BX = TX - half2, BY = TY - half2, BZ = TZ - half2;
            const float fxA = floorf(BX.x), fyA = floorf(BY.x), fzA = floorf(BZ.x);
            const float fxB = floorf(BX.y), fyB = floorf(BY.y), fzB = floorf(BZ.y);
            A.wx1 = BX.x - fxA; A.wy1 = BY.x - fyA; A.wz1 = BZ.x - fzA;
            B.wx1 = BX.y - fxB; B.wy1 = BY.y - fyB; B.wz1 = BZ.y - fzB;
            constexpr int SH = PAIRS ? 3 : 2;
            {
                // Index arithmetic without float -> int conversions: f + 1.5 * 2^23 is exact for the integer f (|f| < 2^22)
                // and its bit pattern is 0x4B400000 + f, so the LOW 24 BITS are 2^22 + f -- exactly what the 24-bit
                // multiply-adds read.  One packed add serves both voxels; the constant parts (2^22 * pitch, the shifted
                // exponent bits) are folded into `origin4m` by the host (modulo 2^32, like every step of this sum).  A lane
                // outside the window gets a meaningless offset here -- and never uses it (CHECK path below).
                const f32x2 magic = {12582912.0f, 12582912.0f};
                const f32x2 MX = (f32x2){fxA, fxB} + magic, MY = (f32x2){fyA, fyB} + magic, MZ = (f32x2){fzA, fzB} + magic;
                auto mad24 = [](unsigned x, unsigned y, unsigned z) {
                    unsigned r;
                    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(x), "s"(y), "v"(z));
                    return r;
                };
                const unsigned row = (unsigned)a.row4, plane = (unsigned)a.plane4, org = (unsigned)a.origin4m;
                A.off = mad24(__float_as_uint(MZ.x), plane, mad24(__float_as_uint(MY.x), row, (__float_as_uint(MX.x) << SH) + org));
                B.off = mad24(__float_as_uint(MZ.y), plane, mad24(__float_as_uint(MY.y), row, (__float_as_uint(MX.y) << SH) + org));
            }
            if (CHECK) {
                A.off = A.in ? A.off : 0u;
                B.off = B.in ? B.off : 0u;
            }
            A.tv = target_load(tbase, toff);
            B.tv = target_load(tbase, toff + 4u * kTileX);
        };
        auto finish = [&](const Vox &p, float r, double &ss, double &st) {
            const float sv = CHECK ? __uint_as_float(p.in ? __float_as_uint(r) : 0u) : r;
            ss = __builtin_fma((double)sv, (double)sv, ss);
            st = __builtin_fma((double)sv, (double)p.tv, st);
        };
        if constexpr (!PAIRS) {
            // plain layout: 8-byte gathers of (x0, x1) at rows y0, y1 of the lower and the upper plane
            auto accumulate = [&](const Vox &p, f32x2 lo0, f32x2 lo1, f32x2 up0, f32x2 up1, double &ss, double &st) {
                const f32x2 wx = {1.0f - p.wx1, p.wx1}, wy = {1.0f - p.wy1, p.wy1}, wz = {1.0f - p.wz1, p.wz1};
                const f32x2 p00 = wx * lo0, p01 = wx * lo1, p10 = wx * up0, p11 = wx * up1;
                const f32x2 lower = {p00.x + p00.y, p01.x + p01.y}, upper = {p10.x + p10.y, p11.x + p11.y};
                const f32x2 el = wy * lower, eu = wy * upper;
                const f32x2 e = {el.x + el.y, eu.x + eu.y};
                const f32x2 rz = wz * e;
                finish(p, rz.x + rz.y, ss, st);
            };
            struct Taps { f32x2 r0, r1; };  // the two (x0, x1) pairs of one step: rows y0, y1 of a plane / planes z0, z1 of a row
            auto plane = [&](unsigned off, bool upper) {
                return Taps{gather_pair(rsrc, off, upper ? sStep : 0u), gather_pair(rsrc, off, upper ? sBoth : sSide)};
            };
            // the interpolation takes (z0 y0, z0 y1, z1 y0, z1 y1)
            auto interp = [&](const Vox &p, const Taps &lo, const Taps &up, double &ss, double &st) {
                if constexpr (MARCHY) accumulate(p, lo.r0, up.r0, lo.r1, up.r1, ss, st);
                else accumulate(p, lo.r0, lo.r1, up.r0, up.r1, ss, st);
            };
            Vox cA, cB;
            prepare(z0, toff, cA, cB);
            Taps aP = plane(cA.off, false), aQ = plane(cA.off, true), aR = aQ;
            Taps bP = plane(cB.off, false), bQ = plane(cB.off, true), bR = bQ;
            // One z step.  (lo, up, nx) name three register sets whose roles rotate from step to step, so the plane
            // reuse -- the upper plane of this step is the lower plane of the next -- is a renaming, not a copy: a lane
            // that did NOT move exactly one plane gathers its new lower plane INTO the set that held the upper one.
            auto step = [&](int z, Taps &aLo, Taps &aUp, Taps &aNx, Taps &bLo, Taps &bUp, Taps &bNx) {
                toff += tstep;
                Vox nA, nB;
                prepare(z, toff, nA, nB);
                if (CHECK && REUSE) {  // out of the window: ride on the offset the plane reuse expects
                    nA.off = nA.in ? nA.off : cA.off + sStep;
                    nB.off = nB.in ? nB.off : cB.off + sStep;
                }
                aNx = plane(nA.off, true);
                bNx = plane(nB.off, true);
                Taps al, bl;
                if (!REUSE) {
                    al = plane(nA.off, false);
                    bl = plane(nB.off, false);
                }
                interp(cA, aLo, aUp, ssA, stA);
                interp(cB, bLo, bUp, ssB, stB);
                if (REUSE) {
                    // keep the (rare) re-gather BELOW the interpolation that still reads the upper plane (the empty asm
                    // pins the four sums here): issued above it, the load needs a copy of the register set -- 4 moves
                    // per voxel, measured in the ISA
                    asm volatile("" : "+v"(ssA), "+v"(stA), "+v"(ssB), "+v"(stB) : : "memory");
                    if (nA.off != cA.off + sStep) aUp = plane(nA.off, false);
                    if (nB.off != cB.off + sStep) bUp = plane(nB.off, false);
                } else {
                    aUp = al;
                    bUp = bl;
                }
                cA = nA;
                cB = nB;
            };
            int z = z0 + 1;
            for (; z + 2 < z1; z += 3) {
                step(z, aP, aQ, aR, bP, bQ, bR);      // afterwards: lower = Q, upper = R
                step(z + 1, aQ, aR, aP, bQ, bR, bP);  // lower = R, upper = P
                step(z + 2, aR, aP, aQ, bR, bP, bQ);  // lower = P, upper = Q
            }
            for (; z < z1; ++z) {  // remainder (a z segment cut short by the volume): rotate by copying
                step(z, aP, aQ, aR, bP, bQ, bR);
                aP = aQ; aQ = aR; bP = bQ; bQ = bR;
            }
            interp(cA, aP, aQ, ssA, stA);
            interp(cB, bP, bQ, ssB, stB);
        } else {
            // row-pair layout: one 16-byte gather = (y0 x0, y1 x0, y0 x1, y1 x1) of a plane
            auto accumulate = [&](const Vox &p, f32x4 lo, f32x4 up, double &ss, double &st) {
                const float wx0 = 1.0f - p.wx1;
                const f32x2 wy = {1.0f - p.wy1, p.wy1}, wz = {1.0f - p.wz1, p.wz1};
                const f32x2 lo_a = {lo.x, lo.y}, lo_b = {lo.z, lo.w}, up_a = {up.x, up.y}, up_b = {up.z, up.w};
                const f32x2 cl = lo_a * wx0 + lo_b * p.wx1;
                const f32x2 cu = up_a * wx0 + up_b * p.wx1;
                const f32x2 el = wy * cl, eu = wy * cu;
                const f32x2 e = {el.x + el.y, eu.x + eu.y};
                const f32x2 rz = wz * e;
                finish(p, rz.x + rz.y, ss, st);
            };
            Vox cA, cB;
            prepare(z0, toff, cA, cB);
            f32x4 aP = gather_quad(rsrc, cA.off, 0u), aQ = gather_quad(rsrc, cA.off, sPlane), aR = aQ;
            f32x4 bP = gather_quad(rsrc, cB.off, 0u), bQ = gather_quad(rsrc, cB.off, sPlane), bR = bQ;
            auto step = [&](int z, f32x4 &aLo, f32x4 &aUp, f32x4 &aNx, f32x4 &bLo, f32x4 &bUp, f32x4 &bNx) {
                toff += tstep;
                Vox nA, nB;
                prepare(z, toff, nA, nB);
                if (CHECK && REUSE) {
                    nA.off = nA.in ? nA.off : cA.off + sPlane;
                    nB.off = nB.in ? nB.off : cB.off + sPlane;
                }
                aNx = gather_quad(rsrc, nA.off, sPlane);
                bNx = gather_quad(rsrc, nB.off, sPlane);
                f32x4 al, bl;
                if (!REUSE) {
                    al = gather_quad(rsrc, nA.off, 0u);
                    bl = gather_quad(rsrc, nB.off, 0u);
                }
                accumulate(cA, aLo, aUp, ssA, stA);
                accumulate(cB, bLo, bUp, ssB, stB);
                if (REUSE) {
                    asm volatile("" : "+v"(ssA), "+v"(stA), "+v"(ssB), "+v"(stB) : : "memory");  // as in the plain layout
                    if (nA.off != cA.off + sPlane) aUp = gather_quad(rsrc, nA.off, 0u);
                    if (nB.off != cB.off + sPlane) bUp = gather_quad(rsrc, nB.off, 0u);
                } else {
                    aUp = al;
                    bUp = bl;
                }
                cA = nA;
                cB = nB;
            };
            int z = z0 + 1;
            for (; z + 2 < z1; z += 3) {
                step(z, aP, aQ, aR, bP, bQ, bR);
                step(z + 1, aQ, aR, aP, bQ, bR, bP);
                step(z + 2, aR, aP, aQ, bR, bP, bQ);
            }
            for (; z < z1; ++z) {
                step(z, aP, aQ, aR, bP, bQ, bR);
                aP = aQ; aQ = aR; bP = bQ; bQ = bR;
            }
            accumulate(cA, aP, aQ, ssA, stA);
            accumulate(cB, bP, bQ, ssB, stB);
        }
    };
    if (active) {
        if (fast) run(std::false_type{});
        else run(std::true_type{});
    }
    // Voxel A is a lane of one 64-wide workgroup of the one-voxel-per-lane decomposition, voxel B the same lane of the next
    // one: the same shuffle tree on each, the same order over the waves, two publishes.
    ssA = corr_wave_sum(ssA);
    stA = corr_wave_sum(stA);
    ssB = corr_wave_sum(ssB);
    stB = corr_wave_sum(stB);
    if (lane == 0) {
        lds[wave] = ssA;
        lds[kTileY + wave] = stA;
        lds[2 * kTileY + wave] = ssB;
        lds[3 * kTileY + wave] = stB;
    }
    __syncthreads();
    double v0[2] = {0.0, 0.0}, v1[2] = {0.0, 0.0};
    if (threadIdx.x == 0) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int w = 0; w < kTileY; ++w) {
                v0[h] += lds[h * 2 * kTileY + w];
                v1[h] += lds[h * 2 * kTileY + kTileY + w];
            }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int obx = 2 * bx + h;  // workgroup x index of the 64-wide decomposition
        if (obx < old_grid_x)
            corr_publish_as<kThreads>(obx + old_grid_x * (by + a.gy * bz), v0[h], v1[h], tail, lds, last_p);
    }
}

template <bool PAIRS, bool REUSE, bool MARCHY = false>
__global__ __launch_bounds__(kThreads) void corr_pair_kernel(const float *__restrict__ tgt,
                                                           const float *__restrict__ srcp, PaddedArgs a,
                                                           CorrTail tail, int old_grid_x) {
    __shared__ double lds[4 * kTileY];
    __shared__ int last;
    corr_pair_body<PAIRS, REUSE, MARCHY>(tgt, srcp, a, tail, old_grid_x, lds, &last);
}

// ------------------------------------------------------------------------------------------------------------------
// Chunked form of the two-voxel kernels (round 5).  The tiled kernels above are bound by LATENCY, not bytes: a wave has ONE
// step in flight, a step is one memory round trip under load (~1.4 us), and every 16-step tile adds a workgroup launch, a
// cold first step (lower + upper plane, target) and two publish round trips -- about 8 us of the ~23 us a tile occupies its
// slot (z steps per tile 8 / 16 / 32: 0.198 / 0.146 / 0.145 ms, the last one losing to the 3.2 -> 4 rounds of its 4 096
// tiles on 1 280 slots what it gains per tile).  Here the grid is what the chip holds at once and a workgroup walks a
// CONTIGUOUS piece of the (column, step) sequence -- columns of 128 x 4 voxels, steps along the marching axis, the pieces
// equal to within one step whatever the volume -- so a slot pays one launch, one or two cold steps and ONE publish per
// evaluation.  The float32 voxel values are those of every other kernel; a lane's float64 sums now run over its whole piece
// (~100 steps) and the number of pieces follows the device, so the two sums agree with the tiled kernels' to ~1e-15 relative
// (like kernel 6), and are bit-reproducible from run to run on one device.
#ifndef CHUNK_WAVES
#define CHUNK_WAVES 5
#endif
constexpr int kSubRun = 24;  // steps between two decisions "may this wave drop the window tests" (a multiple of 3: the
                             // plane roles of the rotating register sets are back in place)

// RAGGED (round 6): target widths that are NOT a multiple of 128 -- every volume that was not cut to an FFT-friendly size, e.g.
// 500 x 500 x 250, which used to fall back to the one-voxel tiled kernels (0.171 against 0.123 ms per evaluation at the identity
// class).  A lane beyond the width follows the row's LAST voxel -- valid offsets, valid target address -- and is kept out of both
// sums like a lane outside the window (in = false); a wave that holds such lanes never drops its window tests.
template <bool PAIRS, bool REUSE, bool MARCHY, bool RAGGED = false>
__global__ __launch_bounds__(kThreads, CHUNK_WAVES) void corr_chunk_kernel(const float *__restrict__ tgt,
                                                            const float *__restrict__ srcp, PaddedArgs a, CorrTail tail,
                                                            unsigned total_steps, int run_len) {
    static_assert(!MARCHY || (!PAIRS && REUSE), "the y-marching form exists for the plain layout with reuse");
    __shared__ double lds[4 * kTileY];
    __shared__ int last;
    const int lane = threadIdx.x & (kTileX - 1), wave = threadIdx.x / kTileX;
    const unsigned long long tot = total_steps;
    const unsigned lo = (unsigned)(tot * blockIdx.x / gridDim.x), hi = (unsigned)(tot * (blockIdx.x + 1) / gridDim.x);
    const f32x2 half2 = {0.5f, 0.5f};
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(srcp), 0, (int)a.bytes, 0x00020000);
    const unsigned sRow = (unsigned)a.row4, sPlane = (unsigned)a.plane4, sBoth = (unsigned)(a.plane4 + a.row4);
    const char *tbase = reinterpret_cast<const char *>(tgt);
    const unsigned tstep = MARCHY ? 4u * (unsigned)a.tx : 4u * (unsigned)a.tx * (unsigned)a.ty;
    const unsigned sStep = MARCHY ? sRow : sPlane, sSide = MARCHY ? sPlane : sRow;
    double ssA = 0.0, stA = 0.0, ssB = 0.0, stB = 0.0;

    struct Vox {
        float wx1, wy1, wz1;
        unsigned off;
        bool in;
        float tv;
    };
    auto inside = [&](float tx_, float ty_, float tz_) {
        return tx_ > 0 && tx_ < a.sxf && ty_ > 0 && ty_ < a.syf && tz_ > 0 && tz_ < a.szf;
    };

    for (unsigned L = lo; L < hi;) {
        const unsigned col = L / (unsigned)run_len;
        const int s0 = (int)(L - col * (unsigned)run_len);
        const int s1 = (int)min((unsigned)run_len, (unsigned)s0 + (hi - L));
        L += (unsigned)(s1 - s0);
        const int bx = (int)(col % (unsigned)a.gx), by = (int)(col / (unsigned)a.gx);
        const int x0 = bx * 2 * kTileX + lane;  // voxel A; voxel B sits kTileX further on; tx % 128 == 0 unless RAGGED (host check)
        const bool okA = !RAGGED || x0 < a.tx, okB = !RAGGED || x0 + kTileX < a.tx;
        const int x = RAGGED ? min(x0, a.tx - 1) : x0, xB = RAGGED ? min(x0 + kTileX, a.tx - 1) : x0 + kTileX;
        const unsigned dB = 4u * (unsigned)(xB - x);  // byte distance of voxel B's target value from voxel A's
        const int fixed = by * kTileY + wave;  // this wave's coordinate across the marching axis (y, or z when marching along y)
        if (fixed >= (MARCHY ? a.tz : a.ty)) continue;  // wave-uniform
        const float ifix = (float)fixed;
        const f32x2 ixs = {(float)x, (float)xB};
        // t = ((m0*x + m1*y) + m2*z) + m3, then + 0.5: the reference's left-to-right float32 evaluation (corr_pair_body)
        const f32x2 PX = ixs * a.m[0], PY = ixs * a.m[4], PZ = ixs * a.m[8];
        const float cx = (MARCHY ? a.m[2] : a.m[1]) * ifix, cy = (MARCHY ? a.m[6] : a.m[5]) * ifix, cz = (MARCHY ? a.m[10] : a.m[9]) * ifix;
        const f32x2 HX = PX + (f32x2){cx, cx}, HY = PY + (f32x2){cy, cy}, HZ = PZ + (f32x2){cz, cz};  // z-marching form only
        auto coords = [&](int step, f32x2 &TX, f32x2 &TY, f32x2 &TZ) {
            const float is = (float)step;
            if constexpr (MARCHY) {
                const float yx = a.m[1] * is, yy = a.m[5] * is, yz = a.m[9] * is;
                TX = (((PX + (f32x2){yx, yx}) + (f32x2){cx, cx}) + (f32x2){a.m[3], a.m[3]}) + half2;
                TY = (((PY + (f32x2){yy, yy}) + (f32x2){cy, cy}) + (f32x2){a.m[7], a.m[7]}) + half2;
                TZ = (((PZ + (f32x2){yz, yz}) + (f32x2){cz, cz}) + (f32x2){a.m[11], a.m[11]}) + half2;
            } else {
                const float zx = a.m[2] * is, zy = a.m[6] * is, zz = a.m[10] * is;
                TX = ((HX + (f32x2){zx, zx}) + (f32x2){a.m[3], a.m[3]}) + half2;
                TY = ((HY + (f32x2){zy, zy}) + (f32x2){a.m[7], a.m[7]}) + half2;
                TZ = ((HZ + (f32x2){zz, zz}) + (f32x2){a.m[11], a.m[11]}) + half2;
            }
        };
        unsigned toff = MARCHY ? 4u * ((unsigned)x + (unsigned)a.tx * ((unsigned)s0 + (unsigned)a.ty * (unsigned)fixed))
                               : 4u * ((unsigned)x + (unsigned)a.tx * ((unsigned)fixed + (unsigned)a.ty * (unsigned)s0));
        // chk (wave-uniform, a scalar branch): false = every lane of the wave is known to lie in the window at this step, so
        // the six compares and the offset substitutions are jumped over.  ONE body for both cases: as two instantiations
        // that hand the rotating register sets to each other the kernel needed 113 registers instead of 78.
        // ride: a lane outside the window takes the offset the reuse expects (prev + one step) instead of 0, so that it
        // never triggers the wave-wide re-gather.
        auto prepare = [&](bool chk, bool ride, int z, unsigned toff, const Vox &pA, const Vox &pB, Vox &A, Vox &B) {
            f32x2 TX, TY, TZ;
            coords(z, TX, TY, TZ);
            const f32x2 BX = TX - half2, BY = TY - half2, BZ = TZ - half2;
            const float fxA = floorf(BX.x), fyA = floorf(BY.x), fzA = floorf(BZ.x);
            const float fxB = floorf(BX.y), fyB = floorf(BY.y), fzB = floorf(BZ.y);
            A.wx1 = BX.x - fxA; A.wy1 = BY.x - fyA; A.wz1 = BZ.x - fzA;
            B.wx1 = BX.y - fxB; B.wy1 = BY.y - fyB; B.wz1 = BZ.y - fzB;
            constexpr int SH = PAIRS ? 3 : 2;
            const f32x2 magic = {12582912.0f, 12582912.0f};  // index arithmetic on float bit patterns: see corr_pair_body
            const f32x2 MX = (f32x2){fxA, fxB} + magic, MY = (f32x2){fyA, fyB} + magic, MZ = (f32x2){fzA, fzB} + magic;
            auto mad24 = [](unsigned x, unsigned y, unsigned z) {
                unsigned r;
                asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(x), "s"(y), "v"(z));
                return r;
            };
            const unsigned row = (unsigned)a.row4, plane = (unsigned)a.plane4, org = (unsigned)a.origin4m;
            A.off = mad24(__float_as_uint(MZ.x), plane, mad24(__float_as_uint(MY.x), row, (__float_as_uint(MX.x) << SH) + org));
            B.off = mad24(__float_as_uint(MZ.y), plane, mad24(__float_as_uint(MY.y), row, (__float_as_uint(MX.y) << SH) + org));
            A.in = B.in = true;
            if (chk) {
                A.in = okA && inside(TX.x, TY.x, TZ.x);
                B.in = okB && inside(TX.y, TY.y, TZ.y);
                A.off = A.in ? A.off : (ride ? pA.off + sStep : 0u);
                B.off = B.in ? B.off : (ride ? pB.off + sStep : 0u);
            }
            A.tv = target_load(tbase, toff);
            B.tv = target_load(tbase, toff + (RAGGED ? dB : 4u * kTileX));
        };
        auto finish = [&](const Vox &p, float r, double &ss, double &st) {
            const float sv = __uint_as_float(p.in ? __float_as_uint(r) : 0u);
            // float x float is exact in double, so the fused form rounds exactly like multiply-then-add
            ss = __builtin_fma((double)sv, (double)sv, ss);
            st = __builtin_fma((double)sv, (double)p.tv, st);
        };
        // May the wave drop the tests for the steps (z - 1 .. ze - 1]?  The coordinates are monotone along the marching axis
        // under float32 evaluation: in the window at step z - 1 (the one waiting to be interpolated) and at step ze - 1 means
        // in the window throughout.
        auto all_inside = [&](const Vox &A, const Vox &B, int ze) {
            f32x2 TX, TY, TZ;
            coords(ze - 1, TX, TY, TZ);
            return __all(okA && okB && A.in && B.in && inside(TX.x, TY.x, TZ.x) && inside(TX.y, TY.y, TZ.y)) != 0;
        };
        if constexpr (!PAIRS) {
            auto accumulate = [&](const Vox &p, f32x2 lo0, f32x2 lo1, f32x2 up0, f32x2 up1, double &ss, double &st) {
                const f32x2 wx = {1.0f - p.wx1, p.wx1}, wy = {1.0f - p.wy1, p.wy1}, wz = {1.0f - p.wz1, p.wz1};
                const f32x2 p00 = wx * lo0, p01 = wx * lo1, p10 = wx * up0, p11 = wx * up1;
                const f32x2 lower = {p00.x + p00.y, p01.x + p01.y}, upper = {p10.x + p10.y, p11.x + p11.y};
                const f32x2 el = wy * lower, eu = wy * upper;
                const f32x2 e = {el.x + el.y, eu.x + eu.y};
                const f32x2 rz = wz * e;
                finish(p, rz.x + rz.y, ss, st);
            };
            struct Taps { f32x2 r0, r1; };
            auto plane = [&](unsigned off, bool upper) {
                return Taps{gather_pair(rsrc, off, upper ? sStep : 0u), gather_pair(rsrc, off, upper ? sBoth : sSide)};
            };
            auto interp = [&](const Vox &p, const Taps &lo, const Taps &up, double &ss, double &st) {
                if constexpr (MARCHY) accumulate(p, lo.r0, up.r0, lo.r1, up.r1, ss, st);
                else accumulate(p, lo.r0, lo.r1, up.r0, up.r1, ss, st);
            };
            Vox cA, cB;
            prepare(true, false, s0, toff, cA, cB, cA, cB);
            Taps aP = plane(cA.off, false), aQ = plane(cA.off, true), aR = aQ;
            Taps bP = plane(cB.off, false), bQ = plane(cB.off, true), bR = bQ;
            auto step = [&](bool chk, int z, Taps &aLo, Taps &aUp, Taps &aNx, Taps &bLo, Taps &bUp, Taps &bNx) {
                toff += tstep;
                Vox nA, nB;
                prepare(chk, REUSE, z, toff, cA, cB, nA, nB);
                aNx = plane(nA.off, true);
                bNx = plane(nB.off, true);
                Taps al, bl;
                if (!REUSE) {
                    al = plane(nA.off, false);
                    bl = plane(nB.off, false);
                }
                interp(cA, aLo, aUp, ssA, stA);
                interp(cB, bLo, bUp, ssB, stB);
                if (REUSE) {
                    asm volatile("" : "+v"(ssA), "+v"(stA), "+v"(ssB), "+v"(stB) : : "memory");  // see corr_pair_body
                    if (nA.off != cA.off + sStep) aUp = plane(nA.off, false);
                    if (nB.off != cB.off + sStep) bUp = plane(nB.off, false);
                } else {
                    aUp = al;
                    bUp = bl;
                }
                cA = nA;
                cB = nB;
            };
            for (int z = s0 + 1; z < s1;) {
                const int ze = min(z + kSubRun, s1);
                const bool chk = !all_inside(cA, cB, ze);
                for (; z + 2 < ze; z += 3) {
                    step(chk, z, aP, aQ, aR, bP, bQ, bR);      // afterwards: lower = Q, upper = R
                    step(chk, z + 1, aQ, aR, aP, bQ, bR, bP);  // lower = R, upper = P
                    step(chk, z + 2, aR, aP, aQ, bR, bP, bQ);  // lower = P, upper = Q
                }
                for (; z < ze; ++z) {  // the end of a run: rotate by copying
                    step(chk, z, aP, aQ, aR, bP, bQ, bR);
                    aP = aQ; aQ = aR; bP = bQ; bQ = bR;
                }
            }
            interp(cA, aP, aQ, ssA, stA);
            interp(cB, bP, bQ, ssB, stB);
        } else {
            auto accumulate = [&](const Vox &p, f32x4 lo, f32x4 up, double &ss, double &st) {
                const float wx0 = 1.0f - p.wx1;
                const f32x2 wy = {1.0f - p.wy1, p.wy1}, wz = {1.0f - p.wz1, p.wz1};
                const f32x2 lo_a = {lo.x, lo.y}, lo_b = {lo.z, lo.w}, up_a = {up.x, up.y}, up_b = {up.z, up.w};
                const f32x2 cl = lo_a * wx0 + lo_b * p.wx1;
                const f32x2 cu = up_a * wx0 + up_b * p.wx1;
                const f32x2 el = wy * cl, eu = wy * cu;
                const f32x2 e = {el.x + el.y, eu.x + eu.y};
                const f32x2 rz = wz * e;
                finish(p, rz.x + rz.y, ss, st);
            };
            Vox cA, cB;
            prepare(true, false, s0, toff, cA, cB, cA, cB);
            f32x4 aP = gather_quad(rsrc, cA.off, 0u), aQ = gather_quad(rsrc, cA.off, sPlane), aR = aQ;
            f32x4 bP = gather_quad(rsrc, cB.off, 0u), bQ = gather_quad(rsrc, cB.off, sPlane), bR = bQ;
            auto step = [&](bool chk, int z, f32x4 &aLo, f32x4 &aUp, f32x4 &aNx, f32x4 &bLo, f32x4 &bUp, f32x4 &bNx) {
                toff += tstep;
                Vox nA, nB;
                prepare(chk, REUSE, z, toff, cA, cB, nA, nB);
                aNx = gather_quad(rsrc, nA.off, sPlane);
                bNx = gather_quad(rsrc, nB.off, sPlane);
                f32x4 al, bl;
                if (!REUSE) {
                    al = gather_quad(rsrc, nA.off, 0u);
                    bl = gather_quad(rsrc, nB.off, 0u);
                }
                accumulate(cA, aLo, aUp, ssA, stA);
                accumulate(cB, bLo, bUp, ssB, stB);
                if (REUSE) {
                    asm volatile("" : "+v"(ssA), "+v"(stA), "+v"(ssB), "+v"(stB) : : "memory");
                    if (nA.off != cA.off + sPlane) aUp = gather_quad(rsrc, nA.off, 0u);
                    if (nB.off != cB.off + sPlane) bUp = gather_quad(rsrc, nB.off, 0u);
                } else {
                    aUp = al;
                    bUp = bl;
                }
                cA = nA;
                cB = nB;
            };
            for (int z = s0 + 1; z < s1;) {
                const int ze = min(z + kSubRun, s1);
                const bool chk = !all_inside(cA, cB, ze);
                for (; z + 2 < ze; z += 3) {
                    step(chk, z, aP, aQ, aR, bP, bQ, bR);
                    step(chk, z + 1, aQ, aR, aP, bQ, bR, bP);
                    step(chk, z + 2, aR, aP, aQ, bR, bP, bQ);
                }
                for (; z < ze; ++z) {
                    step(chk, z, aP, aQ, aR, bP, bQ, bR);
                    aP = aQ; aQ = aR; bP = bQ; bQ = bR;
                }
            }
            accumulate(cA, aP, aQ, ssA, stA);
            accumulate(cB, bP, bQ, ssB, stB);
        }
    }
    // one publish per workgroup: lanes -> wave (shuffle tree) -> workgroup (waves in order) -> partial blockIdx.x
    double ss = corr_wave_sum(ssA + ssB), st = corr_wave_sum(stA + stB);
    if (lane == 0) {
        lds[wave] = ss;
        lds[kTileY + wave] = st;
    }
    __syncthreads();
    double v0 = 0.0, v1 = 0.0;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 0; w < kTileY; ++w) {
            v0 += lds[w];
            v1 += lds[kTileY + w];
        }
    }
    corr_publish_flat<kThreads>((int)blockIdx.x, v0, v1, tail, lds, &last);
}

// dst[z+1][y+1][x+1] = (src[z][y][x], src[z][y+1][x]) with edge replication; y runs over [-1, sy-1].
__global__ __launch_bounds__(256) void pad_rowpair_kernel(float2 *__restrict__ dst, const float *__restrict__ src,
                                                        int sx, int sy, int sz) {
    const int xp = blockIdx.x * 256 + threadIdx.x, yp = blockIdx.y, zp = blockIdx.z;
    if (xp >= sx + 2) return;
    const int x = min(max(xp - 1, 0), sx - 1), z = min(max(zp - 1, 0), sz - 1);
    const int y0 = min(max(yp - 1, 0), sy - 1), y1 = min(max(yp, 0), sy - 1);
    const float *pl = src + (size_t)sx * sy * z;
    dst[(size_t)xp + (size_t)(sx + 2) * ((size_t)yp + (size_t)(sy + 1) * zp)] =
        make_float2(pl[(size_t)x + (size_t)sx * y0], pl[(size_t)x + (size_t)sx * y1]);
}

// dst[z+1][y+1][x+1] = src[z][y][x]; border voxels replicate the nearest edge voxel.  Rows are `pitch` floats apart (a
// multiple of 4: 16-byte aligned rows); the 0-2 floats of padding at the end of a row repeat the edge voxel too.
__global__ __launch_bounds__(256) void pad_replicate_kernel(float *__restrict__ dst, const float *__restrict__ src,
                                                          int sx, int sy, int sz, int pitch) {
    const int xp = blockIdx.x * 256 + threadIdx.x, yp = blockIdx.y, zp = blockIdx.z;
    if (xp >= pitch) return;
    const int x = min(max(xp - 1, 0), sx - 1), y = min(max(yp - 1, 0), sy - 1), z = min(max(zp - 1, 0), sz - 1);
    dst[(size_t)xp + (size_t)pitch * ((size_t)yp + (size_t)(sy + 2) * zp)] =
        src[(size_t)x + (size_t)sx * ((size_t)y + (size_t)sy * z)];
}

}  // namespace

long long PaddedSource::plain_pitch(Dims s) { return (s.nx + 2 + 3) / 4 * 4; }

bool PaddedSource::eligible(Dims t, Dims s) {
    if (t.nx <= 0 || t.ny <= 0 || t.nz <= 0 || s.nx <= 0 || s.ny <= 0 || s.nz <= 0) return false;
    const long long xp = plain_pitch(s), yp = s.ny + 2, zp = s.nz + 2;
    // every byte offset fits 32 bits (the buffer's record count is a signed int: 2 GiB); both operands of each 24-bit
    // multiply fit 23 bits + sign; float32 holds every target index exactly
    return xp * yp * zp * 4 < (1ll << 31) && t.count() * 4 < (1ll << 32) && xp * yp * 4 < (1ll << 23) &&
           zp < (1ll << 23) && t.nx <= (1 << 24) && t.ny <= (1 << 24) && t.nz <= (1 << 24);
}

bool PaddedSource::row_pairs_fit(Dims s) {
    const long long xp = s.nx + 2, yp2 = s.ny + 1, zp = s.nz + 2;
    return xp * yp2 * zp * 8 < (1ll << 31) && xp * yp2 * 8 < (1ll << 23);
}

void PaddedSource::build(const float *src, Dims d, hipStream_t s) {
    if (d.nx <= 0 || d.ny <= 0 || d.nz <= 0) throw Error(-1, "PaddedSource: empty volume");
    if (d.ny + 2 > 65535 || d.nz + 2 > 65535) throw Error(-1, "PaddedSource: volume exceeds launch limits");
    dims = d;
    pitch = plain_pitch(d);
    // + 16 floats of slack behind the last row
    const size_t n = (size_t)pitch * (d.ny + 2) * (d.nz + 2) + 16;
    if (data.size() < n) data.alloc(n);
    dim3 grid((unsigned)div_up(pitch, 256), (unsigned)(d.ny + 2), (unsigned)(d.nz + 2));
    hipLaunchKernelGGL(pad_replicate_kernel, grid, dim3(256), 0, s, data.get(), src, (int)d.nx, (int)d.ny, (int)d.nz,
                       (int)pitch);
    MIL_HIP(hipGetLastError());
    has_pairs = row_pairs_fit(d);
    if (has_pairs) {
        const size_t n2 = (size_t)2 * (d.nx + 2) * (d.ny + 1) * (d.nz + 2);
        if (pairs.size() < n2) pairs.alloc(n2);
        dim3 grid2((unsigned)div_up(d.nx + 2, 256), (unsigned)(d.ny + 1), (unsigned)(d.nz + 2));
        hipLaunchKernelGGL(pad_rowpair_kernel, grid2, dim3(256), 0, s, reinterpret_cast<float2 *>(pairs.get()), src,
                           (int)d.nx, (int)d.ny, (int)d.nz);
        MIL_HIP(hipGetLastError());
    }
}

namespace {

// Which layout / reuse serves this matrix best (measured at 512x512x256, tools/bench_corr.py --sweep; all three give the same
// sums).  The plane reuse of corr_padded_kernel skips a gather only when EVERY lane of a wave reuses, so what counts is
// how often, per z step, some lane of a 64-wide wave crosses a source column, row or plane out of turn: the crossing rate
// of one lane (|m2|, |m6|, |m10 - 1| per step) times the number of distinct phases the wave's lanes carry
// (1 + 64 * distance of the per-x increment m0 / m4 / m8 from an integer, at most 64).
//   < 0.05  plain layout + plane reuse           (translations, pure scales, rotations up to ~1 degree)
//   < 2.0   row pairs + plane reuse              (the usual registration matrix: a few degrees AND a few per cent)
//   else    row pairs, both planes gathered      (large rotations)
std::atomic<int> g_chunk_wgs_per_cu{0};  // corr_set_chunk_wgs_per_cu (experiments); 0: kChunkWgsPerCu
std::atomic<int> g_forced_kernel{-1};  // test / profiling aid (process-wide), corr_padded_force_kernel(): 0-2 one voxel per lane, 3-5 two, 6 two + marching along y

// Expected number of wave-wide re-gathers per step when a wave of 64 consecutive x marches along z / along y (see the
// comment above): the drift of the two coordinates across the marching axis, and the index skips of the one along it,
// each times the number of distinct sub-voxel phases the wave's lanes carry for that coordinate.
void exception_rates(const float *m, double &along_z, double &along_y) {
    auto frac_dist = [](double v) { return std::fabs(v - std::nearbyint(v)); };
    auto phases = [&](double per_x) { return std::min(64.0, 1.0 + 64.0 * frac_dist(per_x)); };
    const double px = phases(m[0]), py = phases(m[4]), pz = phases(m[8]);
    along_z = std::fabs((double)m[2]) * px + std::fabs((double)m[6]) * py + std::fabs((double)m[10] - 1.0) * pz;
    along_y = std::fabs((double)m[1]) * px + std::fabs((double)m[5] - 1.0) * py + std::fabs((double)m[9]) * pz;
}

// Thresholds fitted to 337 matrices a real optimisation visits, each timed with kernels 3, 6, 5, 4 at 512 x 512 x 256
// (tools/tune_corr_pick.py, profiles/r04/reg_cost_pick.txt): the rule averages 0.1636 ms where the per-matrix best of the four
// is 0.1602 (round 3's thresholds 0.08 / 0.6 / 0.08 / 0.5, no spread test: 0.1687).
constexpr double kPlainMaxRate = 0.05;       // plain layout + plane reuse below this re-gather rate along z
constexpr double kPairsReuseMaxRate = 2.0;   // row pairs keep their plane reuse up to here
constexpr double kMarchYMaxRate = 0.05;      // kernel 6: re-gather rate along y below this, and below the rate along z,
constexpr double kMarchYMaxSpread = 6.0;     // ... and the lanes of a wave within this many source rows + planes
int pick_layout(const float *m, bool has_pairs) {
    if (!has_pairs) return 0;
    double rate, rate_y;
    exception_rates(m, rate, rate_y);
    if (!(rate >= kPlainMaxRate)) return 0;  // also NaN
    return rate < kPairsReuseMaxRate ? 2 : 1;
}

// 0 plain + reuse | 1 row pairs | 2 row pairs + reuse: one voxel per lane; 3 / 4 / 5: the same with two voxels per lane
// (needs a target width that is a multiple of 128); 6: plain + reuse, two voxels per lane, MARCHING ALONG Y.
// Measured at 512 x 512 x 256 (tools/bench_corr.py, ms per evaluation):
// the two-voxel form wins wherever the plane reuse mostly holds (identity 0.156 vs 0.171, 2 % scale 0.152 vs 0.157,
// 3 deg + 2 % 0.191 vs 0.199); with both planes gathered every step (large rotations) the one-voxel form keeps fewer
// cache lines per gather in flight and wins (45 deg: 0.500 vs 0.541).  Round 3: where the reuse would hold along y but not
// along z (rotations about y, as in bench.py: 83 % against 2 % of the wave-steps re-gather), kernel 6 runs the plain layout at
// 0.161 ms where the row pairs take 0.175 and the plain z-marching form 0.202.
int pick_kernel(const float *m, bool has_pairs, bool wide) {
    int forced = g_forced_kernel;
    if (forced >= 7 && forced <= 10) forced -= 4;  // the chunked forms of 3-6: same layouts, same per-matrix fall-backs
    if (forced == 6) return wide ? 6 : 0;
    if (forced >= 0 && forced <= 5) {
        const int layout = forced % 3;
        return ((layout == 0 || has_pairs) ? layout : 0) + ((forced >= 3 && wide) ? 3 : 0);
    }
    if (wide) {
        double rz, ry;
        exception_rates(m, rz, ry);
        // ... and only while the 64 lanes of a wave stay within a few source rows / planes: spread wider (rotations beyond
        // ~5 degrees about y or z) every gather touches a line per lane whichever axis the wave marches along, and the row
        // pairs' half instruction count wins (three matrices of the trajectory sample cost 0.56-0.70 ms with kernel 6 against
        // 0.39-0.45 with kernel 5: profiles/r03/reg_cost_pick_on_trajectory_matrices.txt, profiles/r04/reg_cost_pick.txt)
        const double spread = 64.0 * (std::fabs((double)m[4]) + std::fabs((double)m[8]));
        if (ry < kMarchYMaxRate && ry < rz && spread < kMarchYMaxSpread) return 6;
    }
    const int layout = pick_layout(m, has_pairs);
    return layout + ((wide && layout != 1) ? 3 : 0);
}

// The two-voxel kernels (3-6) run in their chunked form (corr_chunk_kernel) unless a tiled one is forced (0-6) or
// MIL_CORR_CHUNK=0 asks for the tiled kernels everywhere (A/B measurements; the tiled forms return the bits of kernels 0-2).
bool use_chunks() {
    const int forced = g_forced_kernel;
    if (forced >= 7 && forced <= 10) return true;
    if (forced >= 0) return false;
    static const bool off = getenv("MIL_CORR_CHUNK") && getenv("MIL_CORR_CHUNK")[0] == '0';
    return !off;
}

// Workgroups of a chunked launch: a CONSTANT per kernel -- kChunkWgsPerCu x 256, what an MI355X (256 compute units) holds at
// once when the registers allow as many -- but no more than one per kMinChunk steps.  The grid fixes how many partial sums
// there are and so the rounding of the two sums: it must not depend on the device a host thread happens to be on, on an
// occupancy query, or on the compiler's register allocation (round 5 derived it from both; advisor).  A part with fewer
// compute units, or a build that fits fewer workgroups per unit, runs the same pieces in more than one wave of workgroups:
// the pieces are independent, nothing waits for a co-resident workgroup.  corr_set_chunk_wgs_per_cu overrides the count
// (experiments); MIL_CORR_CHUNK=0 is the bit-reference path (the tiled kernels return the bits of kernels 0-2).
constexpr unsigned kMinChunk = 24;
constexpr unsigned kChunkCus = 256;
// workgroups (of four waves) per compute unit, by kernel 3 .. 6 (ms per evaluation at 512 x 512 x 256, identity / 3 deg about
// y + 2 %, profiles/r05/reg_cost_chunked_ab.txt: kernel 3 with 4 | 5 | 6: 0.146 | 0.146 | 0.142; kernel 5: 0.156 | 0.169 | 0.163
// at the identity, 0.176 | 0.183 | 0.180; kernel 6: 0.167 | 0.1645 | 0.170 at 3 deg + 2 %).  Round 6: the plain-layout kernels
// (3 and 6) run TWELVE workgroups per compute unit -- twice what is resident, pieces of ~43 steps -- so that the dispatcher
// hands the second half of the pieces to whichever slots free first: what a piece costs depends on where it lies (rim,
// re-gathers, what its neighbours left in L2), and with one piece per resident workgroup the evaluation ends with the slowest
// one.  profiles/r06/reg_cost_chunk_grid.txt: kernel 3 at the identity 0.1289 (6) -> 0.1233 (12), kernel 6 at 3 deg + 2 %
// 0.1484 (5) -> 0.1400 (12), at 10 deg 0.205 -> 0.192; the row-pair kernel (5) is best at 4 - 5 and stays.  Static pieces:
// the partial sums and their order are fixed by the grid, as before.
constexpr int kChunkWgsPerCu[4] = {12, 0, 4, 12};
unsigned chunk_grid(int which, unsigned long long total_steps) {
    int per_cu = kChunkWgsPerCu[which & 3];
    if (const int forced = g_chunk_wgs_per_cu.load(std::memory_order_relaxed)) per_cu = forced;  // experiments
    const unsigned long long want = (unsigned long long)kChunkCus * (unsigned)std::max(per_cu, 1);
    const unsigned long long most = (total_steps + kMinChunk - 1) / kMinChunk;
    return (unsigned)std::max(1ull, std::min(want, most));
}

// The two-voxel kernels form their indices from the bit patterns of floor + 1.5 * 2^23 (corr_pair_kernel, prepare): the
// 24-bit multiplies then see 2^22 + f, the shift sees 0x4B400000 + f.  What that adds to the sum is a constant; it goes into
// the origin here, modulo 2^32.
int magic_origin(int origin4, int row4, int plane4, int shift) {
    const unsigned bias24 = 0x400000u, bits = 0x4B400000u;
    const unsigned k = bias24 * (unsigned)plane4 + bias24 * (unsigned)row4 + (bits << shift);
    return (int)((unsigned)origin4 - k);
}

}  // namespace

void corr_padded_force_kernel(int kernel) { g_forced_kernel = kernel; }
void corr_set_chunk_wgs_per_cu(int n) { g_chunk_wgs_per_cu = n > 0 ? n : 0; }

int corr_padded_pick_kernel(const Affine12 &m, bool wide) {
    return wide ? pick_kernel(m.m, true, true) : pick_layout(m.m, true);
}

void corr_sums_padded_async(const float *target, const PaddedSource &src, Dims t, const Affine12 &m, CorrWorkspace &ws,
                            hipStream_t s) {
    if (!PaddedSource::eligible(t, src.dims) || !src.data.get())
        throw Error(-1, "corr_sums_padded: volume outside the 32-bit offset range");
    PaddedArgs a;
    for (int i = 0; i < 12; ++i) a.m[i] = m.m[i];
    a.tx = (int)t.nx; a.ty = (int)t.ny; a.tz = (int)t.nz;
    a.sxf = (float)src.dims.nx; a.syf = (float)src.dims.ny; a.szf = (float)src.dims.nz;
    const long long xp = src.dims.nx + 2, yp = src.dims.ny + 2, yp2 = src.dims.ny + 1, zp = src.dims.nz + 2;
    dim3 grid((a.tx + kTileX - 1) / kTileX, (a.ty + kTileY - 1) / kTileY, (a.tz + kStepsZ - 1) / kStepsZ);
    if (grid.y > 65535 || grid.z > 65535) throw Error(-1, "corr_sums: volume exceeds launch limits");
    // two voxels per lane (x and x + 64): whole 128-wide workgroups only
    // ... and source extents the bit-pattern index arithmetic of corr_pair_kernel holds exactly (|index| < 2^22)
    const bool in_range = xp < (1 << 22) && yp < (1 << 22) && zp < (1 << 22);
    const bool wide_exact = a.tx % (2 * kTileX) == 0 && in_range;
    // ... or, for the chunked kernels only, ragged 128-wide columns while they waste at most 15 % of their lanes (500 -> 512:
    // 2.4 %; 320 -> 384 would be 20 %: five one-voxel tiles of 64 serve it without waste)
    const long long up128 = (a.tx + 2 * kTileX - 1) / (2 * kTileX) * (2 * kTileX);
    const bool wide_ragged = !wide_exact && in_range && a.tx >= 2 * kTileX && up128 * 100 <= 115ll * a.tx && use_chunks();
    int kernel = pick_kernel(a.m, src.has_pairs, wide_exact || wide_ragged);
    if (wide_ragged && kernel == 4) kernel = 1;  // row pairs without reuse: the two-voxel form exists tiled only
    const int layout = kernel == 6 ? 0 : kernel % 3;
    if (kernel == 6) {  // marching along y: tiles of 128 (x) x 16 (y) x 4 (z)
        grid.y = (unsigned)((a.tz + kTileY - 1) / kTileY);
        grid.z = (unsigned)((a.ty + kStepsZ - 1) / kStepsZ);
        if (grid.y > 65535 || grid.z > 65535) throw Error(-1, "corr_sums: volume exceeds launch limits");
    }
    const long long pitch = src.pitch;
    if (layout != 0) {  // float2 volume [sz+2][sy+1][sx+2]
        a.row4 = (int)(8 * xp);
        a.plane4 = (int)(8 * xp * yp2);
        a.origin4 = (int)(8 * ((yp2 + 1) * xp + 1));
        a.bytes = (unsigned)(8 * xp * yp2 * zp);
        a.origin4m = magic_origin(a.origin4, a.row4, a.plane4, 3);
    } else {            // float volume [sz+2][sy+2][pitch]
        a.row4 = (int)(4 * pitch);
        a.plane4 = (int)(4 * pitch * yp);
        a.origin4 = (int)(4 * (pitch * yp + pitch + 1));
        a.bytes = (unsigned)(4 * pitch * yp * zp);
        a.origin4m = magic_origin(a.origin4, a.row4, a.plane4, 2);
    }
    const float *volume = layout != 0 ? src.pairs.get() : src.data.get();
    // (kernel 4 -- row pairs, both planes gathered every step -- needs 130 registers in this form and runs 4 x slower: tiled)
    if (kernel >= 3 && kernel != 4 && use_chunks()) {
        // columns of 128 (x) x 4 (the waves: y, or z when marching along y), walked along the marching axis
        const int run_len = kernel == 6 ? a.ty : a.tz;
        a.gx = (a.tx + 2 * kTileX - 1) / (2 * kTileX);
        a.gy = ((kernel == 6 ? a.tz : a.ty) + kTileY - 1) / kTileY;
        const unsigned long long total = (unsigned long long)a.gx * a.gy * run_len;
        if (total > 0xffffffffull) throw Error(-1, "corr_sums: volume exceeds launch limits");
        auto launch = [&](auto kern) {
            const unsigned wgs = chunk_grid(kernel - 3, total);
            const CorrTail tail = ws.next(wgs, s);
            hipLaunchKernelGGL(kern, dim3(wgs), dim3(kThreads), 0, s, target, volume, a, tail, (unsigned)total, run_len);
        };
        if (wide_ragged) {
            switch (kernel) {
                case 3: launch(corr_chunk_kernel<false, true, false, true>); break;
                case 5: launch(corr_chunk_kernel<true, true, false, true>); break;
                default: launch(corr_chunk_kernel<false, true, true, true>); break;
            }
        } else {
            switch (kernel) {
                case 3: launch(corr_chunk_kernel<false, true, false>); break;
                case 5: launch(corr_chunk_kernel<true, true, false>); break;
                default: launch(corr_chunk_kernel<false, true, true>); break;
            }
        }
        MIL_HIP(hipGetLastError());
        return;
    }
    // the partials are those of the 64 x 4 x 16 decomposition whichever z-marching kernel runs (64 x 16 x 4 for kernel 6)
    const CorrTail tail = ws.next((size_t)grid.x * grid.y * grid.z, s);
    const int old_gx = (int)grid.x;
    if (kernel >= 3) grid.x = (grid.x + 1) / 2;  // 128 x per workgroup
    a.gx = (int)grid.x;
    a.gy = (int)grid.y;
    const size_t n_wg = (size_t)grid.x * grid.y * grid.z;
    if (n_wg > 0x7fffffffu) throw Error(-1, "corr_sums: volume exceeds launch limits");
    grid = dim3((unsigned)n_wg);  // one-dimensional, decoded by corr_tile_of_block
    switch (kernel) {
        case 0: hipLaunchKernelGGL(corr_padded_kernel, grid, dim3(kThreads), 0, s, target, volume, a, tail); break;
        case 1: hipLaunchKernelGGL(corr_rowpair_kernel<false>, grid, dim3(kThreads), 0, s, target, volume, a, tail); break;
        case 2: hipLaunchKernelGGL(corr_rowpair_kernel<true>, grid, dim3(kThreads), 0, s, target, volume, a, tail); break;
        case 3: hipLaunchKernelGGL((corr_pair_kernel<false, true>), grid, dim3(kThreads), 0, s, target, volume, a, tail, old_gx); break;
        case 4: hipLaunchKernelGGL((corr_pair_kernel<true, false>), grid, dim3(kThreads), 0, s, target, volume, a, tail, old_gx); break;
        case 5: hipLaunchKernelGGL((corr_pair_kernel<true, true>), grid, dim3(kThreads), 0, s, target, volume, a, tail, old_gx); break;
        default: hipLaunchKernelGGL((corr_pair_kernel<false, true, true>), grid, dim3(kThreads), 0, s, target, volume, a, tail, old_gx); break;
    }
    MIL_HIP(hipGetLastError());
}

void corr_sums_padded_sync(const float *target, const PaddedSource &src, Dims t, const Affine12 &m, CorrWorkspace &ws,
                           hipStream_t s, double *ss, double *st) {
    corr_sums_padded_async(target, src, t, m, ws, s);
    ws.wait(s);
    *ss = ws.host[0];
    *st = ws.host[1];
}

}  // namespace k
}  // namespace mil

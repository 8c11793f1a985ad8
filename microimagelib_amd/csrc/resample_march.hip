// affineTransform (reference include/cukernel.cuh:501-524) as a MARCHING kernel, and the fused resample + z projection of
// mip3dgpu (src/apifunc.cpp:610-613) with the taps a rotation series does not need left out.
//
// resample_kernel (affine.hip) gives every output voxel its own thread: four 8-byte gathers, no reuse, 262 144 workgroups
// of one voxel per thread at 512 x 512 x 256 -- 0.25 ms, a quarter of the 2 R roofline, bound by the lines its gathers touch
// and by launch overhead.  Here a lane owns two output voxels (x, x + 64) and marches along z or y like the registration's
// cost kernels (corr_padded.hip): the upper source plane (row) of one step is the lower plane (row) of the next, decided by
// offset equality, so a step costs two gathers per voxel instead of four; the grid is what the chip holds at once and a
// workgroup walks a contiguous piece of the (column, step) sequence.  The arithmetic is fetch_linear's -- the same float32
// operations in the same order -- so the output is the one-voxel kernel's bit for bit (tests/test_gpu_kernels.py).
//
// The source here is NOT border-padded (one resample does not pay for a padded copy), so the edge clamp of the reference's
// texture fetch is real work -- but only on the rim: where every lane of a wave has all eight taps strictly inside the
// volume (floor(t - 0.5) in [0, n - 2] on each axis; the coordinates are monotone along the march, so both ends of a piece
// decide it) the clamps are identities and the loop runs without them, without window tests and with plain 32-bit offsets.
// Pieces that touch the rim or leave the window go through fetch_linear_small voxel by voxel.
#include "kernels.h"
#include "trilinear_dev.h"

#include <algorithm>
#include <cmath>
#include <type_traits>
#include <atomic>
#include <cstdlib>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <utility>
#include <vector>

namespace mil {
namespace k {

namespace {

using namespace tri;

constexpr int kLanes = 64;    // one wave = 64 consecutive x
constexpr int kWaves = 4;     // waves per workgroup: four neighbours across the marching axis
constexpr int kThreads = kLanes * kWaves;
#ifndef MARCH_PIECE
#define MARCH_PIECE 32
#endif
constexpr int kPiece = MARCH_PIECE;    // steps of one column a workgroup draws at a time; off the rim or not is decided per piece
constexpr unsigned kMinChunk = 32;
constexpr int kWgsPerCu = 10;

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

struct MarchArgs {
    float m[12];
    int tx, ty, tz;       // output extents
    int sx, sy, sz;       // source extents
    float bx1, by1, bz1;  // s - 1 as float: interior <=> 0 <= t - 0.5 < s - 1
    int row4, plane4;     // source pitches in bytes
    unsigned bytes;       // size of the source
    int gx, gy;           // columns along x (128 wide) and across the marching axis (4 deep)
};

__device__ __forceinline__ f32x2 gather_pair(__amdgpu_buffer_rsrc_t rsrc, unsigned voffset, unsigned soffset) {
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)voffset, (int)soffset, 0);
    return __builtin_bit_cast(f32x2, v);
}

// MARCHY: march along y inside one output plane (tiles of 128 x 4 z), reusing source ROWS -- the quiet axis of a rotation
// about y; otherwise along z, reusing source planes.  STRICT: the 2-D variant's window 0 < t (affineTransform2Dkernel).
template <bool MARCHY, bool STRICT>
__global__ __launch_bounds__(kThreads) void resample_march_kernel(float *__restrict__ out, const float *__restrict__ src,
                                                                MarchArgs a, unsigned *__restrict__ ticket,
                                                                unsigned *__restrict__ next_ticket, int run_len) {
    __shared__ unsigned next_piece;
    const int lane = threadIdx.x & (kLanes - 1), wave = threadIdx.x / kLanes;
    const f32x2 half2 = {0.5f, 0.5f};
    const __amdgpu_buffer_rsrc_t rsrc = small_rsrc(src, a.sx, a.sy, a.sz);
    const unsigned sRow = (unsigned)a.row4, sPlane = (unsigned)a.plane4, sBoth = (unsigned)(a.plane4 + a.row4);
    const unsigned sStep = MARCHY ? sRow : sPlane, sSide = MARCHY ? sPlane : sRow;
    const unsigned ostep = MARCHY ? 4u * (unsigned)a.tx : 4u * (unsigned)a.tx * (unsigned)a.ty;
    char *obase = reinterpret_cast<char *>(out);
    const float sxf = (float)a.sx, syf = (float)a.sy, szf = (float)a.sz;
    // Pieces of kPiece steps of one column, handed out by an atomic ticket: what a piece costs depends on where it lies (off
    // the rim: two gathers per voxel; on it: the one-voxel code; outside the window: nothing), and equal shares of STEPS left
    // the workgroups that drew the volume's last x column -- on the rim for every near-identity matrix -- running twice as
    // long as the rest (0.215 ms where the average piece predicts 0.15).  The next ticket is drawn before the current piece
    // is worked on, so its round trip is hidden.
    const unsigned pieces_per_run = (unsigned)((run_len + kPiece - 1) / kPiece);
    const unsigned n_pieces = pieces_per_run * (unsigned)a.gx * (unsigned)a.gy;
    // A workgroup's FIRST piece is its own number, the others come from the counter (ticket t = piece gridDim.x + t): one atomic
    // per piece on a single address is what the pieces cost beyond their work -- ~5 ns each, serialised (with a second atomic
    // per workgroup to install an epoch in a 64-bit word, 2 560 more of them, the identity resample ran 0.158 instead of
    // 0.146 ms: profiles/r06/resample_ticket.txt).  The counter of THIS launch is zero when it starts -- the previous launch of
    // the stream zeroed it while it ran on the other word of the pair (ticket_of, below) -- and this launch zeroes the next one's.
    auto draw = [&] { return gridDim.x + __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(next_ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned mine = blockIdx.x;
    for (;;) {
        if (threadIdx.x == 0) next_piece = mine;
        __syncthreads();
        const unsigned piece = next_piece;
        __syncthreads();  // everybody has read it before thread 0 writes the next one
        if (piece >= n_pieces) break;
        if (threadIdx.x == 0) mine = draw();
        const unsigned col = piece / pieces_per_run;
        const int s0 = (int)(piece - col * pieces_per_run) * kPiece;
        const int s1 = min(run_len, s0 + kPiece);
        const int bx = (int)(col % (unsigned)a.gx), by = (int)(col / (unsigned)a.gx);
        const int x = bx * 2 * kLanes + lane;  // voxel A; voxel B sits 64 further on
        const int fixed = by * kWaves + wave;  // this wave's coordinate across the marching axis
        const bool live = fixed < (MARCHY ? a.tz : a.ty);  // wave-uniform; a dead wave still meets the others at the barriers
        const bool okA = x < a.tx, okB = x + kLanes < a.tx;
        const float ifix = (float)fixed;
        // a lane beyond the output's width follows the last voxel of the row: it takes part in the rim decision with that
        // voxel's coordinates, so its offsets are as valid as everybody's, and stores nothing
        const f32x2 ixs = {(float)min(x, a.tx - 1), (float)min(x + kLanes, a.tx - 1)};
        // t = ((m0*x + m1*y) + m2*z) + m3, then + 0.5: coord() of affine.hip for both voxels at once
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
        auto out_offset = [&](int step) {
            return MARCHY ? 4u * ((unsigned)x + (unsigned)a.tx * ((unsigned)step + (unsigned)a.ty * (unsigned)fixed))
                          : 4u * ((unsigned)x + (unsigned)a.tx * ((unsigned)fixed + (unsigned)a.ty * (unsigned)step));
        };
        // all eight taps of a voxel strictly inside the source: floor(t - 0.5) in [0, n - 2] on every axis.  XFREE: on the
        // y and z axes only -- what the edge-tolerant loop needs (it clamps and window-tests x itself)
        auto interior = [&](bool xfree, float tx_, float ty_, float tz_) {
            const float bx_ = tx_ - 0.5f, by_ = ty_ - 0.5f, bz_ = tz_ - 0.5f;
            return (xfree || (bx_ >= 0 && bx_ < a.bx1)) && by_ >= 0 && by_ < a.by1 && bz_ >= 0 && bz_ < a.bz1;
        };
        // 2: every lane of the wave interior at both ends of [z, ze) -- interior throughout, the coordinates are monotone
        // along the march; 1: on y and z only; 0: neither
        auto rim_class = [&](int z, int ze) {
            f32x2 TX, TY, TZ, UX, UY, UZ;
            coords(z, TX, TY, TZ);
            coords(ze - 1, UX, UY, UZ);
            const bool yz = interior(true, TX.x, TY.x, TZ.x) && interior(true, TX.y, TY.y, TZ.y) &&
                            interior(true, UX.x, UY.x, UZ.x) && interior(true, UX.y, UY.y, UZ.y);
            if (!__all(yz)) return 0;
            const bool all = interior(false, TX.x, TY.x, TZ.x) && interior(false, TX.y, TY.y, TZ.y) &&
                             interior(false, UX.x, UY.x, UZ.x) && interior(false, UX.y, UY.y, UZ.y);
            return __all(all) ? 2 : 1;
        };

        // ---- steps [z, ze) off the rim: no clamps, plane / row reuse ----------------------------------------------------
        // EDGE = false: every tap of every lane lies inside the source -- no window test either.  EDGE = true: that holds on y
        // and z only; along x a lane may sit on the first / last column (the clamp folds both taps onto the edge voxel) or
        // outside the window (value 0).  For near-identity matrices the LAST x column of the output is such a lane in every
        // row, i.e. a quarter of all waves at 512 wide: sending those through the one-voxel code made them the critical
        // path.  The pair is then gathered at b = clamp(i, 0, sx - 2) like fetch_linear does, and one select per tap puts the
        // edge voxel on both; the reuse test compares the UNclamped offsets, so equal offsets still mean equal taps.
        struct Vox {
            float wx1, wy1, wz1;
            unsigned off;   // unclamped: what the reuse test compares
            unsigned load;  // EDGE: offset of the clamped pair (off, off + 4 on the first column, off - 4 on the last)
            bool lo_edge, hi_edge, in_x;
        };
        struct Taps { f32x2 r0, r1; };  // the two (x0, x1) pairs of one step: rows y0, y1 of a plane / planes z0, z1 of a row
        auto fast = [&](auto edge_tag, int z, const int ze) {
            constexpr bool EDGE = decltype(edge_tag)::value;
            unsigned ooff = out_offset(z);
            auto prepare = [&](int step, Vox &A, Vox &B) {
                f32x2 TX, TY, TZ;
                coords(step, TX, TY, TZ);
                const f32x2 BX = TX - half2, BY = TY - half2, BZ = TZ - half2;
                float fxA = floorf(BX.x), fxB = floorf(BX.y);
                const float fyA = floorf(BY.x), fzA = floorf(BZ.x), fyB = floorf(BY.y), fzB = floorf(BZ.y);
                A.wx1 = BX.x - fxA; A.wy1 = BY.x - fyA; A.wz1 = BZ.x - fzA;
                B.wx1 = BX.y - fxB; B.wy1 = BY.y - fyB; B.wz1 = BZ.y - fzB;
                if (EDGE) {
                    A.in_x = (STRICT ? TX.x > 0 : TX.x >= 0) && TX.x < sxf;
                    B.in_x = (STRICT ? TX.y > 0 : TX.y >= 0) && TX.y < sxf;
                    // a lane outside the window computes with the nearest edge column: every load stays inside the source
                    fxA = __builtin_amdgcn_fmed3f(fxA, -1.0f, a.bx1);
                    fxB = __builtin_amdgcn_fmed3f(fxB, -1.0f, a.bx1);
                    A.lo_edge = fxA == -1.0f; A.hi_edge = fxA == a.bx1;
                    B.lo_edge = fxB == -1.0f; B.hi_edge = fxB == a.bx1;
                }
                A.off = __umul24((unsigned)(int)fzA, sPlane) + __umul24((unsigned)(int)fyA, sRow) + ((unsigned)(int)fxA << 2);
                B.off = __umul24((unsigned)(int)fzB, sPlane) + __umul24((unsigned)(int)fyB, sRow) + ((unsigned)(int)fxB << 2);
                if (EDGE) {
                    A.load = A.off + (A.lo_edge ? 4u : 0u) - (A.hi_edge ? 4u : 0u);
                    B.load = B.off + (B.lo_edge ? 4u : 0u) - (B.hi_edge ? 4u : 0u);
                } else {
                    A.load = A.off;
                    B.load = B.off;
                }
            };
            auto fix = [&](const Vox &p, f32x2 r) {  // (x0, x1) of a row from the pair gathered at the clamped column
                if (!EDGE) return r;
                return (f32x2){p.hi_edge ? r.y : r.x, p.lo_edge ? r.x : r.y};
            };
            auto plane = [&](const Vox &p, bool upper) {
                return Taps{fix(p, gather_pair(rsrc, p.load, upper ? sStep : 0u)), fix(p, gather_pair(rsrc, p.load, upper ? sBoth : sSide))};
            };
            // fetch_linear's order: x inside each row, then y, then z
            auto value = [&](const Vox &p, const Taps &lo, const Taps &up) {
                const f32x2 lo0 = lo.r0, lo1 = MARCHY ? up.r0 : lo.r1, up0 = MARCHY ? lo.r1 : up.r0, up1 = up.r1;
                const f32x2 wx = {1.0f - p.wx1, p.wx1}, wy = {1.0f - p.wy1, p.wy1}, wz = {1.0f - p.wz1, p.wz1};
                const f32x2 p00 = wx * lo0, p01 = wx * lo1, p10 = wx * up0, p11 = wx * up1;
                const f32x2 lower = {p00.x + p00.y, p01.x + p01.y}, upper = {p10.x + p10.y, p11.x + p11.y};
                const f32x2 el = wy * lower, eu = wy * upper;
                const f32x2 e = {el.x + el.y, eu.x + eu.y};
                const f32x2 rz = wz * e;
                const float v = rz.x + rz.y;
                return EDGE ? __uint_as_float(p.in_x ? __float_as_uint(v) : 0u) : v;
            };
            // non-temporal stores: the output is written once and not read by this kernel; without the hint the 2 x 256 bytes
            // a wave stores per step cost 0.20 ms per volume instead of 0.18 (no stores at all: 0.15; profiles/r05/resample_ab.txt)
            auto store = [&](float va, float vb) {
                if (okA) __builtin_nontemporal_store(va, reinterpret_cast<float *>(obase + ooff));
                if (okB) __builtin_nontemporal_store(vb, reinterpret_cast<float *>(obase + ooff + 4u * kLanes));
                ooff += ostep;
            };
            Vox cA, cB;
            prepare(z, cA, cB);
            Taps aP = plane(cA, false), aQ = plane(cA, true), aR = aQ;
            Taps bP = plane(cB, false), bQ = plane(cB, true), bR = bQ;
            // (lo, up, nx): three register sets whose roles rotate, so the reuse is a renaming (corr_pair_body)
            auto step = [&](int zn, Taps &aLo, Taps &aUp, Taps &aNx, Taps &bLo, Taps &bUp, Taps &bNx) {
                Vox nA, nB;
                prepare(zn, nA, nB);
                aNx = plane(nA, true);
                bNx = plane(nB, true);
                const float va = value(cA, aLo, aUp), vb = value(cB, bLo, bUp);
                store(va, vb);
                if (nA.off != cA.off + sStep) aUp = plane(nA, false);
                if (nB.off != cB.off + sStep) bUp = plane(nB, false);
                cA = nA;
                cB = nB;
            };
            ++z;
            for (; z + 2 < ze; z += 3) {
                step(z, aP, aQ, aR, bP, bQ, bR);      // afterwards: lower = Q, upper = R
                step(z + 1, aQ, aR, aP, bQ, bR, bP);  // lower = R, upper = P
                step(z + 2, aR, aP, aQ, bR, bP, bQ);  // lower = P, upper = Q
            }
            for (; z < ze; ++z) {
                step(z, aP, aQ, aR, bP, bQ, bR);
                aP = aQ; aQ = aR; bP = bQ; bQ = bR;
            }
            store(value(cA, aP, aQ), value(cB, bP, bQ));
        };
        // ---- steps [z, ze) on the y / z rim or outside the window: the one-voxel kernel's arithmetic, voxel by voxel ---------
        // Branch-free, so that the loads of several steps are in flight together: a voxel outside the window fetches the
        // taps of t = (0.5, 0.5, 0.5) and discards them.
        auto slow = [&](int z, const int ze) {
            unsigned ooff = out_offset(z);
#pragma unroll 2
            for (; z < ze; ++z, ooff += ostep) {
                f32x2 TX, TY, TZ;
                coords(z, TX, TY, TZ);
                auto one = [&](float tx_, float ty_, float tz_) {
                    const bool lower = STRICT ? (tx_ > 0 && ty_ > 0 && tz_ > 0) : (tx_ >= 0 && ty_ >= 0 && tz_ >= 0);
                    const bool in = lower && tx_ < sxf && ty_ < syf && tz_ < szf;
                    const float v = fetch_linear_small(rsrc, a.sx, a.sy, a.sz, in ? tx_ : 0.5f, in ? ty_ : 0.5f, in ? tz_ : 0.5f, 0.0f);
                    return __uint_as_float(in ? __float_as_uint(v) : 0u);
                };
                const float va = one(TX.x, TY.x, TZ.x), vb = one(TX.y, TY.y, TZ.y);
                if (okA) __builtin_nontemporal_store(va, reinterpret_cast<float *>(obase + ooff));
                if (okB) __builtin_nontemporal_store(vb, reinterpret_cast<float *>(obase + ooff + 4u * kLanes));
            }
        };
        // a stretch that is not off the rim as a whole is halved until it is (or is down to 8 steps): the steps next to a
        // face of the source (the last z of a near-identity matrix, say) then cost one slow stretch of 8, not the piece
        if (live) {
            for (int z = s0; z < s1;) {
                int len = s1 - z;
                int cls = len >= 4 ? rim_class(z, z + len) : 0;
                while (cls == 0 && len > 8) {
                    len = (len + 1) / 2;
                    cls = rim_class(z, z + len);
                }
                if (cls == 2) fast(std::false_type{}, z, z + len);
                else if (cls == 1) fast(std::true_type{}, z, z + len);
                else slow(z, z + len);
                z += len;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// mip3dgpu's inner pair -- affineTransform into the rotated box, then the maximum along the box's z (src/apifunc.cpp:610-613,
// include/cukernel.cuh:501-524, 396-418) -- in one pass, second version.  What the first (resample_mip_kernel, affine.hip)
// leaves on the table:
//   * a rotation series turns about x or about y, and rot2matrix (src/api_subfunc.cu:626-713) leaves the row of that axis as
//     (1, 0, 0, k) resp. (0, 1, 0, k) with an integer k: t = index + k + 0.5 EXACTLY, so the upper weight of that axis is
//     exactly 0 and the lower exactly 1 -- w0 * q0 + w1 * q1 = q0 for finite data -- and the four taps at index + 1 need not
//     be read at all: 4 instead of 8 taps per sample.  SKIP = 1 (x) / 2 (y); the host checks the row bit by bit;
//   * the rotated box is 2.5 x the volume: 60 % of its samples fall outside the window.  A wave now clips its z range to
//     a conservative union of its lanes' entry / exit steps (estimated in float with a margin; the exact window test stays
//     in the loop, so the bounds only have to be generous) instead of testing every sample of the box.
// Same samples, same float32 arithmetic on them, and max is exact and order-independent: the plane equals the two-kernel
// sequence bit for bit on finite data.  PRECONDITION (as before): finite, NaN-free volumes -- the reference's sequential
// `a = a > v ? a : v` lets a NaN wipe the running maximum and survive only as the last sample of a column; a segmented
// maximum cannot reproduce that order dependence, and with SKIP a non-finite value under a zero weight no longer turns the
// sample into NaN.
struct MipArgs {
    float m[12];
    int tx, ty, tz;  // rotated box
    int sx, sy, sz;  // source
    int segments;
};

template <int LX, int SKIP>
__global__ __launch_bounds__(kThreads) void resample_mip2_kernel(int *__restrict__ out, const float *__restrict__ src, MipArgs a) {
    constexpr int LZ = 64 / LX;
    const int lane = threadIdx.x & 63;
    const int x = blockIdx.x * LX + (lane % LX);
    const int y = blockIdx.y * kWaves + (threadIdx.x >> 6);
    if (y >= a.ty) return;  // whole wave
    const int per = (a.tz + a.segments - 1) / a.segments;
    int z0 = blockIdx.z * per, z1 = min(a.tz, z0 + per);
    const float ix = (float)min(x, a.tx - 1), iy = (float)y;  // lanes beyond the plane follow its last column, store nothing
    const float sxf = (float)a.sx, syf = (float)a.sy, szf = (float)a.sz;
    // ---- conservative z range of this wave: where may ANY of its lanes be inside the window? ---------------------------
    {
        float lo = (float)z0, hi = (float)z1;  // the lane's range [lo, hi) in float
        const float ext[3] = {sxf, syf, szf};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float s = a.m[4 * k + 2];
            const float t0 = a.m[4 * k] * ix + a.m[4 * k + 1] * iy + a.m[4 * k + 3] + 0.5f;  // an estimate: the margin below covers the rounding
            if (fabsf(s) >= 0.01f) {
                const float za = (0.0f - t0) / s, zb = (ext[k] - t0) / s;
                lo = fmaxf(lo, fminf(za, zb) - 2.0f);
                hi = fminf(hi, fmaxf(za, zb) + 2.0f);
            } else if (t0 < -1.0f - 0.01f * (float)a.tz || t0 > ext[k] + 1.0f + 0.01f * (float)a.tz) {
                hi = lo;  // (nearly) constant along z and clearly outside: nothing to do for this lane
            }
        }
        if (!(lo < hi)) {  // also NaN
            lo = (float)z1;
            hi = (float)z1;
        }
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            lo = fminf(lo, __shfl_xor(lo, off, 64));
            hi = fmaxf(hi, __shfl_xor(hi, off, 64));
        }
        z0 = max(z0, (int)floorf(lo));
        z1 = min(z1, (int)ceilf(hi));
    }
    const __amdgpu_buffer_rsrc_t rsrc = small_rsrc(src, a.sx, a.sy, a.sz);
    float best = 0.0f;
    // t = ((m0*x + m1*y) + m2*z) + m3 + 0.5 (coord() of affine.hip); the x / y part does not change along z
    const float hx = a.m[0] * ix + a.m[1] * iy, hy = a.m[4] * ix + a.m[5] * iy, hz = a.m[8] * ix + a.m[9] * iy;
#pragma unroll 2
    for (int z = z0 + lane / LX; z < z1; z += LZ) {
        const float iz = (float)z;
        const float tx = ((hx + a.m[2] * iz) + a.m[3]) + 0.5f;
        const float ty = ((hy + a.m[6] * iz) + a.m[7]) + 0.5f;
        const float tz = ((hz + a.m[10] * iz) + a.m[11]) + 0.5f;
        float v = 0.0f;
        if (tx >= 0 && ty >= 0 && tz >= 0 && tx < sxf && ty < syf && tz < szf) {
            if constexpr (SKIP == 0) {
                v = fetch_linear_small(rsrc, a.sx, a.sy, a.sz, tx, ty, tz, 0.0f);
            } else {
                const Tap X = make_tap<false>(tx, a.sx, 0.0f), Y = make_tap<false>(ty, a.sy, 0.0f), Z = make_tap<false>(tz, a.sz, 0.0f);
                const unsigned plane = __umul24((unsigned)a.sx, (unsigned)a.sy);
                const unsigned z0o = __umul24((unsigned)Z.i0, plane), z1o = __umul24((unsigned)Z.i1, plane);
                if constexpr (SKIP == 1) {
                    // X.w0 == 1, X.w1 == 0: each row contributes its voxel at X.i0 alone
                    const unsigned y0o = __umul24((unsigned)Y.i0, (unsigned)a.sx) + (unsigned)X.i0;
                    const unsigned y1o = __umul24((unsigned)Y.i1, (unsigned)a.sx) + (unsigned)X.i0;
                    auto tap = [&](unsigned o) { return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)(o << 2), 0, 0)); };
                    const float c00 = tap(z0o + y0o), c01 = tap(z0o + y1o), c10 = tap(z1o + y0o), c11 = tap(z1o + y1o);
                    const float c0 = Y.w0 * c00 + Y.w1 * c01;
                    const float c1 = Y.w0 * c10 + Y.w1 * c11;
                    v = Z.w0 * c0 + Z.w1 * c1;
                } else {
                    // Y.w0 == 1, Y.w1 == 0: each plane contributes its row Y.i0 alone
                    const int b = X.i0 < a.sx - 2 ? X.i0 : a.sx - 2;
                    const bool lo0 = X.i0 == b, lo1 = X.i1 == b;
                    const unsigned y0o = __umul24((unsigned)Y.i0, (unsigned)a.sx) + (unsigned)b;
                    auto pair = [&](unsigned o) {
                        const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)(o << 2), 0, 0);
                        return Pair{__uint_as_float(w.x), __uint_as_float(w.y)};
                    };
                    const Pair q00 = pair(z0o + y0o), q10 = pair(z1o + y0o);
                    const float c0 = X.w0 * (lo0 ? q00.lo : q00.hi) + X.w1 * (lo1 ? q00.lo : q00.hi);
                    const float c1 = X.w0 * (lo0 ? q10.lo : q10.hi) + X.w1 * (lo1 ? q10.lo : q10.hi);
                    v = Z.w0 * c0 + Z.w1 * c1;
                }
            }
        }
        best = (best > v) ? best : v;
    }
#pragma unroll
    for (int off = LX; off < 64; off <<= 1) {
        const float o = __shfl_xor(best, off, 64);
        best = (best > o) ? best : o;
    }
    if (lane < LX && x < a.tx && best > 0.0f) atomicMax(out + (size_t)x + (size_t)a.tx * y, __float_as_int(best));
}

// Rotation about x (mip3dgpu, rAxis 1): rot2matrix leaves x alone -- row 0 is (1, 0, 0, k) -- and the other two rows do not
// look at x (m4 = m8 = 0): the 64 lanes of a wave (64 consecutive x, one y) share ty, tz, the y / z taps, their weights and
// their row / plane offsets; only the column differs, and it is the same for every z of the ray.  resample_mip2_kernel<64, 1>
// let every lane redo that arithmetic (~55 vector instructions per sample, 0.11 ms per angle with 16-byte-coalesced loads
// that would take a quarter of that).  Here a wave computes the taps of 64 z steps AT ONCE -- lane l does step z + l -- and
// then walks the 64 steps with the per-step quantities read back as scalars (v_readlane): per sample four coalesced 4-byte
// loads at (uniform offset) + (this lane's column), nine float operations with scalar weights, one maximum.  A step outside
// the window gets zero weights and offsets (its value is exactly 0, as the reference's), so the walk has no branches and the
// loads of several steps are in flight together.  Same samples, same float32 operations on them: bit-identical planes.
__global__ __launch_bounds__(kThreads) void mip_rot_x_kernel(int *__restrict__ out, const float *__restrict__ src, MipArgs a) {
    const int lane = threadIdx.x & 63;
    const int x = blockIdx.x * 64 + lane;
    const int y = blockIdx.y * kWaves + (threadIdx.x >> 6);
    if (y >= a.ty) return;  // whole wave
    const int per = (a.tz + a.segments - 1) / a.segments;
    int z0 = blockIdx.z * per, z1 = min(a.tz, z0 + per);
    const float iy = (float)y;
    const float syf = (float)a.sy, szf = (float)a.sz;
    const float hy = a.m[5] * iy, hz = a.m[9] * iy;  // (m4*x + m5*y) with m4 = +-0: exactly m5*y
    auto coord_y = [&](float iz) { return ((hy + a.m[6] * iz) + a.m[7]) + 0.5f; };
    auto coord_z = [&](float iz) { return ((hz + a.m[10] * iz) + a.m[11]) + 0.5f; };
    {   // conservative z range of the ray (the same for all lanes); the exact window test follows per step
        float lo = (float)z0, hi = (float)z1;
        const float ext[2] = {syf, szf}, slope[2] = {a.m[6], a.m[10]}, t0[2] = {coord_y(0.0f), coord_z(0.0f)};
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (fabsf(slope[k]) >= 0.01f) {
                const float za = (0.0f - t0[k]) / slope[k], zb = (ext[k] - t0[k]) / slope[k];
                lo = fmaxf(lo, fminf(za, zb) - 2.0f);
                hi = fminf(hi, fmaxf(za, zb) + 2.0f);
            } else if (t0[k] < -1.0f - 0.01f * (float)a.tz || t0[k] > ext[k] + 1.0f + 0.01f * (float)a.tz) {
                hi = lo;
            }
        }
        if (lo < hi) {
            z0 = max(z0, (int)floorf(lo));
            z1 = min(z1, (int)ceilf(hi));
        } else if (!(lo != lo || hi != hi)) {
            z1 = z0;
        }
    }
    // this lane's column: tx = x + k + 0.5 exactly, window 0 <= x + k <= sx - 1, both x taps on column x + k
    const int col = x + (int)a.m[3];
    const bool in_x = x < a.tx && col >= 0 && col < a.sx;
    const unsigned voff = 4u * (unsigned)(in_x ? col : 0);
    const __amdgpu_buffer_rsrc_t rsrc = small_rsrc(src, a.sx, a.sy, a.sz);
    const unsigned plane = (unsigned)a.sx * (unsigned)a.sy;
    float best = 0.0f;
    for (int zb = z0; zb < z1; zb += 64) {
        // the taps of steps zb .. zb + 63, one per lane
        const int z = zb + lane;
        const float iz = (float)z;
        const float ty = coord_y(iz), tz = coord_z(iz);
        const bool in = z < z1 && ty >= 0 && tz >= 0 && ty < syf && tz < szf;
        const Tap Y = make_tap<false>(in ? ty : 0.5f, a.sy, 0.0f), Z = make_tap<false>(in ? tz : 0.5f, a.sz, 0.0f);
        const unsigned o00 = 4u * ((unsigned)Z.i0 * plane + (unsigned)Y.i0 * (unsigned)a.sx);
        const unsigned dy = 4u * (unsigned)(Y.i1 - Y.i0) * (unsigned)a.sx, dz = 4u * (unsigned)(Z.i1 - Z.i0) * plane;
        const float wy0 = in ? Y.w0 : 0.0f, wy1 = in ? Y.w1 : 0.0f, wz0 = in ? Z.w0 : 0.0f, wz1 = in ? Z.w1 : 0.0f;
        const int n = min(64, z1 - zb);
        // eight steps (32 row loads) per trip: the waves spend 62 % of their cycles waiting for loads (profiles/r06/aux_pmc.json);
        // 4 / 8 / 16 steps per trip: 2.74 / 2.68 / 3.49 ms per 36-angle series about x (profiles/r06/mip3d_grid.txt)
#ifndef MIP_STEPS_PER_TRIP
#define MIP_STEPS_PER_TRIP 8
#endif
        for (int i0 = 0; i0 < n; i0 += MIP_STEPS_PER_TRIP) {  // steps per trip (a step beyond z1 has zero weights: its value is 0)
#pragma unroll
            for (int k = 0; k < MIP_STEPS_PER_TRIP; ++k) {
                const int i = i0 + k;
                const unsigned s00 = __builtin_amdgcn_readlane(o00, i);
                const unsigned sdy = __builtin_amdgcn_readlane(dy, i), sdz = __builtin_amdgcn_readlane(dz, i);
                const float sy0 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(wy0), i));
                const float sy1 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(wy1), i));
                const float sz0 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(wz0), i));
                const float sz1 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(wz1), i));
                auto tap = [&](unsigned so) { return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)voff, (int)so, 0)); };
                const float c00 = tap(s00), c01 = tap(s00 + sdy), c10 = tap(s00 + sdz), c11 = tap(s00 + sdy + sdz);
                const float c0 = sy0 * c00 + sy1 * c01;
                const float c1 = sy0 * c10 + sy1 * c11;
                const float v = sz0 * c0 + sz1 * c1;
                best = (best > v) ? best : v;
            }
        }
    }
    if (in_x && best > 0.0f) atomicMax(out + (size_t)x + (size_t)a.tx * y, __float_as_int(best));
}

// row r of m is (unit vector of axis r, integer shift): t_r = index + k + 0.5 exactly for every index of the box
bool passes_through(const float *m, int r, long long box_extent, long long src_extent) {
    for (int c = 0; c < 3; ++c)
        if (m[4 * r + c] != (c == r ? 1.0f : 0.0f)) return false;  // -0.0f == 0.0f: adding a signed zero changes nothing
    const float k = m[4 * r + 3];
    return k == std::nearbyint(k) && std::fabs(k) < 4194304.0f && box_extent < 4194304 && src_extent < 4194304;
}

// re-gather rates of a wave of 64 consecutive x marching along z / along y: pick_kernel's estimate (corr_padded.hip)
void exception_rates(const float *m, double &along_z, double &along_y) {
    auto frac_dist = [](double v) { return std::fabs(v - std::nearbyint(v)); };
    auto phases = [&](double per_x) { return std::min(64.0, 1.0 + 64.0 * frac_dist(per_x)); };
    const double px = phases(m[0]), py = phases(m[4]), pz = phases(m[8]);
    along_z = std::fabs((double)m[2]) * px + std::fabs((double)m[6]) * py + std::fabs((double)m[10] - 1.0) * pz;
    along_y = std::fabs((double)m[1]) * px + std::fabs((double)m[5] - 1.0) * py + std::fabs((double)m[9]) * pz;
}

// Workgroups of a marching resample: kWgsPerCu per compute unit -- twice what a compute unit holds at once (five), so that
// 2 560 of the 4 096 pieces of a 512 x 512 x 256 volume are somebody's first piece and the counter hands out the rest.
// Swept on one box (tools/sweep_resample_wgs.py, profiles/r06/resample_ticket.txt; identity | 3 deg + 2 % | 10 deg about z, ms):
// 6: 0.128 | 0.142 | 0.170, 8: 0.123 | 0.143 | 0.158, 10: 0.122 | 0.138 | 0.146, 12: 0.131 | 0.146 | 0.157, 16 (one piece
// each, no counter): 0.140 | 0.159 | 0.161.  The output does not depend on the grid.
unsigned march_grid(unsigned long long total_steps) {
    static std::atomic<int> cus_of[64];  // compute units by device number (0: not asked yet)
    int dev = 0;
    MIL_HIP(hipGetDevice(&dev));
    int cus = dev < 64 ? cus_of[dev].load(std::memory_order_relaxed) : 0;
    if (cus == 0) {
        MIL_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        cus = std::max(cus, 1);
        if (dev < 64) cus_of[dev].store(cus, std::memory_order_relaxed);
    }
    int per_cu = kWgsPerCu;
    if (const char *e = getenv("MIL_RESAMPLE_WGS_PER_CU")) per_cu = std::max(1, atoi(e));
    const unsigned long long want = (unsigned long long)cus * per_cu;
    const unsigned long long most = (total_steps + kMinChunk - 1) / kMinChunk;
    return (unsigned)std::max(1ull, std::min(want, most));
}

// Ticket counters.  A stream owns a PAIR of 32-bit counters and its launches alternate between them: launch n draws from word
// n % 2, which launch n - 1 zeroed while it ran (its first workgroup, a plain agent-scope store; launches of a stream run one
// after the other, so nobody else touches that word meanwhile), and zeroes word (n + 1) % 2 for its successor -- no reset launch
// between resamples (a 4-byte hipMemsetAsync in front of every resample cost 3-4 us of 150), no re-arming by "the last
// workgroup", nothing that depends on how far an earlier launch got.  The host zeroes both words on the stream before the
// pair's first use, and -- as memset NODES around the kernel -- when the launch is being captured into a graph, so that every
// replay finds its word zero and leaves both words zero for whatever runs next on the stream.  The per-thread stream handle names
// a different stream in every host thread and is keyed by the thread as well.  mil_stream_destroy hands a stream's pair back
// with its launch count (release_march_ticket); pairs of foreign streams live for the life of the process, 8 bytes each.
struct TicketKey {
    int dev;
    hipStream_t s;
    std::thread::id tid;  // only for hipStreamPerThread
    bool operator==(const TicketKey &o) const { return dev == o.dev && s == o.s && tid == o.tid; }
};
struct TicketKeyHash {
    size_t operator()(const TicketKey &k) const {
        return std::hash<const void *>()(k.s) * 1000003u ^ std::hash<std::thread::id>()(k.tid) * 31u ^ (size_t)k.dev;
    }
};
struct TicketWord {
    int dev = 0;
    unsigned *p = nullptr;   // the pair
    unsigned launches = 0;   // queued on the pair so far
    bool fresh = true;       // never used: contents undefined
    bool captured = false;   // a graph holds launches on this pair: it may replay them after the stream is gone, so the pair is
                             // never handed to another stream (8 bytes stay behind)
};
struct TicketTable {
    std::mutex mu;
    std::unordered_map<TicketKey, TicketWord, TicketKeyHash> slots;
    std::vector<TicketWord> free_words;
};
TicketTable &tickets() {
    static TicketTable *t = new TicketTable;  // never destroyed: streams may be released during process teardown
    return *t;
}
struct Ticket {
    unsigned *cur, *next;
    bool captured;  // the launch is being recorded into a graph: zero `cur` again behind the kernel
};
// the counters of the launch about to be queued on stream `s`; the caller holds tickets().mu until that launch IS queued (two
// host threads queuing on one stream must enter the stream in the order in which they were dealt their words)
Ticket ticket_of(hipStream_t s) {
    TicketKey key{0, s, s == hipStreamPerThread ? std::this_thread::get_id() : std::thread::id()};
    MIL_HIP(hipGetDevice(&key.dev));
    hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &capture) != hipSuccess) {
        (void)hipGetLastError();
        capture = hipStreamCaptureStatusNone;
    }
    TicketTable &t = tickets();
    auto it = t.slots.find(key);
    if (it == t.slots.end()) {
        size_t f = 0;
        for (; f < t.free_words.size(); ++f)
            if (t.free_words[f].dev == key.dev) break;
        if (f == t.free_words.size()) {  // a 4 KB block of 512 pairs
            unsigned *p = nullptr;
            MIL_HIP(hipMalloc((void **)&p, 4096));
            for (int i = 511; i >= 0; --i) t.free_words.push_back(TicketWord{key.dev, p + 2 * i, 0u, true, false});
            f = t.free_words.size() - 1;
        }
        it = t.slots.emplace(key, t.free_words[f]).first;
        t.free_words.erase(t.free_words.begin() + (long)f);
    }
    TicketWord &w = it->second;
    const bool captured = capture != hipStreamCaptureStatusNone;
    unsigned *cur = w.p + (w.launches & 1u), *next = w.p + ((w.launches + 1u) & 1u);
    ++w.launches;
    if (w.fresh) MIL_HIP(hipMemsetAsync(w.p, 0, 2 * sizeof(unsigned), s));
    else if (captured) MIL_HIP(hipMemsetAsync(cur, 0, sizeof(unsigned), s));
    w.fresh = false;
    w.captured = w.captured || captured;
    return Ticket{cur, next, captured};
}

}  // namespace

// The stream is about to be destroyed (mil_stream_destroy): its counter goes back to the free list of its device.
void release_march_ticket(hipStream_t s) {
    if (s == hipStreamPerThread || s == nullptr) return;
    TicketTable &t = tickets();
    std::lock_guard<std::mutex> g(t.mu);
    for (auto it = t.slots.begin(); it != t.slots.end();) {
        if (it->first.s == s) {
            if (!it->second.captured) t.free_words.push_back(it->second);  // with its launch count: the next owner continues the alternation
            it = t.slots.erase(it);
        } else {
            ++it;
        }
    }
}

// Volumes the marching kernel takes: fetch_linear_small's range (below 2 GiB, planes below 2^24 voxels, sx >= 2, exact
// float32 weights) and an output whose byte offsets fit 32 bits.  MIL_RESAMPLE_MARCH=0 keeps the one-voxel kernel (A/B).
bool resample_march_eligible(Dims out, Dims s) {
    const char *e = getenv("MIL_RESAMPLE_MARCH");
    if (e && e[0] == '0') return false;
    return small_source(s) && out.count() * 4 < (1ll << 32) && out.nx > 0 && out.ny > 0 && out.nz > 0 &&
           s.nx * s.ny * 4 < (1ll << 24) && s.nz < (1ll << 24);
}

void affine_resample_march(float *out, const float *src, Dims o, Dims sd, const Affine12 &m, hipStream_t s, bool strict_window) {
    MarchArgs a;
    for (int i = 0; i < 12; ++i) a.m[i] = m.m[i];
    a.tx = (int)o.nx; a.ty = (int)o.ny; a.tz = (int)o.nz;
    a.sx = (int)sd.nx; a.sy = (int)sd.ny; a.sz = (int)sd.nz;
    a.bx1 = (float)(sd.nx - 1); a.by1 = (float)(sd.ny - 1); a.bz1 = (float)(sd.nz - 1);
    a.row4 = (int)(4 * sd.nx);
    a.plane4 = (int)(4 * sd.nx * sd.ny);
    a.bytes = (unsigned)(4 * sd.count());
    double rz, ry;
    exception_rates(a.m, rz, ry);
    const bool march_y = ry < rz && o.ny >= 16;  // the axis along which a wave re-gathers less often
    const int run_len = march_y ? a.ty : a.tz;
    a.gx = (a.tx + 2 * kLanes - 1) / (2 * kLanes);
    a.gy = ((march_y ? a.tz : a.ty) + kWaves - 1) / kWaves;
    const unsigned long long pieces = (unsigned long long)a.gx * a.gy * ((run_len + kPiece - 1) / kPiece);
    if (pieces > 0x7fffffffull) throw Error(-1, "affine_resample: volume exceeds launch limits");
    std::lock_guard<std::mutex> queue_order(tickets().mu);
    const Ticket ticket = ticket_of(s);
    const unsigned wgs = march_grid(pieces * kPiece);
    auto launch = [&](auto kern) {
        hipLaunchKernelGGL(kern, dim3(wgs), dim3(kThreads), 0, s, out, src, a, ticket.cur, ticket.next, run_len);
    };
    if (march_y && strict_window) launch(resample_march_kernel<true, true>);
    else if (march_y) launch(resample_march_kernel<true, false>);
    else if (strict_window) launch(resample_march_kernel<false, true>);
    else launch(resample_march_kernel<false, false>);
    MIL_HIP(hipGetLastError());
    if (ticket.captured) MIL_HIP(hipMemsetAsync(ticket.cur, 0, sizeof(unsigned), s));  // a replay leaves both words zero
}

// MIL_MIP3D_V2=0 keeps resample_mip_kernel (affine.hip) for comparisons.
bool resample_mip2_eligible(Dims box, Dims sd) {
    const char *e = getenv("MIL_MIP3D_V2");
    if (e && e[0] == '0') return false;
    return small_source(sd) && box.nx > 0 && box.ny > 0 && box.nz > 0 && box.nx < (1 << 24) && box.ny < (1 << 24) && box.nz < (1 << 24);
}

void resample_mip2_z(float *out_plane, const float *src, Dims box, Dims sd, const Affine12 &m, hipStream_t s, bool plane_is_zero) {
    MipArgs a;
    for (int i = 0; i < 12; ++i) a.m[i] = m.m[i];
    a.tx = (int)box.nx; a.ty = (int)box.ny; a.tz = (int)box.nz;
    a.sx = (int)sd.nx; a.sy = (int)sd.ny; a.sz = (int)sd.nz;
    const int skip = passes_through(a.m, 0, box.nx, sd.nx) ? 1 : (passes_through(a.m, 1, box.ny, sd.ny) ? 2 : 0);
    // source rows / planes a wave's gather spreads over, with its 64 lanes along x or along z (resample_mip_kernel)
    const float along_x = 64.f * (fabsf(m.m[4]) + fabsf(m.m[8])), along_z = 64.f * (fabsf(m.m[6]) + fabsf(m.m[10]));
    const int lx = along_x < 8.f ? 64 : (along_z < 8.f ? 1 : 8);
    const int lz = 64 / lx;
    a.segments = std::max(1, std::min(4, a.tz / (lz * 32)));
    dim3 grid((a.tx + lx - 1) / lx, (a.ty + kWaves - 1) / kWaves, a.segments);
    if (grid.y > 65535) throw Error(-1, "resample_mip_z: plane exceeds launch limits");
    if (!plane_is_zero) MIL_HIP(hipMemsetAsync(out_plane, 0, (size_t)a.tx * a.ty * sizeof(float), s));
    int *o = reinterpret_cast<int *>(out_plane);
    static const bool no_rot_x = getenv("MIL_MIP3D_ROTX") && getenv("MIL_MIP3D_ROTX")[0] == '0';
    if (skip == 1 && a.m[4] == 0.0f && a.m[8] == 0.0f && !no_rot_x) {  // a rotation about x: the wave-uniform form
        // z segments of >= 64 steps (eight for the 572 steps of a 512 x 512 x 256 volume's rotated box): 4 576 workgroups were 2.2
        // rounds of what the chip holds, the last one a quarter full; 192 / 128 / 64 / 36 steps per segment: 2.77 / 2.68 / 2.53 /
        // 2.53 ms per series about x, 3.21 / 3.12 / 2.97 / 2.97 about y (profiles/r06/mip3d_grid.txt)
#ifndef MIP_SEG_STEPS
#define MIP_SEG_STEPS 64
#endif
        a.segments = std::max(1, std::min(16, a.tz / MIP_SEG_STEPS));
        dim3 gridx((a.tx + 63) / 64, (a.ty + kWaves - 1) / kWaves, a.segments);
        hipLaunchKernelGGL(mip_rot_x_kernel, gridx, dim3(kThreads), 0, s, o, src, a);
        MIL_HIP(hipGetLastError());
        return;
    }
#define MIL_MIP2(LX)                                                                                          \
    switch (skip) {                                                                                           \
        case 1: hipLaunchKernelGGL((resample_mip2_kernel<LX, 1>), grid, dim3(kThreads), 0, s, o, src, a); break; \
        case 2: hipLaunchKernelGGL((resample_mip2_kernel<LX, 2>), grid, dim3(kThreads), 0, s, o, src, a); break; \
        default: hipLaunchKernelGGL((resample_mip2_kernel<LX, 0>), grid, dim3(kThreads), 0, s, o, src, a); break; \
    }
    if (lx == 64) { MIL_MIP2(64) }
    else if (lx == 8) { MIL_MIP2(8) }
    else { MIL_MIP2(1) }
#undef MIL_MIP2
    MIL_HIP(hipGetLastError());
}

// Rotation about y (mip3dgpu, rAxis 2) as a rotation about x of the TRANSPOSED volume.  rot2matrix leaves y alone -- row 1
// is (0, 1, 0, k) -- and rows 0 and 2 do not look at y (m1 = m9 = 0): the x / z taps of a sample are the same for every y, but
// y is the strided axis, so a wave cannot run along it.  With x and y exchanged (one transpose of the volume per SERIES, 0.2 ms)
// the same sample is  Z.w0 (X.w0 a + X.w1 b) + Z.w1 (...)  with 64 consecutive y in the lanes and wave-uniform taps: exactly
// mip_rot_x_kernel's walk with the matrix rows permuted -- new row 0 = (1, 0, 0, k), new row 1 = (0, m0, m2, m3), new row 2 =
// (0, m8, m10, m11) -- the same float operations in the same order, the plane comes out transposed.  36 angles at 512 x 512 x 256:
// 7.1 ms with resample_mip2_kernel<., 2> (one row per wave, bound by the latency of its two gathers per sample) -> see
// profiles/r05/resample_ab.txt.  (A kernel that kept 16 rows per lane to share the taps was measured first: 10.6 ms -- sixteen
// times the footprint per wave; not kept.)
bool mip_about_y_by_transpose(const Affine12 &m, Dims box, Dims sd, Affine12 &mt) {
    static const bool off = getenv("MIL_MIP3D_ROTY") && getenv("MIL_MIP3D_ROTY")[0] == '0';
    if (off || !resample_mip2_eligible(box, sd)) return false;
    if (!passes_through(m.m, 1, box.ny, sd.ny) || m.m[1] != 0.0f || m.m[9] != 0.0f) return false;
    if (passes_through(m.m, 0, box.nx, sd.nx) && m.m[4] == 0.0f && m.m[8] == 0.0f) return false;  // already a rotation about x
    const float r[12] = {1.0f, 0.0f, 0.0f, m.m[7], 0.0f, m.m[0], m.m[2], m.m[3], 0.0f, m.m[8], m.m[10], m.m[11]};
    for (int i = 0; i < 12; ++i) mt.m[i] = r[i];
    return true;
}

}  // namespace k
}  // namespace mil

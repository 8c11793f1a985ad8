// Hand-written gfx950 kernels for the registration hot path: explicit trilinear
// resampling (gfx950 has no texture unit -- hipcc rejects tex3D for this target)
// and the fused interpolate + correlate + reduce cost kernel that Powell drives.
//
// Semantics restated from the reference CUDA kernels (include/cukernel.cuh:501-556):
//   t = M*(x,y,z,1) + 0.5 evaluated left to right in float32 (no FMA: this file is
//   built with -ffp-contract=off -- the oracle's reading of the source, bit-exact against oracle/reg_oracle.py, whose
//   trilinear arithmetic is pinned to the reference's own itrilerp2 / affinetransformcpu compiled by gcc; an nvcc build
//   with --fmad=true may contract the coordinate expression), validity window on t, then a linear "texture"
//   fetch at t with texel centres at +0.5 and neighbour indices clamped to the edge.
// Interpolation weights are exact float32 (CUDA's texture unit quantises them to 8
// fractional bits; see DESIGN.md "Registration numerics").
//
// corr_kernel here is the GENERAL cost kernel (any volume size, 2-D images, the raw C-ABI entry mil_corr_sums): per target
// voxel one coalesced target read and four 8-byte source gathers with clamped indices, two of them usually skipped by
// plane reuse.  The 3-D registration itself evaluates the border-padded kernels of corr_padded.hip, which return the
// same sums bit for bit with about half the instructions; DESIGN.md section 4 has the measurements (what bounds these
// kernels is the texture-address path and VALU issue, not HBM) and the three LDS-staged variants that were built,
// measured slower and dropped.  Every cost kernel ends in corr_publish (corr_finish.h): one launch per evaluation.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <chrono>

#include "corr_finish.h"
#include "kernels.h"
#include "trilinear_dev.h"

#include <cstdlib>


namespace mil {
namespace k {

namespace {

constexpr int kTileX = 64;   // one wave spans 64 consecutive x -> coalesced target reads
constexpr int kTileY = 4;    // 4 waves per workgroup
constexpr int kCorrZ = 16;   // z-steps per thread in the cost kernel
constexpr int kBlockThreads = kTileX * kTileY;

using namespace tri;

// Plane cache of the cost kernel.  A lane walks z; for the near-axis-aligned matrices registration visits, the lower
// source plane of step z+1 is the upper plane of step z (same row pair, same x pair), so its two 8-byte gathers were
// already fetched.  Reuse is decided by ADDRESS equality -- it never changes a value, only skips a gather -- and cuts
// the gathers per voxel from 4 to ~2 (the kernel is bound by gather issue).
struct PlaneCache {
    size_t a0 = ~(size_t)0, a1 = ~(size_t)0;  // element offsets of the cached upper-plane pairs (rows y0, y1)
    Pair q0{0.f, 0.f}, q1{0.f, 0.f};
};

__device__ __forceinline__ float fetch_linear_cached(const float *__restrict__ src, int sx, int sy, int sz, float tx,
                                                     float ty, float tz, PlaneCache &pc, float wq) {
    Tap X = make_tap(tx, sx, wq), Y = make_tap(ty, sy, wq), Z = make_tap(tz, sz, wq);
    const size_t row = (size_t)sx, plane = (size_t)sx * sy;
    const int b = X.i0 < sx - 2 ? X.i0 : sx - 2;  // sx >= 2 guaranteed by the launcher
    const bool lo0 = X.i0 == b, lo1 = X.i1 == b;
    const size_t o00 = Z.i0 * plane + Y.i0 * row + b, o01 = Z.i0 * plane + Y.i1 * row + b;
    const size_t o10 = Z.i1 * plane + Y.i0 * row + b, o11 = Z.i1 * plane + Y.i1 * row + b;
    Pair q00, q01;
    if (o00 == pc.a0 && o01 == pc.a1) {
        q00 = pc.q0;
        q01 = pc.q1;
    } else {
        q00 = *reinterpret_cast<const Pair *>(src + o00);
        q01 = *reinterpret_cast<const Pair *>(src + o01);
    }
    const Pair q10 = *reinterpret_cast<const Pair *>(src + o10), q11 = *reinterpret_cast<const Pair *>(src + o11);
    pc.a0 = o10; pc.a1 = o11; pc.q0 = q10; pc.q1 = q11;
    float c00 = X.w0 * (lo0 ? q00.lo : q00.hi) + X.w1 * (lo1 ? q00.lo : q00.hi);
    float c01 = X.w0 * (lo0 ? q01.lo : q01.hi) + X.w1 * (lo1 ? q01.lo : q01.hi);
    float c10 = X.w0 * (lo0 ? q10.lo : q10.hi) + X.w1 * (lo1 ? q10.lo : q10.hi);
    float c11 = X.w0 * (lo0 ? q11.lo : q11.hi) + X.w1 * (lo1 ? q11.lo : q11.hi);
    float c0 = Y.w0 * c00 + Y.w1 * c01;
    float c1 = Y.w0 * c10 + Y.w1 * c11;
    return Z.w0 * c0 + Z.w1 * c1;
}

// affinetransformkernel, reference include/cukernel.cuh:501-524 (window 0 <= t < s2).
// STRICT: the 2-D variant tests 0 < t (affineTransform2Dkernel, include/cukernel.cuh:558-573).
// SMALL: source below 2 GiB with planes below 2^24 voxels and sx >= 2 (small_source()): fetch_linear_small.
template <bool STRICT, bool SMALL>
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
    const bool lower = STRICT ? (tx > 0 && ty > 0 && tz > 0) : (tx >= 0 && ty >= 0 && tz >= 0);
    if (lower && tx < (float)a.sx && ty < (float)a.sy && tz < (float)a.sz) {
        if constexpr (SMALL) v = fetch_linear_small(small_rsrc(src, a.sx, a.sy, a.sz), a.sx, a.sy, a.sz, tx, ty, tz, a.wq);
        else v = fetch_linear(src, a.sx, a.sy, a.sz, tx, ty, tz, a.wq);
    }
    out[(size_t)x + (size_t)a.tx * ((size_t)y + (size_t)a.ty * z)] = v;
}

// mip3dgpu's inner pair (src/apifunc.cpp:610-613: affineTransform into the rotated box, then maxprojection along z) in ONE
// pass: a thread owns an (x, y) of the projection and a segment of z, forms the resampled values exactly as resample_kernel
// does and keeps their maximum -- the rotated box (670 MB at 512 x 512 x 256, written and read back 36 times per series) never
// exists.  The maximum is exact and order-independent, so the plane equals the two-kernel result bit for bit.  Segments meet
// in an atomic max on the integer pattern: the accumulator starts at 0 (maxprojectionkernel :396-418), non-negative floats
// order like their bit patterns and negative ones stay below 0.
// Lane layout: LX lanes along x times 64 / LX along z.  A wave's gather then covers an LX x (64 / LX) patch of the (x, z)
// plane, which a rotation about y maps to a patch of the source's (x, z) plane: 64 x 1 is right while the source planes a wave
// touches stay few (small angles: 2-3 cache lines per gather, 45 at 45 degrees), 1 x 64 when the march direction has
// become the source's x axis (near 90 / 270 degrees), 8 x 8 in between.  The z lanes of an (x, y) meet in a wave-level
// maximum, the z segments of the grid in the atomic.
template <int LX, bool SMALL>
__global__ __launch_bounds__(kBlockThreads) void resample_mip_kernel(int *__restrict__ out, const float *__restrict__ src,
                                                                     AffineArgs a, int segments) {
    constexpr int LZ = 64 / LX;
    const int lane = threadIdx.x & 63;
    const int x = blockIdx.x * LX + (lane % LX);
    const int y = blockIdx.y * kTileY + (threadIdx.x >> 6);
    if (y >= a.ty) return;  // whole wave
    const int per = (a.tz + segments - 1) / segments;
    const int z0 = blockIdx.z * per, z1 = min(a.tz, z0 + per);
    const float ix = (float)x, iy = (float)y;
    float best = 0.0f;
    if (x < a.tx) {
#pragma unroll 4
        for (int z = z0 + lane / LX; z < z1; z += LZ) {
            const float iz = (float)z;
            const float tx = coord(a.m + 0, ix, iy, iz);
            const float ty = coord(a.m + 4, ix, iy, iz);
            const float tz = coord(a.m + 8, ix, iy, iz);
            float v = 0.0f;
            if (tx >= 0 && ty >= 0 && tz >= 0 && tx < (float)a.sx && ty < (float)a.sy && tz < (float)a.sz) {
                if constexpr (SMALL) v = fetch_linear_small(small_rsrc(src, a.sx, a.sy, a.sz), a.sx, a.sy, a.sz, tx, ty, tz, a.wq);
                else v = fetch_linear(src, a.sx, a.sy, a.sz, tx, ty, tz, a.wq);
            }
            best = (best > v) ? best : v;
        }
    }
#pragma unroll
    for (int off = LX; off < 64; off <<= 1) {
        const float o = __shfl_xor(best, off, 64);
        best = (best > o) ? best : o;
    }
    if (lane < LX && x < a.tx && best > 0.0f) atomicMax(out + (size_t)x + (size_t)a.tx * y, __float_as_int(best));
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
//
// FLOAT_PRODUCTS: the 2-D variant of the reference (corr2Dkernel, include/cukernel.cuh:575-597) forms t*t and s*t in
// float32 and sums those in float64 on the host (src/api_subfunc.cu:1027-1030); a 2-D image is a one-plane volume here.
// CACHE: reuse the previous step's upper source plane when it is this step's lower one (needs sx >= 2).
template <bool FLOAT_PRODUCTS, bool CACHE>
__global__ __launch_bounds__(kBlockThreads) void corr_kernel(const float *__restrict__ tgt,
                                                           const float *__restrict__ src, AffineArgs a, CorrTail tail) {
    __shared__ double lds[2 * kTileY];
    __shared__ int last;
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
        PlaneCache pc;
#pragma unroll 4
        for (int z = z0; z < z1; ++z, tp += tstep) {
            const float iz = (float)z;
            const float tx = coord(a.m + 0, ix, iy, iz);
            const float ty = coord(a.m + 4, ix, iy, iz);
            const float tz = coord(a.m + 8, ix, iy, iz);
            float sv = 0.0f;
            if (tx > 0 && tx < (float)a.sx && ty > 0 && ty < (float)a.sy && tz > 0 && tz < (float)a.sz)
                sv = CACHE ? fetch_linear_cached(src, a.sx, a.sy, a.sz, tx, ty, tz, pc, a.wq)
                           : fetch_linear(src, a.sx, a.sy, a.sz, tx, ty, tz, a.wq);
            const float tv = __builtin_nontemporal_load(tp);  // read once per evaluation: leave the Infinity Cache to the source (corr_padded.hip)
            if (FLOAT_PRODUCTS) {
                const float q = sv * sv, c = tv * sv;
                ss += (double)q;
                st += (double)c;
            } else {
                // float x float is exact in double, so the fused form rounds exactly like multiply-then-add
                ss = __builtin_fma((double)sv, (double)sv, ss);
                st = __builtin_fma((double)sv, (double)tv, st);
            }
        }
    }
    ss = wave_sum(ss);
    st = wave_sum(st);
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
    corr_publish<kBlockThreads>(a0, a1, tail, lds, &last);
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
    a.wq = texture_weight_bits() == 8 ? 256.0f : 0.0f;
    return a;
}

std::atomic<int> g_weight_bits{-1};  // -1: not set through the API, MIL_TEX_WEIGHT_BITS decides
std::atomic<int> g_corr_fences{-1};  // -1: MIL_CORR_FENCES (read once) decides

}  // namespace

// fetch_linear_small applies: exact float32 weights (no texture-unit emulation), 32-bit byte offsets, UNSIGNED 24-bit multiplies
// (planes below 2^24 voxels), 8-byte pairs inside a row.  MIL_RESAMPLE_SMALL=0 (read per call: the tests flip it) keeps the
// 64-bit address arithmetic.
bool small_source(Dims s) {
    const char *e = getenv("MIL_RESAMPLE_SMALL");
    return !(e && e[0] == '0') && texture_weight_bits() == 0 && s.nx >= 2 && s.count() * 4 < (1ll << 31) && s.nx * s.ny < (1ll << 24) &&
           s.nz < (1ll << 24);
}

// 0 (default): exact float32 interpolation weights; 8: weights rounded to 8 fractional bits like the CUDA texture unit
// the reference samples through (include/cukernel.cuh:510-522, 545-546) -- see make_tap.  Process-wide.
int texture_weight_bits() {
    const int v = g_weight_bits.load();
    if (v >= 0) return v;
    const char *e = getenv("MIL_TEX_WEIGHT_BITS");
    return (e && atoi(e) == 8) ? 8 : 0;
}
// MIL_CORR_FENCES=1 / corr_set_fences(1): release / acquire fences around the completion counters of the cost kernels (a
// diagnostic, csrc/corr_finish.h).  No getenv on the evaluation path: a registration makes ~2 000 of them back to back.
bool corr_fences() {
    const int v = g_corr_fences.load(std::memory_order_relaxed);
    if (v >= 0) return v != 0;
    static const bool env = getenv("MIL_CORR_FENCES") && getenv("MIL_CORR_FENCES")[0] == '1';
    return env;
}
void corr_set_fences(int on) { g_corr_fences.store(on < 0 ? -1 : (on ? 1 : 0)); }

void set_texture_weight_bits(int bits) { g_weight_bits.store(bits == 8 ? 8 : bits == 0 ? 0 : -1); }

void affine_resample(float *out, const float *src, Dims out_dims, Dims src_dims, const Affine12 &m, hipStream_t s,
                     bool strict_window) {
    AffineArgs a = make_args(out_dims, src_dims, m);
    dim3 grid((a.tx + kTileX - 1) / kTileX, (a.ty + kTileY - 1) / kTileY, a.tz);
    if (grid.y > 65535 || grid.z > 65535) throw Error(-1, "affine_resample: volume exceeds launch limits");
    if (resample_march_eligible(out_dims, src_dims)) {  // two voxels per lane, marching with plane / row reuse
        affine_resample_march(out, src, out_dims, src_dims, m, s, strict_window);
        return;
    }
    const bool small = small_source(src_dims);
    if (strict_window && small) hipLaunchKernelGGL((resample_kernel<true, true>), grid, dim3(kBlockThreads), 0, s, out, src, a);
    else if (strict_window) hipLaunchKernelGGL((resample_kernel<true, false>), grid, dim3(kBlockThreads), 0, s, out, src, a);
    else if (small) hipLaunchKernelGGL((resample_kernel<false, true>), grid, dim3(kBlockThreads), 0, s, out, src, a);
    else hipLaunchKernelGGL((resample_kernel<false, false>), grid, dim3(kBlockThreads), 0, s, out, src, a);
    MIL_HIP(hipGetLastError());
}

void resample_mip_z(float *out_plane, const float *src, Dims box, Dims src_dims, const Affine12 &m, hipStream_t s, bool plane_is_zero) {
    if (resample_mip2_eligible(box, src_dims)) {  // taps under exact zero weights left out, z range clipped (resample_march.hip)
        resample_mip2_z(out_plane, src, box, src_dims, m, s, plane_is_zero);
        return;
    }
    AffineArgs a = make_args(box, src_dims, m);
    // source rows / planes a wave's gather spreads over, with its 64 lanes along x or along z (see the kernel)
    const float along_x = 64.f * (fabsf(m.m[4]) + fabsf(m.m[8])), along_z = 64.f * (fabsf(m.m[6]) + fabsf(m.m[10]));
    const int lx = along_x < 8.f ? 64 : (along_z < 8.f ? 1 : 8);
    const int lz = 64 / lx;
    const int segments = std::max(1, std::min(4, a.tz / (lz * 32)));
    dim3 grid((a.tx + lx - 1) / lx, (a.ty + kTileY - 1) / kTileY, segments);
    if (grid.y > 65535) throw Error(-1, "resample_mip_z: plane exceeds launch limits");
    if (!plane_is_zero) MIL_HIP(hipMemsetAsync(out_plane, 0, (size_t)a.tx * a.ty * sizeof(float), s));
    int *o = reinterpret_cast<int *>(out_plane);
#define MIL_MIP(LX)                                                                                                   \
    if (small_source(src_dims)) hipLaunchKernelGGL((resample_mip_kernel<LX, true>), grid, dim3(kBlockThreads), 0, s, o, src, a, segments); \
    else hipLaunchKernelGGL((resample_mip_kernel<LX, false>), grid, dim3(kBlockThreads), 0, s, o, src, a, segments);
    if (lx == 64) { MIL_MIP(64) }
    else if (lx == 8) { MIL_MIP(8) }
    else { MIL_MIP(1) }
#undef MIL_MIP
    MIL_HIP(hipGetLastError());
}

CorrWorkspace::CorrWorkspace() {
    // coherent: the kernel's system-scope store of the ticket is visible to the polling host while the stream runs on
    MIL_HIP(hipHostMalloc((void **)&host, 3 * sizeof(double), hipHostMallocCoherent));
    host[0] = host[1] = host[2] = 0.0;
    counter.alloc(kCorrCounters + 1);  // zeroed by the first next(), ON THE STREAM the evaluation runs on
}
CorrWorkspace::~CorrWorkspace() {
    if (host) (void)hipHostFree(host);
}

CorrTail CorrWorkspace::next(size_t nblocks, hipStream_t s) {
    if (nblocks == 0 || nblocks > (size_t)(1 << 30)) throw Error(-1, "corr_sums: volume exceeds launch limits");
    if (!counters_zeroed) {
        // hipMemset (no stream) is asynchronous to the host and runs on the NULL stream, which our non-blocking streams do not
        // wait for: under load (several host threads on one device) it could land in the middle of the first evaluations and
        // wipe the completion counts -- no workgroup was "last", nothing was published.  On the evaluation's own stream it is
        // ordered before the first kernel.
        MIL_HIP(hipMemsetAsync(counter.get(), 0, (kCorrCounters + 1) * sizeof(unsigned), s));
        counters_zeroed = true;
    }
    if (partials.size() < 2 * nblocks + 2 * kCorrCounters) partials.alloc(2 * nblocks + 2 * kCorrCounters);
    ticket += 1.0;
    return CorrTail{partials.get(), (int)nblocks, counter.get(), host, ticket, corr_fences() ? 1 : 0};
}

void CorrWorkspace::wait(hipStream_t s) {
    // The last workgroup stores the ticket after the sums (system scope).  Poll it: a stream synchronisation costs ~10 us
    // more per evaluation.  Should the ticket not show up within 100 ms of wall clock (a failed launch, a fault), fall
    // back to the synchronisation, which reports the error; on any failure the completion counters are zeroed and the
    // ticket advanced, so that stale counts cannot make a later evaluation elect the wrong "last" workgroup.
    volatile const double *flag = host + 2;
    static const bool no_poll = getenv("MIL_CORR_POLL") && getenv("MIL_CORR_POLL")[0] == '0';
    auto recover = [&]() {
        (void)hipStreamSynchronize(s);
        (void)hipMemsetAsync(counter.get(), 0, (kCorrCounters + 1) * sizeof(unsigned), s);
        (void)hipStreamSynchronize(s);
        ticket += 1.0;
    };
    if (no_poll) {
        const hipError_t e = hipStreamSynchronize(s);
        if (e != hipSuccess || *flag != ticket) {
            recover();
            check_hip(e, "hipStreamSynchronize", __FILE__, __LINE__);
            throw Error(-1, "corr_sums: the cost kernel did not publish its result");
        }
        return;
    }
    const auto t0 = std::chrono::steady_clock::now();
    for (long spin = 0;; ++spin) {
        if (*flag == ticket) {
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
            return;
        }
        __builtin_ia32_pause();
        if ((spin & 0xfff) == 0xfff &&
            std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 0.1)
            break;
    }
    const hipError_t e = hipStreamSynchronize(s);
    if (e == hipSuccess && *flag == ticket) return;  // a slow evaluation (a huge volume), not a failed one
    recover();
    check_hip(e, "hipStreamSynchronize", __FILE__, __LINE__);
    throw Error(-1, "corr_sums: the cost kernel did not publish its result");
}

void corr_sums_async(const float *target, const float *src, Dims t_dims, Dims s_dims, const Affine12 &m,
                     CorrWorkspace &ws, hipStream_t s, bool float_products) {
    AffineArgs a = make_args(t_dims, s_dims, m);
    dim3 grid((a.tx + kTileX - 1) / kTileX, (a.ty + kTileY - 1) / kTileY, (a.tz + kCorrZ - 1) / kCorrZ);
    if (grid.y > 65535 || grid.z > 65535) throw Error(-1, "corr_sums: volume exceeds launch limits");
    const CorrTail tail = ws.next((size_t)grid.x * grid.y * grid.z, s);
    static const bool no_cache = getenv("MIL_CORR_NOCACHE") && getenv("MIL_CORR_NOCACHE")[0] == '1';
    if (float_products)  // 2-D images (one plane): nothing to reuse along z
        hipLaunchKernelGGL((corr_kernel<true, false>), grid, dim3(kBlockThreads), 0, s, target, src, a, tail);
    else if (a.sx >= 2 && !no_cache)
        hipLaunchKernelGGL((corr_kernel<false, true>), grid, dim3(kBlockThreads), 0, s, target, src, a, tail);
    else
        hipLaunchKernelGGL((corr_kernel<false, false>), grid, dim3(kBlockThreads), 0, s, target, src, a, tail);
    MIL_HIP(hipGetLastError());
}

void corr_sums_sync(const float *target, const float *src, Dims t_dims, Dims s_dims, const Affine12 &m,
                    CorrWorkspace &ws, hipStream_t s, double *ss, double *st, bool float_products) {
    corr_sums_async(target, src, t_dims, s_dims, m, ws, s, float_products);
    ws.wait(s);
    *ss = ws.host[0];
    *st = ws.host[1];
}

}  // namespace k
}  // namespace mil

// Hand-written gfx950 (CDNA4) kernels for the Richardson-Lucy hot path:
// element-wise maps, float64 reductions and the pad / crop / flip family.
//
// All of these are HBM-bound (no contraction anywhere -> no MFMA).  Design rules
// applied (cdna_hip_programming.md G2/G11/G13, Appendix B "Element-wise"):
//   * 16 B per lane per access (float4), 64-lane waves fully coalesced: one wave
//     instruction moves 1 KiB;
//   * several independent loads in flight per lane before the first use;
//   * grids of >> 256 workgroups so all 8 XCDs / 256 CUs stream;
//   * x (TIFF fastest axis) is always the lane axis -- the reference's pad / crop /
//     flip kernels put threadIdx.x on the slowest axis (include/cukernel.cuh:694,735,
//     750,768) and are fully uncoalesced; these are not.
// This file is compiled with -ffp-contract=off: every float op rounds on its own, which is how the ORACLE reads the
// reference's C source (bit-exact against oracle/decon_oracle.py, itself pinned to the reference's CPU helpers compiled
// by gcc: tests/test_oracle_pinned.py).  It is NOT a statement about a CUDA build of the reference: nvcc's default
// --fmad=true contracts a*b+c inside e.g. multicomplex3Dkernel, a difference far below the 1e-5 tolerance.
#include "kernels.h"

#include <cstdint>

namespace mil {
namespace k {

namespace {

constexpr int kBlock = 256;      // 4 waves per workgroup
constexpr int kUnroll = 4;       // float4 per lane in flight
constexpr int kMaxGrid = 16384;  // cap; grid-stride beyond

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

inline int ew_grid(size_t n4) {
    size_t per_block = (size_t)kBlock * kUnroll;
    size_t g = (n4 + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > (size_t)kMaxGrid) g = kMaxGrid;
    return (int)g;
}

// ---- functors -------------------------------------------------------------
struct RatioOp {  // div3Dkernel, reference include/cukernel.cuh:196-206 (unguarded divide)
    __device__ __forceinline__ float operator()(float t, float a) const { return a / t; }
};
struct UpdateOp {  // multi3Dkernel :115-126 fused with maxvalue3Dgpukernel :383-392
    float eps;
    __device__ __forceinline__ float operator()(float e, float t) const {
        float v = e * t;
        return (v > eps) ? v : eps;
    }
};

// io[i] = op(io[i], in[i]) -- 2 reads + 1 write per element.
template <class Op>
__global__ __launch_bounds__(kBlock) void ew_inplace2_kernel(float *__restrict__ io, const float *__restrict__ in,
                                                           size_t n, Op op) {
    const size_t n4 = n >> 2;
    float4 *io4 = reinterpret_cast<float4 *>(io);
    const float4 *in4 = reinterpret_cast<const float4 *>(in);
    const size_t tile = (size_t)kBlock * kUnroll;
    for (size_t base = (size_t)blockIdx.x * tile; base < n4; base += (size_t)gridDim.x * tile) {
        float4 a[kUnroll], b[kUnroll];
#pragma unroll
        for (int j = 0; j < kUnroll; ++j) {
            size_t i = base + threadIdx.x + (size_t)j * kBlock;
            if (i < n4) {
                a[j] = io4[i];
                b[j] = in4[i];
            }
        }
#pragma unroll
        for (int j = 0; j < kUnroll; ++j) {
            size_t i = base + threadIdx.x + (size_t)j * kBlock;
            if (i < n4) {
                float4 r;
                r.x = op(a[j].x, b[j].x);
                r.y = op(a[j].y, b[j].y);
                r.z = op(a[j].z, b[j].z);
                r.w = op(a[j].w, b[j].w);
                io4[i] = r;
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {  // ragged tail
        size_t i = (n4 << 2) + threadIdx.x;
        io[i] = op(io[i], in[i]);
    }
}

template <class Op>
__global__ __launch_bounds__(kBlock) void ew_inplace2_scalar_kernel(float *io, const float *in, size_t n, Op op) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock)
        io[i] = op(io[i], in[i]);
}

template <class Op>
void launch_inplace2(float *io, const float *in, size_t n, Op op, hipStream_t s) {
    if (n == 0) return;
    if (aligned16(io) && aligned16(in)) {
        hipLaunchKernelGGL(ew_inplace2_kernel<Op>, dim3(ew_grid(n >> 2)), dim3(kBlock), 0, s, io, in, n, op);
    } else {  // misaligned sub-buffer: scalar path, still coalesced
        size_t g = (n + kBlock - 1) / kBlock;
        if (g > (size_t)kMaxGrid) g = kMaxGrid;
        hipLaunchKernelGGL(ew_inplace2_scalar_kernel<Op>, dim3((int)g), dim3(kBlock), 0, s, io, in, n, op);
    }
    MIL_HIP(hipGetLastError());
}

// Complex product in place: S[i] = S[i] * O[i]; float4 = two complex64 per lane.
// multicomplex3Dkernel, reference include/cukernel.cuh:140-152.
__global__ __launch_bounds__(kBlock) void cmul_kernel(float2 *__restrict__ S, const float2 *__restrict__ O, size_t n) {
    const size_t n2 = n >> 1;
    float4 *S4 = reinterpret_cast<float4 *>(S);
    const float4 *O4 = reinterpret_cast<const float4 *>(O);
    const size_t tile = (size_t)kBlock * kUnroll;
    for (size_t base = (size_t)blockIdx.x * tile; base < n2; base += (size_t)gridDim.x * tile) {
        float4 a[kUnroll], b[kUnroll];
#pragma unroll
        for (int j = 0; j < kUnroll; ++j) {
            size_t i = base + threadIdx.x + (size_t)j * kBlock;
            if (i < n2) {
                a[j] = S4[i];
                b[j] = O4[i];
            }
        }
#pragma unroll
        for (int j = 0; j < kUnroll; ++j) {
            size_t i = base + threadIdx.x + (size_t)j * kBlock;
            if (i < n2) {
                float4 r;
                r.x = a[j].x * b[j].x - a[j].y * b[j].y;
                r.y = a[j].x * b[j].y + a[j].y * b[j].x;
                r.z = a[j].z * b[j].z - a[j].w * b[j].w;
                r.w = a[j].z * b[j].w + a[j].w * b[j].z;
                S4[i] = r;
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && (n & 1)) {
        float2 a = S[n - 1], b = O[n - 1];
        S[n - 1] = make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
    }
}

__global__ __launch_bounds__(kBlock) void cmul_scalar_kernel(float2 *S, const float2 *O, size_t n) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
        float2 a = S[i], b = O[i];
        S[i] = make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
    }
}

// Generic 1-input / up-to-2-output map used by the (cold) init kernels.
enum MapKind { kClamp, kScale, kAddValue, kFill };
template <int KIND>
__global__ __launch_bounds__(kBlock) void map_kernel(float *out, const float *in, size_t n, float c) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
        float v = (KIND == kFill) ? c : in[i];
        if (KIND == kClamp) v = (v > c) ? v : c;  // maxvalue3Dgpukernel
        if (KIND == kScale) v = v * c;            // multivaluekernel
        if (KIND == kAddValue) v = v + c;         // addvaluekernel
        out[i] = v;
    }
}

__global__ __launch_bounds__(kBlock) void init_single_kernel(float *A, float *E, size_t n, float eps) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
        float a = A[i];
        a = (a > eps) ? a : eps;
        A[i] = a;
        E[i] = a;
    }
}

__global__ __launch_bounds__(kBlock) void init_dual_kernel(float *A, float *B, float *E, size_t n, float eps) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
        float a = A[i], b = B[i];
        a = (a > eps) ? a : eps;
        b = (b > eps) ? b : eps;
        A[i] = a;
        B[i] = b;
        E[i] = (a + b) * 0.5f;  // add3Dgpu then multivaluegpu(0.5), reference src/api_subfunc.cu:3616-3617
    }
}

inline int flat_grid(size_t n) {
    size_t g = (n + kBlock - 1) / kBlock;
    if (g < 1) g = 1;
    if (g > 8192) g = 8192;
    return (int)g;
}

// ---- float64 reduction ------------------------------------------------------
constexpr int kMaxReduceBlocks = 1024;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

__device__ __forceinline__ double block_sum(double v, double *lds /* >= kBlock/64 */) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) lds[wave] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 0; w < kBlock / 64; ++w) r += lds[w];
    }
    __syncthreads();
    return r;  // valid in thread 0
}

template <int OP>
__device__ __forceinline__ double red_term(float x, float c) {
    if (OP == kSum) return (double)x;
    float v = x + c;  // addvaluegpu
    float q = v * v;  // multi3Dgpu (float32 square)
    return (double)q;
}

template <int OP>
__global__ __launch_bounds__(kBlock) void reduce_kernel(const float *__restrict__ in, size_t n, float c,
                                                      double *__restrict__ partials, int vec_ok) {
    __shared__ double lds[kBlock / 64];
    double acc = 0.0;
    const size_t n4 = vec_ok ? (n >> 2) : 0;
    const float4 *in4 = reinterpret_cast<const float4 *>(in);
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (size_t)gridDim.x * kBlock) {
        float4 v = in4[i];
        acc += red_term<OP>(v.x, c);
        acc += red_term<OP>(v.y, c);
        acc += red_term<OP>(v.z, c);
        acc += red_term<OP>(v.w, c);
    }
    for (size_t i = (n4 << 2) + (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock)
        acc += red_term<OP>(in[i], c);
    double r = block_sum(acc, lds);
    if (threadIdx.x == 0) partials[blockIdx.x] = r;
}

// Fixed-order final pass: one workgroup, `count` partials per channel.
__global__ __launch_bounds__(kBlock) void reduce_final_kernel(const double *__restrict__ partials, int count,
                                                            int channels, double *__restrict__ out) {
    __shared__ double lds[kBlock / 64];
    for (int ch = 0; ch < channels; ++ch) {
        double acc = 0.0;
        for (int i = threadIdx.x; i < count; i += kBlock) acc += partials[(size_t)ch * count + i];
        double r = block_sum(acc, lds);
        if (threadIdx.x == 0) out[ch] = r;
    }
}

// ---- geometry ---------------------------------------------------------------
struct Geo {
    int fx, fy, fz;  // output box
    int sx, sy, sz;  // input box
    int ox, oy, oz;  // offsets / origins
};

__device__ __forceinline__ bool out_xyz(const Geo &g, int &x, int &y, int &z) {
    x = blockIdx.x * blockDim.x + threadIdx.x;
    y = blockIdx.y;
    z = blockIdx.z;
    return x < g.fx;
}

// padPSFgpukernel (reference include/cukernel.cuh:680-697) turned inside out: for
// every voxel d of the F box, the PSF sample that lands there is i = d + o (front part)
// or i = d + o - F (wrapped part), o = P/2; nothing else does -> 0.
__device__ __forceinline__ int psf_src(int d, int F, int P, int o) {
    int i = d + o;
    if (i < P) return i;  // d in [0, P - o)
    i -= F;               // d in [F - o, F)
    return (i >= 0 && i < o) ? i : -1;
}

__global__ void pad_psf_kernel(float *__restrict__ out, const float *__restrict__ psf, Geo g, float scale, int flip) {
    int x, y, z;
    if (!out_xyz(g, x, y, z)) return;
    int ix = psf_src(x, g.fx, g.sx, g.ox);
    int iy = psf_src(y, g.fy, g.sy, g.oy);
    int iz = psf_src(z, g.fz, g.sz, g.oz);
    float v = 0.0f;
    if (ix >= 0 && iy >= 0 && iz >= 0) {
        if (flip) {  // flipgpukernel :668-677 folded in
            ix = g.sx - 1 - ix;
            iy = g.sy - 1 - iy;
            iz = g.sz - 1 - iz;
        }
        v = psf[(size_t)ix + (size_t)g.sx * ((size_t)iy + (size_t)g.sy * iz)] * scale;
    }
    out[(size_t)x + (size_t)g.fx * ((size_t)y + (size_t)g.fy * z)] = v;
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// padstackgpukernel :700-737 -- edge replication == clamped source index.
__global__ void pad_stack_kernel(float *__restrict__ out, const float *__restrict__ in, Geo g) {
    int x, y, z;
    if (!out_xyz(g, x, y, z)) return;
    int ix = clampi(x - g.ox, 0, g.sx - 1);
    int iy = clampi(y - g.oy, 0, g.sy - 1);
    int iz = clampi(z - g.oz, 0, g.sz - 1);
    // one pass, nothing re-read: non-temporal on both sides (the 256 MB Infinity Cache keeps what the next kernel wants)
    __builtin_nontemporal_store(__builtin_nontemporal_load(in + (size_t)ix + (size_t)g.sx * ((size_t)iy + (size_t)g.sy * iz)),
                                out + (size_t)x + (size_t)g.fx * ((size_t)y + (size_t)g.fy * z));
}

// cropgpukernel :740-753 (g.f* = cropped output box, g.s* = big input box).
__global__ void crop_kernel(float *__restrict__ out, const float *__restrict__ in, Geo g) {
    int x, y, z;
    if (!out_xyz(g, x, y, z)) return;
    int ix = x + g.ox, iy = y + g.oy, iz = z + g.oz;
    if (ix < g.sx && iy < g.sy && iz < g.sz)
        __builtin_nontemporal_store(__builtin_nontemporal_load(in + (size_t)ix + (size_t)g.sx * ((size_t)iy + (size_t)g.sy * iz)),
                                    out + (size_t)x + (size_t)g.fx * ((size_t)y + (size_t)g.fy * z));
}

// pad_stack / crop for output rows of a multiple of four voxels (every FFT box; most images): a thread moves FOUR consecutive
// output voxels -- four scalar loads (the source row starts wherever the offset puts it), one 16-byte store -- and the grid is
// what the chip holds, striding over (row, quad) items: the one-voxel kernels above launch a workgroup per half row (262 144
// of them for a 512 x 512 x 256 box, 1 KB each) and sat at 0.38 of the HBM roofline (profiles/r05/aux_pmc.json).
typedef float f32x4_nt __attribute__((ext_vector_type(4)));
typedef float f32x2_nt __attribute__((ext_vector_type(2)));
template <bool PAD>
__global__ __launch_bounds__(256) void geo_rows4_kernel(float *__restrict__ out, const float *__restrict__ in, Geo g) {
    const unsigned quads = (unsigned)g.fx / 4u;
    const unsigned long long items = (unsigned long long)quads * (unsigned)g.fy * (unsigned)g.fz;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256u + threadIdx.x; i < items; i += (unsigned long long)gridDim.x * 256u) {
        const unsigned q = (unsigned)(i % quads);
        const unsigned row = (unsigned)(i / quads);
        const int y = (int)(row % (unsigned)g.fy), z = (int)(row / (unsigned)g.fy), x = (int)(4u * q);
        f32x4_nt v;
        // four consecutive source voxels: one 16-byte or two 8-byte loads where the address allows (a 500-wide image centred in
        // a 512-wide box starts 24 bytes into the quad grid: 8-byte aligned) -- four scalar loads per thread touch every source
        // line four times (round 6: pad 0.146 -> see profiles/r06/pad_crop_vector_loads.txt)
        auto quad = [](const float *p) {
            f32x4_nt q;
            const size_t a = reinterpret_cast<size_t>(p);
            if ((a & 15) == 0) {
                q = __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt *>(p));
            } else if ((a & 7) == 0) {
                const f32x2_nt lo = __builtin_nontemporal_load(reinterpret_cast<const f32x2_nt *>(p));
                const f32x2_nt hi = __builtin_nontemporal_load(reinterpret_cast<const f32x2_nt *>(p + 2));
                q.x = lo.x; q.y = lo.y; q.z = hi.x; q.w = hi.y;
            } else {
                q.x = __builtin_nontemporal_load(p);
                q.y = __builtin_nontemporal_load(p + 1);
                q.z = __builtin_nontemporal_load(p + 2);
                q.w = __builtin_nontemporal_load(p + 3);
            }
            return q;
        };
        if (PAD) {  // padstackgpukernel :700-737: edge replication
            const int iy = clampi(y - g.oy, 0, g.sy - 1), iz = clampi(z - g.oz, 0, g.sz - 1);
            const float *src = in + (size_t)g.sx * ((size_t)iy + (size_t)g.sy * iz);
            const int x0 = x - g.ox;
            if (x0 >= 0 && x0 + 3 < g.sx) {
                v = quad(src + x0);
            } else {
                v.x = __builtin_nontemporal_load(src + clampi(x0, 0, g.sx - 1));
                v.y = __builtin_nontemporal_load(src + clampi(x0 + 1, 0, g.sx - 1));
                v.z = __builtin_nontemporal_load(src + clampi(x0 + 2, 0, g.sx - 1));
                v.w = __builtin_nontemporal_load(src + clampi(x0 + 3, 0, g.sx - 1));
            }
        } else {    // cropgpukernel :740-753 (the crop lies inside the box: host check)
            v = quad(in + (size_t)(x + g.ox) + (size_t)g.sx * ((size_t)(y + g.oy) + (size_t)g.sy * (z + g.oz)));
        }
        __builtin_nontemporal_store(v, reinterpret_cast<f32x4_nt *>(out + (size_t)x + (size_t)g.fx * ((size_t)y + (size_t)g.fy * z)));
    }
}

// pad_psf for output rows of a multiple of four voxels (every FFT box): the box is zeros but for the 2 x 2 x 2 corner blocks the
// PSF wraps into, so this is a 16-byte store per thread (rows4_grid workgroups striding over (row, quad) items) and a PSF read
// only where one lands; the one-voxel kernel wrote the 268 MB of a 512 x 512 x 256 box in 118 us (2.3 TB/s).  Plain stores:
// the forward transform reads the box next.
__global__ __launch_bounds__(256) void pad_psf_rows4_kernel(float *__restrict__ out, const float *__restrict__ psf, Geo g, float scale,
                                                          int flip) {
    const unsigned quads = (unsigned)g.fx / 4u;
    const unsigned long long items = (unsigned long long)quads * (unsigned)g.fy * (unsigned)g.fz;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256u + threadIdx.x; i < items; i += (unsigned long long)gridDim.x * 256u) {
        const unsigned q = (unsigned)(i % quads);
        const unsigned row = (unsigned)(i / quads);
        const int y = (int)(row % (unsigned)g.fy), z = (int)(row / (unsigned)g.fy), x = (int)(4u * q);
        int iy = psf_src(y, g.fy, g.sy, g.oy), iz = psf_src(z, g.fz, g.sz, g.oz);
        float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (iy >= 0 && iz >= 0) {
            if (flip) {  // flipgpukernel :668-677 folded in
                iy = g.sy - 1 - iy;
                iz = g.sz - 1 - iz;
            }
            const float *src = psf + (size_t)g.sx * ((size_t)iy + (size_t)g.sy * iz);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int ix = psf_src(x + k, g.fx, g.sx, g.ox);
                if (ix >= 0) v[k] = src[flip ? g.sx - 1 - ix : ix] * scale;
            }
        }
        *reinterpret_cast<float4 *>(out + (size_t)x + (size_t)g.fx * ((size_t)y + (size_t)g.fy * z)) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// alignsize3Dgpukernel :755-769.
__global__ void align_kernel(float *__restrict__ out, const float *__restrict__ in, Geo g) {
    int x, y, z;
    if (!out_xyz(g, x, y, z)) return;
    int ix = x - g.ox, iy = y - g.oy, iz = z - g.oz;
    float v = 0.0f;
    if (ix >= 0 && iy >= 0 && iz >= 0 && ix < g.sx && iy < g.sy && iz < g.sz)
        v = in[(size_t)ix + (size_t)g.sx * ((size_t)iy + (size_t)g.sy * iz)];
    out[(size_t)x + (size_t)g.fx * ((size_t)y + (size_t)g.fy * z)] = v;
}

__global__ void flip_kernel(float *__restrict__ out, const float *__restrict__ in, Geo g) {
    int x, y, z;
    if (!out_xyz(g, x, y, z)) return;
    out[(size_t)x + (size_t)g.fx * ((size_t)y + (size_t)g.fy * z)] =
        in[(size_t)(g.fx - 1 - x) + (size_t)g.fx * ((size_t)(g.fy - 1 - y) + (size_t)g.fy * (g.fz - 1 - z))];
}

// rotbyyaxiskernel :438-453.  Input dims (sx,sy,sz) x-fastest; output dims (sz,sy,sx).
//  +1: out[(sx-1-i)*sy*sz + j*sz + k] = in[k*sy*sx + j*sx + i]
//  -1: out[i*sy*sz + j*sz + (sz-1-k)] = in[k*sy*sx + j*sx + i]
// For one y this is a transpose of the (x, z) plane (with one axis mirrored): a 64 x 64 tile goes through LDS so that both
// the reads (consecutive i) and the writes (consecutive k) are whole 256-byte rows.  (Output-driven with strided reads, as
// the reference does it, this took 1.14 ms at 512 x 512 x 256 -- 0.06 of the HBM rate; tiled: see DESIGN.md.)
// out[z][x][y] = in[z][y][x]: 64 x 64 tiles through LDS (row pitch 65: conflict-free on both sides), 256-byte rows in and out
__global__ __launch_bounds__(256) void transpose_xy_kernel(float *__restrict__ out, const float *__restrict__ in, int nx, int ny) {
    __shared__ float tile[64 * 65];
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
    const int x0 = blockIdx.x * 64, y0 = blockIdx.y * 64;
    const size_t plane = (size_t)nx * ny * blockIdx.z;
#pragma unroll 4
    for (int r = ly; r < 64; r += 4) {
        const int x = x0 + lx, y = y0 + r;
        if (x < nx && y < ny) tile[r * 65 + lx] = in[plane + (size_t)y * nx + x];
    }
    __syncthreads();
#pragma unroll 4
    for (int r = ly; r < 64; r += 4) {
        const int y = y0 + lx, x = x0 + r;
        if (x < nx && y < ny) out[plane + (size_t)x * ny + y] = tile[lx * 65 + r];
    }
}

__global__ __launch_bounds__(256) void rot_y90_kernel(float *__restrict__ out, const float *__restrict__ in, Geo g,
                                                     int direction) {
    // g.f* = output dims (fx = in.sz, fy = in.sy, fz = in.sx); g.s* = input dims
    __shared__ float tile[64][65];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i0 = blockIdx.x * 64, j = blockIdx.y, k0 = blockIdx.z * 64;
#pragma unroll 4
    for (int kk = w; kk < 64; kk += 4) {
        const int i = i0 + lane, k = k0 + kk;
        if (i < g.sx && k < g.sz) tile[kk][lane] = __builtin_nontemporal_load(in + (size_t)i + (size_t)g.sx * ((size_t)j + (size_t)g.sy * k));
    }
    __syncthreads();
#pragma unroll 4
    for (int ii = w; ii < 64; ii += 4) {
        const int i = i0 + ii, k = k0 + lane;
        if (i < g.sx && k < g.sz) {
            const int ox = direction == 1 ? k : g.sz - 1 - k, oz = direction == 1 ? g.sx - 1 - i : i;
            __builtin_nontemporal_store(tile[lane][ii], out + (size_t)ox + (size_t)g.fx * ((size_t)j + (size_t)g.fy * oz));
        }
    }
}

// maxprojectionkernel :396-418 -- accumulator starts at 0 (negative data projects to 0).  Whatever the direction, the
// lanes of a wave read consecutive x (the reference -- and round 1 here -- put the lane index on the z or y stride for the
// projections along y and x: 122 us for one 512 x 512 x 256 projection where a coalesced pass takes ~50).
//   dir 1  along z: image [sy][sx]          thread = (x, y), loop over z
//   dir 2  along y: image [sx][sz] (w = sz) thread = (x, z), loop over y; the (small) output is written strided
//   dir 3  along x: image [sz][sy] (w = sy) one wave per volume row, lanes stride x, wave-level maximum
// max is exact and order-independent on NaN-free data, so the tree of dir 3 and the segments of dir 1 / 2 (published through
// an integer atomic max on positive values) return the reference's value.  PRECONDITION: finite, NaN-free volumes.  The
// reference's sequential `a = a > v ? a : v` lets a NaN wipe the running maximum and survive only as the LAST sample of a
// column; a segmented maximum cannot reproduce that order dependence (a NaN segment maximum is never published here), so on
// NaN input the planes differ from the reference's -- the same holds for the fused resample + projection kernels
// (affine.hip, resample_march.hip).  A deconvolution that has diverged to NaN shows up in the stack, not in its MIPs.
__global__ void max_projection_kernel(float *__restrict__ out, const float *__restrict__ in, int sx, int sy, int sz,
                                      int dir) {
    const size_t zs = (size_t)sx * sy;
    if (dir == 3) {
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        const long long row = (long long)blockIdx.x * (blockDim.x >> 6) + wave;  // row = y + sy * z
        if (row >= (long long)sy * sz) return;
        const float *p = in + (size_t)row * sx;
        float a = 0.0f;
        if ((sx & 3) == 0 && (reinterpret_cast<size_t>(in) & 15) == 0) {  // 16 bytes per lane: two loads for a 512-wide row instead of eight
            const float4 *p4 = reinterpret_cast<const float4 *>(p);
            for (int x = lane; x < sx / 4; x += 64) {
                const float4 v = p4[x];
                const float m0 = (v.x > v.y) ? v.x : v.y, m1 = (v.z > v.w) ? v.z : v.w, m = (m0 > m1) ? m0 : m1;
                a = (a > m) ? a : m;
            }
        } else {
            for (int x = lane; x < sx; x += 64) { const float v = p[x]; a = (a > v) ? a : v; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { const float v = __shfl_down(a, off, 64); a = (a > v) ? a : v; }
        if (lane == 0) out[row] = a;
        return;
    }
    // dir 1 / 2: the reduction axis is cut into gridDim.z segments that meet in an atomic maximum on the bit pattern (the
    // output starts at 0 -- the reference's accumulator -- and non-negative floats order like their patterns): four times
    // the waves in flight for a loop that is one dependent load after another
    const int x = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
    if (x >= sx) return;
    const int n = dir == 1 ? sz : sy, per = (n + (int)gridDim.z - 1) / (int)gridDim.z;
    const int k0 = (int)blockIdx.z * per, k1 = min(n, k0 + per);
    float a = 0.0f;
    if (dir == 1) {  // j = y
#pragma unroll 4
        for (int k = k0; k < k1; ++k) { const float v = in[x + (size_t)sx * j + zs * k]; a = (a > v) ? a : v; }
        if (a > 0.0f) atomicMax(reinterpret_cast<int *>(out) + x + (size_t)j * sx, __float_as_int(a));
    } else {         // dir 2: j = z, out[z + x * sz]
#pragma unroll 4
        for (int k = k0; k < k1; ++k) { const float v = in[x + (size_t)sx * k + zs * j]; a = (a > v) ? a : v; }
        if (a > 0.0f) atomicMax(reinterpret_cast<int *>(out) + j + (size_t)x * sz, __float_as_int(a));
    }
}

// circshiftgpukernel / imshiftgpukernel, reference include/cukernel.cuh:457-489: out[p] = in[p - d], circular
// (wrap != 0) or zero-filled.  g.o* carry the shift, g.f* = g.s* = the volume.
__global__ void shift_kernel(float *__restrict__ out, const float *__restrict__ in, Geo g, int wrap) {
    int x, y, z;
    if (!out_xyz(g, x, y, z)) return;
    int tx = x - g.ox, ty = y - g.oy, tz = z - g.oz;
    float v = 0.0f;
    if (wrap) {
        tx = ((tx % g.fx) + g.fx) % g.fx;
        ty = ((ty % g.fy) + g.fy) % g.fy;
        tz = ((tz % g.fz) + g.fz) % g.fz;
        v = in[(size_t)tx + (size_t)g.fx * ((size_t)ty + (size_t)g.fy * tz)];
    } else if (tx >= 0 && tx < g.fx && ty >= 0 && ty < g.fy && tz >= 0 && tz < g.fz) {
        v = in[(size_t)tx + (size_t)g.fx * ((size_t)ty + (size_t)g.fy * tz)];
    }
    out[(size_t)x + (size_t)g.fx * ((size_t)y + (size_t)g.fy * z)] = v;
}

// multicomplexnorm3Dkernel with conj3Dkernel folded in, reference include/cukernel.cuh:155-177, 209-219:
// out = conj(s1) * s2 / |conj(s1) * s2|, zero where the product vanishes.
__global__ __launch_bounds__(kBlock) void cross_power_kernel(float2 *__restrict__ out, const float2 *__restrict__ s1,
                                                           const float2 *__restrict__ s2, size_t n) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
        const float2 a = make_float2(s1[i].x, -s1[i].y), b = s2[i];
        const float c = a.x * b.x - a.y * b.y;
        const float d = a.x * b.y + a.y * b.x;
        const float e = sqrtf(c * c + d * d);
        out[i] = (e != 0.0f) ? make_float2(c / e, d / e) : make_float2(0.0f, 0.0f);
    }
}

// maxZkernel, reference include/cukernel.cuh:364-380: per (x, y) column the first maximum along z (strict <).
__global__ void argmax_z_kernel(float *__restrict__ col_max, int *__restrict__ col_arg, const float *__restrict__ in,
                                int sx, int sy, int sz) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
    if (i >= sx || j >= sy) return;
    const size_t zs = (size_t)sx * sy, base = (size_t)i + (size_t)sx * j;
    float p = in[base];
    int z0 = 0;
    for (int k = 0; k < sz; ++k) {
        const float v = in[base + zs * k];
        if (p < v) {
            p = v;
            z0 = k;
        }
    }
    col_max[base] = p;
    col_arg[base] = z0;
}

// sum of float32 products (a+ca)*(b+cb) in float64: the reference's addvaluegpu + multi3Dgpu + sum3Dgpu chain of zncc1
__global__ __launch_bounds__(kBlock) void reduce_dot_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                          size_t n, float ca, float cb, double *__restrict__ partials) {
    __shared__ double lds[kBlock / 64];
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
        const float u = a[i] + ca, v = b[i] + cb;
        const float q = u * v;
        acc += (double)q;
    }
    const double r = block_sum(acc, lds);
    if (threadIdx.x == 0) partials[blockIdx.x] = r;
}

inline void geo_launch_dims(const Geo &g, dim3 &grid, dim3 &block) {
    int bx = g.fx >= 256 ? 256 : (g.fx > 64 ? 128 : 64);
    block = dim3(bx, 1, 1);
    grid = dim3((g.fx + bx - 1) / bx, g.fy, g.fz);
}

inline Geo make_geo(Dims F, Dims S) {
    if (F.nx <= 0 || F.ny <= 0 || F.nz <= 0 || S.nx <= 0 || S.ny <= 0 || S.nz <= 0)
        throw Error(-1, "geometry kernel: empty volume");
    if (F.ny > 65535 || F.nz > 65535 || F.nx > (1LL << 30) || S.count() > (1LL << 40))
        throw Error(-1, "geometry kernel: volume extent exceeds launch limits");
    Geo g;
    g.fx = (int)F.nx; g.fy = (int)F.ny; g.fz = (int)F.nz;
    g.sx = (int)S.nx; g.sy = (int)S.ny; g.sz = (int)S.nz;
    g.ox = g.oy = g.oz = 0;
    return g;
}

inline int cdiv2(long long a) { return (int)(a / 2); }  // C truncation toward zero

}  // namespace

// 16-bit stacks on the device: what readtifstack / writetifstack do on the host (reference src/apifunc.cpp:166-183: every
// sample widened to float; :255: the C cast `(unsigned short)f`, i.e. truncation toward zero, low 16 bits of the integer --
// csrc/tiff_io.cpp writes (uint16_t)(long long)f, and so does this kernel; values beyond the long long range and NaN give
// 0 like the x86 conversion's "integer indefinite" does).
__global__ __launch_bounds__(kBlock) void u16_to_f32_kernel(float *__restrict__ out, const unsigned short *__restrict__ in, size_t n) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) out[i] = (float)in[i];
}
__global__ __launch_bounds__(kBlock) void f32_to_u16_kernel(unsigned short *__restrict__ out, const float *__restrict__ in, size_t n) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
        const float v = in[i];
        const bool fits = v > -9.2e18f && v < 9.2e18f;
        out[i] = fits ? (unsigned short)(long long)v : (unsigned short)0;
    }
}

// ---- launchers --------------------------------------------------------------
void u16_to_f32(float *out, const unsigned short *in, size_t n, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(u16_to_f32_kernel, dim3(flat_grid(n)), dim3(kBlock), 0, s, out, in, n);
    MIL_HIP(hipGetLastError());
}
void f32_to_u16(unsigned short *out, const float *in, size_t n, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(f32_to_u16_kernel, dim3(flat_grid(n)), dim3(kBlock), 0, s, out, in, n);
    MIL_HIP(hipGetLastError());
}

void rl_ratio(float *T, const float *A, size_t n, hipStream_t s) { launch_inplace2(T, A, n, RatioOp{}, s); }

void rl_update(float *E, const float *T, size_t n, float eps, hipStream_t s) {
    launch_inplace2(E, T, n, UpdateOp{eps}, s);
}

void cmul_inplace(float2 *S, const float2 *OTF, size_t n, hipStream_t s) {
    if (n == 0) return;
    if (aligned16(S) && aligned16(OTF)) {
        hipLaunchKernelGGL(cmul_kernel, dim3(ew_grid(n >> 1)), dim3(kBlock), 0, s, S, OTF, n);
    } else {
        hipLaunchKernelGGL(cmul_scalar_kernel, dim3(flat_grid(n)), dim3(kBlock), 0, s, S, OTF, n);
    }
    MIL_HIP(hipGetLastError());
}

void clamp_min(float *out, const float *in, size_t n, float eps, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(map_kernel<kClamp>, dim3(flat_grid(n)), dim3(kBlock), 0, s, out, in, n, eps);
    MIL_HIP(hipGetLastError());
}
void scale(float *out, const float *in, float c, size_t n, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(map_kernel<kScale>, dim3(flat_grid(n)), dim3(kBlock), 0, s, out, in, n, c);
    MIL_HIP(hipGetLastError());
}
void add_value(float *out, const float *in, float c, size_t n, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(map_kernel<kAddValue>, dim3(flat_grid(n)), dim3(kBlock), 0, s, out, in, n, c);
    MIL_HIP(hipGetLastError());
}
void fill(float *out, float v, size_t n, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(map_kernel<kFill>, dim3(flat_grid(n)), dim3(kBlock), 0, s, out, (const float *)nullptr, n, v);
    MIL_HIP(hipGetLastError());
}
void init_single(float *A, float *E, size_t n, float eps, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(init_single_kernel, dim3(flat_grid(n)), dim3(kBlock), 0, s, A, E, n, eps);
    MIL_HIP(hipGetLastError());
}
void init_dual(float *A, float *B, float *E, size_t n, float eps, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(init_dual_kernel, dim3(flat_grid(n)), dim3(kBlock), 0, s, A, B, E, n, eps);
    MIL_HIP(hipGetLastError());
}

ReduceWorkspace::ReduceWorkspace() {
    partials.alloc(2 * kMaxReduceBlocks + 2);
    MIL_HIP(hipHostMalloc((void **)&host, 2 * sizeof(double), hipHostMallocDefault));
}
ReduceWorkspace::~ReduceWorkspace() {
    if (host) (void)hipHostFree(host);
}

double reduce_sync(ReduceOp op, const float *in, size_t n, float c, ReduceWorkspace &ws, hipStream_t s) {
    if (n == 0) return 0.0;
    size_t want = (n / 4 + kBlock - 1) / kBlock;
    int blocks = (int)(want < 1 ? 1 : (want > (size_t)kMaxReduceBlocks ? kMaxReduceBlocks : want));
    int vec_ok = aligned16(in) ? 1 : 0;
    double *out = ws.partials.get() + 2 * kMaxReduceBlocks;
    if (op == kSum)
        hipLaunchKernelGGL(reduce_kernel<kSum>, dim3(blocks), dim3(kBlock), 0, s, in, n, c, ws.partials.get(), vec_ok);
    else
        hipLaunchKernelGGL(reduce_kernel<kSumSqShifted>, dim3(blocks), dim3(kBlock), 0, s, in, n, c,
                           ws.partials.get(), vec_ok);
    MIL_HIP(hipGetLastError());
    hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(kBlock), 0, s, ws.partials.get(), blocks, 1, out);
    MIL_HIP(hipGetLastError());
    MIL_HIP(hipMemcpyAsync(ws.host, out, sizeof(double), hipMemcpyDeviceToHost, s));
    MIL_HIP(hipStreamSynchronize(s));
    return ws.host[0];
}

// workgroups of geo_rows4_kernel: 16 per compute unit of a 256-CU part, fewer for small boxes
static unsigned rows4_grid(const Geo &g) {
    const unsigned long long items = (unsigned long long)(g.fx / 4) * g.fy * g.fz;
    return (unsigned)std::max<unsigned long long>(1, std::min<unsigned long long>(4096, (items + 255) / 256));
}

void pad_psf(float *out, const float *psf, Dims F, Dims P, float scale_v, bool flip, hipStream_t s) {
    if (P.nx > F.nx || P.ny > F.ny || P.nz > F.nz) throw Error(-1, "pad_psf: PSF larger than FFT box");
    Geo g = make_geo(F, P);
    g.ox = (int)(P.nx / 2); g.oy = (int)(P.ny / 2); g.oz = (int)(P.nz / 2);  // reference src/api_subfunc.cu:1695
    if (g.fx % 4 == 0 && (reinterpret_cast<size_t>(out) & 15) == 0) {
        hipLaunchKernelGGL(pad_psf_rows4_kernel, dim3(rows4_grid(g)), dim3(256), 0, s, out, psf, g, scale_v, flip ? 1 : 0);
        MIL_HIP(hipGetLastError());
        return;
    }
    dim3 grid, block;
    geo_launch_dims(g, grid, block);
    hipLaunchKernelGGL(pad_psf_kernel, grid, block, 0, s, out, psf, g, scale_v, flip ? 1 : 0);
    MIL_HIP(hipGetLastError());
}

void pad_stack(float *out, const float *in, Dims F, Dims S, hipStream_t s) {
    if (S.nx > F.nx || S.ny > F.ny || S.nz > F.nz) throw Error(-1, "pad_stack: image larger than FFT box");
    Geo g = make_geo(F, S);
    g.ox = (int)((F.nx - S.nx) / 2); g.oy = (int)((F.ny - S.ny) / 2); g.oz = (int)((F.nz - S.nz) / 2);  // :1717-1719
    dim3 grid, block;
    geo_launch_dims(g, grid, block);
    if (g.fx % 4 == 0 && (reinterpret_cast<size_t>(out) & 15) == 0) {
        hipLaunchKernelGGL(geo_rows4_kernel<true>, dim3(rows4_grid(g)), dim3(256), 0, s, out, in, g);
        MIL_HIP(hipGetLastError());
        return;
    }
    hipLaunchKernelGGL(pad_stack_kernel, grid, block, 0, s, out, in, g);
    MIL_HIP(hipGetLastError());
}

void crop_stack(float *out, const float *in, Dims S, Dims F, hipStream_t s) {
    if (S.nx > F.nx || S.ny > F.ny || S.nz > F.nz) throw Error(-1, "crop_stack: crop larger than input");
    Geo g = make_geo(S, F);
    g.ox = (int)((F.nx - S.nx) / 2); g.oy = (int)((F.ny - S.ny) / 2); g.oz = (int)((F.nz - S.nz) / 2);  // :1740-1742
    dim3 grid, block;
    geo_launch_dims(g, grid, block);
    if (g.fx % 4 == 0 && (reinterpret_cast<size_t>(out) & 15) == 0) {
        hipLaunchKernelGGL(geo_rows4_kernel<false>, dim3(rows4_grid(g)), dim3(256), 0, s, out, in, g);
        MIL_HIP(hipGetLastError());
        return;
    }
    hipLaunchKernelGGL(crop_kernel, grid, block, 0, s, out, in, g);
    MIL_HIP(hipGetLastError());
}

void align_size(float *out, const float *in, Dims O, Dims I, hipStream_t s) {
    Geo g = make_geo(O, I);
    g.ox = cdiv2(O.nx - I.nx); g.oy = cdiv2(O.ny - I.ny); g.oz = cdiv2(O.nz - I.nz);  // :1783-1785
    dim3 grid, block;
    geo_launch_dims(g, grid, block);
    hipLaunchKernelGGL(align_kernel, grid, block, 0, s, out, in, g);
    MIL_HIP(hipGetLastError());
}

void flip3(float *out, const float *in, Dims d, hipStream_t s) {
    Geo g = make_geo(d, d);
    dim3 grid, block;
    geo_launch_dims(g, grid, block);
    hipLaunchKernelGGL(flip_kernel, grid, block, 0, s, out, in, g);
    MIL_HIP(hipGetLastError());
}

void transpose_xy(float *out, const float *in, Dims d, hipStream_t s) {
    if (d.count() == 0) return;
    const dim3 grid((unsigned)((d.nx + 63) / 64), (unsigned)((d.ny + 63) / 64), (unsigned)d.nz);
    if (d.nz > 65535 || grid.y > 65535) throw Error(-1, "transpose_xy: volume exceeds launch limits");
    hipLaunchKernelGGL(transpose_xy_kernel, grid, dim3(256), 0, s, out, in, (int)d.nx, (int)d.ny);
    MIL_HIP(hipGetLastError());
}

void rot_y90(float *out, const float *in, Dims in_dims, int direction, hipStream_t s) {
    if (direction != 1 && direction != -1) throw Error(-1, "rot_y90: direction must be +1 or -1");
    Dims od{in_dims.nz, in_dims.ny, in_dims.nx};
    Geo g = make_geo(od, in_dims);
    const dim3 grid((unsigned)((in_dims.nx + 63) / 64), (unsigned)in_dims.ny, (unsigned)((in_dims.nz + 63) / 64));
    if (grid.y > 65535 || grid.z > 65535) throw Error(-1, "rot_y90: volume exceeds launch limits");
    hipLaunchKernelGGL(rot_y90_kernel, grid, dim3(256), 0, s, out, in, g, direction);
    MIL_HIP(hipGetLastError());
}

void max_projection(float *out, const float *in, Dims d, int direction, hipStream_t s) {
    // image extents per direction: reference src/api_subfunc.cu:492-501 (1: sx x sy, 2: sz x sx, 3: sy x sz)
    const int sx = (int)d.nx, sy = (int)d.ny, sz = (int)d.nz;
    if (direction < 1 || direction > 3) throw Error(-1, "max_projection: direction must be 1, 2 or 3");
    if (direction == 3) {
        const long long rows = (long long)sy * sz;
        hipLaunchKernelGGL(max_projection_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, out, in, sx, sy, sz, 3);
    } else {
        const int bx = sx >= 256 ? 256 : 64;
        const int n = direction == 1 ? sz : sy, segments = n >= 128 ? 4 : 1;
        MIL_HIP(hipMemsetAsync(out, 0, (size_t)sx * (direction == 1 ? sy : sz) * sizeof(float), s));
        hipLaunchKernelGGL(max_projection_kernel, dim3((sx + bx - 1) / bx, direction == 1 ? sy : sz, segments), dim3(bx), 0, s,
                           out, in, sx, sy, sz, direction);
    }
    MIL_HIP(hipGetLastError());
}

void shift_volume(float *out, const float *in, Dims d, long long dx, long long dy, long long dz, bool wrap,
                  hipStream_t s) {
    Geo g = make_geo(d, d);
    g.ox = (int)dx; g.oy = (int)dy; g.oz = (int)dz;
    dim3 grid, block;
    geo_launch_dims(g, grid, block);
    hipLaunchKernelGGL(shift_kernel, grid, block, 0, s, out, in, g, wrap ? 1 : 0);
    MIL_HIP(hipGetLastError());
}

void crop_offset(float *out, const float *in, Dims S, Dims F, long long ox, long long oy, long long oz, hipStream_t s) {
    if (ox < 0 || oy < 0 || oz < 0 || ox + S.nx > F.nx || oy + S.ny > F.ny || oz + S.nz > F.nz)
        throw Error(-1, "crop_offset: crop outside the volume");
    Geo g = make_geo(S, F);
    g.ox = (int)ox; g.oy = (int)oy; g.oz = (int)oz;
    dim3 grid, block;
    geo_launch_dims(g, grid, block);
    hipLaunchKernelGGL(crop_kernel, grid, block, 0, s, out, in, g);
    MIL_HIP(hipGetLastError());
}

void cross_power_norm(float2 *out, const float2 *s1, const float2 *s2, size_t n, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(cross_power_kernel, dim3(flat_grid(n)), dim3(kBlock), 0, s, out, s1, s2, n);
    MIL_HIP(hipGetLastError());
}

void argmax_columns(float *col_max, int *col_arg, const float *in, Dims d, hipStream_t s) {
    const int bx = d.nx >= 256 ? 256 : 64;
    hipLaunchKernelGGL(argmax_z_kernel, dim3((unsigned)((d.nx + bx - 1) / bx), (unsigned)d.ny, 1), dim3(bx), 0, s, col_max,
                       col_arg, in, (int)d.nx, (int)d.ny, (int)d.nz);
    MIL_HIP(hipGetLastError());
}

double reduce_dot_sync(const float *a, const float *b, size_t n, float ca, float cb, ReduceWorkspace &ws, hipStream_t s) {
    if (n == 0) return 0.0;
    size_t want = (n + kBlock - 1) / kBlock;
    const int blocks = (int)(want > (size_t)kMaxReduceBlocks ? kMaxReduceBlocks : want);
    double *out = ws.partials.get() + 2 * kMaxReduceBlocks;
    hipLaunchKernelGGL(reduce_dot_kernel, dim3(blocks), dim3(kBlock), 0, s, a, b, n, ca, cb, ws.partials.get());
    MIL_HIP(hipGetLastError());
    hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(kBlock), 0, s, ws.partials.get(), blocks, 1, out);
    MIL_HIP(hipGetLastError());
    MIL_HIP(hipMemcpyAsync(ws.host, out, sizeof(double), hipMemcpyDeviceToHost, s));
    MIL_HIP(hipStreamSynchronize(s));
    return ws.host[0];
}

}  // namespace k
}  // namespace mil

// What does the registration cost kernel's MEMORY ACCESS PATTERN cost on its own?  The same tiles (128 x, 4 waves = 4 y, 16 z
// steps), the same loads per pair-step at the identity matrix -- two 4-byte target loads (x, x + 64) and four 8-byte gathers
// of the new upper source plane (rows y0, y1 for both voxels), software-pipelined one step ahead -- and NO interpolation: the
// loaded words are xor-ed together.  Variants: the marching axis (z as kernels 0-5, y as kernel 6), waves per SIMD, and a
// streaming read of the same 2 R for reference.  DESIGN.md section 4 (round 3).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Wno-unused-value -o build/probe_corr_mem tools/probe_corr_mem.hip && build/probe_corr_mem
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                  \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            printf("%s: %s\n", #x, hipGetErrorString(e_));                        \
            return 1;                                                             \
        }                                                                         \
    } while (0)

constexpr int NX = 512, NY = 512, NZ = 256;
constexpr int PITCH = 516;  // padded source row (sx + 2 rounded up to 4 floats)
constexpr int PY = NY + 2, PZ = NZ + 2;

// MARCHY 0: 16 steps along z at fixed y | 1: 16 steps along y at fixed z.  VALU = dummy dependent VALU instructions per step
// (0: the pattern alone).
template <int MARCHY, int VALU, int WAVES_PER_EU, int ORDER = 0, int PARTS = 7, int TAIL = 0, int DWORDS = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES_PER_EU, WAVES_PER_EU)))
void pattern(const float *__restrict__ tgt, const float *__restrict__ src, unsigned *out, double *partials = nullptr,
             unsigned *counters = nullptr) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gx = NX / 128, gy = (MARCHY ? NZ : NY) / 4;
    // ORDER 1: every XCD (workgroups are dealt round-robin over the 8) takes a CONTIGUOUS eighth of the tiles
    const int tile = ORDER == 1 ? (int)(blockIdx.x % 8) * (int)(gridDim.x / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
    const int bx = tile % gx, by = (tile / gx) % gy, bz = tile / (gx * gy);
    const int x = bx * 128 + lane, fixed = by * 4 + wave, s0 = bz * 16;
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0, PITCH * PY * PZ * 4, 0x00020000);
    const unsigned row = PITCH * 4, plane = PITCH * PY * 4;
    const unsigned sStep = MARCHY ? row : plane, sSide = MARCHY ? plane : row;
    const int y0 = MARCHY ? s0 : fixed, z0 = MARCHY ? fixed : s0;
    unsigned offA = (unsigned)(z0 * (int)plane + y0 * (int)row + x * 4), offB = offA + 256u;
    unsigned toff = 4u * ((unsigned)x + NX * ((unsigned)y0 + NY * (unsigned)z0));
    const char *tb = reinterpret_cast<const char *>(tgt);
    unsigned acc = 0;
    const unsigned tstep = MARCHY ? 4u * NX : 4u * NX * NY;
        // PARTS: 1 = the target loads, 2 = the gathers of side 0 (row y0 / plane z0), 4 = those of side 1
    auto g = [&](unsigned off, unsigned so) {
        const bool side1 = so != sStep;
        if (!(PARTS & (side1 ? 4 : 2))) return u32x2{off, so};
        if (DWORDS) {  // the (x0, x1) pair as two 4-byte gathers (the +4 hidden from the compiler, which would fuse them again)
            unsigned four = 4u;
            asm volatile("" : "+s"(four));
            return u32x2{__builtin_amdgcn_raw_buffer_load_b32(rs, (int)off, (int)so, 0),
                         __builtin_amdgcn_raw_buffer_load_b32(rs, (int)off, (int)(so + four), 0)};
        }
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)off, (int)so, 0);
        return v;
    };
    auto tl = [&](unsigned o) { return (PARTS & 1) ? *reinterpret_cast<const unsigned *>(tb + o) : o; };
    u32x2 a0 = g(offA, sStep), a1 = g(offA, sStep + sSide), b0 = g(offB, sStep), b1 = g(offB, sStep + sSide);
    unsigned ta = tl(toff), tbv = tl(toff + 256);
#pragma unroll 3
    for (int s = 1; s < 16; ++s) {
        offA += sStep; offB += sStep; toff += tstep;
        const u32x2 n0 = g(offA, sStep), n1 = g(offA, sStep + sSide), m0 = g(offB, sStep), m1 = g(offB, sStep + sSide);
        const unsigned na = tl(toff), nb = tl(toff + 256);
        unsigned v = a0.x ^ a0.y ^ a1.x ^ a1.y ^ b0.x ^ b0.y ^ b1.x ^ b1.y ^ ta ^ tbv;
        float f = __uint_as_float(v & 0x3fffffffu);
#pragma unroll
        for (int k = 0; k < VALU / 2; ++k) f = f * 1.0001f + 0.5f;  // v_mul_f32 + v_add_f32 (no contraction, no folding): two VALU each
        acc ^= __float_as_uint(f);
        a0 = n0; a1 = n1; b0 = m0; b1 = m1; ta = na; tbv = nb;
    }
    acc ^= a0.x ^ a0.y ^ a1.x ^ a1.y ^ b0.x ^ b0.y ^ b1.x ^ b1.y ^ ta ^ tbv;
    if (acc == 0x12345678u) out[threadIdx.x] = acc;
    if constexpr (TAIL != 0) {
        // The cost kernels' epilogue: four float64 wave sums, LDS, then thread 0 publishes the workgroup's partials at agent
        // scope -- store, wait for the acknowledgement, count -- TAIL 1: twice in a row (the two virtual workgroups of the
        // two-voxel kernels), 2: both in one round (four stores, one wait, two counts in flight together), 3: as 2, and the
        // waves that have nothing to publish leave before the round trips.
        __shared__ double lds[16];
        __shared__ int flag;
        double v[4] = {(double)acc, (double)(acc >> 3), (double)(acc >> 5), (double)(acc >> 7)};
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_down(v[k], off, 64);
        if (lane == 0)
#pragma unroll
            for (int k = 0; k < 4; ++k) lds[4 * k + wave] = v[k];
        __syncthreads();
        if (TAIL == 3 && wave != 0) return;
        const int b = 2 * (int)blockIdx.x, nb = 2 * (int)gridDim.x;
        auto st = [&](double *p_, double x_) { __hip_atomic_store(p_, x_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
        auto cnt = [&](unsigned *c_) { return __hip_atomic_fetch_add(c_, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
        if (threadIdx.x == 0) {
            double s[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) s[k] = lds[4 * k] + lds[4 * k + 1] + lds[4 * k + 2] + lds[4 * k + 3];
            if (TAIL == 1) {
                st(partials + b, s[0]); st(partials + nb + b, s[1]);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                flag = cnt(counters + b % 64) == 0xffffffffu;
            } else {
                st(partials + b, s[0]); st(partials + nb + b, s[1]); st(partials + b + 1, s[2]); st(partials + nb + b + 1, s[3]);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const unsigned c0 = cnt(counters + b % 64), c1 = cnt(counters + (b + 1) % 64);
                flag = (c0 == 0xffffffffu) | (c1 == 0xffffffffu);
            }
        }
        if (TAIL != 3) __syncthreads();
        if (flag) out[1] = 1;
        if (TAIL == 1) {
            if (threadIdx.x == 0) {
                st(partials + b + 1, lds[8]); st(partials + nb + b + 1, lds[12]);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                flag = cnt(counters + (b + 1) % 64) == 0xffffffffu;
            }
            __syncthreads();
            if (flag) out[2] = 1;
        }
    }
}


// What would lanes that own y-ADJACENT voxels cost?  A lane owns NV voxels (x, y .. y + NV - 1); per step it loads NV target
// words and gathers the NV + 1 rows y .. y + NV of the new upper plane (the middle rows serve two voxels each): 1 + 1/NV gathers
// per voxel instead of 2.  Tiles: 64 x, 4 waves x NV rows, 16 z steps.
template <int NV>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
void pattern_ypairs(const float *__restrict__ tgt, const float *__restrict__ src, unsigned *out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gx = NX / 64, gy = NY / (4 * NV);
    const int bx = blockIdx.x % gx, by = (blockIdx.x / gx) % gy, bz = blockIdx.x / (gx * gy);
    const int x = bx * 64 + lane, y0 = (by * 4 + wave) * NV, z0 = bz * 16;
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0, PITCH * PY * PZ * 4, 0x00020000);
    const unsigned row = PITCH * 4, plane = PITCH * PY * 4;
    unsigned off = (unsigned)(z0 * (int)plane + y0 * (int)row + x * 4);
    unsigned toff = 4u * ((unsigned)x + NX * ((unsigned)y0 + NY * (unsigned)z0));
    const char *tb = reinterpret_cast<const char *>(tgt);
    unsigned acc = 0;
    u32x2 g[NV + 1], n[NV + 1];
    unsigned t[NV], tn[NV];
#pragma unroll
    for (int r = 0; r <= NV; ++r) g[r] = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)off, (int)(plane + r * row), 0);
#pragma unroll
    for (int r = 0; r < NV; ++r) t[r] = *reinterpret_cast<const unsigned *>(tb + toff + r * 4u * NX);
#pragma unroll 2
    for (int s = 1; s < 16; ++s) {
        off += plane; toff += 4u * NX * NY;
#pragma unroll
        for (int r = 0; r <= NV; ++r) n[r] = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)off, (int)(plane + r * row), 0);
#pragma unroll
        for (int r = 0; r < NV; ++r) tn[r] = *reinterpret_cast<const unsigned *>(tb + toff + r * 4u * NX);
#pragma unroll
        for (int r = 0; r <= NV; ++r) acc ^= g[r].x ^ g[r].y;
#pragma unroll
        for (int r = 0; r < NV; ++r) acc ^= t[r];
#pragma unroll
        for (int r = 0; r <= NV; ++r) g[r] = n[r];
#pragma unroll
        for (int r = 0; r < NV; ++r) t[r] = tn[r];
    }
#pragma unroll
    for (int r = 0; r <= NV; ++r) acc ^= g[r].x ^ g[r].y;
#pragma unroll
    for (int r = 0; r < NV; ++r) acc ^= t[r];
    if (acc == 0x12345678u) out[threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void stream(const u32x4 *__restrict__ a, size_t n, unsigned *out) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const u32x4 v = a[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[threadIdx.x] = acc;
}


// VALU calibration: what does one wave-wide instruction of each kind cost?  The cost kernel's grid and step structure, no
// loads, 128 instructions of ONE kind per step (two independent chains).
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int OP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) void valu(unsigned *out, float seed) {
    float a = seed + threadIdx.x, b = seed * 3.f + threadIdx.x;
    f32x2 pa = {a, b}, pb = {b, a};
    double da = a, db = b;
    unsigned ua = threadIdx.x * 7u + 1u, ub = threadIdx.x * 13u + 5u;
    for (int s = 0; s < 16; ++s) {
#pragma unroll
        for (int k = 0; k < 64; ++k) {
            if (OP == 0) { a = a * 1.0001f; b = b * 0.9999f; }                       // v_mul_f32
            if (OP == 1) { a = a + 1.5f; b = b + 2.5f; }                            // v_add_f32
            if (OP == 2) { pa = pa * (f32x2){1.0001f, 0.9999f}; pb = pb * (f32x2){0.9999f, 1.0001f}; }  // v_pk_mul_f32
            if (OP == 3) { pa = pa + (f32x2){1.5f, 2.5f}; pb = pb + (f32x2){2.5f, 1.5f}; }              // v_pk_add_f32
            if (OP == 4) { da = __builtin_fma(da, 1.0001, 0.5); db = __builtin_fma(db, 0.9999, 0.25); } // v_fma_f64
            if (OP == 5) { da = da + (double)a; a = a + 1.0f; }                     // v_cvt_f64_f32 + v_add_f64 + v_add_f32
            if (OP == 6) { a = floorf(a) + 0.5f; b = floorf(b) + 1.5f; }            // v_floor_f32 + v_add_f32
            if (OP == 7) { asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(ua) : "s"(s)); asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(ub) : "s"(s)); }
            if (OP == 8) { da = da * 1.0001; db = db * 0.9999; }                    // v_mul_f64
            if (OP == 9) { da = da + 0.5; db = db + 0.25; }                         // v_add_f64
            if (OP == 10) { ua = __builtin_amdgcn_ubfe(ua, 3, 7) + ub; ub = ub ^ ua; }  // integer: bfe / add / xor
        }
    }
    const double r = da + db + a + b + pa.x + pa.y + pb.x + pb.y + ua + ub;
    if (r == 123456.789) out[threadIdx.x] = (unsigned)r;
}

template <typename F>
static float time_us(F launch, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0, nullptr);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1, nullptr);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.f / reps;
}

int main() {
    const size_t nt = (size_t)NX * NY * NZ, ns = (size_t)PITCH * PY * PZ + 64;
    float *tgt, *src;
    unsigned *out;
    CHECK(hipMalloc(&tgt, nt * 4));
    CHECK(hipMalloc(&src, ns * 4));
    CHECK(hipMalloc(&out, 4096));
    CHECK(hipMemset(tgt, 1, nt * 4));
    CHECK(hipMemset(src, 2, ns * 4));
    const int grid = (NX / 128) * (NY / 4) * (NZ / 16);
    const double bytes = 2.0 * nt * 4;
#define RUN(M, V, W)                                                                                                   \
    {                                                                                                                  \
        const float us = time_us([&] { hipLaunchKernelGGL((pattern<M, V, W>), dim3(grid), dim3(256), 0, nullptr, tgt, src, out); }, 20); \
        printf("pattern march-%s, %3d dummy VALU per pair-step, %d waves/SIMD: %7.1f us  (%.2f TB/s on 2 R)\n", M ? "y" : "z", V, W, us, \
               bytes / us / 1e6);                                                                                      \
    }
    RUN(0, 0, 5) RUN(0, 0, 8) RUN(0, 0, 3) RUN(1, 0, 5)
    RUN(0, 40, 5) RUN(0, 80, 5) RUN(0, 120, 5) RUN(0, 160, 5) RUN(0, 80, 8)
#define RUNP(P, LABEL)                                                                                                 \
    {                                                                                                                  \
        const float us = time_us([&] { hipLaunchKernelGGL((pattern<0, 0, 5, 0, P>), dim3(grid), dim3(256), 0, nullptr, tgt, src, out); }, 20); \
        printf("march-z, only %s: %7.1f us\n", LABEL, us);                                                            \
    }
    RUNP(1, "the target loads (R)") RUNP(6, "the four gathers (R unique)") RUNP(2, "the gathers of row y0") RUNP(3, "target + gathers of row y0")
    {
        const float us = time_us([&] { hipLaunchKernelGGL((pattern<0, 160, 5, 0, 0>), dim3(grid), dim3(256), 0, nullptr, tgt, src, out); }, 20);
        printf("no loads at all, 160 dummy VALU per pair-step (VALU calibration): %7.1f us\n", us);
        const float us2 = time_us([&] { hipLaunchKernelGGL((pattern<0, 320, 5, 0, 0>), dim3(grid), dim3(256), 0, nullptr, tgt, src, out); }, 20);
        printf("no loads at all, 320 dummy VALU per pair-step: %7.1f us\n", us2);
        const float us3 = time_us([&] { hipLaunchKernelGGL((pattern<0, 320, 5, 0, 7>), dim3(grid), dim3(256), 0, nullptr, tgt, src, out); }, 20);
        printf("all loads, 320 dummy VALU per pair-step: %7.1f us\n", us3);
        const float us4 = time_us([&] { hipLaunchKernelGGL((pattern<0, 240, 5, 0, 7>), dim3(grid), dim3(256), 0, nullptr, tgt, src, out); }, 20);
        printf("all loads, 240 dummy VALU per pair-step: %7.1f us\n", us4);
    }
    {
        const float us = time_us([&] { hipLaunchKernelGGL((pattern<0, 0, 5, 1>), dim3(grid), dim3(256), 0, nullptr, tgt, src, out); }, 20);
        printf("XCD-contiguous tiles, march-z: %7.1f us\n", us);
    }
    {
        double *partials; unsigned *counters;
        CHECK(hipMalloc(&partials, (size_t)4 * grid * 8 + 1024));
        CHECK(hipMalloc(&counters, 1024));
        CHECK(hipMemset(counters, 0, 1024));
#define RUNT(T, LABEL)                                                                                                 \
        {                                                                                                              \
            const float us = time_us([&] { hipLaunchKernelGGL((pattern<0, 160, 5, 0, 7, T>), dim3(grid), dim3(256), 0, nullptr, tgt, src, out, partials, counters); }, 20); \
            printf("all loads, 160 dummy VALU, epilogue %-70s: %7.1f us\n", LABEL, us);                                \
        }
        RUNT(0, "none") RUNT(1, "as the cost kernels: two publishes in a row") RUNT(2, "one round: four stores, one wait, two counts")
        RUNT(3, "one round, waves 1-3 leave before it") RUNT(0, "none") RUNT(1, "as the cost kernels: two publishes in a row")
    }
    {
        const float us = time_us([&] { hipLaunchKernelGGL((pattern<0, 0, 5, 0, 7, 0, 1>), dim3(grid), dim3(256), 0, nullptr, tgt, src, out, nullptr, nullptr); }, 20);
        printf("march-z, every 8-byte gather as two 4-byte gathers: %7.1f us\n", us);
        const float us6 = time_us([&] { hipLaunchKernelGGL((pattern<0, 0, 5, 0, 6, 0, 1>), dim3(grid), dim3(256), 0, nullptr, tgt, src, out, nullptr, nullptr); }, 20);
        printf("march-z, only the gathers, as 4-byte gathers: %7.1f us\n", us6);
    }
    {
        const float u2 = time_us([&] { hipLaunchKernelGGL((pattern_ypairs<2>), dim3((NX / 64) * (NY / 8) * (NZ / 16)), dim3(256), 0, nullptr, tgt, src, out); }, 20);
        printf("lanes own 2 y-adjacent voxels (3 gathers per 2 voxels): %7.1f us\n", u2);
        const float u4 = time_us([&] { hipLaunchKernelGGL((pattern_ypairs<4>), dim3((NX / 64) * (NY / 16) * (NZ / 16)), dim3(256), 0, nullptr, tgt, src, out); }, 20);
        printf("lanes own 4 y-adjacent voxels (5 gathers per 4 voxels): %7.1f us\n", u4);
        const float u1 = time_us([&] { hipLaunchKernelGGL((pattern_ypairs<1>), dim3((NX / 64) * (NY / 4) * (NZ / 16)), dim3(256), 0, nullptr, tgt, src, out); }, 20);
        printf("lanes own 1 voxel (2 gathers per voxel, 64-wide tiles):    %7.1f us\n", u1);
    }
#define RUNV(OP, LABEL, PER)                                                                                           \
    {                                                                                                                  \
        const float us = time_us([&] { hipLaunchKernelGGL((valu<OP>), dim3(grid), dim3(256), 0, nullptr, out, 1.25f); }, 10); \
        printf("VALU calibration, 128 x %-44s per step: %7.1f us  = %.2f ns per instruction and SIMD\n", LABEL, us,      \
               us * 1e3 / (128.0 * PER * 16 * 32));                                                                    \
    }
    RUNV(0, "v_mul_f32", 1) RUNV(1, "v_add_f32", 1) RUNV(2, "v_pk_mul_f32", 1) RUNV(3, "v_pk_add_f32", 1) RUNV(4, "v_fma_f64", 1)
    RUNV(8, "v_mul_f64", 1) RUNV(9, "v_add_f64", 1) RUNV(5, "(v_cvt_f64_f32 + v_add_f64 + v_add_f32) / 2", 1.5)
    RUNV(6, "(v_floor_f32 + v_add_f32)", 2) RUNV(7, "v_mad_u32_u24", 1) RUNV(10, "(v_bfe_u32 + v_add_u32 + v_xor) / 1.5", 1.5)
    {
        const size_t n16 = (nt + ns) / 4;
        // one buffer is enough for the reference figure: stream tgt then src
        const float us = time_us([&] {
            hipLaunchKernelGGL(stream, dim3(256 * 8), dim3(256), 0, nullptr, reinterpret_cast<const u32x4 *>(tgt), nt / 4, out);
            hipLaunchKernelGGL(stream, dim3(256 * 8), dim3(256), 0, nullptr, reinterpret_cast<const u32x4 *>(src), ns / 4, out);
        }, 20);
        printf("streaming read of both volumes (%zu x 16 B): %7.1f us  (%.2f TB/s)\n", n16, us, (nt + ns) * 4.0 / us / 1e6);
    }
    CHECK(hipDeviceSynchronize());
    return 0;
}

// Does a BRICKED source layout pay for the registration cost kernel?  (Round-3 review: "8x x 4y x 4z floats = one 512-byte brick
// of four 128-byte lines ... so that the taps of a wave land in 1-2 bricks whatever the rotation axis".)
// The access pattern of a z-marching cost evaluation for a REAL matrix, without the interpolation: per target voxel the floor of
// its source coordinate, the 4-byte target load and the gathers of the tap rows (y0, y1) of the new upper source plane -- the
// lower plane is re-gathered wave-wide when some lane did not advance by exactly one plane, as in corr_padded.hip -- xor-ed
// together.  Layouts of the border-padded source:
//   plain   [z][y][x], 8-byte gathers (x0, x1)                                        (kernels 0 / 3 of the library)
//   brick   8 x 4 x 4 bricks (x fastest inside a brick, bricks x-fastest), 4-byte gathers: a pair may straddle two bricks
// Lanes: 64 consecutive x of one (y, z0 .. z0 + 15), or an 8 x 8 patch of (x, y) -- the patch is what could make a brick layout
// pay (a wave's taps in few bricks).  Matrices: identity, 3 deg about y + 2 %, 10 deg, 45 deg.
//   hipcc --offload-arch=gfx950 -O3 -o build/probe_brick tools/probe_brick.hip && build/probe_brick
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>

#define CHECK(x)                                                   \
    do {                                                           \
        hipError_t e_ = (x);                                       \
        if (e_ != hipSuccess) {                                    \
            printf("%s: %s\n", #x, hipGetErrorString(e_));         \
            return 1;                                              \
        }                                                          \
    } while (0)

constexpr int NX = 512, NY = 512, NZ = 256;
constexpr int PX = 520, PY = 516, PZ = 260;  // padded extents, multiples of the brick (8, 4, 4)
constexpr int BXN = PX / 8, BYN = PY / 4;

struct Mat {
    float m[12];
};

template <int LAYOUT>
__device__ __forceinline__ unsigned src_off(int x, int y, int z) {  // float index of padded voxel (x + 1, y + 1, z + 1)
    x += 1; y += 1; z += 1;
    if (LAYOUT == 0) return (unsigned)((z * PY + y) * PX + x);
    return (unsigned)((((z >> 2) * BYN + (y >> 2)) * BXN + (x >> 3)) * 128 + (z & 3) * 32 + (y & 3) * 8 + (x & 7));
}

// PATCH 0: lanes = 64 consecutive x; 1: lanes = 8 x times 8 y.  Tiles of 64 x 4 (or 8 x 32) x 16 z per workgroup of 4 waves.
template <int LAYOUT, int PATCH>
__global__ __launch_bounds__(256) void pattern(const float *__restrict__ tgt, const float *__restrict__ src, unsigned *out, Mat M,
                                               unsigned *regathers) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int x, y;
    const int gx = PATCH ? NX / 8 : NX / 64, gy = PATCH ? NY / 32 : NY / 4;
    const int bx = blockIdx.x % gx, by = (blockIdx.x / gx) % gy, bz = blockIdx.x / (gx * gy);
    if (PATCH) { x = bx * 8 + (lane & 7); y = by * 32 + wave * 8 + (lane >> 3); }
    else { x = bx * 64 + lane; y = by * 4 + wave; }
    const int z0 = bz * 16;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0, PX * PY * PZ * 4, 0x00020000);
    const float hx = M.m[0] * x + M.m[1] * y + M.m[3], hy = M.m[4] * x + M.m[5] * y + M.m[7], hz = M.m[8] * x + M.m[9] * y + M.m[11];
    auto coords = [&](int z, int &fx, int &fy, int &fz) {
        fx = (int)floorf(hx + M.m[2] * z); fy = (int)floorf(hy + M.m[6] * z); fz = (int)floorf(hz + M.m[10] * z);
        fx = min(max(fx, -1), NX - 1); fy = min(max(fy, -1), NY - 1); fz = min(max(fz, -1), NZ - 1);
    };
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    // the (x0, x1) taps of the rows y0, y1 of one plane: the loaded words are only combined one step later (as the cost kernel
    // interpolates one step behind its gathers), so nothing waits for a load it has just issued
    struct Plane { u2 a, b; };
    auto plane = [&](int fx, int fy, int fz) -> Plane {
        Plane p;
        if (LAYOUT == 0) {
            p.a = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)(4u * src_off<0>(fx, fy, fz)), 0, 0);
            p.b = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)(4u * src_off<0>(fx, fy + 1, fz)), 0, 0);
        } else {
            p.a = u2{__builtin_amdgcn_raw_buffer_load_b32(rs, (int)(4u * src_off<1>(fx, fy, fz)), 0, 0),
                     __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(4u * src_off<1>(fx + 1, fy, fz)), 0, 0)};
            p.b = u2{__builtin_amdgcn_raw_buffer_load_b32(rs, (int)(4u * src_off<1>(fx, fy + 1, fz)), 0, 0),
                     __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(4u * src_off<1>(fx + 1, fy + 1, fz)), 0, 0)};
        }
        return p;
    };
    auto mix = [](const Plane &p) { return p.a.x ^ p.a.y ^ p.b.x ^ p.b.y; };
    unsigned acc = 0, count = 0;
    int fx, fy, fz;
    coords(z0, fx, fy, fz);
    Plane lo = plane(fx, fy, fz), up = plane(fx, fy, fz + 1);
    unsigned tv = tgt[(size_t)x + NX * ((size_t)y + (size_t)NY * z0)];
#pragma unroll 3
    for (int z = z0 + 1; z < z0 + 16; ++z) {
        int nx_, ny_, nz_;
        coords(z, nx_, ny_, nz_);
        const Plane nu = plane(nx_, ny_, nz_ + 1);
        const unsigned nt = tgt[(size_t)x + NX * ((size_t)y + (size_t)NY * z)];
        acc ^= mix(lo) ^ mix(up) ^ tv;
        const bool moved_one = nx_ == fx && ny_ == fy && nz_ == fz + 1;
        if (__builtin_amdgcn_ballot_w64(!moved_one)) {  // wave-wide re-gather of the lower plane, like the library's kernels
            lo = plane(nx_, ny_, nz_);
            ++count;
        } else {
            lo = up;
        }
        up = nu; tv = nt; fx = nx_; fy = ny_; fz = nz_;
    }
    acc ^= mix(lo) ^ mix(up) ^ tv;
    if (acc == 0x12345678u) out[threadIdx.x] = acc;
    if (regathers && lane == 0 && count) atomicAdd(regathers, count);  // counted in an untimed launch only (262 144 atomics on one word take 0.8 ms)
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

static Mat rot_y(double deg, double scale, double sx, double sy, double sz) {
    const double c = std::cos(deg * M_PI / 180) * scale, s = std::sin(deg * M_PI / 180) * scale;
    const double cx = NX / 2.0, cy = NY / 2.0, cz = NZ / 2.0;
    Mat M;
    const double lin[9] = {c, 0, -s, 0, scale, 0, s, 0, c};
    const double ctr[3] = {cx, cy, cz}, sh[3] = {sx, sy, sz};
    for (int r = 0; r < 3; ++r) {
        double t = ctr[r] + sh[r];
        for (int k = 0; k < 3; ++k) {
            M.m[4 * r + k] = (float)lin[3 * r + k];
            t -= lin[3 * r + k] * ctr[k];
        }
        M.m[4 * r + 3] = (float)t;
    }
    return M;
}

int main() {
    const size_t nt = (size_t)NX * NY * NZ, ns = (size_t)PX * PY * PZ;
    float *tgt, *src;
    unsigned *out, *cnt;
    CHECK(hipMalloc(&tgt, nt * 4));
    CHECK(hipMalloc(&src, ns * 4));
    CHECK(hipMalloc(&out, 4096));
    CHECK(hipMalloc(&cnt, 4));
    CHECK(hipMemset(tgt, 1, nt * 4));
    CHECK(hipMemset(src, 2, ns * 4));
    const int grid = (NX / 64) * (NY / 4) * (NZ / 16);
    struct { const char *name; Mat M; } cases[] = {{"identity", rot_y(0, 1, 0, 0, 0)}, {"3 deg about y + 2 % + shift", rot_y(3, 1.02, 5, -3, 2)},
                                                   {"10 deg about y", rot_y(10, 1, 0, 0, 0)}, {"45 deg about y", rot_y(45, 1, 0, 0, 0)}};
    printf("%-30s %-28s %10s %12s\n", "matrix", "layout / lanes", "us", "re-gathers");
    for (auto &c : cases) {
#define RUN(L, P, LABEL)                                                                                                        \
        {                                                                                                                       \
            (void)hipMemset(cnt, 0, 4);                                                                                         \
            hipLaunchKernelGGL((pattern<L, P>), dim3(grid), dim3(256), 0, nullptr, tgt, src, out, c.M, cnt);                    \
            unsigned n = 0;                                                                                                     \
            (void)hipMemcpy(&n, cnt, 4, hipMemcpyDeviceToHost);                                                                 \
            const float us = time_us([&] { hipLaunchKernelGGL((pattern<L, P>), dim3(grid), dim3(256), 0, nullptr, tgt, src, out, c.M, (unsigned *)nullptr); }, 10); \
            printf("%-30s %-28s %10.1f %11.0f%%\n", c.name, LABEL, us, 100.0 * n / (15.0 * grid * 4));                           \
        }
        RUN(0, 0, "plain,  64 x 1 lanes") RUN(1, 0, "bricks, 64 x 1 lanes") RUN(0, 1, "plain,  8 x 8 lanes") RUN(1, 1, "bricks, 8 x 8 lanes")
    }
    CHECK(hipDeviceSynchronize());
    return 0;
}

// Access-pattern probe for a two-pass-per-convolution RL step ("two-pass probe"):
// how fast can 256-thread workgroups that keep a 128 KB tile in REGISTERS (64 float2 per thread) read and write the
// spectrum in the access patterns such a design needs?  Each kernel reads its tile, scales it, writes it back in place.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_twopass.hip -o build/probe_twopass ; build/probe_twopass
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                  \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            printf("%s failed: %s\n", #x, hipGetErrorString(e_));                 \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

constexpr int NZ = 256, NY = 512, NC = 256;  // complex columns per row
constexpr int PITCH = 272;                    // row pitch of the product layout (columns + 16)
constexpr int N1 = 64, N2 = 8, CW = 8;        // y = N2 * n1 + n2 ; CW columns per z-side segment

// ---- row layout [z][y][PITCH] -------------------------------------------------------------------------------------
// x-side tile: fixed (z, n2): rows y = N2 * k + n2, k = 0..63, thread = column.
__global__ __launch_bounds__(256, 2) void rows_xside(float2 *s) {
    const int z = blockIdx.x / N2, n2 = blockIdx.x % N2, c = threadIdx.x;
    float2 v[N1];
    float2 *base = s + ((size_t)z * NY + n2) * PITCH + c;
#pragma unroll
    for (int k = 0; k < N1; ++k) v[k] = base[(size_t)k * N2 * PITCH];
    float acc = 0.f;
#pragma unroll
    for (int q = 0; q < N1; ++q) acc += v[q].y;
    acc *= 1e-30f;
#pragma unroll
    for (int k = 0; k < N1; ++k) v[k].x = v[k].x * 1.0001f + acc;
#pragma unroll
    for (int k = 0; k < N1; ++k) base[(size_t)k * N2 * PITCH] = v[k];
}
// z-side tile: fixed (k1, column chunk): rows y = N2 * k1 + n2 (adjacent), CW columns, all z.
// lane = col + CW * n2 (8 segments of 64 B per wave access), wave w and register i <-> z = w + 4 * i.
__global__ __launch_bounds__(256, 2) void rows_zside(float2 *s) {
    const int k1 = blockIdx.x % N1, chunk = blockIdx.x / N1;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int col = chunk * CW + (lane & 7), n2 = lane >> 3;
    float2 v[64];
    float2 *base = s + ((size_t)w * NY + (size_t)k1 * N2 + n2) * PITCH + col;
#pragma unroll
    for (int i = 0; i < 64; ++i) v[i] = base[(size_t)i * 4 * NY * PITCH];
    float acc = 0.f;
#pragma unroll
    for (int q = 0; q < 64; ++q) acc += v[q].y;
    acc *= 1e-30f;
#pragma unroll
    for (int i = 0; i < 64; ++i) v[i].x = v[i].x * 1.0001f + acc;
#pragma unroll
    for (int i = 0; i < 64; ++i) base[(size_t)i * 4 * NY * PITCH] = v[i];
}

// ---- tiled layout [z][k1][chunk][n2][CW] ---------------------------------------------------------------------------
__device__ inline size_t tiled(int z, int k1, int chunk, int n2, int c) {
    return ((((size_t)z * N1 + k1) * (NC / CW) + chunk) * N2 + n2) * CW + c;
}
__global__ __launch_bounds__(256, 2) void tiled_xside(float2 *s) {
    const int z = blockIdx.x / N2, n2 = blockIdx.x % N2, c = threadIdx.x;
    float2 v[N1];
    float2 *base = s + tiled(z, 0, c / CW, n2, c % CW);
#pragma unroll
    for (int k = 0; k < N1; ++k) v[k] = base[(size_t)k * (NC / CW) * N2 * CW];
    float acc = 0.f;
#pragma unroll
    for (int q = 0; q < N1; ++q) acc += v[q].y;
    acc *= 1e-30f;
#pragma unroll
    for (int k = 0; k < N1; ++k) v[k].x = v[k].x * 1.0001f + acc;
#pragma unroll
    for (int k = 0; k < N1; ++k) base[(size_t)k * (NC / CW) * N2 * CW] = v[k];
}
__global__ __launch_bounds__(256, 2) void tiled_zside(float2 *s) {
    const int k1 = blockIdx.x % N1, chunk = blockIdx.x / N1;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float2 v[64];
    float2 *base = s + tiled(w, k1, chunk, 0, 0) + lane;
#pragma unroll
    for (int i = 0; i < 64; ++i) v[i] = base[(size_t)i * 4 * N1 * (NC / CW) * N2 * CW];
    float acc = 0.f;
#pragma unroll
    for (int q = 0; q < 64; ++q) acc += v[q].y;
    acc *= 1e-30f;
#pragma unroll
    for (int i = 0; i < 64; ++i) v[i].x = v[i].x * 1.0001f + acc;
#pragma unroll
    for (int i = 0; i < 64; ++i) base[(size_t)i * 4 * N1 * (NC / CW) * N2 * CW] = v[i];
}
// same with 16-byte accesses: lane <-> 2 columns, 32 lanes per 512 B block, two z planes per wave access
__global__ __launch_bounds__(256, 2) void tiled_zside16(float4 *s4) {
    const int k1 = blockIdx.x % N1, chunk = blockIdx.x / N1;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float4 v[32];
    const size_t plane = (size_t)N1 * (NC / CW) * N2 * CW / 2;  // float4 per z
    float4 *base = s4 + tiled(0, k1, chunk, 0, 0) / 2 + (size_t)(w * 2 + (lane >> 5)) * plane + (lane & 31);
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = base[(size_t)i * 8 * plane];
    float acc = 0.f;
#pragma unroll
    for (int q = 0; q < 32; ++q) acc += v[q].y;
    acc *= 1e-30f;
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i].x = v[i].x * 1.0001f + acc;
#pragma unroll
    for (int i = 0; i < 32; ++i) base[(size_t)i * 8 * plane] = v[i];
}
// x-side with 16-byte accesses in the row layout: thread <-> 2 columns, 128 threads of the 256 per row pair
__global__ __launch_bounds__(256, 2) void rows_xside16(float4 *s4) {
    const int z = blockIdx.x / N2, n2 = blockIdx.x % N2, t = threadIdx.x;
    const int c2 = t & 127, half = t >> 7;  // rows k = 2 * i + half
    float4 v[32];
    float4 *base = s4 + (((size_t)z * NY + n2 + (size_t)half * N2) * PITCH) / 2 + c2;
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = base[(size_t)i * 2 * N2 * PITCH / 2];
    float acc = 0.f;
#pragma unroll
    for (int q = 0; q < 32; ++q) acc += v[q].y;
    acc *= 1e-30f;
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i].x = v[i].x * 1.0001f + acc;
#pragma unroll
    for (int i = 0; i < 32; ++i) base[(size_t)i * 2 * N2 * PITCH / 2] = v[i];
}

// z-side, row layout, chunk pairs (one 128 B line) on the same XCD in consecutive dispatch slots
__global__ __launch_bounds__(256, 2) void rows_zside_paired(float2 *s) {
    const int b = blockIdx.x, q = b % 8, r = b / 8, half = r % 2, t = r / 2;
    const int k1 = q + 8 * (t % 8), chunk = 2 * (t / 8) + half;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int col = chunk * CW + (lane & 7), n2 = lane >> 3;
    float2 v[64];
    float2 *base = s + ((size_t)w * NY + (size_t)k1 * N2 + n2) * PITCH + col;
#pragma unroll
    for (int i = 0; i < 64; ++i) v[i] = base[(size_t)i * 4 * NY * PITCH];
    float acc = 0.f;
#pragma unroll
    for (int q2 = 0; q2 < 64; ++q2) acc += v[q2].y;
    acc *= 1e-30f;
#pragma unroll
    for (int i = 0; i < 64; ++i) v[i].x = v[i].x * 1.0001f + acc;
#pragma unroll
    for (int i = 0; i < 64; ++i) base[(size_t)i * 4 * NY * PITCH] = v[i];
}
// z-side, row layout, 16 columns (128 B segments): 512 threads, one workgroup per CU; lane = col + 16 * (n2 & 3),
// wave w: n2 half = w & 1, z = (w >> 1) + 4 * i
__global__ __launch_bounds__(512, 1) void rows_zside_w16(float2 *s) {
    const int k1 = blockIdx.x % N1, chunk = blockIdx.x / N1;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int col = chunk * 16 + (lane & 15), n2 = (lane >> 4) + 4 * (w & 1);
    float2 v[64];
    float2 *base = s + ((size_t)(w >> 1) * NY + (size_t)k1 * N2 + n2) * PITCH + col;
#pragma unroll
    for (int i = 0; i < 64; ++i) v[i] = base[(size_t)i * 4 * NY * PITCH];
    float acc = 0.f;
#pragma unroll
    for (int q2 = 0; q2 < 64; ++q2) acc += v[q2].y;
    acc *= 1e-30f;
#pragma unroll
    for (int i = 0; i < 64; ++i) v[i].x = v[i].x * 1.0001f + acc;
#pragma unroll
    for (int i = 0; i < 64; ++i) base[(size_t)i * 4 * NY * PITCH] = v[i];
}
// same tile with 16-byte accesses: lane = 2 columns (8 lanes per 128 B segment), 8 rows per wave access
__global__ __launch_bounds__(512, 1) void rows_zside_w16v4(float4 *s4) {
    const int k1 = blockIdx.x % N1, chunk = blockIdx.x / N1;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;  // 8 waves: z = w + 8 * i
    const int c2 = chunk * 8 + (lane & 7), n2 = lane >> 3;
    float4 v[32];
    float4 *base = s4 + (((size_t)w * NY + (size_t)k1 * N2 + n2) * PITCH) / 2 + c2;
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = base[(size_t)i * 8 * NY * PITCH / 2];
    float acc = 0.f;
#pragma unroll
    for (int q2 = 0; q2 < 32; ++q2) acc += v[q2].y;
    acc *= 1e-30f;
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i].x = v[i].x * 1.0001f + acc;
#pragma unroll
    for (int i = 0; i < 32; ++i) base[(size_t)i * 8 * NY * PITCH / 2] = v[i];
}
// 8 columns with 16-byte accesses: lane = 2 columns (4 lanes per 64 B segment), 8 rows x 2 planes per wave access
__global__ __launch_bounds__(256, 2) void rows_zside_v4(float4 *s4) {
    const int k1 = blockIdx.x % N1, chunk = blockIdx.x / N1;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;  // 4 waves x 2 planes: z = 2 * w + (lane >> 5) + 8 * i
    const int c2 = chunk * 4 + (lane & 3), n2 = (lane >> 2) & 7, zz = 2 * w + (lane >> 5);
    float4 v[32];
    float4 *base = s4 + (((size_t)zz * NY + (size_t)k1 * N2 + n2) * PITCH) / 2 + c2;
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = base[(size_t)i * 8 * NY * PITCH / 2];
    float acc = 0.f;
#pragma unroll
    for (int q2 = 0; q2 < 32; ++q2) acc += v[q2].y;
    acc *= 1e-30f;
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i].x = v[i].x * 1.0001f + acc;
#pragma unroll
    for (int i = 0; i < 32; ++i) base[(size_t)i * 8 * NY * PITCH / 2] = v[i];
}
// plain in-place float4 scale of the same bytes
__global__ __launch_bounds__(256) void flat(float4 *s4, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float4 v = s4[i];
        v.x *= 1.0001f;
        s4[i] = v;
    }
}

int main() {
    const size_t n = (size_t)NZ * NY * PITCH;
    float2 *s;
    CHECK(hipMalloc(&s, n * sizeof(float2)));
    CHECK(hipMemset(s, 0, n * sizeof(float2)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const double bytes = 2.0 * NZ * NY * NC * sizeof(float2);
    auto run = [&](const char *name, auto launch) {
        for (int i = 0; i < 5; ++i) launch();
        CHECK(hipDeviceSynchronize());
        float best = 1e9f, sum = 0;
        for (int rep = 0; rep < 5; ++rep) {
            CHECK(hipEventRecord(e0));
            for (int i = 0; i < 20; ++i) launch();
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            ms /= 20;
            best = ms < best ? ms : best;
            sum += ms;
        }
        printf("%-16s best %.4f ms  mean %.4f ms  -> %.0f GB/s\n", name, best, sum / 5, bytes / best * 1e-6);
    };
    const int wg = NZ * N2;  // 2048 either way
    run("flat float4", [&] { flat<<<8192, 256>>>((float4 *)s, (size_t)NZ * NY * NC / 2); });
    run("rows_xside", [&] { rows_xside<<<wg, 256>>>(s); });
    run("rows_xside16", [&] { rows_xside16<<<wg, 256>>>((float4 *)s); });
    run("rows_zside", [&] { rows_zside<<<N1 * (NC / CW), 256>>>(s); });
    run("rows_zside_pair", [&] { rows_zside_paired<<<N1 * (NC / CW), 256>>>(s); });
    run("rows_zside_v4", [&] { rows_zside_v4<<<N1 * (NC / CW), 256>>>((float4 *)s); });
    run("rows_zside_w16", [&] { rows_zside_w16<<<N1 * (NC / 16), 512>>>(s); });
    run("rows_zside_w16v4", [&] { rows_zside_w16v4<<<N1 * (NC / 16), 512>>>((float4 *)s); });
    run("tiled_xside", [&] { tiled_xside<<<wg, 256>>>(s); });
    run("tiled_zside", [&] { tiled_zside<<<N1 * (NC / CW), 256>>>(s); });
    run("tiled_zside16", [&] { tiled_zside16<<<N1 * (NC / CW), 256>>>((float4 *)s); });
    CHECK(hipGetLastError());
    CHECK(hipDeviceSynchronize());
    printf("done\n");
    return 0;
}

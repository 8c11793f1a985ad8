// Micro-benchmark of the vector-memory address path (TA) of one gfx950 CU: what does a wave-wide buffer load cost when it
// carries 4 / 8 / 16 bytes per lane, and what when only a few lanes are active?  (the registration cost
// kernel is bound by this path; its wave-wide "exception" gathers have 3-5 active lanes.)  All addresses hit a 16 KB window,
// so the numbers are issue / tag-lookup rates, not HBM.
//   hipcc --offload-arch=gfx950 -O3 -o build/probe_ta tools/probe_ta.hip && build/probe_ta
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int BYTES>
__device__ __forceinline__ unsigned load(__amdgpu_buffer_rsrc_t r, unsigned off) {
    if constexpr (BYTES == 4) return __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0);
    else if constexpr (BYTES == 8) {
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, 0);
        return v.x ^ v.y;
    } else {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
        return v.x ^ v.y ^ v.z ^ v.w;
    }
}

// MODE 0: all lanes | 1: lanes 0..3 (one quad) | 2: lanes 0, 16, 32, 48 | 3: lanes 0..15 | 4: every 4th lane (16 lanes)
template <int BYTES, int MODE>
__global__ __launch_bounds__(256) void probe(const unsigned *buf, unsigned *out, int iters) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned *>(buf), 0, 1 << 20, 0x00020000);
    const int lane = threadIdx.x & 63;
    bool on = true;
    if (MODE == 1) on = lane < 4;
    if (MODE == 2) on = (lane & 15) == 0;
    if (MODE == 3) on = lane < 16;
    if (MODE == 4) on = (lane & 3) == 0;
    unsigned acc = 0, off = (unsigned)lane * BYTES;
    if (on) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 8; ++k) acc ^= load<BYTES>(r, (off + (unsigned)k * 1024u) & 16383u);
            off += 64u * BYTES;
        }
    }
    if (acc == 0x12345678u) out[threadIdx.x] = acc;
}

// address patterns of a near-identity gather: PAT 0 consecutive | 1 a gap of one element every 16 lanes | 2 lanes 32..63 on
// another row (+4 KB) | 3 unaligned (+4 bytes) | 4 pseudo-random element per lane inside the window
template <int BYTES, int PAT>
__global__ __launch_bounds__(256) void probe_pat(const unsigned *buf, unsigned *out, int iters) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned *>(buf), 0, 1 << 20, 0x00020000);
    const int lane = threadIdx.x & 63;
    unsigned e = (unsigned)lane;
    if (PAT == 1) e = (unsigned)(lane + (lane >> 4));
    if (PAT == 2) e = (unsigned)((lane & 31) + ((lane >> 5) ? 4096 / BYTES : 0));
    if (PAT == 4) e = (unsigned)((lane * 2654435761u) >> 22) & 1023u;
    unsigned acc = 0, off = e * BYTES + (PAT == 3 ? 4u : 0u);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) acc ^= load<BYTES>(r, (off + (unsigned)k * 1024u) & 16383u);
        off += 64u * BYTES;
    }
    if (acc == 0x12345678u) out[threadIdx.x] = acc;
}

template <int BYTES, int PAT>
void run_pat(const char *name, const unsigned *buf, unsigned *out) {
    const int iters = 2000, blocks = 256 * 8;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL((probe_pat<BYTES, PAT>), dim3(blocks), dim3(256), 0, 0, buf, out, 50);
    hipEventRecord(a);
    hipLaunchKernelGGL((probe_pat<BYTES, PAT>), dim3(blocks), dim3(256), 0, 0, buf, out, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double wave_instr_per_cu = (double)blocks * 4 * iters * 8 / 256.0;
    printf("%-44s %8.3f ms  %7.2f ns per wave-instruction per CU  (= %5.1f cycles at 2.1 GHz)\n", name, ms,
           ms * 1e6 / wave_instr_per_cu, ms * 1e6 / wave_instr_per_cu * 2.1);
}

// plain streaming read of `bytes` (16-byte loads, grid-stride): the HBM read rate a cost evaluation could reach at best
__global__ __launch_bounds__(256) void stream_read(const uint4 *p, size_t n16, unsigned *out) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        const uint4 v = p[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[threadIdx.x] = acc;
}

void run_stream(const char *name, size_t bytes, unsigned *out) {
    uint4 *p;
    hipMalloc(&p, bytes);
    hipMemset(p, 1, bytes);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int blocks : {2048, 8192}) {
        hipLaunchKernelGGL(stream_read, dim3(blocks), dim3(256), 0, 0, p, bytes / 16, out);
        hipEventRecord(a);
        for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(stream_read, dim3(blocks), dim3(256), 0, 0, p, bytes / 16, out);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        printf("%-30s %5d blocks  %8.1f us per pass  %7.1f GB/s\n", name, blocks, ms * 1e3 / 20, bytes / (ms * 1e-3 / 20) / 1e9);
    }
    hipFree(p);
}

template <int BYTES, int MODE>
void run(const char *name, const unsigned *buf, unsigned *out) {
    const int iters = 2000, blocks = 256 * 8;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL((probe<BYTES, MODE>), dim3(blocks), dim3(256), 0, 0, buf, out, 50);
    hipEventRecord(a);
    hipLaunchKernelGGL((probe<BYTES, MODE>), dim3(blocks), dim3(256), 0, 0, buf, out, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double wave_instr_per_cu = (double)blocks * 4 * iters * 8 / 256.0;
    printf("%-44s %8.3f ms  %7.2f ns per wave-instruction per CU  (= %5.1f cycles at 2.1 GHz)\n", name, ms,
           ms * 1e6 / wave_instr_per_cu, ms * 1e6 / wave_instr_per_cu * 2.1);
}

int main() {
    unsigned *buf, *out;
    hipMalloc(&buf, 1 << 20);
    hipMalloc(&out, 4096);
    hipMemset(buf, 1, 1 << 20);
    run<4, 0>("4 B / lane, 64 lanes", buf, out);
    run<8, 0>("8 B / lane, 64 lanes", buf, out);
    run<16, 0>("16 B / lane, 64 lanes", buf, out);
    run<16, 1>("16 B / lane, lanes 0-3", buf, out);
    run<16, 2>("16 B / lane, lanes 0,16,32,48", buf, out);
    run<16, 3>("16 B / lane, lanes 0-15", buf, out);
    run<16, 4>("16 B / lane, every 4th lane", buf, out);
    run<8, 1>("8 B / lane, lanes 0-3", buf, out);
    run<8, 2>("8 B / lane, lanes 0,16,32,48", buf, out);
    run<8, 4>("8 B / lane, every 4th lane", buf, out);
    run<4, 1>("4 B / lane, lanes 0-3", buf, out);
    run<4, 2>("4 B / lane, lanes 0,16,32,48", buf, out);
    run_pat<4, 0>("4 B consecutive", buf, out);
    run_pat<4, 1>("4 B, a gap every 16 lanes", buf, out);
    run_pat<4, 2>("4 B, lanes 32-63 on another row", buf, out);
    run_pat<4, 4>("4 B, scattered lanes", buf, out);
    run_pat<8, 0>("8 B consecutive", buf, out);
    run_pat<8, 3>("8 B consecutive, 4-byte aligned only", buf, out);
    run_pat<8, 1>("8 B, a gap every 16 lanes", buf, out);
    run_pat<8, 4>("8 B, scattered lanes", buf, out);
    run_pat<16, 0>("16 B consecutive", buf, out);
    run_pat<16, 3>("16 B consecutive, 4-byte aligned only", buf, out);
    run_pat<16, 1>("16 B, a gap every 16 lanes", buf, out);
    run_pat<16, 4>("16 B, scattered lanes", buf, out);
    run_stream("read 2R = 537 MB", 536870912, out);
    run_stream("read 3R = 805 MB", 805306368, out);
    return 0;
}

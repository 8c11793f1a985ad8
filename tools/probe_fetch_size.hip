// Is the "FETCH_SIZE x 2" correction (MI355X_MICROARCH.md: the counter under-reports 16-byte streaming reads by 2 on gfx950) also
// right for the 8-byte GATHERS of the registration cost kernel?  Three kernels that read a known number of bytes exactly once:
//   stream16   16 bytes per lane, contiguous            (the case the guide's rule was stated for)
//   gather8    8 bytes per lane through a buffer descriptor (buffer_load_dwordx2), contiguous across the wave
//   gather8s   the same, every lane 128 bytes apart      (one 8-byte word per cache line: what does a line cost?)
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace -- build/probe_fetch_size     (tools/gpu_profile_run_r04.sh, profiles/r04/)
#include <hip/hip_runtime.h>

#include <cstdio>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void stream16(const u32x4 *__restrict__ a, size_t n, unsigned *out) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const u32x4 v = a[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[threadIdx.x] = acc;
}

template <int STRIDE_BYTES, int AUX = 0>
__global__ __launch_bounds__(256) void gather8(const float *__restrict__ a, unsigned bytes, unsigned *out) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a), 0, (int)bytes, 0x00020000);
    unsigned acc = 0;
    const unsigned step = (unsigned)gridDim.x * 256u * STRIDE_BYTES;
    for (unsigned off = ((unsigned)blockIdx.x * 256u + threadIdx.x) * STRIDE_BYTES; off < bytes; off += step) {
        const auto v = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)off, 0, AUX);
        acc ^= v[0] ^ v[1];
    }
    if (acc == 0x12345678u) out[threadIdx.x] = acc;
}

// round 5: the same reads with the non-temporal hint (how are THEY tallied?  csrc/fft2p.hip and corr_padded.hip use it)
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void stream16_nt(const f32x4 *__restrict__ a, size_t n, unsigned *out) {
    float acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const f32x4 v = __builtin_nontemporal_load(a + i);
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 0.12345678f) out[threadIdx.x] = 1;
}

int main() {
    const size_t bytes = (size_t)1 << 30;  // 1 GiB, read once by every kernel (gather8s: one word of each 128-byte line)
    float *a;
    unsigned *out;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&out, 4096) != hipSuccess) return 1;
    (void)hipMemset(a, 1, bytes);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(stream16, dim3(4096), dim3(256), 0, nullptr, reinterpret_cast<const u32x4 *>(a), bytes / 16, out);
        hipLaunchKernelGGL(gather8<8>, dim3(4096), dim3(256), 0, nullptr, a, (unsigned)bytes, out);
        hipLaunchKernelGGL(gather8<128>, dim3(4096), dim3(256), 0, nullptr, a, (unsigned)bytes, out);
        hipLaunchKernelGGL(stream16_nt, dim3(4096), dim3(256), 0, nullptr, reinterpret_cast<const f32x4 *>(a), bytes / 16, out);
        hipLaunchKernelGGL((gather8<8, 2>), dim3(4096), dim3(256), 0, nullptr, a, (unsigned)bytes, out);
    }
    (void)hipDeviceSynchronize();
    printf("each kernel read %zu bytes (gather8<128>: %zu bytes requested in %zu lines)\n", bytes, bytes / 16, bytes / 128);
    return 0;
}

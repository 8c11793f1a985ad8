// Probe: how long is the round trip "kernel publishes a result -> host reacts -> next dependent kernel starts", for
//   A  the library's scheme: poll a pinned result slot, then hipLaunchKernelGGL the next kernel;
//   B  the next kernel PRE-QUEUED behind a stream memory wait (hipStreamWaitValue64 on a pinned mailbox) that the host
//      releases by writing the mailbox after it has seen the result -- the command processor waits, no wave spins;
//   C  the next kernel pre-queued behind a one-wave GATE kernel that spins on the mailbox (bounded) and then exits.
// The kernel is a stand-in of adjustable length that reads its "matrix" from kernel arguments (A) or from the mailbox (B, C)
// and writes {value, ticket} to pinned memory at its end.  Prints the mean microseconds per dependent evaluation for each
// scheme at two kernel lengths: the difference between schemes is the gap they save.
//   hipcc --offload-arch=gfx950 -O2 -o gpurun_out/probe_gate tools/probe_gate.hip && gpurun_out/probe_gate
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>

#define CHECK(e)                                                                   \
    do {                                                                           \
        hipError_t err_ = (e);                                                     \
        if (err_ != hipSuccess) {                                                  \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(err_)); \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

struct Mail {
    volatile unsigned long long ticket;  // released evaluations
    volatile float m[12];
};

__global__ void work_kernel(const float *data, int spins, float m0, const Mail *mail, volatile double *host, double ticket) {
    float acc = mail ? mail->m[0] : m0;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const unsigned long long t0 = wall_clock64();
        while (wall_clock64() - t0 < (unsigned long long)spins) __builtin_amdgcn_s_sleep(8);
        host[0] = (double)acc + data[0];
        __threadfence_system();
        host[1] = ticket;
    }
}

// D: the gate COPIES the matrix into device memory once it is released; the work kernel reads it from there (B and C let every
//    wave of the work kernel read mail->m[0] from host memory: 6 144 uncached reads over PCIe -- that, not the waiting, is
//    their 0.6 ms)
__global__ void gate_copy_kernel(const Mail *mail, unsigned long long want, float *d_m, int *timed_out) {
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(&mail->ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < want) {
        if (wall_clock64() - t0 > 100000000ull) {
            *timed_out = 1;
            return;
        }
        __builtin_amdgcn_s_sleep(2);
    }
    if (threadIdx.x < 12) d_m[threadIdx.x] = mail->m[threadIdx.x];
}
__global__ void work_dev_kernel(const float *data, int spins, const float *d_m, volatile double *host, double ticket) {
    float acc = d_m[0];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const unsigned long long t0 = wall_clock64();
        while (wall_clock64() - t0 < (unsigned long long)spins) __builtin_amdgcn_s_sleep(8);
        host[0] = (double)acc + data[0];
        __threadfence_system();
        host[1] = ticket;
    }
}
// E: ONE kernel per evaluation, pre-queued: workgroup 0 waits for the release, copies the matrix into device memory and raises a
//    device flag (agent scope); every other workgroup spins on that flag (L2, not PCIe)
__global__ void work_gated_kernel(const float *data, int spins, const Mail *mail, unsigned long long want, float *d_m,
                                  unsigned long long *d_flag, volatile double *host, double ticket, int *timed_out) {
    __shared__ float m0;
    if (threadIdx.x == 0) {
        const unsigned long long t0 = wall_clock64();
        if (blockIdx.x == 0) {
            while (__hip_atomic_load(&mail->ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < want) {
                if (wall_clock64() - t0 > 100000000ull) { *timed_out = 1; break; }
                __builtin_amdgcn_s_sleep(2);
            }
            for (int i = 0; i < 12; ++i) __hip_atomic_store(d_m + i, mail->m[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_s_waitcnt(0);
            __hip_atomic_store(d_flag, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (__hip_atomic_load(d_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                if (wall_clock64() - t0 > 100000000ull) break;
                __builtin_amdgcn_s_sleep(1);
            }
        }
        m0 = __hip_atomic_load(d_m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    float acc = m0;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const unsigned long long t0 = wall_clock64();
        while (wall_clock64() - t0 < (unsigned long long)spins) __builtin_amdgcn_s_sleep(8);
        host[0] = (double)acc + data[0];
        __threadfence_system();
        host[1] = ticket;
    }
}

__global__ void gate_kernel(const Mail *mail, unsigned long long want, int *timed_out) {
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(&mail->ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < want) {
        if (wall_clock64() - t0 > 100000000ull) {  // 1 s at 100 MHz
            *timed_out = 1;
            return;
        }
        __builtin_amdgcn_s_sleep(2);
    }
}

int main() {
    hipStream_t s;
    CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    float *d_data;
    CHECK(hipMalloc(&d_data, 4096));
    CHECK(hipMemset(d_data, 0, 4096));
    double *host;
    CHECK(hipHostMalloc((void **)&host, 2 * sizeof(double), hipHostMallocCoherent));
    Mail *mail;
    CHECK(hipHostMalloc((void **)&mail, sizeof(Mail), hipHostMallocCoherent));
    int *d_to;
    CHECK(hipMalloc(&d_to, 4));
    CHECK(hipMemset(d_to, 0, 4));
    const int N = 2000;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto wait_ticket = [&](double t) {
        while (host[1] != t) __builtin_ia32_pause();
    };
    for (int spins : {0, 10000}) {  // wall_clock64 ticks of 10 ns: 0 and 100 us
        // ---- A
        host[1] = 0;
        double ticket = 0;
        auto t0 = now();
        for (int i = 0; i < N; ++i) {
            ticket += 1;
            hipLaunchKernelGGL(work_kernel, dim3(1536), dim3(256), 0, s, d_data, spins, (float)i, (const Mail *)nullptr, host, ticket);
            wait_ticket(ticket);
        }
        const double a = std::chrono::duration<double>(now() - t0).count() / N * 1e6;
        // ---- B: stream wait value
        double b = -1;
        {
            mail->ticket = 0;
            host[1] = 0;
            ticket = 0;
            unsigned long long rel = 0;
            hipError_t e = hipStreamWaitValue64(s, (void *)&mail->ticket, 1, hipStreamWaitValueGte, 0xffffffffffffffffull);
            if (e == hipSuccess) {
                hipLaunchKernelGGL(work_kernel, dim3(1536), dim3(256), 0, s, d_data, spins, 0.f, (const Mail *)mail, host, 1.0);
                t0 = now();
                for (int i = 0; i < N; ++i) {
                    ticket += 1;
                    mail->m[0] = (float)i;  // "the matrix", then the release
                    __atomic_thread_fence(__ATOMIC_RELEASE);
                    mail->ticket = ++rel;
                    if (i + 1 < N) {  // pre-queue the next evaluation while this one runs
                        CHECK(hipStreamWaitValue64(s, (void *)&mail->ticket, rel + 1, hipStreamWaitValueGte, 0xffffffffffffffffull));
                        hipLaunchKernelGGL(work_kernel, dim3(1536), dim3(256), 0, s, d_data, spins, 0.f, (const Mail *)mail, host, ticket + 1);
                    }
                    wait_ticket(ticket);
                }
                b = std::chrono::duration<double>(now() - t0).count() / N * 1e6;
            } else {
                fprintf(stderr, "hipStreamWaitValue64: %s\n", hipGetErrorString(e));
                (void)hipGetLastError();
            }
            CHECK(hipStreamSynchronize(s));
        }
        // ---- C: gate kernel
        mail->ticket = 0;
        host[1] = 0;
        ticket = 0;
        unsigned long long rel = 0;
        hipLaunchKernelGGL(gate_kernel, dim3(1), dim3(64), 0, s, (const Mail *)mail, 1ull, d_to);
        hipLaunchKernelGGL(work_kernel, dim3(1536), dim3(256), 0, s, d_data, spins, 0.f, (const Mail *)mail, host, 1.0);
        t0 = now();
        for (int i = 0; i < N; ++i) {
            ticket += 1;
            mail->m[0] = (float)i;
            __atomic_thread_fence(__ATOMIC_RELEASE);
            mail->ticket = ++rel;
            if (i + 1 < N) {
                hipLaunchKernelGGL(gate_kernel, dim3(1), dim3(64), 0, s, (const Mail *)mail, rel + 1, d_to);
                hipLaunchKernelGGL(work_kernel, dim3(1536), dim3(256), 0, s, d_data, spins, 0.f, (const Mail *)mail, host, ticket + 1);
            }
            wait_ticket(ticket);
        }
        const double c = std::chrono::duration<double>(now() - t0).count() / N * 1e6;
        CHECK(hipStreamSynchronize(s));
        // ---- D: gate kernel that copies the matrix to device memory
        float *d_m;
        unsigned long long *d_flag;
        CHECK(hipMalloc(&d_m, 64));
        CHECK(hipMalloc(&d_flag, 8));
        CHECK(hipMemset(d_flag, 0, 8));
        mail->ticket = 0;
        host[1] = 0;
        ticket = 0;
        rel = 0;
        hipLaunchKernelGGL(gate_copy_kernel, dim3(1), dim3(64), 0, s, (const Mail *)mail, 1ull, d_m, d_to);
        hipLaunchKernelGGL(work_dev_kernel, dim3(1536), dim3(256), 0, s, d_data, spins, (const float *)d_m, host, 1.0);
        t0 = now();
        for (int i = 0; i < N; ++i) {
            ticket += 1;
            mail->m[0] = (float)i;
            __atomic_thread_fence(__ATOMIC_RELEASE);
            mail->ticket = ++rel;
            if (i + 1 < N) {
                hipLaunchKernelGGL(gate_copy_kernel, dim3(1), dim3(64), 0, s, (const Mail *)mail, rel + 1, d_m, d_to);
                hipLaunchKernelGGL(work_dev_kernel, dim3(1536), dim3(256), 0, s, d_data, spins, (const float *)d_m, host, ticket + 1);
            }
            wait_ticket(ticket);
        }
        const double d = std::chrono::duration<double>(now() - t0).count() / N * 1e6;
        CHECK(hipStreamSynchronize(s));
        // ---- E: the gate inside the work kernel
        mail->ticket = 0;
        host[1] = 0;
        ticket = 0;
        rel = 0;
        hipLaunchKernelGGL(work_gated_kernel, dim3(1536), dim3(256), 0, s, d_data, spins, (const Mail *)mail, 1ull, d_m, d_flag, host, 1.0, d_to);
        t0 = now();
        for (int i = 0; i < N; ++i) {
            ticket += 1;
            mail->m[0] = (float)i;
            __atomic_thread_fence(__ATOMIC_RELEASE);
            mail->ticket = ++rel;
            if (i + 1 < N)
                hipLaunchKernelGGL(work_gated_kernel, dim3(1536), dim3(256), 0, s, d_data, spins, (const Mail *)mail, rel + 1, d_m, d_flag, host, ticket + 1, d_to);
            wait_ticket(ticket);
        }
        const double e5 = std::chrono::duration<double>(now() - t0).count() / N * 1e6;
        CHECK(hipStreamSynchronize(s));
        printf("kernel body %4d us: D gate kernel copies the matrix to device memory %.2f us | E gate inside the work kernel %.2f us per dependent evaluation\n", spins / 100, d, e5);
        CHECK(hipFree(d_m));
        CHECK(hipFree(d_flag));
        int to = 0;
        CHECK(hipMemcpy(&to, d_to, 4, hipMemcpyDeviceToHost));
        printf("kernel body %4d us: A launch after poll %.2f us | B pre-queued behind hipStreamWaitValue64 %.2f us | C pre-queued behind a gate "
               "kernel %.2f us per dependent evaluation (gate timed out: %d)\n", spins / 100, a, b, c, to);
    }
    return 0;
}

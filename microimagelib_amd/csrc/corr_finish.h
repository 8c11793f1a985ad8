// Tail of every registration cost kernel: the workgroup that finishes LAST sums the per-workgroup partials -- in an
// order that does not depend on which workgroup that is -- and publishes {sum s*s, sum s*t, ticket} to pinned host
// memory.  One launch per evaluation instead of two (the reference uses three plus two host loops,
// src/api_subfunc.cu:954-988), and the host learns the result by polling the ticket instead of a stream
// synchronisation: ~15 us less per evaluation, of ~165.
#pragma once

#include "kernels.h"

namespace mil {
namespace k {

__device__ __forceinline__ double corr_wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// Visibility between workgroups (which may sit on different XCDs, each with its own L2) WITHOUT release / acquire
// fences: an agent-scope fence is a write-back + invalidate of the whole L2 (buffer_wbl2 sc1 / buffer_inv sc1), measured
// at ~6 us apiece -- 0.75 ms per evaluation with one per workgroup.  Instead every datum that crosses workgroups is
// itself stored and loaded at agent scope (sc1: written through to / read from the memory side), and a workgroup waits
// for its stores to be acknowledged (s_waitcnt vmcnt(0)) before it counts itself done.
//
// Why this is sound on gfx950, and what it rests on.  The LLVM AMDGPU back end's memory-model table for gfx942 (which
// gfx950 shares; AMDGPUUsage, "Memory Model", code sequences for GFX942) implements
//   * `store atomic monotonic, agent scope, global`   as  `global_store sc1=1`      -- the store bypasses the non-coherent
//     per-XCD L2 and completes at the memory side;
//   * `load atomic monotonic, agent scope, global`    as  `global_load sc1=1`       -- served from the memory side, never
//     from a stale L2 line of this XCD;
//   * `atomicrmw monotonic, agent scope`              as  `global_atomic sc1=1`     -- executed at the memory side;
//   * the release half of `fence release, agent`      as  `buffer_wbl2 sc1=1; s_waitcnt vmcnt(0)`: write back dirty L2
//     lines, then wait until every earlier store has been acknowledged.
// Our data are never dirty in L2 (every cross-workgroup store is itself sc1), so the write-back has nothing to do and what
// remains of the release is exactly the `s_waitcnt vmcnt(0)` issued here; the acquire half (`buffer_inv sc1`) would only
// discard lines that the sc1 loads do not consult.  This is a hardware argument, outside the HIP memory model, so the
// fenced form is kept selectable: MIL_CORR_FENCES=1 adds `fence release` before a workgroup counts itself done and
// `fence acquire` before the last one sums (CorrTail::fenced); tests/test_gpu_reg.py demands the same bits from both.
__device__ __forceinline__ void corr_store_agent(double *p, double v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double corr_load_agent(const double *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void corr_stores_done() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ unsigned corr_count(unsigned *c) {
    return __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Sum of n (ss, st) pairs p[i * stride], q[i * stride] in a fixed order: thread-strided, then waves, then thread 0.
// A thread issues 8 loads per array before the first add, so the tail costs one memory round trip per 8 * THREADS
// elements, not one per element.  Result valid in thread 0.
template <int THREADS>
__device__ __forceinline__ void corr_block_sum(const double *p, const double *q, int n, int stride, double *lds,
                                               double &r0, double &r1) {
    constexpr int WAVES = THREADS / 64;
    double s0 = 0.0, s1 = 0.0;
    for (int base = 0; base < n; base += 8 * THREADS) {  // one trip up to 2048 partials per group (512 x 512 x 256: 512)
        double v0[8], v1[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = base + (int)threadIdx.x + k * THREADS;
            const bool ok = i < n;
            v0[k] = ok ? corr_load_agent(p + (size_t)i * stride) : 0.0;
            v1[k] = ok ? corr_load_agent(q + (size_t)i * stride) : 0.0;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            s0 += v0[k];
            s1 += v1[k];
        }
    }
    s0 = corr_wave_sum(s0);
    s1 = corr_wave_sum(s1);
    __syncthreads();  // lds may still be in use by the caller's previous reduction
    if ((threadIdx.x & 63) == 0) {
        lds[threadIdx.x >> 6] = s0;
        lds[WAVES + (threadIdx.x >> 6)] = s1;
    }
    __syncthreads();
    r0 = r1 = 0.0;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            r0 += lds[w];
            r1 += lds[WAVES + w];
        }
    }
}

// Call from all THREADS threads of the workgroup; (a0, a1) = this workgroup's sums, valid in thread 0.
// lds: >= 2 * (THREADS / 64) doubles; flag: one int of shared memory.
//
// Two levels, both for the count and for the sum: workgroup b belongs to group b % G (G = kCorrCounters).  The last
// workgroup of a group to finish sums that group's partials into the group slot; the last GROUP to finish sums the G
// slots and publishes.  (~3 * 10^4 workgroups on ONE counter serialise in the memory-side atomic unit, and one
// workgroup summing 6 * 10^4 doubles alone is a long latency-bound tail.)  Which workgroup is "last" varies from run to
// run; what it adds, and in which order, does not.
//
// corr_publish_as: the same with an explicit partial index b -- a workgroup that covers the voxels of several
// "virtual" workgroups of the 64 x 4 x 16 decomposition (corr_pair_kernel: two of them) publishes once per virtual
// workgroup, so that the partials, their grouping and the order of every addition are those of the one-voxel-per-lane
// kernels: all cost kernels return the same bits.
template <int THREADS>
__device__ __forceinline__ void corr_publish_as(int b, double a0, double a1, CorrTail t, double *lds, int *flag);

template <int THREADS>
__device__ __forceinline__ void corr_publish(double a0, double a1, CorrTail t, double *lds, int *flag) {
    corr_publish_as<THREADS>(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), a0, a1, t, lds, flag);
}

template <int THREADS>
__device__ __forceinline__ void corr_publish_as(int b, double a0, double a1, CorrTail t, double *lds, int *flag) {
    __syncthreads();  // *flag and lds may still be read by a previous publish of this workgroup
    const int groups = t.nblocks < kCorrCounters ? t.nblocks : kCorrCounters;
    const int grp = b % groups;
    const int members = (t.nblocks - grp + groups - 1) / groups;
    double *gsum = t.partials + 2 * (size_t)t.nblocks;             // [2][kCorrCounters]
    if (threadIdx.x == 0) {
        corr_store_agent(t.partials + b, a0);
        corr_store_agent(t.partials + t.nblocks + b, a1);
        corr_stores_done();
        if (t.fenced) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        *flag = corr_count(t.counter + grp) == (unsigned)members - 1u;
        if (t.fenced && *flag) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (!*flag) return;
    double r0, r1;
    corr_block_sum<THREADS>(t.partials + grp, t.partials + t.nblocks + grp, members, groups, lds, r0, r1);
    if (threadIdx.x == 0) {
        corr_store_agent(gsum + grp, r0);
        corr_store_agent(gsum + kCorrCounters + grp, r1);
        corr_stores_done();
        if (t.fenced) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        *flag = corr_count(t.counter + kCorrCounters) == (unsigned)groups - 1u;
        if (t.fenced && *flag) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (!*flag) return;
    corr_block_sum<THREADS>(gsum, gsum + kCorrCounters, groups, 1, lds, r0, r1);
    if (threadIdx.x == 0) {
        // ready for the next evaluation: kernels of one stream run in order, and the end of a kernel writes L2 back
        for (int i = 0; i <= kCorrCounters; ++i)
            __hip_atomic_store(t.counter + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(t.host + 0, r0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(t.host + 1, r1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        corr_stores_done();  // the sums are in host memory before the ticket that announces them
        __hip_atomic_store(t.host + 2, t.ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// The chunked kernels publish at most a few thousand partials (one per resident workgroup): the workgroup that finishes last
// sums ALL of them itself -- thread-strided, then waves, then thread 0: a fixed order -- instead of the group sums of
// corr_publish_as.  The COUNT stays two-level (kCorrCounters group counters + one on top: 1 536 workgroups finishing within a
// few microseconds on one counter would queue at the memory-side atomic unit), the SUM loses a store -> wait -> count round
// trip: ~2 us less between the last workgroup's last step and the ticket.
template <int THREADS>
__device__ __forceinline__ void corr_publish_flat(int b, double a0, double a1, CorrTail t, double *lds, int *flag) {
    __syncthreads();
    const int groups = t.nblocks < kCorrCounters ? t.nblocks : kCorrCounters;
    const int grp = b % groups;
    const int members = (t.nblocks - grp + groups - 1) / groups;
    if (threadIdx.x == 0) {
        corr_store_agent(t.partials + b, a0);
        corr_store_agent(t.partials + t.nblocks + b, a1);
        corr_stores_done();
        if (t.fenced) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        bool last = corr_count(t.counter + grp) == (unsigned)members - 1u;
        if (last) last = corr_count(t.counter + kCorrCounters) == (unsigned)groups - 1u;
        if (t.fenced && last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        *flag = last;
    }
    __syncthreads();
    if (!*flag) return;
    double r0, r1;
    corr_block_sum<THREADS>(t.partials, t.partials + t.nblocks, t.nblocks, 1, lds, r0, r1);
    if (threadIdx.x == 0) {
        for (int i = 0; i <= kCorrCounters; ++i)
            __hip_atomic_store(t.counter + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(t.host + 0, r0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(t.host + 1, r1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        corr_stores_done();  // the sums are in host memory before the ticket that announces them
        __hip_atomic_store(t.host + 2, t.ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

}  // namespace k
}  // namespace mil

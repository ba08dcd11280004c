// grid_sync.h - device-wide barrier for persistent kernels whose blocks are all resident (at most one per CU).
//
// MI355X: 8 XCDs with private L2s that are not coherent with each other, a per-CU L1 that is never refreshed by other
// CUs' stores (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility").  Data written
// with PLAIN stores before the barrier is handed to plain loads after it by the guide's release / acquire form:
//   every storing wave: s_waitcnt vmcnt(0) -> workgroup barrier -> ONE lane: agent-scope release fence ->
//   s_waitcnt vmcnt(0) (inline asm: the compiler may drop the fence's own wait) -> arrival (agent-scope atomic add) ->
//   relaxed sc1 poll with s_sleep -> ONE agent-scope acquire fence -> s_waitcnt vmcnt(0) -> workgroup barrier.
// The arrival counter is hierarchical (one counter per group of blocks b % 8 - blocks that share an XCD under the
// observed round-robin placement, which matters for speed only -, the group's last arriver adds to the top counter
// every block polls): 256 arrivals on one word serialise at ~12 ns each.
// Two counter sets are used by alternate launches; a launch clears the set of the next one (no memset node per
// launch).  Every wait is bounded: a block that gives up sets the abort word, all blocks return false and the kernel
// drains instead of hanging.
#pragma once
#include <hip/hip_runtime.h>

namespace rbl {

constexpr int GS_STRIDE = 32;                  // one 128-byte line per counter
constexpr int GS_SET = 10 * GS_STRIDE;         // 8 group counters | top | abort
constexpr int GS_UINTS = 2 * GS_SET;           // both sets
constexpr unsigned GS_SPIN_CAP = 1u << 24;

typedef unsigned gs_u32 __attribute__((address_space(1)));

struct GridBarrier {
    unsigned* set;
    unsigned ngroups, gsize, epoch;
};

__device__ inline GridBarrier gs_init(unsigned* bar, int parity) {
    GridBarrier b;
    b.set = bar + parity * GS_SET;
    if (blockIdx.x == 0)    // the other set was used by the previous launch, which has completed
        for (int i = threadIdx.x; i < GS_SET; i += blockDim.x) bar[(parity ^ 1) * GS_SET + i] = 0u;
    b.ngroups = gridDim.x < 8u ? gridDim.x : 8u;
    const unsigned g = blockIdx.x % b.ngroups;
    b.gsize = (gridDim.x - g + b.ngroups - 1) / b.ngroups;
    b.epoch = 0;
    return b;
}

// arrival + wait of ONE lane (the caller's lane 0); returns 0 when some block gave up
__device__ inline int gs_arrive_and_wait(GridBarrier& b) {
    ++b.epoch;
    const unsigned g = blockIdx.x % b.ngroups;
    const unsigned old = __hip_atomic_fetch_add((gs_u32*)b.set + g * GS_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1 == b.epoch * b.gsize)
        (void)__hip_atomic_fetch_add((gs_u32*)b.set + 8 * GS_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned want = b.epoch * b.ngroups;
    unsigned spins = 0;
    while (__hip_atomic_load((gs_u32*)b.set + 8 * GS_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        if (++spins > GS_SPIN_CAP ||
            __hip_atomic_load((gs_u32*)b.set + 9 * GS_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
            __hip_atomic_store((gs_u32*)b.set + 9 * GS_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return 0;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    return 1;
}

// Full barrier with release / acquire of plain stores and loads.  lds_flag: one int of LDS.
__device__ inline bool gs_barrier(GridBarrier& b, int* lds_flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every storing wave
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int ok = gs_arrive_and_wait(b);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        *lds_flag = ok;
    }
    __syncthreads();
    return *lds_flag != 0;
}

}  // namespace rbl

// anakin_amd/csrc/coop_sync.h - the hand-off between cooperating workgroups of ONE XCD (conv_chain_coop.hip, conv_stage_coop.hip):
// arrival counters in the XCD's L2, scalar-load spins, wait-count helpers. Internal to those two translation units.
#pragma once
#include "epilogue_pack.h"

#ifndef SABER_COOP_AUX
#define SABER_COOP_AUX 16      // cache-policy bits of the loads that read a partner's tile: sc1 (scripts/probe/timeline_probe.hip tries others)
#endif

namespace saber_mi355x {

namespace {

typedef int c2i __attribute__((ext_vector_type(2)));

// s_waitcnt vmcnt(N) as the BUILTIN, not inline asm: global_load_lds is a FLAT-encoded instruction that touches LDS, which leaves
// the compiler's wait-count pass in its "pending flat" state - its next vmcnt wait is forced to 0 (the whole weight ring landed
// before the first MFMA) unless it SEES a wait that retires the DMA. (gfx9 encoding: vmcnt [3:0] + [15:14], expcnt [6:4], lgkmcnt [11:8].)
template <int N>
__device__ __forceinline__ void wait_vm_older_than() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ unsigned long long coop_sload(const unsigned long long* p) {
    unsigned long long v;
    asm volatile("s_dcache_inv\n\ts_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(v) : "s"(p) : "memory");
    return v;
}

// The pair's barrier, split in two so that the wait can sit behind work that does not need the partner: ARRIVE (per wave: its
// stores are in the XCD's L2, then one arrival on the pair's counter) ... work on this workgroup's own half ... WAIT (per wave:
// spin on scalar loads - they do not queue behind the wave's weight loads - until all 16 waves of the pair have arrived).
// The counter is never reset: 16 arrivals per launch bring it back to a multiple of 16.
// NO cache invalidate anywhere: `buffer_inv sc1` is a DEVICE-scope acquire, which on this multi-XCD part also drops the L2's
// non-coherent lines - every workgroup passing a barrier wiped its XCD's copy of the weight stream for all 28 workgroups sharing that
// L2 (measured with the in-kernel stamps: 13 us in the first barrier, 14 us for the 1 us third phase at batch 8). What the partner
// wrote is read with sc1 LOADS instead (L2Reader, conv_igemm_impl.h): they miss the L1 and hit the L2.
// The arrival is a WORKGROUP-scope atomic on purpose: it executes in this XCD's L2, which is all the pair needs (both halves run on
// one XCD); an AGENT-scope atomic is performed beyond the L2 on this multi-XCD part and took 1.6 us per arrival (in-kernel stamps).
// The arrival is a NON-returning atomic: a returning one is turned into mbcnt / readfirstlane code by the compiler, whose
// s_waitcnt vmcnt(0) stalled the wave for the atomic's round trip right at the arrival. The wave instead waits for its arrival to be
// performed at the start of coop_wait (one vmcnt for loads, stores and atomics on gfx950), after the work that needs no partner.
__device__ __forceinline__ void coop_arrive(unsigned long long* ctr) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if ((threadIdx.x & 63) == 0) (void)__hip_atomic_fetch_add(ctr, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void coop_wait_mask(const unsigned long long* ctr, unsigned long long mask, unsigned* err);
template <unsigned long long MASK = 15ull>     // arrivals per barrier - 1: 2 workgroups x 8 waves (15) or 4 x 8 (31)
__device__ __forceinline__ void coop_wait(const unsigned long long* ctr, unsigned* err) {
    coop_wait_mask(ctr, MASK, err);
}
__device__ __forceinline__ void coop_wait_mask(const unsigned long long* ctr, unsigned long long mask, unsigned* err) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's own arrival has been PERFORMED before it looks at the counter
    int spins = 0;
    while ((coop_sload(ctr) & mask) != 0ull) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > 200000) {                          // ~20 ms: give up loudly (see the file header)
            if (err && (threadIdx.x & 63) == 0) __hip_atomic_fetch_add(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
        }
    }
}

// Two counters in one loop (the stage kernels' edge barriers: the edge above and the edge below a tile row): each is latched when it is
// first seen at its multiple, so neither sample can be missed while the other is still being waited for.
__device__ __forceinline__ void coop_wait_mask2(const unsigned long long* a, unsigned long long mask_a, const unsigned long long* b,
                                                unsigned long long mask_b, unsigned* err) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's own arrivals have been PERFORMED before it looks at the counters
    int spins = 0;
    bool da = false, db = false;
    for (;;) {
        if (!da) da = (coop_sload(a) & mask_a) == 0ull;
        if (!db) db = (coop_sload(b) & mask_b) == 0ull;
        if (da && db) break;
        __builtin_amdgcn_s_sleep(1);
        if (++spins > 200000) {                          // ~20 ms: give up loudly (see the file header)
            if (err && (threadIdx.x & 63) == 0) __hip_atomic_fetch_add(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
        }
    }
}

// LDS-DMA of 16 bytes per lane with sc1: misses this CU's L1 and reads the XCD's L2 - what another workgroup of the XCD stored earlier
__device__ __forceinline__ void lds_dma16_l2(const void* src, void* dst_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst_wave_base, 16, 0, 16);
}

}  // namespace

}  // namespace saber_mi355x

// 16-byte LDS-DMA loads (global memory -> LDS without a register round trip) for gfx950.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace asrfe {

// LDS[lds_addr + 16 * lane] = *(sbase + voff): scalar 64-bit base, unsigned 32-bit per-lane byte offset.
// Issued through inline assembly on purpose: the compiler models a load that touches both global memory and
// LDS as a flat access and, while one is pending, turns every wait it inserts into vmcnt(0) / lgkmcnt(0); hidden
// from it, its own LDS-read waits stay counted.  The kernel orders these loads itself with counted
// s_waitcnt vmcnt(N) (loads complete in issue order).
// The s_nop is required: on this generation an LDS-DMA instruction that issues in the cycle after a scalar write of M0
// may still read the previous M0 (the compiler's hazard recognizer puts the wait state into code it generates itself,
// but does not look inside an inline-assembly block), and the load then lands at the previous load's LDS address.
__device__ __forceinline__ void lds_dma16(const void* sbase, uint32_t voff, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_addr), "v"(voff), "s"(sbase) : "memory");
}

// The same load in two parts, for loops that count issue cycles: the M0 write, then (with at least one other
// instruction issued in between, which serves as the wait state, and nothing that writes M0) the load itself.  Both are
// volatile asm statements, so the compiler keeps their order relative to each other and to the asm MFMAs a caller puts
// between them; tools/check_vm_queue.py checks the distance in the generated assembly of every kernel.
__device__ __forceinline__ void lds_dma_set_m0(uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %0" ::"s"(lds_addr) : "memory");
}
__device__ __forceinline__ void lds_dma16_at_m0(const void* sbase, uint32_t voff) {
    asm volatile("global_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase) : "memory");
}

// LDS byte address of a pointer into a __shared__ array
__device__ __forceinline__ uint32_t lds_addr_of(const void* p) {
    return static_cast<uint32_t>(reinterpret_cast<uintptr_t>((const __attribute__((address_space(3))) char*)p));
}

}  // namespace asrfe

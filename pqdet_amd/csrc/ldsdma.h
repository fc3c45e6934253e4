// LDS-DMA (global_load_lds_*) helpers shared by the kernels that stream operand tiles HBM/L2 -> LDS without staging
// registers.  The loads are issued from inline asm: hipcc then neither drains vmcnt(0) in front of every ds_read (it does
// for its own builtin, because the DMA "may alias" the LDS array) nor counts them; ordering is by hand:
//   RAW  issuing wave: s_waitcnt vmcnt(N) -> barrier -> any wave may ds_read the tile
//   WAR  a buffer is refilled only after the barrier that follows its last ds_read.
// A DMA instruction moves 64 lanes x 16 B (or 4 B) from per-lane global addresses to ONE contiguous 1 KiB (256 B) block of
// LDS whose byte address is wave-uniform (M0).
#pragma once
#include "common.h"

namespace pq {

__device__ __forceinline__ void glds16(const void* g, unsigned lds_byte) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds_byte) : "memory", "m0");
}
__device__ __forceinline__ void glds4(const void* g, unsigned lds_byte) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(g), "s"(lds_byte) : "memory", "m0");
}
// same, address = wave-uniform 64-bit base (SGPR pair) + per-lane unsigned 32-bit byte offset: no 64-bit VALU address math
__device__ __forceinline__ void glds16s(const void* base, unsigned voff, unsigned lds_byte) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(lds_byte) : "memory", "m0");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const float* p) {
    return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float*)p;
}

}  // namespace pq

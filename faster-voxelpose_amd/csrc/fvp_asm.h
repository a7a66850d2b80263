// The inline-assembly primitives of the LDS-DMA paths (k_conv_dma, k_conv_wino): the only gfx950 instructions the
// sources spell out by hand.  They are asm rather than builtins on purpose: with __builtin_amdgcn_global_load_lds hipcc
// knows that LDS is written behind its back and drains the whole DMA queue (s_waitcnt vmcnt(0)) in front of the next LDS
// read, which disables the chunk ring (Winograd kernel); as asm the instruction is invisible to the waitcnt
// pass and the ordering is what the kernels say (counted s_waitcnt vmcnt + s_barrier).
//
// A replacement <hip/hip_runtime.h> that cannot assemble gfx950 code - the CPU execution-model emulator of tests/hipemu -
// defines FVP_ASM_PRIMITIVES_PROVIDED and supplies the same four names with the measured hardware semantics
// (tools/micro/buflds.hip); the kernel sources themselves contain no emulator code.
#pragma once

#ifndef FVP_ASM_PRIMITIVES_PROVIDED

// Byte address of an LDS array as the DMA's M0 base wants it.  A macro on the bare array: the generic -> LDS cast of the
// array folds to a constant, while the cast of a pointer VARIABLE makes this hipcc emit a null check that it mis-selects.
#define FVP_LDS_BYTE_ADDRESS(arr) unsigned(size_t((const __attribute__((address_space(3))) float*)(arr)))

namespace fvp {

typedef int fvp_i32x4 __attribute__((ext_vector_type(4)));

// global_load_lds_dwordx4: lane l copies 16 bytes from its global address g to LDS byte address lds_addr + 16 l
// (lds_addr wave-uniform)
__device__ __forceinline__ void asm_global_load_lds16(const float* g, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(g), "s"(lds_addr) : "memory", "m0");
}

// buffer_load_dwordx4 ... offen lds (raw buffer rs, stride 0): lane l copies 16 bytes from rs.base + vo + so to LDS byte
// address la + 16 l; every dword whose offset fails the range check (scalar offset included) is written as zero
__device__ __forceinline__ void asm_buffer_load_lds16(unsigned la, unsigned vo, const fvp_i32x4& rs, unsigned so) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :
               : "s"(la), "v"(vo), "s"(rs), "s"(so)
               : "memory", "m0");
}

}  // namespace fvp

#endif  // FVP_ASM_PRIMITIVES_PROVIDED

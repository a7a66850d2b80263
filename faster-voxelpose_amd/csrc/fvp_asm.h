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

// buffer_load_dwordx2 / buffer_store_dword(x2) ... offen through a raw descriptor held as four SGPRs (the Winograd
// epilogue: the descriptor of the unit's first plane, ONE per-lane byte offset for every access of the lane and a scalar
// byte offset per (cout, row) - no address arithmetic per access; a per-lane offset with bit 31 set fails the range check:
// loads return 0, stores are dropped).  As asm the accesses are invisible to hipcc's waitcnt pass: the CALLER waits
// (s_waitcnt vmcnt) before it uses a loaded value - the epilogue's single vmcnt(0) does.
typedef float fvp_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ fvp_f32x2 asm_buffer_load_f32x2(unsigned vo, const fvp_i32x4& rs, unsigned so) {
  fvp_f32x2 v;
  asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen" : "=v"(v) : "v"(vo), "s"(rs), "s"(so) : "memory");
  return v;
}
__device__ __forceinline__ void asm_buffer_store_f32x2(fvp_f32x2 v, unsigned vo, const fvp_i32x4& rs, unsigned so) {
  asm volatile("buffer_store_dwordx2 %0, %1, %2, %3 offen" : : "v"(v), "v"(vo), "s"(rs), "s"(so) : "memory");
}
__device__ __forceinline__ void asm_buffer_store_f32(float v, unsigned vo, const fvp_i32x4& rs, unsigned so) {
  asm volatile("buffer_store_dword %0, %1, %2, %3 offen" : : "v"(v), "v"(vo), "s"(rs), "s"(so) : "memory");
}

// ---- raw buffer addressing for ordinary loads / stores (k_conv_reg): descriptor in 4 SGPRs + ONE per-lane 32-bit byte
// offset + a scalar byte offset per row.  "Uniform 64-bit row pointer + per-lane offset" - what round 4 wrote - costs an
// SGPR PAIR per row in flight (64 rows at K = 128: the kernel spilled 143-628 SGPRs); here a row is one scalar add.
// Range check (raw buffer, stride 0): a dword whose voffset + soffset + 4 exceeds num_records reads as 0 / is not written,
// which replaces the `cout < a.cout` predicates of padded output rows.
typedef __amdgpu_buffer_rsrc_t fvp_rsrc;
__device__ __forceinline__ fvp_rsrc make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, int(bytes), 0x00020000);
}
__device__ __forceinline__ float buf_load_f32(fvp_rsrc r, unsigned vo, unsigned so) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, int(vo), int(so), 0));
}
// (two dword loads: in this hipcc - ROCm 7.2 - __builtin_amdgcn_raw_buffer_load_b64 / _b96 / _b128 all lower to the i32
// intrinsic and return element 0 in every lane of the result; found on the GPU, tools/micro/bufrange.hip prints it)
__device__ __forceinline__ float2 buf_load_f32x2(fvp_rsrc r, unsigned vo, unsigned so) {
  return make_float2(buf_load_f32(r, vo, so), buf_load_f32(r, vo + 4u, so));
}
__device__ __forceinline__ void buf_store_f32(float x, fvp_rsrc r, unsigned vo, unsigned so) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, x), r, int(vo), int(so), 0);
}
__device__ __forceinline__ void buf_store_f32x2(float2 x, fvp_rsrc r, unsigned vo, unsigned so) {
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  const u32x2 v = {__builtin_bit_cast(unsigned, x.x), __builtin_bit_cast(unsigned, x.y)};
  __builtin_amdgcn_raw_buffer_store_b64(v, r, int(vo), int(so), 0);
}

}  // namespace fvp

#endif  // FVP_ASM_PRIMITIVES_PROVIDED

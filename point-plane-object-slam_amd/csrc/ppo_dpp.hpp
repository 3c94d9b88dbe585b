// ppo_dpp.hpp -- FP64 DPP helpers shared by the dense factorisation (ppo_chol.hip: 16x16 register factorisations) and the cuboid blocks of
// the reduction (ppo_kernels.hpp: 9x9 factorisation + inverse, one cuboid per row of 16 lanes).
#pragma once
#include <hip/hip_runtime.h>

namespace ppo {
#ifdef __HIPCC__
// ---- f64 DPP: v_fmac_f64 / v_mov_b64 accept `row_newbcast:N` (source operand taken from lane N of the reader's own row of
// 16 lanes), which turns "broadcast + multiply-add" into ONE instruction.  The wave that factors a 16x16 block is issue-bound
// (tools/chol_probe.hip: ~4.5 cycles per instruction whatever the dependencies), so instruction count is what matters: the
// v_readlane formulation needed 2 broadcasts + 1 FMA per element update.  `s_nop 1` covers the VALU-write -> DPP-read hazard,
// which the compiler cannot see inside inline assembly.
template <int C, bool FRESH>
__device__ __forceinline__ void fmac_bcast(double& acc, double l_src, double mul) {  // acc -= l_src[lane C of my row] * mul
  if constexpr (FRESH)  // l_src was written by the preceding VALU instruction
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(l_src), "v"(mul), "n"(C));
  else
    asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(l_src), "v"(mul), "n"(C));
}
template <int C>
__device__ __forceinline__ double row_bcast(double v) {
  double d;
  asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(d) : "v"(v), "n"(C));
  return d;
}

#endif  // __HIPCC__
}  // namespace ppo

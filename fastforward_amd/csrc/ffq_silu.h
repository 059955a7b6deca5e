// ffq_silu.h — F.silu on a bf16 tensor as a table in LDS.
//
// The reference's MLP (docs/examples/doc_helpers/quantized_llama/mlp.py:36-38) applies F.silu to the bf16 output of
// gate_proj: ATen evaluates x / (1 + exp(-x)) in fp32 and rounds to bf16. A function of a bf16 argument has 65536 values,
// so the ~30 VALU instructions of the exact expf + IEEE division per element (what made silu_mul_quantize VALU-bound at
// 50 % of the HBM rate and cost the gate+up GEMM launch 7.5 %) are replaced by one 2-byte LDS read:
//   * 32 binades of |x|, 2^-24 <= |x| < 2^8, both signs: 8192 entries = 16 KiB, FILLED BY THE KERNEL THAT USES THEM
//     with the exact expression below (no state outside the launch, nothing to initialise, graph-capture safe);
//   * |x| < 2^-24: exp(-x) rounds to within one ulp of 1, the denominator rounds to exactly 2 (ties to even), so the
//     value is x / 2, whose bf16 rounding (denormals) is the one rounding the exact expression would do;
//   * |x| >= 2^8, Inf, NaN: exp(-x) underflows or overflows and the quotient is x, -0 or NaN (silu_outside below).
// The values outside the window are patched under a wave-uniform branch nobody takes on ordinary activations. Whatever
// the table holds was produced by silu_exact on this device, and tests/test_parity_gpu.py feeds all 65536 bf16 patterns
// through the kernels that use it and compares with ATen's silu, so the equality is checked over the whole domain.
#pragma once
#include "ffq_common.h"
#include "ffq_vec.h"

#include <math.h>

namespace ffq {

constexpr uint32_t kSiluBase = 103u << 7;    // bf16 pattern of 2^-24
constexpr uint32_t kSiluSpan = 32u << 7;     // 32 binades of 128 mantissas
constexpr uint32_t kSiluEntries = 2 * kSiluSpan;
constexpr uint32_t kSiluBytes = kSiluEntries * 2;

__device__ __forceinline__ float silu_exact(float x) { return x / (1.0f + expf(-x)); }  // ATen's silu kernel in fp32

// all `nthreads` threads of the block; the caller's next barrier publishes the table
__device__ __forceinline__ void silu_table_fill(uint16_t* table, uint32_t tid, uint32_t nthreads) {
  for (uint32_t e = tid; e < kSiluEntries; e += nthreads) {
    const uint32_t u = ((e >> 12) << 15) | (kSiluBase + (e & (kSiluSpan - 1)));
    const float x = __builtin_bit_cast(float, u << 16);
    table[e] = (uint16_t)pack2<bf16_t>(silu_exact(x), 0.0f);
  }
}

// bf16(silu(x)) for the two bf16 values packed in w, packed the same way — in two steps, so that a caller can issue the
// LDS reads of a whole batch of pairs back to back and branch ONCE for the batch (a branch per pair serialises the reads):
//   r[j] = silu_pair_lookup(w[j], table, bad) for the batch;  if (silu_any_outside(bad)) r[j] = silu_pair_patch(w[j], r[j]).
// `bad` collects bits 13..15 of 2 (|u| - base) mod 2^16, which are clear exactly inside the window; the byte address is
// ((2 u - 2 base) & 0x1FFE) | (sign << 13), formed on the packed word without separating the halves first (the stray low
// bit of w >> 15 falls to the mask). Values outside the window read some entry of the table and are patched afterwards.
__device__ __forceinline__ uint32_t silu_pair_lookup(uint32_t w, const uint16_t* table, uint32_t& bad) {
  const uint32_t a0 = (w << 1) - 2u * kSiluBase, a1 = (w >> 15) - 2u * kSiluBase;
  bad |= a0 | a1;
  const unsigned char* t = reinterpret_cast<const unsigned char*>(table);
  const uint32_t r0 = *reinterpret_cast<const uint16_t*>(t + ((a0 & 0x1FFEu) | ((w >> 2) & 0x2000u)));
  const uint32_t r1 = *reinterpret_cast<const uint16_t*>(t + ((a1 & 0x1FFEu) | ((w >> 18) & 0x2000u)));
  return r0 | (r1 << 16);
}
__device__ __forceinline__ bool silu_any_outside(uint32_t bad) { return __any((bad & 0xE000u) != 0); }

// Outside the window the exact expression has closed forms (each one is what x / (1 + expf(-x)) evaluates to in fp32):
//   |x| < 2^-24          x / 2       (see the header)
//   x >= 2^8, +Inf       x           exp(-x) underflows below half an ulp of 1
//   -Inf < x <= -2^8     -0          exp(-x) overflows: x / Inf
//   -Inf, NaN            NaN         -Inf / Inf
__device__ __forceinline__ uint32_t silu_outside(uint32_t u, uint32_t half_x) {  // 16-bit patterns
  const uint32_t mag = u & 0x7FFFu;
  if (mag < kSiluBase) return half_x;
  if (mag > 0x7F80u) return u | 0x0040u;
  if (!(u & 0x8000u)) return u;
  return mag == 0x7F80u ? 0x7FC0u : 0x8000u;
}
__device__ __forceinline__ uint32_t silu_pair_patch(uint32_t w, uint32_t r) {
  const uint32_t u0 = w & 0xFFFFu, u1 = w >> 16;
  const bool in0 = (u0 & 0x7FFFu) - kSiluBase < kSiluSpan, in1 = (u1 & 0x7FFFu) - kSiluBase < kSiluSpan;
  const uint32_t h = pack2<bf16_t>(__builtin_bit_cast(float, w << 16) * 0.5f, __builtin_bit_cast(float, w & 0xFFFF0000u) * 0.5f);
  const uint32_t e0 = silu_outside(u0, h & 0xFFFFu), e1 = silu_outside(u1, h >> 16);
  return (in0 ? r & 0xFFFFu : e0) | (in1 ? r & 0xFFFF0000u : e1 << 16);
}

}  // namespace ffq

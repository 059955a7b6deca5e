// ffq_affine.h — the per-element arithmetic of A1 (quantize) shared by every kernel that produces
// integer codes: the streaming quantizers (ffq_quantize.hip) and the producer-fused quantizers
// (ffq_producers.hip). Reference: quantize_by_tile_impl, _quantizer_impl.py:154-169.
#pragma once

#include "ffq_common.h"
#include "ffq_vec.h"

namespace ffq {

// x / s, correctly rounded. DIVMODE 0 is the compiler's IEEE sequence (v_div_scale / v_rcp /
// v_fma x4 / v_div_fmas / v_div_fixup, ~11 VALU + hazard nops per element). DIVMODE 1 replaces it
// by Markstein's FMA iteration around r = RN(1/s), which is computed once per chunk with the IEEE
// sequence: q0 = RN(x r) is within 1.5 ulp, the first residual step makes it faithful, the second
// one makes it the correctly rounded quotient (Markstein 1990; Muller et al., Handbook of
// Floating-Point Arithmetic, division by FMA iteration: y = RN(1/b) and q faithful imply
// RN(q + RN(a - b q) y) = RN(a/b)). The theorem needs the residuals a - b q to be exact, i.e. no
// underflow, hence the guards: the iteration is used only for 2^-40 < |s| < 2^40 AND
// 2^-40 < |q0| < 2^40 (so |x| > 2^-80 and every residual bit is a normal number). Outside:
//   |q0| >= 2^40 (or Inf/NaN): every candidate clamps to the same bound; q0 carries sign/Inf/NaN;
//   |q0| <= 2^-40 (incl. +-0):  round(q0 - o) cannot depend on the last bit of q0; q0 keeps the
//                               sign of the quotient, which decides between -0.0 and +0.0;
//   |s| outside the window:     the whole chunk takes the IEEE sequence.
// tests/test_parity_gpu.py compares both modes bit-for-bit on adversarial data.
template <int DIVMODE>
struct Divider {
  float s, r;
  bool safe;
  __device__ __forceinline__ explicit Divider(float s_) : s(s_), r(0.0f), safe(false) {
    if constexpr (DIVMODE == 1) {
      r = 1.0f / s;
      const float as = __builtin_fabsf(s);
      safe = as > 0x1p-40f && as < 0x1p40f;
    }
  }
  __device__ __forceinline__ float fast(float x) const {
    const float q0 = x * r;
    const float q1 = __builtin_fmaf(__builtin_fmaf(-q0, s, x), r, q0);
    const float q2 = __builtin_fmaf(__builtin_fmaf(-q1, s, x), r, q1);
    const float a0 = __builtin_fabsf(q0);
    return (a0 > 0x1p-40f && a0 < 0x1p40f) ? q2 : q0;
  }
};

// round(x / s - o) for E elements sharing one parameter pair; the divider (its 1/s costs an IEEE division) can be shared
// between chunks that hold the same scale.
template <int DIVMODE, int E>
__device__ __forceinline__ void quantize_chunk_with(const Divider<DIVMODE>& d, const float (&x)[E], float o, float (&r)[E]) {
  if (DIVMODE == 1 && d.safe) {
#pragma unroll
    for (int i = 0; i < E; ++i) r[i] = rne(d.fast(x[i]) - o);
  } else {
#pragma unroll
    for (int i = 0; i < E; ++i) r[i] = rne(x[i] / d.s - o);  // separate roundings: -ffp-contract=off
  }
}
template <int DIVMODE, int E>
__device__ __forceinline__ void quantize_chunk(const float (&x)[E], float s, float o, float (&r)[E]) {
  const Divider<DIVMODE> d(s);
  quantize_chunk_with<DIVMODE, E>(d, x, o, r);
}

// A1 for a chunk whose quotients are known to be ordinary numbers — the form the VALU-bound kernels take on their common path.
// quantize_chunk + finalize_chunk cost ~17 VALU ops per element into a byte container (unpack, 5 for the division, 3-4 for the
// window test, subtract, round, convert, clamp, pack): 58.7 M elements x 17 ops / (1024 SIMDs x 16 lanes) is 27 us of issue at
// 2.2 GHz — these kernels are as much VALU-bound as HBM-bound. Here the same values take ~6:
//   * the Markstein iteration (see Divider) on PAIRS of elements: v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32 are full-rate on
//     two fp32 lanes, and each lane's result is the same IEEE operation as the scalar instruction;
//   * no window test per element: the CALLER guarantees 2^-40 < |s| < 2^40 and |x r| < 2^39 for every element of the chunk
//     (no Inf, no NaN, no overflow: the iteration's residuals are exact or — below 2^-40 — irrelevant to an INTEGER container:
//     whatever the last bits of a quotient of that size, it rounds to 0 after the integer offset is subtracted; the sign of a
//     zero, which the float containers keep, is why those stay on quantize_chunk);
//   * clamp first (v_med3_f32; clamp and round-half-even commute for integer bounds), then round AND convert in one packed add of
//     1.5 * 2^23: the sum's mantissa holds the two's-complement integer in its low bits, rounded half-even by the adder itself
//     (|v| <= 2^15 after the clamp, far inside the trick's 2^22 range);
//   * the low bytes of four such words are one int8 quadruple: three v_perm_b32.
// Bit-identical to quantize_chunk + finalize_chunk on every chunk that meets the precondition
// (tests/test_parity_gpu.py::test_fast_chunk_arithmetic_equals_the_reference_chain).
typedef float fq_f32x2 __attribute__((ext_vector_type(2)));
// CHECK: `*check` receives the sum of the E values x / s - o; it is NaN whenever the precondition on x failed (an Inf or NaN
// element, or x r overflowing: each of these turns the iteration's residual into Inf - Inf), so a caller that knows nothing
// about its elements tests `check == check` once per chunk instead of three VALU ops per element.
template <int E, bool CHECK>
__device__ __forceinline__ void quantize_chunk_bytes_fast(const float (&x)[E], float s, float r, float o, float lo, float hi,
                                                          uint32_t (&words)[E / 4], float* check = nullptr) {
  static_assert(E % 4 == 0, "whole dwords of codes");
  const fq_f32x2 S = {s, s}, R = {r, r}, O = {o, o}, M = {12582912.0f, 12582912.0f};
  fq_f32x2 acc = {0.0f, 0.0f};
  uint32_t b[E];
#pragma unroll
  for (int i = 0; i < E; i += 2) {
    const fq_f32x2 X = {x[i], x[i + 1]};
    const fq_f32x2 q0 = X * R;
    const fq_f32x2 q1 = __builtin_elementwise_fma(__builtin_elementwise_fma(-q0, S, X), R, q0);
    const fq_f32x2 q2 = __builtin_elementwise_fma(__builtin_elementwise_fma(-q1, S, X), R, q1);
    const fq_f32x2 d = q2 - O;
    if constexpr (CHECK) acc = acc + d;
    const fq_f32x2 c = {__builtin_amdgcn_fmed3f(d.x, lo, hi), __builtin_amdgcn_fmed3f(d.y, lo, hi)};
    const fq_f32x2 e = c + M;
    // (through scalar temporaries: __builtin_bit_cast applied to a vector ELEMENT reads element 0 with this hipcc)
    const float e0 = e.x, e1 = e.y;
    b[i] = __builtin_bit_cast(uint32_t, e0);
    b[i + 1] = __builtin_bit_cast(uint32_t, e1);
  }
#pragma unroll
  for (int i = 0; i < E; i += 4) {
    const uint32_t t01 = __builtin_amdgcn_perm(b[i + 1], b[i], 0x0c0c0400u);      // [b0.0, b1.0, 0, 0]
    const uint32_t t23 = __builtin_amdgcn_perm(b[i + 3], b[i + 2], 0x0c0c0400u);  // [b2.0, b3.0, 0, 0]
    words[i >> 2] = __builtin_amdgcn_perm(t23, t01, 0x05040100u);
  }
  if constexpr (CHECK) *check = acc.x + acc.y;
}
// the caller's precondition for one scale and a bound on |x| over the elements it covers (an Inf / NaN bound fails it)
__device__ __forceinline__ bool fast_chunk_ok(float s, float r, float abs_x_max) {
  const float as = __builtin_fabsf(s);
  return as > 0x1p-40f && as < 0x1p40f && abs_x_max * __builtin_fabsf(r) < 0x1p39f;
}

// clamp + cast of E rounded values. Float containers: v_med3_f32 with NaN passed through
// (torch.clamp propagates NaN). Integer containers: convert first (v_cvt_i32_f32 saturates and
// maps NaN to 0, the value the reference's CPU cast yields for int8/int16), then v_med3_i32.
template <typename TOut, int E>
__device__ __forceinline__ void finalize_chunk(const float (&r)[E], float lo, float hi, Chunk<TOut, E>& y) {
  if constexpr (TypeTag<TOut>::value == FFQ_I8 || TypeTag<TOut>::value == FFQ_I16 ||
                TypeTag<TOut>::value == FFQ_I32) {
    const int ilo = (int)lo, ihi = (int)hi;
    int c[E];
#pragma unroll
    for (int i = 0; i < E; ++i) {
      int v = (int)r[i];
      v = v < ilo ? ilo : (v > ihi ? ihi : v);
      if constexpr (TypeTag<TOut>::value == FFQ_I32) v = r[i] != r[i] ? INT32_MIN : v;
      c[i] = v;
    }
    y.pack_int(c);
  } else {
    float c[E];
#pragma unroll
    for (int i = 0; i < E; ++i) c[i] = r[i] != r[i] ? r[i] : __builtin_amdgcn_fmed3f(r[i], lo, hi);
    y.pack(c);
  }
}

// A1 of one chunk into a BYTE container, parameters (s, o) with o already rounded: the packed form when its self-check passes
// (always, on ordinary data), the reference chain otherwise — same codes either way.
template <int E>
__device__ __forceinline__ void quantize_chunk_to_bytes(const float (&x)[E], float s, float o, float lo, float hi,
                                                        Chunk<int8_t, E>& y) {
  const Divider<1> d(s);
  if (d.safe) {
    float check;
    quantize_chunk_bytes_fast<E, true>(x, s, d.r, o, lo, hi, y.w, &check);
    if (check == check) return;
  }
  float r[E];
  quantize_chunk_with<1, E>(d, x, o, r);
  finalize_chunk<int8_t, E>(r, lo, hi, y);
}

// ---- the same tensor, quantized again by a quantizer whose parameters may or may not be the earlier one's -----------------------
// q_proj / k_proj / v_proj (gate_proj / up_proj) each quantize the same hidden state with their own input quantizer (reference
// nn/linear.py:32-39). While range estimators rewrite the parameters on every step the host cannot know whether two of them are
// equal without reading them back, so the question is put to the device: A1's codes are a function of the scale's bits and of the
// ROUNDED offset (_quantizer_impl.py:140-141), hence two quantizers that agree in both produce the same bytes. The later one's A1
// launch (ffq_quantize_by_tile_unless_same) returns before its first load when they agree, and whoever consumes its codes is given
// the earlier quantizer's as well and asks the same question (`codes_in_force`). Whether or not the later launch has written, the
// earlier codes ARE the later quantizer's codes whenever the answer is yes — a consumer may always take them then.
struct EarlierCodes {
  const int8_t* codes;  // nullptr: there is no earlier quantizer
  const float* scale;   // one element each (per-tensor quantizers); offset nullable = 0
  const float* offset;
};

__device__ __forceinline__ bool same_parameters(const float* scale, const float* offset, const float* scale2, const float* offset2) {
  const float o = offset ? rne(offset[0]) : 0.0f, o2 = offset2 ? rne(offset2[0]) : 0.0f;
  // (a NaN offset is "not the same": the later quantizer then runs as itself, whatever it makes of the NaN)
  return __builtin_bit_cast(uint32_t, scale[0]) == __builtin_bit_cast(uint32_t, scale2[0]) && o == o2;
}

__device__ __forceinline__ const int8_t* codes_in_force(const int8_t* own, const float* scale, const float* offset, const EarlierCodes& e) {
  return e.codes && same_parameters(scale, offset, e.scale, e.offset) ? e.codes : own;
}

}  // namespace ffq

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

}  // namespace ffq

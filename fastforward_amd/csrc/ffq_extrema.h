// ffq_extrema.h — a producer kernel leaves [min, max] of the bf16 tensor it writes (A4 of its own output, fused).
//
// During range estimation every per-tensor activation quantizer starts with a reduction over its input (RunningMinMax,
// reference range_setting/minmax.py:215-239) — a tensor some kernel of this library wrote a moment ago. That kernel sees every
// value on its way out: each lane keeps a running min / max (NaN apart: torch.min / torch.max propagate it), the block reduces
// them, ONE thread per block merges the block's result into three words with integer atomics on order-preserving keys, takes a
// ticket, and the last block to arrive writes the pair in the data dtype and puts the words back to their initial state. The
// estimator step then merges two numbers (ffq_running_minmax_step on a 2-element tensor) instead of reading the tensor again.
// Same values as ffq_minmax_by_tile on the finished tensor: min and max are exact, the keys order -0.0 below +0.0 as the 16-bit
// pattern accumulators of ffq_minmax.hip do.
#pragma once
#include "ffq_common.h"

namespace ffq {

// words: [0] min key (initially 0xFFFFFFFF), [1] max key (0), [2] NaN seen (0), [3] arrivals (0) — left in that state by every launch
struct ExtremaSink {
  uint32_t* words;
  void* pair;     // [min, max] in `pair_dt`
  int pair_dt;
};

__device__ __forceinline__ uint32_t extrema_key(float f) {
  const uint32_t u = __builtin_bit_cast(uint32_t, f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float extrema_value(uint32_t key) {
  return __builtin_bit_cast(float, (key & 0x80000000u) ? (key ^ 0x80000000u) : ~key);
}

// the words -> the pair, and the words back to their initial state (one thread, after every block has merged)
__device__ __forceinline__ void extrema_finish(const ExtremaSink& s) {
  const uint32_t kmin = __hip_atomic_load(s.words + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const uint32_t kmax = __hip_atomic_load(s.words + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const uint32_t knan = __hip_atomic_load(s.words + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // (nothing merged: +inf / -inf, the estimator's own initial state)
  const float lo = knan ? NAN : (kmin == 0xFFFFFFFFu ? INFINITY : extrema_value(kmin));
  const float hi = knan ? NAN : (kmax == 0u ? -INFINITY : extrema_value(kmax));
  store_any(s.pair, s.pair_dt, 0, (double)lo);
  store_any(s.pair, s.pair_dt, 1, (double)hi);
  __hip_atomic_store(s.words + 0, 0xFFFFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(s.words + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(s.words + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(s.words + 3, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One thread per block, after the block's reduction. A block first LOOKS at the words and merges only what would move them: a
// stale look costs an atomic that changes nothing, never a wrong result (min / max are idempotent), and after the first few
// arrivals almost no block of a 16 k-block launch has anything to add — the words are read-shared, not a serialised hot spot.
// `arrivals` = blocks of the launch that call this (all of them): the last one finishes.
// (Tried on the RMSNorm producer, 16 k one-row blocks with a one-thread finishing launch: the kernel went from 84 to 142 us, more
// than the 26 us reduction it replaces — the extra VALU work and the block-end reduction do not hide in a kernel that short.)
__device__ __forceinline__ void extrema_publish(const ExtremaSink& s, float mn, float mx, bool nan, bool any, uint32_t arrivals) {
  if (any) {  // (returning atomics: complete at L2 before the ticket below is taken)
    const uint32_t kmn = extrema_key(mn), kmx = extrema_key(mx);
    if (kmn < __hip_atomic_load(s.words + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
      (void)__hip_atomic_fetch_min(s.words + 0, kmn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (kmx > __hip_atomic_load(s.words + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
      (void)__hip_atomic_fetch_max(s.words + 1, kmx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (nan && __hip_atomic_load(s.words + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u)
      (void)__hip_atomic_fetch_or(s.words + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const uint32_t t = __hip_atomic_fetch_add(s.words + 3, 1u, FFQ_TICKET_ORDER, __HIP_MEMORY_SCOPE_AGENT);
  if (t != arrivals - 1) return;
  asm volatile("" ::: "memory");
  extrema_finish(s);
}

}  // namespace ffq

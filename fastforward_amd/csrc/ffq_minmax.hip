// ffq_minmax.hip — A4 (RunningMinMax reduction), A5 (range -> scale/offset), A3 (dynamic quantize).
//
// References
//   A4  RunningMinMaxEstimator.estimate_step, src/fastforward/range_setting/minmax.py:215-239
//       (torch.min + torch.max over a tiles_to_rows view, isinf().any() host sync, running merge)
//   A5  parameters_for_range, src/fastforward/quantization/affine/range.py:54-122, plus the copy in
//       LinearQuantizer.quantization_range.setter, src/fastforward/nn/linear_quantizer.py:350-357
//   A3  quantize_dynamic_by_tile_impl, src/fastforward/quantization/_quantizer_impl.py:243-285
//
// The reference reads the tensor twice (min, then max) and synchronises with the host twice per
// quantizer per step. Here one pass produces both extrema (2 B/elem for bf16), the Inf test becomes
// a flag word on the device, and the global one-sided decision of A5 is taken inside the kernel,
// so a calibration step enqueues without ever waiting for the GPU.
//
// Reduction shape: 16 B per lane loads, min/max in VGPRs, NaN tracked as a wave-level predicate,
// wavefront (64-lane) xor-shuffle butterflies, one LDS hop across the 4 waves of a block.
#include "ffq_affine.h"
#include "ffq_common.h"
#include "ffq_vec.h"

#include <math.h>
#include <stdlib.h>

namespace ffq {

int quantize_impl(const void* data, int data_dt, const void* scale, int scale_dt, int64_t scale_numel,
                  const void* offset, int offset_dt, int64_t offset_numel, const ffq_tiling* tiling,
                  double num_bits, void* out, int out_dt, hipStream_t stream);

// torch.min / torch.max propagate NaN; v_min_f32 / v_max_f32 do not, so NaN travels separately.
struct MinMax {
  float mn, mx;
  bool nan;
  __device__ __forceinline__ void init() { mn = INFINITY; mx = -INFINITY; nan = false; }
  __device__ __forceinline__ void add(float v) {
    mn = __builtin_fminf(mn, v);
    mx = __builtin_fmaxf(mx, v);
    nan |= v != v;
  }
  __device__ __forceinline__ void merge(const MinMax& o) {
    mn = __builtin_fminf(mn, o.mn);
    mx = __builtin_fmaxf(mx, o.mx);
    nan |= o.nan;
  }
};

// bf16 / fp16 data: extrema on the RAW 16-bit patterns, two elements per VALU op and no conversion, no NaN test per
// element (6 VALU ops per element become 2). Sign-magnitude patterns order like this:
//   as signed int16   positives order like their values and beat every negative; among negatives the order is reversed
//   as unsigned int16 negatives (>= 0x8000) beat every positive and order by magnitude
// so  max = (signed max >= 0) ? signed max : signed min      (no positive element: the negative of least magnitude)
//     min = (unsigned max >= 0x8000) ? unsigned max : unsigned min
// and a NaN (exponent all ones, mantissa != 0) is the only pattern above +Inf in the signed order / above -Inf in the
// unsigned order. Four v_pk_{max,min}_{i16,u16} per dword.
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
template <typename T>
struct PackedMinMax16 {
  static_assert(sizeof(T) == 2, "16-bit floating point only");
  static constexpr int kPosInf = TypeTag<T>::value == FFQ_BF16 ? 0x7F80 : 0x7C00;
  s16x2 imax, imin;
  u16x2 umax, umin;
  bool any;
  __device__ __forceinline__ void init() {
    imax = s16x2{(short)-32768, (short)-32768}; imin = s16x2{(short)32767, (short)32767};
    umax = u16x2{0, 0}; umin = u16x2{(unsigned short)0xFFFF, (unsigned short)0xFFFF};
    any = false;
  }
  __device__ __forceinline__ void add(uint32_t w) {
    const s16x2 si = __builtin_bit_cast(s16x2, w);
    const u16x2 ui = __builtin_bit_cast(u16x2, w);
    imax = __builtin_elementwise_max(imax, si); imin = __builtin_elementwise_min(imin, si);
    umax = __builtin_elementwise_max(umax, ui); umin = __builtin_elementwise_min(umin, ui);
    any = true;
  }
  __device__ __forceinline__ MinMax finish() const {
    MinMax m;
    m.init();
    if (!any) return m;
    const int smax = imax.x > imax.y ? imax.x : imax.y, smin = imin.x < imin.y ? imin.x : imin.y;
    const unsigned xmax = umax.x > umax.y ? umax.x : umax.y, xmin = umin.x < umin.y ? umin.x : umin.y;
    const uint16_t mx_bits = (uint16_t)(smax >= 0 ? smax : smin);
    const uint16_t mn_bits = (uint16_t)(xmax >= 0x8000u ? xmax : xmin);
    m.mx = to_f32(__builtin_bit_cast(T, mx_bits));
    m.mn = to_f32(__builtin_bit_cast(T, mn_bits));
    m.nan = smax > kPosInf || xmax > (0x8000u | (unsigned)kPosInf);
    return m;
  }
};

// E elements of one chunk into the accumulator of the data type: packed patterns for 16-bit floats, floats otherwise.
template <typename T> struct Accum { typedef MinMax type; };
template <> struct Accum<bf16_t> { typedef PackedMinMax16<bf16_t> type; };
template <> struct Accum<f16_t> { typedef PackedMinMax16<f16_t> type; };
template <typename T, int E>
__device__ __forceinline__ void add_chunk(typename Accum<T>::type& acc, const Chunk<T, E>& x) {
  if constexpr (sizeof(T) == 2) {
#pragma unroll
    for (int i = 0; i < Chunk<T, E>::kWords; ++i) acc.add(x.w[i]);
  } else {
#pragma unroll
    for (int i = 0; i < E; ++i) acc.add(x.get(i));
  }
}
template <typename T>
__device__ __forceinline__ MinMax finish_accum(const typename Accum<T>::type& acc) {
  if constexpr (sizeof(T) == 2) return acc.finish(); else return acc;
}

// butterfly over `width` lanes (power of two <= 64) of a wavefront
template <int WIDTH>
__device__ __forceinline__ void wave_reduce(MinMax& m) {
#pragma unroll
  for (int d = WIDTH / 2; d >= 1; d >>= 1) {
    const float omn = __shfl_xor(m.mn, d, 64);
    const float omx = __shfl_xor(m.mx, d, 64);
    const int onan = __shfl_xor((int)m.nan, d, 64);
    m.mn = __builtin_fminf(m.mn, omn);
    m.mx = __builtin_fmaxf(m.mx, omx);
    m.nan |= onan != 0;
  }
}

// The same butterfly on the VALU where the data-parallel primitives reach (quad_perm xor 1 / xor 2, row_half_mirror, row_mirror:
// every lane of a 16-lane row ends up with the row's result in 4 steps of ~2 cycles each) and lane reads for the rest: a
// __shfl_xor is a ds_bpermute_b32, i.e. an LDS-pipe round trip per step and value (18 of them for a wave's min, max and NaN
// flag) — the latency chain of a block that lives for one row. Every lane of the group holds the result.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false); }
template <int CTRL>
__device__ __forceinline__ void dpp_step(float& mn, float& mx, int& nan) {
  mn = __builtin_fminf(mn, dpp_f<CTRL>(mn));
  mx = __builtin_fmaxf(mx, dpp_f<CTRL>(mx));
  nan |= dpp_i<CTRL>(nan);
}
template <int WIDTH>
__device__ __forceinline__ void wave_allreduce(MinMax& m) {
  float mn = m.mn, mx = m.mx;
  int nan = m.nan ? 1 : 0;
  if constexpr (WIDTH >= 2) dpp_step<0xB1>(mn, mx, nan);    // quad_perm [1,0,3,2]: lane ^ 1
  if constexpr (WIDTH >= 4) dpp_step<0x4E>(mn, mx, nan);    // quad_perm [2,3,0,1]: lane ^ 2
  if constexpr (WIDTH >= 8) dpp_step<0x141>(mn, mx, nan);   // row_half_mirror: lane -> 7 - lane within 8
  if constexpr (WIDTH >= 16) dpp_step<0x140>(mn, mx, nan);  // row_mirror: lane -> 15 - lane within 16
  if constexpr (WIDTH == 32) {
    mn = __builtin_fminf(mn, __shfl_xor(mn, 16, 64));
    mx = __builtin_fmaxf(mx, __shfl_xor(mx, 16, 64));
    nan |= __shfl_xor(nan, 16, 64);
  }
  if constexpr (WIDTH == 64) {  // the four rows' results through scalar registers
    const float a0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mn), 0));
    const float a1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mn), 16));
    const float a2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mn), 32));
    const float a3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mn), 48));
    const float b0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mx), 0));
    const float b1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mx), 16));
    const float b2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mx), 32));
    const float b3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mx), 48));
    mn = __builtin_fminf(__builtin_fminf(a0, a1), __builtin_fminf(a2, a3));
    mx = __builtin_fmaxf(__builtin_fmaxf(b0, b1), __builtin_fmaxf(b2, b3));
    nan = __builtin_amdgcn_readlane(nan, 0) | __builtin_amdgcn_readlane(nan, 16) | __builtin_amdgcn_readlane(nan, 32) | __builtin_amdgcn_readlane(nan, 48);
  }
  m.mn = mn; m.mx = mx; m.nan = nan != 0;
}

// all 256 lanes of the block -> result valid in thread 0
__device__ __forceinline__ void block_reduce(MinMax& m, float* lds /* 3 * 4 floats */) {
  wave_reduce<64>(m);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) {
    lds[wave] = m.mn;
    lds[4 + wave] = m.mx;
    lds[8 + wave] = m.nan ? 1.0f : 0.0f;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      m.mn = __builtin_fminf(m.mn, lds[w]);
      m.mx = __builtin_fmaxf(m.mx, lds[4 + w]);
      m.nan |= lds[8 + w] != 0.0f;
    }
  }
}

// Partial record in the workspace: {min, max, nan-flag, pad}
struct Partial { float mn, mx, nan, pad; };

template <typename T>
__device__ __forceinline__ void write_result(T* mn_out, T* mx_out, uint32_t t, MinMax m, int accumulate,
                                             int32_t* flags) {
  float mn = m.nan ? NAN : m.mn;
  float mx = m.nan ? NAN : m.mx;
  int f = 0;
  // flags describe THIS batch (minmax.py:233 tests data_min / data_max, not the running values)
  if (__builtin_isinf(mn) || __builtin_isinf(mx)) f |= FFQ_FLAG_INF;
  if (m.nan) f |= FFQ_FLAG_NAN;
  if (accumulate) {
    const float pmn = to_f32(mn_out[t]), pmx = to_f32(mx_out[t]);
    // torch.min(self.min, data_min) / torch.max(self.max, data_max): NaN propagates    (:236-237)
    mn = (pmn != pmn || mn != mn) ? NAN : __builtin_fminf(pmn, mn);
    mx = (pmx != pmx || mx != mx) ? NAN : __builtin_fmaxf(pmx, mx);
  }
  mn_out[t] = from_f32<T>(mn);
  mx_out[t] = from_f32<T>(mx);
  if (f && flags) atomicOr(flags, f);
}

// ---- A5 (range -> scale / offset) of ONE tile, shared by parameters_for_range_kernel and the one-launch estimator step ------------
struct RangeArgs {
  int range_dt, scale_dt, offset_dt;
  int64_t ntiles;
  int symmetric, allow_one_sided, round_offset;
  float abs_int_min, abs_int_max, num_steps, int_min;
};

__device__ __forceinline__ void range_to_parameters(float lo, float hi, int one_sided, const RangeArgs& a, float& scale, float& offset) {
  if (a.symmetric && one_sided) lo = 0.0f;                               // (range.py:104-105)
  if (a.symmetric && !one_sided) {
    const float neg = __builtin_fabsf(lo) / a.abs_int_min;               // (:108)
    const float pos = __builtin_fabsf(hi) / a.abs_int_max;               // (:109)
    scale = (neg != neg || pos != pos) ? NAN : __builtin_fmaxf(neg, pos);  // torch.max  (:110)
    offset = 0.0f;  // reference returns None; the setter fills the buffer with 0
  } else {
    const float interval = hi - lo;                                      // (:118)
    scale = interval / a.num_steps;                                      // (:119)
    scale = scale != scale ? scale : __builtin_fmaxf(scale, 1.1920928955078125e-07f);  // clamp(eps) (:120)
    const float q = lo / scale;
    offset = q - a.int_min;                                              // (:121)
    if (a.round_offset) offset = rne(offset);                            // dynamic path, _quantizer_impl.py:275
  }
}

// ---- stage 1, one tile: grid-stride over chunks, one Partial per block -------------------------
// what the LAST block of the one-launch form does with the tile's result
struct ScalarFinish {
  void* mn_out; void* mx_out;   // T*
  int accumulate;
  int32_t* flags;
  int32_t* ticket;              // zero on entry, zero on exit
  void* scale_out; void* offset_out;  // nullable: A5 of the (merged) range straight into the quantizer's parameters
  RangeArgs range;
};

// LAST = false: one Partial per block, minmax_finalize_kernel follows. LAST = true (round 4): ONE launch — every block publishes
// its Partial with write-through stores, takes a ticket, and the last block to arrive reduces the Partials, merges into the
// running extrema, sets the status flags and (estimator step) writes scale / offset: the finalize and parameters_for_range
// launches of a per-tensor quantizer are gone (3 launches -> 1; [8, 2048, 4096] bf16: 31.5 -> ~22 us).
template <typename T, int E, int U, bool LAST>
__global__ __launch_bounds__(kBlock) void minmax_scalar_partial_kernel(const T* __restrict__ in,
                                                                       uint32_t nchunks, int64_t numel,
                                                                       Partial* __restrict__ partial, ScalarFinish fin) {
  __shared__ float lds[12];
  typename Accum<T>::type acc;
  acc.init();
  const uint32_t stride = gridDim.x * (uint32_t)(kBlock * U);
  for (uint32_t base = blockIdx.x * (uint32_t)(kBlock * U) + threadIdx.x; base < nchunks; base += stride) {
    Chunk<T, E> x[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t c = base + u * kBlock;
      if (c < nchunks) x[u].load_nt(in + (size_t)c * E);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t c = base + u * kBlock;
      if (c < nchunks) add_chunk<T, E>(acc, x[u]);
    }
  }
  MinMax m = finish_accum<T>(acc);
  if (blockIdx.x == 0 && threadIdx.x == 0) {  // numel % E trailing elements
    for (int64_t i = (int64_t)nchunks * E; i < numel; ++i) m.add(to_f32(in[i]));
  }
  block_reduce(m, lds);
  if constexpr (!LAST) {
    if (threadIdx.x == 0) partial[blockIdx.x] = Partial{m.mn, m.mx, m.nan ? 1.0f : 0.0f, 0.0f};
  } else {
    __shared__ int last_s;
    unsigned long long* cells = reinterpret_cast<unsigned long long*>(partial);
    if (threadIdx.x == 0) {
      // two 8-byte agent-scope stores (write-through: visible to the last block without a release fence), drained, then the ticket
      const unsigned long long lohi = (unsigned long long)__builtin_bit_cast(uint32_t, m.mn) | ((unsigned long long)__builtin_bit_cast(uint32_t, m.mx) << 32);
      __hip_atomic_store(cells + 2 * blockIdx.x, lohi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(cells + 2 * blockIdx.x + 1, (unsigned long long)(m.nan ? 1u : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const int t = __hip_atomic_fetch_add(fin.ticket, 1, FFQ_TICKET_ORDER, __HIP_MEMORY_SCOPE_AGENT);
      last_s = t == (int)gridDim.x - 1;
      if (last_s) __hip_atomic_store(fin.ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // zero again for the next launch
    }
    __syncthreads();
    if (!last_s) return;
    asm volatile("" ::: "memory");  // the partial cells are read after the ticket said every block has published (reader side of the hand-over)
    MinMax r;
    r.init();
    for (uint32_t k = threadIdx.x; k < gridDim.x; k += kBlock) {  // agent-scope loads bypass this CU's L1
      const unsigned long long lohi = __hip_atomic_load(cells + 2 * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned long long nan = __hip_atomic_load(cells + 2 * k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      r.mn = __builtin_fminf(r.mn, __builtin_bit_cast(float, (uint32_t)lohi));
      r.mx = __builtin_fmaxf(r.mx, __builtin_bit_cast(float, (uint32_t)(lohi >> 32)));
      r.nan |= nan != 0;
    }
    __syncthreads();  // lds is reused by the second reduction
    block_reduce(r, lds);
    if (threadIdx.x == 0) {
      T* mn_out = static_cast<T*>(fin.mn_out);
      T* mx_out = static_cast<T*>(fin.mx_out);
      write_result<T>(mn_out, mx_out, 0, r, fin.accumulate, fin.flags);
      if (fin.scale_out) {  // A5 on what the estimator now holds, read back in the data dtype as the eager chain does (.to(float32), range.py:90)
        const float lo = to_f32(mn_out[0]), hi = to_f32(mx_out[0]);
        const int one_sided = fin.range.symmetric && fin.range.allow_one_sided && lo == lo && lo >= 0.0f;
        float scale, offset;
        range_to_parameters(lo, hi, one_sided, fin.range, scale, offset);
        store_any(fin.scale_out, fin.range.scale_dt, 0, (double)scale);
        if (fin.offset_out) store_any(fin.offset_out, fin.range.offset_dt, 0, (double)offset);
      }
    }
  }
}

// ---- contiguous runs: a group of P lanes (P <= 64, power of two) owns one tile ------------------
struct RowsArgs {
  uint32_t ntiles;
  uint32_t chunks_per_run;
  int accumulate;
};

// UF x 16 B non-temporal loads in flight per lane. Measured on [14336, 4096] bf16 (interleaved A/B):
// temporal UF=4 23.3 us, non-temporal UF=4 21.6 us, non-temporal UF=8 21.4 us (5.49 TB/s).
template <typename T, int E, int P, int UF = 8, bool NT = true>
__global__ __launch_bounds__(kBlock) void minmax_rows_kernel(const T* __restrict__ in, T* __restrict__ mn_out,
                                                             T* __restrict__ mx_out, int32_t* flags,
                                                             RowsArgs a) {
  constexpr int TILES_PER_BLOCK = kBlock / P;
  const uint32_t t = blockIdx.x * TILES_PER_BLOCK + threadIdx.x / P;
  const uint32_t lane = threadIdx.x % P;
  typename Accum<T>::type acc;
  acc.init();
  if (t < a.ntiles) {
    const T* row = in + (size_t)t * a.chunks_per_run * E;
    uint32_t c = lane;
    // UF chunks (UF x 16 B) in flight per lane
    for (; c + (UF - 1) * P < a.chunks_per_run; c += UF * P) {
      Chunk<T, E> x[UF];
#pragma unroll
      for (int u = 0; u < UF; ++u) {
        if constexpr (NT) x[u].load_nt(row + (size_t)(c + u * P) * E); else x[u].load(row + (size_t)(c + u * P) * E);
      }
#pragma unroll
      for (int u = 0; u < UF; ++u) add_chunk<T, E>(acc, x[u]);
    }
    for (; c < a.chunks_per_run; c += P) {
      Chunk<T, E> x0;
      x0.load(row + (size_t)c * E);
      add_chunk<T, E>(acc, x0);
    }
  }
  MinMax m = finish_accum<T>(acc);
  wave_reduce<P>(m);
  if (lane == 0 && t < a.ntiles) write_result<T>(mn_out, mx_out, t, m, a.accumulate, flags);
}

// ---- few large tiles: 2-D grid (split, tile) -> Partial[tile][split] ----------------------------
template <typename T, int E>
__global__ __launch_bounds__(kBlock) void minmax_rows_split_kernel(const T* __restrict__ in,
                                                                   uint32_t chunks_per_run,
                                                                   Partial* __restrict__ partial) {
  __shared__ float lds[12];
  const uint32_t t = blockIdx.y, splits = gridDim.x;
  const T* row = in + (size_t)t * chunks_per_run * E;
  typename Accum<T>::type acc;
  acc.init();
  for (uint32_t c = blockIdx.x * kBlock + threadIdx.x; c < chunks_per_run; c += splits * kBlock) {
    Chunk<T, E> x;
    x.load(row + (size_t)c * E);
    add_chunk<T, E>(acc, x);
  }
  MinMax m = finish_accum<T>(acc);
  block_reduce(m, lds);
  if (threadIdx.x == 0) partial[(size_t)t * splits + blockIdx.x] = Partial{m.mn, m.mx, m.nan ? 1.0f : 0.0f, 0.0f};
}

// ---- one channel per column: lane keeps E running pairs, walks its row group --------------------
struct ColsArgs {
  uint32_t col_chunks, rows, row_groups;
  FastDiv col_chunks_div;
};
template <typename T, int E>
__global__ __launch_bounds__(kBlock) void minmax_columns_partial_kernel(const T* __restrict__ in,
                                                                        Partial* __restrict__ partial,
                                                                        ColsArgs a) {
  const uint32_t g = blockIdx.x * (uint32_t)kBlock + threadIdx.x;
  const uint32_t group = fdiv(g, a.col_chunks_div);
  if (group >= a.row_groups) return;
  const uint32_t cc = g - group * a.col_chunks;
  MinMax m[E];
#pragma unroll
  for (int i = 0; i < E; ++i) m[i].init();
  const size_t row_elems = (size_t)a.col_chunks * E;
  for (uint32_t r = group; r < a.rows; r += a.row_groups) {
    Chunk<T, E> x;
    x.load(in + (size_t)r * row_elems + (size_t)cc * E);
#pragma unroll
    for (int i = 0; i < E; ++i) m[i].add(x.get(i));
  }
  // Partial[channel][group]
#pragma unroll
  for (int i = 0; i < E; ++i)
    partial[(size_t)(cc * E + i) * a.row_groups + group] =
        Partial{m[i].mn, m[i].mx, m[i].nan ? 1.0f : 0.0f, 0.0f};
}

// ---- stage 2: Partial[tile][count] -> min/max in data dtype, running merge, flags ---------------
template <typename T>
__global__ __launch_bounds__(kBlock) void minmax_finalize_kernel(const Partial* __restrict__ partial,
                                                                 uint32_t ntiles, uint32_t count,
                                                                 T* __restrict__ mn_out, T* __restrict__ mx_out,
                                                                 int accumulate, int32_t* flags) {
  // one wavefront per tile
  const uint32_t t = blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint32_t lane = threadIdx.x & 63;
  MinMax m;
  m.init();
  if (t < ntiles) {
    for (uint32_t k = lane; k < count; k += 64) {
      const Partial p = partial[(size_t)t * count + k];
      m.mn = __builtin_fminf(m.mn, p.mn);
      m.mx = __builtin_fmaxf(m.mx, p.mx);
      m.nan |= p.nan != 0.0f;
    }
  }
  wave_reduce<64>(m);
  if (lane == 0 && t < ntiles) write_result<T>(mn_out, mx_out, t, m, accumulate, flags);
}

// ---- generic tilings / wide dtypes: ordered-integer atomics on a double carrier -----------------
__device__ __forceinline__ unsigned long long ordered_bits(double v) {
  const unsigned long long u = (unsigned long long)__builtin_bit_cast(long long, v);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double from_ordered_bits(unsigned long long k) {
  const unsigned long long u = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
  return __builtin_bit_cast(double, (long long)u);
}
struct GenericCell { unsigned long long mn, mx; unsigned int nan; unsigned int pad; };

__global__ void minmax_generic_init_kernel(GenericCell* cells, int64_t ntiles) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < ntiles) cells[t] = GenericCell{ordered_bits(INFINITY), ordered_bits(-INFINITY), 0u, 0u};
}
__global__ __launch_bounds__(kBlock) void minmax_generic_scatter_kernel(const void* __restrict__ data, int dt,
                                                                        int64_t numel, GenericTiling g,
                                                                        GenericCell* cells) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < numel; i += stride) {
    const int64_t t = generic_tile_of(g, i);
    const double v = load_any(data, dt, i);
    if (v != v) {
      atomicOr(&cells[t].nan, 1u);
    } else {
      const unsigned long long k = ordered_bits(v);
      atomicMin(&cells[t].mn, k);
      atomicMax(&cells[t].mx, k);
    }
  }
}
__global__ void minmax_generic_finalize_kernel(const GenericCell* cells, int64_t ntiles, void* mn_out,
                                               void* mx_out, int dt, int accumulate, int32_t* flags) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ntiles) return;
  const GenericCell c = cells[t];
  double mn = c.nan ? NAN : from_ordered_bits(c.mn);
  double mx = c.nan ? NAN : from_ordered_bits(c.mx);
  int f = 0;
  if (__builtin_isinf(mn) || __builtin_isinf(mx)) f |= FFQ_FLAG_INF;
  if (c.nan) f |= FFQ_FLAG_NAN;
  if (accumulate) {
    const double pmn = load_any(mn_out, dt, t), pmx = load_any(mx_out, dt, t);
    mn = (pmn != pmn || mn != mn) ? NAN : (pmn < mn ? pmn : mn);
    mx = (pmx != pmx || mx != mx) ? NAN : (pmx > mx ? pmx : mx);
  }
  store_any(mn_out, dt, t, mn);
  store_any(mx_out, dt, t, mx);
  if (f && flags) atomicOr(flags, f);
}

// ---- launch plan ----------------------------------------------------------------------------
enum Plan { PLAN_SCALAR, PLAN_ROWS_WAVE, PLAN_ROWS_SPLIT, PLAN_COLUMNS, PLAN_GENERIC };
struct MinMaxPlan {
  int plan;
  int lanes_per_tile;   // ROWS_WAVE
  uint32_t partials;    // Partial records per tile (SCALAR / ROWS_SPLIT / COLUMNS)
  size_t workspace;
};

constexpr int kE = 8;                  // elements per 16 B chunk for 16-bit data (f32: two loads)
constexpr uint32_t kScalarBlocks = 2048;  // 8 blocks per CU on 256 CUs
#ifndef FFQ_MM_ONE_BLOCKS
#define FFQ_MM_ONE_BLOCKS 512
#endif
constexpr uint32_t kScalarBlocksOneLaunch = FFQ_MM_ONE_BLOCKS;  // the one-launch form (a ticket per block): 2 blocks per CU, 8 chunks in flight per lane

static MinMaxPlan plan_for(const TileInfo& info, int data_dt) {
  MinMaxPlan p;
  p.plan = PLAN_GENERIC;
  p.lanes_per_tile = 0;
  p.partials = 0;
  p.workspace = sizeof(GenericCell) * (size_t)info.ntiles;
  const bool fast_dt = data_dt == FFQ_F32 || data_dt == FFQ_BF16 || data_dt == FFQ_F16;
  if (!fast_dt || info.numel >= ((int64_t)1 << 32) - 4096 || generic_kernels_forced()) return p;
  if (info.layout == LAYOUT_SCALAR) {
    const uint32_t nchunks = (uint32_t)(info.numel / kE);
    uint32_t blocks = (nchunks + kBlock * 4 - 1) / (kBlock * 4);
    if (blocks < 1) blocks = 1;
    if (blocks > kScalarBlocks) blocks = kScalarBlocks;
    p.plan = PLAN_SCALAR;
    p.partials = blocks;
    p.workspace = sizeof(Partial) * blocks;
    return p;
  }
  if (info.layout == LAYOUT_ROWS && info.run % kE == 0) {
    const int64_t chunks = info.run / kE;
    // enough tiles to fill the chip with one (sub-)wave per tile?
    if (info.ntiles >= 2048 || chunks <= 64) {
      int lanes = 64;
      while (lanes > 1 && lanes / 2 >= chunks) lanes /= 2;
      p.plan = PLAN_ROWS_WAVE;
      p.lanes_per_tile = lanes;
      p.workspace = 0;
      return p;
    }
    uint32_t splits = (uint32_t)((chunks + kBlock * 4 - 1) / (kBlock * 4));
    const uint32_t want = (uint32_t)(kScalarBlocks / info.ntiles) + 1;
    if (splits > want) splits = want;
    if (splits < 1) splits = 1;
    p.plan = PLAN_ROWS_SPLIT;
    p.partials = splits;
    p.workspace = sizeof(Partial) * (size_t)splits * (size_t)info.ntiles;
    return p;
  }
  if (info.layout == LAYOUT_CHANNEL && info.inner == 1 && info.channels % kE == 0) {
    const uint32_t rows = (uint32_t)(info.numel / info.channels);
    uint32_t groups = (rows + 15) / 16;
    if (groups > 64) groups = 64;
    if (groups < 1) groups = 1;
    p.plan = PLAN_COLUMNS;
    p.partials = groups;
    p.workspace = sizeof(Partial) * (size_t)groups * (size_t)info.channels;
    return p;
  }
  return p;
}

// what a caller may add to a min/max call: a ticket word (zero on entry, zero on exit) turns the per-tensor plan into ONE launch,
// and (estimator step) the quantizer's parameters are written from the merged range in that same launch
struct StepExtras {
  int32_t* ticket = nullptr;
  void* scale_out = nullptr; void* offset_out = nullptr;
  RangeArgs range = {};
  bool params_done = false;  // set when the launch wrote scale / offset itself
};

template <typename T>
static int run_fast(const MinMaxPlan& p, const TileInfo& info, const void* data, void* mn, void* mx,
                    int accumulate, int32_t* flags, void* workspace, hipStream_t stream, StepExtras* ex) {
  const T* in = static_cast<const T*>(data);
  T* mn_out = static_cast<T*>(mn);
  T* mx_out = static_cast<T*>(mx);
  Partial* partial = static_cast<Partial*>(workspace);
  switch (p.plan) {
    case PLAN_SCALAR: {
      const uint32_t nchunks = (uint32_t)(info.numel / kE);
      ScalarFinish fin = {};
      if (ex && ex->ticket) {  // one launch: the last block to arrive finishes the tile
        fin.mn_out = mn_out; fin.mx_out = mx_out; fin.accumulate = accumulate; fin.flags = flags; fin.ticket = ex->ticket;
        fin.scale_out = ex->scale_out; fin.offset_out = ex->offset_out; fin.range = ex->range;
        // every block takes a ticket on ONE word (~12 ns each, serialised at the L2's atomic unit): 2048 blocks cost 25 us of
        // tickets (measured: 41 us against 31.5 us for the two launches), so this form runs a quarter of the blocks with twice
        // the loads in flight per lane
        uint32_t blocks = (nchunks + kBlock * 8 - 1) / (kBlock * 8);
        if (blocks > kScalarBlocksOneLaunch) blocks = kScalarBlocksOneLaunch;
        if (blocks < 1) blocks = 1;
        minmax_scalar_partial_kernel<T, kE, 8, true><<<blocks, kBlock, 0, stream>>>(in, nchunks, info.numel, partial, fin);
        ex->params_done = ex->scale_out != nullptr;
        break;
      }
      minmax_scalar_partial_kernel<T, kE, 4, false><<<p.partials, kBlock, 0, stream>>>(in, nchunks, info.numel, partial, fin);
      minmax_finalize_kernel<T><<<1, kBlock, 0, stream>>>(partial, 1, p.partials, mn_out, mx_out, accumulate, flags);
      break;
    }
    case PLAN_ROWS_WAVE: {
      RowsArgs a;
      a.ntiles = (uint32_t)info.ntiles;
      a.chunks_per_run = (uint32_t)(info.run / kE);
      a.accumulate = accumulate;
#define FFQ_ROWS(P)                                                                              \
  case P: {                                                                                      \
    const unsigned grid = (unsigned)((info.ntiles + (kBlock / P) - 1) / (kBlock / P));           \
    minmax_rows_kernel<T, kE, P><<<grid, kBlock, 0, stream>>>(in, mn_out, mx_out, flags, a);     \
    break;                                                                                       \
  }
      switch (p.lanes_per_tile) {
        FFQ_ROWS(64) FFQ_ROWS(32) FFQ_ROWS(16) FFQ_ROWS(8) FFQ_ROWS(4) FFQ_ROWS(2) FFQ_ROWS(1)
      }
#undef FFQ_ROWS
      break;
    }
    case PLAN_ROWS_SPLIT: {
      const dim3 grid(p.partials, (unsigned)info.ntiles);
      minmax_rows_split_kernel<T, kE><<<grid, kBlock, 0, stream>>>(in, (uint32_t)(info.run / kE), partial);
      minmax_finalize_kernel<T><<<(unsigned)((info.ntiles + 3) / 4), kBlock, 0, stream>>>(
          partial, (uint32_t)info.ntiles, p.partials, mn_out, mx_out, accumulate, flags);
      break;
    }
    default: {  // PLAN_COLUMNS
      ColsArgs a;
      a.col_chunks = (uint32_t)(info.channels / kE);
      a.rows = (uint32_t)(info.numel / info.channels);
      a.row_groups = p.partials;
      a.col_chunks_div = make_fastdiv(a.col_chunks);
      const uint64_t lanes = (uint64_t)a.row_groups * a.col_chunks;
      minmax_columns_partial_kernel<T, kE><<<(unsigned)((lanes + kBlock - 1) / kBlock), kBlock, 0, stream>>>(in, partial, a);
      minmax_finalize_kernel<T><<<(unsigned)((info.channels + 3) / 4), kBlock, 0, stream>>>(
          partial, (uint32_t)info.channels, p.partials, mn_out, mx_out, accumulate, flags);
      break;
    }
  }
  return check_launch("minmax");
}

static int minmax_impl(const void* data, int data_dt, const ffq_tiling* tiling, void* mn, void* mx,
                       int accumulate, int32_t* flags, void* workspace, size_t workspace_bytes,
                       hipStream_t stream, StepExtras* ex = nullptr) {
  TileInfo info;
  int rc = analyse(tiling, &info);
  if (rc) return rc;
  if (!dt_valid(data_dt)) return fail(FFQ_ERR_ARG, "bad dtype tag");
  if (info.numel == 0) return fail(FFQ_ERR_EMPTY, "min/max of an empty tensor");
  if (!data || !mn || !mx) return fail(FFQ_ERR_ARG, "NULL buffer");
  const MinMaxPlan p = plan_for(info, data_dt);
  if (p.workspace > workspace_bytes || (p.workspace && !workspace))
    return fail(FFQ_ERR_WORKSPACE, "min/max needs %zu workspace bytes, got %zu", p.workspace, workspace_bytes);
  if (p.plan != PLAN_GENERIC && !aligned16(data)) {
    // misaligned view: fall through to the element-wise kernels
  } else if (p.plan != PLAN_GENERIC) {
    switch (data_dt) {
      case FFQ_F32: return run_fast<float>(p, info, data, mn, mx, accumulate, flags, workspace, stream, ex);
      case FFQ_BF16: return run_fast<bf16_t>(p, info, data, mn, mx, accumulate, flags, workspace, stream, ex);
      default: return run_fast<f16_t>(p, info, data, mn, mx, accumulate, flags, workspace, stream, ex);
    }
  }
  const size_t need = sizeof(GenericCell) * (size_t)info.ntiles;
  if (need > workspace_bytes) return fail(FFQ_ERR_WORKSPACE, "min/max needs %zu workspace bytes, got %zu", need, workspace_bytes);
  GenericCell* cells = static_cast<GenericCell*>(workspace);
  const unsigned tb = (unsigned)((info.ntiles + kBlock - 1) / kBlock);
  minmax_generic_init_kernel<<<tb, kBlock, 0, stream>>>(cells, info.ntiles);
  int64_t blocks = (info.numel + kBlock - 1) / kBlock;
  if (blocks > 8192) blocks = 8192;
  minmax_generic_scatter_kernel<<<(unsigned)blocks, kBlock, 0, stream>>>(data, data_dt, info.numel, make_generic(tiling), cells);
  minmax_generic_finalize_kernel<<<tb, kBlock, 0, stream>>>(cells, info.ntiles, mn, mx, data_dt, accumulate, flags);
  return check_launch("minmax_generic");
}

static size_t minmax_workspace(const ffq_tiling* tiling, int data_dt) {
  TileInfo info;
  if (analyse(tiling, &info)) return 0;
  if (info.numel == 0) return 0;
  const MinMaxPlan p = plan_for(info, data_dt);
  // a misaligned pointer may force the generic kernels at call time: size for both
  const size_t generic = sizeof(GenericCell) * (size_t)info.ntiles;
  size_t need = p.workspace > generic ? p.workspace : generic;
  return (need + 255) & ~(size_t)255;
}

// ---- A5 -------------------------------------------------------------------------------------
constexpr int kRangeBlock = 1024;

__global__ __launch_bounds__(kRangeBlock) void parameters_for_range_kernel(const void* __restrict__ min_range,
                                                                           const void* __restrict__ max_range,
                                                                           void* __restrict__ scale_out,
                                                                           void* __restrict__ offset_out,
                                                                           RangeArgs a) {
  __shared__ float lds_min[16];
  __shared__ int lds_nan[16];
  __shared__ int one_sided_s;
  // one_sided = min_range.min() >= 0 and allow_one_sided — global over all tiles        (:100)
  int one_sided = 0;
  if (a.symmetric && a.allow_one_sided) {
    float mn = INFINITY;
    int nan = 0;
    for (int64_t t = threadIdx.x; t < a.ntiles; t += kRangeBlock) {
      const float v = (float)load_any(min_range, a.range_dt, t);  // .to(torch.float32)      (:90)
      mn = __builtin_fminf(mn, v);
      nan |= v != v;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      mn = __builtin_fminf(mn, __shfl_xor(mn, d, 64));
      nan |= __shfl_xor(nan, d, 64);
    }
    if ((threadIdx.x & 63) == 0) { lds_min[threadIdx.x >> 6] = mn; lds_nan[threadIdx.x >> 6] = nan; }
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int w = 1; w < kRangeBlock / 64; ++w) { mn = __builtin_fminf(mn, lds_min[w]); nan |= lds_nan[w]; }
      one_sided_s = (!nan && mn >= 0.0f) ? 1 : 0;  // NaN >= 0 is False
    }
    __syncthreads();
    one_sided = one_sided_s;
  }
  for (int64_t t = threadIdx.x; t < a.ntiles; t += kRangeBlock) {
    const float lo = (float)load_any(min_range, a.range_dt, t);
    const float hi = (float)load_any(max_range, a.range_dt, t);
    float scale, offset;
    range_to_parameters(lo, hi, one_sided, a, scale, offset);
    store_any(scale_out, a.scale_dt, t, (double)scale);
    if (offset_out) store_any(offset_out, a.offset_dt, t, (double)offset);
  }
}

// ---- A5 as a grid (round 5) ---------------------------------------------------------------------------------------------------
// One 1024-lane block walking 458,752 tiles (gate_proj at group 128) through the runtime-typed `double` accessors took 368 us;
// above kRangeGridTiles tiles the work is spread over the chip with typed loads. The only thing that ties the tiles together is
// the GLOBAL one-sided test (range.py:100), needed when symmetric and allow_one_sided: a first launch leaves one minimum per
// block in `partial` (NaN when the block saw one: NaN >= 0 is False), every block of the second launch reduces those <= 1024
// numbers itself (4 KiB from L2) and then writes its tiles — two short launches, no atomics, no tickets, nothing to zero.
// Without the global test it is one launch.
constexpr int64_t kRangeGridTiles = 8192;
constexpr uint32_t kRangePartials = 1024;

template <typename R>
__global__ __launch_bounds__(kBlock) void range_min_partial_kernel(const R* __restrict__ min_range, int64_t ntiles,
                                                                   float* __restrict__ partial) {
  __shared__ float lds[12];
  MinMax m;
  m.init();
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < ntiles; t += stride) m.add(to_f32(min_range[t]));
  block_reduce(m, lds);
  if (threadIdx.x == 0) partial[blockIdx.x] = m.nan ? NAN : m.mn;
}

template <typename R, bool F32OUT>
__global__ __launch_bounds__(kBlock) void parameters_for_range_grid_kernel(const R* __restrict__ min_range,
                                                                           const R* __restrict__ max_range,
                                                                           void* __restrict__ scale_out,
                                                                           void* __restrict__ offset_out,
                                                                           const float* __restrict__ partial,
                                                                           uint32_t npartial, RangeArgs a) {
  __shared__ float lds[12];
  __shared__ int one_sided_s;
  int one_sided = 0;
  if (partial) {  // min_range.min() >= 0 over ALL tiles                                   (:100)
    MinMax m;
    m.init();
    for (uint32_t k = threadIdx.x; k < npartial; k += kBlock) m.add(partial[k]);
    block_reduce(m, lds);
    if (threadIdx.x == 0) one_sided_s = (!m.nan && m.mn >= 0.0f) ? 1 : 0;
    __syncthreads();
    one_sided = one_sided_s;
  }
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < a.ntiles; t += stride) {
    const float lo = to_f32(min_range[t]), hi = to_f32(max_range[t]);  // .to(torch.float32)   (:90)
    float scale, offset;
    range_to_parameters(lo, hi, one_sided, a, scale, offset);
    if constexpr (F32OUT) {
      static_cast<float*>(scale_out)[t] = scale;
      if (offset_out) static_cast<float*>(offset_out)[t] = offset;
    } else {
      store_any(scale_out, a.scale_dt, t, (double)scale);
      if (offset_out) store_any(offset_out, a.offset_dt, t, (double)offset);
    }
  }
}

static RangeArgs make_range_args(int range_dt, int64_t ntiles, double num_bits, int symmetric, int allow_one_sided, int scale_dt, int offset_dt,
                                 int round_offset) {
  RangeArgs a;
  a.range_dt = range_dt; a.scale_dt = scale_dt; a.offset_dt = offset_dt;
  a.ntiles = ntiles;
  a.symmetric = symmetric; a.allow_one_sided = allow_one_sided; a.round_offset = round_offset;
  const double int_min = -pow(2.0, num_bits - 1.0), int_max = -int_min - 1.0;
  a.abs_int_min = (float)fabs(int_min);
  a.abs_int_max = (float)fabs(int_max);
  a.num_steps = (float)(pow(2.0, num_bits) - 1.0);
  a.int_min = (float)int_min;
  return a;
}

template <typename R>
static int parameters_grid(const void* min_range, const void* max_range, void* scale_out, void* offset_out, const RangeArgs& a,
                           float* partial, hipStream_t stream) {
  const R* mn = static_cast<const R*>(min_range);
  const R* mx = static_cast<const R*>(max_range);
  uint32_t npartial = 0;
  if (partial) {
    const int64_t want = (a.ntiles + kBlock * 4 - 1) / (kBlock * 4);
    npartial = (uint32_t)(want > (int64_t)kRangePartials ? kRangePartials : want);
    range_min_partial_kernel<R><<<npartial, kBlock, 0, stream>>>(mn, a.ntiles, partial);
  }
  int64_t blocks = (a.ntiles + kBlock - 1) / kBlock;
  if (blocks > 2048) blocks = 2048;
  const bool f32out = a.scale_dt == FFQ_F32 && (!offset_out || a.offset_dt == FFQ_F32);
  if (f32out) parameters_for_range_grid_kernel<R, true><<<(unsigned)blocks, kBlock, 0, stream>>>(mn, mx, scale_out, offset_out, partial, npartial, a);
  else parameters_for_range_grid_kernel<R, false><<<(unsigned)blocks, kBlock, 0, stream>>>(mn, mx, scale_out, offset_out, partial, npartial, a);
  return check_launch("parameters_for_range_grid_kernel");
}

// bytes of scratch the grid form wants (0: none — few tiles, or nothing global to decide)
static size_t parameters_workspace(int64_t ntiles, int symmetric, int allow_one_sided) {
  if (ntiles <= kRangeGridTiles || !(symmetric && allow_one_sided)) return 0;
  return sizeof(float) * kRangePartials;
}

static int parameters_impl(const void* min_range, const void* max_range, int range_dt, int64_t ntiles,
                           double num_bits, int symmetric, int allow_one_sided, void* scale_out, int scale_dt,
                           void* offset_out, int offset_dt, int round_offset, void* workspace, size_t workspace_bytes,
                           hipStream_t stream) {
  if (!min_range || !max_range || !scale_out || ntiles <= 0) return fail(FFQ_ERR_ARG, "bad argument");
  if (!dt_valid(range_dt) || !dt_valid(scale_dt) || (offset_out && !dt_valid(offset_dt)))
    return fail(FFQ_ERR_ARG, "bad dtype tag");
  const RangeArgs a = make_range_args(range_dt, ntiles, num_bits, symmetric, allow_one_sided, scale_dt, offset_dt, round_offset);
  const bool typed = range_dt == FFQ_F32 || range_dt == FFQ_BF16 || range_dt == FFQ_F16;
  if (typed && ntiles > kRangeGridTiles && !generic_kernels_forced()) {
    const size_t need = parameters_workspace(ntiles, symmetric, allow_one_sided);
    float* partial = need ? static_cast<float*>(workspace) : nullptr;
    if (!need || (workspace && workspace_bytes >= need && (reinterpret_cast<uintptr_t>(workspace) & 3u) == 0)) {
      switch (range_dt) {
        case FFQ_F32: return parameters_grid<float>(min_range, max_range, scale_out, offset_out, a, partial, stream);
        case FFQ_BF16: return parameters_grid<bf16_t>(min_range, max_range, scale_out, offset_out, a, partial, stream);
        default: return parameters_grid<f16_t>(min_range, max_range, scale_out, offset_out, a, partial, stream);
      }
    }
    // no scratch for the global decision: the one-block kernel below decides it in LDS (never fails for lack of scratch)
  }
  parameters_for_range_kernel<<<1, kRangeBlock, 0, stream>>>(min_range, max_range, scale_out, offset_out, a);
  return check_launch("parameters_for_range_kernel");
}

// ---- A3 in ONE launch (round 5): per-token / per-row / per-group dynamic quantization ---------------------------------------------
// quantize_dynamic_by_tile_impl (_quantizer_impl.py:243-285) is min, max, parameters_for_range, round(offset), quantize: five
// passes in the reference, three launches (A4, A5, A1: 2 + 3 = 5 B/elem for bf16 -> int8) in rounds 1-4. When every tile is a
// contiguous run that fits the registers of the lanes that own it and nothing global has to be decided — asymmetric, or
// symmetric without the one-sided fallback (range.py:100 is the only cross-tile dependency of the whole op) — one pass does it:
// a group of P lanes loads its run (U chunks of E elements per lane), reduces min / max (raw 16-bit patterns for bf16 / fp16,
// wave butterflies, one LDS hop when the group is the whole block), every lane evaluates A5 on the two numbers, and the chunks
// still sitting in registers are quantized with A1's arithmetic (ffq_affine.h) and stored: 2 R + 1 W = 3 B/elem.
// Codes, scales and offsets are bit-identical to the composed form (same min / max, same A5 expression, same division).
#ifndef FFQ_DYN_WAVE_ROWS
#define FFQ_DYN_WAVE_ROWS 1  // 129..256-chunk runs (4096-wide rows into int8): 1 = one WAVE per run, no block barrier (round 6: 39.2 -> 37.7 us on
#endif                       // [8, 2048, 4096], [4096, 4096] per channel 15.0 -> 14.0; two runs per wave 42.7, the block form with four in flight 44.8: profiles/r06_a3_variants.txt)
#ifndef FFQ_DYN_WAVE_ROWS_IN_FLIGHT
#define FFQ_DYN_WAVE_ROWS_IN_FLIGHT 1  // runs a wave of the wave-per-run plan works on at once (A/B: 1, 2)
#endif
#ifndef FFQ_DYN_ROWS_IN_FLIGHT
#define FFQ_DYN_ROWS_IN_FLIGHT 2  // tiles a block of the 129..256-chunk plan works on at once (A/B: 1, 2, 4)
#endif
struct DynRowsArgs {
  uint32_t ntiles, chunks_per_run;
  float lo, hi;  // clamp bounds
  RangeArgs range;
  // RUNNING (ffq_running_minmax_quantize: a RunningMinMax estimator step and the quantizer's own forward in one pass): the
  // estimator's per-tile running min / max in the data dtype, merged in place, and its status word (flags of THIS batch)
  void* run_min; void* run_max;
  int32_t* flags;
};

// R = tiles a group of lanes works on at once (R > 1: whole-block groups only): all their loads are issued up front, the R reductions
// meet in ONE barrier, and tile r + 1's memory latency hides behind tile r's arithmetic — a block that lives for one 8 KiB row spends
// most of its life waiting ([8, 2048, 4096] bf16 per token: 39.7 us with R = 1).
//
// MODE (symmetric AND allow_one_sided: range.py:100 asks whether the smallest minimum of ALL tiles is >= 0):
//   DYN_PLAIN  nothing global to decide.
//   DYN_GUESS  first launch. A tile whose own minimum is negative (or NaN) has answered the global question — "not one-sided" — by
//              itself and is finished here, two-sided. A tile with a minimum >= 0 cannot know: it leaves its maximum in scale_out, its
//              minimum with the SIGN BIT SET in offset_out (a finished two-sided tile holds +0.0 there) and is counted in ticket[0].
//   DYN_SETTLE second launch, a few blocks striding over the first launch's blocks. ticket[0] == 0: nothing was left open, return
//              (activations with both signs in every row: 60.7 -> ~44 us for [8, 2048, 4096], 3 B/elem + an empty launch).
//              Otherwise the verdict is known — one-sided iff EVERY tile was left open — and the open tiles are read again and
//              finished with it (all-non-negative data: 2 + 3 = 5 B/elem, what the composed form moves); the last block to leave
//              zeroes the two ticket words.
enum { DYN_PLAIN = 0, DYN_GUESS = 1, DYN_SETTLE = 2 };

// RUNNING: the tile's fresh extrema are merged into the estimator's running pair (read before the block's barrier, written by the
// group's first lane after it), the status flags describe this batch, and A5 runs on the MERGED range read back in the data dtype —
// range_setting/minmax.py:215-239 followed by the range setter and the quantizer's forward (common.py:218-238) in 2 R + 1 W bytes
// per element instead of 2 + 2 R + 1 W. With the global one-sided question the open tiles need no stash: their merged range is
// in the running buffers; offset_out holds -0.0 as the marker.
template <typename TIn, typename TOut, int E, int P, int U, int R = 1, int MODE = DYN_PLAIN, bool RUNNING = false>
__global__ __launch_bounds__(kBlock) void quantize_dynamic_rows_kernel(const TIn* __restrict__ in, TOut* __restrict__ out,
                                                                       float* __restrict__ scale_out, float* __restrict__ offset_out,
                                                                       DynRowsArgs a, int32_t* __restrict__ ticket, uint32_t logical_blocks) {
  static_assert(P <= 64 || P == kBlock, "a group is part of a wave or the whole block");
  static_assert(R == 1 || P == kBlock || P == 64, "several tiles in flight: whole-block or whole-wave groups");
  constexpr int TILES_PER_BLOCK = kBlock / P;
  const uint32_t lane = threadIdx.x % P;
  [[maybe_unused]] int verdict = 0;     // DYN_SETTLE: one-sided?
  [[maybe_unused]] int open_here = 0;   // DYN_GUESS: tiles this thread left open (counted once per tile, by the group's first lane)
  if constexpr (MODE == DYN_SETTLE) {
    const int open = __hip_atomic_load(ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (open == 0) return;
    verdict = (uint32_t)open == a.ntiles;
  }
  for (uint32_t blk = blockIdx.x; blk < logical_blocks; blk += gridDim.x) {  // (one trip unless DYN_SETTLE)
    const uint32_t t0 = (blk * TILES_PER_BLOCK + threadIdx.x / P) * R;
    Chunk<TIn, E> x[R][U];
    MinMax m[R];
    [[maybe_unused]] bool mine[R];  // DYN_SETTLE: the tile was left open
#pragma unroll
    for (int r = 0; r < R; ++r) {
      mine[r] = true;
      if constexpr (MODE == DYN_SETTLE) {
        mine[r] = t0 + r < a.ntiles && (__builtin_bit_cast(uint32_t, offset_out[t0 + r]) >> 31) != 0;
        if (mine[r]) {
          if constexpr (RUNNING) {
            m[r].mn = to_f32(static_cast<const TIn*>(a.run_min)[t0 + r]);
            m[r].mx = to_f32(static_cast<const TIn*>(a.run_max)[t0 + r]);
          } else {
            m[r].mn = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, offset_out[t0 + r]) & 0x7FFFFFFFu);
            m[r].mx = scale_out[t0 + r];
          }
          m[r].nan = false;  // (a NaN minimum is "not >= 0": such a tile was finished by the first launch)
        }
      }
      const size_t row = (size_t)(t0 + r) * a.chunks_per_run;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t c = lane + u * P;
        if (mine[r] && t0 + r < a.ntiles && c < a.chunks_per_run) x[r][u].load_nt(in + (row + c) * E);
      }
    }
    // (a whole-block group: every wave has read the tile's marker before the first lane replaces it with the parameters)
    if constexpr (MODE == DYN_SETTLE && P > 64) __syncthreads();
    [[maybe_unused]] float prev_mn[R], prev_mx[R];
    if constexpr (RUNNING && MODE != DYN_SETTLE) {  // every lane of the group, ahead of the barrier its first lane writes behind
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const uint32_t t = t0 + r < a.ntiles ? t0 + r : a.ntiles - 1;
        prev_mn[r] = to_f32(static_cast<const TIn*>(a.run_min)[t]);
        prev_mx[r] = to_f32(static_cast<const TIn*>(a.run_max)[t]);
      }
    }
    if constexpr (MODE != DYN_SETTLE) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        typename Accum<TIn>::type acc;
        acc.init();
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const uint32_t c = lane + u * P;
          if (t0 + r < a.ntiles && c < a.chunks_per_run) add_chunk<TIn, E>(acc, x[r][u]);
        }
        m[r] = finish_accum<TIn>(acc);
        wave_allreduce<(P <= 64 ? P : 64)>(m[r]);
      }
      if constexpr (P > 64) {
        __shared__ float lds[R][12];
        const int wave = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) {
#pragma unroll
          for (int r = 0; r < R; ++r) {
            lds[r][wave] = m[r].mn;
            lds[r][4 + wave] = m[r].mx;
            lds[r][8 + wave] = m[r].nan ? 1.0f : 0.0f;
          }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            m[r].mn = __builtin_fminf(m[r].mn, lds[r][w]);
            m[r].mx = __builtin_fmaxf(m[r].mx, lds[r][4 + w]);
            m[r].nan |= lds[r][8 + w] != 0.0f;
          }
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const uint32_t t = t0 + r;
      if (t >= a.ntiles || !mine[r]) continue;
      const size_t row = (size_t)t * a.chunks_per_run;
      // torch.min / torch.max propagate NaN; the extrema are elements of the data, so .to(float32) is exact   (:257-258, range.py:90)
      float mn = m[r].nan ? NAN : m[r].mn, mx = m[r].nan ? NAN : m[r].mx;
      if constexpr (RUNNING && MODE != DYN_SETTLE) {  // write_result() of the two-step form, on this tile
        int f = 0;
        if (__builtin_isinf(mn) || __builtin_isinf(mx)) f |= FFQ_FLAG_INF;
        if (m[r].nan) f |= FFQ_FLAG_NAN;
        mn = (prev_mn[r] != prev_mn[r] || mn != mn) ? NAN : __builtin_fminf(prev_mn[r], mn);
        mx = (prev_mx[r] != prev_mx[r] || mx != mx) ? NAN : __builtin_fmaxf(prev_mx[r], mx);
        if (lane == 0) {
          static_cast<TIn*>(a.run_min)[t] = from_f32<TIn>(mn);
          static_cast<TIn*>(a.run_max)[t] = from_f32<TIn>(mx);
          if (f && a.flags) atomicOr(a.flags, f);
        }
      }
      if constexpr (MODE == DYN_GUESS) {
        if (mn >= 0.0f) {  // the answer depends on the other tiles: leave the range behind and count the tile
          if (lane == 0) {
            if constexpr (RUNNING) {
              offset_out[t] = -0.0f;  // (the merged range is in the running buffers)
            } else {
              scale_out[t] = mx;
              offset_out[t] = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, mn) | 0x80000000u);
            }
            ++open_here;
          }
          continue;
        }
      }
      float scale, offset;
      range_to_parameters(mn, mx, MODE == DYN_SETTLE ? verdict : 0, a.range, scale, offset);  // offset None -> zeros; offset = round(offset)   (:266-275)
      if (lane == 0) {
        scale_out[t] = scale;
        offset_out[t] = offset;
      }
      if constexpr (RUNNING) offset = rne(offset);  // the parameter keeps its fraction, A1 rounds it (_quantizer_impl.py:140-141); A3's is rounded already
      const Divider<1> d(scale);
      bool fast = false;
      // the run's own extrema bound every |x|: one test per tile decides for the packed arithmetic of ffq_affine.h
      if constexpr (sizeof(TOut) == 1) fast = fast_chunk_ok(scale, d.r, __builtin_fmaxf(__builtin_fabsf(mn), __builtin_fabsf(mx)));
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t c = lane + u * P;
        if (c >= a.chunks_per_run) continue;
        float xf[E];
#pragma unroll
        for (int i = 0; i < E; ++i) xf[i] = x[r][u].get(i);
        Chunk<TOut, E> y;
        if constexpr (sizeof(TOut) == 1) {
          if (fast) {
            quantize_chunk_bytes_fast<E, false>(xf, scale, d.r, offset, a.lo, a.hi, y.w);
            y.store(out + (row + c) * E);
            continue;
          }
        }
        float q[E];
        quantize_chunk_with<1, E>(d, xf, offset, q);  // round(row / scale - offset), clamp, cast               (:277-284)
        finalize_chunk<TOut, E>(q, a.lo, a.hi, y);
        y.store(out + (row + c) * E);
      }
    }
    if constexpr (MODE != DYN_SETTLE) break;
  }
  if constexpr (MODE == DYN_GUESS) {  // one global atomic per block that left something open
    __shared__ int open_s;
    if (threadIdx.x == 0) open_s = 0;
    __syncthreads();
    if (open_here) atomicAdd(&open_s, open_here);
    __syncthreads();
    if (threadIdx.x == 0 && open_s) __hip_atomic_fetch_add(ticket, open_s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if constexpr (MODE == DYN_SETTLE) {  // (reached only when something was open) the last block to leave zeroes the words
    __syncthreads();
    if (threadIdx.x == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const int left = __hip_atomic_fetch_add(ticket + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (left == (int)gridDim.x - 1) {
        __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(ticket + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

template <typename TIn, typename TOut>
static bool launch_dynamic_rows(const void* data, void* out, float* scale_out, float* offset_out, const TileInfo& info,
                                const DynRowsArgs& base, int32_t* ticket, hipStream_t stream) {
  constexpr int E = sizeof(TOut) == 1 ? 16 : 8;
  if (info.run % E != 0) return false;
  const int64_t chunks = info.run / E;
  if (chunks > 4 * kBlock) return false;  // the run no longer fits the block's registers
  DynRowsArgs a = base;
  a.chunks_per_run = (uint32_t)chunks;
  const TIn* in = static_cast<const TIn*>(data);
  TOut* o = static_cast<TOut*>(out);
  const bool global_question = a.range.symmetric && a.range.allow_one_sided;  // (the caller checked the ticket words)
#define FFQ_DYN_M(P, U, R, RUN)                                                                                   \
  do {                                                                                                            \
    const int64_t per_block = (int64_t)(kBlock / P) * R;                                                          \
    const unsigned grid = (unsigned)((info.ntiles + per_block - 1) / per_block);                                  \
    if (!global_question) {                                                                                       \
      quantize_dynamic_rows_kernel<TIn, TOut, E, P, U, R, DYN_PLAIN, RUN><<<grid, kBlock, 0, stream>>>(in, o, scale_out, offset_out, a, nullptr, grid); \
    } else {                                                                                                      \
      quantize_dynamic_rows_kernel<TIn, TOut, E, P, U, R, DYN_GUESS, RUN><<<grid, kBlock, 0, stream>>>(in, o, scale_out, offset_out, a, ticket, grid);  \
      const unsigned settle = grid < 2048u ? grid : 2048u;                                                        \
      quantize_dynamic_rows_kernel<TIn, TOut, E, P, U, R, DYN_SETTLE, RUN><<<settle, kBlock, 0, stream>>>(in, o, scale_out, offset_out, a, ticket, grid); \
    }                                                                                                             \
  } while (0)
#define FFQ_DYN_R(P, U, R) do { if (a.run_min) FFQ_DYN_M(P, U, R, true); else FFQ_DYN_M(P, U, R, false); } while (0)
#define FFQ_DYN(P, U) FFQ_DYN_R(P, U, 1)
  if (chunks <= 1) FFQ_DYN(1, 1);
  else if (chunks <= 2) FFQ_DYN(2, 1);
  else if (chunks <= 4) FFQ_DYN(4, 1);
  else if (chunks <= 8) FFQ_DYN(8, 1);
  else if (chunks <= 16) FFQ_DYN(16, 1);
  else if (chunks <= 32) FFQ_DYN(32, 1);
  else if (chunks <= 64) FFQ_DYN(64, 1);
  else if (chunks <= 128) FFQ_DYN(64, 2);
#if FFQ_DYN_WAVE_ROWS  // A/B: one wave per run up to 256 chunks (no block barrier, four loads in flight per lane)
  else if (chunks <= 256) FFQ_DYN_R(64, 4, FFQ_DYN_WAVE_ROWS_IN_FLIGHT);
#else
  else if (chunks <= 256) FFQ_DYN_R(256, 1, FFQ_DYN_ROWS_IN_FLIGHT);
#endif
  else if (chunks <= 512) FFQ_DYN(256, 2);
  else FFQ_DYN(256, 4);
#undef FFQ_DYN
#undef FFQ_DYN_R
#undef FFQ_DYN_M
  return true;
}

template <typename TIn>
static bool launch_dynamic_rows_out(int out_dt, const void* data, void* out, float* scale_out, float* offset_out,
                                    const TileInfo& info, const DynRowsArgs& a, int32_t* ticket, hipStream_t stream) {
  switch (out_dt) {
    case FFQ_I8: return launch_dynamic_rows<TIn, int8_t>(data, out, scale_out, offset_out, info, a, ticket, stream);
    case FFQ_F32: return launch_dynamic_rows<TIn, float>(data, out, scale_out, offset_out, info, a, ticket, stream);
    case FFQ_BF16: return launch_dynamic_rows<TIn, bf16_t>(data, out, scale_out, offset_out, info, a, ticket, stream);
    default: return false;
  }
}

// true: the one-launch kernel was enqueued (*rc holds the launch status); false: the caller composes A4 -> A5 -> A1
static bool dynamic_one_launch(const void* data, int data_dt, const TileInfo& info, double num_bits, int symmetric,
                               int allow_one_sided, void* out, int out_dt, float* scale_out, float* offset_out,
                               int32_t* ticket, hipStream_t stream, int* rc, void* run_min = nullptr, void* run_max = nullptr,
                               int32_t* flags = nullptr) {
  if (info.layout != LAYOUT_ROWS || info.ntiles >= ((int64_t)1 << 31) || info.numel >= ((int64_t)1 << 36)) return false;
  // the one decision that is global over the tiles (range.py:100) takes two ticket words (DYN_GUESS / DYN_SETTLE); a single tile
  // answers it in the reduction's last block (below)
  if (symmetric && allow_one_sided && (!ticket || info.ntiles == 1)) return false;
  if (num_bits != floor(num_bits) || num_bits < 1 || num_bits > 32 || generic_kernels_forced()) return false;
  if (!aligned16(data) || !aligned16(out)) return false;
  DynRowsArgs a;
  a.ntiles = (uint32_t)info.ntiles;
  a.chunks_per_run = 0;
  const double lo = -pow(2.0, num_bits - 1.0);
  a.lo = (float)lo; a.hi = (float)(-lo - 1.0);
  // A3 rounds the offset it returns (_quantizer_impl.py:275); a static quantizer's parameter keeps its fraction (range.py:121)
  a.range = make_range_args(data_dt, info.ntiles, num_bits, symmetric, allow_one_sided, FFQ_F32, FFQ_F32, run_min ? 0 : 1);
  a.run_min = run_min; a.run_max = run_max; a.flags = flags;
  bool done = false;
  switch (data_dt) {
    case FFQ_BF16: done = launch_dynamic_rows_out<bf16_t>(out_dt, data, out, scale_out, offset_out, info, a, ticket, stream); break;
    case FFQ_F16: done = launch_dynamic_rows_out<f16_t>(out_dt, data, out, scale_out, offset_out, info, a, ticket, stream); break;
    case FFQ_F32: done = launch_dynamic_rows_out<float>(out_dt, data, out, scale_out, offset_out, info, a, ticket, stream); break;
    default: break;
  }
  if (done) *rc = check_launch("quantize_dynamic_rows_kernel");
  return done;
}

}  // namespace ffq

using namespace ffq;

extern "C" {

size_t ffq_minmax_workspace_bytes(const ffq_tiling* tiling, int data_dt) { return minmax_workspace(tiling, data_dt); }

int ffq_minmax_by_tile(const void* data, int data_dt, const ffq_tiling* tiling, void* min_inout,
                       void* max_inout, int accumulate, int32_t* status_flags, void* workspace,
                       size_t workspace_bytes, int32_t* ticket, void* stream) {
  StepExtras ex;
  ex.ticket = ticket;
  return minmax_impl(data, data_dt, tiling, min_inout, max_inout, accumulate, status_flags, workspace,
                     workspace_bytes, static_cast<hipStream_t>(stream), &ex);
}

// One RunningMinMax estimator step (range_setting/minmax.py:215-239 + the range setter nn/linear_quantizer.py:350-357 +
// affine/range.py:54-122): running min / max merged in place (A4), then scale / offset of the merged range (A5) written into the
// quantizer's own parameter tensors. A per-tensor quantizer with a ticket word takes ONE launch for all of it.
int ffq_running_minmax_step(const void* data, int data_dt, const ffq_tiling* tiling, void* min_inout, void* max_inout,
                            int32_t* status_flags, double num_bits, int symmetric, int allow_one_sided, void* scale_out,
                            int scale_dt, void* offset_out, int offset_dt, void* workspace, size_t workspace_bytes,
                            int32_t* ticket, void* stream) {
  if (!scale_out || !dt_valid(scale_dt) || (offset_out && !dt_valid(offset_dt))) return fail(FFQ_ERR_ARG, "bad parameter output");
  TileInfo info;
  int rc = analyse(tiling, &info);
  if (rc) return rc;
  StepExtras ex;
  ex.ticket = ticket;
  ex.scale_out = scale_out; ex.offset_out = offset_out;
  ex.range = make_range_args(data_dt, info.ntiles, num_bits, symmetric, allow_one_sided, scale_dt, offset_dt, 0);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if ((rc = minmax_impl(data, data_dt, tiling, min_inout, max_inout, 1, status_flags, workspace, workspace_bytes, s, &ex))) return rc;
  if (ex.params_done) return FFQ_OK;
  // the reduction's scratch is free again (stream order): the grid form of A5 keeps its per-block minima there
  return parameters_impl(min_inout, max_inout, data_dt, info.ntiles, num_bits, symmetric, allow_one_sided, scale_out, scale_dt, offset_out,
                         offset_dt, 0, workspace, workspace_bytes, s);
}

// The estimator step above AND the quantizer's forward on the same data (range_setting/common.py:218-238: estimate_step, then
// the quantizer's own forward) in one pass over it — see include/ffq.h
int ffq_running_minmax_quantize(const void* data, int data_dt, const ffq_tiling* tiling, void* min_inout, void* max_inout,
                                int32_t* status_flags, double num_bits, int symmetric, int allow_one_sided, float* scale_out,
                                float* offset_out, void* out, int out_dt, int32_t* ticket, void* stream) {
  if (!data || !min_inout || !max_inout || !scale_out || !offset_out || !out) return fail(FFQ_ERR_ARG, "NULL buffer");
  if (!dt_valid(data_dt) || !dt_valid(out_dt)) return fail(FFQ_ERR_ARG, "bad dtype tag");
  TileInfo info;
  int rc = analyse(tiling, &info);
  if (rc) return rc;
  if (info.numel == 0) return fail(FFQ_ERR_DTYPE, "running min/max + quantize: empty tensor (take the two steps)");
  if (!ffq_can_support_bitwidth(out_dt, num_bits))
    return fail(FFQ_ERR_PRECISION, "Provided dtype (%d) is not enough to store %g bits quantized values.", out_dt, num_bits);
  if (info.ntiles == 1 || !dynamic_one_launch(data, data_dt, info, num_bits, symmetric, allow_one_sided, out, out_dt, scale_out, offset_out, ticket,
                                              static_cast<hipStream_t>(stream), &rc, min_inout, max_inout, status_flags))
    return fail(FFQ_ERR_DTYPE, "running min/max + quantize: tiling outside the one-pass kernel (take ffq_running_minmax_step, then ffq_quantize_by_tile)");
  return rc;
}

size_t ffq_parameters_for_range_workspace_bytes(int64_t ntiles, int symmetric, int allow_one_sided) {
  return parameters_workspace(ntiles, symmetric, allow_one_sided);
}

int ffq_parameters_for_range(const void* min_range, const void* max_range, int range_dt, int64_t ntiles,
                             double num_bits, int symmetric, int allow_one_sided, void* scale_out,
                             int scale_dt, void* offset_out, int offset_dt, void* workspace, size_t workspace_bytes,
                             void* stream) {
  return parameters_impl(min_range, max_range, range_dt, ntiles, num_bits, symmetric, allow_one_sided,
                         scale_out, scale_dt, offset_out, offset_dt, 0, workspace, workspace_bytes,
                         static_cast<hipStream_t>(stream));
}

// workspace layout: [ minmax scratch | min (ntiles, data dtype) | max (ntiles, data dtype) ]
size_t ffq_quantize_dynamic_workspace_bytes(const ffq_tiling* tiling, int data_dt) {
  TileInfo info;
  if (analyse(tiling, &info) || info.numel == 0) return 0;
  size_t mm = minmax_workspace(tiling, data_dt);
  const size_t a5 = (parameters_workspace(info.ntiles, 1, 1) + 255) & ~(size_t)255;
  if (mm < a5) mm = a5;
  const size_t ranges = (((size_t)info.ntiles * dt_size(data_dt)) + 255) & ~(size_t)255;
  return mm + 2 * ranges;
}

int ffq_quantize_dynamic_by_tile(const void* data, int data_dt, const ffq_tiling* tiling, double num_bits,
                                 int symmetric, int allow_one_sided, void* out, int out_dt,
                                 float* scale_out, float* offset_out, void* workspace,
                                 size_t workspace_bytes, int32_t* ticket, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  TileInfo info;
  int rc = analyse(tiling, &info);
  if (rc) return rc;
  // torch.min over an empty row raises IndexError -> QuantizationError                 (:259-264)
  if (info.numel == 0) return fail(FFQ_ERR_EMPTY, "Cannot dynamically quantize an empty tensor");
  if (!data || !out || !scale_out || !offset_out) return fail(FFQ_ERR_ARG, "NULL buffer");
  if (!dt_valid(data_dt) || !dt_valid(out_dt)) return fail(FFQ_ERR_ARG, "bad dtype tag");
  if (!ffq_can_support_bitwidth(out_dt, num_bits))
    return fail(FFQ_ERR_PRECISION, "Provided dtype (%d) is not enough to store %g bits quantized values.",
                out_dt, num_bits);
  // contiguous-run tiles that fit a block's registers and no global one-sided decision: ONE launch, 3 B/elem
  if (dynamic_one_launch(data, data_dt, info, num_bits, symmetric, allow_one_sided, out, out_dt, scale_out, offset_out, ticket, s, &rc)) return rc;
  const size_t need = ffq_quantize_dynamic_workspace_bytes(tiling, data_dt);
  if (need > workspace_bytes || !workspace)
    return fail(FFQ_ERR_WORKSPACE, "dynamic quantize needs %zu workspace bytes, got %zu", need, workspace_bytes);
  const size_t ranges = (((size_t)info.ntiles * dt_size(data_dt)) + 255) & ~(size_t)255;
  const size_t mm = need - 2 * ranges;
  char* base = static_cast<char*>(workspace);
  void* mn = base + mm;
  void* mx = base + mm + ranges;
  // one tile and a ticket word: the reduction's last block also evaluates A5 (round(offset) included)   (:266-275)
  StepExtras ex;
  ex.ticket = ticket;
  ex.scale_out = scale_out; ex.offset_out = offset_out;
  ex.range = make_range_args(data_dt, info.ntiles, num_bits, symmetric, allow_one_sided, FFQ_F32, FFQ_F32, 1);
  if ((rc = minmax_impl(data, data_dt, tiling, mn, mx, 0, nullptr, base, mm, s, &ex))) return rc;
  // parameters_for_range; offset None -> zeros; offset = round(offset)                  (:266-275)
  if (!ex.params_done &&
      (rc = parameters_impl(mn, mx, data_dt, info.ntiles, num_bits, symmetric, allow_one_sided, scale_out,
                            FFQ_F32, offset_out, FFQ_F32, 1, base, mm, s)))
    return rc;
  // round(row / scale - offset), clamp, cast                                            (:277-284)
  return quantize_impl(data, data_dt, scale_out, FFQ_F32, info.ntiles, offset_out, FFQ_F32, info.ntiles,
                       tiling, num_bits, out, out_dt, s);
}

}  // extern "C"

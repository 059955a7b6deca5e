// ffq_quantize.hip — A1: fastforward::quantize_by_tile on gfx950.
//
// Reference: quantize_by_tile_impl, src/fastforward/quantization/_quantizer_impl.py:144-169
//   q = cast(clamp(round(x / s_t - round(o_t)), -2^(b-1), 2^(b-1)-1))
// The reference runs this as five ATen passes over a tiles_to_rows view (a full copy for strided
// channels). Here it is ONE pass: 16 B per lane in, packed codes out, the tile -> parameter map
// folded into index arithmetic. HBM-bound: 3 B/elem for bf16 -> int8, 4 B/elem for bf16 -> bf16.
//
// Kernel families
//   quantize_stream_kernel   fp32 stages, fp32 parameters, layouts SCALAR / ROWS / CHANNEL(block)
//   quantize_columns_kernel  fp32 stages, fp32 parameters, one channel per COLUMN (PerChannel(-1)):
//                            each lane keeps the parameters of its 8 columns in registers and
//                            walks down the rows, so nothing is re-fetched and nothing is copied
//   quantize_generic_kernel  any dtype mix / any N-d tiling, one element per lane, every
//                            intermediate rounding of the eager chain reproduced
#include "ffq_affine.h"
#include "ffq_common.h"
#include "ffq_vec.h"

#include <math.h>
#include <stdlib.h>

namespace ffq {

struct StreamArgs {
  float lo, hi;
  uint32_t nchunks;
  uint32_t scale_stride;   // 0 when scale is a single broadcast value, else 1
  uint32_t offset_stride;  // same for offset
  FastDiv chunks_per_run;  // ROWS: run / E             CHANNEL: inner / E
  FastDiv channels;        // CHANNEL: number of channels
};

template <int LAYOUT>
__device__ __forceinline__ uint32_t tile_of_chunk(uint32_t chunk, const StreamArgs& a) {
  if constexpr (LAYOUT == LAYOUT_SCALAR) {
    return 0;
  } else if constexpr (LAYOUT == LAYOUT_ROWS) {
    return fdiv(chunk, a.chunks_per_run);
  } else {
    const uint32_t outer = fdiv(chunk, a.chunks_per_run);  // flat / inner
    return outer - fdiv(outer, a.channels) * a.channels.div;
  }
}

// One block = kBlock lanes x U chunks of E elements; chunk c of the block is read by lane
// (c % kBlock) so that every load instruction of a wave covers one contiguous 64 * 16 B span.
template <typename TIn, typename TOut, int LAYOUT, int E, int U, bool HAS_OFFSET, int DIVMODE, int NT = 0>
__global__ __launch_bounds__(kBlock) void quantize_stream_kernel(const TIn* __restrict__ in,
                                                                 TOut* __restrict__ out,
                                                                 const float* __restrict__ scale,
                                                                 const float* __restrict__ offset,
                                                                 StreamArgs a) {
  const uint32_t first = blockIdx.x * (uint32_t)(kBlock * U) + threadIdx.x;
  Chunk<TIn, E> x[U];
  float s[U], o[U];
  // all HBM loads first, then the (L2-resident) parameter loads: nothing below waits in series
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const uint32_t c = first + u * kBlock;
    if (c < a.nchunks) {
      if constexpr (NT & 1) x[u].load_nt(in + (size_t)c * E); else x[u].load(in + (size_t)c * E);
    }
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const uint32_t c = first + u * kBlock;
    s[u] = 1.0f;
    o[u] = 0.0f;
    if (c < a.nchunks) {
      const uint32_t t = tile_of_chunk<LAYOUT>(c, a);
      s[u] = scale[t * a.scale_stride];
      if constexpr (HAS_OFFSET) o[u] = offset[t * a.offset_stride];
    }
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const uint32_t c = first + u * kBlock;
    if (c >= a.nchunks) continue;
    float xf[E];
#pragma unroll
    for (int i = 0; i < E; ++i) xf[i] = x[u].get(i);
    Chunk<TOut, E> y;
    if constexpr (TypeTag<TOut>::value == FFQ_I8 && DIVMODE == 1 && E % 4 == 0) {
      quantize_chunk_to_bytes<E>(xf, s[u], HAS_OFFSET ? rne(o[u]) : 0.0f, a.lo, a.hi, y);  // packed arithmetic, self-checked
    } else {
      float r[E];
      quantize_chunk<DIVMODE, E>(xf, s[u], HAS_OFFSET ? rne(o[u]) : 0.0f, r);
      finalize_chunk<TOut, E>(r, a.lo, a.hi, y);
    }
    if constexpr (NT & 2) y.store_nt(out + (size_t)c * E); else y.store(out + (size_t)c * E);
  }
}


struct ColumnArgs {
  float lo, hi;
  uint32_t col_chunks;   // channels / E
  uint32_t rows;
  uint32_t row_groups;   // rows are visited as row = group + k * row_groups
  uint32_t scale_stride;
  uint32_t offset_stride;
  FastDiv col_chunks_div;
};

// data viewed as [rows, channels]; lane g owns column chunk g % col_chunks for every row of its
// row group. The 2 x E parameters stay in VGPRs across the whole walk.
template <typename TIn, typename TOut, int E, bool HAS_OFFSET>
__global__ __launch_bounds__(kBlock) void quantize_columns_kernel(const TIn* __restrict__ in,
                                                                  TOut* __restrict__ out,
                                                                  const float* __restrict__ scale,
                                                                  const float* __restrict__ offset,
                                                                  ColumnArgs a) {
  const uint32_t g = blockIdx.x * (uint32_t)kBlock + threadIdx.x;
  const uint32_t group = fdiv(g, a.col_chunks_div);
  if (group >= a.row_groups) return;
  const uint32_t cc = g - group * a.col_chunks;
  float s[E], o[E];
#pragma unroll
  for (int i = 0; i < E; ++i) {
    s[i] = scale[(cc * E + i) * a.scale_stride];
    o[i] = HAS_OFFSET ? rne(offset[(cc * E + i) * a.offset_stride]) : 0.0f;
  }
  const size_t row_elems = (size_t)a.col_chunks * E;
  for (uint32_t r = group; r < a.rows; r += a.row_groups) {
    const size_t at = (size_t)r * row_elems + (size_t)cc * E;
    Chunk<TIn, E> x;
    x.load(in + at);
    float q[E];
#pragma unroll
    for (int i = 0; i < E; ++i) q[i] = rne(x.get(i) / s[i] - o[i]);
    Chunk<TOut, E> y;
    finalize_chunk<TOut, E>(q, a.lo, a.hi, y);
    y.store(out + at);
  }
}

struct GenericArgs {
  int data_dt, scale_dt, offset_dt, out_dt;
  int div_dt, sub_dt;
  int has_offset;
  int64_t start, count;
  int64_t scale_numel, offset_numel;
  double lo, hi;
  GenericTiling g;
};

// Follows oracle-independent reading of the eager chain: each ATen op evaluates in float opmath
// (double for f64 tensors) and rounds its result into the promoted dtype.
__global__ __launch_bounds__(kBlock) void quantize_generic_kernel(const void* __restrict__ data,
                                                                  const void* __restrict__ scale,
                                                                  const void* __restrict__ offset,
                                                                  void* __restrict__ out, GenericArgs a) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x; k < a.count; k += stride) {
    const int64_t i = a.start + k;
    const int64_t t = generic_tile_of(a.g, i);
    const double xs = load_any(data, a.data_dt, i);
    const double ss = load_any(scale, a.scale_dt, a.scale_numel == 1 ? 0 : t);
    double os = 0.0;
    if (a.has_offset) {
      os = load_any(offset, a.offset_dt, a.offset_numel == 1 ? 0 : t);
      if (dt_is_float(a.offset_dt)) os = rne(os);  // torch.round; identity on integer tensors
    }
    double q;
    if (a.sub_dt == FFQ_F64) {
      double d = (a.div_dt == FFQ_F64) ? xs / ss
                                       : (double)round_stage(round_stage((float)xs, a.div_dt) /
                                                                 round_stage((float)ss, a.div_dt),
                                                             a.div_dt);
      q = clamp_nan(rne(d - os), a.lo, a.hi);
    } else {
      const float x = round_stage((float)xs, a.div_dt);
      const float s = round_stage((float)ss, a.div_dt);
      float d = round_stage(x / s, a.div_dt);
      const float o = round_stage((float)os, a.sub_dt);
      d = round_stage(round_stage(d, a.sub_dt) - o, a.sub_dt);
      q = (double)clamp_nan(rne(d), round_stage((float)a.lo, a.sub_dt), round_stage((float)a.hi, a.sub_dt));
    }
    store_any(out, a.out_dt, i, q);
  }
}

// (-DFFQ_EXPERIMENTS builds) FFQ_DIV_MODE=0 forces the compiler's IEEE division sequence in the streaming kernels;
// the default is the Markstein iteration (bit-identical, see Divider).
static int div_mode() {
#ifdef FFQ_EXPERIMENTS
  static const int mode = [] {
    const char* e = getenv("FFQ_DIV_MODE");
    return e ? atoi(e) : 1;
  }();
  return mode;
#else
  return 1;
#endif
}

static unsigned grid_for(int64_t work_items, int per_block) {
  int64_t blocks = (work_items + per_block - 1) / per_block;
  if (blocks < 1) blocks = 1;
  return (unsigned)blocks;
}

static int launch_generic(const void* data, int data_dt, const void* scale, int scale_dt,
                          int64_t scale_numel, const void* offset, int offset_dt, int64_t offset_numel,
                          const ffq_tiling* tiling, double lo, double hi, int div_dt, int sub_dt,
                          void* out, int out_dt, int64_t start, int64_t count, hipStream_t stream) {
  if (count <= 0) return FFQ_OK;
  GenericArgs a;
  a.data_dt = data_dt; a.scale_dt = scale_dt; a.offset_dt = offset_dt; a.out_dt = out_dt;
  a.div_dt = div_dt; a.sub_dt = sub_dt;
  a.has_offset = offset != nullptr;
  a.start = start; a.count = count;
  a.scale_numel = scale_numel; a.offset_numel = offset_numel;
  a.lo = lo; a.hi = hi;
  a.g = make_generic(tiling);
  int64_t blocks = (count + kBlock - 1) / kBlock;
  if (blocks > 8192) blocks = 8192;
  quantize_generic_kernel<<<dim3((unsigned)blocks), dim3(kBlock), 0, stream>>>(data, scale, offset, out, a);
  return check_launch("quantize_generic_kernel");
}

// chunks per lane: measured on MI355X (interleaved A/B): short blocks win. Also measured and rejected for 2-byte ->
// 1-byte: two coalesced 16-B loads per lane (chunks l and l + 64) with a lane-pair exchange into 16-B stores (4.2-5.3 TB/s)
// or with two 8-B stores (4.1-5.4 TB/s) against 6.0 TB/s for the 16-element chunk below —
// bf16 -> int8 [14336, 4096]: E=16/U=1 29.2 us vs E=16/U=2 32.4 us vs E=8/U=4 34.3 us; U=1 also wins for
// bf16 -> bf16 (38.5 vs 41.0 us at U=4) and per-tensor activations (33.2 vs 39.2 us).
static int stream_u_override() {
#ifdef FFQ_EXPERIMENTS
  const char* e = getenv("FFQ_STREAM_U");
  return e ? atoi(e) : 0;
#else
  return 0;
#endif
}

template <typename TIn, typename TOut, int E, int U>
static int launch_stream_u(const TIn* in, TOut* out, const float* scale, int64_t scale_numel,
                           const float* offset, int64_t offset_numel, const TileInfo& info, float lo,
                           float hi, hipStream_t stream) {
  StreamArgs a;
  a.lo = lo; a.hi = hi;
  a.nchunks = (uint32_t)(info.numel / E);
  a.scale_stride = scale_numel == 1 ? 0u : 1u;
  a.offset_stride = offset_numel == 1 ? 0u : 1u;
  a.chunks_per_run = make_fastdiv(1);
  a.channels = make_fastdiv(1);
  const unsigned grid = grid_for(a.nchunks, kBlock * U);
  const dim3 block(kBlock);
#define FFQ_LAUNCH_D(LAYOUT, D)                                                                      \
  do {                                                                                               \
    if (offset)                                                                                      \
      quantize_stream_kernel<TIn, TOut, LAYOUT, E, U, true, D><<<grid, block, 0, stream>>>(in, out, scale, offset, a); \
    else                                                                                             \
      quantize_stream_kernel<TIn, TOut, LAYOUT, E, U, false, D><<<grid, block, 0, stream>>>(in, out, scale, offset, a); \
  } while (0)
#ifdef FFQ_EXPERIMENTS  // the IEEE-division form of the streaming kernels exists in tuning builds only (round 4: 1,637 -> ~500 device kernels)
#define FFQ_LAUNCH(LAYOUT)                                                                           \
  do {                                                                                               \
    if (div_mode() == 1) FFQ_LAUNCH_D(LAYOUT, 1); else FFQ_LAUNCH_D(LAYOUT, 0);                      \
  } while (0)
#else
#define FFQ_LAUNCH(LAYOUT) FFQ_LAUNCH_D(LAYOUT, 1)
#endif
  switch (info.layout) {
    case LAYOUT_SCALAR: FFQ_LAUNCH(LAYOUT_SCALAR); break;
    case LAYOUT_ROWS:
      a.chunks_per_run = make_fastdiv((uint32_t)(info.run / E));
      FFQ_LAUNCH(LAYOUT_ROWS);
      break;
    default:
      a.chunks_per_run = make_fastdiv((uint32_t)(info.inner / E));
      a.channels = make_fastdiv((uint32_t)info.channels);
      FFQ_LAUNCH(LAYOUT_CHANNEL);
      break;
  }
#undef FFQ_LAUNCH
#undef FFQ_LAUNCH_D
  return check_launch("quantize_stream_kernel");
}

template <typename TIn, typename TOut, int E>
static int launch_stream(const TIn* in, TOut* out, const float* scale, int64_t scale_numel,
                         const float* offset, int64_t offset_numel, const TileInfo& info, float lo,
                         float hi, hipStream_t stream) {
#ifdef FFQ_EXPERIMENTS  // chunks per lane: one in the product (measured fastest, see above); 2 / 4 in tuning builds only
  int u = stream_u_override();
  if (u == 0) u = 1;
  switch (u) {
    case 1: return launch_stream_u<TIn, TOut, E, 1>(in, out, scale, scale_numel, offset, offset_numel, info, lo, hi, stream);
    case 4: return launch_stream_u<TIn, TOut, E, 4>(in, out, scale, scale_numel, offset, offset_numel, info, lo, hi, stream);
    default: return launch_stream_u<TIn, TOut, E, 2>(in, out, scale, scale_numel, offset, offset_numel, info, lo, hi, stream);
  }
#else
  return launch_stream_u<TIn, TOut, E, 1>(in, out, scale, scale_numel, offset, offset_numel, info, lo, hi, stream);
#endif
}

template <typename TIn, typename TOut, int E>
static int launch_columns(const TIn* in, TOut* out, const float* scale, int64_t scale_numel,
                          const float* offset, int64_t offset_numel, const TileInfo& info, float lo,
                          float hi, hipStream_t stream) {
  ColumnArgs a;
  a.lo = lo; a.hi = hi;
  a.col_chunks = (uint32_t)(info.channels / E);
  a.rows = (uint32_t)(info.numel / info.channels);
  a.scale_stride = scale_numel == 1 ? 0u : 1u;
  a.offset_stride = offset_numel == 1 ? 0u : 1u;
  a.col_chunks_div = make_fastdiv(a.col_chunks);
  // ~8 rows per lane amortise the 2*E parameter loads; never more groups than rows.
  uint32_t groups = (a.rows + 7) / 8;
  if (groups < 1) groups = 1;
  a.row_groups = groups;
  const uint64_t lanes = (uint64_t)groups * a.col_chunks;
  const unsigned grid = grid_for((int64_t)lanes, kBlock);
  if (offset)
    quantize_columns_kernel<TIn, TOut, E, true><<<grid, dim3(kBlock), 0, stream>>>(in, out, scale, offset, a);
  else
    quantize_columns_kernel<TIn, TOut, E, false><<<grid, dim3(kBlock), 0, stream>>>(in, out, scale, offset, a);
  return check_launch("quantize_columns_kernel");
}

// Decide whether the streaming kernels apply: fp32 stages + fp32 parameters + 16 B alignment + a
// layout whose tiles never split a chunk; everything else goes to the generic kernel.
template <typename TIn, typename TOut, int E>
static int dispatch_fast_e(const void* data, const void* scale, int64_t scale_numel, const void* offset,
                           int64_t offset_numel, const TileInfo& info, float lo, float hi, void* out,
                           hipStream_t stream, int64_t* done) {
  *done = 0;
  const TIn* in = static_cast<const TIn*>(data);
  TOut* o = static_cast<TOut*>(out);
  const float* s = static_cast<const float*>(scale);
  const float* f = static_cast<const float*>(offset);
  if (info.layout == LAYOUT_SCALAR || (info.layout == LAYOUT_ROWS && info.run % E == 0) ||
      (info.layout == LAYOUT_CHANNEL && info.inner % E == 0)) {
    if (info.numel / E == 0) return FFQ_OK;
    // only the SCALAR layout can leave a tail (numel % E); ROWS / CHANNEL tiles are whole chunks
    *done = info.layout == LAYOUT_SCALAR ? (info.numel / E) * E : info.numel;
    return launch_stream<TIn, TOut, E>(in, o, s, scale_numel, f, offset_numel, info, lo, hi, stream);
  }
  return FFQ_OK;
}

// Decide whether the streaming kernels apply: fp32 stages + fp32 parameters + 16 B alignment + a
// layout whose tiles never split a chunk; everything else goes to the generic kernel.
// One-byte containers use 16-element chunks so that the store is 16 B per lane as well.
template <typename TIn, typename TOut>
static int dispatch_fast(const void* data, const void* scale, int64_t scale_numel, const void* offset,
                         int64_t offset_numel, const TileInfo& info, float lo, float hi, void* out,
                         hipStream_t stream, int64_t* done) {
  *done = 0;
  if (info.numel >= ((int64_t)1 << 32) - 4096) return FFQ_OK;  // 32-bit chunk index
  if (!aligned16(data) || !aligned16(out)) return FFQ_OK;
  int rc = FFQ_OK;
  if constexpr (sizeof(TOut) == 1) {
    rc = dispatch_fast_e<TIn, TOut, 16>(data, scale, scale_numel, offset, offset_numel, info, lo, hi, out, stream, done);
    if (rc || *done) return rc;
  }
  rc = dispatch_fast_e<TIn, TOut, 8>(data, scale, scale_numel, offset, offset_numel, info, lo, hi, out, stream, done);
  if (rc || *done) return rc;
  if (info.layout == LAYOUT_CHANNEL && info.inner == 1 && info.channels % 8 == 0) {
    *done = info.numel;
    return launch_columns<TIn, TOut, 8>(static_cast<const TIn*>(data), static_cast<TOut*>(out),
                                        static_cast<const float*>(scale), scale_numel,
                                        static_cast<const float*>(offset), offset_numel, info, lo, hi, stream);
  }
  return FFQ_OK;
}

template <typename TIn>
static int dispatch_out(int out_dt, const void* data, const void* scale, int64_t scale_numel,
                        const void* offset, int64_t offset_numel, const TileInfo& info, float lo,
                        float hi, void* out, hipStream_t stream, int64_t* handled) {
  switch (out_dt) {
    case FFQ_F32: return dispatch_fast<TIn, float>(data, scale, scale_numel, offset, offset_numel, info, lo, hi, out, stream, handled);
    case FFQ_BF16: return dispatch_fast<TIn, bf16_t>(data, scale, scale_numel, offset, offset_numel, info, lo, hi, out, stream, handled);
    case FFQ_F16: return dispatch_fast<TIn, f16_t>(data, scale, scale_numel, offset, offset_numel, info, lo, hi, out, stream, handled);
    case FFQ_I8: return dispatch_fast<TIn, int8_t>(data, scale, scale_numel, offset, offset_numel, info, lo, hi, out, stream, handled);
    case FFQ_I16: return dispatch_fast<TIn, int16_t>(data, scale, scale_numel, offset, offset_numel, info, lo, hi, out, stream, handled);
    case FFQ_I32: return dispatch_fast<TIn, int32_t>(data, scale, scale_numel, offset, offset_numel, info, lo, hi, out, stream, handled);
    default: *handled = 0; return FFQ_OK;
  }
}

int quantize_impl(const void* data, int data_dt, const void* scale, int scale_dt, int64_t scale_numel,
                  const void* offset, int offset_dt, int64_t offset_numel, const ffq_tiling* tiling,
                  double num_bits, void* out, int out_dt, hipStream_t stream) {
  TileInfo info;
  int rc = analyse(tiling, &info);
  if (rc) return rc;
  if (!dt_valid(data_dt) || !dt_valid(scale_dt) || !dt_valid(out_dt) || (offset && !dt_valid(offset_dt)))
    return fail(FFQ_ERR_ARG, "bad dtype tag");
  if (info.numel != 0) {
    if ((rc = check_param_numel("scale", scale_numel, info.ntiles))) return rc;
    if (offset && (rc = check_param_numel("offset", offset_numel, info.ntiles))) return rc;
  }
  const int off_dt = offset ? offset_dt : scale_dt;
  // min_threshold = -(2 ** (num_bits - 1)); max_threshold = -min_threshold - 1      (:158-159)
  const double lo = -pow(2.0, num_bits - 1.0), hi = -lo - 1.0;
  int div_dt = ffq_promote_types(data_dt, scale_dt);
  if (!dt_is_float(div_dt)) div_dt = FFQ_F32;  // integer / integer is a true division in float
  const int sub_dt = ffq_promote_types(div_dt, off_dt);
  if (!ffq_can_support_bitwidth(out_dt, num_bits))
    return fail(FFQ_ERR_PRECISION, "Provided dtype (%d) is not enough to store %g bits quantized values.",
                out_dt, num_bits);
  if (info.numel == 0) return FFQ_OK;
  if (!data || !scale || !out) return fail(FFQ_ERR_ARG, "NULL buffer");

  const bool fast_types = num_bits == floor(num_bits) && num_bits >= 1 && num_bits <= 32 &&
                          div_dt == FFQ_F32 && sub_dt == FFQ_F32 && scale_dt == FFQ_F32 &&
                          (!offset || offset_dt == FFQ_F32) && !generic_kernels_forced();
  int64_t done = 0;
  if (fast_types) {
    switch (data_dt) {
      case FFQ_F32: rc = dispatch_out<float>(out_dt, data, scale, scale_numel, offset, offset_numel, info, (float)lo, (float)hi, out, stream, &done); break;
      case FFQ_BF16: rc = dispatch_out<bf16_t>(out_dt, data, scale, scale_numel, offset, offset_numel, info, (float)lo, (float)hi, out, stream, &done); break;
      case FFQ_F16: rc = dispatch_out<f16_t>(out_dt, data, scale, scale_numel, offset, offset_numel, info, (float)lo, (float)hi, out, stream, &done); break;
      default: break;
    }
    if (rc) return rc;
  }
  return launch_generic(data, data_dt, scale, scale_dt, scale_numel, offset, offset_dt, offset_numel,
                        tiling, lo, hi, div_dt, sub_dt, out, out_dt, done, info.numel - done, stream);
}

// A1 for a [rows, cols] weight with one parameter pair per row, int8 container, that also leaves the row sums of the codes
// (the zero-point term of A6) — one pass instead of A1 + a reduction over the codes. cols % 1024 == 0: the 64 chunks of a wave
// lie in one row, so a wave reduces its 1024 codes in registers (v_dot4 on the packed bytes, a shuffle butterfly) and adds one
// int32 to rowsum[row] (integer atomics: exact, order-independent; the caller zeroes the sums). Codes are those of
// quantize_stream_kernel.
template <bool HAS_OFFSET>
__global__ __launch_bounds__(kBlock) void quantize_rows_rowsum_kernel(const bf16_t* __restrict__ in, int8_t* __restrict__ out,
                                                                      const float* __restrict__ scale,
                                                                      const float* __restrict__ offset,
                                                                      int32_t* __restrict__ rowsum, StreamArgs a) {
  const uint32_t c = blockIdx.x * (uint32_t)kBlock + threadIdx.x;
  const bool live = c < a.nchunks;  // nchunks % 64 == 0: whole waves are live or not
  Chunk<bf16_t, 16> x;
  if (live) x.load(in + (size_t)c * 16);
  else {
#pragma unroll
    for (int i = 0; i < Chunk<bf16_t, 16>::kWords; ++i) x.w[i] = 0;
  }
  const uint32_t row = fdiv(live ? c : a.nchunks - 1, a.chunks_per_run);
  const float s = scale[row];
  const float o = HAS_OFFSET ? rne(offset[row]) : 0.0f;
  float xf[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) xf[i] = x.get(i);
  Chunk<int8_t, 16> y;
  quantize_chunk_to_bytes<16>(xf, s, o, a.lo, a.hi, y);
  if (live) y.store(out + (size_t)c * 16);
  int sum = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) sum = __builtin_amdgcn_sdot4((int)y.w[i], 0x01010101, sum, false);
  if (!live) sum = 0;
  // wave sum on the VALU (DPP inclusive scan: row_shr 1 / 2 / 4 / 8, then row_bcast15 / row_bcast31): no LDS round trips in
  // a wave that lives for one chunk; lane 63 ends up with the total
  sum += __builtin_amdgcn_update_dpp(0, sum, 0x111, 0xf, 0xf, false);
  sum += __builtin_amdgcn_update_dpp(0, sum, 0x112, 0xf, 0xf, false);
  sum += __builtin_amdgcn_update_dpp(0, sum, 0x114, 0xf, 0xf, false);
  sum += __builtin_amdgcn_update_dpp(0, sum, 0x118, 0xf, 0xf, false);
  sum += __builtin_amdgcn_update_dpp(0, sum, 0x142, 0xa, 0xf, false);
  sum += __builtin_amdgcn_update_dpp(0, sum, 0x143, 0xc, 0xf, false);
  // one atomic per block where its 4 waves share a row (cols % 4096 == 0, or the block does not straddle a row end): 64 waves
  // hammering the 16 sums of one cache line serialise in L2
  __shared__ int wave_sum[4];
  __shared__ uint32_t wave_row[4];
  const uint32_t wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 63 && live) {
    wave_sum[wave] = sum;
    wave_row[wave] = row;
  }
  __syncthreads();
  if (threadIdx.x == 63) {
    // only the LAST block can be short (nchunks % 64 == 0): count the waves this block really has
    const uint32_t waves = min(4u, (a.nchunks - blockIdx.x * (uint32_t)kBlock + 63u) / 64u);
    int acc = wave_sum[0];
    uint32_t cur = wave_row[0];
    for (uint32_t w = 1; w < waves; ++w) {
      if (wave_row[w] != cur) {
        atomicAdd(rowsum + cur, acc);
        acc = 0;
        cur = wave_row[w];
      }
      acc += wave_sum[w];
    }
    atomicAdd(rowsum + cur, acc);
  }
}


// ---- A1 of SEVERAL row-quantized weights in one launch (ffq_quantize_rows_batch) ---------------------------------------------------
// The reference re-quantizes all seven linears' weights of a decoder layer on every forward (nn/linear.py:34): seven launches,
// two of them (k_proj / v_proj, 4 M elements) far too short to reach the streaming rate (34 % of the HBM peak, profiles/r03_micro.md).
// One grid covers all members: every member's chunk count is a multiple of the block size, so a block lies inside ONE member and
// finds it with a handful of scalar compares; the arithmetic is quantize_stream_kernel's ROWS case (bf16 -> int8, one (scale,
// offset) per row, Markstein division).
struct BatchMember {
  const bf16_t* in;
  int8_t* out;
  const float* scale;
  const float* offset;
  int32_t* rowsum;          // ROWSUM: += sum of the member's codes per row (zero on entry; cols % 1024 == 0: a wave lies inside one row)
  uint32_t first_block;     // blocks [first_block, next member's first_block) belong to this member
  FastDiv chunks_per_row;   // cols / 16
};
struct BatchArgs {
  BatchMember m[FFQ_MAX_BATCH];
  int count;
  float lo, hi;
};

// ROWSUM (round 4): the int8 GEMM's zero-point term needs sum_k wq[n, k] of every weight (ffq_linear.hip); taken here from the
// codes while they are in registers (one v_dot4 per 4 codes, a DPP wave sum, at most two atomics per block) it replaces one
// rowsum_i8_kernel launch per linear and forward (224 launches, 2.1 ms of the Llama-3-8B step).
template <bool ROWSUM>
__global__ __launch_bounds__(kBlock) void quantize_rows_batch_kernel(BatchArgs a) {
  int k = 0;
#pragma unroll
  for (int i = 1; i < FFQ_MAX_BATCH; ++i)
    if (i < a.count && blockIdx.x >= a.m[i].first_block) k = i;
  // (k is wave-uniform: the selects below stay in scalar registers)
  const BatchMember& mem = a.m[k];
  const uint32_t c = (blockIdx.x - mem.first_block) * (uint32_t)kBlock + threadIdx.x;
  Chunk<bf16_t, 16> x;
  x.load(mem.in + (size_t)c * 16);
  const uint32_t row = fdiv(c, mem.chunks_per_row);
  const float s = mem.scale[row];
  const float o = mem.offset ? rne(mem.offset[row]) : 0.0f;
  float xf[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) xf[i] = x.get(i);
  Chunk<int8_t, 16> y;
  quantize_chunk_to_bytes<16>(xf, s, o, a.lo, a.hi, y);
  y.store(mem.out + (size_t)c * 16);
  if constexpr (ROWSUM) {
    int sum = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) sum = __builtin_amdgcn_sdot4((int)y.w[i], 0x01010101, sum, false);
    // wave sum on the VALU (DPP inclusive scan, as quantize_rows_rowsum_kernel): lane 63 ends up with the total
    sum += __builtin_amdgcn_update_dpp(0, sum, 0x111, 0xf, 0xf, false);
    sum += __builtin_amdgcn_update_dpp(0, sum, 0x112, 0xf, 0xf, false);
    sum += __builtin_amdgcn_update_dpp(0, sum, 0x114, 0xf, 0xf, false);
    sum += __builtin_amdgcn_update_dpp(0, sum, 0x118, 0xf, 0xf, false);
    sum += __builtin_amdgcn_update_dpp(0, sum, 0x142, 0xa, 0xf, false);
    sum += __builtin_amdgcn_update_dpp(0, sum, 0x143, 0xc, 0xf, false);
    __shared__ int wave_sum[4];
    __shared__ uint32_t wave_row[4];
    const uint32_t wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 63) {
      wave_sum[wave] = sum;
      wave_row[wave] = row;
    }
    __syncthreads();
    if (threadIdx.x == 63) {  // one atomic per row the block touches (its four waves usually share one)
      int acc = wave_sum[0];
      uint32_t cur = wave_row[0];
      for (uint32_t w = 1; w < 4; ++w) {
        if (wave_row[w] != cur) {
          atomicAdd(mem.rowsum + cur, acc);
          acc = 0;
          cur = wave_row[w];
        }
        acc += wave_sum[w];
      }
      atomicAdd(mem.rowsum + cur, acc);
    }
  }
}


// A1 of a per-tensor quantizer into int8 UNLESS an earlier quantizer's parameters are the same (ffq_affine.h: EarlierCodes) — then the
// launch returns before its first load and `out` keeps whatever it held. Otherwise quantize_stream_kernel's codes (its E = 16 body).
// At most kUnlessSameBlocks blocks striding over the chunks: the launch that does nothing is the common one (siblings that have
// seen the same data agree), and what it costs is its blocks' dispatch — 16384 one-chunk blocks for 16 k tokens x 4096: 6.1 us.
constexpr unsigned kUnlessSameBlocks = 2048;
template <typename TIn>
__global__ __launch_bounds__(kBlock) void quantize_unless_same_kernel(const TIn* __restrict__ in, int8_t* __restrict__ out,
                                                                      const float* __restrict__ scale, const float* __restrict__ offset,
                                                                      const float* __restrict__ scale2, const float* __restrict__ offset2,
                                                                      float lo, float hi, uint32_t nchunks) {
  if (same_parameters(scale, offset, scale2, offset2)) return;
  const float s = scale[0], o = offset ? rne(offset[0]) : 0.0f;
  const uint32_t stride = gridDim.x * (uint32_t)kBlock;
  for (uint32_t c = blockIdx.x * (uint32_t)kBlock + threadIdx.x; c < nchunks; c += 2 * stride) {
    const uint32_t c2 = c + stride;
    Chunk<TIn, 16> x, x2;
    x.load(in + (size_t)c * 16);
    if (c2 < nchunks) x2.load(in + (size_t)c2 * 16);
    float xf[16];
    Chunk<int8_t, 16> y;
#pragma unroll
    for (int i = 0; i < 16; ++i) xf[i] = x.get(i);
    quantize_chunk_to_bytes<16>(xf, s, o, lo, hi, y);
    y.store(out + (size_t)c * 16);
    if (c2 < nchunks) {
#pragma unroll
      for (int i = 0; i < 16; ++i) xf[i] = x2.get(i);
      quantize_chunk_to_bytes<16>(xf, s, o, lo, hi, y);
      y.store(out + (size_t)c2 * 16);
    }
  }
}

}  // namespace ffq

extern "C" int ffq_quantize_by_tile(const void* data, int data_dt, const void* scale, int scale_dt,
                                    int64_t scale_numel, const void* offset, int offset_dt,
                                    int64_t offset_numel, const ffq_tiling* tiling, double num_bits,
                                    void* out, int out_dt, void* stream) {
  return ffq::quantize_impl(data, data_dt, scale, scale_dt, scale_numel, offset, offset_dt, offset_numel,
                            tiling, num_bits, out, out_dt, static_cast<hipStream_t>(stream));
}

extern "C" int ffq_quantize_by_tile_unless_same(const void* data, int data_dt, const float* scale, const float* offset, int64_t numel,
                                                double num_bits, const float* earlier_scale, const float* earlier_offset, int8_t* out,
                                                void* stream) {
  using namespace ffq;
  if (numel < 0) return fail(FFQ_ERR_ARG, "negative extent");
  if (!ffq_can_support_bitwidth(FFQ_I8, num_bits))
    return fail(FFQ_ERR_PRECISION, "Provided dtype (%d) is not enough to store %g bits quantized values.", FFQ_I8, num_bits);
  if (numel == 0) return FFQ_OK;
  if (!data || !scale || !earlier_scale || !out) return fail(FFQ_ERR_ARG, "NULL buffer");
  if (num_bits != floor(num_bits) || num_bits < 1 || numel % 16 != 0 || numel / 16 >= ((int64_t)1 << 32) - 4096 || !aligned16(data) || !aligned16(out) ||
      !(data_dt == FFQ_F32 || data_dt == FFQ_BF16 || data_dt == FFQ_F16))
    return fail(FFQ_ERR_DTYPE, "quantize unless same: whole 16-element chunks of f32 / bf16 / f16 data, 16-byte aligned (else ffq_quantize_by_tile)");
  const double lo = -pow(2.0, num_bits - 1.0), hi = -lo - 1.0;
  const uint32_t nchunks = (uint32_t)(numel / 16);
  unsigned grid = grid_for(nchunks, kBlock);
  if (grid > kUnlessSameBlocks) grid = kUnlessSameBlocks;
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (data_dt) {
    case FFQ_F32: quantize_unless_same_kernel<float><<<grid, kBlock, 0, s>>>(static_cast<const float*>(data), out, scale, offset, earlier_scale, earlier_offset, (float)lo, (float)hi, nchunks); break;
    case FFQ_BF16: quantize_unless_same_kernel<bf16_t><<<grid, kBlock, 0, s>>>(static_cast<const bf16_t*>(data), out, scale, offset, earlier_scale, earlier_offset, (float)lo, (float)hi, nchunks); break;
    default: quantize_unless_same_kernel<f16_t><<<grid, kBlock, 0, s>>>(static_cast<const f16_t*>(data), out, scale, offset, earlier_scale, earlier_offset, (float)lo, (float)hi, nchunks); break;
  }
  return check_launch("quantize_unless_same_kernel");
}

extern "C" int ffq_quantize_rows_rowsum(const void* data, int data_dt, const float* scale, const float* offset, int64_t rows,
                                        int64_t cols, double num_bits, int8_t* codes, int32_t* rowsum, void* stream) {
  using namespace ffq;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (rows < 0 || cols < 0) return fail(FFQ_ERR_ARG, "negative extent");
  if (data_dt != FFQ_BF16) return fail(FFQ_ERR_DTYPE, "fused weight quantize + row sums is built for bf16 weights");
  if (cols % 1024 != 0) return fail(FFQ_ERR_DTYPE, "fused weight quantize + row sums needs cols %% 1024 == 0");
  if (!ffq_can_support_bitwidth(FFQ_I8, num_bits))
    return fail(FFQ_ERR_PRECISION, "Provided dtype (%d) is not enough to store %g bits quantized values.", FFQ_I8, num_bits);
  // the kernel clamps BEFORE it rounds (quantize_chunk_to_bytes), which equals the reference's round-then-clamp only for integer clamp
  // bounds: a fractional bit width goes back to ffq_quantize_by_tile
  if (num_bits != floor(num_bits) || num_bits < 1) return fail(FFQ_ERR_DTYPE, "fused weight quantize + row sums needs an integral bit width (else ffq_quantize_by_tile)");
  if (rows == 0 || cols == 0) return FFQ_OK;
  if (!data || !scale || !codes || !rowsum) return fail(FFQ_ERR_ARG, "NULL buffer");
  if (!aligned16(data) || !aligned16(codes)) return fail(FFQ_ERR_ARG, "buffers must be 16-byte aligned");
  const int64_t nchunks = rows * cols / 16;
  if (nchunks >= ((int64_t)1 << 32)) return fail(FFQ_ERR_ARG, "too many elements for one launch");
  StreamArgs a;
  const double lo = -pow(2.0, num_bits - 1.0);
  a.lo = (float)lo; a.hi = (float)(-lo - 1.0);
  a.nchunks = (uint32_t)nchunks;
  a.scale_stride = 1; a.offset_stride = 1;
  a.chunks_per_run = make_fastdiv((uint32_t)(cols / 16));
  a.channels = make_fastdiv(1);
  const unsigned grid = (unsigned)((nchunks + kBlock - 1) / kBlock);
  if (offset) quantize_rows_rowsum_kernel<true><<<grid, kBlock, 0, s>>>(static_cast<const bf16_t*>(data), codes, scale, offset, rowsum, a);
  else quantize_rows_rowsum_kernel<false><<<grid, kBlock, 0, s>>>(static_cast<const bf16_t*>(data), codes, scale, offset, rowsum, a);
  return check_launch("quantize_rows_rowsum_kernel");
}

extern "C" int ffq_quantize_rows_batch(const ffq_rows_batch* batch, int data_dt, void* stream) {
  using namespace ffq;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (!batch || batch->count < 0 || batch->count > FFQ_MAX_BATCH) return fail(FFQ_ERR_ARG, "batch count must be 0..%d", FFQ_MAX_BATCH);
  if (batch->count == 0) return FFQ_OK;
  if (data_dt != FFQ_BF16) return fail(FFQ_ERR_DTYPE, "batched weight quantization is built for bf16 weights");
  if (!ffq_can_support_bitwidth(FFQ_I8, batch->num_bits))
    return fail(FFQ_ERR_PRECISION, "Provided dtype (%d) is not enough to store %g bits quantized values.", FFQ_I8, batch->num_bits);
  if (batch->num_bits != floor(batch->num_bits) || batch->num_bits < 1)  // clamp-before-round needs integer bounds (see ffq_quantize_rows_rowsum)
    return fail(FFQ_ERR_DTYPE, "batched weight quantization needs an integral bit width (else ffq_quantize_by_tile per weight)");
  BatchArgs a;
  a.count = batch->count;
  const double lo = -pow(2.0, batch->num_bits - 1.0);
  a.lo = (float)lo; a.hi = (float)(-lo - 1.0);
  uint64_t blocks = 0;
  const bool rowsums = batch->rowsum[0] != nullptr;
  for (int i = 0; i < batch->count; ++i) {
    const int64_t rows = batch->rows[i], cols = batch->cols[i];
    if ((batch->rowsum[i] != nullptr) != rowsums) return fail(FFQ_ERR_ARG, "row sums for every member of the batch or for none");
    if (rowsums && cols % 1024 != 0) return fail(FFQ_ERR_DTYPE, "batched weight quantization with row sums needs cols %% 1024 == 0");
    if (rows <= 0 || cols <= 0 || cols % 16 != 0 || (rows * cols / 16) % kBlock != 0)
      return fail(FFQ_ERR_DTYPE, "batched weight quantization needs rows * cols %% %d == 0 and cols %% 16 == 0", 16 * kBlock);
    if (!batch->data[i] || !batch->scale[i] || !batch->codes[i]) return fail(FFQ_ERR_ARG, "NULL buffer in batch member %d", i);
    if (!aligned16(batch->data[i]) || !aligned16(batch->codes[i])) return fail(FFQ_ERR_ARG, "buffers must be 16-byte aligned");
    a.m[i].in = static_cast<const bf16_t*>(batch->data[i]);
    a.m[i].out = batch->codes[i];
    a.m[i].scale = batch->scale[i];
    a.m[i].offset = batch->offset[i];
    a.m[i].rowsum = batch->rowsum[i];
    a.m[i].first_block = (uint32_t)blocks;
    a.m[i].chunks_per_row = make_fastdiv((uint32_t)(cols / 16));
    blocks += (uint64_t)(rows * cols / 16) / kBlock;
    if (blocks >= ((uint64_t)1 << 31)) return fail(FFQ_ERR_ARG, "too many elements for one launch");
  }
  for (int i = batch->count; i < FFQ_MAX_BATCH; ++i) a.m[i] = a.m[0];
  if (rowsums) quantize_rows_batch_kernel<true><<<(unsigned)blocks, kBlock, 0, s>>>(a);
  else quantize_rows_batch_kernel<false><<<(unsigned)blocks, kBlock, 0, s>>>(a);
  return check_launch("quantize_rows_batch_kernel");
}

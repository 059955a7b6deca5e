// ffq_dequantize.hip — A2: fastforward::dequantize_by_tile on gfx950.
//
// Reference: dequantize_by_tile_impl, src/fastforward/quantization/_quantizer_impl.py:172-190
//   x^ = cast((q + round(o_t)) * s_t)
// Reference cost: add, mul, _to_copy passes with fp32 temporaries. Here: one pass, 16 B per lane.
// HBM-bound: 3 B/elem for int8 -> bf16, 4 B/elem for a bf16 container.
//
// The add and the multiply are two separately rounded fp32 operations (no FMA), and `q + 0.0f` is
// kept when there is no offset because it turns -0.0 into +0.0 exactly like the eager chain.
#include "ffq_common.h"
#include "ffq_vec.h"

#include <stdlib.h>

namespace ffq {

struct DqStreamArgs {
  uint32_t nchunks;
  uint32_t scale_stride;
  uint32_t offset_stride;
  FastDiv chunks_per_run;
  FastDiv channels;
};

template <int LAYOUT>
__device__ __forceinline__ uint32_t dq_tile_of_chunk(uint32_t chunk, const DqStreamArgs& a) {
  if constexpr (LAYOUT == LAYOUT_SCALAR) {
    return 0;
  } else if constexpr (LAYOUT == LAYOUT_ROWS) {
    return fdiv(chunk, a.chunks_per_run);
  } else {
    const uint32_t outer = fdiv(chunk, a.chunks_per_run);
    return outer - fdiv(outer, a.channels) * a.channels.div;
  }
}

template <typename TIn, typename TOut, int LAYOUT, int E, int U, bool HAS_OFFSET>
__global__ __launch_bounds__(kBlock) void dequantize_stream_kernel(const TIn* __restrict__ in,
                                                                   TOut* __restrict__ out,
                                                                   const float* __restrict__ scale,
                                                                   const float* __restrict__ offset,
                                                                   DqStreamArgs a) {
  const uint32_t first = blockIdx.x * (uint32_t)(kBlock * U) + threadIdx.x;
  Chunk<TIn, E> x[U];
  float s[U], o[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const uint32_t c = first + u * kBlock;
    if (c < a.nchunks) x[u].load(in + (size_t)c * E);
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const uint32_t c = first + u * kBlock;
    s[u] = 1.0f;
    o[u] = 0.0f;
    if (c < a.nchunks) {
      const uint32_t t = dq_tile_of_chunk<LAYOUT>(c, a);
      s[u] = scale[t * a.scale_stride];
      if constexpr (HAS_OFFSET) o[u] = offset[t * a.offset_stride];
    }
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const uint32_t c = first + u * kBlock;
    if (c >= a.nchunks) continue;
    const float off = HAS_OFFSET ? rne(o[u]) : 0.0f;
    float v[E];
#pragma unroll
    for (int i = 0; i < E; ++i) {
      float q = x[u].get(i) + off;
      v[i] = q * s[u];
    }
    Chunk<TOut, E> y;
    y.pack(v);
    y.store(out + (size_t)c * E);
  }
}

struct DqColumnArgs {
  uint32_t col_chunks, rows, row_groups, scale_stride, offset_stride;
  FastDiv col_chunks_div;
};

template <typename TIn, typename TOut, int E, bool HAS_OFFSET>
__global__ __launch_bounds__(kBlock) void dequantize_columns_kernel(const TIn* __restrict__ in,
                                                                    TOut* __restrict__ out,
                                                                    const float* __restrict__ scale,
                                                                    const float* __restrict__ offset,
                                                                    DqColumnArgs a) {
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
  for (uint32_t row = group; row < a.rows; row += a.row_groups) {
    const size_t at = (size_t)row * row_elems + (size_t)cc * E;
    Chunk<TIn, E> x;
    x.load(in + at);
    float v[E];
#pragma unroll
    for (int i = 0; i < E; ++i) {
      float q = x.get(i) + o[i];
      v[i] = q * s[i];
    }
    Chunk<TOut, E> y;
    y.pack(v);
    y.store(out + at);
  }
}

struct DqGenericArgs {
  int data_dt, scale_dt, offset_dt, out_dt;
  int add_dt, mul_dt;
  int has_offset;
  int64_t start, count;
  int64_t scale_numel, offset_numel;
  GenericTiling g;
};

__global__ __launch_bounds__(kBlock) void dequantize_generic_kernel(const void* __restrict__ data,
                                                                    const void* __restrict__ scale,
                                                                    const void* __restrict__ offset,
                                                                    void* __restrict__ out,
                                                                    DqGenericArgs a) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x; k < a.count; k += stride) {
    const int64_t i = a.start + k;
    const int64_t t = generic_tile_of(a.g, i);
    const double qs = load_any(data, a.data_dt, i);
    const double ss = load_any(scale, a.scale_dt, a.scale_numel == 1 ? 0 : t);
    double os = 0.0;
    if (a.has_offset) {
      os = load_any(offset, a.offset_dt, a.offset_numel == 1 ? 0 : t);
      if (dt_is_float(a.offset_dt)) os = rne(os);
    }
    double v;
    if (a.mul_dt == FFQ_F64) {
      double sum;
      if (a.add_dt == FFQ_F64) sum = qs + os;
      else if (dt_is_float(a.add_dt))
        sum = (double)round_stage(round_stage((float)qs, a.add_dt) + round_stage((float)os, a.add_dt), a.add_dt);
      else sum = qs + os;
      v = sum * ss;
    } else {
      float sum;
      if (dt_is_float(a.add_dt))
        sum = round_stage(round_stage((float)qs, a.add_dt) + round_stage((float)os, a.add_dt), a.add_dt);
      else
        sum = (float)(qs + os);  // integer add, exact; converted by the multiply's promotion
      v = (double)round_stage(round_stage(sum, a.mul_dt) * round_stage((float)ss, a.mul_dt), a.mul_dt);
    }
    store_any(out, a.out_dt, i, v);
  }
}

template <typename TIn, typename TOut, int E, int U>
static int dq_launch_stream_u(const TIn* in, TOut* out, const float* scale, int64_t scale_numel,
                              const float* offset, int64_t offset_numel, const TileInfo& info,
                              hipStream_t stream) {
  DqStreamArgs a;
  a.nchunks = (uint32_t)(info.numel / E);
  a.scale_stride = scale_numel == 1 ? 0u : 1u;
  a.offset_stride = offset_numel == 1 ? 0u : 1u;
  a.chunks_per_run = make_fastdiv(1);
  a.channels = make_fastdiv(1);
  const unsigned grid = (unsigned)((a.nchunks + kBlock * U - 1) / (kBlock * U));
  const dim3 block(kBlock);
#define FFQ_LAUNCH(LAYOUT)                                                                             \
  do {                                                                                                 \
    if (offset)                                                                                        \
      dequantize_stream_kernel<TIn, TOut, LAYOUT, E, U, true><<<grid, block, 0, stream>>>(in, out, scale, offset, a); \
    else                                                                                               \
      dequantize_stream_kernel<TIn, TOut, LAYOUT, E, U, false><<<grid, block, 0, stream>>>(in, out, scale, offset, a); \
  } while (0)
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
  return check_launch("dequantize_stream_kernel");
}

// chunks per lane: short blocks win on MI355X (same finding as ffq_quantize.hip::launch_stream)
template <typename TIn, typename TOut, int E>
static int dq_launch_stream(const TIn* in, TOut* out, const float* scale, int64_t scale_numel,
                            const float* offset, int64_t offset_numel, const TileInfo& info,
                            hipStream_t stream) {
#ifdef FFQ_EXPERIMENTS  // 2 / 4 chunks per lane: tuning builds only (the product instantiates the one form it launches)
  int u = 1;
  if (const char* e = getenv("FFQ_STREAM_U")) u = atoi(e) ? atoi(e) : 1;
  switch (u) {
    case 1: return dq_launch_stream_u<TIn, TOut, E, 1>(in, out, scale, scale_numel, offset, offset_numel, info, stream);
    case 4: return dq_launch_stream_u<TIn, TOut, E, 4>(in, out, scale, scale_numel, offset, offset_numel, info, stream);
    default: return dq_launch_stream_u<TIn, TOut, E, 2>(in, out, scale, scale_numel, offset, offset_numel, info, stream);
  }
#else
  return dq_launch_stream_u<TIn, TOut, E, 1>(in, out, scale, scale_numel, offset, offset_numel, info, stream);
#endif
}

template <typename TIn, typename TOut, int E>
static int dq_launch_columns(const TIn* in, TOut* out, const float* scale, int64_t scale_numel,
                             const float* offset, int64_t offset_numel, const TileInfo& info,
                             hipStream_t stream) {
  DqColumnArgs a;
  a.col_chunks = (uint32_t)(info.channels / E);
  a.rows = (uint32_t)(info.numel / info.channels);
  a.scale_stride = scale_numel == 1 ? 0u : 1u;
  a.offset_stride = offset_numel == 1 ? 0u : 1u;
  a.col_chunks_div = make_fastdiv(a.col_chunks);
  uint32_t groups = (a.rows + 7) / 8;
  if (groups < 1) groups = 1;
  a.row_groups = groups;
  const uint64_t lanes = (uint64_t)groups * a.col_chunks;
  const unsigned grid = (unsigned)((lanes + kBlock - 1) / kBlock);
  if (offset)
    dequantize_columns_kernel<TIn, TOut, E, true><<<grid, dim3(kBlock), 0, stream>>>(in, out, scale, offset, a);
  else
    dequantize_columns_kernel<TIn, TOut, E, false><<<grid, dim3(kBlock), 0, stream>>>(in, out, scale, offset, a);
  return check_launch("dequantize_columns_kernel");
}

template <typename TIn, typename TOut, int E>
static int dq_dispatch_fast_e(const void* data, const void* scale, int64_t scale_numel, const void* offset,
                              int64_t offset_numel, const TileInfo& info, void* out, hipStream_t stream,
                              int64_t* done) {
  *done = 0;
  const TIn* in = static_cast<const TIn*>(data);
  TOut* o = static_cast<TOut*>(out);
  const float* s = static_cast<const float*>(scale);
  const float* f = static_cast<const float*>(offset);
  if (info.layout == LAYOUT_SCALAR || (info.layout == LAYOUT_ROWS && info.run % E == 0) ||
      (info.layout == LAYOUT_CHANNEL && info.inner % E == 0)) {
    if (info.numel / E == 0) return FFQ_OK;
    *done = info.layout == LAYOUT_SCALAR ? (info.numel / E) * E : info.numel;
    return dq_launch_stream<TIn, TOut, E>(in, o, s, scale_numel, f, offset_numel, info, stream);
  }
  return FFQ_OK;
}

template <typename TIn, typename TOut>
static int dq_dispatch_fast(const void* data, const void* scale, int64_t scale_numel, const void* offset,
                            int64_t offset_numel, const TileInfo& info, void* out, hipStream_t stream,
                            int64_t* done) {
  *done = 0;
  if (info.numel >= ((int64_t)1 << 32) - 4096) return FFQ_OK;
  if (!aligned16(data) || !aligned16(out)) return FFQ_OK;
  int rc = FFQ_OK;
  // 8 codes per chunk: an 8 B load and ONE dense 16 B store per lane beats 16 codes per chunk (16 B
  // load, two half-dense 16 B stores): 30.1 vs 32.1 us on [14336, 4096] int8 -> bf16 (FFQ_DQ_E16=1 to compare)
#ifdef FFQ_EXPERIMENTS
  const char* e16 = getenv("FFQ_DQ_E16");
  if (sizeof(TIn) == 1 && e16 && e16[0] == '1') {
    rc = dq_dispatch_fast_e<TIn, TOut, 16>(data, scale, scale_numel, offset, offset_numel, info, out, stream, done);
    if (rc || *done) return rc;
  }
#endif
  rc = dq_dispatch_fast_e<TIn, TOut, 8>(data, scale, scale_numel, offset, offset_numel, info, out, stream, done);
  if (rc || *done) return rc;
  if (info.layout == LAYOUT_CHANNEL && info.inner == 1 && info.channels % 8 == 0) {
    *done = info.numel;
    return dq_launch_columns<TIn, TOut, 8>(static_cast<const TIn*>(data), static_cast<TOut*>(out),
                                           static_cast<const float*>(scale), scale_numel,
                                           static_cast<const float*>(offset), offset_numel, info, stream);
  }
  return FFQ_OK;
}

template <typename TIn>
static int dq_dispatch_out(int out_dt, const void* data, const void* scale, int64_t scale_numel,
                           const void* offset, int64_t offset_numel, const TileInfo& info, void* out,
                           hipStream_t stream, int64_t* done) {
  switch (out_dt) {
    case FFQ_F32: return dq_dispatch_fast<TIn, float>(data, scale, scale_numel, offset, offset_numel, info, out, stream, done);
    case FFQ_BF16: return dq_dispatch_fast<TIn, bf16_t>(data, scale, scale_numel, offset, offset_numel, info, out, stream, done);
    case FFQ_F16: return dq_dispatch_fast<TIn, f16_t>(data, scale, scale_numel, offset, offset_numel, info, out, stream, done);
    default: *done = 0; return FFQ_OK;
  }
}

int dequantize_impl(const void* data, int data_dt, const void* scale, int scale_dt, int64_t scale_numel,
                    const void* offset, int offset_dt, int64_t offset_numel, const ffq_tiling* tiling,
                    void* out, int out_dt, hipStream_t stream) {
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
  // (row + offset[:, None]) * scale[:, None]                                           (:182)
  const int add_dt = ffq_promote_types(data_dt, off_dt);
  const int mul_dt = ffq_promote_types(add_dt, scale_dt);
  if (!dt_is_float(mul_dt)) return fail(FFQ_ERR_DTYPE, "integer-only dequantize is not built");
  if (info.numel == 0) return FFQ_OK;
  if (!data || !scale || !out) return fail(FFQ_ERR_ARG, "NULL buffer");

  const bool fast_types = add_dt == FFQ_F32 && mul_dt == FFQ_F32 && scale_dt == FFQ_F32 &&
                          (!offset || offset_dt == FFQ_F32) && !generic_kernels_forced();
  int64_t done = 0;
  if (fast_types) {
    switch (data_dt) {
      case FFQ_I8: rc = dq_dispatch_out<int8_t>(out_dt, data, scale, scale_numel, offset, offset_numel, info, out, stream, &done); break;
      case FFQ_I16: rc = dq_dispatch_out<int16_t>(out_dt, data, scale, scale_numel, offset, offset_numel, info, out, stream, &done); break;
      case FFQ_I32: rc = dq_dispatch_out<int32_t>(out_dt, data, scale, scale_numel, offset, offset_numel, info, out, stream, &done); break;
      case FFQ_F32: rc = dq_dispatch_out<float>(out_dt, data, scale, scale_numel, offset, offset_numel, info, out, stream, &done); break;
      case FFQ_BF16: rc = dq_dispatch_out<bf16_t>(out_dt, data, scale, scale_numel, offset, offset_numel, info, out, stream, &done); break;
      case FFQ_F16: rc = dq_dispatch_out<f16_t>(out_dt, data, scale, scale_numel, offset, offset_numel, info, out, stream, &done); break;
      default: break;
    }
    if (rc) return rc;
  }
  const int64_t rest = info.numel - done;
  if (rest <= 0) return FFQ_OK;
  DqGenericArgs a;
  a.data_dt = data_dt; a.scale_dt = scale_dt; a.offset_dt = offset_dt; a.out_dt = out_dt;
  a.add_dt = add_dt; a.mul_dt = mul_dt;
  a.has_offset = offset != nullptr;
  a.start = done; a.count = rest;
  a.scale_numel = scale_numel; a.offset_numel = offset_numel;
  a.g = make_generic(tiling);
  int64_t blocks = (rest + kBlock - 1) / kBlock;
  if (blocks > 8192) blocks = 8192;
  dequantize_generic_kernel<<<dim3((unsigned)blocks), dim3(kBlock), 0, stream>>>(data, scale, offset, out, a);
  return check_launch("dequantize_generic_kernel");
}

}  // namespace ffq

extern "C" int ffq_dequantize_by_tile(const void* data, int data_dt, const void* scale, int scale_dt,
                                      int64_t scale_numel, const void* offset, int offset_dt,
                                      int64_t offset_numel, const ffq_tiling* tiling, void* out,
                                      int out_dt, void* stream) {
  return ffq::dequantize_impl(data, data_dt, scale, scale_dt, scale_numel, offset, offset_dt, offset_numel,
                              tiling, out, out_dt, static_cast<hipStream_t>(stream));
}

// ffq_backward.hip — A8: fastforward::quantize_by_tile_backward on gfx950.
//
// Reference: quant_dequant_by_tile_grad_impl, src/fastforward/quantization/_quantizer_impl.py:193-237 —
// the straight-through / LSQ gradients of quantize -> dequantize with respect to data, scale and offset:
//   u = x / s_t - round(o_t);  q = round(u);  clip = (q < lo) | (q > hi)
//   dinput  = clip ? 0 : g
//   doffset = sum_tile( clip ? s_t * g : 0 )
//   dscale  = sum_tile( (clip ? (q < lo ? lo : hi) + round(o_t) : q - u) * g )  (_infer_offset rounds, :140)
// The reference runs ~14 ATen passes with fp32 temporaries. Here: ONE streaming pass (x and g in, dinput
// out, 6 B/elem for bf16) that leaves one (dscale, doffset) partial per 2048-element block or per
// 8-element chunk, and a finalize pass over the partials in a FIXED order — the sums are deterministic
// (no floating-point atomics), their summation order is this file's own. Tilings the streaming pass does not cover
// (strided channels, N-d tiles) take quantize_backward_tiles_kernel: the same arithmetic, walked tile by tile.
#ifndef FFQ_NT_STREAMS
#define FFQ_NT_STREAMS 1  // nt loads (ffq_vec.h)
#endif
#include "ffq_common.h"
#include "ffq_vec.h"

#include <math.h>

namespace ffq {

struct BwdArgs {
  float lo, hi;
  uint32_t nchunks;
  uint32_t scale_stride, offset_stride;
  FastDiv chunks_per_run;  // ROWS: run / 8; SCALAR: unused
  int rows;                // 1: ROWS layout, 0: SCALAR
  int per_block;           // 1: every block lies inside one tile -> one partial per block
};

struct Partial2 { float ds, dof; };

template <typename T, bool HAS_OFFSET>
__global__ __launch_bounds__(kBlock) void quantize_backward_kernel(const T* __restrict__ x, const T* __restrict__ g,
                                                                  T* __restrict__ dinput,
                                                                  const float* __restrict__ scale,
                                                                  const float* __restrict__ offset,
                                                                  Partial2* __restrict__ partials, BwdArgs a) {
  constexpr int E = 8;
  __shared__ Partial2 wave_part[kBlock / 64];
  const uint32_t c = blockIdx.x * (uint32_t)kBlock + threadIdx.x;
  float ds = 0.0f, dof = 0.0f;
  if (c < a.nchunks) {
    Chunk<T, E> cx, cg;
    cx.FFQ_SLOAD(x + (size_t)c * E);
    cg.FFQ_SLOAD(g + (size_t)c * E);
    const uint32_t t = a.rows ? fdiv(c, a.chunks_per_run) : 0u;
    const float s = scale[t * a.scale_stride];
    const float o = HAS_OFFSET ? offset[t * a.offset_stride] : 0.0f;
    const float ro = rne(o);
    float di[E];
#pragma unroll
    for (int i = 0; i < E; ++i) {
      const float xv = cx.get(i), gv = cg.get(i);
      const float u = xv / s - ro;
      const float q = rne(u);
      const bool below = q < a.lo, above = q > a.hi;
      const bool clip = below || above;
      di[i] = clip ? 0.0f : gv;
      const float bound = (below ? a.lo : a.hi) + ro;
      const float term = (clip ? bound : q - u) * gv;
      ds = ds + term;
      if constexpr (HAS_OFFSET) dof = dof + (clip ? s * gv : 0.0f);
    }
    Chunk<T, E> out;
    out.pack(di);
    out.FFQ_SSTORE(dinput + (size_t)c * E);
  }
  if (!a.per_block) {
    if (c < a.nchunks) partials[c] = Partial2{ds, dof};
    return;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    ds = ds + __shfl_xor(ds, d, 64);
    dof = dof + __shfl_xor(dof, d, 64);
  }
  if ((threadIdx.x & 63) == 0) wave_part[threadIdx.x >> 6] = Partial2{ds, dof};
  __syncthreads();
  if (threadIdx.x == 0) {
    Partial2 p = wave_part[0];
#pragma unroll
    for (int w = 1; w < kBlock / 64; ++w) { p.ds = p.ds + wave_part[w].ds; p.dof = p.dof + wave_part[w].dof; }
    partials[blockIdx.x] = p;
  }
}

// One block per tile: fixed-order sum of the tile's `units` consecutive partials.
__global__ __launch_bounds__(kBlock) void backward_finalize_kernel(const Partial2* __restrict__ partials, uint32_t units,
                                                                  float* __restrict__ dscale, float* __restrict__ doffset) {
  __shared__ Partial2 wave_part[kBlock / 64];
  const Partial2* p = partials + (size_t)blockIdx.x * units;
  float ds = 0.0f, dof = 0.0f;
  for (uint32_t u = threadIdx.x; u < units; u += kBlock) { ds = ds + p[u].ds; dof = dof + p[u].dof; }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    ds = ds + __shfl_xor(ds, d, 64);
    dof = dof + __shfl_xor(dof, d, 64);
  }
  if ((threadIdx.x & 63) == 0) wave_part[threadIdx.x >> 6] = Partial2{ds, dof};
  __syncthreads();
  if (threadIdx.x == 0) {
    Partial2 r = wave_part[0];
#pragma unroll
    for (int w = 1; w < kBlock / 64; ++w) { r.ds = r.ds + wave_part[w].ds; r.dof = r.dof + wave_part[w].dof; }
    dscale[blockIdx.x] = r.ds;
    if (doffset) doffset[blockIdx.x] = r.dof;
  }
}

// Tiles with few partials (group-128: 16 chunks): one lane per tile.
__global__ __launch_bounds__(kBlock) void backward_finalize_small_kernel(const Partial2* __restrict__ partials, uint32_t units,
                                                                        uint32_t ntiles, float* __restrict__ dscale,
                                                                        float* __restrict__ doffset) {
  const uint32_t t = blockIdx.x * (uint32_t)kBlock + threadIdx.x;
  if (t >= ntiles) return;
  const Partial2* p = partials + (size_t)t * units;
  float ds = 0.0f, dof = 0.0f;
  for (uint32_t u = 0; u < units; ++u) { ds = ds + p[u].ds; dof = dof + p[u].dof; }
  dscale[t] = ds;
  if (doffset) doffset[t] = dof;
}

// Any tiling (strided channels such as PerChannel(-1), N-d tiles): tile-major walk, one block per tile (or one LANE per
// tile when tiles are shorter than a wave), dinput written element by element, the two sums reduced in a fixed order and
// written straight into dscale / doffset — no workspace, no atomics. Latency-bound (an index decomposition per element):
// the coverage path, the streaming kernel above stays the fast one.
template <typename T, bool HAS_OFFSET>
__global__ __launch_bounds__(kBlock) void quantize_backward_tiles_kernel(const T* __restrict__ x, const T* __restrict__ g,
                                                                        T* __restrict__ dinput, const float* __restrict__ scale,
                                                                        const float* __restrict__ offset, float* __restrict__ dscale,
                                                                        float* __restrict__ doffset, TileWalk w, int64_t ntiles,
                                                                        float lo, float hi, uint32_t scale_stride,
                                                                        uint32_t offset_stride, int lane_per_tile) {
  __shared__ Partial2 wave_part[kBlock / 64];
  const int64_t tile = lane_per_tile ? (int64_t)blockIdx.x * kBlock + threadIdx.x : (int64_t)blockIdx.x;
  float ds = 0.0f, dof = 0.0f;
  if (tile < ntiles) {
    const float s = scale[tile * scale_stride];
    const float ro = HAS_OFFSET ? rne(offset[tile * offset_stride]) : 0.0f;
    const int64_t origin = tile_origin(w.g, tile);
    const int64_t first = lane_per_tile ? 0 : threadIdx.x, step = lane_per_tile ? 1 : kBlock;
    for (int64_t e = first; e < w.tile_elems; e += step) {
      const int64_t at = tile_element(w.g, origin, e);
      const float xv = to_f32(x[at]), gv = to_f32(g[at]);
      const float u = xv / s - ro;
      const float q = rne(u);
      const bool below = q < lo, above = q > hi;
      const bool clip = below || above;
      dinput[at] = from_f32<T>(clip ? 0.0f : gv);
      const float bound = (below ? lo : hi) + ro;
      ds = ds + (clip ? bound : q - u) * gv;
      if constexpr (HAS_OFFSET) dof = dof + (clip ? s * gv : 0.0f);
    }
  }
  if (lane_per_tile) {
    if (tile < ntiles) {
      dscale[tile] = ds;
      if (HAS_OFFSET) doffset[tile] = dof;
    }
    return;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    ds = ds + __shfl_xor(ds, d, 64);
    dof = dof + __shfl_xor(dof, d, 64);
  }
  if ((threadIdx.x & 63) == 0) wave_part[threadIdx.x >> 6] = Partial2{ds, dof};
  __syncthreads();
  if (threadIdx.x == 0) {
    Partial2 p = wave_part[0];
#pragma unroll
    for (int v = 1; v < kBlock / 64; ++v) { p.ds = p.ds + wave_part[v].ds; p.dof = p.dof + wave_part[v].dof; }
    dscale[tile] = p.ds;
    if (HAS_OFFSET) doffset[tile] = p.dof;
  }
}

static bool backward_plan(const TileInfo& info, uint32_t* nchunks, int* per_block, uint32_t* units, uint32_t* nparts) {
  if (info.numel % 8 != 0 || info.numel / 8 >= ((int64_t)1 << 32) - kBlock) return false;
  const int64_t chunks = info.numel / 8;
  int64_t cpr;  // chunks per tile
  if (info.layout == LAYOUT_SCALAR) cpr = chunks;
  else if (info.layout == LAYOUT_ROWS && info.run % 8 == 0) cpr = info.run / 8;
  else return false;
  *nchunks = (uint32_t)chunks;
  if (cpr % kBlock == 0) {
    *per_block = 1;
    *units = (uint32_t)(cpr / kBlock);
    *nparts = (uint32_t)(chunks / kBlock);
  } else {
    *per_block = 0;
    *units = (uint32_t)cpr;
    *nparts = (uint32_t)chunks;
  }
  return true;
}

}  // namespace ffq

using namespace ffq;

// small problems of a few tiles take the by-tile kernel: ONE launch, no workspace (see ffq_quantize_by_tile_backward)
static bool backward_small(const TileInfo& info) {
  return info.numel <= ((int64_t)1 << 16) && info.ntiles >= 4 && info.numel / info.ntiles <= 4096;
}

extern "C" size_t ffq_quantize_backward_workspace_bytes(const ffq_tiling* tiling) {
  TileInfo info;
  if (analyse(tiling, &info)) return 0;
  if (backward_small(info)) return 0;
  uint32_t nchunks, units, nparts;
  int per_block;
  if (!backward_plan(info, &nchunks, &per_block, &units, &nparts)) return 0;
  return (size_t)nparts * sizeof(Partial2);
}

extern "C" int ffq_quantize_by_tile_backward(const void* data, const void* output_grad, int dt, const float* scale,
                                             int64_t scale_numel, const float* offset, int64_t offset_numel,
                                             const ffq_tiling* tiling, double num_bits, void* dinput, float* dscale,
                                             float* doffset, void* workspace, size_t workspace_bytes, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  TileInfo info;
  int rc = analyse(tiling, &info);
  if (rc) return rc;
  if (!(dt == FFQ_F32 || dt == FFQ_BF16 || dt == FFQ_F16)) return fail(FFQ_ERR_DTYPE, "backward is built for f32 / bf16 / f16 data");
  if (info.numel != 0) {
    if ((rc = check_param_numel("scale", scale_numel, info.ntiles))) return rc;
    if (offset && (rc = check_param_numel("offset", offset_numel, info.ntiles))) return rc;
  }
  if (info.numel == 0) return FFQ_OK;
  if (!data || !output_grad || !scale || !dinput || !dscale) return fail(FFQ_ERR_ARG, "NULL buffer");
  if (offset && !doffset) return fail(FFQ_ERR_ARG, "doffset is required when offset is given");
  uint32_t nchunks, units, nparts;
  int per_block;
  const double lo_d = -pow(2.0, num_bits - 1.0);
  // small problems of a few tiles (an eager quantizer of a small model: the host pays per launch, bench.py host_us_per_op) take the by-tile
  // kernel as well: ONE launch instead of the streaming pass + its finalize (round 6: 11.8 us per call on the host with two launches)
  const bool small = backward_small(info);
  if (small || !backward_plan(info, &nchunks, &per_block, &units, &nparts) || !aligned16(data) || !aligned16(output_grad) || !aligned16(dinput) ||
      (scale_numel == 1 && info.ntiles != 1)) {
    // strided channels, N-d tiles, odd sizes, broadcast parameters: the by-tile kernel
    const TileWalk w = make_tile_walk(tiling);
    const int lane_per_tile = w.tile_elems < 64;
    const int64_t blocks = lane_per_tile ? (info.ntiles + kBlock - 1) / kBlock : info.ntiles;
    if (blocks >= ((int64_t)1 << 31)) return fail(FFQ_ERR_DTYPE, "too many tiles for the by-tile backward kernel");
    const uint32_t ss = scale_numel == 1 ? 0u : 1u, os = offset_numel == 1 ? 0u : 1u;
#define FFQ_BWD_TILES(T)                                                                                             \
  do {                                                                                                               \
    if (offset) quantize_backward_tiles_kernel<T, true><<<(unsigned)blocks, kBlock, 0, s>>>(static_cast<const T*>(data), static_cast<const T*>(output_grad), static_cast<T*>(dinput), scale, offset, dscale, doffset, w, info.ntiles, (float)lo_d, (float)(-lo_d - 1.0), ss, os, lane_per_tile); \
    else quantize_backward_tiles_kernel<T, false><<<(unsigned)blocks, kBlock, 0, s>>>(static_cast<const T*>(data), static_cast<const T*>(output_grad), static_cast<T*>(dinput), scale, offset, dscale, doffset, w, info.ntiles, (float)lo_d, (float)(-lo_d - 1.0), ss, os, lane_per_tile); \
  } while (0)
    switch (dt) {
      case FFQ_F32: FFQ_BWD_TILES(float); break;
      case FFQ_BF16: FFQ_BWD_TILES(bf16_t); break;
      default: FFQ_BWD_TILES(f16_t); break;
    }
#undef FFQ_BWD_TILES
    return check_launch("quantize_backward_tiles_kernel");
  }
  const size_t need = (size_t)nparts * sizeof(Partial2);
  if (!workspace || workspace_bytes < need) return fail(FFQ_ERR_WORKSPACE, "backward needs %zu workspace bytes, got %zu", need, workspace_bytes);
  BwdArgs a;
  const double lo = -pow(2.0, num_bits - 1.0);
  a.lo = (float)lo; a.hi = (float)(-lo - 1.0);
  a.nchunks = nchunks;
  a.scale_stride = scale_numel == 1 ? 0u : 1u;
  a.offset_stride = offset_numel == 1 ? 0u : 1u;
  a.rows = info.layout == LAYOUT_ROWS;
  a.chunks_per_run = make_fastdiv(a.rows ? (uint32_t)(info.run / 8) : 1u);
  a.per_block = per_block;
  Partial2* parts = static_cast<Partial2*>(workspace);
  const unsigned grid = (nchunks + kBlock - 1) / kBlock;
#define FFQ_BWD(T)                                                                                                   \
  do {                                                                                                               \
    if (offset) quantize_backward_kernel<T, true><<<grid, kBlock, 0, s>>>(static_cast<const T*>(data), static_cast<const T*>(output_grad), static_cast<T*>(dinput), scale, offset, parts, a); \
    else quantize_backward_kernel<T, false><<<grid, kBlock, 0, s>>>(static_cast<const T*>(data), static_cast<const T*>(output_grad), static_cast<T*>(dinput), scale, offset, parts, a); \
  } while (0)
  switch (dt) {
    case FFQ_F32: FFQ_BWD(float); break;
    case FFQ_BF16: FFQ_BWD(bf16_t); break;
    default: FFQ_BWD(f16_t); break;
  }
#undef FFQ_BWD
  if ((rc = check_launch("quantize_backward_kernel"))) return rc;
  const uint32_t ntiles = (uint32_t)info.ntiles;
  if (units <= 32)
    backward_finalize_small_kernel<<<(ntiles + kBlock - 1) / kBlock, kBlock, 0, s>>>(parts, units, ntiles, dscale, offset ? doffset : nullptr);
  else
    backward_finalize_kernel<<<ntiles, kBlock, 0, s>>>(parts, units, dscale, offset ? doffset : nullptr);
  return check_launch("backward_finalize_kernel");
}

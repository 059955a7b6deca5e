// ffq_griderror.hip — the inner loop of the min-error (MSE grid) range estimator on gfx950.
//
// Reference: _MinAvgErrorGridEstimator.estimate_step, src/fastforward/range_setting/min_error.py:218-231 —
// for each of `num_candidates` (default 100) candidate ranges: quantize the batch, dequantize it, and take
// the mean squared difference per tile: 100 x (A1 + A2 + sub + pow + mean) = ~1300 ATen passes per quantizer
// per step. Here the batch is read ONCE: every lane keeps its 8 elements in registers and evaluates all
// candidates on them (16 accumulators at a time), so the kernel is VALU-bound (~18 ops per element and
// candidate) instead of HBM-bound; per-tile sums are reduced in a fixed order (no floating-point atomics).
// Every step of the eager chain keeps its rounding: quantize and dequantize in fp32, the dequantized value,
// the difference and the square each rounded to the data dtype, the sum in fp32 (ATen's mean accumulates
// half-precision inputs in fp32).
#include "ffq_affine.h"
#include "ffq_common.h"
#include "ffq_vec.h"

#include <math.h>

namespace ffq {

constexpr int kCandBatch = 16;

struct GridArgs {
  float lo, hi;
  uint32_t nchunks;
  uint32_t ntiles, ncand;
  FastDiv chunks_per_run;  // ROWS: run / 8
  int rows;                // 1: ROWS layout, 0: one tile
  int mode;                // 0: one partial row per block (block inside one tile); 1: sub-wave groups of `group` lanes
  uint32_t group;          // lanes per tile in mode 1 (power of two <= 64)
  int accumulate;          // mode 1 writes err directly: += or =
};

template <typename T>
__device__ __forceinline__ float round_to(float v) {
  if constexpr (sizeof(T) == 4) return v;
  else return to_f32(from_f32<T>(v));
}

template <typename T, bool HAS_OFFSET>
__global__ __launch_bounds__(kBlock) void grid_sqerror_kernel(const T* __restrict__ x, const float* __restrict__ scales,
                                                             const float* __restrict__ offsets, float* __restrict__ partials,
                                                             float* __restrict__ err, GridArgs a) {
  constexpr int E = 8;
  __shared__ float wave_part[kBlock / 64][kCandBatch];
  const uint32_t c = blockIdx.x * (uint32_t)kBlock + threadIdx.x;
  const bool active = c < a.nchunks;
  float xv[E];
  uint32_t t = 0;
  if (active) {
    Chunk<T, E> cx;
    cx.load(x + (size_t)c * E);
#pragma unroll
    for (int i = 0; i < E; ++i) xv[i] = cx.get(i);
    t = a.rows ? fdiv(c, a.chunks_per_run) : 0u;
  } else {
#pragma unroll
    for (int i = 0; i < E; ++i) xv[i] = 0.0f;
  }
  for (uint32_t cb = 0; cb < a.ncand; cb += kCandBatch) {
    float acc[kCandBatch];
#pragma unroll
    for (int k = 0; k < kCandBatch; ++k) {
      acc[k] = 0.0f;
      const uint32_t cand = cb + k;
      if (cand < a.ncand && active) {  // cand < ncand is wave-uniform
        const float s = scales[(size_t)cand * a.ntiles + t];
        const float ro = HAS_OFFSET ? rne(offsets[(size_t)cand * a.ntiles + t]) : 0.0f;
        float r[E];
        quantize_chunk<1, E>(xv, s, ro, r);
        float sum = 0.0f;
#pragma unroll
        for (int i = 0; i < E; ++i) {
          const float q = r[i] != r[i] ? r[i] : __builtin_amdgcn_fmed3f(r[i], a.lo, a.hi);
          const float y = round_to<T>((q + ro) * s);          // dequantize, cast to the data dtype
          const float d = round_to<T>(y - xv[i]);             // quantized - original
          sum = sum + round_to<T>(d * d);                     // ** 2
        }
        acc[k] = sum;
      }
    }
    if (a.mode == 1) {
      // tiles of `group` lanes: butterfly inside the group, its first lane owns err[cand][tile]
      for (uint32_t d = a.group >> 1; d >= 1; d >>= 1) {
#pragma unroll
        for (int k = 0; k < kCandBatch; ++k) acc[k] = acc[k] + __shfl_xor(acc[k], (int)d, 64);
      }
      if (active && (threadIdx.x & (a.group - 1)) == 0) {
#pragma unroll
        for (int k = 0; k < kCandBatch; ++k) {
          const uint32_t cand = cb + k;
          if (cand < a.ncand) {
            float* dst = err + (size_t)cand * a.ntiles + t;
            *dst = a.accumulate ? *dst + acc[k] : acc[k];
          }
        }
      }
    } else {
#pragma unroll
      for (int k = 0; k < kCandBatch; ++k) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) acc[k] = acc[k] + __shfl_xor(acc[k], d, 64);
      }
      __syncthreads();  // wave_part is reused by every candidate batch
      if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < kCandBatch; ++k) wave_part[threadIdx.x >> 6][k] = acc[k];
      }
      __syncthreads();
      if (threadIdx.x < kCandBatch && cb + threadIdx.x < a.ncand) {
        const float v = ((wave_part[0][threadIdx.x] + wave_part[1][threadIdx.x]) + wave_part[2][threadIdx.x]) + wave_part[3][threadIdx.x];
        partials[(size_t)blockIdx.x * a.ncand + cb + threadIdx.x] = v;
      }
    }
  }
}

// err[cand][tile] (+)= sum over the tile's `units` consecutive blocks of partials[block][cand]; one wave per (tile, cand).
__global__ __launch_bounds__(kBlock) void grid_finalize_kernel(const float* __restrict__ partials, uint32_t units, uint32_t ntiles,
                                                              uint32_t ncand, float* __restrict__ err, int accumulate) {
  const uint32_t item = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (item >= ntiles * ncand) return;
  const uint32_t tile = item / ncand, cand = item - tile * ncand;
  const float* p = partials + (size_t)tile * units * ncand + cand;
  float s = 0.0f;
  for (uint32_t u = threadIdx.x & 63u; u < units; u += 64) s = s + p[(size_t)u * ncand];
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) s = s + __shfl_xor(s, d, 64);
  if ((threadIdx.x & 63u) == 0) {
    float* dst = err + (size_t)cand * ntiles + tile;
    *dst = accumulate ? *dst + s : s;
  }
}

// Any tiling: one block per tile (one lane per tile below a wave's worth of elements) walks the tile once per batch of 16
// candidates; fixed-order sums written straight to err[cand][tile]. The coverage path for strided channels / N-d tiles.
template <typename T, bool HAS_OFFSET>
__global__ __launch_bounds__(kBlock) void grid_sqerror_tiles_kernel(const T* __restrict__ x, const float* __restrict__ scales,
                                                                   const float* __restrict__ offsets, float* __restrict__ err,
                                                                   TileWalk w, uint32_t ntiles, uint32_t ncand, float lo, float hi,
                                                                   int accumulate, int lane_per_tile) {
  __shared__ float wave_part[kBlock / 64][kCandBatch];
  const int64_t tile = lane_per_tile ? (int64_t)blockIdx.x * kBlock + threadIdx.x : (int64_t)blockIdx.x;
  const bool live = tile < (int64_t)ntiles;
  const int64_t origin = live ? tile_origin(w.g, tile) : 0;
  const int64_t first = lane_per_tile ? 0 : threadIdx.x, step = lane_per_tile ? 1 : kBlock;
  for (uint32_t cb = 0; cb < ncand; cb += kCandBatch) {
    float acc[kCandBatch], sc[kCandBatch], ro[kCandBatch];
#pragma unroll
    for (int k = 0; k < kCandBatch; ++k) {
      acc[k] = 0.0f;
      const bool on = live && cb + k < ncand;
      sc[k] = on ? scales[(size_t)(cb + k) * ntiles + tile] : 1.0f;
      ro[k] = (on && HAS_OFFSET) ? rne(offsets[(size_t)(cb + k) * ntiles + tile]) : 0.0f;
    }
    if (live) {
      for (int64_t e = first; e < w.tile_elems; e += step) {
        const float xv = to_f32(x[tile_element(w.g, origin, e)]);
#pragma unroll
        for (int k = 0; k < kCandBatch; ++k) {
          const float r = rne(xv / sc[k] - ro[k]);
          const float q = r != r ? r : __builtin_amdgcn_fmed3f(r, lo, hi);
          const float y = round_to<T>((q + ro[k]) * sc[k]);
          const float d = round_to<T>(y - xv);
          acc[k] = acc[k] + round_to<T>(d * d);
        }
      }
    }
    if (lane_per_tile) {
      if (live) {
#pragma unroll
        for (int k = 0; k < kCandBatch; ++k)
          if (cb + k < ncand) {
            float* dst = err + (size_t)(cb + k) * ntiles + tile;
            *dst = accumulate ? *dst + acc[k] : acc[k];
          }
      }
      continue;
    }
#pragma unroll
    for (int k = 0; k < kCandBatch; ++k) {
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) acc[k] = acc[k] + __shfl_xor(acc[k], d, 64);
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
      for (int k = 0; k < kCandBatch; ++k) wave_part[threadIdx.x >> 6][k] = acc[k];
    }
    __syncthreads();
    if (threadIdx.x < kCandBatch && cb + threadIdx.x < ncand) {
      const float v = ((wave_part[0][threadIdx.x] + wave_part[1][threadIdx.x]) + wave_part[2][threadIdx.x]) + wave_part[3][threadIdx.x];
      float* dst = err + (size_t)(cb + threadIdx.x) * ntiles + tile;
      *dst = accumulate ? *dst + v : v;
    }
  }
}

static bool grid_plan(const TileInfo& info, GridArgs* a, uint32_t* units, uint32_t* nblocks) {
  if (info.numel % 8 != 0 || info.numel / 8 >= ((int64_t)1 << 32) - kBlock) return false;
  const int64_t chunks = info.numel / 8;
  int64_t cpr;
  if (info.layout == LAYOUT_SCALAR) cpr = chunks;
  else if (info.layout == LAYOUT_ROWS && info.run % 8 == 0) cpr = info.run / 8;
  else return false;
  a->nchunks = (uint32_t)chunks;
  a->rows = info.layout == LAYOUT_ROWS;
  a->chunks_per_run = make_fastdiv(a->rows ? (uint32_t)cpr : 1u);
  *nblocks = (uint32_t)((chunks + kBlock - 1) / kBlock);
  if (cpr % kBlock == 0) {
    a->mode = 0;
    a->group = 0;
    *units = (uint32_t)(cpr / kBlock);
    return true;
  }
  if (cpr <= 64 && (cpr & (cpr - 1)) == 0) {
    a->mode = 1;
    a->group = (uint32_t)cpr;
    *units = 0;
    return true;
  }
  return false;
}

}  // namespace ffq

using namespace ffq;

extern "C" size_t ffq_grid_sqerror_workspace_bytes(const ffq_tiling* tiling, int64_t ncand) {
  TileInfo info;
  if (analyse(tiling, &info) || ncand <= 0) return 0;
  GridArgs a;
  uint32_t units, nblocks;
  if (!grid_plan(info, &a, &units, &nblocks) || a.mode != 0) return 0;
  return (size_t)nblocks * (size_t)ncand * sizeof(float);
}

extern "C" int ffq_grid_sqerror_by_tile(const void* data, int dt, const float* scales, const float* offsets, int64_t ncand,
                                        const ffq_tiling* tiling, double num_bits, float* err, int accumulate, void* workspace,
                                        size_t workspace_bytes, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  TileInfo info;
  int rc = analyse(tiling, &info);
  if (rc) return rc;
  if (ncand <= 0 || ncand > 4096) return fail(FFQ_ERR_ARG, "number of candidates must be 1..4096");
  if (!(dt == FFQ_F32 || dt == FFQ_BF16 || dt == FFQ_F16)) return fail(FFQ_ERR_DTYPE, "grid error is built for f32 / bf16 / f16 data");
  if (info.numel == 0) return fail(FFQ_ERR_EMPTY, "grid error over an empty tensor");
  if (!data || !scales || !err) return fail(FFQ_ERR_ARG, "NULL buffer");
  GridArgs a;
  uint32_t units, nblocks;
  const double lo = -pow(2.0, num_bits - 1.0);
  if (info.ntiles * ncand >= ((int64_t)1 << 31)) return fail(FFQ_ERR_DTYPE, "too many (tile, candidate) pairs");
  if (!grid_plan(info, &a, &units, &nblocks) || !aligned16(data)) {
    // strided channels, N-d tiles, run lengths the streaming kernel does not cover: the by-tile kernel
    const TileWalk w = make_tile_walk(tiling);
    const int lane_per_tile = w.tile_elems < 64;
    const int64_t blocks = lane_per_tile ? (info.ntiles + kBlock - 1) / kBlock : info.ntiles;
    if (blocks >= ((int64_t)1 << 31)) return fail(FFQ_ERR_DTYPE, "too many tiles for the by-tile grid kernel");
#define FFQ_GRID_TILES(T)                                                                                                 \
  do {                                                                                                                    \
    if (offsets) grid_sqerror_tiles_kernel<T, true><<<(unsigned)blocks, kBlock, 0, s>>>(static_cast<const T*>(data), scales, offsets, err, w, (uint32_t)info.ntiles, (uint32_t)ncand, (float)lo, (float)(-lo - 1.0), accumulate, lane_per_tile); \
    else grid_sqerror_tiles_kernel<T, false><<<(unsigned)blocks, kBlock, 0, s>>>(static_cast<const T*>(data), scales, offsets, err, w, (uint32_t)info.ntiles, (uint32_t)ncand, (float)lo, (float)(-lo - 1.0), accumulate, lane_per_tile); \
  } while (0)
    switch (dt) {
      case FFQ_F32: FFQ_GRID_TILES(float); break;
      case FFQ_BF16: FFQ_GRID_TILES(bf16_t); break;
      default: FFQ_GRID_TILES(f16_t); break;
    }
#undef FFQ_GRID_TILES
    return check_launch("grid_sqerror_tiles_kernel");
  }
  a.lo = (float)lo; a.hi = (float)(-lo - 1.0);
  a.ntiles = (uint32_t)info.ntiles;
  a.ncand = (uint32_t)ncand;
  a.accumulate = accumulate;
  float* parts = static_cast<float*>(workspace);
  if (a.mode == 0) {
    const size_t need = (size_t)nblocks * (size_t)ncand * sizeof(float);
    if (!workspace || workspace_bytes < need) return fail(FFQ_ERR_WORKSPACE, "grid error needs %zu workspace bytes, got %zu", need, workspace_bytes);
  }
#define FFQ_GRID(T)                                                                                                       \
  do {                                                                                                                    \
    if (offsets) grid_sqerror_kernel<T, true><<<nblocks, kBlock, 0, s>>>(static_cast<const T*>(data), scales, offsets, parts, err, a); \
    else grid_sqerror_kernel<T, false><<<nblocks, kBlock, 0, s>>>(static_cast<const T*>(data), scales, offsets, parts, err, a);        \
  } while (0)
  switch (dt) {
    case FFQ_F32: FFQ_GRID(float); break;
    case FFQ_BF16: FFQ_GRID(bf16_t); break;
    default: FFQ_GRID(f16_t); break;
  }
#undef FFQ_GRID
  if ((rc = check_launch("grid_sqerror_kernel"))) return rc;
  if (a.mode == 0) {
    const uint32_t items = a.ntiles * a.ncand;
    grid_finalize_kernel<<<(items + 3) / 4, kBlock, 0, s>>>(parts, units, a.ntiles, a.ncand, err, accumulate);
    return check_launch("grid_finalize_kernel");
  }
  return FFQ_OK;
}

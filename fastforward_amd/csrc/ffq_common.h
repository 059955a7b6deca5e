// ffq_common.h — shared host/device helpers of libffq_hip.so (gfx950 only).
//
// Numerics contract (see include/ffq.h and DESIGN.md §3): every arithmetic step of the reference's
// eager chain is one correctly rounded IEEE operation in the promoted dtype. The library is
// compiled with -ffp-contract=off and without fast-math, so `a / b - c` below is a correctly
// rounded divide followed by a correctly rounded subtract — never an FMA, never a reciprocal.
#pragma once

#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/ffq.h"

// ---------------------------------------------------------------------------------------------
// Cross-block hand-overs (split-K slabs of ffq_wlinear.hip / ffq_wmid.hip / ffq_wskinny.hip, the one-launch min/max of
// ffq_minmax.hip, the extrema words of ffq_extrema.h): a block publishes data with sc1 WRITE-THROUGH stores, drains them
// (`s_waitcnt vmcnt(0)`), takes a ticket with an agent-scope read-modify-write, and the block whose ticket says "everybody has
// published" reads its peers' data with sc1 loads (which bypass the reading CU's L1 and cannot hit its L2: a slab line is written
// once and read once per launch). The ticket RMW itself is RELAXED. Round 6 measured the alternative the HIP memory model offers —
// ACQ_REL on the RMW, i.e. `buffer_wbl2 sc1` (write back the XCD's whole L2) ahead of every ticket and `buffer_inv sc1` (invalidate
// it) behind — as two builds on one box (profiles/r06_ticket_order_ab.txt): the skinny weight-only GEMM at 64 rows 20.4 -> 46.2 us
// (q/o), 29.7 -> 138.0 us (gate/up), at 128 rows 29.2 -> 53.9 us, the 256-row tiles' exchange at 512 rows 44 -> 50 us, the per-tensor
// min/max of a 470 MB activation 76 -> 88 us: an L2 flush per arriving wave in kernels whose other blocks are mid-stream. So the
// relaxed form stays, and what it rests on is pinned by a CPU test on the emitted ISA (tests/test_ticket_isa.py): every slab store
// and load carries sc1, and `s_waitcnt vmcnt(0)` stands between a block's last slab store and its ticket RMW.
// -DFFQ_TICKET_ORDER=__ATOMIC_ACQ_REL builds the in-model form (tools/build_variant_multi.sh) for whoever wants to re-measure.
// ---------------------------------------------------------------------------------------------
#ifndef FFQ_TICKET_ORDER
#define FFQ_TICKET_ORDER __ATOMIC_RELAXED
#endif
// acquire behind a relaxed polling loop (one fence when the loop leaves, none per poll)
#define ffq_ticket_acquire()                                                      \
  do {                                                                            \
    if (FFQ_TICKET_ORDER != __ATOMIC_RELAXED) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); \
  } while (0)

namespace ffq {

// ---------------------------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------------------------
char* err_buf();
int fail(int code, const char* fmt, ...);
// ffq_force_generic_kernels (include/ffq.h): the one-element-per-lane kernels instead of the streaming ones, for tests that
// compare the two families. The library reads NO environment variables (tuning knobs exist only under -DFFQ_EXPERIMENTS).
bool generic_kernels_forced();
bool mid_register_form_forced();  // test hook (bit 2): the register-staged 128-column-tile kernel instead of the LDS-DMA one (ffq_wmid.hip)
bool splitk_abandon_forced();  // test hook (bit 1 of ffq_force_generic_kernels): odd K slices of a split tile abandon their wait at once
int check_launch(const char* what);

// ---------------------------------------------------------------------------------------------
// dtype tags
// ---------------------------------------------------------------------------------------------
__host__ __device__ inline bool dt_valid(int dt) { return dt >= FFQ_F32 && dt <= FFQ_U8; }
__host__ __device__ inline bool dt_is_float(int dt) {
  return dt == FFQ_F32 || dt == FFQ_BF16 || dt == FFQ_F16 || dt == FFQ_F64;
}
__host__ __device__ inline int dt_size(int dt) {
  switch (dt) {
    case FFQ_F32: case FFQ_I32: return 4;
    case FFQ_BF16: case FFQ_F16: case FFQ_I16: return 2;
    case FFQ_F64: case FFQ_I64: return 8;
    default: return 1;
  }
}

// ---------------------------------------------------------------------------------------------
// scalar conversions (device + host). bf16 is handled as raw uint16 bits; f16 through _Float16.
// ---------------------------------------------------------------------------------------------
__host__ __device__ inline float bf16_bits_to_f32(uint16_t h) {
  return __builtin_bit_cast(float, (uint32_t)h << 16);
}
// round-to-nearest-even, NaN stays a quiet NaN (matches c10::BFloat16)
__host__ __device__ inline uint16_t f32_to_bf16_bits(float f) {
  uint32_t u = __builtin_bit_cast(uint32_t, f);
  if (f != f) return 0x7FC0;
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
__host__ __device__ inline float f16_bits_to_f32(uint16_t h) {
  return (float)__builtin_bit_cast(_Float16, h);
}
__host__ __device__ inline uint16_t f32_to_f16_bits(float f) {
  return __builtin_bit_cast(uint16_t, (_Float16)f);  // v_cvt_f16_f32: RNE, overflow -> inf
}

// Value a tensor of dtype `dt` holds after an op whose exact result (in float opmath) is `v`.
__host__ __device__ inline float round_stage(float v, int dt) {
  if (dt == FFQ_BF16) return bf16_bits_to_f32(f32_to_bf16_bits(v));
  if (dt == FFQ_F16) return f16_bits_to_f32(f32_to_f16_bits(v));
  return v;
}

// torch.round == rint in round-to-nearest-even mode.
__host__ __device__ inline float rne(float v) { return __builtin_rintf(v); }
__host__ __device__ inline double rne(double v) { return __builtin_rint(v); }

// torch.clamp(x, lo, hi): NaN propagates.
__host__ __device__ inline float clamp_nan(float v, float lo, float hi) {
  float c = v < lo ? lo : v;
  c = c > hi ? hi : c;
  return v != v ? v : c;
}
__host__ __device__ inline double clamp_nan(double v, double lo, double hi) {
  double c = v < lo ? lo : v;
  c = c > hi ? hi : c;
  return v != v ? v : c;
}

// Runtime-typed element access, used by the generic kernels and for parameter tables.
__device__ inline double load_any(const void* p, int dt, int64_t i) {
  switch (dt) {
    case FFQ_F32: return (double)((const float*)p)[i];
    case FFQ_BF16: return (double)bf16_bits_to_f32(((const uint16_t*)p)[i]);
    case FFQ_F16: return (double)f16_bits_to_f32(((const uint16_t*)p)[i]);
    case FFQ_F64: return ((const double*)p)[i];
    case FFQ_I8: return (double)((const int8_t*)p)[i];
    case FFQ_I16: return (double)((const int16_t*)p)[i];
    case FFQ_I32: return (double)((const int32_t*)p)[i];
    case FFQ_I64: return (double)((const int64_t*)p)[i];
    default: return (double)((const uint8_t*)p)[i];
  }
}
// NaN -> integer follows the x86 conversion of the reference's CPU path (INT_MIN, low bits for
// narrower types), see oracle/ffq_oracle.c st().
__device__ inline void store_any(void* p, int dt, int64_t i, double v) {
  const bool nan = v != v;
  switch (dt) {
    case FFQ_F32: ((float*)p)[i] = (float)v; break;
    case FFQ_BF16: ((uint16_t*)p)[i] = f32_to_bf16_bits((float)v); break;
    case FFQ_F16: ((uint16_t*)p)[i] = f32_to_f16_bits((float)v); break;
    case FFQ_F64: ((double*)p)[i] = v; break;
    case FFQ_I8: ((int8_t*)p)[i] = nan ? 0 : (int8_t)(int64_t)v; break;
    case FFQ_I16: ((int16_t*)p)[i] = nan ? 0 : (int16_t)(int64_t)v; break;
    case FFQ_I32: ((int32_t*)p)[i] = nan ? INT32_MIN : (int32_t)(int64_t)v; break;
    case FFQ_I64: ((int64_t*)p)[i] = nan ? INT64_MIN : (int64_t)v; break;
    default: ((uint8_t*)p)[i] = nan ? 0 : (uint8_t)(int64_t)v; break;
  }
}

// ---------------------------------------------------------------------------------------------
// exact unsigned 32-bit division by a launch-invariant divisor (no integer divider on CDNA):
// q = (mulhi(n, mul) + n) >> shift evaluated in 64 bits, the classic round-up magic number.
// ---------------------------------------------------------------------------------------------
struct FastDiv {
  uint32_t mul;
  uint32_t shift;
  uint32_t div;
};
inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f;
  f.div = d;
  if (d == 1) { f.mul = 0; f.shift = 0; return f; }
  uint32_t l = 32 - (uint32_t)__builtin_clz(d - 1);  // ceil(log2 d)
  uint64_t m = (((uint64_t)1 << 32) * (((uint64_t)1 << l) - d)) / d + 1;
  f.mul = (uint32_t)m;
  f.shift = l;
  return f;
}
__device__ inline uint32_t fdiv(uint32_t n, const FastDiv& f) {
  uint64_t t = (uint64_t)__umulhi(n, f.mul) + n;
  return (uint32_t)(t >> f.shift);
}

// ---------------------------------------------------------------------------------------------
// host-side tiling analysis
// ---------------------------------------------------------------------------------------------
enum Layout {
  LAYOUT_SCALAR = 0,   // one tile: parameters are wave-uniform
  LAYOUT_ROWS = 1,     // every tile is a contiguous run of `run` elements: tile = flat / run
  LAYOUT_CHANNEL = 2,  // tile = (flat / inner) % channels  (one strided channel dimension)
  LAYOUT_GENERIC = 3   // N-d tile grid, resolved per element
};

struct TileInfo {
  int layout;
  int64_t numel;
  int64_t ntiles;
  int64_t run;       // ROWS
  int64_t inner;     // CHANNEL
  int64_t channels;  // CHANNEL
};

int check_tiling(const ffq_tiling* t);
int analyse(const ffq_tiling* t, TileInfo* info);
int check_param_numel(const char* what, int64_t numel, int64_t ntiles);

// Generic N-d tile lookup descriptor passed by value to the generic kernels.
struct GenericTiling {
  int32_t ndim;
  int64_t shape[FFQ_MAX_DIMS];
  int64_t tile[FFQ_MAX_DIMS];
  int64_t gstride[FFQ_MAX_DIMS];
};
GenericTiling make_generic(const ffq_tiling* t);

__device__ inline int64_t generic_tile_of(const GenericTiling& g, int64_t flat) {
  int64_t tile = 0;
  for (int k = g.ndim - 1; k >= 0; --k) {
    const int64_t extent = g.shape[k];
    const int64_t idx = flat % extent;
    flat /= extent;
    tile += (idx / g.tile[k]) * g.gstride[k];
  }
  return tile;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per device: "done" flags are bit masks indexed by the current device,
// so a process that launches on several GPUs raises the attribute on each of them. The bit is set AFTER the attribute call
// returned (a second host thread that finds it clear repeats the call, which is harmless; one that finds it set may launch);
// devices beyond the mask's 64 bits are not cached at all.
inline void ensure_dynamic_lds(uint64_t* mask, const void* kernel, int bytes) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const uint64_t bit = (dev >= 0 && dev < 64) ? (uint64_t)1 << dev : 0;
  if (bit && (__atomic_load_n(mask, __ATOMIC_ACQUIRE) & bit)) return;
  (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (bit) __atomic_fetch_or(mask, bit, __ATOMIC_RELEASE);
}

// The inverse walk: flat offset of element `e` (row-major inside the tile) of tile `tile` (row-major over the tile grid),
// i.e. element (tile, e) of the reference's tiles_to_rows view (quantization/tiled_tensor.py:71-98) without materialising
// it. Used by the by-tile kernels that cover ANY tiling (strided channels, N-d tiles): one block or one lane per tile.
struct TileWalk {
  GenericTiling g;
  int64_t tile_elems;
};
inline TileWalk make_tile_walk(const ffq_tiling* t) {
  TileWalk w;
  w.g = make_generic(t);
  w.tile_elems = 1;
  for (int k = 0; k < t->ndim; ++k) w.tile_elems *= t->tile[k];
  return w;
}
__device__ inline int64_t tile_origin(const GenericTiling& g, int64_t tile) {  // flat offset of the tile's first element
  int64_t flat = 0, stride = 1;
  for (int k = g.ndim - 1; k >= 0; --k) {
    const int64_t grid = g.shape[k] / g.tile[k];
    const int64_t tc = (tile / g.gstride[k]) % grid;
    flat += tc * g.tile[k] * stride;
    stride *= g.shape[k];
  }
  return flat;
}
__device__ inline int64_t tile_element(const GenericTiling& g, int64_t origin, int64_t e) {
  int64_t flat = origin, stride = 1;
  for (int k = g.ndim - 1; k >= 0; --k) {
    const int64_t ec = e % g.tile[k];
    e /= g.tile[k];
    flat += ec * stride;
    stride *= g.shape[k];
  }
  return flat;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

constexpr int kBlock = 256;       // 4 waves of 64 lanes
constexpr int kMaxGridBlocks = 1 << 20;

}  // namespace ffq

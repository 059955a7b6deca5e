// ffq_vec.h — 16-byte-per-lane global memory access for the streaming kernels (gfx950).
//
// A wave64 issuing one 16 B load per lane moves 1 KiB per instruction, the coalescing sweet spot
// on CDNA4 (cdna_hip_programming.md §2 / Guideline 13). hipcc does not vectorise bf16/fp16/int8
// element loops by itself, so every streaming kernel here reads and writes through these helpers.
#pragma once

#include "ffq_common.h"

// Streaming tensors (read once / written once per launch) of the producer, W4 and backward kernels carry non-temporal
// hints where an A/B on the MI355X showed a gain (tools/arith_ab.py, profiles/r02_nt_ab.txt; [14336, 4096] bf16):
//   backward: nt loads 5.44 -> 6.02 TB/s (nt stores on top: 5.9);  W4 quantize+pack: nt loads + stores 4.5 -> 5.4;
//   W4 unpack+dequantize: both 5.59 -> 5.75;  add+RMSNorm+quantize: nt loads 5.47 -> 5.62 (stores: no gain);  SiLU*up: none.
// A translation unit picks its policy by defining FFQ_NT_STREAMS (bit 0: loads, bit 1: stores) before including this header.
#ifndef FFQ_NT_STREAMS
#define FFQ_NT_STREAMS 0
#endif
#if FFQ_NT_STREAMS & 1
#define FFQ_SLOAD load_nt
#else
#define FFQ_SLOAD load
#endif
#if FFQ_NT_STREAMS & 2
#define FFQ_SSTORE store_nt
#else
#define FFQ_SSTORE store
#endif

namespace ffq {

struct bf16_t { uint16_t bits; };
struct f16_t { uint16_t bits; };

template <typename T> struct TypeTag;
template <> struct TypeTag<float> { static constexpr int value = FFQ_F32; };
template <> struct TypeTag<bf16_t> { static constexpr int value = FFQ_BF16; };
template <> struct TypeTag<f16_t> { static constexpr int value = FFQ_F16; };
template <> struct TypeTag<int8_t> { static constexpr int value = FFQ_I8; };
template <> struct TypeTag<int16_t> { static constexpr int value = FFQ_I16; };
template <> struct TypeTag<int32_t> { static constexpr int value = FFQ_I32; };

__device__ inline float to_f32(float v) { return v; }
__device__ inline float to_f32(bf16_t v) { return bf16_bits_to_f32(v.bits); }
__device__ inline float to_f32(f16_t v) { return f16_bits_to_f32(v.bits); }
__device__ inline float to_f32(int8_t v) { return (float)v; }
__device__ inline float to_f32(int16_t v) { return (float)v; }
__device__ inline float to_f32(int32_t v) { return (float)v; }  // RNE, as ATen's int -> float copy

template <typename T> __device__ inline T from_f32(float v);
template <> __device__ inline float from_f32<float>(float v) { return v; }
template <> __device__ inline bf16_t from_f32<bf16_t>(float v) { return bf16_t{f32_to_bf16_bits(v)}; }
template <> __device__ inline f16_t from_f32<f16_t>(float v) { return f16_t{f32_to_f16_bits(v)}; }
// Integer containers receive integer-valued, in-range numbers (or NaN, which follows the x86
// conversion the reference's CPU path executes: INT_MIN, and its low bits for narrower types).
template <> __device__ inline int8_t from_f32<int8_t>(float v) { return v != v ? (int8_t)0 : (int8_t)(int)v; }
template <> __device__ inline int16_t from_f32<int16_t>(float v) { return v != v ? (int16_t)0 : (int16_t)(int)v; }
template <> __device__ inline int32_t from_f32<int32_t>(float v) { return v != v ? INT32_MIN : (int32_t)v; }

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// Two floats -> one dword of two 16-bit values (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32: RNE).
template <typename T> __device__ inline uint32_t pack2(float a, float b);
template <> __device__ inline uint32_t pack2<bf16_t>(float a, float b) {
  f32x2 v; v.x = a; v.y = b;
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
template <> __device__ inline uint32_t pack2<f16_t>(float a, float b) {
  f32x2 v; v.x = a; v.y = b;
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
}
template <> __device__ inline uint32_t pack2<int16_t>(float a, float b) {
  return ((uint32_t)(uint16_t)from_f32<int16_t>(a)) | ((uint32_t)(uint16_t)from_f32<int16_t>(b) << 16);
}
__device__ inline uint32_t pack_bytes(int a, int b, int c, int d) {
  return ((uint32_t)a & 0xFFu) | (((uint32_t)b & 0xFFu) << 8) | (((uint32_t)c & 0xFFu) << 16) | ((uint32_t)d << 24);
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// E consecutive elements of T, held as raw dwords. E * sizeof(T) is 4, 8, 16, 32 or 64 bytes.
template <typename T, int E>
struct Chunk {
  static constexpr int kBytes = E * (int)sizeof(T);
  static_assert(kBytes == 4 || kBytes == 8 || kBytes % 16 == 0, "unsupported chunk size");
  static constexpr int kWords = kBytes / 4;
  uint32_t w[kWords];

  __device__ inline void load(const T* p) {
    if constexpr (kBytes == 4) {
      w[0] = *reinterpret_cast<const uint32_t*>(p);
    } else if constexpr (kBytes == 8) {
      u32x2 v = *reinterpret_cast<const u32x2*>(p);
      w[0] = v.x; w[1] = v.y;
    } else {
#pragma unroll
      for (int i = 0; i < kBytes / 16; ++i) {
        u32x4 v = reinterpret_cast<const u32x4*>(p)[i];
        w[4 * i + 0] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
      }
    }
  }
  __device__ inline void load_nt(const T* p) {
    if constexpr (kBytes == 4) {
      w[0] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(p));
    } else if constexpr (kBytes == 8) {
      u32x2 v = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(p));
      w[0] = v.x; w[1] = v.y;
    } else {
#pragma unroll
      for (int i = 0; i < kBytes / 16; ++i) {
        u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p) + i);
        w[4 * i + 0] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
      }
    }
  }
  __device__ inline void store(T* p) const {
    if constexpr (kBytes == 4) {
      *reinterpret_cast<uint32_t*>(p) = w[0];
    } else if constexpr (kBytes == 8) {
      u32x2 v; v.x = w[0]; v.y = w[1];
      *reinterpret_cast<u32x2*>(p) = v;
    } else {
#pragma unroll
      for (int i = 0; i < kBytes / 16; ++i) {
        u32x4 v; v.x = w[4 * i + 0]; v.y = w[4 * i + 1]; v.z = w[4 * i + 2]; v.w = w[4 * i + 3];
        reinterpret_cast<u32x4*>(p)[i] = v;
      }
    }
  }
  __device__ inline void store_nt(T* p) const {
    if constexpr (kBytes == 4) {
      __builtin_nontemporal_store(w[0], reinterpret_cast<uint32_t*>(p));
    } else if constexpr (kBytes == 8) {
      u32x2 v; v.x = w[0]; v.y = w[1];
      __builtin_nontemporal_store(v, reinterpret_cast<u32x2*>(p));
    } else {
#pragma unroll
      for (int i = 0; i < kBytes / 16; ++i) {
        u32x4 v; v.x = w[4 * i + 0]; v.y = w[4 * i + 1]; v.z = w[4 * i + 2]; v.w = w[4 * i + 3];
        __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p) + i);
      }
    }
  }

  // element accessors on the packed words (compile-time index after unrolling)
  __device__ inline float get(int i) const {
    if constexpr (sizeof(T) == 4) {
      return to_f32(__builtin_bit_cast(T, w[i]));
    } else if constexpr (sizeof(T) == 2) {
      const uint16_t h = (uint16_t)(w[i >> 1] >> ((i & 1) * 16));
      return to_f32(__builtin_bit_cast(T, h));
    } else {
      const uint8_t b = (uint8_t)(w[i >> 2] >> ((i & 3) * 8));
      return to_f32(__builtin_bit_cast(T, b));
    }
  }
  // Pack E real values (float containers: RNE conversion, one v_cvt_pk_* per pair for 16-bit).
  __device__ inline void pack(const float (&v)[E]) {
    if constexpr (sizeof(T) == 4) {
#pragma unroll
      for (int i = 0; i < E; ++i) w[i] = __builtin_bit_cast(uint32_t, from_f32<T>(v[i]));
    } else if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int i = 0; i < E; i += 2) w[i >> 1] = pack2<T>(v[i], v[i + 1]);
    } else {
#pragma unroll
      for (int i = 0; i < E; i += 4)
        w[i >> 2] = pack_bytes((int)from_f32<T>(v[i]), (int)from_f32<T>(v[i + 1]),
                               (int)from_f32<T>(v[i + 2]), (int)from_f32<T>(v[i + 3]));
    }
  }
  // Pack E integer codes (already clamped to the container's range).
  __device__ inline void pack_int(const int (&v)[E]) {
    if constexpr (sizeof(T) == 4) {
#pragma unroll
      for (int i = 0; i < E; ++i) w[i] = (uint32_t)v[i];
    } else if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int i = 0; i < E; i += 2) w[i >> 1] = ((uint32_t)v[i] & 0xFFFFu) | ((uint32_t)v[i + 1] << 16);
    } else {
#pragma unroll
      for (int i = 0; i < E; i += 4) w[i >> 2] = pack_bytes(v[i], v[i + 1], v[i + 2], v[i + 3]);
    }
  }
};

}  // namespace ffq

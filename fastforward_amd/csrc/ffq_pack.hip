// ffq_pack.hip — A7: 4-bit storage in the GGUF Q4_0 nibble order.
//
// Reference convention: pack_q4_0_blocks, src/fastforward/export/stages/gguf/_packing.py:44-53
//   qs = code + 8 (clamped to [0, 15]); within each block of `block` codes the first half goes to
//   the low nibbles and the second half to the high nibbles: byte[j] = qs[j] | qs[j + block/2] << 4.
// The reference itself keeps 4-bit codes unpacked; this pair makes W4 weights occupy 0.5 B/elem in
// HBM while `unpack(pack(q)) == q` holds exactly.
#ifndef FFQ_NT_STREAMS
#define FFQ_NT_STREAMS 3  // nt loads and stores (ffq_vec.h)
#endif
#include "ffq_affine.h"
#include "ffq_common.h"
#include "ffq_vec.h"

#include <math.h>

namespace ffq {

// One lane produces 4 packed bytes (= 8 codes: 4 from each half of its block).
template <typename T>
__global__ __launch_bounds__(kBlock) void pack_int4_kernel(const T* __restrict__ codes, uint32_t* __restrict__ packed,
                                                           uint32_t nwords, FastDiv words_per_block,
                                                           uint32_t block) {
  const uint32_t stride = gridDim.x * (uint32_t)kBlock;
  for (uint32_t w = blockIdx.x * (uint32_t)kBlock + threadIdx.x; w < nwords; w += stride) {
    const uint32_t b = fdiv(w, words_per_block);
    const uint32_t j = (w - b * words_per_block.div) * 4;  // first byte index inside the block
    const T* lo = codes + (size_t)b * block + j;
    const T* hi = lo + block / 2;
    uint32_t word = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int l = (int)to_f32(lo[i]) + 8, h = (int)to_f32(hi[i]) + 8;
      l = l < 0 ? 0 : (l > 15 ? 15 : l);
      h = h < 0 ? 0 : (h > 15 ? 15 : h);
      word |= (uint32_t)(l | (h << 4)) << (8 * i);
    }
    packed[w] = word;
  }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void unpack_int4_kernel(const uint32_t* __restrict__ packed, T* __restrict__ codes,
                                                             uint32_t nwords, FastDiv words_per_block,
                                                             uint32_t block) {
  const uint32_t stride = gridDim.x * (uint32_t)kBlock;
  for (uint32_t w = blockIdx.x * (uint32_t)kBlock + threadIdx.x; w < nwords; w += stride) {
    const uint32_t b = fdiv(w, words_per_block);
    const uint32_t j = (w - b * words_per_block.div) * 4;
    const uint32_t word = packed[w];
    T* lo = codes + (size_t)b * block + j;
    T* hi = lo + block / 2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t byte = (word >> (8 * i)) & 0xFFu;
      lo[i] = from_f32<T>((float)((int)(byte & 15u) - 8));
      hi[i] = from_f32<T>((float)((int)(byte >> 4) - 8));
    }
  }
}

// any block size / dtype / alignment: one byte per lane
__global__ __launch_bounds__(kBlock) void pack_int4_generic_kernel(const void* __restrict__ codes, int dt,
                                                                   uint8_t* __restrict__ packed, int64_t nbytes,
                                                                   int64_t block) {
  const int64_t stride = (int64_t)gridDim.x * kBlock, half = block / 2;
  for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < nbytes; p += stride) {
    const int64_t b = p / half, j = p % half;
    int64_t l = (int64_t)load_any(codes, dt, b * block + j) + 8;
    int64_t h = (int64_t)load_any(codes, dt, b * block + half + j) + 8;
    l = l < 0 ? 0 : (l > 15 ? 15 : l);
    h = h < 0 ? 0 : (h > 15 ? 15 : h);
    packed[p] = (uint8_t)(l | (h << 4));
  }
}
__global__ __launch_bounds__(kBlock) void unpack_int4_generic_kernel(const uint8_t* __restrict__ packed,
                                                                     void* __restrict__ codes, int dt,
                                                                     int64_t nbytes, int64_t block) {
  const int64_t stride = (int64_t)gridDim.x * kBlock, half = block / 2;
  for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < nbytes; p += stride) {
    const int64_t b = p / half, j = p % half;
    const uint8_t byte = packed[p];
    store_any(codes, dt, b * block + j, (double)((int)(byte & 15) - 8));
    store_any(codes, dt, b * block + half + j, (double)((int)(byte >> 4) - 8));
  }
}


// ---- A1 + A7 in one pass: x -> 4-bit codes -> nibbles (2 + 0.5 B/elem for bf16) -------------------------
// One lane owns 16 consecutive elements of the low half of a packing block and the 16 elements at the same
// position of the high half: two 32-byte loads in, one dense 16-byte store out.
struct Pack4Args {
  uint32_t nitems;         // numel / 32
  uint32_t block;          // packing block (codes)
  FastDiv items_per_block; // block / 32
  FastDiv chunks_per_run;  // ROWS: run / 16 (16-element chunks per tile)
  int rows;                // 1: ROWS layout, 0: one tile
  uint32_t scale_stride, offset_stride;
};

// ITEM = elements of each half a lane owns (pack4_item). With ITEM = 8 on 16-bit data every load / store instruction of a
// wave covers whole 128-byte lines; with 16 a lane's 32 bytes take two instructions that each touch every line of the wave's
// span and use half of it. Both halves of a packing block that lie in one tile (group >= block) share one reciprocal.
template <typename T, bool HAS_OFFSET, int ITEM>
__global__ __launch_bounds__(kBlock) void quantize_pack_int4_kernel(const T* __restrict__ x, const float* __restrict__ scale,
                                                                   const float* __restrict__ offset,
                                                                   uint8_t* __restrict__ packed, Pack4Args a) {
  constexpr int U = 1;  // items per lane (2 measured slower: see pack4_item)
  const uint32_t first = blockIdx.x * (uint32_t)(kBlock * U) + threadIdx.x;
  Chunk<T, ITEM> cl[U], ch[U];
  uint32_t b[U], j[U], e_lo[U], e_hi[U];
  bool live[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const uint32_t item = first + u * kBlock;
    live[u] = item < a.nitems;
    b[u] = fdiv(live[u] ? item : 0u, a.items_per_block);
    j[u] = (live[u] ? item : 0u) - b[u] * a.items_per_block.div;
    e_lo[u] = b[u] * a.block + j[u] * ITEM;
    e_hi[u] = e_lo[u] + a.block / 2;
    if (live[u]) {
      cl[u].FFQ_SLOAD(x + e_lo[u]);
      ch[u].FFQ_SLOAD(x + e_hi[u]);
    }
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    if (!live[u]) continue;
    const uint32_t t_lo = a.rows ? fdiv(e_lo[u] / 16, a.chunks_per_run) : 0u;
    const uint32_t t_hi = a.rows ? fdiv(e_hi[u] / 16, a.chunks_per_run) : 0u;
    const float s_lo = scale[t_lo * a.scale_stride];
    const float o_lo = HAS_OFFSET ? rne(offset[t_lo * a.offset_stride]) : 0.0f;
    float xl[ITEM], xh[ITEM];
#pragma unroll
    for (int i = 0; i < ITEM; ++i) { xl[i] = cl[u].get(i); xh[i] = ch[u].get(i); }
    // A1 into int8 bytes clamped to [-8, 7] by ffq_affine.h's packed arithmetic (two elements per VALU op, its NaN self-check and
    // the reference chain behind it: the codes of ffq_quantize_by_tile at 4 bits), then nibble = (byte & 0xF) ^ 8 = code + 8, four
    // codes per op. (Round 5; before: one IEEE division chain, a convert and two clamps per element.)
    Chunk<int8_t, ITEM> bl, bh;
    quantize_chunk_to_bytes<ITEM>(xl, s_lo, o_lo, -8.0f, 7.0f, bl);
    if (t_hi == t_lo) {  // both halves of the packing block inside one tile (group >= block): the same parameters
      quantize_chunk_to_bytes<ITEM>(xh, s_lo, o_lo, -8.0f, 7.0f, bh);
    } else {
      const float s_hi = scale[t_hi * a.scale_stride];
      const float o_hi = HAS_OFFSET ? rne(offset[t_hi * a.offset_stride]) : 0.0f;
      quantize_chunk_to_bytes<ITEM>(xh, s_hi, o_hi, -8.0f, 7.0f, bh);
    }
    Chunk<uint8_t, ITEM> out;
#pragma unroll
    for (int w = 0; w < ITEM / 4; ++w)
      out.w[w] = ((bl.w[w] & 0x0F0F0F0Fu) ^ 0x08080808u) | (((bh.w[w] & 0x0F0F0F0Fu) ^ 0x08080808u) << 4);
    out.FFQ_SSTORE(packed + (size_t)b[u] * (a.block / 2) + j[u] * ITEM);
  }
}

// ---- A7 + A2 in one pass: nibbles -> codes -> (q + round(o)) * s (0.5 + 2 B/elem for bf16) -------------
template <typename T, bool HAS_OFFSET, int ITEM>
__global__ __launch_bounds__(kBlock) void unpack_dequantize_int4_kernel(const uint8_t* __restrict__ packed,
                                                                       const float* __restrict__ scale,
                                                                       const float* __restrict__ offset, T* __restrict__ out,
                                                                       Pack4Args a) {
  const uint32_t item = blockIdx.x * (uint32_t)kBlock + threadIdx.x;
  if (item >= a.nitems) return;
  const uint32_t b = fdiv(item, a.items_per_block);
  const uint32_t j = item - b * a.items_per_block.div;
  const uint32_t e_lo = b * a.block + j * ITEM, e_hi = e_lo + a.block / 2;
  Chunk<uint8_t, ITEM> in;
  in.FFQ_SLOAD(packed + (size_t)b * (a.block / 2) + j * ITEM);
  const uint32_t t_lo = a.rows ? fdiv(e_lo / 16, a.chunks_per_run) : 0u;
  const uint32_t t_hi = a.rows ? fdiv(e_hi / 16, a.chunks_per_run) : 0u;
  const float s_lo = scale[t_lo * a.scale_stride], s_hi = scale[t_hi * a.scale_stride];
  const float o_lo = HAS_OFFSET ? rne(offset[t_lo * a.offset_stride]) : 0.0f;
  const float o_hi = HAS_OFFSET ? rne(offset[t_hi * a.offset_stride]) : 0.0f;
  float yl[ITEM], yh[ITEM];
#pragma unroll
  for (int i = 0; i < ITEM; ++i) {
    const uint32_t byte = (in.w[i >> 2] >> (8 * (i & 3))) & 0xFFu;
    yl[i] = ((float)((int)(byte & 15u) - 8) + o_lo) * s_lo;  // add and multiply are separate fp32 roundings
    yh[i] = ((float)((int)(byte >> 4) - 8) + o_hi) * s_hi;
  }
  Chunk<T, ITEM> cl, ch;
  cl.pack(yl);
  ch.pack(yh);
  cl.FFQ_SSTORE(out + e_lo);
  ch.FFQ_SSTORE(out + e_hi);
}

// Elements of each half per lane. Measured on [14336, 4096] bf16, group 128 (tools/arith_ab.py, A/B on one box):
// unpack+dequantize 5.03-5.10 TB/s with 16, 5.51-5.59 with 8 (its two 16-bit stores become whole-line instructions);
// quantize+pack 5.25-5.31 with 16, 5.20-5.29 with 8 and 4.6-4.7 with 8 x 2 items per lane (a second reciprocal per 8
// elements, and short blocks beat unrolled ones here as in A1) — so the two directions take different shapes.
static int pack4_item(int dt, bool unpack) {
#ifdef FFQ_EXPERIMENTS
  static const int forced = getenv("FFQ_PACK_ITEM") ? atoi(getenv("FFQ_PACK_ITEM")) : 0;
#else
  constexpr int forced = 0;
#endif
  if (dt == FFQ_F32) return 16;
  if (forced == 8 || forced == 16) return forced;
  return unpack ? 8 : 16;
}

static int pack4_plan(const ffq_tiling* tiling, int64_t block, int64_t scale_numel, const float* offset, int64_t offset_numel,
                      Pack4Args* a, int64_t* numel, int item = 16) {
  TileInfo info;
  int rc = analyse(tiling, &info);
  if (rc) return rc;
  *numel = info.numel;
  if (block <= 0 || (block & 1) || info.numel % block) return fail(FFQ_ERR_ARG, "numel %% block != 0 or odd block");
  if (info.numel != 0) {
    if ((rc = check_param_numel("scale", scale_numel, info.ntiles))) return rc;
    if (offset && (rc = check_param_numel("offset", offset_numel, info.ntiles))) return rc;
  }
  const bool layout_ok = info.layout == LAYOUT_SCALAR || (info.layout == LAYOUT_ROWS && info.run % 16 == 0);
  if (!layout_ok || block % 32 != 0 || info.numel >= ((int64_t)1 << 32))
    return fail(FFQ_ERR_DTYPE, "fused 4-bit kernels cover per-tensor / contiguous-run tiles (run %% 16 == 0) and block %% 32 == 0");
  a->nitems = (uint32_t)(info.numel / (2 * item));
  a->block = (uint32_t)block;
  a->items_per_block = make_fastdiv((uint32_t)(block / (2 * item)));
  a->rows = info.layout == LAYOUT_ROWS;
  a->chunks_per_run = make_fastdiv(a->rows ? (uint32_t)(info.run / 16) : 1u);
  a->scale_stride = scale_numel == 1 ? 0u : 1u;
  a->offset_stride = offset_numel == 1 ? 0u : 1u;
  return FFQ_OK;
}

// ---- GGUF block-32 writers: Q4_0 = [fp16 d | 16 nibble bytes] (18 B), Q8_0 = [fp16 d | 32 int8] (34 B) ------------
// Reference: pack_q4_0_blocks / pack_q8_0_blocks, src/fastforward/export/stages/gguf/_packing.py:23-72 — FastForward's
// signed codes and positive per-block scales written in the byte layout llama.cpp dequantizes (d = +scale,
// qs = code + 8 nibble-packed low/high halves; Q8_0 codes clipped to [-127, 127]). One lane builds one block; the
// block's bytes go through LDS so that the 18- / 34-byte records leave as dense 16-byte stores.
template <int FORMAT>  // 4: Q4_0, 8: Q8_0
__global__ __launch_bounds__(kBlock) void pack_gguf_blocks_kernel(const int8_t* __restrict__ codes, const float* __restrict__ scales,
                                                                 uint8_t* __restrict__ out, uint32_t nblocks) {
  constexpr int REC = FORMAT == 4 ? 18 : 34;
  __shared__ __attribute__((aligned(16))) uint8_t stage[kBlock * REC];
  const uint32_t b = blockIdx.x * (uint32_t)kBlock + threadIdx.x;
  if (b < nblocks) {
    Chunk<int8_t, 16> lo, hi;
    lo.load(codes + (size_t)b * 32);
    hi.load(codes + (size_t)b * 32 + 16);
    uint16_t* rec16 = reinterpret_cast<uint16_t*>(stage + threadIdx.x * REC);  // REC is even: 2-byte aligned
    rec16[0] = f32_to_f16_bits(scales[b]);
    if constexpr (FORMAT == 4) {
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        uint32_t pair = 0;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          int l = (int)(int8_t)(lo.w[(i + k) >> 2] >> (8 * ((i + k) & 3))) + 8;
          int h = (int)(int8_t)(hi.w[(i + k) >> 2] >> (8 * ((i + k) & 3))) + 8;
          l = l < 0 ? 0 : (l > 15 ? 15 : l);
          h = h < 0 ? 0 : (h > 15 ? 15 : h);
          pair |= (uint32_t)(l | (h << 4)) << (8 * k);
        }
        rec16[1 + i / 2] = (uint16_t)pair;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 32; i += 2) {
        const uint32_t w = i < 16 ? lo.w[i >> 2] : hi.w[(i - 16) >> 2];
        int c0 = (int)(int8_t)(w >> (8 * (i & 3))), c1 = (int)(int8_t)(w >> (8 * ((i + 1) & 3)));
        c0 = c0 < -127 ? -127 : c0;
        c1 = c1 < -127 ? -127 : c1;
        rec16[1 + i / 2] = (uint16_t)((c0 & 0xFF) | ((c1 & 0xFF) << 8));
      }
    }
  }
  __syncthreads();
  // this thread block's records are contiguous in `out`, starting at a multiple of 256 * REC bytes (16-byte aligned)
  const uint32_t first = blockIdx.x * (uint32_t)kBlock;
  const uint32_t valid = (nblocks - first < (uint32_t)kBlock ? nblocks - first : (uint32_t)kBlock) * REC;
  uint8_t* dst = out + (size_t)first * REC;
  for (uint32_t off = threadIdx.x * 16; off < valid; off += kBlock * 16) {
    if (off + 16 <= valid) {
      *reinterpret_cast<u32x4*>(dst + off) = *reinterpret_cast<const u32x4*>(stage + off);
    } else {
      for (uint32_t k = off; k < valid; ++k) dst[k] = stage[k];
    }
  }
}

static bool fast_ok(const void* a, const void* b, int64_t numel, int64_t block) {
  return block % 8 == 0 && numel < ((int64_t)1 << 32) && (reinterpret_cast<uintptr_t>(a) & 3u) == 0 &&
         (reinterpret_cast<uintptr_t>(b) & 3u) == 0;
}
static unsigned grid_of(int64_t items) {
  int64_t blocks = (items + kBlock - 1) / kBlock;
  if (blocks > 16384) blocks = 16384;
  if (blocks < 1) blocks = 1;
  return (unsigned)blocks;
}

}  // namespace ffq

using namespace ffq;

extern "C" int ffq_pack_int4(const void* codes, int codes_dt, int64_t numel, int64_t block, uint8_t* packed,
                             void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (!dt_valid(codes_dt)) return fail(FFQ_ERR_ARG, "bad dtype tag");
  if (block <= 0 || (block & 1) || numel < 0 || numel % block) return fail(FFQ_ERR_ARG, "numel %% block != 0 or odd block");
  if (numel == 0) return FFQ_OK;
  if (!codes || !packed) return fail(FFQ_ERR_ARG, "NULL buffer");
  const int64_t nbytes = numel / 2;
  if (fast_ok(codes, packed, numel, block) && (codes_dt == FFQ_I8 || codes_dt == FFQ_BF16 || codes_dt == FFQ_F16 || codes_dt == FFQ_F32)) {
    const uint32_t nwords = (uint32_t)(nbytes / 4);
    const FastDiv wpb = make_fastdiv((uint32_t)(block / 8));
    uint32_t* out = reinterpret_cast<uint32_t*>(packed);
    const unsigned grid = grid_of(nwords);
    switch (codes_dt) {
      case FFQ_I8: pack_int4_kernel<int8_t><<<grid, kBlock, 0, s>>>((const int8_t*)codes, out, nwords, wpb, (uint32_t)block); break;
      case FFQ_BF16: pack_int4_kernel<bf16_t><<<grid, kBlock, 0, s>>>((const bf16_t*)codes, out, nwords, wpb, (uint32_t)block); break;
      case FFQ_F16: pack_int4_kernel<f16_t><<<grid, kBlock, 0, s>>>((const f16_t*)codes, out, nwords, wpb, (uint32_t)block); break;
      default: pack_int4_kernel<float><<<grid, kBlock, 0, s>>>((const float*)codes, out, nwords, wpb, (uint32_t)block); break;
    }
    return check_launch("pack_int4_kernel");
  }
  pack_int4_generic_kernel<<<grid_of(nbytes), kBlock, 0, s>>>(codes, codes_dt, packed, nbytes, block);
  return check_launch("pack_int4_generic_kernel");
}

extern "C" int ffq_unpack_int4(const uint8_t* packed, int64_t numel, int64_t block, void* codes_out, int codes_dt,
                               void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (!dt_valid(codes_dt)) return fail(FFQ_ERR_ARG, "bad dtype tag");
  if (block <= 0 || (block & 1) || numel < 0 || numel % block) return fail(FFQ_ERR_ARG, "numel %% block != 0 or odd block");
  if (numel == 0) return FFQ_OK;
  if (!codes_out || !packed) return fail(FFQ_ERR_ARG, "NULL buffer");
  const int64_t nbytes = numel / 2;
  if (fast_ok(codes_out, packed, numel, block) && (codes_dt == FFQ_I8 || codes_dt == FFQ_BF16 || codes_dt == FFQ_F16 || codes_dt == FFQ_F32)) {
    const uint32_t nwords = (uint32_t)(nbytes / 4);
    const FastDiv wpb = make_fastdiv((uint32_t)(block / 8));
    const uint32_t* in = reinterpret_cast<const uint32_t*>(packed);
    const unsigned grid = grid_of(nwords);
    switch (codes_dt) {
      case FFQ_I8: unpack_int4_kernel<int8_t><<<grid, kBlock, 0, s>>>(in, (int8_t*)codes_out, nwords, wpb, (uint32_t)block); break;
      case FFQ_BF16: unpack_int4_kernel<bf16_t><<<grid, kBlock, 0, s>>>(in, (bf16_t*)codes_out, nwords, wpb, (uint32_t)block); break;
      case FFQ_F16: unpack_int4_kernel<f16_t><<<grid, kBlock, 0, s>>>(in, (f16_t*)codes_out, nwords, wpb, (uint32_t)block); break;
      default: unpack_int4_kernel<float><<<grid, kBlock, 0, s>>>(in, (float*)codes_out, nwords, wpb, (uint32_t)block); break;
    }
    return check_launch("unpack_int4_kernel");
  }
  unpack_int4_generic_kernel<<<grid_of(nbytes), kBlock, 0, s>>>(packed, codes_out, codes_dt, nbytes, block);
  return check_launch("unpack_int4_generic_kernel");
}

extern "C" int ffq_quantize_pack_int4(const void* data, int data_dt, const float* scale, int64_t scale_numel,
                                      const float* offset, int64_t offset_numel, const ffq_tiling* tiling, int64_t block,
                                      uint8_t* packed, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  Pack4Args a;
  int64_t numel = 0;
  const int item = pack4_item(data_dt, false);
  int rc = pack4_plan(tiling, block, scale_numel, offset, offset_numel, &a, &numel, item);
  if (rc) return rc;
  if (numel == 0) return FFQ_OK;
  if (!data || !scale || !packed) return fail(FFQ_ERR_ARG, "NULL buffer");
  if (!(data_dt == FFQ_BF16 || data_dt == FFQ_F16 || data_dt == FFQ_F32) || !aligned16(data) || !aligned16(packed))
    return fail(FFQ_ERR_DTYPE, "fused quantize+pack is built for 16-byte aligned f32 / bf16 / f16 data");
  const unsigned grid = (a.nitems + kBlock - 1) / kBlock;
#define FFQ_QP(T, I)                                                                                                \
  do {                                                                                                              \
    if (offset) quantize_pack_int4_kernel<T, true, I><<<grid, kBlock, 0, s>>>(static_cast<const T*>(data), scale, offset, packed, a); \
    else quantize_pack_int4_kernel<T, false, I><<<grid, kBlock, 0, s>>>(static_cast<const T*>(data), scale, offset, packed, a);       \
  } while (0)
  switch (data_dt) {
    case FFQ_BF16: if (item == 8) FFQ_QP(bf16_t, 8); else FFQ_QP(bf16_t, 16); break;
    case FFQ_F16: if (item == 8) FFQ_QP(f16_t, 8); else FFQ_QP(f16_t, 16); break;
    default: FFQ_QP(float, 16); break;
  }
#undef FFQ_QP
  return check_launch("quantize_pack_int4_kernel");
}

extern "C" int ffq_unpack_dequantize_int4(const uint8_t* packed, const float* scale, int64_t scale_numel, const float* offset,
                                          int64_t offset_numel, const ffq_tiling* tiling, int64_t block, void* out, int out_dt,
                                          void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  Pack4Args a;
  int64_t numel = 0;
  const int item = pack4_item(out_dt, true);
  int rc = pack4_plan(tiling, block, scale_numel, offset, offset_numel, &a, &numel, item);
  if (rc) return rc;
  if (numel == 0) return FFQ_OK;
  if (!packed || !scale || !out) return fail(FFQ_ERR_ARG, "NULL buffer");
  if (!(out_dt == FFQ_BF16 || out_dt == FFQ_F16 || out_dt == FFQ_F32) || !aligned16(out) || !aligned16(packed))
    return fail(FFQ_ERR_DTYPE, "fused unpack+dequantize is built for 16-byte aligned f32 / bf16 / f16 outputs");
  const unsigned grid = (a.nitems + kBlock - 1) / kBlock;
#define FFQ_UD(T, I)                                                                                                \
  do {                                                                                                              \
    if (offset) unpack_dequantize_int4_kernel<T, true, I><<<grid, kBlock, 0, s>>>(packed, scale, offset, static_cast<T*>(out), a); \
    else unpack_dequantize_int4_kernel<T, false, I><<<grid, kBlock, 0, s>>>(packed, scale, offset, static_cast<T*>(out), a);       \
  } while (0)
  switch (out_dt) {
    case FFQ_BF16: if (item == 8) FFQ_UD(bf16_t, 8); else FFQ_UD(bf16_t, 16); break;
    case FFQ_F16: if (item == 8) FFQ_UD(f16_t, 8); else FFQ_UD(f16_t, 16); break;
    default: FFQ_UD(float, 16); break;
  }
#undef FFQ_UD
  return check_launch("unpack_dequantize_int4_kernel");
}

extern "C" int ffq_pack_gguf_blocks(const int8_t* codes, const float* scales, int64_t nblocks, int format, uint8_t* out, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (format != 4 && format != 8) return fail(FFQ_ERR_ARG, "GGUF block format must be 4 (Q4_0) or 8 (Q8_0)");
  if (nblocks < 0 || nblocks >= ((int64_t)1 << 31)) return fail(FFQ_ERR_ARG, "bad block count");
  if (nblocks == 0) return FFQ_OK;
  if (!codes || !scales || !out) return fail(FFQ_ERR_ARG, "NULL buffer");
  if (!aligned16(codes) || !aligned16(out)) return fail(FFQ_ERR_ARG, "codes and output must be 16-byte aligned");
  const unsigned grid = (unsigned)((nblocks + kBlock - 1) / kBlock);
  if (format == 4) pack_gguf_blocks_kernel<4><<<grid, kBlock, 0, s>>>(codes, scales, out, (uint32_t)nblocks);
  else pack_gguf_blocks_kernel<8><<<grid, kBlock, 0, s>>>(codes, scales, out, (uint32_t)nblocks);
  return check_launch("pack_gguf_blocks_kernel");
}

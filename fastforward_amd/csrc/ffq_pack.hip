// ffq_pack.hip — A7: 4-bit storage in the GGUF Q4_0 nibble order.
//
// Reference convention: pack_q4_0_blocks, src/fastforward/export/stages/gguf/_packing.py:44-53
//   qs = code + 8 (clamped to [0, 15]); within each block of `block` codes the first half goes to
//   the low nibbles and the second half to the high nibbles: byte[j] = qs[j] | qs[j + block/2] << 4.
// The reference itself keeps 4-bit codes unpacked; this pair makes W4 weights occupy 0.5 B/elem in
// HBM while `unpack(pack(q)) == q` holds exactly.
#include "ffq_common.h"
#include "ffq_vec.h"

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

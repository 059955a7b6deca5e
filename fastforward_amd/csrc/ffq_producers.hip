// ffq_producers.hip — A1 fused into the kernels that PRODUCE the tensors the Llama recipe quantizes.
//
// In the reference's quantized Llama (docs/examples/doc_helpers/quantized_llama/) every quantized
// linear quantizes its own input (nn/linear.py:33 -> input_quantizer -> quantize_by_tile), and the
// tensors it quantizes come straight out of three elementwise producers that run as eager ATen chains:
//   * RMSNorm (rms_norm.py:17-35; 8 passes: to fp32, pow, mean, add, rsqrt, mul, to bf16, mul) behind a
//     residual add (decoder.py:60-90) -> feeds q/k/v_proj and gate/up_proj,
//   * SiLU(gate) * up (mlp.py:30-40; 2 passes)                            -> feeds down_proj,
//   * rotary embedding (attention.py:20-41; 7 passes incl. two cat copies) between q/k_proj and SDPA.
// Each producer + A1 is one pass here: the value the eager chain would have written in bf16 is formed
// in registers with the same roundings (every ATen op rounds to the tensor dtype), optionally stored,
// and pushed through the A1 arithmetic of ffq_affine.h for up to three static per-tensor quantizers.
// HBM-bound like A1: algorithmic bytes per element are stated at each kernel.
#ifndef FFQ_NT_STREAMS
#define FFQ_NT_STREAMS 3  // nt loads AND nt stores of the code tensors (ffq_vec.h). Round 4, A/B of two builds on one box, two rounds:
#endif                    // SiLU*up + quantize 54.5 / 55.2 -> 51.4 / 50.2 us (0.67 -> 0.71-0.73 of 8 TB/s); RMSNorm unchanged (0.70)
#ifndef FFQ_SILU_GRID
#define FFQ_SILU_GRID 512  // blocks of the table-driven SiLU*up kernel: two 512-thread blocks per CU
#endif
#include "ffq_affine.h"
#include "ffq_common.h"
#include "ffq_silu.h"
#include "ffq_vec.h"

#include <math.h>
#include <stdlib.h>

namespace ffq {

struct FanOut {
  const float* scale[FFQ_MAX_FANOUT];
  const float* offset[FFQ_MAX_FANOUT];
  int8_t* codes[FFQ_MAX_FANOUT];
  int n;
  float lo, hi;
};

// Round two fp32 values to bf16 and back with ONE v_cvt_pk_bf16_f32 (RNE) + two bit moves, instead of the
// ~6-instruction integer sequence per value: the producers round after every ATen op of the eager chain.
__device__ __forceinline__ void bf16_round2(float& a, float& b) {
  const uint32_t w = pack2<bf16_t>(a, b);
  a = __builtin_bit_cast(float, w << 16);
  b = __builtin_bit_cast(float, w & 0xFFFF0000u);
}

// 16 bf16-valued floats -> int8 codes for every quantizer of the fan-out. Quantizers that hold the same
// (scale, offset) — q/k/v_proj see the same tensor, so RunningMinMax gives them the same range — reuse
// the codes of the first one (wave-uniform test).
struct FanParams {
  float s[FFQ_MAX_FANOUT], o[FFQ_MAX_FANOUT];
};
__device__ __forceinline__ FanParams load_fan(const FanOut& f) {
  FanParams p;
#pragma unroll
  for (int j = 0; j < FFQ_MAX_FANOUT; ++j) {
    p.s[j] = 1.0f;
    p.o[j] = 0.0f;
    if (j < f.n) {
      p.s[j] = f.scale[j][0];
      p.o[j] = f.offset[j] ? rne(f.offset[j][0]) : 0.0f;
    }
  }
  return p;
}
__device__ __forceinline__ void fan_store(const FanOut& f, const FanParams& p, const float (&z)[16], size_t at) {
  Chunk<int8_t, 16> y[FFQ_MAX_FANOUT];
#pragma unroll
  for (int j = 0; j < FFQ_MAX_FANOUT; ++j) {
    if (j >= f.n) break;
    bool reuse = false;
#pragma unroll
    for (int i = 0; i < j; ++i) {
      if (!reuse && p.s[i] == p.s[j] && p.o[i] == p.o[j]) {
        y[j] = y[i];
        reuse = true;
      }
    }
    if (!reuse) quantize_chunk_to_bytes<16>(z, p.s[j], p.o[j], f.lo, f.hi, y[j]);
    y[j].FFQ_SSTORE(f.codes[j] + at);
  }
}

// ---------------------------------------------------------------------------------------------------
// P1: sum = x + delta (residual add, bf16); z = weight * bf16(sum_f32 * rsqrt(mean(sum_f32^2) + eps));
//     codes_j = A1(z; s_j, o_j). WPR wavefronts per row (1 for short rows, 4 = the whole block for
//     hidden sizes above 1024), the row stays in registers between the reduction and the normalisation
//     (CPL chunks of 16 elements per lane: cols <= 1024 * WPR * CPL). Few registers per lane on purpose:
//     the one-wave-per-row form of a 4096-wide row needs 117 VGPRs (4 waves/SIMD) and reaches 4.5 TB/s.
//     Algorithmic bytes / element: 2 (x) [+ 2 (delta) + 2 (sum)] [+ 2 (z)] + 1 per distinct code tensor.
// ---------------------------------------------------------------------------------------------------
template <int CPL, int WPR>
__global__ __launch_bounds__(kBlock) void add_rmsnorm_quantize_kernel(const bf16_t* __restrict__ x,
                                                                      const bf16_t* __restrict__ delta,
                                                                      bf16_t* __restrict__ sum_out,
                                                                      const bf16_t* __restrict__ weight,
                                                                      bf16_t* __restrict__ norm_out, FanOut f,
                                                                      uint32_t rows, uint32_t chunks_per_row,
                                                                      float inv_cols, float eps) {
  constexpr uint32_t LPR = 64u * WPR;  // lanes per row
  const uint32_t lane = threadIdx.x % LPR;
  const uint32_t row = blockIdx.x * (kBlock / LPR) + threadIdx.x / LPR;
  if (row >= rows) return;  // block-uniform when WPR == 4
  const size_t base = (size_t)row * chunks_per_row * 16;
  Chunk<bf16_t, 16> h[CPL];
#pragma unroll
  for (int u = 0; u < CPL; ++u) {
    const uint32_t c = lane + LPR * u;
    if (c < chunks_per_row) h[u].FFQ_SLOAD(x + base + (size_t)c * 16);
  }
  if (delta) {
    Chunk<bf16_t, 16> d[CPL];
#pragma unroll
    for (int u = 0; u < CPL; ++u) {
      const uint32_t c = lane + LPR * u;
      if (c < chunks_per_row) d[u].FFQ_SLOAD(delta + base + (size_t)c * 16);
    }
#pragma unroll
    for (int u = 0; u < CPL; ++u) {
      const uint32_t c = lane + LPR * u;
      if (c >= chunks_per_row) continue;
      float v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = h[u].get(i) + d[u].get(i);  // bf16 + bf16 -> bf16 (one rounding)
      h[u].pack(v);
      if (sum_out) h[u].store(sum_out + base + (size_t)c * 16);
    }
  }
  // variance = mean(h^2) in fp32 (rms_norm.py:28); the summation order is this kernel's own
  float ss = 0.0f;
#pragma unroll
  for (int u = 0; u < CPL; ++u) {
    const uint32_t c = lane + LPR * u;
    if (c >= chunks_per_row) continue;
    float part = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float v = h[u].get(i);
      part = part + v * v;
    }
    ss = ss + part;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) ss = ss + __shfl_xor(ss, d, 64);
  if constexpr (WPR > 1) {
    __shared__ float wave_ss[kBlock / 64];
    if ((threadIdx.x & 63u) == 0) wave_ss[threadIdx.x >> 6] = ss;
    __syncthreads();
    ss = ((wave_ss[0] + wave_ss[1]) + wave_ss[2]) + wave_ss[3];
  }
  const float r = rsqrtf(ss * inv_cols + eps);
  const FanParams p = load_fan(f);
#pragma unroll
  for (int u = 0; u < CPL; ++u) {
    const uint32_t c = lane + LPR * u;
    if (c >= chunks_per_row) continue;
    Chunk<bf16_t, 16> w;
    w.load(weight + (size_t)c * 16);
    float z[16];
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
      float n0 = h[u].get(i) * r, n1 = h[u].get(i + 1) * r;
      bf16_round2(n0, n1);                      // (hidden * rsqrt).to(bf16)
      z[i] = w.get(i) * n0;
      z[i + 1] = w.get(i + 1) * n1;
      bf16_round2(z[i], z[i + 1]);              // weight * hidden in bf16
    }
    if (norm_out) {
      Chunk<bf16_t, 16> zc;
      zc.pack(z);
      zc.store(norm_out + base + (size_t)c * 16);
    }
    fan_store(f, p, z, base + (size_t)c * 16);
  }
}

// ---------------------------------------------------------------------------------------------------
// P2: z = bf16(silu(gate)) * up in bf16 (mlp.py:36-38: F.silu rounds to bf16, the product rounds again);
//     codes = A1(z). silu(x) = x / (1 + exp(-x)) evaluated in fp32 exactly as ATen's silu kernel does.
//     Algorithmic bytes / element: 2 + 2 [+ 2 (z)] + 1.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void silu_mul_quantize_kernel(const bf16_t* __restrict__ gate,
                                                                   const bf16_t* __restrict__ up,
                                                                   bf16_t* __restrict__ product_out, FanOut f,
                                                                   uint32_t nchunks) {
  const uint32_t c = blockIdx.x * (uint32_t)kBlock + threadIdx.x;
  if (c >= nchunks) return;
  Chunk<bf16_t, 16> g, u;
  g.load(gate + (size_t)c * 16);
  u.load(up + (size_t)c * 16);
  const FanParams p = load_fan(f);
  float z[16];
#pragma unroll
  for (int i = 0; i < 16; i += 2) {
    const float x0 = g.get(i), x1 = g.get(i + 1);
    float a0 = x0 / (1.0f + expf(-x0)), a1 = x1 / (1.0f + expf(-x1));
    bf16_round2(a0, a1);                        // F.silu rounds to bf16
    z[i] = a0 * u.get(i);
    z[i + 1] = a1 * u.get(i + 1);
    bf16_round2(z[i], z[i + 1]);                // the product rounds again
  }
  if (product_out) {
    Chunk<bf16_t, 16> zc;
    zc.pack(z);
    zc.store(product_out + (size_t)c * 16);
  }
  fan_store(f, p, z, (size_t)c * 16);
}

// P2 for large tensors: silu through the 16 KiB LDS table of ffq_silu.h (a function of a bf16 argument has 65536 values;
// the exact expf + IEEE division made the kernel above VALU-bound at half the HBM rate). 512 threads fill the table with
// silu_exact (16 entries each), then walk the tensor grid-stride with the next chunk's loads in flight.
constexpr int kSiluBlock = 512;
__global__ __launch_bounds__(kSiluBlock) void silu_mul_quantize_table_kernel(const bf16_t* __restrict__ gate,
                                                                             const bf16_t* __restrict__ up,
                                                                             bf16_t* __restrict__ product_out, FanOut f,
                                                                             uint32_t nchunks) {
  __shared__ uint16_t table[kSiluEntries];
  const uint32_t stride = gridDim.x * (uint32_t)kSiluBlock;
  uint32_t c = blockIdx.x * (uint32_t)kSiluBlock + threadIdx.x;
  Chunk<bf16_t, 16> g, u;
  if (c < nchunks) {
    g.load(gate + (size_t)c * 16);
    u.load(up + (size_t)c * 16);
  }
  silu_table_fill(table, threadIdx.x, kSiluBlock);
  const FanParams p = load_fan(f);
  __syncthreads();
  while (c < nchunks) {
    const uint32_t cn = c + stride;
    Chunk<bf16_t, 16> gn, un;   // a ring of two chunks or non-temporal loads: no faster (A/B on one box)
    if (cn < nchunks) {
      gn.load(gate + (size_t)cn * 16);
      un.load(up + (size_t)cn * 16);
    }
    uint32_t a[8];              // F.silu rounds to bf16
    uint32_t bad = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = silu_pair_lookup(g.w[j], table, bad);
    if (__builtin_expect(silu_any_outside(bad), 0)) {
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] = silu_pair_patch(g.w[j], a[j]);
    }
    float z[16];
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
      const uint32_t w = a[i >> 1];
      z[i] = __builtin_bit_cast(float, w << 16) * u.get(i);
      z[i + 1] = __builtin_bit_cast(float, w & 0xFFFF0000u) * u.get(i + 1);
      bf16_round2(z[i], z[i + 1]);  // the product rounds again
    }
    if (product_out) {
      Chunk<bf16_t, 16> zc;
      zc.pack(z);
      zc.store(product_out + (size_t)c * 16);
    }
    fan_store(f, p, z, (size_t)c * 16);
    g = gn;
    u = un;
    c = cn;
  }
}

// ---------------------------------------------------------------------------------------------------
// P3: rotary embedding in place on the q/k projections as they leave the GEMM ([tokens, heads * D]):
//     out = bf16(bf16(q * cos) + bf16(rotate_half(q) * sin))          (attention.py:20-41)
//     A lane owns elements [8j, 8j+8) of both halves of one head row, so it reads everything it
//     overwrites. Algorithmic bytes / element: 2 + 2 (tables are L2-resident).
// ---------------------------------------------------------------------------------------------------
struct RopeArgs {
  bf16_t* q; bf16_t* k;
  const bf16_t* cos; const bf16_t* sin;
  uint32_t q_heads, k_heads, half_chunks;  // half_chunks = D / 16
  uint32_t seq_len, head_dim;
  uint32_t nitems;                         // tokens * (q_heads + k_heads) * half_chunks
  FastDiv by_half_chunks, by_heads, by_seq;
};
__global__ __launch_bounds__(kBlock) void rope_kernel(RopeArgs a) {
  const uint32_t item = blockIdx.x * (uint32_t)kBlock + threadIdx.x;
  if (item >= a.nitems) return;
  const uint32_t hrow = fdiv(item, a.by_half_chunks);          // token * heads + head
  const uint32_t j = item - hrow * a.half_chunks;
  const uint32_t token = fdiv(hrow, a.by_heads);
  const uint32_t head = hrow - token * (a.q_heads + a.k_heads);
  const uint32_t pos = token - fdiv(token, a.by_seq) * a.seq_len;
  const uint32_t half = a.head_dim / 2;
  bf16_t* row = head < a.q_heads ? a.q + ((size_t)token * a.q_heads + head) * a.head_dim
                                 : a.k + ((size_t)token * a.k_heads + (head - a.q_heads)) * a.head_dim;
  Chunk<bf16_t, 8> lo, hi, cl, ch, sl, sh;
  lo.load(row + j * 8);
  hi.load(row + half + j * 8);
  const bf16_t* cr = a.cos + (size_t)pos * a.head_dim;
  const bf16_t* sr = a.sin + (size_t)pos * a.head_dim;
  cl.load(cr + j * 8); ch.load(cr + half + j * 8);
  sl.load(sr + j * 8); sh.load(sr + half + j * 8);
  float ol[8], oh[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float x1 = lo.get(i), x2 = hi.get(i);
    float a = x1 * cl.get(i), b = (-x2) * sl.get(i), c = x2 * ch.get(i), d = x1 * sh.get(i);
    bf16_round2(a, b);
    bf16_round2(c, d);
    ol[i] = a + b;
    oh[i] = c + d;
  }
  lo.pack(ol);
  hi.pack(oh);
  lo.store(row + j * 8);
  hi.store(row + half + j * 8);
}

static int fan_from_abi(const ffq_fanout* fan, int64_t numel, FanOut* out) {
  out->n = 0;
  out->lo = out->hi = 0.0f;
  for (int j = 0; j < FFQ_MAX_FANOUT; ++j) { out->scale[j] = nullptr; out->offset[j] = nullptr; out->codes[j] = nullptr; }
  if (!fan) return FFQ_OK;
  if (fan->count < 0 || fan->count > FFQ_MAX_FANOUT) return fail(FFQ_ERR_ARG, "fan-out count must be 0..%d", FFQ_MAX_FANOUT);
  if (fan->count == 0) return FFQ_OK;
  if (!(fan->num_bits >= 1 && fan->num_bits <= 8 && fan->num_bits == floor(fan->num_bits)))
    return fail(FFQ_ERR_PRECISION, "Provided dtype (%d) is not enough to store %g bits quantized values.", FFQ_I8, fan->num_bits);
  out->n = fan->count;
  const double lo = -pow(2.0, fan->num_bits - 1.0);
  out->lo = (float)lo;
  out->hi = (float)(-lo - 1.0);
  for (int j = 0; j < fan->count; ++j) {
    if (!fan->scale[j] || !fan->codes[j]) return fail(FFQ_ERR_ARG, "NULL scale / codes in fan-out %d", j);
    if (numel && !aligned16(fan->codes[j])) return fail(FFQ_ERR_ARG, "codes buffer %d must be 16-byte aligned", j);
    out->scale[j] = fan->scale[j];
    out->offset[j] = fan->offset[j];
    out->codes[j] = fan->codes[j];
  }
  return FFQ_OK;
}

}  // namespace ffq

using namespace ffq;

extern "C" int ffq_add_rmsnorm_quantize(const void* x, const void* delta, void* sum_out, const void* weight,
                                        int dt, int64_t rows, int64_t cols, double eps, void* norm_out,
                                        const ffq_fanout* fan, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (rows < 0 || cols < 0) return fail(FFQ_ERR_ARG, "negative extent");
  if (dt != FFQ_BF16) return fail(FFQ_ERR_DTYPE, "fused RMSNorm is built for bf16 activations");
  if (cols == 0) return fail(FFQ_ERR_EMPTY, "RMSNorm over an empty row");
  if (cols % 16 != 0 || cols > 8192) return fail(FFQ_ERR_DTYPE, "fused RMSNorm needs cols %% 16 == 0 and cols <= 8192 (got %lld)", (long long)cols);
  if (rows >= ((int64_t)1 << 31)) return fail(FFQ_ERR_ARG, "too many rows");
  FanOut f;
  int rc = fan_from_abi(fan, rows * cols, &f);
  if (rc) return rc;
  if (rows == 0) return FFQ_OK;
  if (!x || !weight) return fail(FFQ_ERR_ARG, "NULL buffer");
  if (!aligned16(x) || !aligned16(weight) || (delta && !aligned16(delta)) || (sum_out && !aligned16(sum_out)) ||
      (norm_out && !aligned16(norm_out)))
    return fail(FFQ_ERR_ARG, "buffers must be 16-byte aligned");
  const uint32_t cpr = (uint32_t)(cols / 16);
  const float inv = (float)(1.0 / (double)cols);  // ATen's mean multiplies the sum by float(1/N)
#define FFQ_P1(CPL, WPR)                                                                                  \
  add_rmsnorm_quantize_kernel<CPL, WPR><<<(unsigned)((rows + 4 / WPR - 1) / (4 / WPR)), kBlock, 0, s>>>(  \
      static_cast<const bf16_t*>(x), static_cast<const bf16_t*>(delta), static_cast<bf16_t*>(sum_out),    \
      static_cast<const bf16_t*>(weight), static_cast<bf16_t*>(norm_out), f, (uint32_t)rows, cpr, inv, (float)eps)
  if (cpr <= 64) FFQ_P1(1, 1);
  else if (cpr <= 256) FFQ_P1(1, 4);
  else FFQ_P1(2, 4);
#undef FFQ_P1
  return check_launch("add_rmsnorm_quantize_kernel");
}

extern "C" int ffq_silu_mul_quantize(const void* gate, const void* up, int dt, int64_t numel, void* product_out,
                                     const ffq_fanout* fan, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (numel < 0) return fail(FFQ_ERR_ARG, "negative extent");
  if (dt != FFQ_BF16) return fail(FFQ_ERR_DTYPE, "fused SiLU*up is built for bf16 activations");
  if (numel % 16 != 0 || numel >= ((int64_t)1 << 35)) return fail(FFQ_ERR_DTYPE, "fused SiLU*up needs numel %% 16 == 0 and numel < 2^35");
  FanOut f;
  int rc = fan_from_abi(fan, numel, &f);
  if (rc) return rc;
  if (numel == 0) return FFQ_OK;
  if (!gate || !up) return fail(FFQ_ERR_ARG, "NULL buffer");
  if (!aligned16(gate) || !aligned16(up) || (product_out && !aligned16(product_out)))
    return fail(FFQ_ERR_ARG, "buffers must be 16-byte aligned");
  const uint32_t nchunks = (uint32_t)(numel / 16);
  if (nchunks >= 8u * kSiluBlock * 256u) {  // >= 4 chunks per thread of the two-blocks-per-CU grid: the table pays
    silu_mul_quantize_table_kernel<<<FFQ_SILU_GRID, kSiluBlock, 0, s>>>(
        static_cast<const bf16_t*>(gate), static_cast<const bf16_t*>(up), static_cast<bf16_t*>(product_out), f, nchunks);
    return check_launch("silu_mul_quantize_table_kernel");
  }
  silu_mul_quantize_kernel<<<(nchunks + kBlock - 1) / kBlock, kBlock, 0, s>>>(
      static_cast<const bf16_t*>(gate), static_cast<const bf16_t*>(up), static_cast<bf16_t*>(product_out), f, nchunks);
  return check_launch("silu_mul_quantize_kernel");
}

extern "C" int ffq_rope_inplace(void* q, int64_t q_heads, void* k, int64_t k_heads, int dt, int64_t tokens,
                                int64_t seq_len, int64_t head_dim, const void* cos_table, const void* sin_table,
                                void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (tokens < 0 || q_heads < 0 || k_heads < 0 || seq_len <= 0 || head_dim <= 0) return fail(FFQ_ERR_ARG, "bad extent");
  if (dt != FFQ_BF16) return fail(FFQ_ERR_DTYPE, "fused rotary embedding is built for bf16 activations");
  if (head_dim % 16 != 0) return fail(FFQ_ERR_DTYPE, "fused rotary embedding needs head_dim %% 16 == 0");
  if (tokens % seq_len != 0) return fail(FFQ_ERR_ARG, "tokens must be a multiple of seq_len");
  const int64_t heads = q_heads + k_heads;
  const int64_t items = tokens * heads * (head_dim / 16);
  if (items == 0) return FFQ_OK;
  if (items >= ((int64_t)1 << 32)) return fail(FFQ_ERR_ARG, "too many elements for one launch");
  if ((q_heads && !q) || (k_heads && !k) || !cos_table || !sin_table) return fail(FFQ_ERR_ARG, "NULL buffer");
  if ((q && !aligned16(q)) || (k && !aligned16(k)) || !aligned16(cos_table) || !aligned16(sin_table))
    return fail(FFQ_ERR_ARG, "buffers must be 16-byte aligned");
  RopeArgs a;
  a.q = static_cast<bf16_t*>(q); a.k = static_cast<bf16_t*>(k);
  a.cos = static_cast<const bf16_t*>(cos_table); a.sin = static_cast<const bf16_t*>(sin_table);
  a.q_heads = (uint32_t)q_heads; a.k_heads = (uint32_t)k_heads;
  a.half_chunks = (uint32_t)(head_dim / 16);
  a.seq_len = (uint32_t)seq_len; a.head_dim = (uint32_t)head_dim;
  a.nitems = (uint32_t)items;
  a.by_half_chunks = make_fastdiv(a.half_chunks);
  a.by_heads = make_fastdiv((uint32_t)heads);
  a.by_seq = make_fastdiv((uint32_t)seq_len);
  rope_kernel<<<(unsigned)((items + kBlock - 1) / kBlock), kBlock, 0, s>>>(a);
  return check_launch("rope_kernel");
}

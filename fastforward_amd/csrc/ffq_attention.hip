// ffq_attention.hip — the attention between q/k/v_proj and o_proj of the reference's quantized Llama
// (docs/examples/doc_helpers/quantized_llama/attention.py:45-92: repeat_kv, matmul, scale, causal mask,
// fp32 softmax, matmul; the three quantizers inside are stubs in the recipe), fused with the INPUT
// QUANTIZER of o_proj (nn/linear.py:33 -> A1), so the context tensor never has to visit HBM as bf16.
//
// Flash-style: no S x S matrix. One workgroup = 8 waves = 256 query rows of one (batch, head); a wave owns
// 32 rows. K/V tiles of 64 keys are register-staged into a 2-deep LDS ring (global loads issued before the
// tile's matrix work, LDS writes after it, one barrier per tile).
//   * scores transposed: S^T[key][query] = mfma_32x32x16_bf16(A = K rows, B = Q rows), so a lane holds one
//     query column: the row max / row sum of the online softmax are 32 in-register ops + one cross-half
//     exchange, and the running (m, l) are lane-local;
//   * context transposed: O^T[d][query] = mfma(A = V^T, B = P): P's bf16 fragments are the exponentiated
//     score registers in place (a contraction is invariant under a permutation of its index applied to both
//     operands), V^T fragments come from the row-major V image through ds_read_b64_tr_b16 (row pitch 320 B:
//     the four key rows of a transposing read land in four different 64-byte bank groups);
//   * K image XOR-swizzled (16-byte slot ^= row & 15) so the ds_read_b128 of a fragment is conflict-free;
//   * causal: tiles above a wave's rows are skipped wave-uniformly, the diagonal tiles mask in registers,
//     heavy query blocks are dispatched first; blockIdx -> (kv head = XCD) so the 4 query heads of a GQA
//     group and the query blocks of one sequence share their K/V in one XCD's L2;
//   * epilogue: O^T / l -> bf16 -> per-wave LDS tile -> whole rows back out: bf16 context (optional) and the
//     int8 codes of A1 with the arithmetic of ffq_affine.h (bit-identical to ffq_quantize_by_tile on the
//     bf16 context this call produces).
// MFMA-bound (4 * S^2 * D / 2 flops per (batch, head) causal); bf16 operands, fp32 accumulation and softmax.
#include "ffq_affine.h"
#include "ffq_common.h"
#include "ffq_vec.h"

#include <math.h>
#include <stdlib.h>


namespace ffq {
namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

constexpr int kD = 128;             // head dim
constexpr int kKeys = 64;           // keys per tile
#ifndef FFQ_ATTN_WAVES
#define FFQ_ATTN_WAVES 8  // waves (of 32 query rows) per workgroup; A/B hook (tools/build_variant.sh)
#endif
constexpr int kWaves = FFQ_ATTN_WAVES;
constexpr int kRowsPerWave = 32;
constexpr int kQBlock = kWaves * kRowsPerWave;  // 256 query rows per workgroup
constexpr int kKPitch = 256;        // bytes per K row in LDS (swizzled slots)
constexpr int kVPitch = 320;        // bytes per V row in LDS (64-byte skew per row)
constexpr int kKBytes = kKeys * kKPitch;
constexpr int kVBytes = kKeys * kVPitch;
constexpr int kLdsBytes = 2 * kKBytes + 2 * kVBytes;   // 73728
constexpr float kDefer = 8.0f;      // log2 of the growth of a row maximum tolerated before O / l are rescaled
constexpr int kOPitch = 272;        // bytes per output row in the epilogue image
static_assert(kWaves * kRowsPerWave * kOPitch <= kLdsBytes, "epilogue image must fit the K/V ring");

struct AttnArgs {
  const uint16_t* q;
  const uint16_t* k;
  const uint16_t* v;
  uint16_t* ctx;          // nullable
  int8_t* codes;          // nullable
  const float* scale;     // A1 parameters of the codes
  const float* offset;    // nullable
  int32_t B, S, H, HKV;
  int32_t nqb;            // query blocks per sequence
  float c;                // softmax scale * log2(e)
  float lo, hi;           // clamp bounds of the codes
  const uint16_t* q_cos;  // nullable: rotary tables [S][128] — q arrives UN-rotated and is rotated as it is loaded
  const uint16_t* q_sin;
};

// V^T fragment of one MFMA: 8 keys x 1 column per lane = two transposing reads (ds_read_b64_tr_b16: lane i of a 16-lane
// group supplies the address of 4 consecutive bf16 — key row i / 4, columns 4 (i % 4) .. + 3 of a [4 keys][16 columns]
// block — and receives column i of that block, keys 0..3).
__device__ __forceinline__ bf16x8 load_vt(const unsigned char* vbuf, uint32_t lane_off, int imm) {
  typedef s16x4 __attribute__((address_space(3))) * lds_ptr;
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(vbuf + lane_off + imm));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(vbuf + lane_off + imm + 8 * kVPitch));
  const s16x8 ab = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, ab);
}

template <bool CAUSAL>
__global__ __launch_bounds__(kWaves * 64) void attention_fwd_kernel(AttnArgs a) {
  constexpr int QBLOCK = kQBlock;                 // query rows per workgroup
  constexpr int PASSES = kKeys / (kWaves * 4);    // staging passes: kWaves * 4 rows of 16 slots each

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63, wave = tid >> 6;
  const uint32_t r32 = lane & 31, h = lane >> 5;

  // blockIdx -> (kv head, head in group, query block (heavy first), batch)
  const uint32_t groups = (uint32_t)(a.H / a.HKV);
  uint32_t bid = blockIdx.x;
  const uint32_t kvh = bid % (uint32_t)a.HKV;
  bid /= (uint32_t)a.HKV;
  const uint32_t hg = bid % groups;
  bid /= groups;
  const uint32_t qb = (uint32_t)a.nqb - 1 - bid % (uint32_t)a.nqb;
  const uint32_t b = bid / (uint32_t)a.nqb;
  const uint32_t head = kvh * groups + hg;

  const int32_t q0 = (int32_t)qb * QBLOCK;
  const int32_t qw0 = q0 + (int32_t)wave * kRowsPerWave;
  const bool wave_valid = qw0 < a.S;
  const int32_t q_end = q0 + QBLOCK < a.S ? q0 + QBLOCK : a.S;
  const int32_t ntiles = (CAUSAL ? q_end : a.S) / kKeys;

  // ---- K/V staging: thread -> (row = tid / 16 [+32], 16-byte slot = tid % 16)
  const uint32_t srow = tid >> 4, sslot = tid & 15;
  const size_t kv_stride = (size_t)a.HKV * kD;  // elements between consecutive keys
  const uint16_t* kbase = a.k + ((size_t)b * a.S * a.HKV + kvh) * kD + sslot * 8;
  const uint16_t* vbase = a.v + ((size_t)b * a.S * a.HKV + kvh) * kD + sslot * 8;
  u32x4 sk[PASSES], sv[PASSES];
  auto stage_load = [&](int32_t tile) {
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const size_t row = (size_t)tile * kKeys + srow + kWaves * 4 * p;
      sk[p] = *reinterpret_cast<const u32x4*>(kbase + row * kv_stride);
      sv[p] = *reinterpret_cast<const u32x4*>(vbase + row * kv_stride);
    }
  };
  auto stage_store = [&](int buf) {
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const uint32_t row = srow + kWaves * 4 * p;
      *reinterpret_cast<u32x4*>(smem + buf * kKBytes + row * kKPitch + ((sslot ^ (row & 15)) << 4)) = sk[p];
      *reinterpret_cast<u32x4*>(smem + 2 * kKBytes + buf * kVBytes + row * kVPitch + (sslot << 4)) = sv[p];
    }
  };

  // the first tile's K / V rows are requested BEFORE the query fragments are loaded and rotated: one round trip for all of them,
  // and the rotation's arithmetic (a prologue of this one-workgroup-per-CU kernel that nothing else overlaps) runs under it
  stage_load(0);

  // ---- Q fragments: lane (r32, h) holds columns [(2t+h)*8, +8) of row qw0 + r32, t = 0..7
  bf16x8 qf[8];
  {
    const uint16_t* qrow = a.q + (((size_t)b * a.S + (size_t)(wave_valid ? qw0 + (int32_t)r32 : 0)) * a.H + head) * kD;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      u32x4 v = *reinterpret_cast<const u32x4*>(qrow + (2 * t + h) * 8);
      if (!wave_valid) v = u32x4{0, 0, 0, 0};
      qf[t] = __builtin_bit_cast(bf16x8, v);
    }
    // Rotary embedding of q on the way in (attention.py:20-41: out = bf16(bf16(q * cos) + bf16(rotate_half(q) * sin)), the
    // arithmetic of ffq_rope_inplace): a lane's fragments t and t + 4 are columns c .. c + 7 and c + 64 .. c + 71 of ITS row — the
    // two halves the rotation pairs — so nothing leaves the lane, and a query row is rotated once, by the one wave that owns it.
    // The q pass of the stand-alone rotary kernel (a read and a write of the whole q projection per layer) is gone.
    if (a.q_cos && wave_valid) {
      const size_t trow = (size_t)(qw0 + (int32_t)r32) * kD;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int c = (2 * t + (int)h) * 8;
        const u32x4 cl = *reinterpret_cast<const u32x4*>(a.q_cos + trow + c), ch = *reinterpret_cast<const u32x4*>(a.q_cos + trow + 64 + c);
        const u32x4 sl = *reinterpret_cast<const u32x4*>(a.q_sin + trow + c), sh = *reinterpret_cast<const u32x4*>(a.q_sin + trow + 64 + c);
        const u32x4 lo = __builtin_bit_cast(u32x4, qf[t]), hi = __builtin_bit_cast(u32x4, qf[t + 4]);
        const uint32_t lw[4] = {lo.x, lo.y, lo.z, lo.w}, hw[4] = {hi.x, hi.y, hi.z, hi.w};
        const uint32_t clw[4] = {cl.x, cl.y, cl.z, cl.w}, chw[4] = {ch.x, ch.y, ch.z, ch.w};
        const uint32_t slw[4] = {sl.x, sl.y, sl.z, sl.w}, shw[4] = {sh.x, sh.y, sh.z, sh.w};
        uint32_t ol[4], oh[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          float r[2][2];
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            auto el = [&](uint32_t word) { return __builtin_bit_cast(float, e ? (word & 0xFFFF0000u) : (word << 16)); };
            const float x1 = el(lw[w]), x2 = el(hw[w]);
            float p = x1 * el(clw[w]), q2 = (-x2) * el(slw[w]), u = x2 * el(chw[w]), d = x1 * el(shw[w]);
            const uint32_t pq = pack2<bf16_t>(p, q2), ud = pack2<bf16_t>(u, d);  // every product rounds to bf16
            r[0][e] = __builtin_bit_cast(float, pq << 16) + __builtin_bit_cast(float, pq & 0xFFFF0000u);
            r[1][e] = __builtin_bit_cast(float, ud << 16) + __builtin_bit_cast(float, ud & 0xFFFF0000u);
          }
          ol[w] = pack2<bf16_t>(r[0][0], r[0][1]);  // ... and so does the sum
          oh[w] = pack2<bf16_t>(r[1][0], r[1][1]);
        }
        qf[t] = __builtin_bit_cast(bf16x8, u32x4{ol[0], ol[1], ol[2], ol[3]});
        qf[t + 4] = __builtin_bit_cast(bf16x8, u32x4{oh[0], oh[1], oh[2], oh[3]});
      }
    }
  }

  f32x16 o[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[i][e] = 0.0f;
  float m_run = -1.0e30f, l_run = 0.0f;

  // per-lane part of the transposing-read address: 16-lane group g = lane / 16 -> (h = g / 2, column half = g % 2),
  // lane i of the group supplies the address of key row 4h + i / 4, columns 16 (g % 2) + 4 (i % 4) .. + 3
  const uint32_t i16 = lane & 15, g16 = lane >> 4;
  const uint32_t vt_lane_off = (4 * (g16 >> 1) + (i16 >> 2)) * kVPitch + (16 * (g16 & 1) + 4 * (i16 & 3)) * 2;

  stage_store(0);
  __syncthreads();

  for (int32_t tile = 0; tile < ntiles; ++tile) {
    const int buf = tile & 1;
    const bool more = tile + 1 < ntiles;
    if (more) stage_load(tile + 1);
    const int32_t kv0 = tile * kKeys;
    const bool active = wave_valid && (!CAUSAL || kv0 <= qw0 + kRowsPerWave - 1);
    if (active) {
      const unsigned char* kbuf = smem + buf * kKBytes;
      const unsigned char* vbuf = smem + 2 * kKBytes + buf * kVBytes;
      // ---- S^T = K Q^T: all 16 K fragments requested up front (one LDS latency per tile, not one per MFMA), the two
      //      32-key halves accumulate alternately so consecutive MFMAs are independent
      f32x16 s[2];
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int e = 0; e < 16; ++e) s[sub][e] = 0.0f;
      bf16x8 kf[2][8];
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
          const uint32_t krow = 32 * sub + r32;
          kf[sub][t] = *reinterpret_cast<const bf16x8*>(kbuf + krow * kKPitch + (((2 * t + h) ^ (krow & 15)) << 4));
        }
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
          s[sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[sub][t], qf[t], s[sub], 0, 0, 0);
      // keep that order: the scheduler otherwise pairs each read with its MFMA to save registers
      __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);  // 16 DS reads
      __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);  // 16 MFMAs
      // ---- causal mask on the diagonal tiles: key kv0 + 32 sub + crow(e, h) against query qw0 + r32
      if (CAUSAL && kv0 + kKeys - 1 > qw0) {
        const int32_t qi = qw0 + (int32_t)r32;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int32_t key = kv0 + 32 * sub + (e & 3) + 8 * (e >> 2) + 4 * (int32_t)h;
            if (key > qi) s[sub][e] = -INFINITY;
          }
      }
      // ---- online softmax (base 2, scores scaled by c = scale * log2 e)
      float mx = s[0][0];
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int e = 0; e < 16; ++e) mx = fmaxf(mx, s[sub][e]);
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      // Deferred rescale: while no row's maximum grows by more than 2^kDefer the running maximum stays where it is (the
      // probabilities of this tile are then bounded by 2^kDefer instead of 1, harmless in bf16 / fp32) and the 64
      // accumulator registers are not touched. The branch is wave-uniform, the new maximum is per row, and it is taken
      // before this tile's P exists, so everything at the old scale (O, l) is rescaled exactly once and nothing at the
      // new scale is.
      const float mxs = mx * a.c;
      if (!__all(mxs - m_run <= kDefer)) {
        // only the rows that grew move: a row's result never depends on what its neighbours in the wave hold
        const float m_new = mxs - m_run > kDefer ? mxs : m_run;
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        l_run *= alpha;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int e = 0; e < 16; ++e) o[i][e] *= alpha;
      }
      float rs = 0.0f;
      bf16x8 pf[2][2];
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[sub][e], a.c, -m_run));
          rs += p;
          pf[sub][e >> 3][e & 7] = (__bf16)p;
        }
      l_run += rs;
      // ---- O^T += V^T P
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int imm = (32 * sub + 16 * u) * kVPitch + 32 * db * 2;
            const bf16x8 vf = load_vt(vbuf, vt_lane_off, imm);
            o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[sub][u], o[db], 0, 0, 0);
          }
    }
    if (more) stage_store(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: normalise, bf16, per-wave LDS tile [32 rows][128 d], then whole rows out
  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = 1.0f / l_tot;
  unsigned char* obuf = smem + wave * (kRowsPerWave * kOPitch);
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      u32x2 w;
      w.x = pack2<bf16_t>(o[db][4 * rr + 0] * inv, o[db][4 * rr + 1] * inv);
      w.y = pack2<bf16_t>(o[db][4 * rr + 2] * inv, o[db][4 * rr + 3] * inv);
      *reinterpret_cast<u32x2*>(obuf + r32 * kOPitch + (32 * db + 8 * rr + 4 * h) * 2) = w;
    }
  __syncthreads();
  if (!wave_valid) return;
  float sc = 1.0f, of = 0.0f;
  if (a.codes) {
    sc = a.scale[0];
    of = a.offset ? rne(a.offset[0]) : 0.0f;
  }
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const uint32_t slot = it * 64 + lane;
    const uint32_t row = slot >> 3, seg = slot & 7;
    Chunk<bf16_t, 16> z;
    z.load(reinterpret_cast<const bf16_t*>(obuf + row * kOPitch + seg * 32));
    const size_t at = (((size_t)b * a.S + (size_t)(qw0 + (int32_t)row)) * a.H + head) * kD + seg * 16;
    if (a.ctx) z.store(reinterpret_cast<bf16_t*>(a.ctx) + at);
    if (a.codes) {
      float x[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) x[i] = z.get(i);
      Chunk<int8_t, 16> y;
      quantize_chunk_to_bytes<16>(x, sc, of, a.lo, a.hi, y);
      y.store(a.codes + at);
    }
  }
}

}  // namespace
}  // namespace ffq

using namespace ffq;

extern "C" int ffq_attention(const void* q, const void* k, const void* v, int dt, int64_t batch, int64_t seq_len,
                             int64_t q_heads, int64_t kv_heads, int64_t head_dim, double softmax_scale, int causal,
                             void* ctx_out, int8_t* codes_out, const float* out_scale, const float* out_offset,
                             double out_num_bits, const void* q_cos, const void* q_sin, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (batch < 0 || seq_len < 0 || q_heads <= 0 || kv_heads <= 0) return fail(FFQ_ERR_ARG, "bad extent");
  if (dt != FFQ_BF16) return fail(FFQ_ERR_DTYPE, "attention is built for bf16 activations");
  if (head_dim != kD) return fail(FFQ_ERR_DTYPE, "attention is built for head_dim 128");
  if (seq_len % kKeys != 0) return fail(FFQ_ERR_DTYPE, "attention needs seq_len %% 64 == 0");
  if (q_heads % kv_heads != 0) return fail(FFQ_ERR_ARG, "q_heads must be a multiple of kv_heads");
  if (batch == 0 || seq_len == 0) return FFQ_OK;
  if (!q || !k || !v || (!ctx_out && !codes_out)) return fail(FFQ_ERR_ARG, "NULL buffer");
  if (codes_out && !out_scale) return fail(FFQ_ERR_ARG, "codes need a scale");
  if (codes_out && !(out_num_bits >= 1.0 && out_num_bits <= 8.0 && out_num_bits == (double)(int)out_num_bits))
    return fail(FFQ_ERR_ARG, "codes need an integral bit-width in 1..8");
  if (!aligned16(q) || !aligned16(k) || !aligned16(v) || (ctx_out && !aligned16(ctx_out)) || (codes_out && !aligned16(codes_out)))
    return fail(FFQ_ERR_ARG, "buffers must be 16-byte aligned");
  if (batch * seq_len * q_heads * head_dim >= ((int64_t)1 << 40)) return fail(FFQ_ERR_ARG, "too many elements for one launch");
  if ((q_cos == nullptr) != (q_sin == nullptr)) return fail(FFQ_ERR_ARG, "q_cos and q_sin come together");
  if (q_cos && (!aligned16(q_cos) || !aligned16(q_sin))) return fail(FFQ_ERR_ARG, "buffers must be 16-byte aligned");
  AttnArgs a;
  a.q_cos = static_cast<const uint16_t*>(q_cos); a.q_sin = static_cast<const uint16_t*>(q_sin);
  a.q = static_cast<const uint16_t*>(q); a.k = static_cast<const uint16_t*>(k); a.v = static_cast<const uint16_t*>(v);
  a.ctx = static_cast<uint16_t*>(ctx_out); a.codes = codes_out; a.scale = out_scale; a.offset = out_offset;
  a.B = (int32_t)batch; a.S = (int32_t)seq_len; a.H = (int32_t)q_heads; a.HKV = (int32_t)kv_heads;
  a.nqb = (int32_t)((seq_len + kQBlock - 1) / kQBlock);
  a.c = (float)(softmax_scale * 1.4426950408889634);
  const double half = ldexp(1.0, (int)out_num_bits - 1);
  a.lo = codes_out ? (float)-half : 0.0f;
  a.hi = codes_out ? (float)(half - 1.0) : 0.0f;
  const int64_t blocks = (int64_t)a.nqb * q_heads * batch;
  if (blocks >= ((int64_t)1 << 31)) return fail(FFQ_ERR_ARG, "too many workgroups");
  static uint64_t once_causal = 0, once_full = 0;  // 72 KiB of dynamic LDS: above the 64 KiB a kernel gets without asking
  if (causal) ensure_dynamic_lds(&once_causal, reinterpret_cast<const void*>(attention_fwd_kernel<true>), kLdsBytes);
  else ensure_dynamic_lds(&once_full, reinterpret_cast<const void*>(attention_fwd_kernel<false>), kLdsBytes);
  const dim3 grid((unsigned)blocks), block(kWaves * 64);
  if (causal) attention_fwd_kernel<true><<<grid, block, kLdsBytes, s>>>(a);
  else attention_fwd_kernel<false><<<grid, block, kLdsBytes, s>>>(a);
  return check_launch("attention_fwd_kernel");
}

// ffq_linear.hip — A6: W8A8 linear on the int8 matrix cores of gfx950.
//
// Replaces fallback.linear, src/fastforward/_gen/fallback.py:77-112: the reference dequantizes
// the activation codes and the weight codes into bf16 tensors (two extra HBM round trips, 3 B/elem
// of weight traffic each forward), runs a float GEMM and optionally re-quantizes. Here the integer
// codes feed the int8 MFMA directly, accumulate exactly in int32, and the affine parameters are
// applied once per output element in the epilogue:
//
//   y[m,n] = sx[m'] * sw[n'] * ( acc[m,n] + ox[m'] * rowsum_w[n] + ow[n'] * rowsum_x[m]
//                                + K * ox[m'] * ow[n'] )  (+ bias[n])
//
// with acc = sum_k xq[m,k] * wq[n,k], ox / ow = round_half_even(offset) (A2), and the row sums
// produced by one-pass int8 reductions only when the corresponding offset exists. With an output
// quantizer (fallback.py:110-111) the epilogue rounds y to the dtype the linear would have returned and
// applies A1 to it: the codes leave the launch, the real-valued tensor never visits HBM.
//
// Two kernels (round 3: the superseded generations — 256^2 single-phase, 64-byte-row ping-pong, non-persistent
// full-line, 32x32x32 persistent, one-wave-per-SIMD — are gone; their measurements are in DESIGN.md §4):
//   * w8a8_gemm256fq_kernel — the product path: persistent, 256 x 256 tile, 128 k-bytes per super-step,
//     LDS-DMA staging, ping-pong wave groups, v_mfma_i32_16x16x64_i8;
//   * w8a8_gemm_kernel      — 128 x 128 x 64, register-staged: small problems and K % 128 != 0.
// Layout: xq [M,K] and wq [N,K] are both K-contiguous, which is the operand order MFMA wants: a lane supplies
// the same 16 consecutive k-bytes of a row to both operands, so no transpose is ever needed.
#include "ffq_affine.h"
#include "ffq_common.h"
#include "ffq_vec.h"
#include "ffq_extrema.h"
#include "ffq_silu.h"

#include <math.h>

#include <type_traits>

namespace ffq {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int kTileBytes = BM * BK;  // 8 KiB per operand per stage

struct LinearArgs {
  const int8_t* xq;
  const int8_t* wq;
  const float* x_scale; const float* x_offset;
  const float* w_scale; const float* w_offset;
  const int32_t* rowsum_x; const int32_t* rowsum_w;
  int rowsum_w_inside;  // tail kernel: no side reduction — every block sums the codes of its own 128 weight rows as they pass through its registers
  // persistent kernel with a weight offset: 0 = every rounded weight offset is zero (rowsum_x was not produced and is
  // not read); written by offsets_nonzero_kernel ahead of the launch — no host read of the offsets anywhere
  const int32_t* woff_live;
  const void* bias; int bias_dt;
  void* out; int out_dt;
  const float* out_scale; const float* out_offset;
  float out_lo, out_hi;
  int y_dt;  // re-quantizing epilogue: dtype of the real-valued result the output quantizer sees (nn/linear.py:32-39)
  int x_per_row, w_per_row;
  int M, N, K;
  int tiles_m, tiles_n;
  int group_m;  // row tiles per group of the persistent kernel's tile walk (A panels shared by a group's column tiles)
  int group_cols;  // 1: the groups are `group_m` COLUMN tiles x all row tiles
  // MLP mode (gate and up projections in one launch): the second weight matrix
  const int8_t* wq2; const float* w_scale2; const int32_t* rowsum_w2;
  // batched matmul (ffq_bmm_w8a8, tail kernel only): blockIdx.y selects the matrix pair; element strides between consecutive matrices
  int64_t batch_x, batch_w, batch_out;
  // gated output (ffq_linear_w8a8_gated): the bf16 [M, N] tensor whose silu multiplies this linear's bf16 result
  const bf16_t* gate;
  ExtremaSink extrema;  // words == nullptr: not wanted. [min, max] of the gated product (ffq_extrema.h)
  // a launch of a device-side either / or (ffq_mlp_gate_up_w8a8_estimating): it runs iff *run_if == run_when (nullptr: always)
  const int32_t* run_if; int run_when;
  // the activation codes of an EARLIER quantizer of the same tensor (ffq_affine.h): read instead of `xq` when this linear's input
  // quantizer turns out to hold the same parameters (WOFF and GATED instantiations of the persistent kernel only: range estimation)
  EarlierCodes earlier;
  // several weight matrices side by side along N in ONE launch (ffq_linear_w8a8_multi: q / k / v of an attention block): the codes, scales
  // and row sums are ONE [N, K] / [N] run (matrix after matrix), only the outputs are separate tensors. Columns [0, seg_start[0]) belong
  // to `out`, [seg_start[0], seg_start[1]) to seg_out[0], the rest to seg_out[1]; INT32_MAX: absent. Every boundary is a multiple of 256
  // (a tile belongs to one matrix). Plain bf16 / f16 / f32 output of the persistent kernel only.
  int seg_start[2];
  void* seg_out[2];
};

// The value the linear would have returned in dtype `y_dt` (one rounding), as fp32
__device__ __forceinline__ float round_to_dt(float y, int y_dt) {
  if (y_dt == FFQ_BF16) return bf16_bits_to_f32(f32_to_bf16_bits(y));
  if (y_dt == FFQ_F16) return (float)(_Float16)y;
  return y;
}

__device__ __forceinline__ uint32_t swizzled(uint32_t row, uint32_t slot) {
  return row * BK + ((slot ^ ((row >> 2) & 3u)) << 4);
}

// 16 B of row `row` at byte column `kbyte` of a K-contiguous int8 matrix, zero outside the matrix.
__device__ __forceinline__ u32x4 load_slot(const int8_t* base, int row, int rows, int kbyte, int K) {
  u32x4 v = {0u, 0u, 0u, 0u};
  if (row < rows && kbyte < K) v = *reinterpret_cast<const u32x4*>(base + (size_t)row * K + kbyte);
  return v;
}

template <typename TOut>
__device__ __forceinline__ void store_out(TOut* p, float v);
template <> __device__ __forceinline__ void store_out<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void store_out<bf16_t>(bf16_t* p, float v) { *p = from_f32<bf16_t>(v); }
template <> __device__ __forceinline__ void store_out<f16_t>(f16_t* p, float v) { *p = from_f32<f16_t>(v); }
template <> __device__ __forceinline__ void store_out<int8_t>(int8_t* p, float v) { *p = from_f32<int8_t>(v); }

// -------------------------------------------------------------------------------------------------
// The tail kernel: block tile 128 x 128 x 64, 4 wavefronts (2 x 2), each owning 64 x 64 = 2 x 2 tiles of
// v_mfma_i32_32x32x32_i8; register-staged, double-buffered LDS with a 16-byte-slot XOR swizzle
// (slot ^= (row >> 2) & 3) that makes every ds_read_b128 lane group hit 16 distinct slots; any K % 16 == 0,
// any M / N. blockIdx is remapped so that consecutive tiles of one weight panel stay on one XCD's L2.
// -------------------------------------------------------------------------------------------------
template <typename TOut, bool REQUANT>
__global__ __launch_bounds__(256) void w8a8_gemm_kernel(LinearArgs a) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[2][2][kTileBytes];
  if (gridDim.y > 1) {  // one matrix pair of a batch per blockIdx.y (block-uniform): shift every per-matrix pointer
    const int64_t b = blockIdx.y;
    a.xq += b * a.batch_x;
    a.wq += b * a.batch_w;
    a.out = static_cast<uint8_t*>(a.out) + b * a.batch_out * (int64_t)sizeof(TOut);
    if (a.rowsum_x) a.rowsum_x += b * a.M;
    if (a.rowsum_w) a.rowsum_w += b * a.N;
  }

  // XCD-aware tile order: blocks b, b+8, b+16, ... share an XCD (observed dispatch: XCD = b % 8),
  // give each XCD a contiguous range of tiles so a weight panel is fetched into one L2 only.
  const uint32_t nblk = gridDim.x;
  const uint32_t xcd = blockIdx.x & 7u, slot_in_xcd = blockIdx.x >> 3;
  const uint32_t q = nblk >> 3, r = nblk & 7u;
  const uint32_t tile_id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot_in_xcd;
  // n-major inside a group of tiles_m rows: neighbours share the weight panel
  const int tn = tile_id / a.tiles_m, tm = tile_id % a.tiles_m;
  const int m0 = tm * BM, n0 = tn * BN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // staging map: 512 slots of 16 B per operand tile, two per lane
  const int s_row0 = tid >> 2, s_slot = tid & 3;          // rows 0..63
  const int s_row1 = s_row0 + 64;                          // rows 64..127

  v16i acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0;

  const int ksteps = (a.K + BK - 1) / BK;
  u32x4 ra0, ra1, rb0, rb1;
  // rowsum_w_inside: sum_k wq[n, k] of the block's own weight rows (the zero-point term ox * sum_k wq) from the staging registers —
  // 8 byte-sum instructions per 64-deep step beside 8 MFMAs, instead of a launch of its own ahead of this one (a small eager
  // linear costs the host one launch less: bench.py host_us_per_op). Rows past N and bytes past K were loaded as zeros.
  int rs0 = 0, rs1 = 0;
  auto add_rowsums = [&]() {
    if (a.rowsum_w_inside) {
      const uint32_t w0[4] = {rb0.x, rb0.y, rb0.z, rb0.w}, w1[4] = {rb1.x, rb1.y, rb1.z, rb1.w};
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        rs0 = __builtin_amdgcn_sdot4((int)w0[d], 0x01010101, rs0, false);
        rs1 = __builtin_amdgcn_sdot4((int)w1[d], 0x01010101, rs1, false);
      }
    }
  };
  auto fetch = [&](int kt) {
    const int kb = kt * BK + s_slot * 16;
    ra0 = load_slot(a.xq, m0 + s_row0, a.M, kb, a.K);
    ra1 = load_slot(a.xq, m0 + s_row1, a.M, kb, a.K);
    rb0 = load_slot(a.wq, n0 + s_row0, a.N, kb, a.K);
    rb1 = load_slot(a.wq, n0 + s_row1, a.N, kb, a.K);
  };
  auto stash = [&](int stage) {
    *reinterpret_cast<u32x4*>(&lds[stage][0][swizzled(s_row0, s_slot)]) = ra0;
    *reinterpret_cast<u32x4*>(&lds[stage][0][swizzled(s_row1, s_slot)]) = ra1;
    *reinterpret_cast<u32x4*>(&lds[stage][1][swizzled(s_row0, s_slot)]) = rb0;
    *reinterpret_cast<u32x4*>(&lds[stage][1][swizzled(s_row1, s_slot)]) = rb1;
  };

  fetch(0);
  add_rowsums();
  stash(0);
  __syncthreads();

  const uint32_t frag_row = lane & 31, frag_g = lane >> 5;
  for (int kt = 0; kt < ksteps; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < ksteps) fetch(kt + 1);  // global loads fly under the MFMAs below
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      v4i fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const uint32_t row = wm * 64 + i * 32 + frag_row;
        fa[i] = *reinterpret_cast<const v4i*>(&lds[cur][0][swizzled(row, kk * 2 + frag_g)]);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const uint32_t row = wn * 64 + j * 32 + frag_row;
        fb[j] = *reinterpret_cast<const v4i*>(&lds[cur][1][swizzled(row, kk * 2 + frag_g)]);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < ksteps) { add_rowsums(); stash(cur ^ 1); }
    __syncthreads();
  }
  __shared__ int rowsum_s[BN];
  if (a.rowsum_w_inside) {  // the four lanes that staged a row's four 16-byte slots meet; block-uniform branch
    rs0 += __shfl_xor(rs0, 1, 64); rs0 += __shfl_xor(rs0, 2, 64);
    rs1 += __shfl_xor(rs1, 1, 64); rs1 += __shfl_xor(rs1, 2, 64);
    if (s_slot == 0) { rowsum_s[s_row0] = rs0; rowsum_s[s_row1] = rs1; }
    __syncthreads();
  }

  // epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)
  TOut* out = static_cast<TOut*>(a.out);
  const float kf = (float)a.K;
  float oscale = 1.0f, ooff = 0.0f;
  if constexpr (REQUANT) {
    oscale = a.out_scale[0];
    ooff = a.out_offset ? rne(a.out_offset[0]) : 0.0f;
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + wn * 64 + j * 32 + (lane & 31);
    if (n >= a.N) continue;
    const float sw = a.w_scale[a.w_per_row ? n : 0];
    const float ow = a.w_offset ? rne(a.w_offset[a.w_per_row ? n : 0]) : 0.0f;
    const float rsw = a.rowsum_w_inside ? (float)rowsum_s[wn * 64 + j * 32 + (lane & 31)] : a.rowsum_w ? (float)a.rowsum_w[n] : 0.0f;
    const float bias = a.bias ? (float)load_any(a.bias, a.bias_dt, n) : 0.0f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        if (m >= a.M) continue;
        const float sx = a.x_scale[a.x_per_row ? m : 0];
        const float ox = a.x_offset ? rne(a.x_offset[a.x_per_row ? m : 0]) : 0.0f;
        const float rsx = a.rowsum_x ? (float)a.rowsum_x[m] : 0.0f;
        float v = (float)acc[i][j][e];
        v = v + ox * rsw;
        v = v + ow * rsx;
        v = v + kf * ox * ow;
        float y = (sx * sw) * v;
        if (a.bias) y = y + bias;
        if constexpr (REQUANT) {
          // the output quantizer sees the linear's result in the dtype the float GEMM would have returned (fallback.py:110-111)
          y = round_to_dt(y, a.y_dt);
          float qv = rne(y / oscale - ooff);
          qv = clamp_nan(qv, a.out_lo, a.out_hi);
          store_out<TOut>(out + (size_t)m * a.N + n, qv);
        } else {
          store_out<TOut>(out + (size_t)m * a.N + n, y);
        }
      }
    }
  }
}

// -------------------------------------------------------------------------------------------------
// The persistent kernel ("fq"). 256 x 256 output tile per block, 8 wavefronts (2 x 4), each owning 128 x 64 =
// 8 x 4 accumulator tiles of v_mfma_i32_16x16x64_i8 (4 registers each, 128 in all).
//   * Full-line staging: a super-step is 128 k-bytes of 256 + 256 rows = one 64 KiB LDS slot (two slots), so every
//     global_load_lds_dwordx4 moves 8 whole 128-byte cache lines. The bank swizzle slot ^= (row >> 1) & 7 sits on the
//     per-lane SOURCE address (the LDS side of an LDS-DMA is lane-linear) and, identically, on the ds_read address.
//   * DMA sources are (wave-uniform 64-bit tile base) + (32-bit lane offset): the k offset of a super-step goes into the
//     uniform part and the instruction takes its saddr form — no VALU address arithmetic in the loop. The lane offset is
//     passed through an empty asm right before each issue: that keeps hipcc from hoisting its zero-extension out of the
//     loop (which turns the address into a 64-bit VGPR add per piece); the instruction itself is the compiler's builtin,
//     so m0 is set by the compiler (round 2 spelled the instruction out in inline assembly with an m0 clobber, which the
//     compiler flags as a reserved register). The base is per TILE, so operands of any size are covered.
//   * Ping-pong wave groups: a phase is a LOAD segment (the ds_read_b128 of one fragment set) and an MFMA cluster
//     (16 MFMAs under s_setprio 1) separated by raw s_barriers; waves 4-7 (the second wave of every SIMD) run one
//     barrier interval behind waves 0-3, so one wave of a SIMD feeds the matrix pipe while its partner reads LDS.
//     A phase = one 64-byte k-chunk (kq = phase / 2) x one half of the rows: even phases read the 4 weight fragments of
//     the chunk and 4 activation fragments, odd phases the other 4 activation fragments. The 8 LDS-DMA pieces per wave
//     of a super-step are issued in the LOAD segments of phases 0 and 1 (by the group that is not computing, behind its
//     fragment reads) and are waited for with vmcnt(0) one phase before their first read
//     (RAW: wait -> barrier -> read; WAR: the target slot's last reads were issued a barrier interval before the pieces).
//   * Persistent tile loop: one block per CU walks its tiles in the order a plain launch would dispatch them (XCD-aware,
//     `group_m` row tiles deep). The K-loop runs straight across tile boundaries: the last iteration of a tile prefetches
//     the first super-step of the next tile, which lands under the epilogue. The epilogue works in the slot the tile has
//     just consumed.
//   * Accumulator layout (the weight is the MFMA's first operand, so a lane owns ONE output row): tile (mi, nj),
//     register t: row m = 16 mi + lane % 16, column n = 16 nj + 4 (lane / 16) + t.
//   * MLP mode (ffq_mlp_gate_up_w8a8): the B tile holds 128 gate rows and the same 128 up rows, interleaved so that a
//     wave's column tiles nj = 0, 1 are gate_proj and nj + 2 up_proj of the SAME 16 output columns; the epilogue forms
//     bf16(silu(bf16(gate))) * bf16(up) and the down_proj input quantizer's codes with the roundings of the three-launch
//     path; silu comes from the LDS table of ffq_silu.h, filled once per launch behind the two operand slots.
//   * Weight offsets (WOFF): the ow * sum_k xq[m,k] and K * ox * ow terms are added in the epilogue from a side
//     reduction over the activation codes, and only when some rounded offset is non-zero (decided on the device).
// Needs K % 128 == 0, K >= 256.
// -------------------------------------------------------------------------------------------------
constexpr int BM2 = 256;
constexpr int GROUP_M2 = 8;

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;
typedef int v4i32 __attribute__((ext_vector_type(4)));

// Epilogue of the plain mode: 32 rows x 64 columns per wave and round go through LDS (144-byte pitch, conflict-free
// ds_write_b64) and leave as whole 128-byte lines with the non-temporal hint (the output is not read again by this launch
// and must not push the operand panels out of L2: gate/up shape +7.7 %, A/B of two builds on one box).
// GATED (ffq_linear_w8a8_gated; bf16 out, whole 128-byte lines): what leaves is bf16(silu(gate)) * y with y this linear's bf16
// value — the MLP's silu(gate_proj(x)) * up_proj(x) (mlp.py:36-38) formed in up_proj's epilogue from gate_proj's stored result,
// with the roundings of the two-tensor chain (ffq_silu.h's table; ops.silu_mul_quantize's product).
template <typename TOut, bool REQUANT, bool WOFF, bool GATED = false>
__device__ __forceinline__ void gemm256_epilogue_slabs16(const LinearArgs& a, v4i32 (&acc)[8][4], uint8_t* scratch, int wave, int lane,
                                                         int wm, int wn, int m0, int n0, [[maybe_unused]] const uint16_t* silu_table = nullptr,
                                                         [[maybe_unused]] float* zext = nullptr /* GATED: this lane's running {min, max, NaN seen} of the product */) {
  TOut* out = static_cast<TOut*>(a.out);
  // the tile's output matrix: its first element, its row pitch and its first column in the launch's column space (tile-uniform selects)
  int out_n = a.N, col0 = 0;
  if (!GATED && !WOFF && a.seg_start[0] != INT32_MAX) {  // (several outputs: the plain launch only; the either / or instantiations have no register to spare)
    if (n0 >= a.seg_start[1]) { out = static_cast<TOut*>(a.seg_out[1]); col0 = a.seg_start[1]; out_n = a.N - col0; }
    else if (n0 >= a.seg_start[0]) { out = static_cast<TOut*>(a.seg_out[0]); col0 = a.seg_start[0]; out_n = (a.seg_start[1] == INT32_MAX ? a.N : a.seg_start[1]) - col0; }
    else out_n = a.seg_start[0];
  }
  float oscale = 1.0f, ooff = 0.0f;
  if constexpr (REQUANT) {
    oscale = a.out_scale[0];
    ooff = a.out_offset ? rne(a.out_offset[0]) : 0.0f;
  }
  constexpr int ROW_BYTES = 144;
  constexpr int WAVE_BYTES = 32 * ROW_BYTES + 4 * 64 * 4 + 3 * 128 * 4;
  static_assert(8 * WAVE_BYTES <= (BM2 + 256) * 128, "the epilogue scratch must fit the consumed operand slot");
  uint8_t* region = scratch + wave * WAVE_BYTES;
  float* colp = reinterpret_cast<float*>(region + 32 * ROW_BYTES);  // [4][64]: weight scale, weight row sum, bias, weight offset
  // [3][128]: activation scale, rounded activation offset, activation row sum of the wave's 128 rows. Through LDS, not registers:
  // 24 preloaded registers beside the 128 accumulators made every WOFF / REQUANT instantiation spill (round 3: 35-87 VGPRs)
  float* rowp = colp + 4 * 64;
  const int r16 = lane & 15, g4 = lane >> 4;
  const int wave_n0 = n0 + wn * 64;
  const int wave_m0 = m0 + wm * 128;
  // whole-line stores need 16-byte aligned rows: 2-byte containers through the slab, anything else element stores
  const bool lds_path = sizeof(TOut) == 2 && (out_n & 7) == 0 && wave_n0 + 64 <= a.N;
  bool woff_live = false;
  if constexpr (WOFF) woff_live = *a.woff_live != 0;
  // GATED: the gate lines of slab round i + 1 are fetched while round i is computed (round 0's here, ahead of the parameter
  // loads): a load issued where it is used exposes a trip to HBM per round and line batch (+227 us on the gate / up shape)
  [[maybe_unused]] u32x4 gate_next[4];
  [[maybe_unused]] auto fetch_gate = [&](int i) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int c = lane + 64 * t;
      const int row = c >> 3, seg = c & 7;
      int mm = wave_m0 + i * 32 + row;
      mm = mm < a.M ? mm : a.M - 1;
      gate_next[t] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const uint8_t*>(a.gate) + ((size_t)mm * a.N + wave_n0) * 2 + seg * 16);
    }
  };
  if constexpr (GATED) {
    if (lds_path) fetch_gate(0);
  }
  {
    int n = wave_n0 + lane;
    n = n < a.N ? n : a.N - 1;
    colp[lane] = a.w_scale[a.w_per_row ? n : 0];
    colp[64 + lane] = a.rowsum_w ? (float)a.rowsum_w[n] : 0.0f;
    colp[128 + lane] = a.bias ? (float)load_any(a.bias, a.bias_dt, n) : 0.0f;
    if constexpr (WOFF) colp[192 + lane] = woff_live ? rne(a.w_offset[a.w_per_row ? n : 0]) : 0.0f;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private: only this wave's own writes
  // every global load of the epilogue BEFORE its first store: a load inside the slab loop makes the compiler wait with
  // vmcnt(0), i.e. for the previous slab's global stores too (vmcnt counts stores on gfx9)
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int r = lane + 64 * h;
    int m = wave_m0 + r;
    m = m < a.M ? m : a.M - 1;
    rowp[r] = a.x_scale[a.x_per_row ? m : 0];
    rowp[128 + r] = a.x_offset ? rne(a.x_offset[a.x_per_row ? m : 0]) : 0.0f;
    if constexpr (WOFF) rowp[256 + r] = woff_live ? (float)a.rowsum_x[m] : 0.0f;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const float kf = (float)a.K;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int mi = 2 * i + hh;
      const int m = wave_m0 + mi * 16 + r16;
      const bool m_ok = m < a.M;
      const float sx = rowp[mi * 16 + r16], ox = rowp[128 + mi * 16 + r16];
      [[maybe_unused]] float rsx = 0.0f;
      if constexpr (WOFF) rsx = rowp[256 + mi * 16 + r16];
#pragma unroll
      for (int nj = 0; nj < 4; ++nj) {
        const int nb = nj * 16 + 4 * g4;
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        const f32x4 sw4 = *reinterpret_cast<const f32x4*>(colp + nb);
        const f32x4 rs4 = *reinterpret_cast<const f32x4*>(colp + 64 + nb);
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(colp + 128 + nb);
        [[maybe_unused]] f32x4 ow4 = {0.0f, 0.0f, 0.0f, 0.0f};
        if constexpr (WOFF) {
          // re-read per row tile through a pointer the compiler cannot see through: hoisted out of the (unrolled) row-tile loop the
          // four column-parameter arrays are 64 registers next to the 128 accumulators — the last two spilling instantiations
          const float* owp = colp + 192 + nb;
          asm volatile("" : "+v"(owp));
          ow4 = *reinterpret_cast<const f32x4*>(owp);
        }
        float y[4];
        // two columns per VALU instruction (v_pk_mul_f32 / v_pk_add_f32: each half is the scalar instruction's IEEE result; the
        // epilogue's arithmetic is paid in full, nothing overlaps it)
#pragma unroll
        for (int t = 0; t < 4; t += 2) {
          const fq_f32x2 A = {(float)acc[mi][nj][t], (float)acc[mi][nj][t + 1]};
          const fq_f32x2 OX = {ox, ox}, SX = {sx, sx};
          fq_f32x2 V = A + OX * fq_f32x2{rs4[t], rs4[t + 1]};
          if constexpr (WOFF) {  // the order of the tail kernel's terms
            const fq_f32x2 OW = {ow4[t], ow4[t + 1]}, RSX = {rsx, rsx}, KFOX = {kf * ox, kf * ox};
            V = V + OW * RSX;
            V = V + KFOX * OW;
          }
          fq_f32x2 R = (SX * fq_f32x2{sw4[t], sw4[t + 1]}) * V;
          if (a.bias) R = R + fq_f32x2{b4[t], b4[t + 1]};
          float r0 = R.x, r1 = R.y;
          if constexpr (REQUANT) {
            r0 = round_to_dt(r0, a.y_dt);
            r1 = round_to_dt(r1, a.y_dt);
            r0 = clamp_nan(rne(r0 / oscale - ooff), a.out_lo, a.out_hi);
            r1 = clamp_nan(rne(r1 / oscale - ooff), a.out_lo, a.out_hi);
          }
          y[t] = r0;
          y[t + 1] = r1;
        }
        if constexpr (sizeof(TOut) == 2) {
          if (lds_path) {
            u32x2 pk;
            pk.x = pack2<TOut>(y[0], y[1]);
            pk.y = pack2<TOut>(y[2], y[3]);
            *reinterpret_cast<u32x2*>(region + (16 * hh + r16) * ROW_BYTES + nb * 2) = pk;
            continue;
          }
        }
        if (m_ok) {
          const size_t at = (size_t)m * out_n + (wave_n0 - col0) + nb;
#pragma unroll
          for (int t = 0; t < 4; ++t)
            if (wave_n0 + nb + t < a.N) store_out<TOut>(out + at + t, y[t]);
        }
      }
    }
    if (lds_path) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      [[maybe_unused]] u32x4 gate_now[4];
      if constexpr (GATED) {
#pragma unroll
        for (int t = 0; t < 4; ++t) gate_now[t] = gate_next[t];
        if (i < 3) fetch_gate(i + 1);
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int c = lane + 64 * t;
        const int row = c >> 3, seg = c & 7;
        const int mm = wave_m0 + i * 32 + row;
        u32x4 v = *reinterpret_cast<const u32x4*>(region + row * ROW_BYTES + seg * 16);
        if constexpr (GATED) {
          if (mm < a.M) {
            const u32x4 g = gate_now[t];
            const uint32_t gw[4] = {g.x, g.y, g.z, g.w}, uw[4] = {v.x, v.y, v.z, v.w};
            uint32_t sw[4], bad = 0, zw[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) sw[q] = silu_pair_lookup(gw[q], silu_table, bad);
            if (silu_any_outside(bad)) {
#pragma unroll
              for (int q = 0; q < 4; ++q) sw[q] = silu_pair_patch(gw[q], sw[q]);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)  // bf16 x bf16 -> fp32 exactly, one rounding to bf16: the product tensor of the two-tensor chain
              zw[q] = pack2<bf16_t>(__builtin_bit_cast(float, sw[q] << 16) * __builtin_bit_cast(float, uw[q] << 16),
                                    __builtin_bit_cast(float, sw[q] & 0xFFFF0000u) * __builtin_bit_cast(float, uw[q] & 0xFFFF0000u));
            v.x = zw[0]; v.y = zw[1]; v.z = zw[2]; v.w = zw[3];
            if (a.extrema.words) {
              float zmn = zext[0], zmx = zext[512], znan = zext[1024];
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const float z0 = __builtin_bit_cast(float, zw[q] << 16), z1 = __builtin_bit_cast(float, zw[q] & 0xFFFF0000u);
                zmn = __builtin_fminf(zmn, __builtin_fminf(z0, z1));
                zmx = __builtin_fmaxf(zmx, __builtin_fmaxf(z0, z1));
                znan = (z0 != z0 || z1 != z1) ? 1.0f : znan;
              }
              zext[0] = zmn; zext[512] = zmx; zext[1024] = znan;
            }
          }
        }
#ifdef FFQ_ABLATE_STORES  // timing-only builds: everything of the epilogue but its global stores
        if (mm < a.M && v.x == 0x12345678u && v.y == 0x9abcdef0u && v.z == 0x0fedcba9u)
#else
        if (mm < a.M)
#endif
          __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(reinterpret_cast<uint8_t*>(out) + ((size_t)mm * out_n + (wave_n0 - col0)) * 2 + seg * 16));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the slab is re-written by the next i
    }
  }
}

// Epilogue of the MLP mode: column tiles nj = 0, 1 hold gate_proj and nj + 2 up_proj of the same 16 output columns;
//   z = bf16(silu(bf16(gate))) * bf16(up)  (bf16),  codes = A1(z; out_scale, out_offset)
// is formed in registers with exactly the roundings of the three-launch path (GEMM epilogue -> bf16 tensors ->
// silu_mul_quantize_kernel), goes through ONE block-wide LDS tile [256][128 B] and leaves as full 128-byte lines of int8
// codes: a quarter of the bytes of one bf16 projection, instead of two.
__device__ __forceinline__ void mlp_epilogue16_body(const LinearArgs& a, v4i32 (&acc)[8][4], int (&rsw)[2], uint8_t* lds2, int wave,
                                                    int lane, int wm, int wn, int m0, int n0, const uint16_t* silu_table) {
  constexpr int PITCH = 144;  // 128 B of codes + 16 B pad
  const float sx = a.x_scale[0];
  const float ox = a.x_offset ? rne(a.x_offset[0]) : 0.0f;
  const float so = a.out_scale[0];
  const float oo = a.out_offset ? rne(a.out_offset[0]) : 0.0f;
  const Divider<1> div(so);
  __syncthreads();  // every wave is done with the operand ring
  float* rs_lds = reinterpret_cast<float*>(lds2 + 256 * PITCH) + wave * 64;
  if (lane < 32) {
    rs_lds[lane] = (float)rsw[0];
    rs_lds[32 + lane] = (float)rsw[1];
  }
  const int r16 = lane & 15, g4 = lane >> 4;
  const int col0 = n0 + wn * 32;  // this wave's 32 output columns
  // Every VALU instruction of this epilogue is paid in full — it does not hide under another wave's MFMAs (round 2: an ablation
  // without the division and the window test, 10 instructions per element, made the launch 2.9 % faster) — so the arithmetic
  // runs on PAIRS of columns (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32: each half is the scalar instruction's IEEE result):
  //  * y = (sx sw[n]) (acc + ox rowsum[n]) with both column constants formed once per column pair;
  //  * the output quantizer is ffq_affine.h's packed form (Markstein quotient without a window test per element, clamp, then
  //    round + convert as one magic add, bytes gathered with v_perm_b32) and its NaN self-check per 16 codes; a lane whose
  //    check fails (an Inf / NaN product, a scale outside the Markstein window) redoes those 16 codes with the reference chain
  //    (IEEE division, round, convert, clamp) — same codes either way, as tests/test_gemm_gpu.py / test_fullsize_gpu.py assert.
  const fq_f32x2 SO = {so, so}, RO = {div.r, div.r}, OO = {oo, oo}, MAGIC = {12582912.0f, 12582912.0f};
  const float lo = a.out_lo, hi = a.out_hi;
#pragma unroll
  for (int nj = 0; nj < 2; ++nj) {
    const int cb = 16 * nj + 4 * g4;  // this lane's 4 columns: col0 + cb + (0..3)
    fq_f32x2 CG[2], OG[2], CU[2], OU[2];
#pragma unroll
    for (int t = 0; t < 4; t += 2) {
      const int n = col0 + cb + t;  // N % 128 == 0: always inside
      CG[t >> 1] = fq_f32x2{sx * a.w_scale[n], sx * a.w_scale[n + 1]};
      CU[t >> 1] = fq_f32x2{sx * a.w_scale2[n], sx * a.w_scale2[n + 1]};
      OG[t >> 1] = fq_f32x2{ox * rs_lds[cb + t], ox * rs_lds[cb + t + 1]};
      OU[t >> 1] = fq_f32x2{ox * rs_lds[32 + cb + t], ox * rs_lds[32 + cb + t + 1]};
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      // bf16(silu(bf16 gate)) of the 8 pairs of four row tiles: the table reads back to back, one branch for the window
      uint32_t wg[4][2], ws[4][2];
      uint32_t bad = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int mi = 4 * half + q;
#pragma unroll
        for (int t = 0; t < 4; t += 2) {
          const fq_f32x2 A = {(float)acc[mi][nj][t], (float)acc[mi][nj][t + 1]};
          const fq_f32x2 G = CG[t >> 1] * (A + OG[t >> 1]);
          wg[q][t >> 1] = pack2<bf16_t>(G.x, G.y);
          ws[q][t >> 1] = silu_pair_lookup(wg[q][t >> 1], silu_table, bad);
        }
      }
      if (__builtin_expect(silu_any_outside(bad), 0)) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) ws[q][h2] = silu_pair_patch(wg[q][h2], ws[q][h2]);
      }
      // z = bf16(silu) * bf16(up) as bf16 pairs (kept for the fallback), codes by the packed chain
      uint32_t zw[4][2], cw[4];
      fq_f32x2 chk = {0.0f, 0.0f};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int mi = 4 * half + q;
        uint32_t b[4];
#pragma unroll
        for (int t = 0; t < 4; t += 2) {
          const fq_f32x2 A = {(float)acc[mi][nj + 2][t], (float)acc[mi][nj + 2][t + 1]};
          const fq_f32x2 U = CU[t >> 1] * (A + OU[t >> 1]);
          uint32_t w = pack2<bf16_t>(U.x, U.y);
          const fq_f32x2 UB = {__builtin_bit_cast(float, w << 16), __builtin_bit_cast(float, w & 0xFFFF0000u)};
          w = ws[q][t >> 1];
          const fq_f32x2 SB = {__builtin_bit_cast(float, w << 16), __builtin_bit_cast(float, w & 0xFFFF0000u)};
          const fq_f32x2 Z = SB * UB;
          w = pack2<bf16_t>(Z.x, Z.y);
          zw[q][t >> 1] = w;
          const fq_f32x2 X = {__builtin_bit_cast(float, w << 16), __builtin_bit_cast(float, w & 0xFFFF0000u)};
          const fq_f32x2 q0 = X * RO;
          const fq_f32x2 q1 = __builtin_elementwise_fma(__builtin_elementwise_fma(-q0, SO, X), RO, q0);
          const fq_f32x2 q2 = __builtin_elementwise_fma(__builtin_elementwise_fma(-q1, SO, X), RO, q1);
          const fq_f32x2 d = q2 - OO;
          chk = chk + d;
          const fq_f32x2 c = {__builtin_amdgcn_fmed3f(d.x, lo, hi), __builtin_amdgcn_fmed3f(d.y, lo, hi)};
          const fq_f32x2 e = c + MAGIC;
          const float e0 = e.x, e1 = e.y;  // (bit_cast of a vector ELEMENT reads element 0 with this hipcc: scalar temporaries)
          b[t] = __builtin_bit_cast(uint32_t, e0);
          b[t + 1] = __builtin_bit_cast(uint32_t, e1);
        }
        cw[q] = __builtin_amdgcn_perm(__builtin_amdgcn_perm(b[3], b[2], 0x0c0c0400u), __builtin_amdgcn_perm(b[1], b[0], 0x0c0c0400u), 0x05040100u);
      }
      const float chk1 = chk.x + chk.y;
      if (__builtin_expect(!div.safe || chk1 != chk1, 0)) {  // the reference chain for these 16 codes
        const int ilo = (int)lo, ihi = (int)hi;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          int c[4];
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const uint32_t w = zw[q][t >> 1];
            const float z = __builtin_bit_cast(float, (t & 1) ? (w & 0xFFFF0000u) : (w << 16));
            const int v = (int)rne(z / so - oo);  // v_cvt_i32_f32: NaN -> 0, the int8 container's value
            c[t] = v < ilo ? ilo : (v > ihi ? ihi : v);
          }
          cw[q] = pack_bytes(c[0], c[1], c[2], c[3]);
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = wm * 128 + (4 * half + q) * 16 + r16;
        *reinterpret_cast<uint32_t*>(lds2 + row * PITCH + wn * 32 + cb) = cw[q];
      }
    }
  }
  __syncthreads();
  int8_t* out = static_cast<int8_t*>(a.out);
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int row = wave * 32 + t * 8 + (lane >> 3), seg = lane & 7;
    const int m = m0 + row;
    const u32x4 v = *reinterpret_cast<const u32x4*>(lds2 + row * PITCH + seg * 16);
#ifdef FFQ_ABLATE_STORES
    if (m < a.M && v.x == 0x12345678u && v.y == 0x9abcdef0u && v.z == 0x0fedcba9u)
#else
    if (m < a.M)
#endif
      __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(out + (size_t)m * a.N + n0 + seg * 16));
  }
}

// The MLP mode WITHOUT the output quantizer (ffq_mlp_gate_up_w8a8_estimating: range estimation, where down_proj's input quantizer is
// what the forward is calibrating): z as above leaves as bf16 — [256 rows][256 B] per tile, through LDS in two passes of 128 rows
// (pitch 272 B: the consumed operand slot holds 34 KiB at a time) as whole 128-byte lines — and every thread keeps the running
// {min, max, NaN seen} of the z it stores (zext, LDS: ffq_extrema.h).
__device__ __forceinline__ void mlp_epilogue16_product(const LinearArgs& a, v4i32 (&acc)[8][4], int (&rsw)[2], uint8_t* lds2, int wave,
                                                       int lane, int wm, int wn, int m0, int n0, const uint16_t* silu_table, float* zext) {
  constexpr int PITCH = 272;  // 256 B of bf16 + 16 B pad
  const float sx = a.x_scale[0];
  const float ox = a.x_offset ? rne(a.x_offset[0]) : 0.0f;
  __syncthreads();  // every wave is done with the operand ring
  float* rs_lds = reinterpret_cast<float*>(lds2 + 128 * PITCH) + wave * 64;
  if (lane < 32) {
    rs_lds[lane] = (float)rsw[0];
    rs_lds[32 + lane] = (float)rsw[1];
  }
  const int r16 = lane & 15, g4 = lane >> 4;
  const int col0 = n0 + wn * 32;  // this wave's 32 output columns
  bf16_t* out = static_cast<bf16_t*>(a.out);
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int nj = 0; nj < 2; ++nj) {
      const int cb = 16 * nj + 4 * g4;  // this lane's 4 columns: col0 + cb + (0..3)
      fq_f32x2 CG[2], OG[2], CU[2], OU[2];
#pragma unroll
      for (int t = 0; t < 4; t += 2) {
        const int n = col0 + cb + t;  // N % 128 == 0: always inside
        CG[t >> 1] = fq_f32x2{sx * a.w_scale[n], sx * a.w_scale[n + 1]};
        CU[t >> 1] = fq_f32x2{sx * a.w_scale2[n], sx * a.w_scale2[n + 1]};
        OG[t >> 1] = fq_f32x2{ox * rs_lds[cb + t], ox * rs_lds[cb + t + 1]};
        OU[t >> 1] = fq_f32x2{ox * rs_lds[32 + cb + t], ox * rs_lds[32 + cb + t + 1]};
      }
      uint32_t wg[4][2], ws[4][2];
      uint32_t bad = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int mi = 4 * half + q;
#pragma unroll
        for (int t = 0; t < 4; t += 2) {
          const fq_f32x2 A = {(float)acc[mi][nj][t], (float)acc[mi][nj][t + 1]};
          const fq_f32x2 G = CG[t >> 1] * (A + OG[t >> 1]);
          wg[q][t >> 1] = pack2<bf16_t>(G.x, G.y);
          ws[q][t >> 1] = silu_pair_lookup(wg[q][t >> 1], silu_table, bad);
        }
      }
      if (__builtin_expect(silu_any_outside(bad), 0)) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) ws[q][h2] = silu_pair_patch(wg[q][h2], ws[q][h2]);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int mi = 4 * half + q;
        u32x2 zw;
#pragma unroll
        for (int t = 0; t < 4; t += 2) {
          const fq_f32x2 A = {(float)acc[mi][nj + 2][t], (float)acc[mi][nj + 2][t + 1]};
          const fq_f32x2 U = CU[t >> 1] * (A + OU[t >> 1]);
          uint32_t w = pack2<bf16_t>(U.x, U.y);
          const fq_f32x2 UB = {__builtin_bit_cast(float, w << 16), __builtin_bit_cast(float, w & 0xFFFF0000u)};
          w = ws[q][t >> 1];
          const fq_f32x2 SB = {__builtin_bit_cast(float, w << 16), __builtin_bit_cast(float, w & 0xFFFF0000u)};
          const fq_f32x2 Z = SB * UB;
          w = pack2<bf16_t>(Z.x, Z.y);
          if (t == 0) zw.x = w; else zw.y = w;
        }
        const int rowl = wm * 64 + q * 16 + r16;  // row of this pass: tile row wm * 128 + half * 64 + q * 16 + r16
        *reinterpret_cast<u32x2*>(lds2 + rowl * PITCH + wn * 64 + cb * 2) = zw;
      }
    }
    __syncthreads();
#pragma unroll 1  // (one 16-byte piece at a time: the other half's 64 accumulator registers are still live)
    for (int t = 0; t < 4; ++t) {  // 128 rows x 16 segments of 16 B over 512 threads
      const int idx = wave * 256 + t * 64 + lane;
      const int rowl = idx >> 4, seg = idx & 15;
      const int m = m0 + (rowl >> 6) * 128 + half * 64 + (rowl & 63);
      const u32x4 v = *reinterpret_cast<const u32x4*>(lds2 + rowl * PITCH + seg * 16);
      if (m < a.M) {  // (rows past the edge of a ragged M hold garbage)
        __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(reinterpret_cast<uint8_t*>(out) + ((size_t)m * a.N + n0) * 2 + seg * 16));
        if (a.extrema.words) {  // on the way out, where the accumulators' registers are not in the way
          const uint32_t vw[4] = {v.x, v.y, v.z, v.w};
          float zmn = zext[0], zmx = zext[512], znan = zext[1024];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float z0 = __builtin_bit_cast(float, vw[q] << 16), z1 = __builtin_bit_cast(float, vw[q] & 0xFFFF0000u);
            zmn = __builtin_fminf(zmn, __builtin_fminf(z0, z1));
            zmx = __builtin_fmaxf(zmx, __builtin_fmaxf(z0, z1));
            znan = (z0 != z0 || z1 != z1) ? 1.0f : znan;
          }
          zext[0] = zmn; zext[512] = zmx; zext[1024] = znan;
        }
      }
    }
    __syncthreads();  // the tile is re-written by the next pass / the slot is the next tile's DMA target
  }
}

template <typename TOut, bool REQUANT, bool MLP, bool WOFF = false, bool GATED = false>
__global__ __launch_bounds__(512, 2) void w8a8_gemm256fq_kernel(LinearArgs a, int total_tiles) {
  constexpr int BN2 = 256, WAVES_N = 4;
  constexpr int BN_OUT = MLP ? 128 : 256;
  constexpr int SLOT_BYTES = (BM2 + BN2) * 128;
  constexpr int B_IMAGE = BM2 * 128;
  extern __shared__ __attribute__((aligned(16))) uint8_t lds2[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  // tile walk: XCD x (= blockIdx % 8) owns a contiguous range of the grouped tile order, its blocks walk it round-robin,
  // i.e. the order in which a non-persistent launch would dispatch them
  const uint32_t nblk = gridDim.x;
  const uint32_t xcd = blockIdx.x & 7u, j_in_xcd = blockIdx.x >> 3;
  const uint32_t blocks_in_xcd = (nblk >> 3) + (xcd < (nblk & 7u) ? 1u : 0u);
  const uint32_t tq = (uint32_t)total_tiles >> 3, tr = (uint32_t)total_tiles & 7u;
  const uint32_t xcd_first = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
  const uint32_t xcd_count = tq + (xcd < tr ? 1u : 0u);
  const int my_tiles = j_in_xcd < xcd_count ? (int)((xcd_count - j_in_xcd + blocks_in_xcd - 1) / blocks_in_xcd) : 0;
  constexpr bool PRODUCT = MLP && !REQUANT;  // the gate/up launch that leaves the bf16 product (and its extrema)
  if constexpr (WOFF || GATED || PRODUCT) {  // (the instantiations a device-side either / or is built from; the forward's are not among them)
    if (a.run_if && *a.run_if != a.run_when) return;
  }
  const int8_t* xq_in_force = a.xq;
  if constexpr (WOFF || GATED) {
    // (read back through readfirstlane: a pointer that is the result of a select on loaded values is a VECTOR value to hipcc — the four
    // A pieces then carry 64-bit per-lane addresses, the K-loop 64-bit vector adds, and the GATED form spills)
    const uint64_t p = reinterpret_cast<uint64_t>(codes_in_force(a.xq, a.x_scale, a.x_offset, a.earlier));
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)p), hi = __builtin_amdgcn_readfirstlane((uint32_t)(p >> 32));
    xq_in_force = reinterpret_cast<const int8_t*>(((uint64_t)hi << 32) | lo);
  }
  if (my_tiles == 0) {
    if constexpr (GATED || PRODUCT) {
      if (a.extrema.words && tid == 0) extrema_publish(a.extrema, 0.0f, 0.0f, false, false, gridDim.x);
    }
    return;
  }
  uint16_t* const silu_table = reinterpret_cast<uint16_t*>(lds2 + 2 * SLOT_BYTES);
  // GATED: the lane's running {min, max, NaN seen} of the product lives in LDS behind the table (three registers held across the
  // K-loop spill next to the 128 accumulators), lane-interleaved: word j of thread t at [j * 512 + t]
  [[maybe_unused]] float* const zext = reinterpret_cast<float*>(lds2 + 2 * SLOT_BYTES + kSiluBytes) + tid;
  if constexpr (GATED || PRODUCT) {
    zext[0] = INFINITY; zext[512] = -INFINITY; zext[1024] = 0.0f;
  }
  if constexpr (MLP || GATED) silu_table_fill(silu_table, (uint32_t)tid, 512u);

  // A contraction depth that is a large power of two puts the same depth of every row of every tile on the same few memory channels,
  // and all 256 CUs walk the depth in step: XCD x starts x/8 of the way in and wraps. Integer sums: the result does not change.
  // Round 6, A/B of two builds, two rounds (profiles/r06_krot_ab.txt): 70B gate/up (K = 8192) 1.461 -> 1.399 ms at 8192 tokens, the
  // gate+up+SiLU launch 2.97 -> 2.81; K = 4096 / 14336 / 28672 +-0.3 % (off there).
  // (not in the GATED instantiations, the device-side either / or of a calibration step: they have no register to spare)
  const int k_rot = (!GATED && a.K >= 8192 && (a.K & (a.K - 1)) == 0) ? (int)(xcd * (uint32_t)(a.K / 128) / 8u) : 0;
  const int d_row = lane >> 3;
  uint32_t a_voff[4], b_voff[4];   // lane offsets inside the tile's rows: < 256 K + 128
  const int8_t* a_base = xq_in_force;  // wave-uniform: first row of the tile
  const int8_t* b_base[4];         // per piece (MLP mode: gate or up matrix)
  int m0 = 0, n0 = 0;
  auto tile_origin = [&](int it, int& tm0, int& tn0) {
    const uint32_t tile_id = xcd_first + j_in_xcd + (uint32_t)it * blocks_in_xcd;
    const uint32_t gm = (uint32_t)a.group_m;
    // groups of `gm` row tiles x all column tiles (the weight is re-streamed per group), or — group_cols — `gm` column tiles x all
    // row tiles (the activations are): the operand a group re-reads in full should be the one that fits the Infinity Cache
    const uint32_t across = a.group_cols ? (uint32_t)a.tiles_m : (uint32_t)a.tiles_n;
    const uint32_t along = a.group_cols ? (uint32_t)a.tiles_n : (uint32_t)a.tiles_m;
    const uint32_t per_group = gm * across;
    const uint32_t group = tile_id / per_group, in_group = tile_id - group * per_group;
    const uint32_t group_size = min(gm, along - group * gm);
    const uint32_t inner = group * gm + in_group % group_size, outer = in_group / group_size;
    tm0 = (int)(a.group_cols ? outer : inner) * BM2;
    tn0 = (int)(a.group_cols ? inner : outer) * BN_OUT;
  };
  // first byte of row `row0` of a K-contiguous matrix, as a value the compiler keeps in SGPRs: there is no scalar 64-bit
  // multiply, so the product is formed in vector registers once per tile and read back — everything the loop derives from it
  // (the k offset of a super-step, the saddr operand) is then scalar arithmetic
  auto row_base = [&](const int8_t* base, int row0) {
    const uint64_t off = (uint64_t)(uint32_t)row0 * (uint64_t)(uint32_t)a.K;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)off), hi = __builtin_amdgcn_readfirstlane((uint32_t)(off >> 32));
    return base + (((uint64_t)hi << 32) | lo);
  };
  auto set_sources = [&](int tm0, int tn0) {
    a_base = row_base(xq_in_force, tm0);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int row = (wave * 4 + c) * 8 + d_row;
      const int d_slot = (lane & 7) ^ ((row >> 1) & 7);
      int ra = row;
      ra = tm0 + ra < a.M ? ra : a.M - 1 - tm0;  // rows past the edge re-read the last row and are never stored
      // (GATED: as a 24-bit multiply-add — hipcc forms row * K + slot with v_mad_u64_u32, a register PAIR per offset and per addend, and
      // this instantiation has none to spare: it spilled four. ra < 256, K < 2^24. The other instantiations keep their allocation.)
      if constexpr (GATED) a_voff[c] = __umul24((uint32_t)ra, (uint32_t)a.K) + (uint32_t)(d_slot * 16);
      else a_voff[c] = (uint32_t)ra * (uint32_t)a.K + d_slot * 16;
      if constexpr (MLP) {
        // rows 32..63 of a wave's 64 come from the up matrix: row & 32 is the same for all lanes of a piece (8 rows per piece)
        const int rb = (row >> 6) * 32 + (row & 31);
        b_base[c] = row_base((((wave * 4 + c) * 8) & 32) ? a.wq2 : a.wq, tn0);
        b_voff[c] = (uint32_t)rb * (uint32_t)a.K + d_slot * 16;
      } else {
        int rb = row;
        rb = tn0 + rb < a.N ? rb : a.N - 1 - tn0;
        b_base[c] = row_base(a.wq, tn0);
        if constexpr (GATED) b_voff[c] = __umul24((uint32_t)rb, (uint32_t)a.K) + (uint32_t)(d_slot * 16);
        else b_voff[c] = (uint32_t)rb * (uint32_t)a.K + d_slot * 16;
      }
    }
  };
  auto issue_a = [&](int ks, int slot, int c0) {
    uint8_t* base = lds2 + slot * SLOT_BYTES;
#pragma unroll
    for (int c = c0; c < c0 + 2; ++c) {
      asm volatile("" : "+v"(a_voff[c]));  // see the header: keeps the saddr form in every unrolled body
      __builtin_amdgcn_global_load_lds((gbl_void_t*)((a_base + ks * 128) + a_voff[c]), (lds_void_t*)(base + (wave * 4 + c) * 1024), 16, 0, 0);
    }
  };
  auto issue_b = [&](int ks, int slot, int c0) {
    uint8_t* base = lds2 + slot * SLOT_BYTES + B_IMAGE;
#pragma unroll
    for (int c = c0; c < c0 + 2; ++c) {
      asm volatile("" : "+v"(b_voff[c]));
      __builtin_amdgcn_global_load_lds((gbl_void_t*)((b_base[c] + ks * 128) + b_voff[c]), (lds_void_t*)(base + (wave * 4 + c) * 1024), 16, 0, 0);
    }
  };

  // fragment byte offsets inside a slot. The swizzle term of a row depends on the row only through lane % 16 (row tiles are
  // 16 rows apart and the term is (row / 2) mod 8), so ONE register per k-chunk and operand serves every row tile: the row tile
  // goes into the instruction's offset field (mi * 2048 bytes). Round 3 kept 24 such registers; with 4 the WOFF / REQUANT
  // instantiations no longer spill (tools/kernel_resources.py).
  const uint32_t r16 = lane & 15, g4 = lane >> 4;
  uint32_t a_off[2], b_off[2];
  {
    const uint32_t arow = wm * 128 + r16, brow = wn * 64 + r16;
#pragma unroll
    for (int kq = 0; kq < 2; ++kq) {
      a_off[kq] = arow * 128 + (((kq * 4 + g4) ^ ((arow >> 1) & 7u)) << 4);
      b_off[kq] = B_IMAGE + brow * 128 + (((kq * 4 + g4) ^ ((brow >> 1) & 7u)) << 4);
    }
  }

  v4i32 acc[8][4];
  v4i fa[4], fb[4];
  auto read_frags = [&](const uint8_t* st, int phase) {  // phase 0..3 of a super-step (compile-time after unrolling)
    const int kq = phase >> 1, mh = phase & 1;
    if (mh == 0) {
#pragma unroll
      for (int nj = 0; nj < 4; ++nj) fb[nj] = *reinterpret_cast<const v4i*>(st + b_off[kq] + nj * 2048);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) fa[q] = *reinterpret_cast<const v4i*>(st + a_off[kq] + (4 * mh + q) * 2048);
  };
  auto cluster = [&](int mh) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int n_ = 0; n_ < 4; ++n_) {
        // snake order: every MFMA shares one operand with its predecessor, also across the row-tile change
        // (MLP mode +1.3-3.2 %, down +1-2.3 %, A/B on one box: operand reads cost power, and power is what limits this loop)
        const int nj = (q & 1) ? 3 - n_ : n_;
        if (mh == 0) acc[q][nj] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fb[nj], fa[q], acc[q][nj], 0, 0, 0);
        else acc[4 + q][nj] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fb[nj], fa[q], acc[4 + q][nj], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(0);
  };

  const int ksuper = a.K / 128;
  int slot = 0;  // slot of the super-step about to be computed
  tile_origin(0, m0, n0);
  set_sources(m0, n0);
  issue_a(k_rot, 0, 0); issue_a(k_rot, 0, 2); issue_b(k_rot, 0, 0); issue_b(k_rot, 0, 2);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  for (int it = 0; it < my_tiles; ++it) {
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
      for (int nj = 0; nj < 4; ++nj)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[mi][nj][e] = 0;
    int nm0 = m0, nn0 = n0;
    const bool has_next = it + 1 < my_tiles;
    if (has_next) tile_origin(it + 1, nm0, nn0);
    if (wm == 1) __builtin_amdgcn_s_barrier();  // the upper group runs one interval behind
    for (int ks = 0; ks < ksuper; ++ks) {
      const uint8_t* st = lds2 + slot * SLOT_BYTES;
      // what the first two clusters fetch into the other slot: the next super-step of this tile, or the first one of
      // the next tile (nothing after the block's last tile: a re-load of this super-step keeps the waits uniform)
      int fetch = ks + 1;
      if (ks == ksuper - 1) {
        fetch = has_next ? 0 : ks;
        if (has_next) set_sources(nm0, nn0);
      }
      fetch += k_rot;  // the depth this XCD is at (see k_rot)
      fetch = fetch >= ksuper ? fetch - ksuper : fetch;
      // The eight LDS-DMA pieces of a wave are issued in the LOAD segments of phases 0 and 1 — by the group that is NOT
      // computing, behind its own fragment reads (lgkmcnt(0)) and ahead of the barrier. An LDS-DMA instruction blocks its wave's
      // instruction stream for 60+ cycles: inside a cluster (rounds 1-2) that let the matrix pipe run dry behind every piece
      // (round 3, A/B on one box, bit-identical: layer mix 2.52 -> 2.61 POP/s, down_proj +4.3 %, MLP mode +2.1 %; round 2's
      // "pieces in the LOAD segments: -6 %" had them AHEAD of the reads, where they delay the fragments). WAR on the target slot:
      // its last readers, the other group's reads of the previous super-step, were issued a whole barrier interval earlier.
      read_frags(st, 0);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      issue_a(fetch, slot ^ 1, 0); issue_b(fetch, slot ^ 1, 0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      cluster(0);
      __builtin_amdgcn_s_barrier();
      read_frags(st, 1);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      issue_a(fetch, slot ^ 1, 2); issue_b(fetch, slot ^ 1, 2);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      cluster(1);
      __builtin_amdgcn_s_barrier();
      read_frags(st, 2);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      cluster(0);
      __builtin_amdgcn_s_barrier();
      read_frags(st, 3);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the fetched super-step landed (and older epilogue stores)
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      cluster(1);
      __builtin_amdgcn_s_barrier();
      slot ^= 1;
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();  // same number of barriers for both groups
    uint8_t* scratch = lds2 + (slot ^ 1) * SLOT_BYTES;  // the consumed slot: epilogue scratch
    __syncthreads();
    if constexpr (MLP) {
      int rsw[2] = {0, 0};
      if (a.rowsum_w) {
        rsw[0] = a.rowsum_w[n0 + wn * 32 + (lane & 31)];
        rsw[1] = a.rowsum_w2[n0 + wn * 32 + (lane & 31)];
      }
#ifdef FFQ_ABLATE_EPILOGUE  // timing-only builds (tools/build_variant.sh): what the epilogues cost — one store per lane keeps the accumulators alive
      {
        int sum = rsw[0];
#pragma unroll
        for (int mi = 0; mi < 8; ++mi)
#pragma unroll
          for (int nj = 0; nj < 4; ++nj) sum += (acc[mi][nj][0] ^ acc[mi][nj][1]) + (acc[mi][nj][2] ^ acc[mi][nj][3]);
        if (sum == 0x7fffffff) static_cast<int8_t*>(a.out)[lane] = (int8_t)sum;
      }
#else
      if constexpr (REQUANT) mlp_epilogue16_body(a, acc, rsw, scratch, wave, lane, wm, wn, m0, n0, silu_table);
      else mlp_epilogue16_product(a, acc, rsw, scratch, wave, lane, wm, wn, m0, n0, silu_table, zext);
#endif
    } else {
#ifdef FFQ_ABLATE_EPILOGUE
      {
        int sum = 0;
#pragma unroll
        for (int mi = 0; mi < 8; ++mi)
#pragma unroll
          for (int nj = 0; nj < 4; ++nj) sum += (acc[mi][nj][0] ^ acc[mi][nj][1]) + (acc[mi][nj][2] ^ acc[mi][nj][3]);
        if (sum == 0x7fffffff) static_cast<int8_t*>(a.out)[lane] = (int8_t)sum;
      }
#else
      gemm256_epilogue_slabs16<TOut, REQUANT, WOFF, GATED>(a, acc, scratch, wave, lane, wm, wn, m0, n0, silu_table, zext);
#endif
    }
    __syncthreads();  // the scratch slot is the next tile's DMA target
    m0 = nm0; n0 = nn0;
  }
  if constexpr (GATED || PRODUCT) {
    if (a.extrema.words) {  // lanes -> waves -> block -> the launch's three words (the operand ring is free: every tile is done)
      float mn = zext[0], mx = zext[512], nan = zext[1024];
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) {
        mn = __builtin_fminf(mn, __shfl_xor(mn, d, 64));
        mx = __builtin_fmaxf(mx, __shfl_xor(mx, d, 64));
        nan = __builtin_fmaxf(nan, __shfl_xor(nan, d, 64));
      }
      float* red = reinterpret_cast<float*>(lds2);
      if (lane == 0) { red[wave] = mn; red[8 + wave] = mx; red[16 + wave] = nan; }
      __syncthreads();
      if (tid == 0) {
        for (int w = 1; w < 8; ++w) {
          mn = __builtin_fminf(mn, red[w]);
          mx = __builtin_fmaxf(mx, red[8 + w]);
          nan = __builtin_fmaxf(nan, red[16 + w]);
        }
        extrema_publish(a.extrema, mn, mx, nan != 0.0f, true, gridDim.x);
      }
    }
  }
}

// one wavefront per row: sum of K int8 codes. `gate` (nullable): a device word; 0 = the sums are not needed (nothing is read).
// `offsets` (nullable): block 0 also decides whether any rounded entry of the `n_offsets` weight offsets is non-zero and writes
// `flag` — the job of offsets_nonzero_kernel folded into a launch that runs anyway (one launch fewer per linear and calibration step)
__global__ __launch_bounds__(256) void rowsum_i8_kernel(const int8_t* __restrict__ q, int rows, int K,
                                                        int32_t* __restrict__ sums, const int32_t* __restrict__ gate,
                                                        const float* __restrict__ offsets = nullptr, int n_offsets = 0,
                                                        int32_t* __restrict__ flag = nullptr, EarlierCodes earlier = {nullptr, nullptr, nullptr},
                                                        const float* __restrict__ q_scale = nullptr, const float* __restrict__ q_offset = nullptr) {
  if (offsets && blockIdx.x == 0) {
    int any = 0;
    for (int i = threadIdx.x; i < n_offsets; i += 256) any |= rne(offsets[i]) != 0.0f;
    any = __syncthreads_or(any);
    if (threadIdx.x == 0) flag[0] = any ? 1 : 0;
  }
  if (gate && *gate == 0) return;
  if (earlier.codes) q = codes_in_force(q, q_scale, q_offset, earlier);  // (`q` are activation codes their quantizer may not have written)
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  int s = 0;
  if (row < rows) {
    const int8_t* p = q + (size_t)row * K;
    for (int k = lane * 16; k < K; k += 64 * 16) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(p + k);
      s = __builtin_amdgcn_sdot4((int)v.x, 0x01010101, s, false);
      s = __builtin_amdgcn_sdot4((int)v.y, 0x01010101, s, false);
      s = __builtin_amdgcn_sdot4((int)v.z, 0x01010101, s, false);
      s = __builtin_amdgcn_sdot4((int)v.w, 0x01010101, s, false);
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
  if (lane == 0 && row < rows) sums[row] = s;
}

// flag[0] = 1 if any round_half_even(offset[i]) != 0 (a symmetric quantizer carries an all-zero offset BUFFER, reference
// nn/linear_quantizer.py:164-170; A5 re-writes it on every calibration step, so the answer is taken on the device)
__global__ __launch_bounds__(1024) void offsets_nonzero_kernel(const float* __restrict__ offset, int n, int32_t* __restrict__ flag) {
  int any = 0;
  for (int i = threadIdx.x; i < n; i += 1024) any |= rne(offset[i]) != 0.0f;
  any = __syncthreads_or(any);
  if (threadIdx.x == 0) flag[0] = any ? 1 : 0;
}

}  // namespace ffq

using namespace ffq;

// workspace: [M] activation row sums, [N] weight row sums, one flag word (+ padding)
extern "C" size_t ffq_linear_w8a8_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  (void)K;
  if (M < 0 || N < 0) return 0;
  return (size_t)(((M + N + 1) * 4 + 255) & ~(int64_t)255);
}

// the persistent kernel's shape class (the only one whose launches can read an earlier quantizer's codes)
static bool linear_takes_earlier(int64_t M, int64_t N, int64_t K) {
  if (M <= 0 || N <= 0 || K <= 0) return false;
  const int64_t tiles256 = ((M + BM2 - 1) / BM2) * ((N + 255) / 256);
  return K % 128 == 0 && K >= 256 && M >= 128 && N >= 128 && tiles256 >= 64;
}

extern "C" int ffq_linear_w8a8_takes_earlier(int64_t M, int64_t N, int64_t K) { return linear_takes_earlier(M, N, K) ? 1 : 0; }

static int linear_w8a8_impl(const int8_t* xq, const int8_t* wq, const int32_t* w_rowsum, const float* x_scale,
                            const float* x_offset, int x_per_row, const float* w_scale, const float* w_offset,
                            int w_per_row, const void* bias, int bias_dt, void* out, int out_dt,
                            const float* out_scale, const float* out_offset, double out_num_bits, int y_dt, int64_t M,
                            int64_t N, int64_t K, void* workspace, size_t workspace_bytes, void* stream, const void* gate,
                            uint32_t* extrema_words = nullptr, void* extrema_pair = nullptr, const int32_t* run_if = nullptr, int run_when = 0,
                            EarlierCodes earlier = {nullptr, nullptr, nullptr}, int seg_count = 1, const int64_t* seg_ns = nullptr,
                            void* const* seg_outs = nullptr) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (M < 0 || N < 0 || K < 0) return fail(FFQ_ERR_ARG, "negative extent");
  if (M == 0 || N == 0) return FFQ_OK;
  if (earlier.codes && (!earlier.scale || x_per_row || !aligned16(earlier.codes) || !linear_takes_earlier(M, N, K)))
    return fail(FFQ_ERR_DTYPE, "earlier codes: per-tensor activation parameters on the persistent kernel's shapes (ffq_linear_w8a8_takes_earlier)");
  if (!xq || !wq || !x_scale || !w_scale || !out) return fail(FFQ_ERR_ARG, "NULL buffer");
  if (M > INT32_MAX || N > INT32_MAX || K > INT32_MAX) return fail(FFQ_ERR_ARG, "extent exceeds 2^31");
  if (K % 16 != 0 || !aligned16(xq) || !aligned16(wq))
    return fail(FFQ_ERR_DTYPE, "w8a8 linear needs K %% 16 == 0 and 16-byte aligned code pointers");
  if (bias && !dt_valid(bias_dt)) return fail(FFQ_ERR_ARG, "bad bias dtype");
  const bool requant = out_scale != nullptr;
  if (requant) {
    if (!ffq_can_support_bitwidth(out_dt, out_num_bits))
      return fail(FFQ_ERR_PRECISION, "Provided dtype (%d) is not enough to store %g bits quantized values.",
                  out_dt, out_num_bits);
    if (!(y_dt == FFQ_F32 || y_dt == FFQ_BF16 || y_dt == FFQ_F16))
      return fail(FFQ_ERR_DTYPE, "the re-quantized linear's real-valued dtype must be f32, bf16 or f16");
  } else if (!(out_dt == FFQ_F32 || out_dt == FFQ_BF16 || out_dt == FFQ_F16)) {
    return fail(FFQ_ERR_DTYPE, "real-valued output must be f32, bf16 or f16");
  }
  const size_t need = ffq_linear_w8a8_workspace_bytes(M, N, K);
  // what the workspace holds: the weight row sums of a persistent launch (the tail kernel sums its own), the activation row sums and
  // the flag of a launch with weight offsets / earlier codes. A launch with none of them runs with workspace == NULL (include/ffq.h)
  const bool sums_in_workspace = x_offset && !w_rowsum && linear_takes_earlier(M, N, K);
  if ((sums_in_workspace || w_offset || earlier.codes) && (need > workspace_bytes || !workspace))
    return fail(FFQ_ERR_WORKSPACE, "w8a8 linear needs %zu workspace bytes, got %zu", need, workspace_bytes);

  LinearArgs a;
  a.xq = xq; a.wq = wq;
  a.x_scale = x_scale; a.x_offset = x_offset;
  a.w_scale = w_scale; a.w_offset = w_offset;
  a.rowsum_x = nullptr; a.rowsum_w = nullptr; a.woff_live = nullptr; a.rowsum_w_inside = 0;
  a.seg_start[0] = a.seg_start[1] = INT32_MAX; a.seg_out[0] = a.seg_out[1] = nullptr;
  a.wq2 = nullptr; a.w_scale2 = nullptr; a.rowsum_w2 = nullptr;
  a.batch_x = a.batch_w = a.batch_out = 0;
  a.bias = bias; a.bias_dt = bias_dt;
  a.out = out; a.out_dt = out_dt;
  a.out_scale = out_scale; a.out_offset = out_offset;
  const double lo = -pow(2.0, out_num_bits - 1.0);
  a.out_lo = (float)lo; a.out_hi = (float)(-lo - 1.0);
  a.y_dt = y_dt;
  a.x_per_row = x_per_row; a.w_per_row = w_per_row;
  a.M = (int)M; a.N = (int)N; a.K = (int)K;
  a.group_m = GROUP_M2;
  a.group_cols = 0;
  a.gate = static_cast<const bf16_t*>(gate);
  a.run_if = run_if; a.run_when = run_when;
  a.earlier = earlier;
  a.extrema.words = extrema_words; a.extrema.pair = extrema_pair; a.extrema.pair_dt = FFQ_BF16;

  int32_t* ws = static_cast<int32_t*>(workspace);
  const int64_t tiles256 = ((M + BM2 - 1) / BM2) * ((N + 255) / 256);
  const bool persistent = K % 128 == 0 && K >= 256 && M >= 128 && N >= 128 && tiles256 >= 64;
  // the gated epilogue lives in the persistent kernel's whole-line store path: everything else is the caller's two launches
  if (gate && !(persistent && N % 64 == 0 && out_dt == FFQ_BF16 && !requant && !bias && aligned16(gate) && aligned16(out)))
    return fail(FFQ_ERR_DTYPE, "gated w8a8 linear: outside the persistent kernel's whole-line path (bf16 out, N %% 64 == 0, >= 64 tiles of 256 x 256)");
  if (seg_count > 1) {  // several weight matrices side by side (ffq_linear_w8a8_multi): whole tiles per matrix, the persistent kernel's plain epilogue
    if (!persistent || requant || w_offset || gate || earlier.codes) return fail(FFQ_ERR_DTYPE, "w8a8 linears in one launch: the persistent kernel's plain form only (>= 64 tiles, no weight offsets, no output quantizer)");
    int64_t at = 0;
    for (int i = 0; i + 1 < seg_count; ++i) {
      if (seg_ns[i] % 256 != 0) return fail(FFQ_ERR_DTYPE, "w8a8 linears in one launch: every weight matrix but the last needs a multiple of 256 rows");
      at += seg_ns[i];
      a.seg_start[i] = (int)at;
      a.seg_out[i] = seg_outs[i + 1];
    }
  }
  bool flag_written = false;
  if (x_offset) {  // sum_k wq[n, k] for the zero-point term: one pass over the weight codes, unless the caller has them
    if (w_rowsum) {
      a.rowsum_w = w_rowsum;
    } else if (!persistent) {
      a.rowsum_w_inside = 1;  // the tail kernel sums its own weight rows (no launch ahead of it)
    } else {
      const bool with_flag = w_offset && persistent;  // the weight-offset decision rides in this launch
      rowsum_i8_kernel<<<(unsigned)((N + 3) / 4), 256, 0, s>>>(wq, (int)N, (int)K, ws + M, nullptr, with_flag ? w_offset : nullptr,
                                                               w_per_row ? (int)N : 1, with_flag ? ws + M + N : nullptr);
      a.rowsum_w = ws + M;
      flag_written = with_flag;
    }
  }
  if (w_offset) {  // sum_k xq[m, k] for the ow term
    if (persistent) {  // ... only where an offset is really non-zero: decided and consumed on the device
      int32_t* flag = ws + M + N;
      if (!flag_written) offsets_nonzero_kernel<<<1, 1024, 0, s>>>(w_offset, w_per_row ? (int)N : 1, flag);
      rowsum_i8_kernel<<<(unsigned)((M + 3) / 4), 256, 0, s>>>(xq, (int)M, (int)K, ws, flag, nullptr, 0, nullptr, earlier, x_scale, x_offset);
      a.woff_live = flag;
    } else {
      rowsum_i8_kernel<<<(unsigned)((M + 3) / 4), 256, 0, s>>>(xq, (int)M, (int)K, ws, nullptr);
    }
    a.rowsum_x = ws;
  }

  if (persistent) {
    a.tiles_m = (int)((M + BM2 - 1) / BM2);
    a.tiles_n = (int)((N + 255) / 256);
    // 8 row tiles per group; 4 for long contractions (down_proj, K = 14336: +3 %; A/B of 2 / 3 / 4 / 6 / 8 / 16 / 32 on one
    // box) — the group's A panels are 256 x K bytes each
    a.group_m = K >= 8192 ? 4 : GROUP_M2;
#ifdef FFQ_I8_GROUP_COLS  // A/B builds (tools/build_variant.sh): column groups where the activation codes exceed ~200 MB
    if ((size_t)M * (size_t)K > ((size_t)200 << 20)) { a.group_cols = 1; a.group_m = FFQ_I8_GROUP_COLS; }
#endif
    const unsigned total = (unsigned)(a.tiles_m * a.tiles_n);
    const unsigned grid = total < 256u ? total : 256u;  // persistent: one block per CU
    if (gate) {  // bf16 out, no re-quantization (checked above); with or without weight offsets
      const size_t lds_gated = (size_t)2 * (BM2 + 256) * 128 + kSiluBytes + 3 * 512 * 4;
#define FFQ_FQ_GATED(WO)                                                                                                        \
  do {                                                                                                                          \
    static uint64_t attr_set = 0;                                                                                               \
    ensure_dynamic_lds(&attr_set, reinterpret_cast<const void*>(&w8a8_gemm256fq_kernel<bf16_t, false, false, WO, true>), (int)lds_gated); \
    w8a8_gemm256fq_kernel<bf16_t, false, false, WO, true><<<grid, 512, lds_gated, s>>>(a, (int)total);                          \
  } while (0)
      if (w_offset) FFQ_FQ_GATED(true); else FFQ_FQ_GATED(false);
#undef FFQ_FQ_GATED
      return check_launch("w8a8_gemm256fq_kernel (gated)");
    }
    const size_t lds = (size_t)2 * (BM2 + 256) * 128;
#define FFQ_FQ_LAUNCH(T, RQ, WO)                                                                            \
  do {                                                                                                      \
    static uint64_t attr_set = 0;                                                                           \
    ensure_dynamic_lds(&attr_set, reinterpret_cast<const void*>(&w8a8_gemm256fq_kernel<T, RQ, false, WO>), (int)lds); \
    w8a8_gemm256fq_kernel<T, RQ, false, WO><<<grid, 512, lds, s>>>(a, (int)total);                          \
  } while (0)
    // a predicated launch without weight offsets takes the WOFF instantiation too (the forward's own instantiations carry no
    // predicate): its "some offset is live" word is run_if[1], which the either / or's deciding kernel keeps at zero
    if (run_if && !w_offset) a.woff_live = run_if + 1;
    // ... and so does one that may read an earlier quantizer's codes: its word is the workspace's flag slot, zeroed here
    if (earlier.codes && !w_offset && !run_if) {
      hipError_t e = hipMemsetAsync(ws + M + N, 0, 4, s);
      if (e != hipSuccess) return fail(FFQ_ERR_LAUNCH, "hipMemsetAsync: %s", hipGetErrorString(e));
      a.woff_live = ws + M + N;
    }
#define FFQ_FQ(T, RQ) do { if (w_offset || run_if || earlier.codes) FFQ_FQ_LAUNCH(T, RQ, true); else FFQ_FQ_LAUNCH(T, RQ, false); } while (0)
    if (requant) {
      switch (out_dt) {
        case FFQ_I8: FFQ_FQ(int8_t, true); break;
        case FFQ_BF16: FFQ_FQ(bf16_t, true); break;
        case FFQ_F16: FFQ_FQ(f16_t, true); break;
        case FFQ_F32: FFQ_FQ(float, true); break;
        default: return fail(FFQ_ERR_DTYPE, "re-quantized output container must be i8, bf16, f16 or f32");
      }
    } else {
      switch (out_dt) {
        case FFQ_BF16: FFQ_FQ(bf16_t, false); break;
        case FFQ_F16: FFQ_FQ(f16_t, false); break;
        default: FFQ_FQ(float, false); break;
      }
    }
#undef FFQ_FQ
#undef FFQ_FQ_LAUNCH
    return check_launch("w8a8_gemm256fq_kernel");
  }

  a.tiles_m = (int)((M + BM - 1) / BM);
  a.tiles_n = (int)((N + BN - 1) / BN);
  const unsigned grid = (unsigned)(a.tiles_m * a.tiles_n);
  if (requant) {
    switch (out_dt) {
      case FFQ_I8: w8a8_gemm_kernel<int8_t, true><<<grid, 256, 0, s>>>(a); break;
      case FFQ_BF16: w8a8_gemm_kernel<bf16_t, true><<<grid, 256, 0, s>>>(a); break;
      case FFQ_F16: w8a8_gemm_kernel<f16_t, true><<<grid, 256, 0, s>>>(a); break;
      case FFQ_F32: w8a8_gemm_kernel<float, true><<<grid, 256, 0, s>>>(a); break;
      default: return fail(FFQ_ERR_DTYPE, "re-quantized output container must be i8, bf16, f16 or f32");
    }
  } else {
    switch (out_dt) {
      case FFQ_BF16: w8a8_gemm_kernel<bf16_t, false><<<grid, 256, 0, s>>>(a); break;
      case FFQ_F16: w8a8_gemm_kernel<f16_t, false><<<grid, 256, 0, s>>>(a); break;
      default: w8a8_gemm_kernel<float, false><<<grid, 256, 0, s>>>(a); break;
    }
  }
  return check_launch("w8a8_gemm_kernel");
}

extern "C" int ffq_linear_w8a8(const int8_t* xq, const int8_t* wq, const int32_t* w_rowsum, const float* x_scale,
                               const float* x_offset, int x_per_row, const float* w_scale, const float* w_offset,
                               int w_per_row, const void* bias, int bias_dt, void* out, int out_dt,
                               const float* out_scale, const float* out_offset, double out_num_bits, int y_dt, int64_t M,
                               int64_t N, int64_t K, void* workspace, size_t workspace_bytes, void* stream) {
  return linear_w8a8_impl(xq, wq, w_rowsum, x_scale, x_offset, x_per_row, w_scale, w_offset, w_per_row, bias, bias_dt, out, out_dt, out_scale,
                          out_offset, out_num_bits, y_dt, M, N, K, workspace, workspace_bytes, stream, nullptr);
}

// ffq_linear_w8a8 whose activation codes may not have been written (ffq_quantize_by_tile_unless_same) — see include/ffq.h
// q_proj / k_proj / v_proj of a W8A8 attention block (three QuantizedLinear modules on one quantized hidden state, reference nn/linear.py:32-39
// three times over _gen/fallback.py:77-112) as ONE launch: `count` (2 or 3) weight matrices whose int8 codes, scales and (optional) row
// sums lie matrix after matrix in ONE [N, K] / [N] run each (N = the sum of Ns), separate outputs outs[i] = [M, Ns[i]]. Exactly the values
// of `count` ffq_linear_w8a8 calls (the same tiles, the same epilogue): a k/v projection (4 column tiles per row tile) no longer runs
// alone on a corner of the chip. Every matrix but the last needs a multiple of 256 rows; per-tensor or per-row activation parameters,
// one scale per weight row, no weight offsets, no bias, no output quantizer; shapes the persistent kernel does not take return
// FFQ_ERR_DTYPE before touching a buffer (the caller launches the matrices one by one). Workspace: ffq_linear_w8a8_workspace_bytes(M, N, K).
extern "C" int ffq_linear_w8a8_multi(const int8_t* xq, const int8_t* wq, const int32_t* w_rowsum, const float* x_scale, const float* x_offset,
                                     int x_per_row, const float* w_scale, int count, void* const* outs, int out_dt, int64_t M, const int64_t* Ns,
                                     int64_t K, void* workspace, size_t workspace_bytes, void* stream) {
  if (count < 2 || count > 3 || !outs || !Ns) return fail(FFQ_ERR_ARG, "2 or 3 weight matrices");
  int64_t N = 0;
  for (int i = 0; i < count; ++i) {
    if (Ns[i] <= 0 || !outs[i]) return fail(FFQ_ERR_ARG, "empty weight matrix or NULL output");
    if (!aligned16(outs[i])) return fail(FFQ_ERR_DTYPE, "w8a8 linears in one launch need 16-byte aligned outputs");
    N += Ns[i];
  }
  return linear_w8a8_impl(xq, wq, w_rowsum, x_scale, x_offset, x_per_row, w_scale, nullptr, 1, nullptr, 0, outs[0], out_dt, nullptr, nullptr, 8.0, 0, M, N, K,
                          workspace, workspace_bytes, stream, nullptr, nullptr, nullptr, nullptr, 0, EarlierCodes{nullptr, nullptr, nullptr}, count, Ns, outs);
}

extern "C" int ffq_linear_w8a8_earlier(const int8_t* xq, const int8_t* earlier_xq, const float* earlier_scale, const float* earlier_offset,
                                       const int8_t* wq, const int32_t* w_rowsum, const float* x_scale, const float* x_offset,
                                       const float* w_scale, const float* w_offset, int w_per_row, void* out, int out_dt, int64_t M,
                                       int64_t N, int64_t K, void* workspace, size_t workspace_bytes, void* stream) {
  if (!earlier_xq || !earlier_scale) return fail(FFQ_ERR_ARG, "NULL earlier codes / scale");
  return linear_w8a8_impl(xq, wq, w_rowsum, x_scale, x_offset, 0, w_scale, w_offset, w_per_row, nullptr, 0, out, out_dt, nullptr, nullptr, 8.0, 0,
                          M, N, K, workspace, workspace_bytes, stream, nullptr, nullptr, nullptr, nullptr, 0,
                          EarlierCodes{earlier_xq, earlier_scale, earlier_offset});
}

// out = bf16(silu(gate)) * bf16(linear(x, w))  — see include/ffq.h
extern "C" int ffq_linear_w8a8_gated(const int8_t* xq, const int8_t* wq, const int32_t* w_rowsum, const float* x_scale,
                                     const float* x_offset, int x_per_row, const float* w_scale, const float* w_offset,
                                     int w_per_row, const void* gate, void* out, int64_t M, int64_t N, int64_t K,
                                     void* workspace, size_t workspace_bytes, uint32_t* extrema_words, void* extrema_pair, void* stream) {
  if (!gate) return fail(FFQ_ERR_ARG, "NULL gate");
  if ((extrema_words == nullptr) != (extrema_pair == nullptr)) return fail(FFQ_ERR_ARG, "extrema_words and extrema_pair come together");
  return linear_w8a8_impl(xq, wq, w_rowsum, x_scale, x_offset, x_per_row, w_scale, w_offset, w_per_row, nullptr, 0, out, FFQ_BF16, nullptr,
                          nullptr, 8.0, 0, M, N, K, workspace, workspace_bytes, stream, gate, extrema_words, extrema_pair);
}

// ---- bmm: `batch` independent [M, K] x [N, K]^T products with ONE parameter pair per operand, as ONE launch -----------------------
// fallback.bmm (_gen/fallback.py:699-798 pattern: dequantize both operands, torch.bmm, output quantizer) on int8 codes: the
// 128 x 128-tile kernel with the matrix pair chosen by blockIdx.y (round 3 looped over the batch in Python: up to 256 launches
// and a torch.stack). Row sums of both operands over the flattened [batch * rows, K] matrices: two launches for the whole batch.
extern "C" size_t ffq_bmm_w8a8_workspace_bytes(int64_t batch, int64_t M, int64_t N, int64_t K) {
  (void)K;
  if (batch < 0 || M < 0 || N < 0) return 0;
  return (size_t)((batch * (M + N) * 4 + 255) & ~(int64_t)255);
}

extern "C" int ffq_bmm_w8a8(const int8_t* xq, const int8_t* wq, const float* x_scale, const float* x_offset, const float* w_scale,
                            const float* w_offset, void* out, int out_dt, const float* out_scale, const float* out_offset,
                            double out_num_bits, int y_dt, int64_t batch, int64_t M, int64_t N, int64_t K, void* workspace,
                            size_t workspace_bytes, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (batch < 0 || M < 0 || N < 0 || K < 0) return fail(FFQ_ERR_ARG, "negative extent");
  if (batch == 0 || M == 0 || N == 0) return FFQ_OK;
  if (!xq || !wq || !x_scale || !w_scale || !out) return fail(FFQ_ERR_ARG, "NULL buffer");
  if (batch > 65535 || M > INT32_MAX || N > INT32_MAX || K > INT32_MAX || batch * M > INT32_MAX || batch * N > INT32_MAX)
    return fail(FFQ_ERR_ARG, "extent out of range");
  if (K % 16 != 0 || !aligned16(xq) || !aligned16(wq) || (M * K) % 16 != 0 || (N * K) % 16 != 0)
    return fail(FFQ_ERR_DTYPE, "batched w8a8 matmul needs K %% 16 == 0 and 16-byte aligned matrices");
  const bool requant = out_scale != nullptr;
  if (requant) {
    if (!ffq_can_support_bitwidth(out_dt, out_num_bits))
      return fail(FFQ_ERR_PRECISION, "Provided dtype (%d) is not enough to store %g bits quantized values.", out_dt, out_num_bits);
    if (!(y_dt == FFQ_F32 || y_dt == FFQ_BF16 || y_dt == FFQ_F16)) return fail(FFQ_ERR_DTYPE, "the re-quantized product's real-valued dtype must be f32, bf16 or f16");
  } else if (!(out_dt == FFQ_F32 || out_dt == FFQ_BF16 || out_dt == FFQ_F16)) {
    return fail(FFQ_ERR_DTYPE, "real-valued output must be f32, bf16 or f16");
  }
  const size_t need = ffq_bmm_w8a8_workspace_bytes(batch, M, N, K);
  if (w_offset && (need > workspace_bytes || !workspace)) return fail(FFQ_ERR_WORKSPACE, "batched w8a8 matmul needs %zu workspace bytes, got %zu", need, workspace_bytes);
  LinearArgs a;
  a.xq = xq; a.wq = wq;
  a.x_scale = x_scale; a.x_offset = x_offset;
  a.w_scale = w_scale; a.w_offset = w_offset;
  a.rowsum_x = nullptr; a.rowsum_w = nullptr; a.woff_live = nullptr; a.rowsum_w_inside = 0;
  a.seg_start[0] = a.seg_start[1] = INT32_MAX; a.seg_out[0] = a.seg_out[1] = nullptr;
  a.wq2 = nullptr; a.w_scale2 = nullptr; a.rowsum_w2 = nullptr;
  a.batch_x = M * K; a.batch_w = N * K; a.batch_out = M * N;
  a.gate = nullptr; a.extrema.words = nullptr; a.extrema.pair = nullptr; a.extrema.pair_dt = 0; a.run_if = nullptr; a.run_when = 0;
  a.bias = nullptr; a.bias_dt = 0;
  a.out = out; a.out_dt = out_dt;
  a.out_scale = out_scale; a.out_offset = out_offset;
  const double lo = -pow(2.0, out_num_bits - 1.0);
  a.out_lo = (float)lo; a.out_hi = (float)(-lo - 1.0);
  a.y_dt = y_dt;
  a.x_per_row = 0; a.w_per_row = 0;
  a.M = (int)M; a.N = (int)N; a.K = (int)K;
  a.group_m = GROUP_M2;
  a.group_cols = 0;
  int32_t* ws = static_cast<int32_t*>(workspace);
  if (x_offset) a.rowsum_w_inside = 1;  // every block sums its own weight rows (w8a8_gemm_kernel)
  if (w_offset) {
    rowsum_i8_kernel<<<(unsigned)((batch * M + 3) / 4), 256, 0, s>>>(xq, (int)(batch * M), (int)K, ws, nullptr);
    a.rowsum_x = ws;
  }
  a.tiles_m = (int)((M + BM - 1) / BM);
  a.tiles_n = (int)((N + BN - 1) / BN);
  const dim3 grid((unsigned)(a.tiles_m * a.tiles_n), (unsigned)batch);
  if (requant) {
    switch (out_dt) {
      case FFQ_I8: w8a8_gemm_kernel<int8_t, true><<<grid, 256, 0, s>>>(a); break;
      case FFQ_BF16: w8a8_gemm_kernel<bf16_t, true><<<grid, 256, 0, s>>>(a); break;
      case FFQ_F16: w8a8_gemm_kernel<f16_t, true><<<grid, 256, 0, s>>>(a); break;
      case FFQ_F32: w8a8_gemm_kernel<float, true><<<grid, 256, 0, s>>>(a); break;
      default: return fail(FFQ_ERR_DTYPE, "re-quantized output container must be i8, bf16, f16 or f32");
    }
  } else {
    switch (out_dt) {
      case FFQ_BF16: w8a8_gemm_kernel<bf16_t, false><<<grid, 256, 0, s>>>(a); break;
      case FFQ_F16: w8a8_gemm_kernel<f16_t, false><<<grid, 256, 0, s>>>(a); break;
      default: w8a8_gemm_kernel<float, false><<<grid, 256, 0, s>>>(a); break;
    }
  }
  return check_launch("w8a8_gemm_kernel (batched)");
}

// ---- gate_proj + up_proj + SiLU * up + the down_proj input quantizer in one launch -----------------------------
extern "C" size_t ffq_mlp_gate_up_w8a8_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  (void)M; (void)K;
  if (N < 0) return 0;
  return (size_t)((2 * N * 4 + 255) & ~(int64_t)255);
}

// `product_out` (bf16 [M, N]) instead of `codes_out`: the launch leaves silu(gate) * up itself, unquantized (+ its extrema)
static int mlp_gate_up_w8a8_impl(const int8_t* xq, const int8_t* gate_wq, const int8_t* up_wq, const int32_t* gate_rowsum,
                                 const int32_t* up_rowsum, const float* x_scale, const float* x_offset,
                                 const float* gate_w_scale, const float* up_w_scale, int8_t* codes_out,
                                 const float* out_scale, const float* out_offset, double out_num_bits, int64_t M,
                                 int64_t N, int64_t K, void* workspace, size_t workspace_bytes, void* stream,
                                 void* product_out, uint32_t* extrema_words, void* extrema_pair, const int32_t* run_if, int run_when) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool have_sums = gate_rowsum && up_rowsum;
  if (M < 0 || N < 0 || K < 0) return fail(FFQ_ERR_ARG, "negative extent");
  if (M == 0 || N == 0) return FFQ_OK;
  if (!xq || !gate_wq || !up_wq || !x_scale || !gate_w_scale || !up_w_scale || (!product_out && (!codes_out || !out_scale)))
    return fail(FFQ_ERR_ARG, "NULL buffer");
  if (M > INT32_MAX || N > INT32_MAX || K > INT32_MAX) return fail(FFQ_ERR_ARG, "extent exceeds 2^31");
  if (N % 128 != 0 || K % 128 != 0 || K < 256 || !aligned16(xq) || !aligned16(gate_wq) || !aligned16(up_wq) || !aligned16(product_out ? product_out : (void*)codes_out))
    return fail(FFQ_ERR_DTYPE, "fused gate/up kernel needs N %% 128 == 0, K %% 128 == 0, K >= 256 and 16-byte aligned buffers");
  if (!product_out && !(out_num_bits >= 1 && out_num_bits <= 8 && out_num_bits == floor(out_num_bits)))
    return fail(FFQ_ERR_PRECISION, "Provided dtype (%d) is not enough to store %g bits quantized values.", FFQ_I8, out_num_bits);
  const size_t need = ffq_mlp_gate_up_w8a8_workspace_bytes(M, N, K);
  if (x_offset && !have_sums && (need > workspace_bytes || !workspace)) return fail(FFQ_ERR_WORKSPACE, "fused gate/up needs %zu workspace bytes, got %zu", need, workspace_bytes);
  LinearArgs a;
  a.xq = xq; a.wq = gate_wq; a.wq2 = up_wq;
  a.x_scale = x_scale; a.x_offset = x_offset;
  a.w_scale = gate_w_scale; a.w_scale2 = up_w_scale; a.w_offset = nullptr;
  a.rowsum_x = nullptr; a.rowsum_w = nullptr; a.rowsum_w2 = nullptr; a.woff_live = nullptr; a.rowsum_w_inside = 0;
  a.seg_start[0] = a.seg_start[1] = INT32_MAX; a.seg_out[0] = a.seg_out[1] = nullptr;
  a.batch_x = a.batch_w = a.batch_out = 0;
  a.gate = nullptr; a.extrema.words = extrema_words; a.extrema.pair = extrema_pair; a.extrema.pair_dt = FFQ_BF16;
  a.run_if = run_if; a.run_when = run_when;
  a.bias = nullptr; a.bias_dt = 0;
  a.out = product_out ? product_out : (void*)codes_out; a.out_dt = product_out ? FFQ_BF16 : FFQ_I8;
  a.out_scale = out_scale; a.out_offset = out_offset;
  const double lo = -pow(2.0, out_num_bits - 1.0);
  a.out_lo = (float)lo; a.out_hi = (float)(-lo - 1.0);
  a.y_dt = FFQ_BF16;
  a.x_per_row = 0; a.w_per_row = 1;
  a.M = (int)M; a.N = (int)N; a.K = (int)K;
  a.tiles_m = (int)((M + BM2 - 1) / BM2);
  a.tiles_n = (int)(N / 128);
  a.group_m = K >= 8192 ? 4 : GROUP_M2;
  a.group_cols = 0;
  if (x_offset && have_sums) {
    a.rowsum_w = gate_rowsum; a.rowsum_w2 = up_rowsum;
  } else if (x_offset) {
    int32_t* ws = static_cast<int32_t*>(workspace);
    rowsum_i8_kernel<<<(unsigned)((N + 3) / 4), 256, 0, s>>>(gate_wq, (int)N, (int)K, ws, nullptr);
    rowsum_i8_kernel<<<(unsigned)((N + 3) / 4), 256, 0, s>>>(up_wq, (int)N, (int)K, ws + N, nullptr);
    a.rowsum_w = ws; a.rowsum_w2 = ws + N;
  }
  const int total = a.tiles_m * a.tiles_n;
  if (product_out) {
    const size_t lds = (size_t)2 * (BM2 + 256) * 128 + kSiluBytes + 3 * 512 * 4;  // ... + the threads' running extrema
    static uint64_t attr_set = 0;
    ensure_dynamic_lds(&attr_set, reinterpret_cast<const void*>(&w8a8_gemm256fq_kernel<bf16_t, false, true>), (int)lds);
    w8a8_gemm256fq_kernel<bf16_t, false, true><<<(unsigned)(total < 256 ? total : 256), 512, lds, s>>>(a, total);
    return check_launch("w8a8_gemm256fq_kernel (mlp mode, product)");
  }
  const size_t lds = (size_t)2 * (BM2 + 256) * 128 + kSiluBytes;  // two operand slots + the silu table
  static uint64_t attr_set = 0;
  ensure_dynamic_lds(&attr_set, reinterpret_cast<const void*>(&w8a8_gemm256fq_kernel<int8_t, true, true>), (int)lds);
  w8a8_gemm256fq_kernel<int8_t, true, true><<<(unsigned)(total < 256 ? total : 256), 512, lds, s>>>(a, total);
  return check_launch("w8a8_gemm256fq_kernel (mlp mode)");
}

extern "C" int ffq_mlp_gate_up_w8a8(const int8_t* xq, const int8_t* gate_wq, const int8_t* up_wq, const int32_t* gate_rowsum,
                                    const int32_t* up_rowsum, const float* x_scale, const float* x_offset,
                                    const float* gate_w_scale, const float* up_w_scale, int8_t* codes_out,
                                    const float* out_scale, const float* out_offset, double out_num_bits, int64_t M,
                                    int64_t N, int64_t K, void* workspace, size_t workspace_bytes, void* stream) {
  return mlp_gate_up_w8a8_impl(xq, gate_wq, up_wq, gate_rowsum, up_rowsum, x_scale, x_offset, gate_w_scale, up_w_scale, codes_out, out_scale,
                               out_offset, out_num_bits, M, N, K, workspace, workspace_bytes, stream, nullptr, nullptr, nullptr, nullptr, 0);
}

// ---- the gated MLP up to its product while gate_proj's and up_proj's input quantizers are being calibrated ----------------------------
// The one-launch mode takes ONE set of activation codes and weights without offsets; during range estimation the two input
// quantizers are separate estimators whose parameters are rewritten on every step — equal whenever both have seen the same data, which
// the host cannot know without a read per layer and step — and a symmetric weight quantizer's offset buffer is zero unless its
// weight is non-negative, also a fact of the device. So the decision is taken on the device: one block compares the two parameter
// pairs (scale bit for bit, rounded offsets) and tests the two offset buffers, and both routes are enqueued with the flag as
// their predicate — the route that is not taken costs a launch whose blocks return on their first instruction:
//   flag == 1  w8a8_gemm256fq_kernel<bf16, no requant, MLP>: gate + up + SiLU * up -> the bf16 product, its [min, max]
//   flag == 0  gate_proj's linear (weight offsets on the device) into `gate_scratch`, then up_proj's with the gated epilogue
// Either way `product_out` / `extrema_pair` hold what the two linears + SiLU * up + a reduction would have produced.
namespace ffq {
__global__ __launch_bounds__(1024) void mlp_inputs_agree_kernel(const float* __restrict__ xs_g, const float* __restrict__ xo_g,
                                                               const float* __restrict__ xs_u, const float* __restrict__ xo_u,
                                                               const float* __restrict__ wo_g, const float* __restrict__ wo_u, int n,
                                                               int32_t* __restrict__ flag) {
  int bad = 0;
  if (wo_g) for (int i = threadIdx.x; i < n; i += 1024) bad |= rne(wo_g[i]) != 0.0f;
  if (wo_u) for (int i = threadIdx.x; i < n; i += 1024) bad |= rne(wo_u[i]) != 0.0f;
  if (threadIdx.x == 0) {
    bad |= __builtin_bit_cast(uint32_t, xs_g[0]) != __builtin_bit_cast(uint32_t, xs_u[0]);
    const float og = xo_g ? rne(xo_g[0]) : 0.0f, ou = xo_u ? rne(xo_u[0]) : 0.0f;
    bad |= !(og == ou);  // (a NaN offset: not equal, the two-launch route reproduces whatever the linears make of it)
  }
  bad = __syncthreads_or(bad);
  if (threadIdx.x == 0) {
    flag[0] = bad ? 0 : 1;
    flag[1] = 0;  // (the "a weight offset is live" word of a predicated launch without weight offsets: linear_w8a8_impl)
  }
}
}  // namespace ffq

// workspace: [64 int32: the flag] [N gate row sums] [N up row sums] [ffq_linear_w8a8_workspace_bytes(M, N, K) for the two-launch route]
extern "C" size_t ffq_mlp_gate_up_w8a8_estimating_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  if (M < 0 || N < 0) return 0;
  return 256 + (size_t)((2 * N * 4 + 255) & ~(int64_t)255) + ffq_linear_w8a8_workspace_bytes(M, N, K);
}

extern "C" int ffq_mlp_gate_up_w8a8_estimating(const int8_t* xq_gate, const int8_t* xq_up, const int8_t* gate_wq, const int8_t* up_wq,
                                               const float* x_scale_gate, const float* x_offset_gate, const float* x_scale_up,
                                               const float* x_offset_up, const float* gate_w_scale, const float* gate_w_offset,
                                               const float* up_w_scale, const float* up_w_offset, void* gate_scratch, void* product_out,
                                               int64_t M, int64_t N, int64_t K, void* workspace, size_t workspace_bytes,
                                               uint32_t* extrema_words, void* extrema_pair, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (M < 0 || N < 0 || K < 0) return fail(FFQ_ERR_ARG, "negative extent");
  if (M == 0 || N == 0) return FFQ_OK;
  if (!xq_gate || !xq_up || !gate_wq || !up_wq || !x_scale_gate || !x_scale_up || !gate_w_scale || !up_w_scale || !gate_scratch || !product_out)
    return fail(FFQ_ERR_ARG, "NULL buffer");
  if ((extrema_words == nullptr) != (extrema_pair == nullptr)) return fail(FFQ_ERR_ARG, "extrema_words and extrema_pair come together");
  if (M > INT32_MAX || N > INT32_MAX || K > INT32_MAX) return fail(FFQ_ERR_ARG, "extent exceeds 2^31");
  const int64_t tiles256 = ((M + BM2 - 1) / BM2) * ((N + 255) / 256);
  // both routes must be able to run: the one-launch mode's shapes and the persistent kernel's whole-line path of the gated epilogue
  if (N % 128 != 0 || K % 128 != 0 || K < 256 || M < 128 || tiles256 < 64 || !aligned16(xq_gate) || !aligned16(xq_up) || !aligned16(gate_wq) ||
      !aligned16(up_wq) || !aligned16(gate_scratch) || !aligned16(product_out))
    return fail(FFQ_ERR_DTYPE, "gate/up while estimating: needs N %% 128 == 0, K %% 128 == 0, K >= 256, >= 64 tiles of 256 x 256 and 16-byte aligned buffers");
  const size_t need = ffq_mlp_gate_up_w8a8_estimating_workspace_bytes(M, N, K);
  if (!workspace || workspace_bytes < need) return fail(FFQ_ERR_WORKSPACE, "gate/up while estimating needs %zu workspace bytes, got %zu", need, workspace_bytes);
  char* base = static_cast<char*>(workspace);
  int32_t* flag = reinterpret_cast<int32_t*>(base);
  int32_t* rs_gate = reinterpret_cast<int32_t*>(base + 256);
  int32_t* rs_up = rs_gate + N;
  void* lin_ws = base + 256 + ((2 * N * 4 + 255) & ~(int64_t)255);
  const size_t lin_bytes = ffq_linear_w8a8_workspace_bytes(M, N, K);
  mlp_inputs_agree_kernel<<<1, 1024, 0, s>>>(x_scale_gate, x_offset_gate, x_scale_up, x_offset_up, gate_w_offset, up_w_offset, (int)N, flag);
  // the weight row sums serve both routes (the zero-point term of either x offset)
  rowsum_i8_kernel<<<(unsigned)((N + 3) / 4), 256, 0, s>>>(gate_wq, (int)N, (int)K, rs_gate, nullptr);
  rowsum_i8_kernel<<<(unsigned)((N + 3) / 4), 256, 0, s>>>(up_wq, (int)N, (int)K, rs_up, nullptr);
  int rc = check_launch("mlp_inputs_agree_kernel / rowsum_i8_kernel");
  if (rc) return rc;
  rc = mlp_gate_up_w8a8_impl(xq_gate, gate_wq, up_wq, rs_gate, rs_up, x_scale_gate, x_offset_gate ? x_offset_gate : nullptr, gate_w_scale, up_w_scale,
                             nullptr, nullptr, nullptr, 8.0, M, N, K, nullptr, 0, stream, product_out, extrema_words, extrema_pair, flag, 1);
  if (rc) return rc;
  rc = linear_w8a8_impl(xq_gate, gate_wq, rs_gate, x_scale_gate, x_offset_gate, 0, gate_w_scale, gate_w_offset, 1, nullptr, 0, gate_scratch, FFQ_BF16,
                        nullptr, nullptr, 8.0, 0, M, N, K, lin_ws, lin_bytes, stream, nullptr, nullptr, nullptr, flag, 0);
  if (rc) return rc;
  return linear_w8a8_impl(xq_up, up_wq, rs_up, x_scale_up, x_offset_up, 0, up_w_scale, up_w_offset, 1, nullptr, 0, product_out, FFQ_BF16, nullptr, nullptr,
                          8.0, 0, M, N, K, lin_ws, lin_bytes, stream, gate_scratch, extrema_words, extrema_pair, flag, 0,
                          EarlierCodes{xq_gate, x_scale_gate, x_offset_gate});
}

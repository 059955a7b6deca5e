// ffq_linear.hip — A6: W8A8 linear on the int8 matrix cores of gfx950.
//
// Replaces fallback.linear, src/fastforward/_gen/fallback.py:77-112: the reference dequantizes
// the activation codes and the weight codes into bf16 tensors (two extra HBM round trips, 3 B/elem
// of weight traffic each forward), runs a float GEMM and optionally re-quantizes. Here the integer
// codes feed v_mfma_i32_32x32x32_i8 directly, accumulate exactly in int32, and the affine
// parameters are applied once per output element in the epilogue:
//
//   y[m,n] = sx[m'] * sw[n'] * ( acc[m,n] + ox[m'] * rowsum_w[n] + ow[n'] * rowsum_x[m]
//                                + K * ox[m'] * ow[n'] )  (+ bias[n])
//
// with acc = sum_k xq[m,k] * wq[n,k], ox / ow = round_half_even(offset) (A2), and the row sums
// produced by a one-pass int8 reduction only when the corresponding offset exists.
//
// Layout: xq [M,K] and wq [N,K] are both K-contiguous, which is the operand order MFMA wants: lane
// (r = lane % 32, g = lane / 32) of a wavefront supplies 16 consecutive k-bytes of row r for both
// operands, so no transpose is ever needed. Block tile 128 x 128 x 64, 4 wavefronts (2 x 2), each
// owning 64 x 64 = 2 x 2 MFMA tiles; double-buffered LDS with a 16-byte-slot XOR swizzle
// (slot ^= (row >> 2) & 3) that makes every ds_read_b128 lane group hit 16 distinct slots.
// blockIdx is remapped so that consecutive tiles of one weight panel stay on one XCD's L2.
#include "ffq_affine.h"
#include "ffq_common.h"
#include "ffq_vec.h"
#include "ffq_silu.h"
#ifndef FFQ_EPI_NT
#define FFQ_EPI_NT 1  // epilogue stores of whole lines carry the non-temporal hint
#endif
#if FFQ_EPI_NT
#define FFQ_EPI_STORE(v, p) __builtin_nontemporal_store((v), (p))
#else
#define FFQ_EPI_STORE(v, p) (*(p) = (v))
#endif

#include <math.h>
#include <stdlib.h>

#include <type_traits>

#ifndef FFQ_X
#define FFQ_X 0  // experiment selector of the persistent kernel (tools/gemm_variants.sh): 0 = the shipped schedule
#endif

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
  const void* bias; int bias_dt;
  const void* residual;  // nullable, [M, N] of out_dt (half types): out = residual + T(linear), two roundings as the eager add (fq kernel only)
  void* out; int out_dt;
  const float* out_scale; const float* out_offset;
  float out_lo, out_hi;
  int x_per_row, w_per_row;
  int M, N, K;
  int tiles_m, tiles_n;
  int group_m;  // row tiles per group of the persistent kernels' tile walk (A panels shared by a group's column tiles)
  // MLP mode of the v3 kernel (gate and up projections in one launch): the second weight matrix
  const int8_t* wq2; const float* w_scale2; const int32_t* rowsum_w2;
  int debug;  // FFQ_GEMM_DEBUG ablation bits (tools/gemm_time.py): 1 = no global stores, 2 = no epilogue at all.
              // Measured: sc1 / nt / sc0 sc1 policies on the output stores change nothing (the store burst is HBM-write-bound).
};

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

template <typename TOut, bool REQUANT>
__global__ __launch_bounds__(256) void w8a8_gemm_kernel(LinearArgs a) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[2][2][kTileBytes];

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
    if (kt + 1 < ksteps) stash(cur ^ 1);
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
    const float rsw = a.rowsum_w ? (float)a.rowsum_w[n] : 0.0f;
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
          // the reference rounds the linear output to bf16 before the output quantizer sees it
          y = bf16_bits_to_f32(f32_to_bf16_bits(y));
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
// v2: 256 x 256 x 64 block tile, 8 wavefronts (2 x 4), each owning 128 x 64 = 4 x 2 MFMA tiles.
//
// Why a second kernel: the 128^2 tile moves (BM + BN) / (2 BM BN) = 1/128 byte per op from L2 into
// LDS — 9 TB/s at the 1.15 POP/s it reaches, i.e. it is L2-bound. 256^2 halves that, and:
//   * operands go global -> LDS directly (global_load_lds_dwordx4: no staging VGPRs, no ds_write);
//     the LDS image is lane-linear per wave instruction, so the bank swizzle is applied to the
//     per-lane SOURCE address and, identically, to the ds_read address;
//   * 3-stage LDS ring, loads for tile k+2 issued right after the barrier of tile k, waited for with
//     a COUNTED s_waitcnt vmcnt(4) (tile k+1 stays in flight across the barrier), one raw s_barrier
//     per K-step;
//   * the zero-point row sums sum_k wq[n,k] are accumulated with v_dot4 on the B fragments already in
//     registers (VALU work in the shadow of the MFMAs) — no extra pass over the weight codes;
//   * tiles are visited in groups of 8 M-tiles x all N (within each XCD's contiguous range), so the
//     32 CUs of an XCD share 8 activation panels and 4 weight panels per K-slice in their L2.
// Needs K % 64 == 0 (no K tail in the DMA path); M / N tails are handled by clamped loads + guarded
// stores.
constexpr int BM2 = 256, BK2 = 64, STAGES2 = 3;
constexpr int GROUP_M2 = 8;
// Two shapes of the same kernel (each wave always owns 128 x 64):
//   NW = 8: block 256 x 256, one block per CU  (1/256 B of L2->LDS traffic per op)
//   NW = 4: block 256 x 128, TWO blocks per CU (1/171 B per op) — the two blocks have independent
//           barriers, so one block's MFMAs fill the matrix pipe while the other one synchronises.

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

// Epilogue shared by the 256-row kernels: scale / zero-point terms, LDS transpose, 16-byte stores.
template <typename TOut, bool REQUANT, bool WOFF, int NW>
__device__ __forceinline__ void gemm256_epilogue(const LinearArgs& a, v16i (&acc)[4][2], int (&rsw)[2], int (&rsx_acc)[4],
                                                 uint8_t* lds2, int wave, int lane, int wm, int wn, int m0, int n0) {
  // Epilogue. The weight fragment is the MFMA's A operand, so with the 32x32 C/D layout
  // (col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)) lane l holds, for each (i, j, q):
  //   C[m = i*32 + (l & 31)][n = j*32 + 8*q + 4*(l >> 5) + (0..3)],  e = 4*q + (0..3)
  // i.e. FOUR CONSECUTIVE output columns of one row: 8 B of bf16 that go to LDS as one ds_write_b64
  // (32 per lane instead of 128 two-byte writes) and leave as full 16 B per lane / 128 B per line.
  TOut* out = static_cast<TOut*>(a.out);
  const float kf = (float)a.K;
  float oscale = 1.0f, ooff = 0.0f;
  if constexpr (REQUANT) {
    oscale = a.out_scale[0];
    ooff = a.out_offset ? rne(a.out_offset[0]) : 0.0f;
  }
  if (a.debug & 2) {  // ablation: keep the accumulators alive, do nothing with them
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(acc[i][j]));
    return;
  }
  __syncthreads();  // every wave is done with the operand ring: LDS is free for the epilogue
  constexpr int ROW_BYTES = 144;  // 128 B payload + 16 B pad
  constexpr int REGION_BYTES = 128 * ROW_BYTES + 256;
  uint8_t* region = lds2 + wave * REGION_BYTES;
  float* rs_lds = reinterpret_cast<float*>(region + 128 * ROW_BYTES);
  if (lane < 32) {
    rs_lds[lane] = (float)rsw[0];
    rs_lds[32 + lane] = (float)rsw[1];
  }
  const int g = lane >> 5;
  const int wave_n0 = n0 + wn * 64;
  const int wave_m0 = m0 + wm * 128;
  const bool lds_path = sizeof(TOut) == 2 && (a.N & 7) == 0 && wave_n0 + 64 <= a.N;

  // per-row (activation side) parameters of this lane's 4 rows m = wave_m0 + i*32 + (lane & 31)
  float sx[4], ox[4], rsx[4];
  bool m_ok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int m = wave_m0 + i * 32 + (lane & 31);
    m_ok[i] = m < a.M;
    m = m_ok[i] ? m : a.M - 1;
    sx[i] = a.x_scale[a.x_per_row ? m : 0];
    ox[i] = a.x_offset ? rne(a.x_offset[a.x_per_row ? m : 0]) : 0.0f;
    rsx[i] = WOFF ? (float)rsx_acc[i] : 0.0f;  // this lane's MFMA column IS its activation row
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      asm volatile("" ::: "memory");  // keep the parameter quads of different (j, q) from being hoisted together
      const int nb = j * 32 + 8 * q + 4 * g;  // this lane's 4 columns: wave_n0 + nb + (0..3)
      float sw4[4], ow4[4], rs4[4], b4[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        int n = wave_n0 + nb + t;
        n = n < a.N ? n : a.N - 1;
        sw4[t] = a.w_scale[a.w_per_row ? n : 0];
        ow4[t] = a.w_offset ? rne(a.w_offset[a.w_per_row ? n : 0]) : 0.0f;
        rs4[t] = rs_lds[nb + t];
        b4[t] = a.bias ? (float)load_any(a.bias, a.bias_dt, n) : 0.0f;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float y[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          float v = (float)acc[i][j][4 * q + t] + ox[i] * rs4[t];
          v = v + ow4[t] * rsx[i];
          v = v + kf * ox[i] * ow4[t];
          float r = (sx[i] * sw4[t]) * v;
          if (a.bias) r = r + b4[t];
          if constexpr (REQUANT) {
            r = bf16_bits_to_f32(f32_to_bf16_bits(r));
            r = clamp_nan(rne(r / oscale - ooff), a.out_lo, a.out_hi);
          }
          y[t] = r;
        }
        if constexpr (sizeof(TOut) == 2) {
          if (lds_path) {
            u32x2 pk;
            pk.x = pack2<TOut>(y[0], y[1]);
            pk.y = pack2<TOut>(y[2], y[3]);
            *reinterpret_cast<u32x2*>(region + (i * 32 + (lane & 31)) * ROW_BYTES + nb * 2) = pk;
            continue;
          }
        }
        if (m_ok[i]) {
          const size_t at = (size_t)(wave_m0 + i * 32 + (lane & 31)) * a.N + wave_n0 + nb;
#pragma unroll
          for (int t = 0; t < 4; ++t)
            if (wave_n0 + nb + t < a.N) store_out<TOut>(out + at + t, y[t]);
        }
      }
    }
  }
  if (lds_path) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private region: only this wave's own writes
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int c = lane + 64 * t;
      const int row = c >> 3, seg = c & 7;
      const int m = wave_m0 + row;
      const u32x4 v = *reinterpret_cast<const u32x4*>(region + row * ROW_BYTES + seg * 16);
      if (m < a.M && !(a.debug & 1))
        FFQ_EPI_STORE(v, reinterpret_cast<u32x4*>(reinterpret_cast<uint8_t*>(out) + ((size_t)m * a.N + wave_n0) * 2 + seg * 16));
      if (a.debug & 1) asm volatile("" ::"v"(v));
    }
  }
}

template <typename TOut, bool REQUANT, int NW, bool WOFF>
__global__ __launch_bounds__(NW * 64, 2) void w8a8_gemm256_kernel(LinearArgs a) {
  constexpr int BN2 = NW * 32;                       // 256 or 128 columns per block
  constexpr int WAVES_N = BN2 / 64;                  // 4 or 2
  constexpr int A_BYTES = BM2 * BK2;                 // 16 KiB
  constexpr int OPER_BYTES2 = A_BYTES;               // offset of the B operand inside a stage
  constexpr int STAGE_BYTES2 = (BM2 + BN2) * BK2;    // 32 or 24 KiB
  constexpr int A_CHUNKS = (BM2 / 16) / NW;          // 16-row DMA chunks of A per wave: 2 or 4
  constexpr int B_CHUNKS = (BN2 / 16) / NW;          // ... of B per wave: 2
  constexpr int DMA_PER_STEP = A_CHUNKS + B_CHUNKS;  // LDS-DMA instructions per wave per K-step
  extern __shared__ __attribute__((aligned(16))) uint8_t lds2[];

  // XCD-aware, grouped tile order
  const uint32_t nblk = gridDim.x;
  const uint32_t xcd = blockIdx.x & 7u, slot_in_xcd = blockIdx.x >> 3;
  const uint32_t q = nblk >> 3, r = nblk & 7u;
  const uint32_t tile_id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot_in_xcd;
  const uint32_t per_group = GROUP_M2 * (uint32_t)a.tiles_n;
  const uint32_t group = tile_id / per_group, in_group = tile_id - group * per_group;
  const uint32_t group_rows = min((uint32_t)GROUP_M2, (uint32_t)a.tiles_m - group * GROUP_M2);
  const int tm = (int)(group * GROUP_M2 + in_group % group_rows);
  const int tn = (int)(in_group / group_rows);
  const int m0 = tm * BM2, n0 = tn * BN2;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  // DMA map: wave w copies the 16-row chunks {A_CHUNKS w ...} of A and {B_CHUNKS w ...} of B. Inside a
  // chunk lane l lands at LDS slot l (16 B units): row = l / 4, physical k-slot = l % 4,
  // logical k-slot = physical ^ swz(row).
  const int d_row = lane >> 2;
  const int d_slot = (lane & 3) ^ ((d_row >> 2) & 3);
  const int8_t* a_src[A_CHUNKS];
  const int8_t* b_src[B_CHUNKS];
#pragma unroll
  for (int c = 0; c < A_CHUNKS; ++c) {
    int ra = m0 + (wave * A_CHUNKS + c) * 16 + d_row;
    ra = ra < a.M ? ra : a.M - 1;  // rows past the edge are loaded from the last row and never stored
    a_src[c] = a.xq + (size_t)ra * a.K + d_slot * 16;
  }
#pragma unroll
  for (int c = 0; c < B_CHUNKS; ++c) {
    int rb = n0 + (wave * B_CHUNKS + c) * 16 + d_row;
    rb = rb < a.N ? rb : a.N - 1;
    b_src[c] = a.wq + (size_t)rb * a.K + d_slot * 16;
  }
  const int last_tile = a.K / BK2 - 1;
  // Tiles past the end are re-loads of the last tile into a stage nobody reads any more: it keeps
  // the K-loop free of branches (fixed DMA count per iteration => one constant vmcnt).
  auto issue = [&](int kt, int stage) {
    kt = kt < last_tile ? kt : last_tile;
    uint8_t* base = lds2 + stage * STAGE_BYTES2;
#pragma unroll
    for (int c = 0; c < A_CHUNKS; ++c)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(a_src[c] + kt * BK2), (lds_void_t*)(base + (wave * A_CHUNKS + c) * 1024), 16, 0, 0);
#pragma unroll
    for (int c = 0; c < B_CHUNKS; ++c)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(b_src[c] + kt * BK2), (lds_void_t*)(base + OPER_BYTES2 + (wave * B_CHUNKS + c) * 1024), 16, 0, 0);
  };

  v16i acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0;
  int rsw[2] = {0, 0};
  int rsx_acc[4] = {0, 0, 0, 0};  // WOFF: sum_k xq[m,k] for this lane's 4 activation rows, same trick

  const int ksteps = a.K / BK2;
  issue(0, 0);
  issue(1, 1);
  issue(2, 2);

  const uint32_t frag_row = lane & 31, frag_g = lane >> 5;
  // per-lane LDS byte offsets of the fragments (row * 64 + swizzled slot * 16), kk = 0 / 1
  uint32_t a_off[4][2], b_off[2][2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t row = wm * 128 + i * 32 + frag_row;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) a_off[i][kk] = row * BK2 + (((kk * 2 + frag_g) ^ ((row >> 2) & 3u)) << 4);
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const uint32_t row = wn * 64 + j * 32 + frag_row;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) b_off[j][kk] = OPER_BYTES2 + row * BK2 + (((kk * 2 + frag_g) ^ ((row >> 2) & 3u)) << 4);
  }

  // Schedule of one K-step (tile kt lives in stage kt % 3; two fragment register sets):
  //   first half : 8 MFMAs on set0 (tile kt, k-half 0); the 6 ds_reads of set1 (tile kt, k-half 1)
  //                are slotted between them
  //   middle     : counted s_waitcnt — tile kt+1 has landed while tile kt+2 stays in flight —
  //                plus lgkmcnt(0), then ONE raw s_barrier: from here on nobody reads stage kt % 3
  //   second half: 8 MFMAs on set1 with, slotted between them, the 6 ds_reads of set0 for tile kt+1
  //                and the 4 LDS-DMA instructions that refill stage kt % 3 with tile kt+3
  // so every ds_read and every DMA issue sits in the shadow of an MFMA, DMA runs two K-steps ahead,
  // and hipcc's own (conservative) lgkmcnt(0) for a fragment set lands on the first MFMA of a half,
  // where no newer LDS read is outstanding.
  v4i fa0[4], fb0[2], fa1[4], fb1[2];
  auto read_frags = [&](const uint8_t* st, int kk, v4i (&fa)[4], v4i (&fb)[2]) {
#pragma unroll
    for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const v4i*>(st + b_off[j][kk]);
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const v4i*>(st + a_off[i][kk]);
  };
  auto rowsums = [&](auto with_x, const v4i (&fa)[4], const v4i (&fb)[2]) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      rsw[j] = __builtin_amdgcn_sdot4(fb[j].x, 0x01010101, rsw[j], false);
      rsw[j] = __builtin_amdgcn_sdot4(fb[j].y, 0x01010101, rsw[j], false);
      rsw[j] = __builtin_amdgcn_sdot4(fb[j].z, 0x01010101, rsw[j], false);
      rsw[j] = __builtin_amdgcn_sdot4(fb[j].w, 0x01010101, rsw[j], false);
    }
    if constexpr (decltype(with_x)::value) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        rsx_acc[i] = __builtin_amdgcn_sdot4(fa[i].x, 0x01010101, rsx_acc[i], false);
        rsx_acc[i] = __builtin_amdgcn_sdot4(fa[i].y, 0x01010101, rsx_acc[i], false);
        rsx_acc[i] = __builtin_amdgcn_sdot4(fa[i].z, 0x01010101, rsx_acc[i], false);
        rsx_acc[i] = __builtin_amdgcn_sdot4(fa[i].w, 0x01010101, rsx_acc[i], false);
      }
    }
  };
  auto mfma_rest = [&](const v4i (&fa)[4], const v4i (&fb)[2]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        if (i != 0 || j != 0) acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fb[j], fa[i], acc[i][j], 0, 0, 0);
  };

  if constexpr (DMA_PER_STEP == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  read_frags(lds2, 0, fa0, fb0);

  // The activation row sums (needed only for the ow * sum_k xq term) cost 32 extra v_dot4 per K-step
  // and measurably slow the loop (+7 % per launch), while symmetric weight quantizers carry an offset
  // BUFFER that is all zeros (reference nn/linear_quantizer.py:164-170). The block therefore looks at the
  // offsets of ITS columns on the device and takes the loop without them when they are all zero —
  // no host synchronisation, same result.
  bool need_x_sums = false;
  if constexpr (WOFF) {
    int n = n0 + (tid % BN2);
    n = n < a.N ? n : a.N - 1;
    need_x_sums = __syncthreads_or(rne(a.w_offset[a.w_per_row ? n : 0]) != 0.0f) != 0;
  }
  auto k_loop = [&](auto with_x) {
    int stage = 0;
    for (int kt = 0; kt < ksteps; ++kt) {
      const int next = stage + 1 == STAGES2 ? 0 : stage + 1;
      // ---- first half
      __builtin_amdgcn_s_setprio(1);
      rowsums(with_x, fa0, fb0);
      acc[0][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fb0[0], fa0[0], acc[0][0], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      read_frags(lds2 + stage * STAGE_BYTES2, 1, fa1, fb1);
      mfma_rest(fa0, fb0);
  #pragma unroll
      for (int g = 0; g < 6; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(0);
      // ---- middle: tile kt+1 landed (tile kt+2 stays in flight), my LDS reads done, everybody here
      if constexpr (DMA_PER_STEP == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      // ---- second half (the reads of the last iteration fetch a stale stage and are never used)
      __builtin_amdgcn_s_setprio(1);
      rowsums(with_x, fa1, fb1);
      acc[0][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fb1[0], fa1[0], acc[0][0], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      read_frags(lds2 + next * STAGE_BYTES2, 0, fa0, fb0);
      issue(kt + 3, stage);
      mfma_rest(fa1, fb1);
  #pragma unroll
      for (int g = 0; g < 3; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x006, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x010, DMA_PER_STEP == 4 ? 1 : 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x006, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x010, DMA_PER_STEP == 4 ? 1 : 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(0);
      stage = next;
    }
  };
  if (need_x_sums) k_loop(std::true_type{});
  else k_loop(std::false_type{});
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the trailing dummy DMA must not land in the epilogue's LDS
  // lanes l and l+32 hold the two k-halves of the same weight row
#pragma unroll
  for (int j = 0; j < 2; ++j) rsw[j] += __shfl_xor(rsw[j], 32, 64);
  if constexpr (WOFF) {
#pragma unroll
    for (int i = 0; i < 4; ++i) rsx_acc[i] += __shfl_xor(rsx_acc[i], 32, 64);
  }

  gemm256_epilogue<TOut, REQUANT, WOFF, NW>(a, acc, rsw, rsx_acc, lds2, wave, lane, wm, wn, m0, n0);
}

// Epilogue of the MLP mode: acc[i][0] holds gate_proj and acc[i][1] up_proj for the SAME 32 output columns, so
//   z = bf16(silu(bf16(gate))) * bf16(up)  (bf16),  codes = A1(z; out_scale, out_offset)
// is formed in registers with exactly the roundings of the three-launch path (GEMM epilogue -> bf16 tensors ->
// silu_mul_quantize_kernel), goes through ONE block-wide LDS tile [256][128 B] and leaves as full 128-byte lines
// of int8 codes: a quarter of the bytes of one bf16 projection, instead of two.
// `silu_table`: the LDS table of ffq_silu.h (the persistent kernel fills it once per launch), or nullptr = evaluate silu.
template <bool SAFE, bool TABLE>
__device__ __forceinline__ void mlp_epilogue_body(const LinearArgs& a, v16i (&acc)[4][2], int (&rsw)[2], uint8_t* lds2, int wave,
                                             int lane, int wm, int wn, int m0, int n0, const uint16_t* silu_table) {
  constexpr int PITCH = 144;  // 128 B of codes + 16 B pad
  const float sx = a.x_scale[0];
  const float ox = a.x_offset ? rne(a.x_offset[0]) : 0.0f;
  const float so = a.out_scale[0];
  const float oo = a.out_offset ? rne(a.out_offset[0]) : 0.0f;
  const Divider<1> div(so);
  __syncthreads();  // every wave is done with the operand ring
  // row sums of this lane's gate / up weight rows live in other lanes' registers: share them through LDS
  float* rs_lds = reinterpret_cast<float*>(lds2 + 256 * PITCH) + wave * 64;
  if (lane < 32) {
    rs_lds[lane] = (float)rsw[0];
    rs_lds[32 + lane] = (float)rsw[1];
  }
  const int g = lane >> 5;
  const int col0 = n0 + wn * 32;  // this wave's 32 output columns
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int cb = 8 * q + 4 * g;  // this lane's 4 columns: col0 + cb + (0..3)
    float swg[4], swu[4], rsg[4], rsu[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int n = col0 + cb + t;  // N % 128 == 0: always inside
      swg[t] = a.w_scale[n];
      swu[t] = a.w_scale2[n];
      rsg[t] = rs_lds[cb + t];
      rsu[t] = rs_lds[32 + cb + t];
    }
    // the bf16 tensors the two projections would have written (packed pairs), then bf16(silu(gate)) for all 8 pairs of
    // this column group: the table reads go out back to back, ONE wave-uniform branch covers the values outside its window
    uint32_t wg[4][2], ws[4][2];
    uint32_t bad = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int t = 0; t < 4; t += 2) {
        const float g0 = (sx * swg[t]) * ((float)acc[i][0][4 * q + t] + ox * rsg[t]);
        const float g1 = (sx * swg[t + 1]) * ((float)acc[i][0][4 * q + t + 1] + ox * rsg[t + 1]);
        wg[i][t >> 1] = pack2<bf16_t>(g0, g1);
#if FFQ_X == 10  // ablation: what the exact expf + IEEE division of silu cost in the MLP-mode launch (-7.5 %)
        ws[i][t >> 1] = pack2<bf16_t>(g0 * 0.5f, g1 * 0.5f);
#else
        if constexpr (TABLE) {
          ws[i][t >> 1] = silu_pair_lookup(wg[i][t >> 1], silu_table, bad);  // ffq_silu.h
        } else {                                                             // ATen's silu in fp32, rounded to bf16
          const uint32_t w = wg[i][t >> 1];
          ws[i][t >> 1] = pack2<bf16_t>(silu_exact(__builtin_bit_cast(float, w << 16)), silu_exact(__builtin_bit_cast(float, w & 0xFFFF0000u)));
        }
#endif
      }
    }
    if constexpr (TABLE) {
      if (__builtin_expect(silu_any_outside(bad), 0)) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int h = 0; h < 2; ++h) ws[i][h] = silu_pair_patch(wg[i][h], ws[i][h]);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int c[4];
#pragma unroll
      for (int t = 0; t < 4; t += 2) {
        uint32_t w = pack2<bf16_t>((sx * swu[t]) * ((float)acc[i][1][4 * q + t] + ox * rsu[t]),
                                   (sx * swu[t + 1]) * ((float)acc[i][1][4 * q + t + 1] + ox * rsu[t + 1]));
        const float u0 = __builtin_bit_cast(float, w << 16), u1 = __builtin_bit_cast(float, w & 0xFFFF0000u);
        w = ws[i][t >> 1];
        const float a0 = __builtin_bit_cast(float, w << 16), a1 = __builtin_bit_cast(float, w & 0xFFFF0000u);
        float z0 = a0 * u0, z1 = a1 * u1;
        w = pack2<bf16_t>(z0, z1);
        z0 = __builtin_bit_cast(float, w << 16); z1 = __builtin_bit_cast(float, w & 0xFFFF0000u);
        // SAFE: the Markstein division of ffq_affine.h (scale inside its no-underflow window), else the IEEE sequence;
        // decided once per launch — a per-element select would evaluate both
        const float r0 = SAFE ? rne(div.fast(z0) - oo) : rne(z0 / so - oo);
        const float r1 = SAFE ? rne(div.fast(z1) - oo) : rne(z1 / so - oo);
        int c0 = (int)r0, c1 = (int)r1;  // v_cvt_i32_f32: NaN -> 0, the int8 container's value
        const int lo = (int)a.out_lo, hi = (int)a.out_hi;
        c[t] = c0 < lo ? lo : (c0 > hi ? hi : c0);
        c[t + 1] = c1 < lo ? lo : (c1 > hi ? hi : c1);
      }
      const int row = wm * 128 + i * 32 + (lane & 31);
      *reinterpret_cast<uint32_t*>(lds2 + row * PITCH + wn * 32 + cb) = pack_bytes(c[0], c[1], c[2], c[3]);
    }
  }
  __syncthreads();
  int8_t* out = static_cast<int8_t*>(a.out);
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int row = wave * 32 + t * 8 + (lane >> 3), seg = lane & 7;
    const int m = m0 + row;
    const u32x4 v = *reinterpret_cast<const u32x4*>(lds2 + row * PITCH + seg * 16);
    if (m < a.M && !(a.debug & 1)) FFQ_EPI_STORE(v, reinterpret_cast<u32x4*>(out + (size_t)m * a.N + n0 + seg * 16));
  }
}

template <bool TABLE = false>
__device__ __forceinline__ void mlp_epilogue(const LinearArgs& a, v16i (&acc)[4][2], int (&rsw)[2], uint8_t* lds2, int wave,
                                             int lane, int wm, int wn, int m0, int n0, const uint16_t* silu_table = nullptr) {
  const float as = __builtin_fabsf(a.out_scale[0]);
  if (as > 0x1p-40f && as < 0x1p40f) mlp_epilogue_body<true, TABLE>(a, acc, rsw, lds2, wave, lane, wm, wn, m0, n0, silu_table);
  else mlp_epilogue_body<false, TABLE>(a, acc, rsw, lds2, wave, lane, wm, wn, m0, n0, silu_table);
}

// -------------------------------------------------------------------------------------------------
// v3 ("ping-pong"): same 256 x 256 x 64 tile, operand image, swizzle and epilogue as v2, different
// K-loop. v2 lets every wave interleave its own ds_reads, LDS-DMA issues and v_dot4 row sums with its
// own MFMAs, and all 8 waves run the same half-step at the same time (PMC: 27 % of wave time parked at
// s_waitcnt / s_barrier, MFMA busy 55 %). Ablations on the MI355X (tools/gemm_variants.sh, no epilogue,
// down_proj shape): v2 2.23 POP/s; MFMA + barriers alone 3.29; the in-loop v_dot4c row sums cost 18 %,
// LDS-DMA issued next to the ds_reads 20 %. Hence:
//   * a half-step is a LOAD segment (the 6 ds_read_b128 of ONE fragment set) and an MFMA cluster
//     (8 MFMAs under s_setprio 1) separated by raw barriers, and the upper wave group (waves 4-7, the
//     second wave of every SIMD) runs one barrier interval behind the lower group: while one wave of a
//     SIMD feeds the matrix pipe its partner reads LDS — the 8-phase template of the CDNA GEMM
//     playbook restated for int8 32x32x32;
//          interval n     : group 0  C(p)   | group 1  L(p)
//          interval n + 1 : group 0  L(p+1) | group 1  C(p)
//   * the two LDS-DMA issues of a half-step sit INSIDE the cluster, after its second MFMA (3.0 POP/s
//     vs 2.3 with the DMA in the load segment);
//   * the weight row sums of the zero-point term come from a one-pass side kernel (1 B/elem of the
//     weight, once per launch instead of once per M-tile); only a weight offset that is really
//     non-zero (checked on the device) turns the in-loop activation row sums back on.
// Ring of 4 stages (128 KiB): tile kt lives in stage kt & 3. Cluster C(2kt) issues the B half of tile
// kt+2, C(2kt+1) the A half of tile kt+3; the load segment L(2kt+1) waits with vmcnt(4) — everything
// but tile kt+2 has landed, i.e. tile kt+1 — and tile kt+1 is first read one phase (two barriers)
// later. A stage is re-filled at the earliest two full intervals after the slower group retired its
// last read of it. These are the two ordering rules of the playbook (RAW: wait -> barrier -> read;
// WAR: read retired -> barrier -> DMA issue).
constexpr int STAGES3 = 4;

template <typename TOut, bool REQUANT, bool WOFF, bool MLP = false>
__global__ __launch_bounds__(512, 2) void w8a8_gemm256pp_kernel(LinearArgs a) {
  constexpr int NW = 8, BN2 = 256, WAVES_N = 4;
  constexpr int BN_OUT = MLP ? 128 : 256;  // output columns per block (MLP: 128 gate rows + 128 up rows in the B tile)
  constexpr int A_BYTES = BM2 * BK2;
  constexpr int OPER_BYTES2 = A_BYTES;
  constexpr int STAGE_BYTES2 = (BM2 + BN2) * BK2;  // 32 KiB
  extern __shared__ __attribute__((aligned(16))) uint8_t lds2[];

  const uint32_t nblk = gridDim.x;
  const uint32_t xcd = blockIdx.x & 7u, slot_in_xcd = blockIdx.x >> 3;
  const uint32_t q = nblk >> 3, r = nblk & 7u;
  const uint32_t tile_id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot_in_xcd;
  const uint32_t per_group = GROUP_M2 * (uint32_t)a.tiles_n;
  const uint32_t group = tile_id / per_group, in_group = tile_id - group * per_group;
  const uint32_t group_rows = min((uint32_t)GROUP_M2, (uint32_t)a.tiles_m - group * GROUP_M2);
  const int tm = (int)(group * GROUP_M2 + in_group % group_rows);
  const int tn = (int)(in_group / group_rows);
  const int m0 = tm * BM2, n0 = tn * BN_OUT;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;  // wm is also the ping-pong group

  // DMA map as in v2: wave w copies the 16-row chunks {2w, 2w+1} of A and of B.
  const int d_row = lane >> 2;
  const int d_slot = (lane & 3) ^ ((d_row >> 2) & 3);
  const int8_t* a_src[2];
  const int8_t* b_src[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    int ra = m0 + (wave * 2 + c) * 16 + d_row;
    ra = ra < a.M ? ra : a.M - 1;
    a_src[c] = a.xq + (size_t)ra * a.K + d_slot * 16;
    if constexpr (MLP) {
      // B-tile rows [64 wn', 64 wn' + 32) are gate rows n0 + 32 wn' + (0..31), the next 32 the same rows of up
      const int chunk = wave * 2 + c;           // 16-row chunk of the B tile
      const int within = (chunk & 3) * 16;      // row inside the wave-column's 64 rows
      const int rb = n0 + (chunk >> 2) * 32 + (within & 31) + d_row;
      b_src[c] = (within < 32 ? a.wq : a.wq2) + (size_t)rb * a.K + d_slot * 16;
    } else {
      int rb = n0 + (wave * 2 + c) * 16 + d_row;
      rb = rb < a.N ? rb : a.N - 1;
      b_src[c] = a.wq + (size_t)rb * a.K + d_slot * 16;
    }
  }
  const int last_tile = a.K / BK2 - 1;
  // Tiles past the end re-load the last tile into a stage nobody reads any more (constant vmcnt).
  auto issue_a = [&](int kt) {
    const int stage = kt & (STAGES3 - 1);
    kt = kt < last_tile ? kt : last_tile;
    uint8_t* base = lds2 + stage * STAGE_BYTES2;
#pragma unroll
    for (int c = 0; c < 2; ++c)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(a_src[c] + kt * BK2), (lds_void_t*)(base + (wave * 2 + c) * 1024), 16, 0, 0);
  };
  auto issue_b = [&](int kt) {
    const int stage = kt & (STAGES3 - 1);
    kt = kt < last_tile ? kt : last_tile;
    uint8_t* base = lds2 + stage * STAGE_BYTES2 + OPER_BYTES2;
#pragma unroll
    for (int c = 0; c < 2; ++c)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(b_src[c] + kt * BK2), (lds_void_t*)(base + (wave * 2 + c) * 1024), 16, 0, 0);
  };

  v16i acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0;
  int rsx_acc[4] = {0, 0, 0, 0};

  // Symmetric weight quantizers carry an all-zero offset BUFFER (reference nn/linear_quantizer.py:164-170):
  // look at this block's offsets on the device — before any DMA is in flight, the compiler drains vmcnt
  // for an ordinary load — and take the loop without activation row sums when they are all zero.
  bool need_x_sums = false;
  if constexpr (WOFF) {
    int n = n0 + (tid % BN2);
    n = n < a.N ? n : a.N - 1;
    need_x_sums = __syncthreads_or(rne(a.w_offset[a.w_per_row ? n : 0]) != 0.0f) != 0;
  }
  // sum_k wq[n, k] of this lane's two weight rows (side kernel; only read when x has an offset)
  int rsw[2] = {0, 0};
  if (a.rowsum_w) {
    if constexpr (MLP) {
      rsw[0] = a.rowsum_w[n0 + wn * 32 + (lane & 31)];
      rsw[1] = a.rowsum_w2[n0 + wn * 32 + (lane & 31)];
    } else {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        int n = n0 + wn * 64 + j * 32 + (lane & 31);
        n = n < a.N ? n : a.N - 1;
        rsw[j] = a.rowsum_w[n];
      }
    }
  }

  const int ksteps = a.K / BK2;
  // prologue: tiles 0 and 1 entirely, A half of tile 2
  issue_a(0); issue_b(0);
  issue_a(1); issue_b(1);
  issue_a(2);

  const uint32_t frag_row = lane & 31, frag_g = lane >> 5;
  uint32_t a_off[4][2], b_off[2][2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t row = wm * 128 + i * 32 + frag_row;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) a_off[i][kk] = row * BK2 + (((kk * 2 + frag_g) ^ ((row >> 2) & 3u)) << 4);
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const uint32_t row = wn * 64 + j * 32 + frag_row;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) b_off[j][kk] = OPER_BYTES2 + row * BK2 + (((kk * 2 + frag_g) ^ ((row >> 2) & 3u)) << 4);
  }

  v4i fa[4], fb[2];
  auto read_frags = [&](const uint8_t* st, int kk) {
#pragma unroll
    for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const v4i*>(st + b_off[j][kk]);
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const v4i*>(st + a_off[i][kk]);
  };
  auto cluster = [&](auto with_x, auto dma) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fb[j], fa[i], acc[i][j], 0, 0, 0);
      if (i == 0) {
        __builtin_amdgcn_sched_barrier(0);
        dma();
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if constexpr (decltype(with_x)::value) {  // rare: a weight offset that is really non-zero
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        rsx_acc[i] = __builtin_amdgcn_sdot4(fa[i].x, 0x01010101, rsx_acc[i], false);
        rsx_acc[i] = __builtin_amdgcn_sdot4(fa[i].y, 0x01010101, rsx_acc[i], false);
        rsx_acc[i] = __builtin_amdgcn_sdot4(fa[i].z, 0x01010101, rsx_acc[i], false);
        rsx_acc[i] = __builtin_amdgcn_sdot4(fa[i].w, 0x01010101, rsx_acc[i], false);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(0);
  };

  // tile 0 has landed (this wave's share): tile 1 (4) and the A half of tile 2 (2) stay in flight
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_barrier();  // the upper group runs one interval behind

  auto k_loop = [&](auto with_x) {
    for (int kt = 0; kt < ksteps; ++kt) {
      const uint8_t* st = lds2 + (kt & (STAGES3 - 1)) * STAGE_BYTES2;
      // ---- phase 2kt
      read_frags(st, 0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      cluster(with_x, [&] { issue_b(kt + 2); });
      __builtin_amdgcn_s_barrier();
      // ---- phase 2kt + 1
      read_frags(st, 1);
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // tile kt + 1 landed, tile kt + 2 stays in flight
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      cluster(with_x, [&] { issue_a(kt + 3); });
      __builtin_amdgcn_s_barrier();
    }
  };
  if (need_x_sums) k_loop(std::true_type{});
  else k_loop(std::false_type{});
  if (wm == 0) __builtin_amdgcn_s_barrier();  // same number of barriers for both groups
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // trailing dummy DMA must not land in the epilogue's LDS
  if constexpr (WOFF) {
#pragma unroll
    for (int i = 0; i < 4; ++i) rsx_acc[i] += __shfl_xor(rsx_acc[i], 32, 64);
  }
  if constexpr (MLP) mlp_epilogue(a, acc, rsw, lds2, wave, lane, wm, wn, m0, n0);
  else gemm256_epilogue<TOut, REQUANT, WOFF, NW>(a, acc, rsw, rsx_acc, lds2, wave, lane, wm, wn, m0, n0);
}

// -------------------------------------------------------------------------------------------------
// v3 with full-line staging ("fl"): the same ping-pong K-loop, but the LDS image holds 128 k-bytes per row
// (two K-steps) in two 64 KiB slots, so every LDS-DMA instruction moves 8 whole 128-byte cache lines.
// Needs K % 128 == 0.
template <typename TOut, bool REQUANT, bool WOFF, bool MLP = false>
__global__ __launch_bounds__(512, 2) void w8a8_gemm256fl_kernel(LinearArgs a) {
  constexpr int NW = 8, BN2 = 256, WAVES_N = 4;
  constexpr int BN_OUT = MLP ? 128 : 256;  // output columns per block (MLP: 128 gate rows + 128 up rows in the B tile)
  extern __shared__ __attribute__((aligned(16))) uint8_t lds2[];

  const uint32_t nblk = gridDim.x;
  const uint32_t xcd = blockIdx.x & 7u, slot_in_xcd = blockIdx.x >> 3;
  const uint32_t q = nblk >> 3, r = nblk & 7u;
  const uint32_t tile_id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot_in_xcd;
  const uint32_t per_group = GROUP_M2 * (uint32_t)a.tiles_n;
  const uint32_t group = tile_id / per_group, in_group = tile_id - group * per_group;
  const uint32_t group_rows = min((uint32_t)GROUP_M2, (uint32_t)a.tiles_m - group * GROUP_M2);
  const int tm = (int)(group * GROUP_M2 + in_group % group_rows);
  const int tn = (int)(in_group / group_rows);
  const int m0 = tm * BM2, n0 = tn * BN_OUT;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;  // wm is also the ping-pong group

  // DMA map: full 128-byte lines. A slot holds TWO K-steps (128 k-bytes per row): one LDS-DMA instruction copies
  // 8 rows x 128 B — 8 whole cache lines instead of the 16 half lines of the 64-byte-row image (half the TA / L2
  // requests per byte). Wave w copies the 8-row chunks {4w .. 4w+3} of A and of B. Inside a chunk lane l lands at
  // LDS slot l: row = l / 8, physical 16-B slot = l % 8, logical slot = physical ^ ((row >> 1) & 7).
  const int d_row = lane >> 3;
  const int8_t* a_src[4];
  const int8_t* b_src[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int row = (wave * 4 + c) * 8 + d_row;  // row inside the 256-row tile
    const int d_slot = (lane & 7) ^ ((row >> 1) & 7);
    int ra = m0 + row;
    ra = ra < a.M ? ra : a.M - 1;
    a_src[c] = a.xq + (size_t)ra * a.K + d_slot * 16;
    if constexpr (MLP) {
      const int rb = n0 + (row >> 6) * 32 + (row & 31);
      b_src[c] = ((row & 32) ? a.wq2 : a.wq) + (size_t)rb * a.K + d_slot * 16;
    } else {
      int rb = n0 + row;
      rb = rb < a.N ? rb : a.N - 1;
      b_src[c] = a.wq + (size_t)rb * a.K + d_slot * 16;
    }
  }
  constexpr int SLOT_BYTES = (BM2 + BN2) * 128;  // 64 KiB: A image (32 KiB) then B image
  constexpr int B_IMAGE = BM2 * 128;
  const int last_super = a.K / 128 - 1;
  // super-steps past the end re-load the last one into a slot nobody reads any more
  auto issue_a = [&](int ks, int c0) {
    const int slot = ks & 1;
    ks = ks < last_super ? ks : last_super;
    uint8_t* base = lds2 + slot * SLOT_BYTES;
#pragma unroll
    for (int c = c0; c < c0 + 2; ++c)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(a_src[c] + ks * 128), (lds_void_t*)(base + (wave * 4 + c) * 1024), 16, 0, 0);
  };
  auto issue_b = [&](int ks, int c0) {
    const int slot = ks & 1;
    ks = ks < last_super ? ks : last_super;
    uint8_t* base = lds2 + slot * SLOT_BYTES + B_IMAGE;
#pragma unroll
    for (int c = c0; c < c0 + 2; ++c)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(b_src[c] + ks * 128), (lds_void_t*)(base + (wave * 4 + c) * 1024), 16, 0, 0);
  };

  v16i acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0;
  int rsx_acc[4] = {0, 0, 0, 0};

  // Symmetric weight quantizers carry an all-zero offset BUFFER (reference nn/linear_quantizer.py:164-170):
  // look at this block's offsets on the device — before any DMA is in flight, the compiler drains vmcnt
  // for an ordinary load — and take the loop without activation row sums when they are all zero.
  bool need_x_sums = false;
  if constexpr (WOFF) {
    int n = n0 + (tid % BN2);
    n = n < a.N ? n : a.N - 1;
    need_x_sums = __syncthreads_or(rne(a.w_offset[a.w_per_row ? n : 0]) != 0.0f) != 0;
  }
  // sum_k wq[n, k] of this lane's two weight rows (side kernel; only read when x has an offset)
  int rsw[2] = {0, 0};
  if (a.rowsum_w) {
    if constexpr (MLP) {
      rsw[0] = a.rowsum_w[n0 + wn * 32 + (lane & 31)];
      rsw[1] = a.rowsum_w2[n0 + wn * 32 + (lane & 31)];
    } else {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        int n = n0 + wn * 64 + j * 32 + (lane & 31);
        n = n < a.N ? n : a.N - 1;
        rsw[j] = a.rowsum_w[n];
      }
    }
  }

  const int ksuper = a.K / 128;
  // prologue: super-step 0 (both K-steps of the first 128 k-bytes)
  issue_a(0, 0); issue_a(0, 2); issue_b(0, 0); issue_b(0, 2);

  const uint32_t frag_row = lane & 31, frag_g = lane >> 5;
  // fragment byte offsets inside a slot: row * 128 + swizzled 16-B slot; v = 4 p + 2 kk + g (p = K-step parity)
  uint32_t a_off[4][4], b_off[2][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t row = wm * 128 + i * 32 + frag_row;
#pragma unroll
    for (int v = 0; v < 4; ++v) a_off[i][v] = row * 128 + ((((v * 2) + frag_g) ^ ((row >> 1) & 7u)) << 4);
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const uint32_t row = wn * 64 + j * 32 + frag_row;
#pragma unroll
    for (int v = 0; v < 4; ++v) b_off[j][v] = B_IMAGE + row * 128 + ((((v * 2) + frag_g) ^ ((row >> 1) & 7u)) << 4);
  }

  v4i fa[4], fb[2];
  auto read_frags = [&](const uint8_t* st, int kk) {  // kk = 2 * (K-step parity) + k-half
#pragma unroll
    for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const v4i*>(st + b_off[j][kk]);
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const v4i*>(st + a_off[i][kk]);
  };
  auto cluster = [&](auto with_x, auto dma, auto dma2) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fb[j], fa[i], acc[i][j], 0, 0, 0);
      if (i == 0) {
        __builtin_amdgcn_sched_barrier(0);
        dma();
        __builtin_amdgcn_sched_barrier(0);
      }
      if (i == 2) {
        __builtin_amdgcn_sched_barrier(0);
        dma2();
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if constexpr (decltype(with_x)::value) {  // rare: a weight offset that is really non-zero
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        rsx_acc[i] = __builtin_amdgcn_sdot4(fa[i].x, 0x01010101, rsx_acc[i], false);
        rsx_acc[i] = __builtin_amdgcn_sdot4(fa[i].y, 0x01010101, rsx_acc[i], false);
        rsx_acc[i] = __builtin_amdgcn_sdot4(fa[i].z, 0x01010101, rsx_acc[i], false);
        rsx_acc[i] = __builtin_amdgcn_sdot4(fa[i].w, 0x01010101, rsx_acc[i], false);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(0);
  };

  // super-step 0 has landed (this wave's share)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_barrier();  // the upper group runs one interval behind

  // One iteration = one super-step = two K-steps = four phases. The slot that held super-step ks - 1 is re-filled
  // with super-step ks + 1 by the first two clusters (8 LDS-DMA per wave: two after the second MFMA of a cluster, two
  // after the sixth — +1..2 % over four in one place) and waited for with vmcnt(0) in the last load segment, one phase
  // before its first read; nothing newer is in flight then. WAR: the slot's last reads (phase 3 of the previous
  // iteration) were retired by the slower group one barrier before the faster group's first cluster.
  auto k_loop = [&](auto with_x) {
    for (int ks = 0; ks < ksuper; ++ks) {
      const uint8_t* st = lds2 + (ks & 1) * SLOT_BYTES;
      read_frags(st, 0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      cluster(with_x, [&] { issue_a(ks + 1, 0); }, [&] { issue_b(ks + 1, 0); });
      __builtin_amdgcn_s_barrier();
      read_frags(st, 1);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      cluster(with_x, [&] { issue_a(ks + 1, 2); }, [&] { issue_b(ks + 1, 2); });
      __builtin_amdgcn_s_barrier();
      read_frags(st, 2);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      cluster(with_x, [] {}, [] {});
      __builtin_amdgcn_s_barrier();
      read_frags(st, 3);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // super-step ks + 1 landed
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      cluster(with_x, [] {}, [] {});
      __builtin_amdgcn_s_barrier();
    }
  };
  if (need_x_sums) k_loop(std::true_type{});
  else k_loop(std::false_type{});
  if (wm == 0) __builtin_amdgcn_s_barrier();  // same number of barriers for both groups
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // trailing dummy DMA must not land in the epilogue's LDS
  if constexpr (WOFF) {
#pragma unroll
    for (int i = 0; i < 4; ++i) rsx_acc[i] += __shfl_xor(rsx_acc[i], 32, 64);
  }
  if constexpr (MLP) mlp_epilogue(a, acc, rsw, lds2, wave, lane, wm, wn, m0, n0);
  else gemm256_epilogue<TOut, REQUANT, WOFF, NW>(a, acc, rsw, rsx_acc, lds2, wave, lane, wm, wn, m0, n0);
}

// -------------------------------------------------------------------------------------------------
// v3 persistent ("fp"): one block per CU walks its tiles; the K-loop is the full-line ping-pong loop of
// w8a8_gemm256fl_kernel running straight across tile boundaries: the last iteration of a tile prefetches the first
// super-step of the NEXT tile, which then lands under the epilogue (the 64 KiB + latency prologue burst that every
// CU issues at the same moment otherwise costs ~3.5 us of a ~70 us tile). The epilogue works in the slot the tile
// has just consumed: four 32-row slabs per wave (plain mode) or the block-wide code tile (MLP mode).
// Plain (no weight offset) and MLP modes only; weight offsets take the non-persistent kernel.
template <typename TOut, bool REQUANT>
__device__ __forceinline__ void gemm256_epilogue_slabs(const LinearArgs& a, v16i (&acc)[4][2], uint8_t* scratch, int wave, int lane,
                                                       int wm, int wn, int m0, int n0) {
  TOut* out = static_cast<TOut*>(a.out);
  float oscale = 1.0f, ooff = 0.0f;
  if constexpr (REQUANT) {
    oscale = a.out_scale[0];
    ooff = a.out_offset ? rne(a.out_offset[0]) : 0.0f;
  }
  constexpr int ROW_BYTES = 144;
  constexpr int WAVE_BYTES = 32 * ROW_BYTES + 3 * 64 * 4;  // one 32-row slab + the wave's 64 columns' parameters
  uint8_t* region = scratch + wave * WAVE_BYTES;
  float* colp = reinterpret_cast<float*>(region + 32 * ROW_BYTES);  // [3][64]: weight scale, weight row sum, bias
  const int g = lane >> 5;
  const int wave_n0 = n0 + wn * 64;
  const int wave_m0 = m0 + wm * 128;
  const bool lds_path = sizeof(TOut) == 2 && (a.N & 7) == 0 && wave_n0 + 64 <= a.N;
  {
    int n = wave_n0 + lane;
    n = n < a.N ? n : a.N - 1;
    colp[lane] = a.w_scale[a.w_per_row ? n : 0];
    colp[64 + lane] = a.rowsum_w ? (float)a.rowsum_w[n] : 0.0f;
    colp[128 + lane] = a.bias ? (float)load_any(a.bias, a.bias_dt, n) : 0.0f;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private: only this wave's own writes
  // the activation parameters of all four slabs BEFORE the first store: a load inside the slab loop makes the compiler wait
  // with vmcnt(0), i.e. for the previous slab's global stores too (vmcnt counts stores on gfx9): three store round trips
  // per tile in series
  float sx4[4], ox4[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int m = wave_m0 + i * 32 + (lane & 31);
    m = m < a.M ? m : a.M - 1;
    sx4[i] = a.x_scale[a.x_per_row ? m : 0];
    ox4[i] = a.x_offset ? rne(a.x_offset[a.x_per_row ? m : 0]) : 0.0f;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(sx4[i]), "+v"(ox4[i]));  // a use: the compiler's waits for the loads land HERE
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int m = wave_m0 + i * 32 + (lane & 31);
    const bool m_ok = m < a.M;
    m = m_ok ? m : a.M - 1;
    const float sx = sx4[i], ox = ox4[i];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int nb = j * 32 + 8 * q + 4 * g;
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        const f32x4 sw4 = *reinterpret_cast<const f32x4*>(colp + nb);
        const f32x4 rs4 = *reinterpret_cast<const f32x4*>(colp + 64 + nb);
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(colp + 128 + nb);
        float y[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float v = (float)acc[i][j][4 * q + t] + ox * rs4[t];
          float r = (sx * sw4[t]) * v;
          if (a.bias) r = r + b4[t];
          if constexpr (REQUANT) {
            r = bf16_bits_to_f32(f32_to_bf16_bits(r));
            r = clamp_nan(rne(r / oscale - ooff), a.out_lo, a.out_hi);
          }
          y[t] = r;
        }
        if constexpr (sizeof(TOut) == 2) {
          if (lds_path) {
            u32x2 pk;
            pk.x = pack2<TOut>(y[0], y[1]);
            pk.y = pack2<TOut>(y[2], y[3]);
            *reinterpret_cast<u32x2*>(region + (lane & 31) * ROW_BYTES + nb * 2) = pk;
            continue;
          }
        }
        if (m_ok) {
          const size_t at = (size_t)m * a.N + wave_n0 + nb;
#pragma unroll
          for (int t = 0; t < 4; ++t)
            if (wave_n0 + nb + t < a.N) store_out<TOut>(out + at + t, y[t]);
        }
      }
    if (lds_path) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int c = lane + 64 * t;
        const int row = c >> 3, seg = c & 7;
        const int mm = wave_m0 + i * 32 + row;
        const u32x4 v = *reinterpret_cast<const u32x4*>(region + row * ROW_BYTES + seg * 16);
        // non-temporal: the output is not read again by this launch and must not push the operand panels out of L2
        // (gate/up shape: +5 % over plain stores, A/B on one box; the launch without its stores: +10 %)
        if (mm < a.M && !(a.debug & 1))
          FFQ_EPI_STORE(v, reinterpret_cast<u32x4*>(reinterpret_cast<uint8_t*>(out) + ((size_t)mm * a.N + wave_n0) * 2 + seg * 16));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the slab is re-written by the next i
    }
  }
}

template <typename TOut, bool REQUANT, bool MLP>
__global__ __launch_bounds__(512, 2) void w8a8_gemm256fp_kernel(LinearArgs a, int total_tiles) {
  constexpr int BN2 = 256, WAVES_N = 4;
  constexpr int BN_OUT = MLP ? 128 : 256;
  constexpr int SLOT_BYTES = (BM2 + BN2) * 128;
  constexpr int B_IMAGE = BM2 * 128;
  extern __shared__ __attribute__((aligned(16))) uint8_t lds2[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  // this block's tiles: XCD x (= blockIdx % 8) owns a contiguous range of the grouped tile order and its blocks walk
  // it round-robin, i.e. the order in which a non-persistent launch would dispatch them
  const uint32_t nblk = gridDim.x;
  const uint32_t xcd = blockIdx.x & 7u, j_in_xcd = blockIdx.x >> 3;
  const uint32_t blocks_in_xcd = (nblk >> 3) + (xcd < (nblk & 7u) ? 1u : 0u);
  const uint32_t tq = (uint32_t)total_tiles >> 3, tr = (uint32_t)total_tiles & 7u;
  const uint32_t xcd_first = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
  const uint32_t xcd_count = tq + (xcd < tr ? 1u : 0u);
  const int my_tiles = j_in_xcd < xcd_count ? (int)((xcd_count - j_in_xcd + blocks_in_xcd - 1) / blocks_in_xcd) : 0;
  if (my_tiles == 0) return;
  // MLP mode: silu over bf16 as a 16 KiB table behind the two operand slots, filled once per launch (ffq_silu.h);
  // the barriers of the first tile's K-loop publish it long before the first epilogue reads it
  uint16_t* const silu_table = reinterpret_cast<uint16_t*>(lds2 + 2 * SLOT_BYTES);
  if constexpr (MLP) silu_table_fill(silu_table, (uint32_t)tid, 512u);

  const int d_row = lane >> 3;
  const int8_t* a_src[4];
  const int8_t* b_src[4];
  int m0 = 0, n0 = 0;          // tile being computed
  auto tile_origin = [&](int it, int& tm0, int& tn0) {
    const uint32_t tile_id = xcd_first + j_in_xcd + (uint32_t)it * blocks_in_xcd;
    const uint32_t per_group = GROUP_M2 * (uint32_t)a.tiles_n;
    const uint32_t group = tile_id / per_group, in_group = tile_id - group * per_group;
    const uint32_t group_rows = min((uint32_t)GROUP_M2, (uint32_t)a.tiles_m - group * GROUP_M2);
    tm0 = (int)(group * GROUP_M2 + in_group % group_rows) * BM2;
    tn0 = (int)(in_group / group_rows) * BN_OUT;
  };
  auto set_sources = [&](int tm0, int tn0) {
#if FFQ_X == 1  // every tile streams the operands of tile (0, 0): all L2 hits (wrong results: cost of the misses)
    tm0 = 0; tn0 = 0;
#endif
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int row = (wave * 4 + c) * 8 + d_row;
      const int d_slot = (lane & 7) ^ ((row >> 1) & 7);
      int ra = tm0 + row;
      ra = ra < a.M ? ra : a.M - 1;
      a_src[c] = a.xq + (size_t)ra * a.K + d_slot * 16;
      if constexpr (MLP) {
        const int rb = tn0 + (row >> 6) * 32 + (row & 31);
        b_src[c] = ((row & 32) ? a.wq2 : a.wq) + (size_t)rb * a.K + d_slot * 16;
      } else {
        int rb = tn0 + row;
        rb = rb < a.N ? rb : a.N - 1;
        b_src[c] = a.wq + (size_t)rb * a.K + d_slot * 16;
      }
    }
  };
  // LDS-DMA of super-step `ks` of the tile the sources point at, into slot `slot`
  auto issue_a = [&](int ks, int slot, int c0) {
    uint8_t* base = lds2 + slot * SLOT_BYTES;
#pragma unroll
    for (int c = c0; c < c0 + 2; ++c)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(a_src[c] + ks * 128), (lds_void_t*)(base + (wave * 4 + c) * 1024), 16, 0, 0);
  };
  auto issue_b = [&](int ks, int slot, int c0) {
    uint8_t* base = lds2 + slot * SLOT_BYTES + B_IMAGE;
#pragma unroll
    for (int c = c0; c < c0 + 2; ++c)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(b_src[c] + ks * 128), (lds_void_t*)(base + (wave * 4 + c) * 1024), 16, 0, 0);
  };

  // piece p of a super-step: p < 4 the A chunk p, else the B chunk p - 4 (experiment schedules issue them one by one)
  auto issue_piece = [&](int ks, int slot, int p) {
    uint8_t* base = lds2 + slot * SLOT_BYTES + (p < 4 ? 0 : B_IMAGE);
    const int c = p & 3;
    __builtin_amdgcn_global_load_lds((gbl_void_t*)((p < 4 ? a_src[c] : b_src[c]) + ks * 128), (lds_void_t*)(base + (wave * 4 + c) * 1024), 16, 0, 0);
  };

  const uint32_t frag_row = lane & 31, frag_g = lane >> 5;
  uint32_t a_off[4][4], b_off[2][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t row = wm * 128 + i * 32 + frag_row;
#pragma unroll
    for (int v = 0; v < 4; ++v) a_off[i][v] = row * 128 + ((((v * 2) + frag_g) ^ ((row >> 1) & 7u)) << 4);
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const uint32_t row = wn * 64 + j * 32 + frag_row;
#pragma unroll
    for (int v = 0; v < 4; ++v) b_off[j][v] = B_IMAGE + row * 128 + ((((v * 2) + frag_g) ^ ((row >> 1) & 7u)) << 4);
  }

  v16i acc[4][2];
  v4i fa[4], fb[2];
  auto read_frags = [&](const uint8_t* st, int kk) {
#pragma unroll
    for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const v4i*>(st + b_off[j][kk]);
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const v4i*>(st + a_off[i][kk]);
  };
  auto cluster = [&](auto dma, auto dma2) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fb[j], fa[i], acc[i][j], 0, 0, 0);
#if FFQ_X == 3  // every SIMD's computing wave issues its pieces at a different point of the cluster: no TA burst
      if (i == wn) { __builtin_amdgcn_sched_barrier(0); dma(); dma2(); __builtin_amdgcn_sched_barrier(0); }
#elif FFQ_X == 4  // no LDS-DMA inside the loop at all (wrong results: cost of the stream)
      (void)dma; (void)dma2;
#else
      if (i == 0) { __builtin_amdgcn_sched_barrier(0); dma(); __builtin_amdgcn_sched_barrier(0); }
      if (i == 2) { __builtin_amdgcn_sched_barrier(0); dma2(); __builtin_amdgcn_sched_barrier(0); }
#endif
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(0);
  };
  // experiment form: a hook after EVERY pair of MFMAs
  auto cluster4 = [&](auto h0, auto h1, auto h2, auto h3) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fb[j], fa[i], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (i == 0) h0();
      if (i == 1) h1();
      if (i == 2) h2();
      if (i == 3) h3();
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_setprio(0);
  };
  (void)cluster4; (void)issue_piece;
#if FFQ_X == 8  // two k-steps per phase: 16 MFMAs per cluster, half as many barriers
  v4i fa2[4], fb2[2];
  auto read_frags2 = [&](const uint8_t* st, int kk) {  // fragment sets kk and kk + 1
#pragma unroll
    for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const v4i*>(st + b_off[j][kk]);
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const v4i*>(st + a_off[i][kk]);
#pragma unroll
    for (int j = 0; j < 2; ++j) fb2[j] = *reinterpret_cast<const v4i*>(st + b_off[j][kk + 1]);
#pragma unroll
    for (int i = 0; i < 4; ++i) fa2[i] = *reinterpret_cast<const v4i*>(st + a_off[i][kk + 1]);
  };
  auto cluster16 = [&](auto piece) {  // piece(p) after every MFMA pair, p = 0..7
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fb[j], fa[i], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      piece(i);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fb2[j], fa2[i], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      piece(4 + i);
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_setprio(0);
  };
#endif

  const int ksuper = a.K / 128;
  int slot = 0;  // slot of the super-step about to be computed
  tile_origin(0, m0, n0);
  set_sources(m0, n0);
  issue_a(0, 0, 0); issue_a(0, 0, 2); issue_b(0, 0, 0); issue_b(0, 0, 2);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  for (int it = 0; it < my_tiles; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0;
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
#if FFQ_X == 8
      {
        const int os8 = slot ^ 1;
        read_frags2(st, 0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        cluster16([&](int p8) { issue_piece(fetch, os8, p8); });
        __builtin_amdgcn_s_barrier();
        read_frags2(st, 2);
        if (wm == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the slower group waits before the barrier that opens the faster group's next reads
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        cluster16([](int) {});
        if (wm == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the faster group: one segment later
        __builtin_amdgcn_s_barrier();
      }
#elif FFQ_X >= 5 && FFQ_X <= 7
      const int os = slot ^ 1;
      auto none = [] {};
      read_frags(st, 0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
#if FFQ_X == 5    // one piece behind every MFMA pair of the first two clusters
      cluster4([&] { issue_piece(fetch, os, 0); }, [&] { issue_piece(fetch, os, 1); }, [&] { issue_piece(fetch, os, 4); }, [&] { issue_piece(fetch, os, 5); });
#elif FFQ_X == 6  // half of the pieces in the clusters (one per other gap), half in the load segments
      cluster4(none, [&] { issue_piece(fetch, os, 0); }, none, [&] { issue_piece(fetch, os, 4); });
#else
      cluster4(none, none, none, none);
#endif
      __builtin_amdgcn_s_barrier();
      read_frags(st, 1);
      __builtin_amdgcn_sched_barrier(0);
#if FFQ_X == 6
      issue_piece(fetch, os, 1); issue_piece(fetch, os, 5);
      __builtin_amdgcn_sched_barrier(0);
#elif FFQ_X == 7  // every piece in a load segment
      issue_piece(fetch, os, 0); issue_piece(fetch, os, 1); issue_piece(fetch, os, 4); issue_piece(fetch, os, 5);
      __builtin_amdgcn_sched_barrier(0);
#endif
      __builtin_amdgcn_s_barrier();
#if FFQ_X == 5
      cluster4([&] { issue_piece(fetch, os, 2); }, [&] { issue_piece(fetch, os, 3); }, [&] { issue_piece(fetch, os, 6); }, [&] { issue_piece(fetch, os, 7); });
#elif FFQ_X == 6
      cluster4(none, [&] { issue_piece(fetch, os, 2); }, none, [&] { issue_piece(fetch, os, 6); });
#else
      cluster4(none, none, none, none);
#endif
      __builtin_amdgcn_s_barrier();
      read_frags(st, 2);
      __builtin_amdgcn_sched_barrier(0);
#if FFQ_X == 6
      issue_piece(fetch, os, 3); issue_piece(fetch, os, 7);
      __builtin_amdgcn_sched_barrier(0);
#elif FFQ_X == 7
      issue_piece(fetch, os, 2); issue_piece(fetch, os, 3); issue_piece(fetch, os, 6); issue_piece(fetch, os, 7);
      __builtin_amdgcn_sched_barrier(0);
#endif
      __builtin_amdgcn_s_barrier();
      cluster4(none, none, none, none);
      __builtin_amdgcn_s_barrier();
      read_frags(st, 3);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      cluster4(none, none, none, none);
      __builtin_amdgcn_s_barrier();
#else
      read_frags(st, 0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
#if FFQ_X == 2  // all eight pieces in the first cluster (one interval more for the second half to land)
      cluster([&] { issue_a(fetch, slot ^ 1, 0); issue_a(fetch, slot ^ 1, 2); }, [&] { issue_b(fetch, slot ^ 1, 0); issue_b(fetch, slot ^ 1, 2); });
#else
      cluster([&] { issue_a(fetch, slot ^ 1, 0); }, [&] { issue_b(fetch, slot ^ 1, 0); });
#endif
      __builtin_amdgcn_s_barrier();
      read_frags(st, 1);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
#if FFQ_X == 2
      cluster([] {}, [] {});
#else
      cluster([&] { issue_a(fetch, slot ^ 1, 2); }, [&] { issue_b(fetch, slot ^ 1, 2); });
#endif
      __builtin_amdgcn_s_barrier();
      read_frags(st, 2);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      cluster([] {}, [] {});
      __builtin_amdgcn_s_barrier();
      read_frags(st, 3);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the fetched super-step landed (and older epilogue stores)
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      cluster([] {}, [] {});
      __builtin_amdgcn_s_barrier();
#endif
      slot ^= 1;
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();  // same number of barriers for both groups
    // `slot` now names the slot holding the prefetched super-step; the other one has been consumed: epilogue scratch
    uint8_t* scratch = lds2 + (slot ^ 1) * SLOT_BYTES;
    __syncthreads();
    if constexpr (MLP) {
      int rsw[2] = {0, 0};
      if (a.rowsum_w) {
        rsw[0] = a.rowsum_w[n0 + wn * 32 + (lane & 31)];
        rsw[1] = a.rowsum_w2[n0 + wn * 32 + (lane & 31)];
      }
      mlp_epilogue<true>(a, acc, rsw, scratch, wave, lane, wm, wn, m0, n0, silu_table);
    } else {
      gemm256_epilogue_slabs<TOut, REQUANT>(a, acc, scratch, wave, lane, wm, wn, m0, n0);
    }
    __syncthreads();  // the scratch slot is the next tile's DMA target
    m0 = nm0; n0 = nn0;
  }
}

// -------------------------------------------------------------------------------------------------
// "fq": the persistent kernel above on v_mfma_i32_16x16x64_i8. Same 256 x 256 x 128 super-steps, same LDS image, swizzle,
// LDS-DMA stream, ping-pong phases, barriers and waits; the matrix instruction differs. Two 16x16x64 do the work of one
// 32x32x32 from the same operand registers in the same 32 cycles, and the chip runs them faster: a timing-only build of the
// kernel above with its MFMAs swapped one for two measured +5 % on every shape (power: the chip is power-limited on this
// loop; `profiles/r01_mfma_power_probe.txt` had the bare instruction at +2-3 %).
//   * wave tile 128 x 64 = 8 x 4 accumulator tiles of 16 x 16 (4 registers each, 128 in all, as before);
//   * fragments: lane (r = lane % 16, g = lane / 16) holds bytes [16 g, 16 g + 16) of row r of a 64-byte k-chunk — one
//     ds_read_b128 at slot (4 kq + g) ^ swizzle(row); the existing swizzle keeps those reads conflict-free;
//   * a phase = one k-chunk (kq = phase / 2) x one half of the rows (mi in [4 (phase % 2), +4)): even phases read the 4
//     weight fragments of the chunk and 4 activation fragments, odd phases the other 4 activation fragments; 16 MFMAs each;
//   * accumulator layout (operands swapped as before, so a lane owns ONE output row): tile (mi, nj), register t:
//     row m = 16 mi + lane % 16, column n = 16 nj + 4 (lane / 16) + t.
// -------------------------------------------------------------------------------------------------
#ifndef FFQ_Y
#define FFQ_Y 3  // 3 = LDS-DMA as inline assembly in the saddr form (default); 0 = the builtin
#endif
typedef int v4i32 __attribute__((ext_vector_type(4)));

template <typename T> __device__ __forceinline__ float half_bits_to_f32(uint32_t bits);  // 16-bit pattern -> value
template <> __device__ __forceinline__ float half_bits_to_f32<bf16_t>(uint32_t bits) { return __builtin_bit_cast(float, bits << 16); }
template <> __device__ __forceinline__ float half_bits_to_f32<f16_t>(uint32_t bits) { return (float)__builtin_bit_cast(_Float16, (uint16_t)bits); }

// Epilogue of the plain mode for that layout: the slab scheme of gemm256_epilogue_slabs (32 rows x 64 columns per wave and
// round, whole 128-byte lines out with the non-temporal hint), two row tiles per slab.
template <typename TOut, bool REQUANT, bool RESID>
__device__ __forceinline__ void gemm256_epilogue_slabs16(const LinearArgs& a, v4i32 (&acc)[8][4], uint8_t* scratch, int wave, int lane,
                                                         int wm, int wn, int m0, int n0) {
  TOut* out = static_cast<TOut*>(a.out);
  float oscale = 1.0f, ooff = 0.0f;
  if constexpr (REQUANT) {
    oscale = a.out_scale[0];
    ooff = a.out_offset ? rne(a.out_offset[0]) : 0.0f;
  }
  constexpr int ROW_BYTES = 144;
  constexpr int WAVE_BYTES = 32 * ROW_BYTES + 3 * 64 * 4;
  uint8_t* region = scratch + wave * WAVE_BYTES;
  float* colp = reinterpret_cast<float*>(region + 32 * ROW_BYTES);  // [3][64]: weight scale, weight row sum, bias
  const int r16 = lane & 15, g4 = lane >> 4;
  const int wave_n0 = n0 + wn * 64;
  const int wave_m0 = m0 + wm * 128;
  const bool lds_path = sizeof(TOut) == 2 && (a.N & 7) == 0 && wave_n0 + 64 <= a.N;
  {
    int n = wave_n0 + lane;
    n = n < a.N ? n : a.N - 1;
    colp[lane] = a.w_scale[a.w_per_row ? n : 0];
    colp[64 + lane] = a.rowsum_w ? (float)a.rowsum_w[n] : 0.0f;
    colp[128 + lane] = a.bias ? (float)load_any(a.bias, a.bias_dt, n) : 0.0f;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private: only this wave's own writes
  // every global load of the epilogue BEFORE its first store (see gemm256_epilogue_slabs)
  float sx8[8], ox8[8];
#pragma unroll
  for (int mi = 0; mi < 8; ++mi) {
    int m = wave_m0 + mi * 16 + r16;
    m = m < a.M ? m : a.M - 1;
    sx8[mi] = a.x_scale[a.x_per_row ? m : 0];
    ox8[mi] = a.x_offset ? rne(a.x_offset[a.x_per_row ? m : 0]) : 0.0f;
  }
#pragma unroll
  for (int mi = 0; mi < 8; ++mi) asm volatile("" : "+v"(sx8[mi]), "+v"(ox8[mi]));
  // residual rows of the slab about to leave, requested one slab ahead and BEFORE the previous slab's stores go out: the
  // wait for them is a counted vmcnt that leaves those (younger) stores in flight
  [[maybe_unused]] u32x4 res_cur[4], res_nxt[4];
  [[maybe_unused]] auto load_residual = [&](int i, u32x4 (&dst)[4]) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int c = lane + 64 * t;
      const int row = c >> 3, seg = c & 7;
      int mm = wave_m0 + i * 32 + row;
      mm = mm < a.M ? mm : a.M - 1;
      dst[t] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(static_cast<const uint8_t*>(a.residual) + ((size_t)mm * a.N + wave_n0) * 2 + seg * 16));
    }
  };
  constexpr bool with_residual = RESID && sizeof(TOut) == 2 && !REQUANT;  // the launcher admits only shapes whose waves all take the LDS path
  if constexpr (with_residual) {
    if (lds_path) load_residual(0, res_cur);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int mi = 2 * i + hh;
      const int m = wave_m0 + mi * 16 + r16;
      const bool m_ok = m < a.M;
      const float sx = sx8[mi], ox = ox8[mi];
#pragma unroll
      for (int nj = 0; nj < 4; ++nj) {
        const int nb = nj * 16 + 4 * g4;
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        const f32x4 sw4 = *reinterpret_cast<const f32x4*>(colp + nb);
        const f32x4 rs4 = *reinterpret_cast<const f32x4*>(colp + 64 + nb);
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(colp + 128 + nb);
        float y[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float v = (float)acc[mi][nj][t] + ox * rs4[t];
          float r = (sx * sw4[t]) * v;
          if (a.bias) r = r + b4[t];
          if constexpr (REQUANT) {
            r = bf16_bits_to_f32(f32_to_bf16_bits(r));
            r = clamp_nan(rne(r / oscale - ooff), a.out_lo, a.out_hi);
          }
          y[t] = r;
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
          const size_t at = (size_t)m * a.N + wave_n0 + nb;
#pragma unroll
          for (int t = 0; t < 4; ++t)
            if (wave_n0 + nb + t < a.N) store_out<TOut>(out + at + t, y[t]);
        }
      }
    }
    if (lds_path) {
      if constexpr (with_residual) {
        if (i < 3) load_residual(i + 1, res_nxt);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int c = lane + 64 * t;
        const int row = c >> 3, seg = c & 7;
        const int mm = wave_m0 + i * 32 + row;
        u32x4 v = *reinterpret_cast<const u32x4*>(region + row * ROW_BYTES + seg * 16);
        if constexpr (with_residual) {
          {  // residual + T(linear) in the output dtype: the linear's rounding, then the add's (decoder.py:60-90)
            const u32x4 r = res_cur[t];
            const uint32_t vw[4] = {v.x, v.y, v.z, v.w}, rw[4] = {r.x, r.y, r.z, r.w};
            uint32_t ow[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float s0 = half_bits_to_f32<TOut>(vw[e] & 0xFFFFu) + half_bits_to_f32<TOut>(rw[e] & 0xFFFFu);
              const float s1 = half_bits_to_f32<TOut>(vw[e] >> 16) + half_bits_to_f32<TOut>(rw[e] >> 16);
              ow[e] = pack2<TOut>(s0, s1);
            }
            v.x = ow[0]; v.y = ow[1]; v.z = ow[2]; v.w = ow[3];
          }
        }
        // non-temporal both ways (the residual read above, the sum written here): 134 MB streaming through the L2s would
        // push the operand panels out (ordinary accesses: +28 us on o_proj, +66 us on down_proj at T = 16384)
        if (mm < a.M && !(a.debug & 1))
          FFQ_EPI_STORE(v, reinterpret_cast<u32x4*>(reinterpret_cast<uint8_t*>(out) + ((size_t)mm * a.N + wave_n0) * 2 + seg * 16));
      }
      if constexpr (with_residual) {
#pragma unroll
        for (int t = 0; t < 4; ++t) res_cur[t] = res_nxt[t];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the slab is re-written by the next i
    }
  }
}

// Epilogue of the MLP mode for that layout (see mlp_epilogue_body): column tiles nj = 0, 1 hold gate_proj and nj + 2 up_proj
// of the same 16 output columns; silu through the LDS table of ffq_silu.h.
template <bool SAFE>
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
  const int lo = (int)a.out_lo, hi = (int)a.out_hi;
  // Every VALU instruction of this epilogue is paid in full — it does not hide under another wave's MFMAs; the ablation
  // -DFFQ_X=13 (no division, no window test: 10 instructions per element) makes the launch 2.9 % faster — so:
  //  * the quotient is Divider::fast with its window test as ONE v_cmp_class: q2 for every normal q0, q0 itself for
  //    zero / denormal / Inf / NaN (normal quotients outside 2^-40 .. 2^40 round to -o or leave through the clamp whichever
  //    candidate is taken; zero keeps its sign, an overflowed quotient stays Inf instead of the NaN of its residual);
  //  * the clamp is one v_med3_i32 (lo <= hi).
  auto quotient = [&](float x) {
    const float q0 = x * div.r;
    const float q1 = __builtin_fmaf(__builtin_fmaf(-q0, div.s, x), div.r, q0);
    const float q2 = __builtin_fmaf(__builtin_fmaf(-q1, div.s, x), div.r, q1);
    return __builtin_amdgcn_class(q0, 0x108) ? q2 : q0;
  };
  auto clamp_code = [&](int v) {
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(v), "v"(lo), "v"(hi));
    return r;
  };
#pragma unroll
  for (int nj = 0; nj < 2; ++nj) {
    const int cb = 16 * nj + 4 * g4;  // this lane's 4 columns: col0 + cb + (0..3)
    float swg[4], swu[4], rsg[4], rsu[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int n = col0 + cb + t;  // N % 128 == 0: always inside
      swg[t] = a.w_scale[n];
      swu[t] = a.w_scale2[n];
      rsg[t] = rs_lds[cb + t];
      rsu[t] = rs_lds[32 + cb + t];
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
          const float g0 = (sx * swg[t]) * ((float)acc[mi][nj][t] + ox * rsg[t]);
          const float g1 = (sx * swg[t + 1]) * ((float)acc[mi][nj][t + 1] + ox * rsg[t + 1]);
          wg[q][t >> 1] = pack2<bf16_t>(g0, g1);
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
        int c[4];
#pragma unroll
        for (int t = 0; t < 4; t += 2) {
          uint32_t w = pack2<bf16_t>((sx * swu[t]) * ((float)acc[mi][nj + 2][t] + ox * rsu[t]),
                                     (sx * swu[t + 1]) * ((float)acc[mi][nj + 2][t + 1] + ox * rsu[t + 1]));
          const float u0 = __builtin_bit_cast(float, w << 16), u1 = __builtin_bit_cast(float, w & 0xFFFF0000u);
          w = ws[q][t >> 1];
          const float a0 = __builtin_bit_cast(float, w << 16), a1 = __builtin_bit_cast(float, w & 0xFFFF0000u);
          float z0 = a0 * u0, z1 = a1 * u1;
          w = pack2<bf16_t>(z0, z1);
          z0 = __builtin_bit_cast(float, w << 16); z1 = __builtin_bit_cast(float, w & 0xFFFF0000u);
#if FFQ_X == 13  // ablation (wrong codes): what the A1 division + window test cost in this launch
          const float r0 = z0 - oo, r1 = z1 - oo;
#else
          const float r0 = SAFE ? rne(quotient(z0) - oo) : rne(z0 / so - oo);
          const float r1 = SAFE ? rne(quotient(z1) - oo) : rne(z1 / so - oo);
#endif
          const int c0 = (int)r0, c1 = (int)r1;  // v_cvt_i32_f32: NaN -> 0, the int8 container's value
          c[t] = clamp_code(c0);
          c[t + 1] = clamp_code(c1);
        }
        const int row = wm * 128 + mi * 16 + r16;
        *reinterpret_cast<uint32_t*>(lds2 + row * PITCH + wn * 32 + cb) = pack_bytes(c[0], c[1], c[2], c[3]);
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
    if (m < a.M && !(a.debug & 1)) FFQ_EPI_STORE(v, reinterpret_cast<u32x4*>(out + (size_t)m * a.N + n0 + seg * 16));
  }
}

template <typename TOut, bool REQUANT, bool MLP, bool RESID = false>
__global__ __launch_bounds__(512, 2) void w8a8_gemm256fq_kernel(LinearArgs a, int total_tiles) {
  constexpr int BN2 = 256, WAVES_N = 4;
  constexpr int BN_OUT = MLP ? 128 : 256;
  constexpr int SLOT_BYTES = (BM2 + BN2) * 128;
  constexpr int B_IMAGE = BM2 * 128;
  extern __shared__ __attribute__((aligned(16))) uint8_t lds2[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  // tile walk of the fp kernel: XCD x owns a contiguous range of the grouped tile order, its blocks walk it round-robin
  const uint32_t nblk = gridDim.x;
  const uint32_t xcd = blockIdx.x & 7u, j_in_xcd = blockIdx.x >> 3;
  const uint32_t blocks_in_xcd = (nblk >> 3) + (xcd < (nblk & 7u) ? 1u : 0u);
  const uint32_t tq = (uint32_t)total_tiles >> 3, tr = (uint32_t)total_tiles & 7u;
  const uint32_t xcd_first = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
  const uint32_t xcd_count = tq + (xcd < tr ? 1u : 0u);
  const int my_tiles = j_in_xcd < xcd_count ? (int)((xcd_count - j_in_xcd + blocks_in_xcd - 1) / blocks_in_xcd) : 0;
  if (my_tiles == 0) return;
  uint16_t* const silu_table = reinterpret_cast<uint16_t*>(lds2 + 2 * SLOT_BYTES);
  if constexpr (MLP) silu_table_fill(silu_table, (uint32_t)tid, 512u);

  const int d_row = lane >> 3;
  // DMA sources as (wave-uniform base pointer + 32-bit lane offset): the k offset of a super-step goes into the uniform
  // part, so the loop spends SALU, not VALU, on addresses (global_load_lds saddr form; VALU is not hidden under MFMAs)
  uint32_t a_voff[4], b_voff[4];
  const int8_t* b_base[4];
  int m0 = 0, n0 = 0;
  auto tile_origin = [&](int it, int& tm0, int& tn0) {
    const uint32_t tile_id = xcd_first + j_in_xcd + (uint32_t)it * blocks_in_xcd;
    const uint32_t gm = (uint32_t)a.group_m;
    const uint32_t per_group = gm * (uint32_t)a.tiles_n;
    const uint32_t group = tile_id / per_group, in_group = tile_id - group * per_group;
    const uint32_t group_rows = min(gm, (uint32_t)a.tiles_m - group * gm);
    tm0 = (int)(group * gm + in_group % group_rows) * BM2;
    tn0 = (int)(in_group / group_rows) * BN_OUT;
  };
  auto set_sources = [&](int tm0, int tn0) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int row = (wave * 4 + c) * 8 + d_row;
      const int d_slot = (lane & 7) ^ ((row >> 1) & 7);
      int ra = tm0 + row;
      ra = ra < a.M ? ra : a.M - 1;
      a_voff[c] = (uint32_t)ra * (uint32_t)a.K + d_slot * 16;   // M * K < 2^32 (checked by the launcher)
      if constexpr (MLP) {
        const int rb = tn0 + (row >> 6) * 32 + (row & 31);
        // rows 32..63 of a wave's 64 come from the up matrix: row & 32 is the same for all lanes of a piece (8 rows per piece)
        b_base[c] = (((wave * 4 + c) * 8) & 32) ? a.wq2 : a.wq;
        b_voff[c] = (uint32_t)rb * (uint32_t)a.K + d_slot * 16;
      } else {
        int rb = tn0 + row;
        rb = rb < a.N ? rb : a.N - 1;
        b_base[c] = a.wq;
        b_voff[c] = (uint32_t)rb * (uint32_t)a.K + d_slot * 16;
      }
    }
  };
  auto issue_a = [&](int ks, int slot, int c0) {
    uint8_t* base = lds2 + slot * SLOT_BYTES;
#pragma unroll
    for (int c = c0; c < c0 + 2; ++c) {
#if FFQ_Y == 3  // the saddr form spelled out (the builtin gets it in one of the loop's two unrolled bodies only)
      const int8_t* ub = a.xq + ks * 128;
      const uint32_t lds_addr = (uint32_t)(uintptr_t)(base + (wave * 4 + c) * 1024);
      asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(a_voff[c]), "s"(ub), "s"(lds_addr) : "memory", "m0");
#else
      __builtin_amdgcn_global_load_lds((gbl_void_t*)((a.xq + ks * 128) + a_voff[c]), (lds_void_t*)(base + (wave * 4 + c) * 1024), 16, 0, 0);
#endif
    }
  };
  auto issue_b = [&](int ks, int slot, int c0) {
    uint8_t* base = lds2 + slot * SLOT_BYTES + B_IMAGE;
#pragma unroll
    for (int c = c0; c < c0 + 2; ++c) {
#if FFQ_Y == 3
      const int8_t* ub = b_base[c] + ks * 128;
      const uint32_t lds_addr = (uint32_t)(uintptr_t)(base + (wave * 4 + c) * 1024);
      asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(b_voff[c]), "s"(ub), "s"(lds_addr) : "memory", "m0");
#else
      __builtin_amdgcn_global_load_lds((gbl_void_t*)((b_base[c] + ks * 128) + b_voff[c]), (lds_void_t*)(base + (wave * 4 + c) * 1024), 16, 0, 0);
#endif
    }
  };

  // fragment byte offsets inside a slot: [row tile][k-chunk]
  const uint32_t r16 = lane & 15, g4 = lane >> 4;
  uint32_t a_off[8][2], b_off[4][2];
#pragma unroll
  for (int mi = 0; mi < 8; ++mi) {
    const uint32_t row = wm * 128 + mi * 16 + r16;
#pragma unroll
    for (int kq = 0; kq < 2; ++kq) a_off[mi][kq] = row * 128 + (((kq * 4 + g4) ^ ((row >> 1) & 7u)) << 4);
  }
#pragma unroll
  for (int nj = 0; nj < 4; ++nj) {
    const uint32_t row = wn * 64 + nj * 16 + r16;
#pragma unroll
    for (int kq = 0; kq < 2; ++kq) b_off[nj][kq] = B_IMAGE + row * 128 + (((kq * 4 + g4) ^ ((row >> 1) & 7u)) << 4);
  }

  v4i32 acc[8][4];
  v4i fa[4], fb[4];
  auto read_frags = [&](const uint8_t* st, int phase) {  // phase 0..3 of a super-step (compile-time after unrolling)
    const int kq = phase >> 1, mh = phase & 1;
    if (mh == 0) {
#pragma unroll
      for (int nj = 0; nj < 4; ++nj) fb[nj] = *reinterpret_cast<const v4i*>(st + b_off[nj][kq]);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) fa[q] = *reinterpret_cast<const v4i*>(st + a_off[4 * mh + q][kq]);
  };
  auto cluster = [&](int mh, auto dma, auto dma2) {
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
      if (q == 0) { __builtin_amdgcn_sched_barrier(0); dma(); __builtin_amdgcn_sched_barrier(0); }
      if (q == 2) { __builtin_amdgcn_sched_barrier(0); dma2(); __builtin_amdgcn_sched_barrier(0); }
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(0);
  };

  const int ksuper = a.K / 128;
  int slot = 0;  // slot of the super-step about to be computed
  tile_origin(0, m0, n0);
  set_sources(m0, n0);
  issue_a(0, 0, 0); issue_a(0, 0, 2); issue_b(0, 0, 0); issue_b(0, 0, 2);
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
      int fetch = ks + 1;
      if (ks == ksuper - 1) {
        fetch = has_next ? 0 : ks;
        if (has_next) set_sources(nm0, nn0);
      }
      read_frags(st, 0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      cluster(0, [&] { issue_a(fetch, slot ^ 1, 0); }, [&] { issue_b(fetch, slot ^ 1, 0); });
      __builtin_amdgcn_s_barrier();
      read_frags(st, 1);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      cluster(1, [&] { issue_a(fetch, slot ^ 1, 2); }, [&] { issue_b(fetch, slot ^ 1, 2); });
      __builtin_amdgcn_s_barrier();
      read_frags(st, 2);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      cluster(0, [] {}, [] {});
      __builtin_amdgcn_s_barrier();
      read_frags(st, 3);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the fetched super-step landed (and older epilogue stores)
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      cluster(1, [] {}, [] {});
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
      const float as = __builtin_fabsf(a.out_scale[0]);
      if (as > 0x1p-40f && as < 0x1p40f) mlp_epilogue16_body<true>(a, acc, rsw, scratch, wave, lane, wm, wn, m0, n0, silu_table);
      else mlp_epilogue16_body<false>(a, acc, rsw, scratch, wave, lane, wm, wn, m0, n0, silu_table);
    } else {
      gemm256_epilogue_slabs16<TOut, REQUANT, RESID>(a, acc, scratch, wave, lane, wm, wn, m0, n0);
    }
    __syncthreads();  // the scratch slot is the next tile's DMA target
    m0 = nm0; n0 = nn0;
  }
}

// one wavefront per row: sum of K int8 codes
__global__ __launch_bounds__(256) void rowsum_i8_kernel(const int8_t* __restrict__ q, int rows, int K,
                                                        int32_t* __restrict__ sums) {
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

}  // namespace ffq

namespace ffq {
// ffq_linear4w.hip: the one-wave-per-SIMD form of the plain GEMM (experiment, FFQ_GEMM_4W=1)
int linear4w_try(const int8_t* xq, const int8_t* wq, const int32_t* w_rowsum, const float* x_scale, const float* x_offset, const float* w_scale,
                 int w_per_row, void* out, int64_t M, int64_t N, int64_t K, int32_t* workspace, hipStream_t s);
}  // namespace ffq

using namespace ffq;

extern "C" size_t ffq_linear_w8a8_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  (void)K;
  if (M < 0 || N < 0) return 0;
  return (size_t)(((M + N) * 4 + 255) & ~(int64_t)255);
}

extern "C" int ffq_linear_w8a8(const int8_t* xq, const int8_t* wq, const float* x_scale, const float* x_offset,
                               int x_per_row, const float* w_scale, const float* w_offset, int w_per_row,
                               const void* bias, int bias_dt, void* out, int out_dt, const float* out_scale,
                               const float* out_offset, double out_num_bits, int64_t M, int64_t N, int64_t K,
                               void* workspace, size_t workspace_bytes, void* stream) {
  return ffq_linear_w8a8_rs(xq, wq, nullptr, x_scale, x_offset, x_per_row, w_scale, w_offset, w_per_row, bias, bias_dt, out, out_dt,
                            out_scale, out_offset, out_num_bits, M, N, K, workspace, workspace_bytes, stream);
}

static int linear_w8a8_launch(const int8_t* xq, const int8_t* wq, const int32_t* w_rowsum, const float* x_scale,
                              const float* x_offset, int x_per_row, const float* w_scale, const float* w_offset,
                              int w_per_row, const void* bias, int bias_dt, const void* residual, void* out, int out_dt,
                              const float* out_scale, const float* out_offset, double out_num_bits, int64_t M,
                              int64_t N, int64_t K, void* workspace, size_t workspace_bytes, void* stream);

// w_rowsum (nullable): sum_k wq[n, k] already known to the caller (ffq_quantize_rows_rowsum) — no reduction launch here
extern "C" int ffq_linear_w8a8_rs(const int8_t* xq, const int8_t* wq, const int32_t* w_rowsum, const float* x_scale,
                                  const float* x_offset, int x_per_row, const float* w_scale, const float* w_offset,
                                  int w_per_row, const void* bias, int bias_dt, void* out, int out_dt,
                                  const float* out_scale, const float* out_offset, double out_num_bits, int64_t M,
                                  int64_t N, int64_t K, void* workspace, size_t workspace_bytes, void* stream) {
  return linear_w8a8_launch(xq, wq, w_rowsum, x_scale, x_offset, x_per_row, w_scale, w_offset, w_per_row, bias, bias_dt, nullptr, out, out_dt,
                            out_scale, out_offset, out_num_bits, M, N, K, workspace, workspace_bytes, stream);
}

// out = residual + T(linear): the residual add behind o_proj / down_proj (decoder.py:60-90) in the GEMM's epilogue. Covered
// where the persistent kernel runs with whole-line stores; FFQ_ERR_DTYPE elsewhere (the caller adds with a separate pass).
extern "C" int ffq_linear_w8a8_residual(const int8_t* xq, const int8_t* wq, const int32_t* w_rowsum, const float* x_scale,
                                        const float* x_offset, const float* w_scale, const void* residual, void* out, int out_dt,
                                        int64_t M, int64_t N, int64_t K, void* workspace, size_t workspace_bytes, void* stream) {
  if (!residual) return fail(FFQ_ERR_ARG, "NULL residual");
  if (!(out_dt == FFQ_BF16 || out_dt == FFQ_F16)) return fail(FFQ_ERR_DTYPE, "the fused residual add is built for bf16 / fp16 outputs");
  if (!aligned16(residual) || !aligned16(out)) return fail(FFQ_ERR_ARG, "buffers must be 16-byte aligned");
  return linear_w8a8_launch(xq, wq, w_rowsum, x_scale, x_offset, 0, w_scale, nullptr, 1, nullptr, 0, residual, out, out_dt,
                            nullptr, nullptr, 8.0, M, N, K, workspace, workspace_bytes, stream);
}

static int linear_w8a8_launch(const int8_t* xq, const int8_t* wq, const int32_t* w_rowsum, const float* x_scale,
                              const float* x_offset, int x_per_row, const float* w_scale, const float* w_offset,
                              int w_per_row, const void* bias, int bias_dt, const void* residual, void* out, int out_dt,
                              const float* out_scale, const float* out_offset, double out_num_bits, int64_t M,
                              int64_t N, int64_t K, void* workspace, size_t workspace_bytes, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (M < 0 || N < 0 || K < 0) return fail(FFQ_ERR_ARG, "negative extent");
  if (M == 0 || N == 0) return FFQ_OK;
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
  } else if (!(out_dt == FFQ_F32 || out_dt == FFQ_BF16 || out_dt == FFQ_F16)) {
    return fail(FFQ_ERR_DTYPE, "real-valued output must be f32, bf16 or f16");
  }
  const size_t need = ffq_linear_w8a8_workspace_bytes(M, N, K);
  if (((x_offset && !w_rowsum) || w_offset) && (need > workspace_bytes || !workspace))
    return fail(FFQ_ERR_WORKSPACE, "w8a8 linear needs %zu workspace bytes, got %zu", need, workspace_bytes);

  LinearArgs a;
  a.xq = xq; a.wq = wq;
  a.x_scale = x_scale; a.x_offset = x_offset;
  a.w_scale = w_scale; a.w_offset = w_offset;
  a.rowsum_x = nullptr; a.rowsum_w = nullptr;
  a.wq2 = nullptr; a.w_scale2 = nullptr; a.rowsum_w2 = nullptr;
  a.bias = bias; a.bias_dt = bias_dt;
  a.residual = residual;
  a.out = out; a.out_dt = out_dt;
  a.out_scale = out_scale; a.out_offset = out_offset;
  const double lo = -pow(2.0, out_num_bits - 1.0);
  a.out_lo = (float)lo; a.out_hi = (float)(-lo - 1.0);
  a.x_per_row = x_per_row; a.w_per_row = w_per_row;
  a.M = (int)M; a.N = (int)N; a.K = (int)K;
  a.tiles_m = (int)((M + BM - 1) / BM);
  a.tiles_n = (int)((N + BN - 1) / BN);
  static const int debug_bits = getenv("FFQ_GEMM_DEBUG") ? atoi(getenv("FFQ_GEMM_DEBUG")) : 0;
  a.debug = debug_bits;

  int32_t* ws = static_cast<int32_t*>(workspace);
  if (residual) {  // only the persistent 16x16x64 kernel's whole-line epilogue adds it: the same conditions as its dispatch below
    static const int fq_on = (getenv("FFQ_GEMM_FQ") ? atoi(getenv("FFQ_GEMM_FQ")) : 1) && (getenv("FFQ_GEMM_FP") ? atoi(getenv("FFQ_GEMM_FP")) : 1) &&
                             (getenv("FFQ_GEMM_FL") ? atoi(getenv("FFQ_GEMM_FL")) : 1) && !getenv("FFQ_GEMM_V1") && !getenv("FFQ_GEMM_V2") &&
                             !getenv("FFQ_GEMM_NW") && !getenv("FFQ_GEMM_4W");
    const int64_t tiles = ((M + BM2 - 1) / BM2) * ((N + 255) / 256);
    if (!fq_on || K % 128 != 0 || K < 256 || M < 128 || N < 128 || N % 64 != 0 || tiles < 64 ||
        (uint64_t)M * (uint64_t)K >= (1ull << 32) || (uint64_t)N * (uint64_t)K >= (1ull << 32))
      return fail(FFQ_ERR_DTYPE, "fused residual add: shape outside the persistent kernel (K %% 128, N %% 64, >= 64 tiles)");
  }
  static const int use_4w = getenv("FFQ_GEMM_4W") ? atoi(getenv("FFQ_GEMM_4W")) : 0;
  if (use_4w && !requant && out_dt == FFQ_BF16 && !w_offset && !bias && !x_per_row &&
      linear4w_try(xq, wq, w_rowsum, x_scale, x_offset, w_scale, w_per_row, out, M, N, K, ws, s) == 0)
    return check_launch("w8a8_gemm4w_kernel");
  // the direct-to-LDS kernels compute the weight row sums themselves; they need K % 64 == 0
  static const int force_v1 = getenv("FFQ_GEMM_V1") ? 1 : 0;
  static const int force_nw = getenv("FFQ_GEMM_NW") ? atoi(getenv("FFQ_GEMM_NW")) : 0;
  const int64_t tiles256 = ((M + BM2 - 1) / BM2) * ((N + 255) / 256);
  const bool use_v2 = !force_v1 && K % BK2 == 0 && M >= 128 && N >= 128 && tiles256 >= 64;
  if (use_v2) {
    static const int force_v2 = getenv("FFQ_GEMM_V2") ? 1 : 0;
    if (!force_v2 && !force_nw && K / BK2 >= 4) {
      a.tiles_m = (int)((M + BM2 - 1) / BM2);
      a.tiles_n = (int)((N + 255) / 256);
      // 8 row tiles per group; 4 for long contractions (down_proj, K = 14336: +3 %, at the vendor kernel's 2.6 POP/s; A/B of
      // 2 / 3 / 4 / 6 / 8 / 16 / 32 on one box) — the group's A panels are 256 x K bytes each
      a.group_m = K >= 8192 ? 4 : GROUP_M2;
      static const int force_gm = getenv("FFQ_GROUP_M") ? atoi(getenv("FFQ_GROUP_M")) : 0;
      if (force_gm > 0) a.group_m = force_gm;
      if (x_offset) {  // sum_k wq[n, k] for the zero-point term: one pass over the weight codes
        if (w_rowsum) {
          a.rowsum_w = w_rowsum;
        } else {
          rowsum_i8_kernel<<<(unsigned)((N + 3) / 4), 256, 0, s>>>(wq, (int)N, (int)K, ws + M);
          a.rowsum_w = ws + M;
        }
      }
      const unsigned grid3 = (unsigned)(a.tiles_m * a.tiles_n);
      const size_t ring3 = (size_t)STAGES3 * (BM2 + 256) * BK2;
      const size_t epi3 = (size_t)8 * (128 * 144 + 256);
      const size_t lds3 = ring3 > epi3 ? ring3 : epi3;
      static const int use_fl = getenv("FFQ_GEMM_FL") ? atoi(getenv("FFQ_GEMM_FL")) : 1;  // full-line staging: +2.4 % (A/B on one box)
      const bool fl = use_fl && K % 128 == 0;
      static const int use_fp = getenv("FFQ_GEMM_FP") ? atoi(getenv("FFQ_GEMM_FP")) : 1;  // persistent tile loop: +4.9 % (A/B on one box)
      const bool fp = fl && use_fp && !w_offset;
      static const int fq_env = getenv("FFQ_GEMM_FQ") ? atoi(getenv("FFQ_GEMM_FQ")) : 1;  // v_mfma_i32_16x16x64_i8 form of the persistent kernel
      // (its DMA addresses are a uniform base + a 32-bit lane offset: operands below 4 GiB each, else the 32x32x32 form)
      const bool use_fq = fq_env && (uint64_t)M * (uint64_t)K < (1ull << 32) && (uint64_t)N * (uint64_t)K < (1ull << 32);
      const unsigned grid_fp = grid3 < 256u ? grid3 : 256u;  // persistent: one block per CU
      const size_t lds_fp = (size_t)2 * (BM2 + 256) * 128;
#define FFQ_GEMM3_FP(T, RQ)                                                                                \
  do {                                                                                                     \
    static uint64_t attr_set_fp = 0;                                                                       \
    if (first_use_on_this_device(&attr_set_fp)) {                                                                                    \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&w8a8_gemm256fp_kernel<T, RQ, false>),       \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_fp);                  \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&w8a8_gemm256fq_kernel<T, RQ, false>),       \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_fp);                  \
    }                                                                                                      \
    if (use_fq) w8a8_gemm256fq_kernel<T, RQ, false><<<grid_fp, 512, lds_fp, s>>>(a, (int)grid3);           \
    else w8a8_gemm256fp_kernel<T, RQ, false><<<grid_fp, 512, lds_fp, s>>>(a, (int)grid3);                  \
  } while (0)
#define FFQ_GEMM3_W(T, RQ, WO)                                                                             \
  do {                                                                                                     \
    if (fp) { FFQ_GEMM3_FP(T, RQ); break; }                                                                \
    static uint64_t attr_set = 0;                                                                          \
    if (first_use_on_this_device(&attr_set)) {                                                                                       \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&w8a8_gemm256pp_kernel<T, RQ, WO>),          \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);                    \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&w8a8_gemm256fl_kernel<T, RQ, WO>),          \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);                    \
    }                                                                                                      \
    if (fl) w8a8_gemm256fl_kernel<T, RQ, WO><<<grid3, 512, lds3, s>>>(a);                                  \
    else w8a8_gemm256pp_kernel<T, RQ, WO><<<grid3, 512, lds3, s>>>(a);                                     \
  } while (0)
#define FFQ_GEMM3(T, RQ) do { if (w_offset) FFQ_GEMM3_W(T, RQ, true); else FFQ_GEMM3_W(T, RQ, false); } while (0)
#define FFQ_GEMM3_RESID(T)                                                                                 \
  do {                                                                                                     \
    static uint64_t attr_set_res = 0;                                                                      \
    if (first_use_on_this_device(&attr_set_res)) {                                                         \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&w8a8_gemm256fq_kernel<T, false, false, true>), \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_fp);                  \
    }                                                                                                      \
    w8a8_gemm256fq_kernel<T, false, false, true><<<grid_fp, 512, lds_fp, s>>>(a, (int)grid3);              \
  } while (0)
      if (residual) {  // admitted above only where fp && use_fq hold
        if (out_dt == FFQ_BF16) FFQ_GEMM3_RESID(bf16_t); else FFQ_GEMM3_RESID(f16_t);
        return check_launch("w8a8_gemm256fq_kernel (residual)");
      }
      if (requant) {
        switch (out_dt) {
          case FFQ_I8: FFQ_GEMM3(int8_t, true); break;
          case FFQ_BF16: FFQ_GEMM3(bf16_t, true); break;
          case FFQ_F16: FFQ_GEMM3(f16_t, true); break;
          case FFQ_F32: FFQ_GEMM3(float, true); break;
          default: return fail(FFQ_ERR_DTYPE, "re-quantized output container must be i8, bf16, f16 or f32");
        }
      } else {
        switch (out_dt) {
          case FFQ_BF16: FFQ_GEMM3(bf16_t, false); break;
          case FFQ_F16: FFQ_GEMM3(f16_t, false); break;
          default: FFQ_GEMM3(float, false); break;
        }
      }
#undef FFQ_GEMM3
#undef FFQ_GEMM3_W
#undef FFQ_GEMM3_FP
      return check_launch("w8a8_gemm256pp_kernel");
    }
    const int nw = force_nw ? force_nw : 8;
    const int bn = nw * 32;
    a.tiles_m = (int)((M + BM2 - 1) / BM2);
    a.tiles_n = (int)((N + bn - 1) / bn);
    if (residual) return fail(FFQ_ERR_LAUNCH, "internal: the residual add reached a kernel that does not implement it");
    const unsigned grid2 = (unsigned)(a.tiles_m * a.tiles_n);
    const size_t ring_bytes = (size_t)STAGES2 * (BM2 + bn) * BK2;
    const size_t epilogue_bytes = (size_t)nw * (128 * 144 + 256);  // one padded 128 x 64 bf16 tile per wave
    const size_t lds_bytes = ring_bytes > epilogue_bytes ? ring_bytes : epilogue_bytes;
#define FFQ_GEMM2_NW(T, RQ, NW, WO)                                                                        \
  do {                                                                                                     \
    static uint64_t attr_set = 0;                                                                          \
    if (first_use_on_this_device(&attr_set)) {                                                                                       \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&w8a8_gemm256_kernel<T, RQ, NW, WO>),        \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);               \
    }                                                                                                      \
    w8a8_gemm256_kernel<T, RQ, NW, WO><<<grid2, NW * 64, lds_bytes, s>>>(a);                               \
  } while (0)
#define FFQ_GEMM2(T, RQ)                                                                                   \
  do {                                                                                                     \
    if (nw == 8) { if (w_offset) FFQ_GEMM2_NW(T, RQ, 8, true); else FFQ_GEMM2_NW(T, RQ, 8, false); }       \
    else { if (w_offset) FFQ_GEMM2_NW(T, RQ, 4, true); else FFQ_GEMM2_NW(T, RQ, 4, false); }               \
  } while (0)
    if (requant) {
      switch (out_dt) {
        case FFQ_I8: FFQ_GEMM2(int8_t, true); break;
        case FFQ_BF16: FFQ_GEMM2(bf16_t, true); break;
        case FFQ_F16: FFQ_GEMM2(f16_t, true); break;
        case FFQ_F32: FFQ_GEMM2(float, true); break;
        default: return fail(FFQ_ERR_DTYPE, "re-quantized output container must be i8, bf16, f16 or f32");
      }
    } else {
      switch (out_dt) {
        case FFQ_BF16: FFQ_GEMM2(bf16_t, false); break;
        case FFQ_F16: FFQ_GEMM2(f16_t, false); break;
        default: FFQ_GEMM2(float, false); break;
      }
    }
#undef FFQ_GEMM2
#undef FFQ_GEMM2_NW
    return check_launch("w8a8_gemm256_kernel");
  }
  // 128^2 kernel (small problems, K tails): row sums by a separate one-pass reduction
  if (residual) return fail(FFQ_ERR_LAUNCH, "internal: the residual add reached a kernel that does not implement it");
  if (w_offset) {  // needs sum_k xq[m,k]
    rowsum_i8_kernel<<<(unsigned)((M + 3) / 4), 256, 0, s>>>(xq, (int)M, (int)K, ws);
    a.rowsum_x = ws;
  }
  if (x_offset) {  // needs sum_k wq[n,k]
    if (w_rowsum) {
      a.rowsum_w = w_rowsum;
    } else {
      rowsum_i8_kernel<<<(unsigned)((N + 3) / 4), 256, 0, s>>>(wq, (int)N, (int)K, ws + M);
      a.rowsum_w = ws + M;
    }
  }
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

// ---- gate_proj + up_proj + SiLU * up + the down_proj input quantizer in one launch -----------------------------
extern "C" size_t ffq_mlp_gate_up_w8a8_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  (void)M; (void)K;
  if (N < 0) return 0;
  return (size_t)((2 * N * 4 + 255) & ~(int64_t)255);
}

extern "C" int ffq_mlp_gate_up_w8a8(const int8_t* xq, const int8_t* gate_wq, const int8_t* up_wq, const float* x_scale,
                                    const float* x_offset, const float* gate_w_scale, const float* up_w_scale,
                                    int8_t* codes_out, const float* out_scale, const float* out_offset, double out_num_bits,
                                    int64_t M, int64_t N, int64_t K, void* workspace, size_t workspace_bytes, void* stream) {
  return ffq_mlp_gate_up_w8a8_rs(xq, gate_wq, up_wq, nullptr, nullptr, x_scale, x_offset, gate_w_scale, up_w_scale, codes_out, out_scale,
                                 out_offset, out_num_bits, M, N, K, workspace, workspace_bytes, stream);
}

extern "C" int ffq_mlp_gate_up_w8a8_rs(const int8_t* xq, const int8_t* gate_wq, const int8_t* up_wq, const int32_t* gate_rowsum,
                                       const int32_t* up_rowsum, const float* x_scale, const float* x_offset,
                                       const float* gate_w_scale, const float* up_w_scale, int8_t* codes_out,
                                       const float* out_scale, const float* out_offset, double out_num_bits, int64_t M,
                                       int64_t N, int64_t K, void* workspace, size_t workspace_bytes, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool have_sums = gate_rowsum && up_rowsum;
  if (M < 0 || N < 0 || K < 0) return fail(FFQ_ERR_ARG, "negative extent");
  if (M == 0 || N == 0) return FFQ_OK;
  if (!xq || !gate_wq || !up_wq || !x_scale || !gate_w_scale || !up_w_scale || !codes_out || !out_scale)
    return fail(FFQ_ERR_ARG, "NULL buffer");
  if (M > INT32_MAX || N > INT32_MAX || K > INT32_MAX) return fail(FFQ_ERR_ARG, "extent exceeds 2^31");
  if (N % 128 != 0 || K % BK2 != 0 || K / BK2 < 4 || !aligned16(xq) || !aligned16(gate_wq) || !aligned16(up_wq) || !aligned16(codes_out))
    return fail(FFQ_ERR_DTYPE, "fused gate/up kernel needs N %% 128 == 0, K %% 64 == 0, K >= 256 and 16-byte aligned buffers");
  if (!(out_num_bits >= 1 && out_num_bits <= 8 && out_num_bits == floor(out_num_bits)))
    return fail(FFQ_ERR_PRECISION, "Provided dtype (%d) is not enough to store %g bits quantized values.", FFQ_I8, out_num_bits);
  const size_t need = ffq_mlp_gate_up_w8a8_workspace_bytes(M, N, K);
  if (x_offset && !have_sums && (need > workspace_bytes || !workspace)) return fail(FFQ_ERR_WORKSPACE, "fused gate/up needs %zu workspace bytes, got %zu", need, workspace_bytes);
  LinearArgs a;
  a.xq = xq; a.wq = gate_wq; a.wq2 = up_wq;
  a.x_scale = x_scale; a.x_offset = x_offset;
  a.w_scale = gate_w_scale; a.w_scale2 = up_w_scale; a.w_offset = nullptr;
  a.residual = nullptr;
  a.rowsum_x = nullptr; a.rowsum_w = nullptr; a.rowsum_w2 = nullptr;
  a.bias = nullptr; a.bias_dt = 0;
  a.out = codes_out; a.out_dt = FFQ_I8;
  a.out_scale = out_scale; a.out_offset = out_offset;
  const double lo = -pow(2.0, out_num_bits - 1.0);
  a.out_lo = (float)lo; a.out_hi = (float)(-lo - 1.0);
  a.x_per_row = 0; a.w_per_row = 1;
  a.M = (int)M; a.N = (int)N; a.K = (int)K;
  a.tiles_m = (int)((M + BM2 - 1) / BM2);
  a.tiles_n = (int)(N / 128);
  a.group_m = K >= 8192 ? 4 : GROUP_M2;
  static const int force_gm = getenv("FFQ_GROUP_M") ? atoi(getenv("FFQ_GROUP_M")) : 0;
  if (force_gm > 0) a.group_m = force_gm;
  static const int debug_bits = getenv("FFQ_GEMM_DEBUG") ? atoi(getenv("FFQ_GEMM_DEBUG")) : 0;
  a.debug = debug_bits;
  if (x_offset && have_sums) {
    a.rowsum_w = gate_rowsum; a.rowsum_w2 = up_rowsum;
  } else if (x_offset) {
    int32_t* ws = static_cast<int32_t*>(workspace);
    rowsum_i8_kernel<<<(unsigned)((N + 3) / 4), 256, 0, s>>>(gate_wq, (int)N, (int)K, ws);
    rowsum_i8_kernel<<<(unsigned)((N + 3) / 4), 256, 0, s>>>(up_wq, (int)N, (int)K, ws + N);
    a.rowsum_w = ws; a.rowsum_w2 = ws + N;
  }
  const size_t lds = (size_t)STAGES3 * (BM2 + 256) * BK2;  // the ring (128 KiB) also holds the 36 KiB output tile
  static uint64_t attr_set = 0;
  if (first_use_on_this_device(&attr_set)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&w8a8_gemm256pp_kernel<int8_t, true, false, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&w8a8_gemm256fl_kernel<int8_t, true, false, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  static const int use_fl = getenv("FFQ_GEMM_FL") ? atoi(getenv("FFQ_GEMM_FL")) : 1;
  static const int use_fp = getenv("FFQ_GEMM_FP") ? atoi(getenv("FFQ_GEMM_FP")) : 1;  // persistent tile loop: +4.9 % (A/B on one box)
  if (use_fl && use_fp && K % 128 == 0) {
    static uint64_t attr_set_fp = 0;
    const size_t lds_fp = (size_t)2 * (BM2 + 256) * 128 + kSiluBytes;  // two operand slots + the silu table
    if (first_use_on_this_device(&attr_set_fp)) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&w8a8_gemm256fp_kernel<int8_t, true, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_fp);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&w8a8_gemm256fq_kernel<int8_t, true, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_fp);
    }
    const int total = a.tiles_m * a.tiles_n;
    static const int fq_env = getenv("FFQ_GEMM_FQ") ? atoi(getenv("FFQ_GEMM_FQ")) : 1;
    const bool use_fq = fq_env && (uint64_t)M * (uint64_t)K < (1ull << 32) && (uint64_t)N * (uint64_t)K < (1ull << 32);
    if (use_fq) w8a8_gemm256fq_kernel<int8_t, true, true><<<(unsigned)(total < 256 ? total : 256), 512, lds_fp, s>>>(a, total);
    else w8a8_gemm256fp_kernel<int8_t, true, true><<<(unsigned)(total < 256 ? total : 256), 512, lds_fp, s>>>(a, total);
    return check_launch("w8a8_gemm256fp_kernel (mlp mode)");
  }
  if (use_fl && K % 128 == 0) w8a8_gemm256fl_kernel<int8_t, true, false, true><<<(unsigned)(a.tiles_m * a.tiles_n), 512, lds, s>>>(a);
  else w8a8_gemm256pp_kernel<int8_t, true, false, true><<<(unsigned)(a.tiles_m * a.tiles_n), 512, lds, s>>>(a);
  return check_launch("w8a8_gemm256 (mlp mode)");
}

// ffq_linear4w.hip — A6, the W8A8 GEMM as ONE wavefront per SIMD (experiment: FFQ_GEMM_4W=1 routes the plain bf16-output
// launches of ffq_linear_w8a8 here; ffq_linear.hip's 8-wave ping-pong kernel stays the shipped one until this wins an A/B).
//
// Why: the vendor library's int8 GEMM (256 x 256 x 128 macro tile, 32x32 MFMA, hand-scheduled assembly) reaches 2.3-2.6 POP/s
// on the shapes where the 8-wave kernel reaches 2.2-2.4 (profiles/r02_blas_probe.txt). Its shape is the classic one:
// 4 wavefronts, each owning 128 x 128 of the 256 x 256 tile in 256 accumulator registers, so a k-step of 16 MFMAs needs only
// 8 fragment reads (0.5 per MFMA against 0.75 with 128 x 64 per wave), and every memory instruction sits in the shadow of
// the wave's OWN MFMAs — no second wave, no per-phase barriers.
//   * register staging: 16-byte buffer loads (tile base in a wave-uniform descriptor, one per-lane offset, the rest in the
//     scalar offset) bring super-step ks + 2 into 64 VGPRs while ks is computed; they are written to LDS (ds_write_b128, bank
//     swizzle on the write address) during the first k-step of ks + 1;
//   * two 64 KiB LDS buffers (128 k-bytes per row: whole 128-byte lines per wave instruction) and ONE barrier per super-step,
//     placed before its last k-step: everything read from the current buffer has been requested before it, everything written
//     to the other buffer is complete at it, and the last k-step prefetches the next super-step's first fragments behind it;
//   * fragments double-buffered in registers (2 x 8 x 4 VGPRs), read one k-step ahead.
#include "ffq_common.h"
#include "ffq_vec.h"

namespace ffq {

typedef int q_v4i __attribute__((ext_vector_type(4)));
typedef int q_v16i __attribute__((ext_vector_type(16)));

struct Linear4wArgs {
  const int8_t* xq; const int8_t* wq;
  const float* x_scale; const float* x_offset;  // per tensor
  const float* w_scale;                         // per output channel (or one)
  const int32_t* rowsum_w;                      // sum_k wq[n, k] when x has an offset
  void* out;                                    // bf16 [M, N]
  int w_per_row;
  int M, N, K;
  int tiles_m, tiles_n;
};

constexpr int Q4_SLOT = 512 * 128;  // one LDS buffer: X image (256 x 128 B) then W image

__global__ __launch_bounds__(256, 1) void w8a8_gemm4w_kernel(Linear4wArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  // XCD-aware grouped tile order (as the other GEMMs)
  const uint32_t nblk = gridDim.x;
  const uint32_t xcd = blockIdx.x & 7u, slot_in_xcd = blockIdx.x >> 3;
  const uint32_t q8 = nblk >> 3, r8 = nblk & 7u;
  const uint32_t tile_id = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot_in_xcd;
  const uint32_t per_group = 8u * (uint32_t)a.tiles_n;
  const uint32_t group = tile_id / per_group, in_group = tile_id - group * per_group;
  const uint32_t group_rows = min(8u, (uint32_t)a.tiles_m - group * 8u);
  const int m0 = (int)(group * 8u + in_group % group_rows) * 256;
  const int n0 = (int)(in_group / group_rows) * 256;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // staging map: thread t covers rows (t >> 3) + 32 c, c = 0..7, 16-byte slot t & 7 of both operands
  const size_t x_left = (size_t)(a.M - m0) * a.K, w_left = (size_t)(a.N - n0) * a.K;
  const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.xq + (size_t)m0 * a.K), 0, (int)(x_left < 0x7FFFFFFFu ? x_left : 0x7FFFFFFFu), 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.wq + (size_t)n0 * a.K), 0, (int)(w_left < 0x7FFFFFFFu ? w_left : 0x7FFFFFFFu), 0x00020000);
  const uint32_t s_row = (uint32_t)tid >> 3, s_slot = (uint32_t)tid & 7u;
  const uint32_t g_voff = s_row * (uint32_t)a.K + s_slot * 16u;
  const uint32_t l_woff = s_row * 128u + ((s_slot ^ ((s_row >> 1) & 7u)) << 4);  // + c * 32 rows * 128 B: the swizzle term is unchanged

  // two staging sets: a super-step is fetched TWO super-steps before it is written to LDS (an L2 miss served by the
  // Infinity Cache takes longer than one super-step's 2048 MFMA cycles, and one wave per SIMD has nobody to hide behind)
  q_v4i sxp[8], swp[8], sxq[8], swq[8];
  auto fetch = [&](int ks, q_v4i (&sx)[8], q_v4i (&sw)[8]) {
    const uint32_t kb = (uint32_t)ks * 128u;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      sx[c] = __builtin_bit_cast(q_v4i, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, g_voff, kb + (uint32_t)c * 32u * (uint32_t)a.K, 0));
      sw[c] = __builtin_bit_cast(q_v4i, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, g_voff, kb + (uint32_t)c * 32u * (uint32_t)a.K, 0));
    }
  };
  auto stash = [&](uint8_t* buf, const q_v4i (&sx)[8], const q_v4i (&sw)[8]) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      *reinterpret_cast<q_v4i*>(buf + l_woff + c * (32 * 128)) = sx[c];
      *reinterpret_cast<q_v4i*>(buf + 256 * 128 + l_woff + c * (32 * 128)) = sw[c];
    }
  };

  // fragment addresses: k-step v (32 k-bytes): lane (r, g) reads logical slot 2 v + g of its row
  const uint32_t frag_row = lane & 31, frag_g = lane >> 5;
  uint32_t x_off[4], w_off[4];
  {
    const uint32_t xr = wm * 128 + frag_row, wr = wn * 128 + frag_row;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      x_off[v] = xr * 128 + ((((uint32_t)(2 * v) + frag_g) ^ ((xr >> 1) & 7u)) << 4);
      w_off[v] = 256 * 128 + wr * 128 + ((((uint32_t)(2 * v) + frag_g) ^ ((wr >> 1) & 7u)) << 4);
    }
  }
  q_v4i fxa[4], fwa[4], fxb[4], fwb[4];
  auto read_frags = [&](const uint8_t* buf, int v, q_v4i (&fx)[4], q_v4i (&fw)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) fw[j] = *reinterpret_cast<const q_v4i*>(buf + w_off[v] + j * (32 * 128));
#pragma unroll
    for (int i = 0; i < 4; ++i) fx[i] = *reinterpret_cast<const q_v4i*>(buf + x_off[v] + i * (32 * 128));
  };

  q_v16i acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0;
  auto mfmas = [&](const q_v4i (&fx)[4], const q_v4i (&fw)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fw[j], fx[i], acc[i][j], 0, 0, 0);
  };

  const int ksuper = a.K / 128;  // even (checked by the host)
  auto clampk = [&](int ks) { return ks < ksuper ? ks : ksuper - 1; };
  // prologue: super-step 0 into buffer 0, super-steps 1 and 2 into the staging sets
  fetch(0, sxp, swp);
  stash(lds, sxp, swp);
  fetch(clampk(1), sxp, swp);
  fetch(clampk(2), sxq, swq);
  __syncthreads();
  read_frags(lds, 0, fxa, fwa);

  // one super-step: `sx / sw` hold super-step ks + 1 (written to `nxt` now) and are refilled with super-step ks + 3
  auto super_step = [&](int ks, uint8_t* cur, uint8_t* nxt, q_v4i (&sx)[8], q_v4i (&sw)[8]) {
    // ---- k-step 0: MFMAs on set a | write the staged super-step ks + 1 into the other buffer, read k-step 1 into set b
    stash(nxt, sx, sw);
    read_frags(cur, 1, fxb, fwb);
    mfmas(fxa, fwa);
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
      __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);  // 2 LDS writes
    }
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // 1 LDS read
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- k-step 1: MFMAs on set b | fetch super-step ks + 3 into the set just emptied, read k-step 2 into set a
    fetch(clampk(ks + 3), sx, sw);
    read_frags(cur, 2, fxa, fwa);
    mfmas(fxb, fwb);
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // 1 global load
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // 1 LDS read every other MFMA
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- k-step 2: MFMAs on set a | read k-step 3 into set b
    read_frags(cur, 3, fxb, fwb);
    mfmas(fxa, fwa);
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    // everything of `cur` has been requested, everything written to `nxt` is on its way: one barrier per super-step
    __syncthreads();
    // ---- k-step 3: MFMAs on set b | read k-step 0 of super-step ks + 1 into set a
    read_frags(nxt, 0, fxa, fwa);
    mfmas(fxb, fwb);
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  for (int ks = 0; ks < ksuper; ks += 2) {
    super_step(ks, lds, lds + Q4_SLOT, sxp, swp);
    super_step(ks + 1, lds + Q4_SLOT, lds, sxq, swq);
  }
  __syncthreads();

  // ---- epilogue: y = (sx * sw[n]) * (acc + ox * rowsum_w[n]) -> bf16, 32-row slabs through LDS, 16-byte stores.
  // lane l holds C[m = i*32 + (l & 31)][n = j*32 + 8 q + 4 (l >> 5) + (0..3)] in acc[i][j][4 q + (0..3)]
  bf16_t* out = static_cast<bf16_t*>(a.out);
  constexpr int PITCH = 128 * 2 + 16;  // 128 bf16 columns + pad
  uint8_t* slab = lds + wave * (32 * PITCH + 1024);
  float* colp = reinterpret_cast<float*>(slab + 32 * PITCH);  // [2][128]: weight scale, weight row sum
  const int wave_n0 = n0 + wn * 128, wave_m0 = m0 + wm * 128;
  const float sxv = a.x_scale[0];
  const float oxv = a.x_offset ? rne(a.x_offset[0]) : 0.0f;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    int n = wave_n0 + h * 64 + lane;
    n = n < a.N ? n : a.N - 1;
    colp[h * 64 + lane] = a.w_scale[a.w_per_row ? n : 0];
    colp[128 + h * 64 + lane] = a.rowsum_w ? (float)a.rowsum_w[n] : 0.0f;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const bool full = wave_n0 + 128 <= a.N && (a.N & 7) == 0;
#pragma unroll 1
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int nb = j * 32 + 8 * q + 4 * (int)frag_g;
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        const f32x4 sw4 = *reinterpret_cast<const f32x4*>(colp + nb);
        const f32x4 rs4 = *reinterpret_cast<const f32x4*>(colp + 128 + nb);
        float y[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int raw = i == 0 ? acc[0][j][4 * q + t] : i == 1 ? acc[1][j][4 * q + t] : i == 2 ? acc[2][j][4 * q + t] : acc[3][j][4 * q + t];
          const float v = (float)raw + oxv * rs4[t];
          y[t] = (sxv * sw4[t]) * v;
        }
        u32x2 pk;
        pk.x = pack2<bf16_t>(y[0], y[1]);
        pk.y = pack2<bf16_t>(y[2], y[3]);
        *reinterpret_cast<u32x2*>(slab + frag_row * PITCH + nb * 2) = pk;
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (full) {
#pragma unroll
      for (int t = 0; t < 8; ++t) {  // 32 rows x 16 segments of 16 B
        const int c = lane + 64 * t;
        const int row = c >> 4, seg = c & 15;
        const int mm = wave_m0 + i * 32 + row;
        const u32x4 v = *reinterpret_cast<const u32x4*>(slab + row * PITCH + seg * 16);
        if (mm < a.M) *reinterpret_cast<u32x4*>(reinterpret_cast<uint8_t*>(out) + ((size_t)mm * a.N + wave_n0) * 2 + seg * 16) = v;
      }
    } else {
      for (int c = lane; c < 32 * 128; c += 64) {
        const int row = c >> 7, col = c & 127;
        const int mm = wave_m0 + i * 32 + row;
        if (mm < a.M && wave_n0 + col < a.N) out[(size_t)mm * a.N + wave_n0 + col] = *reinterpret_cast<const bf16_t*>(slab + row * PITCH + col * 2);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

// one wavefront per row: sum of K int8 codes (as ffq_linear.hip)
__global__ __launch_bounds__(256) void rowsum4w_i8_kernel(const int8_t* __restrict__ q, int rows, int K, int32_t* __restrict__ sums) {
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

// Experiment entry point (not part of the C ABI): 0 if the launch was taken, non-zero if the caller should go on with its own kernels.
int linear4w_try(const int8_t* xq, const int8_t* wq, const int32_t* w_rowsum, const float* x_scale, const float* x_offset, const float* w_scale,
                 int w_per_row, void* out, int64_t M, int64_t N, int64_t K, int32_t* workspace, hipStream_t s) {
  if (K % 256 != 0 || K < 512 || M < 256 || N < 256) return 1;  // an even number of 128-wide super-steps
  Linear4wArgs a;
  a.xq = xq; a.wq = wq; a.x_scale = x_scale; a.x_offset = x_offset; a.w_scale = w_scale; a.w_per_row = w_per_row;
  a.out = out; a.M = (int)M; a.N = (int)N; a.K = (int)K;
  a.tiles_m = (int)((M + 255) / 256); a.tiles_n = (int)((N + 255) / 256);
  a.rowsum_w = nullptr;
  if (x_offset) {
    if (w_rowsum) a.rowsum_w = w_rowsum;
    else {
      rowsum4w_i8_kernel<<<(unsigned)((N + 3) / 4), 256, 0, s>>>(wq, (int)N, (int)K, workspace + M);
      a.rowsum_w = workspace + M;
    }
  }
  const size_t lds_bytes = 2 * Q4_SLOT;
  static uint64_t attr_set = 0;
  if (first_use_on_this_device(&attr_set))
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&w8a8_gemm4w_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  w8a8_gemm4w_kernel<<<(unsigned)(a.tiles_m * a.tiles_n), 256, lds_bytes, s>>>(a);
  return 0;
}

}  // namespace ffq

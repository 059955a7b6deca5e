// ffq_wmid.hip — the weight-only quantized linear for a FEW HUNDRED token rows (17 <= M <= 512; any M <= 512 whose weight storage the
// skinny form of ffq_wskinny.hip declines): short prompts, speculative / batched decode steps, the tail chunk of a prefill.
//
// Same contract as ffq_wlinear.hip (reference _gen/fallback.py:86-112: y = F.linear(x, dequantize(w)); the B operand of the bf16 MFMA is
// bit for bit A2's bf16 value ((float(q) + round(o)) * s, one RNE rounding), fp32 accumulation, only the summation order is this
// kernel's own), a different regime. Up to round 5 these launches took the 256 x 256 tiles of ffq_wlinear.hip cut along K: a 512-row
// q/o projection is 32 tiles, so every tile was cut into 8 slices whose 256 KiB fp32 partials went through HBM — 64 MB written and
// 64 MB read back for a problem whose operands are 20 MB (missing #1) — and below 129 rows the skinny form streams every
// activation fragment from LDS once per MFMA (one weight tile per wave), which the LDS pipe bounds from 32 rows on. Here
//   * a block owns a BM x 128 output tile, BM = 64 (128 for contractions deeper than 8192 above 256 rows: md_bm): 4 - 8 x the tiles of
//     the 256-row form, so that a 512-row q/o projection fills the chip without any K slice and k/v with 4 (wq_mid_split: at most 4);
//   * TWO kernels of the same tiles, the same K slices and the same k order (hence the same bits, tests/test_mid_gpu.py):
//       - wq_mid_dma_kernel (further down; what runs for int8 containers and packing blocks >= 128): 8 waves side by side along N, both
//         operands by LDS-DMA into a ring of 3 - 4 stages, codes converted on their way from LDS into the MFMA's registers;
//       - wq_mid_kernel (below; packing blocks 32 / 64, and the test hook's bit 2): 4 waves (2 x 2), both operands staged through
//         REGISTERS into a k-ordered, XOR-swizzled LDS image, one 64-k super-step per stage, double buffered; codes converted ONCE PER
//         BLOCK with A2's arithmetic (dequantize4 of ffq_wq.h, the function every form uses) on their way in; the code loads run
//         MD_BDEPTH super-steps ahead of their conversion, the activation loads MD_ADEPTH (they are L2-resident: M x K x 2 <= 15 MB);
//     int8 containers and packed nibbles of every packing block >= 32 (GGUF's 32 / 64, config 4's 128, 256: a piece of 16 codes
//     lives in one nibble position of 16 contiguous bytes) give the SAME operand, so every storage form of one weight gives the same bits;
//   * v_mfma_f32_16x16x32_bf16 on 16-row tiles: every fragment read feeds MI = BM / 16 (DMA form) or 4 / 2 (register form) MFMAs, where
//     the skinny form needs one read per MFMA; two blocks per CU hide each other's barriers and conversion work;
//   * K is cut into S slices across blocks where the tiles alone do not fill the chip. A wave leaves its partial in a write-through
//     slab, takes a ticket for its part of the tile, and the LAST wave to arrive — whoever it is — adds the S partials in slice
//     order and writes the output: nobody waits for anybody (the exchange of ffq_wskinny.hip), the summation order is a function of
//     the plan alone (bit-reproducible), ticket words are zero before and after.
// Covered: everything ffq_linear_wq_supported() admits with M <= 512 — one to three weight matrices on the same activations
// (every matrix but the last a multiple of 128 rows), per-tensor / per-channel / per-group parameters, offsets, bias, bf16 / f32
// output. The gate+up+SiLU*up launch keeps the 256-row-tile kernel.
#include "ffq_wq.h"

#include <math.h>

#include <type_traits>
#include <utility>

namespace ffq {

constexpr int MD_BN = 128;      // weight rows (output columns) per tile
#ifndef FFQ_MD_BDEPTH
#define FFQ_MD_BDEPTH 4
#endif
#ifndef FFQ_MD_ADEPTH
#define FFQ_MD_ADEPTH 2
#endif
constexpr int MD_BDEPTH = FFQ_MD_BDEPTH;  // super-steps of code loads in flight (ring of registers; A/B hook: tools/build_variant.sh)
constexpr int MD_ADEPTH = FFQ_MD_ADEPTH;  // ... of activation loads
constexpr int md_lcm(int a, int b) { int x = a; while (x % b) x += a; return x; }
constexpr int MD_UNROLL = md_lcm(md_lcm(MD_BDEPTH, MD_ADEPTH), 2);  // steps after which (slot, code ring, activation ring) repeat

template <typename F, int... I>
__device__ __forceinline__ void md_each(F&& fn, std::integer_sequence<int, I...>) { (fn(std::integral_constant<int, I>{}), ...); }

struct MidArgs {
  const uint8_t* x;
  const uint8_t* w[3]; const float* scale[3]; const float* offset[3]; void* out[3];
  int seg_n[3];          // rows of each weight matrix (0: absent)
  int seg_tile[3];       // first column tile of matrices 1 and 2 (seg_tile[0] = 0); INT32_MAX: absent
  const void* bias; int bias_dt;
  int out_dt;
  int M, K;
  int tiles_m, tiles_n;  // tiles_n: column tiles over all matrices
  int groups;            // parameters per row along K (1: per channel / per tensor)
  int steps_per_group;   // super-steps of 64 k that share one group
  int per_row;
  int pack_shift;        // WL_B_I4: log2(packing block)
  int S;                 // K slices across blocks
  float* slabs;          // [tile][S][wave][MI * 4][64 lanes] x 16 B
  int* tickets;          // [tile][wave]
};

template <int BKIND, bool GROUPED, bool OFFSET, int BM>
__global__ __launch_bounds__(256, 2) void wq_mid_kernel(MidArgs a) {
  constexpr int MI = BM / 32;                  // 16-row tiles of a wave along M (the wave contracts (BM / 2) x 64)
  constexpr int A_IMAGE = BM * 128, SLOT = A_IMAGE + MD_BN * 128;
  constexpr int AP = BM / 32;                  // 16-byte activation pieces per thread and super-step
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const uint32_t r16 = lane & 15, g4 = lane >> 4;

  // unit = (row tile, column tile, K slice); the units of one (column tile, slice) over the row tiles are tiles_n * S apart: with
  // tiles_n * S a multiple of 8 they run on ONE XCD and share the tile's weight bytes in its L2
  const int per_m = a.tiles_n * a.S;
  const int tm = (int)blockIdx.x / per_m, rest = (int)blockIdx.x - tm * per_m;
  const int tn_all = rest / a.S, slice = rest - tn_all * a.S;
  const int seg = tn_all >= a.seg_tile[2] ? 2 : tn_all >= a.seg_tile[1] ? 1 : 0;  // block-uniform selects: no dynamic indexing of the arguments
  const uint8_t* const w_base = seg == 0 ? a.w[0] : seg == 1 ? a.w[1] : a.w[2];
  const float* const s_base = seg == 0 ? a.scale[0] : seg == 1 ? a.scale[1] : a.scale[2];
  const float* const o_base = seg == 0 ? a.offset[0] : seg == 1 ? a.offset[1] : a.offset[2];
  const int rows = seg == 0 ? a.seg_n[0] : seg == 1 ? a.seg_n[1] : a.seg_n[2];
  const int n0 = (tn_all - (seg == 0 ? 0 : seg == 1 ? a.seg_tile[1] : a.seg_tile[2])) * MD_BN;  // first row of the tile inside its matrix
  const int m0 = tm * BM;
  const int ksuper = a.K / 64;
  const int k0 = (int)((int64_t)slice * ksuper / a.S), k1 = (int)((int64_t)(slice + 1) * ksuper / a.S);
  const int nsteps = k1 - k0;

  // ---- the activation stream: piece i of thread t = row (t / 8) + 32 i of the tile, 16-byte slot t % 8 of the row's 128 bytes
  const uint8_t* x_ptr[AP];
#pragma unroll
  for (int i = 0; i < AP; ++i) {
    int m = m0 + (tid >> 3) + 32 * i;
    m = m < a.M ? m : a.M - 1;  // rows past the edge re-read the last row and are never stored
    x_ptr[i] = a.x + (size_t)m * (size_t)a.K * 2u + (size_t)(tid & 7) * 16u;
  }
  // ---- the code stream: piece j of thread t = row (t / 4) + 64 j of the tile, codes [16 p, 16 p + 16) of the super-step, p = t % 4
  const uint32_t c_piece = (uint32_t)tid & 3u;
  const uint32_t w_row_bytes = BKIND == WL_B_I8 ? (uint32_t)a.K : (uint32_t)a.K / 2u;
  const uint8_t* w_ptr[2];
  const float* s_ptr[2];
  [[maybe_unused]] const float* o_ptr[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    int row = n0 + (tid >> 2) + 64 * j;
    row = row < rows ? row : rows - 1;
    w_ptr[j] = w_base + (size_t)row * w_row_bytes;
    const size_t p_row = a.per_row ? (size_t)row * (size_t)a.groups : 0;
    s_ptr[j] = s_base + p_row;
    if constexpr (OFFSET) o_ptr[j] = o_base + p_row;
  }
  // (scalars the stream lambdas use, copied out of the argument struct: with `a.pack_shift` named inside the unrolled ring hipcc kept
  // the whole by-value struct in scratch memory for the nibble instantiations — 262 scratch instructions, found by tools/kernel_resources.py)
  [[maybe_unused]] const uint32_t pack_shift = (uint32_t)a.pack_shift;
  [[maybe_unused]] const int steps_per_group = a.steps_per_group;
  u32x4 xa[MD_ADEPTH][AP];
  u32x4 raw[MD_BDEPTH][2];
  [[maybe_unused]] uint32_t nib[MD_BDEPTH];  // WL_B_I4: 0 = the low nibbles of the bytes, 4 = the high ones
  [[maybe_unused]] float sc[MD_BDEPTH][2], ro[MD_BDEPTH][2];
  float s_row[2] = {1.0f, 1.0f}, o_row[2] = {0.0f, 0.0f};
  if constexpr (!GROUPED) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      s_row[j] = s_ptr[j][0];
      if constexpr (OFFSET) o_row[j] = rne(o_ptr[j][0]);
    }
  }
  auto load_a = [&](int ks, auto dc) __attribute__((always_inline)) {
    constexpr int d = decltype(dc)::value;
#pragma unroll
    for (int i = 0; i < AP; ++i) xa[d][i] = *reinterpret_cast<const u32x4*>(x_ptr[i] + (size_t)ks * 128u);
  };
  auto load_b = [&](int ks, auto dc) __attribute__((always_inline)) {
    constexpr int d = decltype(dc)::value;
    uint32_t byte0;
    if constexpr (BKIND == WL_B_I8) {
      byte0 = (uint32_t)ks * 64u + c_piece * 16u;
    } else {
      // codes kk .. kk + 15 of a row live in ONE half of ONE packing block (block >= 32): byte j of a block holds code j in its low and
      // code j + block / 2 in its high nibble (ffq_pack_int4, export/stages/gguf/_packing.py:44-53)
      const uint32_t kk = (uint32_t)ks * 64u + c_piece * 16u, lb = pack_shift;
      const uint32_t within = kk & ((1u << lb) - 1u);
      nib[d] = (within >> (lb - 1u)) * 4u;
      byte0 = ((kk >> lb) << (lb - 1u)) + (within & ((1u << (lb - 1u)) - 1u));
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) raw[d][j] = *reinterpret_cast<const u32x4*>(w_ptr[j] + byte0);
    if constexpr (GROUPED) {
      const int grp = ks / steps_per_group;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        sc[d][j] = s_ptr[j][grp];
        if constexpr (OFFSET) ro[d][j] = rne(o_ptr[j][grp]);
      }
    }
  };
  // registers -> the LDS images of `slot`: 16-byte slot q of row r lies at r * 128 + ((q ^ (r & 7)) << 4) (conflict-free stores and
  // fragment reads)
  auto store_a = [&](auto dc, int slot) __attribute__((always_inline)) {
    constexpr int d = decltype(dc)::value;
    uint8_t* image = lds + slot * SLOT;
#pragma unroll
    for (int i = 0; i < AP; ++i) {
      const uint32_t row = (uint32_t)(tid >> 3) + 32u * i;
      *reinterpret_cast<u32x4*>(image + row * 128u + ((((uint32_t)tid & 7u) ^ (row & 7u)) << 4)) = xa[d][i];
    }
  };
  auto store_b = [&](auto dc, int slot) __attribute__((always_inline)) {
    constexpr int d = decltype(dc)::value;
    uint8_t* image = lds + slot * SLOT + A_IMAGE;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const uint32_t row = (uint32_t)(tid >> 2) + 64u * j;
      uint32_t w[4] = {raw[d][j].x, raw[d][j].y, raw[d][j].z, raw[d][j].w};
      float s = GROUPED ? sc[d][j] : s_row[j];
      float c = OFFSET ? (GROUPED ? ro[d][j] : o_row[j]) : 0.0f;
      if constexpr (BKIND == WL_B_I4) {
        // nibble n = code + 8 -> (n ^ 8) << 4 in the byte's high half = 16 * code as a signed byte, and (16 q + 16 o) * (s / 16) is
        // (q + o) * s with the same single rounding wherever s / 16 is exact; a tiny scale takes the codes themselves (ffq_wlinear.hip)
        const bool tiny = __builtin_fabsf(s) < 0x1p-120f && s != 0.0f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          uint32_t b = (((w[q] >> nib[d]) << 4) & 0xF0F0F0F0u) ^ 0x80808080u;
          if (__builtin_expect(tiny, 0)) {
            const uint32_t b0 = (uint32_t)(((int32_t)(b << 24)) >> 28) & 0xFFu, b1 = (uint32_t)(((int32_t)(b << 16)) >> 28) & 0xFFu;
            const uint32_t b2 = (uint32_t)(((int32_t)(b << 8)) >> 28) & 0xFFu, b3 = (uint32_t)(((int32_t)b) >> 28) & 0xFFu;
            b = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
          }
          w[q] = b;
        }
        if (!tiny) { s = s * 0.0625f; c = c * 16.0f; }
      }
      uint32_t o[8];
#pragma unroll
      for (int q = 0; q < 4; ++q) dequantize4<OFFSET>(w[q], s, c, o[2 * q], o[2 * q + 1]);
      const uint32_t sw = row & 7u;
      *reinterpret_cast<u32x4*>(image + row * 128u + (((2u * c_piece) ^ sw) << 4)) = u32x4{o[0], o[1], o[2], o[3]};
      *reinterpret_cast<u32x4*>(image + row * 128u + (((2u * c_piece + 1u) ^ sw) << 4)) = u32x4{o[4], o[5], o[6], o[7]};
    }
  };

  // ---- fragments: lane (r16, g4) reads 8 bf16 of row r16 of a 16-row tile, logical slot kq * 4 + g4; the swizzle depends on the row
  // through r16 only (row tiles are 16 rows apart): one offset per k-chunk and operand, the row tile is a constant (t * 2048 bytes)
  uint32_t a_off[2], b_off[2];
  {
    const uint32_t arow = (uint32_t)wm * (BM / 2) + r16, brow = (uint32_t)wn * 64u + r16;
#pragma unroll
    for (int kq = 0; kq < 2; ++kq) {
      a_off[kq] = arow * 128u + (((kq * 4u + g4) ^ (arow & 7u)) << 4);
      b_off[kq] = A_IMAGE + brow * 128u + (((kq * 4u + g4) ^ (brow & 7u)) << 4);
    }
  }
  wl_v4f acc[MI][4];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int nj = 0; nj < 4; ++nj) acc[mi][nj] = wl_v4f{0.0f, 0.0f, 0.0f, 0.0f};
  auto compute = [&](int slot) __attribute__((always_inline)) {
    const uint8_t* st = lds + slot * SLOT;
#pragma unroll
    for (int kq = 0; kq < 2; ++kq) {
      wl_v4i fa[MI], fb[4];
#pragma unroll
      for (int nj = 0; nj < 4; ++nj) fb[nj] = *reinterpret_cast<const wl_v4i*>(st + b_off[kq] + nj * 2048);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) fa[mi] = *reinterpret_cast<const wl_v4i*>(st + a_off[kq] + mi * 2048);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int n_ = 0; n_ < 4; ++n_) {
          const int nj = (mi & 1) ? 3 - n_ : n_;  // snake order: every MFMA shares an operand with its predecessor
          acc[mi][nj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(wl_v8bf, fb[nj]), __builtin_bit_cast(wl_v8bf, fa[mi]), acc[mi][nj], 0, 0, 0);
        }
    }
  };

  // ---- the K-loop. Step i (super-step k0 + i) computes from LDS slot i % 2 while stage i + 1 goes from its registers into the other slot
  // (nobody reads that slot: its last readers passed the barrier behind step i - 1) and the registers just freed take the stage a ring
  // ahead; ONE barrier per step.
  auto each_b = [&](auto&& fn) { md_each(fn, std::make_integer_sequence<int, MD_BDEPTH>{}); };
  auto each_u = [&](auto&& fn) { md_each(fn, std::make_integer_sequence<int, MD_UNROLL>{}); };
  // Loads past the slice's last super-step are CLAMPED to it, never skipped: the loop body is straight-line code, so the compiler's
  // wait insertion counts the loads in flight exactly (`s_waitcnt vmcnt(N)` with N = what was issued behind the stage being stored).
  // A first version guarded every load with `if (step + depth < nsteps)`: the merged control flow made hipcc wait `vmcnt(0)` in every
  // step — the newest code loads included, i.e. one HBM round trip per 64-k step (0.83 us per step, profiles/r06_wq_mid_sweep_v1.txt).
  const int k_last = k1 - 1;
  auto clamped = [&](int ks) __attribute__((always_inline)) { return ks < k_last ? ks : k_last; };
  each_b([&](auto dc) { load_b(clamped(k0 + decltype(dc)::value), dc); });
  md_each([&](auto dc) { load_a(clamped(k0 + decltype(dc)::value), dc); }, std::make_integer_sequence<int, MD_ADEPTH>{});
  store_a(std::integral_constant<int, 0>{}, 0);
  store_b(std::integral_constant<int, 0>{}, 0);
  load_a(clamped(k0 + MD_ADEPTH), std::integral_constant<int, 0>{});
  load_b(clamped(k0 + MD_BDEPTH), std::integral_constant<int, 0>{});
  __syncthreads();
  // step `step` = i + u: compute from slot u % 2; stage step + 1 goes from its registers into the other slot (when step + 1 is past the
  // end: a copy of the last stage that nobody reads); the registers just freed take the stage a ring ahead
  auto body = [&](int i, auto uc) __attribute__((always_inline)) {
    constexpr int u = decltype(uc)::value;
    constexpr int cur = u & 1;
    constexpr int nb = (u + 1) % MD_BDEPTH, na = (u + 1) % MD_ADEPTH;
    const int step = i + u;
    compute(cur);
    store_a(std::integral_constant<int, na>{}, cur ^ 1);
    store_b(std::integral_constant<int, nb>{}, cur ^ 1);
    load_a(clamped(k0 + step + 1 + MD_ADEPTH), std::integral_constant<int, na>{});
    load_b(clamped(k0 + step + 1 + MD_BDEPTH), std::integral_constant<int, nb>{});
    __syncthreads();
  };
  int i = 0;
  for (; i + MD_UNROLL <= nsteps; i += MD_UNROLL) each_u([&](auto uc) { body(i, uc); });
  each_u([&](auto uc) {  // the last nsteps % MD_UNROLL steps
    if (i + decltype(uc)::value < nsteps) body(i, uc);
  });

  // ---- a wave's result: acc[mi][nj][t] = y[m0 + wm * BM / 2 + 16 mi + r16][n0 + wn * 64 + 16 nj + 4 g4 + t] (partial over this block's k slice)
  const int tile = tm * a.tiles_n + tn_all;
  if (a.S > 1) {
    constexpr size_t unit_bytes = (size_t)MI * 4 * 1024;  // one wave's partial of one slice
    uint8_t* const strip = reinterpret_cast<uint8_t*>(a.slabs) + ((size_t)tile * a.S * 4 + wave) * unit_bytes;  // slice 0 of this quadrant
    const size_t slice_stride = (size_t)4 * unit_bytes;
    {
      const auto mine = __builtin_amdgcn_make_buffer_rsrc(strip + (size_t)slice * slice_stride, 0, (int)unit_bytes, 0x00020000);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int nj = 0; nj < 4; ++nj)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(wl_v4u, acc[mi][nj]), mine, ((mi * 4 + nj) * 64 + lane) * 16, 0, /*sc1*/ 16);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the write-through stores have left before the ticket is taken
    int t = 0;
    if (lane == 0) t = __hip_atomic_fetch_add(a.tickets + tile * 4 + wave, 1, FFQ_TICKET_ORDER, __HIP_MEMORY_SCOPE_AGENT);
    t = __builtin_amdgcn_readfirstlane(t);
    if (t != a.S - 1) return;  // somebody else finishes this quadrant
    if (lane == 0) __hip_atomic_store(a.tickets + tile * 4 + wave, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // zero again for the next launch
    asm volatile("" ::: "memory");  // the partials are read after the ticket said everybody has written
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int nj = 0; nj < 4; ++nj) acc[mi][nj] = wl_v4f{0.0f, 0.0f, 0.0f, 0.0f};
    for (int sl = 0; sl < a.S; ++sl) {  // slice order, whoever reduces: the sum is a function of the plan alone
      const auto peer = __builtin_amdgcn_make_buffer_rsrc(strip + (size_t)sl * slice_stride, 0, (int)unit_bytes, 0x00020000);
      wl_v4u got[MI][4];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int nj = 0; nj < 4; ++nj) got[mi][nj] = __builtin_amdgcn_raw_buffer_load_b128(peer, ((mi * 4 + nj) * 64 + lane) * 16, 0, /*sc1*/ 16);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int nj = 0; nj < 4; ++nj) {
          const wl_v4f g = __builtin_bit_cast(wl_v4f, got[mi][nj]);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[mi][nj][e] = acc[mi][nj][e] + g[e];
        }
    }
  }
  // ---- epilogue: bias, cast, 4 consecutive columns per lane and row
  void* const out = seg == 0 ? a.out[0] : seg == 1 ? a.out[1] : a.out[2];
  const bool rows_by_4 = (rows & 3) == 0;
#pragma unroll
  for (int nj = 0; nj < 4; ++nj) {
    const int ncol = n0 + wn * 64 + 16 * nj + 4 * (int)g4;
    if (ncol >= rows) continue;
    float b4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (a.bias) {
#pragma unroll
      for (int e = 0; e < 4; ++e) b4[e] = ncol + e < rows ? (float)load_any(a.bias, a.bias_dt, ncol + e) : 0.0f;
    }
    const bool whole = ncol + 4 <= rows && rows_by_4;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int m = m0 + wm * (BM / 2) + 16 * mi + (int)r16;
      if (m >= a.M) continue;
      float y[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) y[e] = a.bias ? acc[mi][nj][e] + b4[e] : acc[mi][nj][e];
      const size_t at = (size_t)m * (size_t)rows + (size_t)ncol;
      if (a.out_dt == FFQ_BF16) {
        bf16_t* o = static_cast<bf16_t*>(out) + at;
        if (whole) {
          u32x2 pk;
          pk.x = pack2<bf16_t>(y[0], y[1]);
          pk.y = pack2<bf16_t>(y[2], y[3]);
          *reinterpret_cast<u32x2*>(o) = pk;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (ncol + e < rows) o[e] = from_f32<bf16_t>(y[e]);
        }
      } else {
        float* o = static_cast<float*>(out) + at;
        if (whole) {
          *reinterpret_cast<wl_v4f*>(o) = wl_v4f{y[0], y[1], y[2], y[3]};
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (ncol + e < rows) o[e] = y[e];
        }
      }
    }
  }
}

// ---- the LDS-DMA form (round 6, second version) ------------------------------------------------------------------------------------------
// The register-staged kernel above is bound by its LDS STORES: 32 KiB per 64-k step and block go VGPR -> LDS through `ds_write_b128`
// (13 cycles per wave-instruction, MI355X_MICROARCH.md "LDS"), 0.74 us per step measured with one block per CU against 0.27 for the step's
// 32 MFMAs (profiles/r06_wq_mid_sweep_v2.txt). Here NOTHING is stored to LDS by a wave:
//   * both operands arrive by LDS-DMA (`global_load_lds`, 16 bytes per lane straight from L2 / HBM into LDS) in a ring of 3 - 4
//     stages of one 64-k super-step each — activations as they are (bf16, 128 bytes per row, XOR swizzle on the SOURCE address as in
//     ffq_wlinear.hip), the weight CODES as they are stored (64 bytes per row: int8 containers, or the 64 packed bytes that hold the
//     super-step's nibbles for packing blocks >= 128) and, for grouped parameters, each wave's 32 scales / offsets of the step's group:
//     two to three stages in flight per block, two blocks per CU, without a single register;
//   * a wave owns 16 output columns for ALL BM rows (8 waves side by side along N, two per SIMD: while one issues its LDS-DMA requests
//     — ~60-100 cycles each in its in-order stream — or waits for fragments, the other feeds the matrix pipe; the first version had 4
//     waves of 32 columns, A/B in profiles/r06_wq_mid_waves_ab.txt): lane (r, g) reads the 8 code bytes it multiplies
//     (`ds_read_b64`, conflict-free under slot ^= ((row / 4) % 4) * 2), converts them with A2's arithmetic (dequantize4) INTO the MFMA's
//     operand registers — every weight code is converted exactly once per row tile, by the wave that consumes it (no redundancy across
//     waves, no LDS round trip of the bf16 image) — and 8 VALU conversions ride under each group of 8 (BM = 128) MFMAs;
//   * one `s_waitcnt vmcnt(NI * (RING - 2))` + one raw barrier per step: the stage about to be computed has landed for every wave,
//     and the slot computed a step ago is free for the stage RING - 1 ahead. The waits are written by hand (hipcc does not order
//     `ds_read` behind LDS-DMA), every issue is unconditional (stages past the slice's end re-read its last stage), so the count of
//     requests in flight is the same in every step.
// Same tiles, same K slices, same k order inside a tile (k ascending in steps of 32) as the register-staged kernel: both give the
// SAME bits for the same plan — and so do all storage forms (tests/test_mid_gpu.py). Packing blocks 32 / 64 (GGUF) keep the
// register-staged kernel (their 64 codes of a step are 32 bytes holding both nibbles).
// stages of the ring: 64-row tiles 4 x 16 KiB (two blocks per CU), 128-row tiles 3 x 24 KiB (two blocks per CU). Round-6 sweeps
// (profiles/r06_wq_mid_sweep_v3.txt, _v4.txt): one block per CU with a six-stage ring was no faster at one tile per CU and 18 % slower
// at two (gate/up at 512 rows 87 against 72 us) — from ~256 rows on these launches are bound by the L2 -> LDS traffic of the whole
// chip (8.5 TB/s measured), not by the prefetch distance of a block.
constexpr int mdd_ring(int bm) { return bm == 64 ? 4 : 3; }
#ifndef FFQ_MDD_WAVES
#define FFQ_MDD_WAVES 8  // waves of a block, side by side along N (A/B hook: 4 = 32 columns per wave, round 6's first LDS-DMA version)
#endif
constexpr int MDD_WAVES = FFQ_MDD_WAVES;
static_assert(MDD_WAVES == 4 || MDD_WAVES == 8, "4 x 32 or 8 x 16 columns");

template <int BKIND, bool GROUPED, bool OFFSET, int BM>
__global__ __launch_bounds__(64 * MDD_WAVES, 2) void wq_mid_dma_kernel(MidArgs a) {
  constexpr int MDD_RING = mdd_ring(BM);
  constexpr int W = MDD_WAVES;                 // waves side by side along N
  constexpr int MI = BM / 16, NJ = 8 / W;      // a wave: all BM rows x 16 NJ columns
  constexpr int A_BYTES = BM * 128, B_BYTES = MD_BN * 64, P_BYTES = GROUPED ? W * 256 : 0;
  constexpr int STAGE = A_BYTES + B_BYTES + P_BYTES;
  constexpr int APW = BM / 8 / W;              // 1 KiB activation pieces (8 rows) per wave and stage
  constexpr int BPW = 8 / W;                   // 1 KiB code pieces (16 rows) per wave and stage: the wave's own columns
  constexpr int NI = APW + BPW + (GROUPED ? 1 : 0);  // LDS-DMA instructions per wave and stage
  static_assert(NI * (MDD_RING - 2) <= 63, "vmcnt is a 6-bit counter");
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t r16 = lane & 15, g4 = lane >> 4;

  const int per_m = a.tiles_n * a.S;
  const int tm = (int)blockIdx.x / per_m, rest = (int)blockIdx.x - tm * per_m;
  const int tn_all = rest / a.S, slice = rest - tn_all * a.S;
  const int seg = tn_all >= a.seg_tile[2] ? 2 : tn_all >= a.seg_tile[1] ? 1 : 0;
  const uint8_t* const w_base = seg == 0 ? a.w[0] : seg == 1 ? a.w[1] : a.w[2];
  const float* const s_base = seg == 0 ? a.scale[0] : seg == 1 ? a.scale[1] : a.scale[2];
  const float* const o_base = seg == 0 ? a.offset[0] : seg == 1 ? a.offset[1] : a.offset[2];
  const int rows = seg == 0 ? a.seg_n[0] : seg == 1 ? a.seg_n[1] : a.seg_n[2];
  const int n0 = (tn_all - (seg == 0 ? 0 : seg == 1 ? a.seg_tile[1] : a.seg_tile[2])) * MD_BN;
  const int m0 = tm * BM;
  const int ksuper = a.K / 64;
  const int k0 = (int)((int64_t)slice * ksuper / a.S), k1 = (int)((int64_t)(slice + 1) * ksuper / a.S);
  const int nsteps = k1 - k0, k_last = k1 - 1;
  const uint32_t pack_shift = (uint32_t)a.pack_shift;
  const int steps_per_group = a.steps_per_group;

  // ---- LDS-DMA sources (per lane, fixed for the tile). A piece = 8 rows x 128 bytes: lane L -> row L / 8, 16-byte chunk L % 8 of the
  // LDS image, which holds logical chunk (L % 8) ^ ((row / 2) % 8). B piece = 16 rows x 64 bytes: lane L -> row L / 4, chunk L % 4
  // of the image = logical chunk (L % 4) ^ ((row / 4) % 4). Rows past an edge re-read the last row and are never stored.
  const uint8_t* a_src[APW];
#pragma unroll
  for (int c = 0; c < APW; ++c) {
    const int row = (wave * APW + c) * 8 + (lane >> 3);
    int m = m0 + row;
    m = m < a.M ? m : a.M - 1;
    a_src[c] = a.x + (size_t)m * (size_t)a.K * 2u + (size_t)((((uint32_t)lane & 7u) ^ (((uint32_t)row >> 1) & 7u)) << 4);
  }
  const uint32_t w_row_bytes = BKIND == WL_B_I8 ? (uint32_t)a.K : (uint32_t)a.K / 2u;
  const uint8_t* b_src[BPW];
#pragma unroll
  for (int j = 0; j < BPW; ++j) {
    const int row = (wave * BPW + j) * 16 + (lane >> 2);
    int n = n0 + row;
    n = n < rows ? n : rows - 1;
    b_src[j] = w_base + (size_t)n * w_row_bytes + (size_t)((((uint32_t)lane & 3u) ^ (((uint32_t)row >> 2) & 3u)) << 4);
  }
  // GROUPED: the wave's strip of the step's parameters — dword l of 64: scale of the wave's row l % (16 NJ) for (l / (16 NJ)) even, its
  // offset for odd (NJ = 1: the upper 32 lanes repeat the lower)
  [[maybe_unused]] const float* p_src = nullptr;
  float s_row[NJ], o_row[NJ];
  {
    int n = n0 + wave * 16 * NJ + (lane % (16 * NJ));
    n = n < rows ? n : rows - 1;
    const size_t p_row = a.per_row ? (size_t)n * (size_t)a.groups : 0;
    if constexpr (GROUPED) p_src = ((OFFSET && ((lane / (16 * NJ)) & 1)) ? o_base : s_base) + p_row;
  }
#pragma unroll
  for (int nj = 0; nj < NJ; ++nj) { s_row[nj] = 1.0f; o_row[nj] = 0.0f; }
  if constexpr (!GROUPED) {
#pragma unroll
    for (int nj = 0; nj < NJ; ++nj) {
      int n = n0 + wave * 16 * NJ + nj * 16 + (int)r16;
      n = n < rows ? n : rows - 1;
      const size_t p_row = a.per_row ? (size_t)n : 0;
      s_row[nj] = s_base[p_row];
      if constexpr (OFFSET) o_row[nj] = rne(o_base[p_row]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // nothing of these in the loop's count
#pragma unroll
    for (int nj = 0; nj < NJ; ++nj) asm volatile("" : "+v"(s_row[nj]), "+v"(o_row[nj]));
  }
  // first byte of super-step ks inside a row of codes (and, nibbles: which nibble holds them — one packing-block half spans >= 64 codes)
  auto code_byte0 = [&](int ks, uint32_t& nib) __attribute__((always_inline)) -> uint32_t {
    if constexpr (BKIND == WL_B_I8) {
      nib = 0;
      return (uint32_t)ks * 64u;
    } else {
      const uint32_t kk = (uint32_t)ks * 64u, lb = pack_shift;
      const uint32_t within = kk & ((1u << lb) - 1u);
      nib = (within >> (lb - 1u)) * 4u;
      return ((kk >> lb) << (lb - 1u)) + (within & ((1u << (lb - 1u)) - 1u));
    }
  };
  auto issue = [&](int ks, int slot) __attribute__((always_inline)) {
    ks = ks < k_last ? ks : k_last;
    uint8_t* base = lds + slot * STAGE;
#pragma unroll
    for (int c = 0; c < APW; ++c)
      __builtin_amdgcn_global_load_lds((wl_gbl_t*)(a_src[c] + (size_t)ks * 128u), (wl_lds_t*)(base + (wave * APW + c) * 1024), 16, 0, 0);
    uint32_t nib;
    const uint32_t byte0 = code_byte0(ks, nib);
#pragma unroll
    for (int j = 0; j < BPW; ++j)
      __builtin_amdgcn_global_load_lds((wl_gbl_t*)(b_src[j] + byte0), (wl_lds_t*)(base + A_BYTES + (wave * BPW + j) * 1024), 16, 0, 0);
    if constexpr (GROUPED)
      __builtin_amdgcn_global_load_lds((wl_gbl_t*)(p_src + ks / steps_per_group), (wl_lds_t*)(base + A_BYTES + B_BYTES + wave * 256), 4, 0, 0);
  };

  // ---- fragment addresses inside a stage
  uint32_t a_off[2], b_off[2];
#pragma unroll
  for (int kq = 0; kq < 2; ++kq) {
    a_off[kq] = r16 * 128u + (((kq * 4u + g4) ^ ((r16 >> 1) & 7u)) << 4);
    b_off[kq] = A_BYTES + ((uint32_t)wave * 16u * NJ + r16) * 64u + (((kq * 4u + g4) ^ (((r16 >> 2) & 3u) << 1)) << 3);
  }
  [[maybe_unused]] const uint32_t p_off = A_BYTES + B_BYTES + (uint32_t)wave * 256u + r16 * 4u;

  wl_v4f acc[MI][NJ];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int nj = 0; nj < NJ; ++nj) acc[mi][nj] = wl_v4f{0.0f, 0.0f, 0.0f, 0.0f};

  // Fragment reads are inline assembly with hand-written waits: with LDS-DMA requests in flight hipcc puts `s_waitcnt lgkmcnt(0)` in front
  // of the first use of ANY compiler-visible ds_read result (seen in the ISA of a first version: chunk 1's reads could not fly under chunk
  // 0's MFMAs). A wait names the registers it guards as operands, which keeps their uses behind it.
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds;
#define MDD_READ128(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF))
#define MDD_READ64(DST, ADDR, OFF) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF))
#define MDD_READ32(DST, ADDR, OFF) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF))
  auto compute = [&](int ks, int slot) __attribute__((always_inline)) {
    const uint32_t st = lds_base + (uint32_t)slot * STAGE;
    uint32_t nib;
    (void)code_byte0(ks, nib);
    float s[NJ], c[NJ];
    wl_v4i fa[2][MI];
    u32x2 codes[2][NJ];
    // (macros, not lambdas: clang does not capture a variable that a nested generic lambda names only as an asm operand)
#define MDD_READ_CHUNK(KQ)                                                                                       \
  do {                                                                                                           \
    const uint32_t b_addr = st + b_off[KQ], a_addr = st + a_off[KQ];                                             \
    MDD_READ64(codes[KQ][0], b_addr, 0);                                                                         \
    if constexpr (NJ == 2) MDD_READ64(codes[KQ][NJ - 1], b_addr, 1024);                                          \
    MDD_READ128(fa[KQ][0], a_addr, 0 * 2048);                                                                    \
    MDD_READ128(fa[KQ][1], a_addr, 1 * 2048);                                                                    \
    MDD_READ128(fa[KQ][2], a_addr, 2 * 2048);                                                                    \
    MDD_READ128(fa[KQ][3], a_addr, 3 * 2048);                                                                    \
    if constexpr (MI == 8) {                                                                                     \
      MDD_READ128(fa[KQ][MI - 4], a_addr, 4 * 2048);                                                             \
      MDD_READ128(fa[KQ][MI - 3], a_addr, 5 * 2048);                                                             \
      MDD_READ128(fa[KQ][MI - 2], a_addr, 6 * 2048);                                                             \
      MDD_READ128(fa[KQ][MI - 1], a_addr, 7 * 2048);                                                             \
    }                                                                                                            \
  } while (0)
    // everything read so far is back; every register of the chunk is re-defined behind the wait as far as the compiler is concerned (an
    // empty asm with the register as a read-write operand: volatile statements keep their order, so its uses stay behind the wait)
#define MDD_WAIT(KQ)                                                                       \
  do {                                                                                     \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                     \
    _Pragma("unroll") for (int mi_ = 0; mi_ < MI; ++mi_) asm volatile("" : "+v"(fa[KQ][mi_])); \
    _Pragma("unroll") for (int nj_ = 0; nj_ < NJ; ++nj_) asm volatile("" : "+v"(codes[KQ][nj_]), "+v"(s[nj_]), "+v"(c[nj_])); \
  } while (0)
    auto mfma_chunk = [&](auto kc) __attribute__((always_inline)) {
      constexpr int kq = decltype(kc)::value;
      wl_v4i fb[NJ];
#pragma unroll
      for (int nj = 0; nj < NJ; ++nj) {
        uint32_t w[2] = {codes[kq][nj].x, codes[kq][nj].y};
        float sv = s[nj], cv = OFFSET ? c[nj] : 0.0f;
        if constexpr (BKIND == WL_B_I4) {
          // nibble n = code + 8 -> (n ^ 8) << 4 in the byte's high half = 16 * code as a signed byte; (16 q + 16 o) * (s / 16) is (q + o) * s
          // with the same single rounding wherever s / 16 is exact; a tiny scale takes the codes themselves (ffq_wlinear.hip)
          const bool tiny = __builtin_fabsf(sv) < 0x1p-120f && sv != 0.0f;
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            uint32_t b = (((w[q] >> nib) << 4) & 0xF0F0F0F0u) ^ 0x80808080u;
            if (__builtin_expect(tiny, 0)) {
              const uint32_t b0 = (uint32_t)(((int32_t)(b << 24)) >> 28) & 0xFFu, b1 = (uint32_t)(((int32_t)(b << 16)) >> 28) & 0xFFu;
              const uint32_t b2 = (uint32_t)(((int32_t)(b << 8)) >> 28) & 0xFFu, b3 = (uint32_t)(((int32_t)b) >> 28) & 0xFFu;
              b = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
            }
            w[q] = b;
          }
          if (!tiny) { sv = sv * 0.0625f; cv = cv * 16.0f; }
        }
        uint32_t o[4];
        dequantize4<OFFSET>(w[0], sv, cv, o[0], o[1]);
        dequantize4<OFFSET>(w[1], sv, cv, o[2], o[3]);
        fb[nj] = wl_v4i{(int)o[0], (int)o[1], (int)o[2], (int)o[3]};
      }
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int n_ = 0; n_ < NJ; ++n_) {
          const int nj = (mi & 1) ? NJ - 1 - n_ : n_;
          acc[mi][nj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(wl_v8bf, fb[nj]), __builtin_bit_cast(wl_v8bf, fa[kq][mi]), acc[mi][nj], 0, 0, 0);
        }
    };
#pragma unroll
    for (int nj = 0; nj < NJ; ++nj) { s[nj] = s_row[nj]; c[nj] = o_row[nj]; }
    if constexpr (GROUPED) {  // the step's group parameters of this lane's two weight rows, from the stage's parameter strip
      const uint32_t p_addr = st + p_off;
      MDD_READ32(s[0], p_addr, 0);
      if constexpr (NJ == 2) MDD_READ32(s[NJ - 1], p_addr, 64);
      if constexpr (OFFSET) {
        MDD_READ32(c[0], p_addr, 64 * NJ);
        if constexpr (NJ == 2) MDD_READ32(c[NJ - 1], p_addr, 64 * NJ + 64);
      }
    }
    // (lgkmcnt is a 4-bit counter: a chunk's 10 reads + 4 parameter reads stay below 16 in flight)
    MDD_READ_CHUNK(0);
    MDD_WAIT(0);  // the one exposed LDS round trip of a step
    if constexpr (GROUPED && OFFSET) {
#pragma unroll
      for (int nj = 0; nj < NJ; ++nj) c[nj] = rne(c[nj]);
    }
    MDD_READ_CHUNK(1);
    __builtin_amdgcn_sched_barrier(0);
    mfma_chunk(std::integral_constant<int, 0>{});  // chunk 1's reads fly under these MFMAs and conversions
    __builtin_amdgcn_sched_barrier(0);
    MDD_WAIT(1);
    mfma_chunk(std::integral_constant<int, 1>{});
  };

  // ---- the K-loop
#pragma unroll
  for (int st = 0; st < MDD_RING - 1; ++st) issue(k0 + st, st);
  int slot = 0, slot_free = MDD_RING - 1;
  for (int step = 0; step < nsteps; ++step) {
    // this wave's pieces of stage `step` have landed (the RING - 2 younger stages may still fly); behind the barrier everybody's have,
    // and everybody is done reading the slot computed a step ago
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(NI * (MDD_RING - 2)) : "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    issue(k0 + step + MDD_RING - 1, slot_free);
    compute(k0 + step, slot);
    slot_free = slot;
    slot = slot + 1 == MDD_RING ? 0 : slot + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the trailing requests must not outlive the block's LDS
#undef MDD_WAIT
#undef MDD_READ_CHUNK
#undef MDD_READ128
#undef MDD_READ64
#undef MDD_READ32

  // ---- a wave's result: acc[mi][nj][t] = y[m0 + 16 mi + r16][n0 + 16 NJ wave + 16 nj + 4 g4 + t] (partial over this block's k slice)
  const int tile = tm * a.tiles_n + tn_all;
  if (a.S > 1) {
    constexpr size_t unit_bytes = (size_t)MI * NJ * 1024;
    uint8_t* const strip = reinterpret_cast<uint8_t*>(a.slabs) + ((size_t)tile * a.S * W + wave) * unit_bytes;
    const size_t slice_stride = (size_t)W * unit_bytes;
    {
      const auto mine = __builtin_amdgcn_make_buffer_rsrc(strip + (size_t)slice * slice_stride, 0, (int)unit_bytes, 0x00020000);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int nj = 0; nj < NJ; ++nj)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(wl_v4u, acc[mi][nj]), mine, ((mi * NJ + nj) * 64 + lane) * 16, 0, /*sc1*/ 16);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the write-through stores have left before the ticket is taken
    int t = 0;
    if (lane == 0) t = __hip_atomic_fetch_add(a.tickets + tile * W + wave, 1, FFQ_TICKET_ORDER, __HIP_MEMORY_SCOPE_AGENT);
    t = __builtin_amdgcn_readfirstlane(t);
    if (t != a.S - 1) return;  // somebody else finishes this strip
    if (lane == 0) __hip_atomic_store(a.tickets + tile * W + wave, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("" ::: "memory");
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int nj = 0; nj < NJ; ++nj) acc[mi][nj] = wl_v4f{0.0f, 0.0f, 0.0f, 0.0f};
    // slice order, whoever reduces: acc = ((0 + p0) + p1) + ... The partials of up to GROUP slices are requested before the first is
    // added (the loads of one slice are one L2 / Infinity-Cache round trip: four of them in series were ~3 us of a 18 us launch)
    constexpr int GROUP = MI <= 4 ? 4 : 2;
    for (int sl0 = 0; sl0 < a.S; sl0 += GROUP) {
      wl_v4u got[GROUP][MI][NJ];
#pragma unroll
      for (int g = 0; g < GROUP; ++g) {
        const int sl = sl0 + g < a.S ? sl0 + g : a.S - 1;  // (past the last slice: a re-read that is not added)
        const auto peer = __builtin_amdgcn_make_buffer_rsrc(strip + (size_t)sl * slice_stride, 0, (int)unit_bytes, 0x00020000);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int nj = 0; nj < NJ; ++nj) got[g][mi][nj] = __builtin_amdgcn_raw_buffer_load_b128(peer, ((mi * NJ + nj) * 64 + lane) * 16, 0, /*sc1*/ 16);
      }
#pragma unroll
      for (int g = 0; g < GROUP; ++g) {
        if (sl0 + g >= a.S) break;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int nj = 0; nj < NJ; ++nj) {
            const wl_v4f p = __builtin_bit_cast(wl_v4f, got[g][mi][nj]);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[mi][nj][e] = acc[mi][nj][e] + p[e];
          }
      }
    }
  }
  // ---- epilogue: bias, cast, 4 consecutive columns per lane and row
  void* const out = seg == 0 ? a.out[0] : seg == 1 ? a.out[1] : a.out[2];
  const bool rows_by_4 = (rows & 3) == 0;
#pragma unroll
  for (int nj = 0; nj < NJ; ++nj) {
    const int ncol = n0 + wave * 16 * NJ + 16 * nj + 4 * (int)g4;
    if (ncol >= rows) continue;
    float b4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (a.bias) {
#pragma unroll
      for (int e = 0; e < 4; ++e) b4[e] = ncol + e < rows ? (float)load_any(a.bias, a.bias_dt, ncol + e) : 0.0f;
    }
    const bool whole = ncol + 4 <= rows && rows_by_4;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int m = m0 + 16 * mi + (int)r16;
      if (m >= a.M) continue;
      float y[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) y[e] = a.bias ? acc[mi][nj][e] + b4[e] : acc[mi][nj][e];
      const size_t at = (size_t)m * (size_t)rows + (size_t)ncol;
      if (a.out_dt == FFQ_BF16) {
        bf16_t* o = static_cast<bf16_t*>(out) + at;
        if (whole) {
          u32x2 pk;
          pk.x = pack2<bf16_t>(y[0], y[1]);
          pk.y = pack2<bf16_t>(y[2], y[3]);
          *reinterpret_cast<u32x2*>(o) = pk;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (ncol + e < rows) o[e] = from_f32<bf16_t>(y[e]);
        }
      } else {
        float* o = static_cast<float*>(out) + at;
        if (whole) {
          *reinterpret_cast<wl_v4f*>(o) = wl_v4f{y[0], y[1], y[2], y[3]};
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (ncol + e < rows) o[e] = y[e];
        }
      }
    }
  }
}

// ---- the plan ---------------------------------------------------------------------------------------------------------------------
// Everything below is a function of (M, N, K) alone — not of the container, the packing block or the group size: every storage form
// of one weight takes the same tiles, the same K slices and the same summation order.
// 64-row tiles up to 256 rows and for contractions up to 8192 deep, 128-row tiles for the long contractions of a few hundred rows
// (down_proj at 512 rows: 78 us against 91; q/o at 512 rows the other way round: 31.6 against 35.0 — profiles/r06_wq_mid_sweep_v4.txt)
static int md_bm(int64_t M, int64_t K) { return (M <= 256 || K <= 8192) ? 64 : 128; }
static int64_t md_tiles_m(int64_t M, int64_t K) { return (M + md_bm(M, K) - 1) / md_bm(M, K); }
static int64_t md_tiles_n(int64_t N) { return (N + MD_BN - 1) / MD_BN; }

bool wq_mid_shape_ok(int64_t M, int64_t K) { return M >= 1 && M <= WQ_MID_MAX_M && K % 64 == 0 && K >= 128; }

// where the 128-column tiles are the preferred form: everything they cover up to 256 rows; beyond that the narrow projections — with
// >= 8192 output columns and > 256 rows the 256-row tiles move less through the L2 (gate/up at 512 rows: 71 against 74 us)
bool wq_mid_prefers(int64_t M, int64_t N, int64_t K) { return wq_mid_shape_ok(M, K) && (M <= 256 || N < 8192); }

// K slices: one unit per CU where the tiles alone do not fill the chip, at most 4 (the last arriver of a quadrant reads S partials: from
// 8 slices on the exchange costs more than the idle CUs it fills — k/v at 128 rows 16.3 us with 4 slices, 18.0 with 8), each slice at
// least 8 super-steps long
int wq_mid_split(int64_t M, int64_t N, int64_t K) {
  if (!wq_mid_shape_ok(M, K)) return 1;
  const int64_t tiles = md_tiles_m(M, K) * md_tiles_n(N), ksuper = K / 64;
  int64_t S = (wq_cus() + tiles - 1) / tiles;
  if (S > 4) S = 4;
  if (S > ksuper / 8) S = ksuper / 8;
  return S < 1 ? 1 : (int)S;
}

// ticket words: one per (tile, wave), for the widest launch of this (M, N, K): three matrices, each rounded up to whole tiles
int64_t wq_mid_tickets(int64_t M, int64_t N, int64_t K) {
  if (!wq_mid_shape_ok(M, K)) return 0;
  return md_tiles_m(M, K) * (md_tiles_n(N) + 2) * 8;  // (tile, wave): up to 8 waves per tile
}

size_t wq_mid_slab_bytes(int64_t M, int64_t N, int64_t K, int64_t split) {
  if (!wq_mid_shape_ok(M, K) || split <= 1) return 0;
  return (size_t)(md_tiles_m(M, K) * (md_tiles_n(N) + 2)) * (size_t)split * (size_t)md_bm(M, K) * MD_BN * 4u;
}

bool wq_mid_applies(const WLinearArgs& a) {
  if (generic_kernels_forced()) return false;  // tests: the 256-row-tile kernel on the same operands (ffq_force_generic_kernels)
  if (!wq_mid_shape_ok(a.M, a.K)) return false;
  for (int i = 0; i < 2; ++i)
    if (a.seg_n[i + 1] > 0 && a.seg_n[i] % MD_BN != 0) return false;  // every matrix but the last: whole column tiles
  return true;
}

// the LDS-DMA form covers int8 containers and nibbles whose packing-block half holds a whole 64-code super-step
static bool md_takes_dma(int w_dt, int pack_shift) {
#ifdef FFQ_MID_NO_DMA  // A/B builds (tools/build_variant.sh)
  return false;
#else
  if (mid_register_form_forced()) return false;  // tests: the register-staged kernel on the same operands (ffq_force_generic_kernels bit 2)
  return w_dt != FFQ_U8 || pack_shift >= 7;
#endif
}

template <int BKIND, bool GROUPED, bool OFFSET>
static void md_launch_dma(const MidArgs& m, int bm, unsigned grid, hipStream_t stream) {
#define FFQ_MDD(BM_)                                                                                                         \
  do {                                                                                                                       \
    static uint64_t attr_set = 0;                                                                                            \
    const int lds_bytes = mdd_ring(BM_) * (BM_ * 128 + MD_BN * 64 + (GROUPED ? MDD_WAVES * 256 : 0));                                        \
    ensure_dynamic_lds(&attr_set, reinterpret_cast<const void*>(&wq_mid_dma_kernel<BKIND, GROUPED, OFFSET, BM_>), lds_bytes); \
    wq_mid_dma_kernel<BKIND, GROUPED, OFFSET, BM_><<<grid, 64 * MDD_WAVES, lds_bytes, stream>>>(m);                                      \
  } while (0)
  if (bm == 64) FFQ_MDD(64); else FFQ_MDD(128);
#undef FFQ_MDD
}

template <int BKIND, bool GROUPED, bool OFFSET>
static void md_launch(const MidArgs& m, int bm, unsigned grid, hipStream_t stream) {
#define FFQ_MD(BM_)                                                                                                      \
  do {                                                                                                                   \
    static uint64_t attr_set = 0;                                                                                        \
    const int lds_bytes = 2 * (BM_ * 128 + MD_BN * 128);                                                                 \
    ensure_dynamic_lds(&attr_set, reinterpret_cast<const void*>(&wq_mid_kernel<BKIND, GROUPED, OFFSET, BM_>), lds_bytes); \
    wq_mid_kernel<BKIND, GROUPED, OFFSET, BM_><<<grid, 256, lds_bytes, stream>>>(m);                                      \
  } while (0)
  if (bm == 64) FFQ_MD(64); else FFQ_MD(128);
#undef FFQ_MD
}

int wq_mid_launch(const WLinearArgs& a, int w_dt, int64_t group, int64_t split, void* workspace, size_t workspace_bytes, int32_t* tickets,
                  hipStream_t stream) {
  MidArgs m;
  m.x = a.x;
  m.w[0] = a.w; m.scale[0] = a.w_scale; m.offset[0] = a.w_offset; m.out[0] = a.out;
  int64_t N = 0, tiles_n = 0;
  for (int i = 0; i < 3; ++i) {
    m.seg_n[i] = a.seg_n[i];
    m.seg_tile[i] = i == 0 ? 0 : (a.seg_n[i] > 0 ? (int)tiles_n : INT32_MAX);
    if (i > 0) { m.w[i] = a.seg_w[i - 1]; m.scale[i] = a.seg_scale[i - 1]; m.offset[i] = a.seg_offset[i - 1]; m.out[i] = a.seg_out[i - 1]; }
    N += a.seg_n[i];
    tiles_n += md_tiles_n(a.seg_n[i]);
  }
  m.bias = a.bias; m.bias_dt = a.bias_dt; m.out_dt = a.out_dt;
  m.M = a.M; m.K = a.K;
  m.tiles_m = (int)md_tiles_m(a.M, a.K); m.tiles_n = (int)tiles_n;
  m.groups = a.groups; m.steps_per_group = (int)(group / 64); m.per_row = a.per_row; m.pack_shift = a.pack_shift;
  const int64_t ksuper = a.K / 64;
  int64_t S = split > 0 ? split : wq_mid_split(a.M, N, a.K);
  if (S > ksuper) {
    if (split > 0) return fail(FFQ_ERR_ARG, "weight-only linear (128-column tiles): split %lld exceeds the %lld super-steps of 64 along K", (long long)split, (long long)ksuper);
    S = ksuper;
  }
  const int bm = md_bm(a.M, a.K);
  const int64_t tiles = (int64_t)m.tiles_m * tiles_n;
  const size_t slab = S > 1 ? (size_t)tiles * (size_t)S * (size_t)bm * MD_BN * 4u : 0;
  if (S > 1 && (!tickets || !workspace || workspace_bytes < slab || !aligned16(workspace))) {
    if (split > 1) return fail(FFQ_ERR_ARG, "weight-only linear (128-column tiles): split %lld needs %zu bytes of workspace and a ticket buffer", (long long)S, slab);
    S = 1;  // the plan is a preference: without scratch every block walks the whole K range
  }
  m.S = (int)S;
  m.slabs = S > 1 ? static_cast<float*>(workspace) : nullptr;
  m.tickets = S > 1 ? tickets : nullptr;
  const unsigned grid = (unsigned)(tiles * S);
  const bool grouped = a.groups > 1, offset = a.w_offset != nullptr;
#define FFQ_MD_T(BK)                                                                                                                     \
  do {                                                                                                                                   \
    if (grouped) { if (offset) md_launch<BK, true, true>(m, bm, grid, stream); else md_launch<BK, true, false>(m, bm, grid, stream); }     \
    else { if (offset) md_launch<BK, false, true>(m, bm, grid, stream); else md_launch<BK, false, false>(m, bm, grid, stream); }           \
  } while (0)
  if (md_takes_dma(w_dt, a.pack_shift)) {
#define FFQ_MDD_T(BK)                                                                                                                            \
  do {                                                                                                                                           \
    if (grouped) { if (offset) md_launch_dma<BK, true, true>(m, bm, grid, stream); else md_launch_dma<BK, true, false>(m, bm, grid, stream); }     \
    else { if (offset) md_launch_dma<BK, false, true>(m, bm, grid, stream); else md_launch_dma<BK, false, false>(m, bm, grid, stream); }           \
  } while (0)
    if (w_dt == FFQ_U8) FFQ_MDD_T(WL_B_I4); else FFQ_MDD_T(WL_B_I8);
#undef FFQ_MDD_T
    return check_launch("wq_mid_dma_kernel");
  }
  if (w_dt == FFQ_U8) FFQ_MD_T(WL_B_I4); else FFQ_MD_T(WL_B_I8);
#undef FFQ_MD_T
  return check_launch("wq_mid_kernel");
}

}  // namespace ffq

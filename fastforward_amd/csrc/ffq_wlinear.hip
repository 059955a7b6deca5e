// ffq_wlinear.hip — the weight-only quantized linear: bf16 activations x integer weight codes on the bf16 matrix cores.
//
// Replaces the branch of fallback.linear that a QuantizedTensor WEIGHT and a plain (non-quantized) INPUT take:
// src/fastforward/_gen/fallback.py:77-112 with strict_quantization off (:86-100): the reference dequantizes the weight
// codes into a bf16 tensor in HBM (A2: (q + round(o)) * s, rounded to the data dtype; 1 B read + 2 B written per
// element and 2 B read again by the GEMM, every forward) and calls F.linear. Here the codes are the B operand of the
// GEMM as they are: they travel HBM -> L2 -> LDS as int8 (1 B/elem, once) and are dequantized in registers on the way into
// v_mfma_f32_32x32x16_bf16, with EXACTLY A2's arithmetic ((float(q) + round(o)) * s in fp32, one RNE rounding to bf16) —
// the matrix the MFMA multiplies is bit for bit the reference's dequantized weight; only the fp32 summation order of the
// contraction differs. Granularities: one (scale, offset) per tensor, per output channel (PerChannel(0)), or per group of
// G input channels of an output channel (PerBlock(block_dims=1, block_sizes=G, per_channel_dims=0), G a multiple of 64:
// BASELINE config 4's group-128), parameters in the row order of tiles_to_rows ([N, K / G] row-major).
//
// Tile 256(M) x 256(N) x 64(K), 8 wavefronts (2 x 4), each owning 128 x 64 of the output = 4 x 2 MFMA tiles (128 fp32
// accumulators per lane). Operands go global -> LDS by LDS-DMA (global_load_lds_dwordx4): activations as whole 128-byte
// lines (8 rows per instruction), weight codes as 64-byte row pieces; the bank swizzle sits on the per-lane SOURCE address
// and, identically, on the ds_read address. Three LDS stages of 48 KiB: the pieces of K-step kt + 2 are issued during
// K-step kt and waited for with a counted s_waitcnt one step later, so an L2 miss has more than a whole K-step to land.
// K-loop = the ping-pong template of ffq_linear.hip: a K-step is four phases of {LOAD segment | raw s_barrier | 8 MFMAs
// under s_setprio | raw s_barrier}; waves 4-7 run one interval behind waves 0-3, so on every SIMD one wave feeds the
// matrix pipe while its partner reads fragments AND converts the next 16 weight codes per lane to bf16 (VALU in the
// shadow of the partner's MFMAs). Contraction-order trick: a lane reads 16 consecutive code bytes (two MFMA k-steps)
// and the matching 2 x 16 bytes of bf16 activations; both operands see the same permutation of k, which a dot product
// does not notice — no transposes, 16-byte LDS reads only.
#include "ffq_common.h"
#include "ffq_vec.h"

#include <type_traits>

#ifndef WL_X
#define WL_X 8  // schedule selector (tools/wq_variants.sh builds the others): 8 = conversion inside the clusters, the shipped one
#endif

namespace ffq {

typedef int wl_v4i __attribute__((ext_vector_type(4)));
typedef float wl_v16f __attribute__((ext_vector_type(16)));
typedef __bf16 wl_v8bf __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void wl_lds_t;
typedef __attribute__((address_space(1))) const void wl_gbl_t;

constexpr int WL_BM = 256, WL_BN = 256, WL_BK = 64, WL_STAGES = 3;
constexpr int WL_A_BYTES = WL_BM * WL_BK * 2;              // 32 KiB of bf16 activations per stage
constexpr int WL_B_BYTES = WL_BN * WL_BK;                  // 16 KiB of int8 codes per stage
constexpr int WL_STAGE_BYTES = WL_A_BYTES + WL_B_BYTES;    // 48 KiB
constexpr int WL_PARAM_BYTES = 2 * WL_BN * 4;              // grouped mode: 256 scales + 256 rounded-later offsets per stage
constexpr int WL_GROUP_M = 8;

struct WLinearArgs {
  const uint16_t* x;      // [M, K] bf16
  const int8_t* wq;       // [N, K] codes, int8 container
  const float* w_scale;   // [N * groups] (or [1])
  const float* w_offset;  // same shape, or NULL
  const void* bias; int bias_dt;
  void* out; int out_dt;  // bf16 or f32
  int M, N, K;
  int groups;             // parameters per output channel along K (1 = per channel / per tensor)
  int steps_per_group;    // K-steps of 64 that share one group
  int per_row;            // 0: one parameter pair for the whole tensor
  int tiles_m, tiles_n;
};

// 4 codes (one dword) -> 4 bf16 (two dwords) of (float(q) + ro) * s
struct wl_pair { int lo, hi; };
template <bool OFFSET>
__device__ __forceinline__ wl_pair dequantize4(uint32_t w, float s, float ro) {
#if WL_X == 6 || WL_X == 9  // integer -> float by byte permutation: [0x4B 00 00 u] = 2^23 + u with u = q + 128, then ONE exact add
  const uint32_t u = w ^ 0x80808080u;
  const float c = ro - 8388736.0f;  // -(2^23 + 128) + ro: integers far below 2^24, every sum below is exact
  const float f0 = __builtin_bit_cast(float, __builtin_amdgcn_perm(0x4B000000u, u, 0x070C0C00u)) + c;
  const float f1 = __builtin_bit_cast(float, __builtin_amdgcn_perm(0x4B000000u, u, 0x070C0C01u)) + c;
  const float f2 = __builtin_bit_cast(float, __builtin_amdgcn_perm(0x4B000000u, u, 0x070C0C02u)) + c;
  const float f3 = __builtin_bit_cast(float, __builtin_amdgcn_perm(0x4B000000u, u, 0x070C0C03u)) + c;
#else
  float f0 = (float)(int)(int8_t)(w), f1 = (float)(int)(int8_t)(w >> 8), f2 = (float)(int)(int8_t)(w >> 16), f3 = (float)(int)(int8_t)(w >> 24);
  if constexpr (OFFSET) { f0 = f0 + ro; f1 = f1 + ro; f2 = f2 + ro; f3 = f3 + ro; }
#endif
  wl_pair r;
  r.lo = (int)pack2<bf16_t>(f0 * s, f1 * s);
  r.hi = (int)pack2<bf16_t>(f2 * s, f3 * s);
  return r;
}

// 8 weight codes (two dwords) -> 8 bf16 of A2's value (float(q) + ro) * s, as the 4 dwords of an MFMA operand
template <bool OFFSET>
__device__ __forceinline__ wl_v4i dequantize8(uint32_t lo, uint32_t hi, float s, float ro) {
#if WL_X == 2  // ablation: no conversion (wrong results)
  wl_v4i raw4; raw4[0] = (int)lo; raw4[1] = (int)hi; raw4[2] = (int)lo; raw4[3] = (int)hi;
  return raw4;
#endif
#if WL_X == 6 || WL_X >= 8
  wl_v4i o4;
  const wl_pair a4 = dequantize4<OFFSET>(lo, s, ro), b4 = dequantize4<OFFSET>(hi, s, ro);
  o4[0] = a4.lo; o4[1] = a4.hi; o4[2] = b4.lo; o4[3] = b4.hi;
  return o4;
#endif
  float f[8];
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    f[b] = (float)(int)(int8_t)(lo >> (8 * b));
    f[4 + b] = (float)(int)(int8_t)(hi >> (8 * b));
  }
  wl_v4i out;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    float a = f[2 * p], b = f[2 * p + 1];
    if constexpr (OFFSET) { a = a + ro; b = b + ro; }
    out[p] = (int)pack2<bf16_t>(a * s, b * s);
  }
  return out;
}

template <bool GROUPED, bool OFFSET, typename TOut>
__global__ __launch_bounds__(512, 2) void wq_bf16_gemm256_kernel(WLinearArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  uint8_t* const params_lds = lds + WL_STAGES * WL_STAGE_BYTES;  // grouped mode only

  // XCD-aware grouped tile order (as ffq_linear.hip): an XCD owns a contiguous range of tiles, visited 8 row-tiles deep
  const uint32_t nblk = gridDim.x;
  const uint32_t xcd = blockIdx.x & 7u, slot_in_xcd = blockIdx.x >> 3;
  const uint32_t q8 = nblk >> 3, r8 = nblk & 7u;
  const uint32_t tile_id = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot_in_xcd;
  const uint32_t per_group = WL_GROUP_M * (uint32_t)a.tiles_n;
  const uint32_t group = tile_id / per_group, in_group = tile_id - group * per_group;
  const uint32_t group_rows = min((uint32_t)WL_GROUP_M, (uint32_t)a.tiles_m - group * WL_GROUP_M);
  const int m0 = (int)(group * WL_GROUP_M + in_group % group_rows) * WL_BM;
  const int n0 = (int)(in_group / group_rows) * WL_BN;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;  // wm is also the ping-pong group
  const int frag_row = lane & 31, frag_g = lane >> 5;

  // ---- parameters of this lane's two weight rows (not grouped: registers for the whole tile, loaded before any DMA flies)
  float sc[2] = {1.0f, 1.0f}, ro[2] = {0.0f, 0.0f};
  if constexpr (!GROUPED) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int n = n0 + wn * 64 + j * 32 + frag_row;
      n = n < a.N ? n : a.N - 1;
      sc[j] = a.w_scale[a.per_row ? n : 0];
      if constexpr (OFFSET) ro[j] = rne(a.w_offset[a.per_row ? n : 0]);
    }
    // consume the loads HERE: hipcc waits vmcnt(0) for an ordinary load's first use, and inside the K-loop that wait
    // would drain the LDS-DMA pipeline on every iteration
    asm volatile("" ::"v"(sc[0]), "v"(sc[1]), "v"(ro[0]), "v"(ro[1]));
  }

  // ---- DMA sources. A: wave w copies the 8-row pieces {4w .. 4w+3} (rows 32w .. 32w+31, 128 B each); lane l lands at
  // 16-B slot l of the piece: row l / 8, physical slot l % 8, logical slot = physical ^ ((row >> 1) & 7).
  // B: wave w copies the 16-row pieces {2w, 2w+1} (rows 32w .. 32w+31, 64 B each): row l / 4, physical slot l % 4,
  // logical = physical ^ ((row >> 2) & 3).
  const uint8_t* a_src[4];
  const uint8_t* b_src[2];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int row = wave * 32 + c * 8 + (lane >> 3);
    const int slot = (lane & 7) ^ ((row >> 1) & 7);
    int gm = m0 + row;
    gm = gm < a.M ? gm : a.M - 1;  // rows past the edge re-read the last row and are never stored
    a_src[c] = reinterpret_cast<const uint8_t*>(a.x) + ((size_t)gm * a.K) * 2 + slot * 16;
  }
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int row = wave * 32 + c * 16 + (lane >> 2);
    const int slot = (lane & 3) ^ ((row >> 2) & 3);
    int gn = n0 + row;
    gn = gn < a.N ? gn : a.N - 1;
    b_src[c] = reinterpret_cast<const uint8_t*>(a.wq) + (size_t)gn * a.K + slot * 16;
  }
  // grouped mode: wave w < 4 fetches the scales of rows 64w .. 64w+63 of the K-step's group, wave w >= 4 the offsets
  // (the scales once more when there is no offset: every wave issues the same number of pieces, one constant vmcnt)
  const float* p_src = nullptr;
  if constexpr (GROUPED) {
    int gn = n0 + (wave & 3) * 64 + lane;
    gn = gn < a.N ? gn : a.N - 1;
    p_src = ((wave >= 4 && a.w_offset) ? a.w_offset : a.w_scale) + (size_t)gn * a.groups;
  }
  const int ksteps = a.K / WL_BK;
  auto stage_of = [](int kt) { return kt % WL_STAGES; };
  // pieces of K-step kt (steps past the end re-load the last one into a stage nobody reads any more)
#if WL_X == 10
  // buffer form of the LDS-DMA: the tile's base in a wave-uniform descriptor (SGPRs), one 32-bit per-lane offset, the
  // K-step in the scalar offset — no 64-bit per-lane address arithmetic, half the address registers read per piece.
  // Rows past the matrix edge fall outside the descriptor's range and read as zeros.
  const size_t a_left = (size_t)(a.M - m0) * a.K * 2, b_left = (size_t)(a.N - n0) * a.K;
  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(reinterpret_cast<const uint8_t*>(a.x) + (size_t)m0 * a.K * 2), 0, (int)(a_left < 0x7FFFFFFFu ? a_left : 0x7FFFFFFFu), 0x00020000);
  const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(reinterpret_cast<const uint8_t*>(a.wq) + (size_t)n0 * a.K), 0, (int)(b_left < 0x7FFFFFFFu ? b_left : 0x7FFFFFFFu), 0x00020000);
  uint32_t a_voff[4], b_voff[2];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int row = wave * 32 + c * 8 + (lane >> 3);
    a_voff[c] = (uint32_t)row * (uint32_t)a.K * 2u + (uint32_t)(((lane & 7) ^ ((row >> 1) & 7)) * 16);
  }
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int row = wave * 32 + c * 16 + (lane >> 2);
    b_voff[c] = (uint32_t)row * (uint32_t)a.K + (uint32_t)(((lane & 3) ^ ((row >> 2) & 3)) * 16);
  }
#endif
  auto issue_a = [&](int kt, int c0) {
    const int st = stage_of(kt);
    kt = kt < ksteps ? kt : ksteps - 1;
    uint8_t* base = lds + st * WL_STAGE_BYTES;
#pragma unroll
    for (int c = c0; c < c0 + 2; ++c)
#if WL_X == 10
      __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (wl_lds_t*)(base + (wave * 4 + c) * 1024), 16, a_voff[c], kt * (WL_BK * 2), 0, 0);
#else
      __builtin_amdgcn_global_load_lds((wl_gbl_t*)(a_src[c] + (size_t)kt * (WL_BK * 2)), (wl_lds_t*)(base + (wave * 4 + c) * 1024), 16, 0, 0);
#endif
  };
  auto issue_b = [&](int kt) {
    const int st = stage_of(kt);
    kt = kt < ksteps ? kt : ksteps - 1;
    uint8_t* base = lds + st * WL_STAGE_BYTES + WL_A_BYTES;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#if WL_X == 10
      __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rsrc, (wl_lds_t*)(base + (wave * 2 + c) * 1024), 16, b_voff[c], kt * WL_BK, 0, 0);
#else
      __builtin_amdgcn_global_load_lds((wl_gbl_t*)(b_src[c] + (size_t)kt * WL_BK), (wl_lds_t*)(base + (wave * 2 + c) * 1024), 16, 0, 0);
#endif
    if constexpr (GROUPED) {
      const int g = kt / a.steps_per_group;
      __builtin_amdgcn_global_load_lds((wl_gbl_t*)(p_src + g), (wl_lds_t*)(params_lds + st * WL_PARAM_BYTES + wave * 256), 4, 0, 0);
    }
  };

  // ---- fragment addresses inside a stage. MFMA (j, t) of a K-step (j = 32-wide half, t = 16-wide quarter of it):
  // lane (r, g) contracts k = 32 j + 16 g + 8 t + (0..7): activations from logical slot 4 j + 2 g + t of its row, codes from
  // the low / high half of logical slot 2 j + g. Adding 32 rows leaves both swizzles unchanged: one base per (j, t).
  uint32_t a_off[4], b_off[2];
  {
    const uint32_t row = wm * 128 + frag_row;
#pragma unroll
    for (int v = 0; v < 4; ++v) a_off[v] = row * 128 + ((((uint32_t)(4 * (v >> 1) + 2 * frag_g + (v & 1))) ^ ((row >> 1) & 7u)) << 4);
    const uint32_t brow = wn * 64 + frag_row;
#pragma unroll
    for (int j = 0; j < 2; ++j) b_off[j] = WL_A_BYTES + brow * 64 + ((((uint32_t)(2 * j + frag_g)) ^ ((brow >> 2) & 3u)) << 4);
  }

  wl_v16f acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

  wl_v4i fa[4], raw[2], nxt[2], fb[2];
  // LOAD segment of phase p = 2 j + t. Everything it converts is already in registers: the 16 code bytes of a pair of
  // phases are fetched one pair AHEAD (odd phases issue the reads of the next pair — the last phase of a K-step those of
  // the NEXT step's first pair, plus that step's parameters in grouped mode), so the LDS latency hides behind the
  // conversion of the current phase instead of preceding it.
  auto fetch_codes = [&](const uint8_t* st, int j) {
#pragma unroll
    for (int f = 0; f < 2; ++f) nxt[f] = *reinterpret_cast<const wl_v4i*>(st + b_off[j] + f * (32 * 64));
  };
  auto fetch_params = [&](const uint8_t* pst) {
    if constexpr (GROUPED) {
      const float* ps = reinterpret_cast<const float*>(pst);
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        sc[f] = ps[wn * 64 + f * 32 + frag_row];
        if constexpr (OFFSET) ro[f] = rne(ps[256 + wn * 64 + f * 32 + frag_row]);
      }
    }
  };
  auto load_segment = [&](const uint8_t* st, const uint8_t* st_next, const uint8_t* pst_next, auto phase) {
    constexpr int p = decltype(phase)::value;
#if WL_X == 5  // the CONVERTING wave is the prioritised one
    __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const wl_v4i*>(st + a_off[p] + i * (32 * 128));
    if constexpr (p == 0 || p == 2) {
      raw[0] = nxt[0]; raw[1] = nxt[1];
#pragma unroll
      for (int f = 0; f < 2; ++f) fb[f] = dequantize8<OFFSET>((uint32_t)raw[f].x, (uint32_t)raw[f].y, sc[f], ro[f]);
    } else {
      if constexpr (p == 1) fetch_codes(st, 1);
      else fetch_codes(st_next, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int f = 0; f < 2; ++f) fb[f] = dequantize8<OFFSET>((uint32_t)raw[f].z, (uint32_t)raw[f].w, sc[f], ro[f]);
      if constexpr (p == 3) {
        __builtin_amdgcn_sched_barrier(0);
        fetch_params(pst_next);  // first used by the next K-step's conversions
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#if WL_X == 5
    __builtin_amdgcn_s_setprio(0);
#endif
  };
  auto cluster = [&](auto dma) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#if WL_X != 4 && WL_X != 5
    __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(wl_v8bf, fb[j]), __builtin_bit_cast(wl_v8bf, fa[i]), acc[i][j], 0, 0, 0);
#if WL_X == 3  // ablation: no LDS-DMA inside the loop (wrong results)
      (void)dma;
#else
      if (i == 1) { __builtin_amdgcn_sched_barrier(0); dma(); __builtin_amdgcn_sched_barrier(0); }
#endif
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(0);
  };

#if WL_X >= 8
  // Conversion inside the clusters: the computing wave converts the codes of its OWN next phase in the shadow of its MFMAs
  // (four VALU operations behind each MFMA), the load segment only reads.
  wl_v4i fbn[2];
  auto load_segment2 = [&](const uint8_t* st, const uint8_t* st_next, const uint8_t* pst_next, auto phase) {
    constexpr int p = decltype(phase)::value;
    fb[0] = fbn[0]; fb[1] = fbn[1];
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const wl_v4i*>(st + a_off[p] + i * (32 * 128));
    if constexpr (p == 1) fetch_codes(st, 1);
    if constexpr (p == 0 || p == 2) { raw[0] = nxt[0]; raw[1] = nxt[1]; }
    if constexpr (p == 3) { fetch_codes(st_next, 0); fetch_params(pst_next); }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto cluster2 = [&](auto dma, auto phase) {
    constexpr int p = decltype(phase)::value;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(wl_v8bf, fb[j]), __builtin_bit_cast(wl_v8bf, fa[i]), acc[i][j], 0, 0, 0);
      {  // quarter i of the next phase's 16 codes: fragment i / 2, dword (i & 1) of its 8 bytes
        constexpr bool from_next = (p & 1) == 1;  // odd phases convert the low half of the pair fetched ahead
        const wl_v4i& src = from_next ? nxt[i >> 1] : raw[i >> 1];
        const uint32_t w = (uint32_t)(from_next ? ((i & 1) ? src.y : src.x) : ((i & 1) ? src.w : src.z));
        const wl_pair c4 = dequantize4<OFFSET>(w, sc[i >> 1], ro[i >> 1]);
        fbn[i >> 1][2 * (i & 1)] = c4.lo;
        fbn[i >> 1][2 * (i & 1) + 1] = c4.hi;
      }
      if (i == 1) { __builtin_amdgcn_sched_barrier(0); dma(); __builtin_amdgcn_sched_barrier(0); }
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(0);
  };
#endif

  // ---- prologue: K-steps 0 and 1 entirely; step 0 has landed when only step 1's pieces are outstanding
  issue_a(0, 0); issue_a(0, 2); issue_b(0);
  issue_a(1, 0); issue_a(1, 2); issue_b(1);
  if constexpr (GROUPED) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  fetch_codes(lds, 0);
  fetch_params(params_lds);
#if WL_X >= 8
  raw[0] = nxt[0]; raw[1] = nxt[1];
#pragma unroll
  for (int f = 0; f < 2; ++f) fbn[f] = dequantize8<OFFSET>((uint32_t)raw[f].x, (uint32_t)raw[f].y, sc[f], ro[f]);
#endif
  if (wm == 1) __builtin_amdgcn_s_barrier();  // the upper group runs one interval behind

  // ---- K-loop. During step kt the clusters issue the pieces of step kt + 2 into the stage step kt - 1 occupied (its
  // last reads were retired by the slower group one barrier before the faster group's first cluster: WAR). The load
  // segment of phase 2 waits until only THIS step's four pieces are in flight — step kt + 1 has landed — one barrier
  // before phase 3's look-ahead reads it (RAW: wait -> barrier -> read; the slower group's wait precedes the barrier that
  // opens the faster group's phase 3).
  for (int kt = 0; kt < ksteps; ++kt) {
    const uint8_t* st = lds + stage_of(kt) * WL_STAGE_BYTES;
    const uint8_t* st_next = lds + stage_of(kt + 1) * WL_STAGE_BYTES;
    const uint8_t* pst_next = params_lds + stage_of(kt + 1) * WL_PARAM_BYTES;
#if WL_X == 11  // conversion in the clusters, pieces at the end of the (now light) load segments
    load_segment2(st, st_next, pst_next, std::integral_constant<int, 0>{});
    issue_a(kt + 2, 0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    cluster2([] {}, std::integral_constant<int, 0>{});
    __builtin_amdgcn_s_barrier();
    load_segment2(st, st_next, pst_next, std::integral_constant<int, 1>{});
    issue_a(kt + 2, 2);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    cluster2([] {}, std::integral_constant<int, 1>{});
    __builtin_amdgcn_s_barrier();
    load_segment2(st, st_next, pst_next, std::integral_constant<int, 2>{});
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    issue_b(kt + 2);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    cluster2([] {}, std::integral_constant<int, 2>{});
    __builtin_amdgcn_s_barrier();
    load_segment2(st, st_next, pst_next, std::integral_constant<int, 3>{});
    __builtin_amdgcn_s_barrier();
    cluster2([] {}, std::integral_constant<int, 3>{});
    __builtin_amdgcn_s_barrier();
    continue;
#elif WL_X >= 8
    load_segment2(st, st_next, pst_next, std::integral_constant<int, 0>{});
    __builtin_amdgcn_s_barrier();
    cluster2([&] { issue_a(kt + 2, 0); }, std::integral_constant<int, 0>{});
    __builtin_amdgcn_s_barrier();
    load_segment2(st, st_next, pst_next, std::integral_constant<int, 1>{});
    __builtin_amdgcn_s_barrier();
    cluster2([&] { issue_a(kt + 2, 2); }, std::integral_constant<int, 1>{});
    __builtin_amdgcn_s_barrier();
    load_segment2(st, st_next, pst_next, std::integral_constant<int, 2>{});
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    cluster2([&] { issue_b(kt + 2); }, std::integral_constant<int, 2>{});
    __builtin_amdgcn_s_barrier();
    load_segment2(st, st_next, pst_next, std::integral_constant<int, 3>{});
    __builtin_amdgcn_s_barrier();
    cluster2([] {}, std::integral_constant<int, 3>{});
    __builtin_amdgcn_s_barrier();
    continue;
#elif WL_X == 1  // pieces issued at the end of the load segments instead of inside the clusters
    load_segment(st, st_next, pst_next, std::integral_constant<int, 0>{});
    issue_a(kt + 2, 0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    cluster([] {});
    __builtin_amdgcn_s_barrier();
    load_segment(st, st_next, pst_next, std::integral_constant<int, 1>{});
    issue_a(kt + 2, 2);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    cluster([] {});
    __builtin_amdgcn_s_barrier();
    load_segment(st, st_next, pst_next, std::integral_constant<int, 2>{});
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    issue_b(kt + 2);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    cluster([] {});
    __builtin_amdgcn_s_barrier();
#else
    load_segment(st, st_next, pst_next, std::integral_constant<int, 0>{});
    __builtin_amdgcn_s_barrier();
    cluster([&] { issue_a(kt + 2, 0); });
    __builtin_amdgcn_s_barrier();
    load_segment(st, st_next, pst_next, std::integral_constant<int, 1>{});
    __builtin_amdgcn_s_barrier();
    cluster([&] { issue_a(kt + 2, 2); });
    __builtin_amdgcn_s_barrier();
    load_segment(st, st_next, pst_next, std::integral_constant<int, 2>{});
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    cluster([&] { issue_b(kt + 2); });
    __builtin_amdgcn_s_barrier();
#endif
    load_segment(st, st_next, pst_next, std::integral_constant<int, 3>{});
    __builtin_amdgcn_s_barrier();
    cluster([] {});
    __builtin_amdgcn_s_barrier();
  }
  if (wm == 0) __builtin_amdgcn_s_barrier();  // same number of barriers for both groups
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the trailing dummy pieces must not land in the epilogue's LDS
  __syncthreads();

  // ---- epilogue. The weight fragment is the MFMA's first operand, so with the 32x32 C/D layout lane l holds, for each
  // (i, j, q): C[m = i*32 + (l & 31)][n = j*32 + 8 q + 4 (l >> 5) + (0..3)]: four consecutive output columns of one row.
  // One 32-row slab per wave at a time goes through LDS (144-B pitch) and leaves as 16-byte stores of whole 128-B lines.
  TOut* out = static_cast<TOut*>(a.out);
  constexpr int PITCH = 64 * (int)sizeof(TOut) + 16;
  constexpr int WAVE_BYTES = 32 * PITCH + 256;             // one 32-row slab + the wave's 64 bias values
  uint8_t* slab = lds + wave * WAVE_BYTES;
  float* bias_lds = reinterpret_cast<float*>(slab + 32 * PITCH);
  const int wave_n0 = n0 + wn * 64, wave_m0 = m0 + wm * 128;
  const bool full = wave_n0 + 64 <= a.N && (a.N * (int)sizeof(TOut)) % 16 == 0;
  const bool has_bias = a.bias != nullptr;
  {
    const int n = wave_n0 + lane;
    bias_lds[lane] = has_bias ? (float)load_any(a.bias, a.bias_dt, n < a.N ? n : a.N - 1) : 0.0f;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private: only this wave's own writes
#pragma unroll 1
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int nb = j * 32 + 8 * q + 4 * frag_g;
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias_lds + nb);
        float y[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          // runtime i: pick the accumulator tile without dynamic register indexing
          const float v = i == 0 ? acc[0][j][4 * q + t] : i == 1 ? acc[1][j][4 * q + t] : i == 2 ? acc[2][j][4 * q + t] : acc[3][j][4 * q + t];
          y[t] = has_bias ? v + b4[t] : v;
        }
        if constexpr (sizeof(TOut) == 2) {
          u32x2 pk;
          pk.x = pack2<TOut>(y[0], y[1]);
          pk.y = pack2<TOut>(y[2], y[3]);
          *reinterpret_cast<u32x2*>(slab + frag_row * PITCH + nb * 2) = pk;
        } else {
          u32x4 pk;
          pk.x = __builtin_bit_cast(uint32_t, y[0]); pk.y = __builtin_bit_cast(uint32_t, y[1]);
          pk.z = __builtin_bit_cast(uint32_t, y[2]); pk.w = __builtin_bit_cast(uint32_t, y[3]);
          *reinterpret_cast<u32x4*>(slab + frag_row * PITCH + nb * 4) = pk;
        }
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (full) {
      constexpr int SEGS = 64 * (int)sizeof(TOut) / 16;   // 16-byte segments per row: 8 (bf16) or 16 (f32)
#pragma unroll
      for (int t = 0; t < 32 * SEGS / 64; ++t) {
        const int c = lane + 64 * t;
        const int row = c / SEGS, seg = c % SEGS;
        const int mm = wave_m0 + i * 32 + row;
        const u32x4 v = *reinterpret_cast<const u32x4*>(slab + row * PITCH + seg * 16);
        // non-temporal, as in ffq_linear.hip: the output must not push the operand panels out of L2
        if (mm < a.M) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(reinterpret_cast<uint8_t*>(out) + ((size_t)mm * a.N + wave_n0) * sizeof(TOut) + seg * 16));
      }
    } else {  // ragged right edge / unaligned rows: element stores (correctness path)
      for (int c = lane; c < 32 * 64; c += 64) {
        const int row = c >> 6, col = c & 63;
        const int mm = wave_m0 + i * 32 + row;
        if (mm < a.M && wave_n0 + col < a.N)
          out[(size_t)mm * a.N + wave_n0 + col] = *reinterpret_cast<const TOut*>(slab + row * PITCH + col * sizeof(TOut));
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the slab is re-written by the next i
  }
}

}  // namespace ffq

using namespace ffq;

// 1 if ffq_linear_wq covers the problem with the MFMA kernel, 0 if the caller has to dequantize and use a float GEMM.
extern "C" int ffq_linear_wq_supported(int x_dt, int w_dt, int out_dt, int64_t M, int64_t N, int64_t K, int64_t group) {
  if (x_dt != FFQ_BF16 || w_dt != FFQ_I8 || !(out_dt == FFQ_BF16 || out_dt == FFQ_F32)) return 0;
  if (M <= 0 || N <= 0 || K < 2 * WL_BK || K % WL_BK != 0) return 0;
  if (group <= 0 || K % group != 0) return 0;
  if (group != K && group % WL_BK != 0) return 0;
  if (M > INT32_MAX || N > INT32_MAX || K > INT32_MAX) return 0;
  return 1;
}

extern "C" int ffq_linear_wq(const void* x, int x_dt, const void* w_codes, int w_dt, const float* w_scale, const float* w_offset,
                             int64_t scale_numel, int64_t group, const void* bias, int bias_dt, void* out, int out_dt, int64_t M,
                             int64_t N, int64_t K, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (M < 0 || N < 0 || K < 0) return fail(FFQ_ERR_ARG, "negative extent");
  if (M == 0 || N == 0) return FFQ_OK;
  if (!x || !w_codes || !w_scale || !out) return fail(FFQ_ERR_ARG, "NULL buffer");
  if (!ffq_linear_wq_supported(x_dt, w_dt, out_dt, M, N, K, group))
    return fail(FFQ_ERR_DTYPE, "weight-only linear: needs bf16 activations, int8-container codes, bf16 / f32 output, K %% 64 == 0, K >= 128 and groups of a multiple of 64 input channels");
  if (!aligned16(x) || !aligned16(w_codes) || !aligned16(out)) return fail(FFQ_ERR_DTYPE, "weight-only linear needs 16-byte aligned buffers");
  if (bias && !dt_valid(bias_dt)) return fail(FFQ_ERR_ARG, "bad bias dtype");
  const int64_t groups = K / group;
  if (!(scale_numel == 1 || scale_numel == N * groups))
    return fail(FFQ_ERR_PARAM_NUMEL, "weight-only linear: %lld parameters for %lld x %lld tiles", (long long)scale_numel, (long long)N, (long long)groups);
  if (scale_numel == 1 && groups != 1) return fail(FFQ_ERR_PARAM_NUMEL, "one parameter pair needs group == K");

  WLinearArgs a;
  a.x = static_cast<const uint16_t*>(x);
  a.wq = static_cast<const int8_t*>(w_codes);
  a.w_scale = w_scale; a.w_offset = w_offset;
  a.bias = bias; a.bias_dt = bias_dt;
  a.out = out; a.out_dt = out_dt;
  a.M = (int)M; a.N = (int)N; a.K = (int)K;
  a.groups = (int)groups;
  a.steps_per_group = (int)(group / WL_BK);
  a.per_row = scale_numel != 1;
  a.tiles_m = (int)((M + WL_BM - 1) / WL_BM);
  a.tiles_n = (int)((N + WL_BN - 1) / WL_BN);
  const unsigned grid = (unsigned)(a.tiles_m * a.tiles_n);
  const size_t lds_bytes = (size_t)WL_STAGES * WL_STAGE_BYTES + (size_t)WL_STAGES * WL_PARAM_BYTES;
  const bool grouped = groups > 1;
  const bool offset = w_offset != nullptr;
#define FFQ_WL_LAUNCH(G, O, T)                                                                                              \
  do {                                                                                                                      \
    static uint64_t attr_set = 0;                                                                                           \
    ensure_dynamic_lds(&attr_set, reinterpret_cast<const void*>(&wq_bf16_gemm256_kernel<G, O, T>), (int)lds_bytes);         \
    wq_bf16_gemm256_kernel<G, O, T><<<grid, 512, lds_bytes, s>>>(a);                                                        \
  } while (0)
#define FFQ_WL_DISPATCH(T)                                                                                                  \
  do {                                                                                                                      \
    if (grouped) { if (offset) FFQ_WL_LAUNCH(true, true, T); else FFQ_WL_LAUNCH(true, false, T); }                          \
    else { if (offset) FFQ_WL_LAUNCH(false, true, T); else FFQ_WL_LAUNCH(false, false, T); }                                \
  } while (0)
  if (out_dt == FFQ_BF16) FFQ_WL_DISPATCH(bf16_t);
  else FFQ_WL_DISPATCH(float);
#undef FFQ_WL_DISPATCH
#undef FFQ_WL_LAUNCH
  return check_launch("wq_bf16_gemm256_kernel");
}

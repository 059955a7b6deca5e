// ffq_wq.h — what the three weight-only GEMM translation units share (ffq_wlinear.hip: 256 x 256 tiles; ffq_wmid.hip: 128-column tiles
// for up to 512 token rows; ffq_wskinny.hip: up to 128 token rows): vector types, the launch arguments, A2 of four codes, the plan queries.
#pragma once
#include "ffq_common.h"
#include "ffq_vec.h"

namespace ffq {

typedef int wl_v4i __attribute__((ext_vector_type(4)));
typedef unsigned int wl_v4u __attribute__((ext_vector_type(4)));
typedef float wl_v4f __attribute__((ext_vector_type(4)));
typedef __bf16 wl_v8bf __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void wl_lds_t;
typedef __attribute__((address_space(1))) const void wl_gbl_t;

constexpr int WL_BM = 256, WL_BN = 256, WL_BK = 64;
constexpr int WL_IMAGE = 256 * 128;            // one operand image: 256 rows x 128 bytes
constexpr int WL_SLOT = 2 * WL_IMAGE;          // A image then B image: 64 KiB
constexpr int WL_GROUP_M = 8;
constexpr size_t WL_UNIT_SLAB = (size_t)32 * 8 * 64 * 16;  // split-K: one unit's partial accumulators, [32 pieces][8][64 lanes] x 16 B = 256 KiB
constexpr int64_t WL_TWO_PASS_MIN_TOKENS = 4096;  // from this many tokens on, A2 as its own pass + the bf16-image GEMM

enum { WL_B_BF16 = 0, WL_B_I8 = 1, WL_B_I4 = 2 };

struct WLinearArgs {
  const uint8_t* x;       // [M, K] bf16
  const uint8_t* w;       // WL_B_BF16: [N, K] bf16; WL_B_I8: [N, K] int8 codes; WL_B_I4: [N, K / 2] packed nibbles
  const float* w_scale;   // [N * groups] (or [1])
  const float* w_offset;  // same shape, or NULL
  const void* bias; int bias_dt;
  void* out; int out_dt;  // bf16 or f32
  int M, N, K;
  int groups;             // parameters per output channel along K (1 = per channel / per tensor)
  int steps_per_group;    // super-steps of 64 that share one group
  int per_row;            // 0: one parameter pair for the whole tensor
  int pack_shift;         // WL_B_I4: log2(packing block)
  int tiles_m, tiles_n, group_m;
  int group_cols;  // 0: groups of `group_m` row tiles x all column tiles (the weight is re-streamed per group); 1: groups of `group_m` column tiles x all row tiles (the activations are)
  // MLP mode (ffq_mlp_gate_up_wq): `w` / `w_scale` / `w_offset` describe gate_proj, these up_proj; N = rows of each = output columns
  const uint8_t* w2;
  const float* w_scale2;
  const float* w_offset2;
  // split-K (fewer tiles than CUs): a work unit is (tile, slice of the K range), `split` slices per tile, all units of the launch
  // resident at once; the units of a tile exchange partial accumulators through `slabs` and each finishes a share of the tile
  // (kernel epilogue). `tickets`: two counters per tile (arrived, left), zero on entry and on exit
  // several weight matrices side by side along N in ONE launch (q / k / v of an attention block: the same activations, three
  // weight tensors, three outputs): column tiles [0, seg_tile[0]) belong to matrix 0, [seg_tile[0], seg_tile[1]) to matrix 1, the
  // rest to matrix 2; every matrix but the last has a multiple of 256 rows. `w` / `w_scale` / `w_offset` / `out` / `N` describe
  // matrix 0 (N = ALL columns for the tile walk); 1 and 2 below. Plain mode only.
  int seg_tile[2];
  const uint8_t* seg_w[2]; const float* seg_scale[2]; const float* seg_offset[2]; void* seg_out[2];
  int seg_n[3];  // rows (output columns) of each matrix
  int split;
  int full_tiles;  // tiles [0, full_tiles) of the walk order are whole units (a multiple of the grid: every block gets the same count);
                   // the TAIL tiles [full_tiles, total) are cut into `split` slices each, at most one such unit per block, every block's last
  float* slabs;
  int* tickets;
  int abandon_test;  // test hook: odd slices give up their wait immediately (splitk_abandon_forced)
};

// 4 codes in the bytes of `w` (signed bytes; for nibbles: 16 * code, see the header) -> 4 bf16 of (float(b) + c) * s
template <bool OFFSET>
__device__ __forceinline__ void dequantize4(uint32_t w, float s, float c, uint32_t& lo, uint32_t& hi) {
  float f0 = (float)(int)(int8_t)(w), f1 = (float)(int)(int8_t)(w >> 8), f2 = (float)(int)(int8_t)(w >> 16), f3 = (float)(int)(int8_t)(w >> 24);
  if constexpr (OFFSET) { f0 = f0 + c; f1 = f1 + c; f2 = f2 + c; f3 = f3 + c; }
  lo = pack2<bf16_t>(f0 * s, f1 * s);
  hi = pack2<bf16_t>(f2 * s, f3 * s);
}

// ---- the skinny form (ffq_wskinny.hip): M <= 128 rows, plain launches -------------------------------------------------------------
bool wq_skinny_applies(const WLinearArgs& a, int64_t pack_block);
int wq_skinny_split(int64_t M, int64_t N, int64_t K);                       // K slices across blocks the library's plan takes
int64_t wq_skinny_tickets(int64_t M, int64_t N, int64_t K);                 // int32 counters (zero before, zero after)
size_t wq_skinny_slab_bytes(int64_t M, int64_t N, int64_t K, int64_t split);
int wq_skinny_launch(const WLinearArgs& a, int w_dt, int64_t pack_block, int64_t group, int64_t split, void* workspace, size_t workspace_bytes,
                     int32_t* tickets, hipStream_t stream);
int wq_cus();

// ---- 128-column tiles (ffq_wmid.hip): M <= 512 rows, plain launches, every storage form -------------------------------------------
#ifndef FFQ_MID_MIN_M
#define FFQ_MID_MIN_M 17     // token rows from which the 128-column tiles are preferred to the skinny form (A/B hook: tools/build_variant.sh)
#endif
constexpr int64_t WQ_MID_MAX_M = 512;      // beyond: the 256-row tiles of ffq_wlinear.hip

bool wq_mid_shape_ok(int64_t M, int64_t K);
bool wq_mid_prefers(int64_t M, int64_t N, int64_t K);  // ... and are the preferred form for (M, N = all output columns of the launch, K)
bool wq_mid_applies(const WLinearArgs& a);
int wq_mid_split(int64_t M, int64_t N, int64_t K);
int64_t wq_mid_tickets(int64_t M, int64_t N, int64_t K);
size_t wq_mid_slab_bytes(int64_t M, int64_t N, int64_t K, int64_t split);
int wq_mid_launch(const WLinearArgs& a, int w_dt, int64_t group, int64_t split, void* workspace, size_t workspace_bytes, int32_t* tickets,
                  hipStream_t stream);

}  // namespace ffq

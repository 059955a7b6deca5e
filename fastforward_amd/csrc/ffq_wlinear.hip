// ffq_wlinear.hip — the weight-only quantized linear: bf16 activations x integer weight codes on the bf16 matrix cores.
//
// Replaces the branch of fallback.linear that a QuantizedTensor WEIGHT and a plain (non-quantized) INPUT take:
// src/fastforward/_gen/fallback.py:77-112 with strict_quantization off (:86-100): the reference dequantizes the weight
// codes into a bf16 tensor in HBM (A2: (q + round(o)) * s, rounded to the data dtype; 1 B read + 2 B written per
// element and 2 B read again by the GEMM, every forward) and calls F.linear. Here the codes are the GEMM's B operand as they
// are stored — int8 containers (1 B/elem) or PACKED 4-bit nibbles in the GGUF Q4_0 order of the reference's exporter
// (export/stages/gguf/_packing.py:44-53; 0.5 B/elem: BASELINE config 4's "sub-byte pack/unpack kernel path") — and the
// matrix the MFMA multiplies is bit for bit the reference's dequantized weight: EXACTLY A2's arithmetic
// ((float(q) + round(o)) * s in fp32, one RNE rounding to bf16); only the fp32 summation order of the contraction differs.
// Granularities: one (scale, offset) per tensor, per output channel (PerChannel(0)), or per group of G input channels of
// an output channel (PerBlock(block_dims=1, block_sizes=G, per_channel_dims=0), G a multiple of 64: config 4's group-128),
// parameters in the row order of tiles_to_rows ([N, K / G] row-major).
//
// Round 3 design (round 2's kernel converted the codes per WAVE on the LDS->register path: every code was converted by the two
// waves that share a column range, 128 VALU instructions per wave and 64-deep K-step beside 32 MFMAs, and ran at 1.15 PFLOP/s
// against 1.36 for its own skeleton without the conversion):
//   * the skeleton is the persistent int8 kernel of ffq_linear.hip restated for bf16: 256 x 256 output tile per block,
//     8 wavefronts (2 x 4) of 128 x 64 = 8 x 4 accumulator tiles of v_mfma_f32_16x16x32_bf16, a super-step = 64 k = 128 bytes
//     of every activation row (whole cache lines by LDS-DMA, saddr form, swizzle on the source address), two 64 KiB LDS
//     slots, ping-pong wave groups (LOAD segment | raw s_barrier | 16 MFMAs under s_setprio | raw s_barrier, waves 4-7 one
//     interval behind), one block per CU walking its tiles XCD-aware with the K-loop running across tile boundaries;
//   * the codes are converted ONCE PER BLOCK: every lane fetches 16 code bytes of two weight rows per super-step straight
//     into registers (inline-asm loads in the saddr form, counted by hand: hipcc waits vmcnt(0) for an ordinary load next
//     to an LDS-DMA stream), two super-steps ahead of their use; one super-step ahead it converts them — 2 VALU per code,
//     64 per lane and super-step beside 64 MFMAs, placed inside the first two MFMA clusters — and writes the bf16 image of
//     the B tile into the LDS slot (4 ds_write_b128 per lane, conflict-free under the image's swizzle slot ^= row & 7);
//     all eight waves then read B fragments from that image exactly as they read A fragments;
//   * packed nibbles: a lane's 16 bytes hold its 16 codes in one nibble position; it moves them into the high half of each
//     byte ((n ^ 8) << 4 = 16 * code as a signed byte), converts with the byte-select form of v_cvt_f32_i32 and multiplies
//     by s / 16 — power-of-two scalings commute with the one rounding of (q + o) * s (scales below 2^-120, where s / 16
//     would lose bits, take the codes themselves on that lane);
//   * large token counts may instead hand the kernel a bf16 image of the weight made by A2 in a separate pass
//     (ffq_linear_wq's `workspace`): the conversion then costs nothing per row tile; the B operand is that tensor by LDS-DMA.
#include "ffq_common.h"
#include "ffq_vec.h"
#include "ffq_silu.h"

#include <stdlib.h>

#include <type_traits>

#ifndef FFQ_WL_LOOP_MODE
#define FFQ_WL_LOOP_MODE 0  // K-loop of the bf16-image form: 0 = ping-pong wave groups (8 raw barriers per super-step), 1 = free-running waves
#endif                      //   with fragments fetched one phase ahead and ONE barrier per super-step (experiment, tools/build_variant.sh)
#ifndef FFQ_WL_CLUSTER_MODE
#define FFQ_WL_CLUSTER_MODE 0  // how a conversion cluster mixes its VALU work with its MFMAs: 0 = the compiler's choice, 1 = three / 2 = two VALU behind each MFMA
#endif

#include "ffq_wq.h"

namespace ffq {

// MLP: the B tile holds 128 gate_proj rows and the same 128 up_proj rows, interleaved in runs of 32 so that a wave's column tiles
// nj = 0, 1 are gate and nj + 2 up of the SAME 32 output columns; the epilogue writes bf16(silu(bf16(gate))) * bf16(up) — exactly
// what ffq_silu_mul_quantize makes of the two plain launches' bf16 outputs (reference quantized_llama/mlp.py:30-40) — as one
// 256 x 128 bf16 tile: the two projections never visit HBM.
template <int BKIND, bool GROUPED, bool OFFSET, typename TOut, bool MLP = false>
__global__ __launch_bounds__(512, 2) void wq_gemm256_kernel(WLinearArgs a, int total_tiles) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  constexpr bool CODES = BKIND != WL_B_BF16;
  constexpr int BN_OUT = MLP ? 128 : WL_BN;  // output columns (= rows of each weight matrix) per tile

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;  // wm is also the ping-pong group

  // tile walk of ffq_linear.hip: XCD x (= blockIdx % 8) owns a contiguous range of the grouped tile order, its blocks walk it
  // round-robin, i.e. the order in which a non-persistent launch would dispatch them
  const uint32_t nblk = gridDim.x;
  const uint32_t xcd = blockIdx.x & 7u, j_in_xcd = blockIdx.x >> 3;
  const uint32_t blocks_in_xcd = (nblk >> 3) + (xcd < (nblk & 7u) ? 1u : 0u);
  // work units: the whole tiles, then the units of the tail tiles slice-major ([slice][tail tile in walk order], so that the
  // blocks of an XCD work on ONE K range of neighbouring tiles); both lists are dealt to the XCDs as contiguous ranges
  const uint32_t full_tiles = (uint32_t)a.full_tiles, tail_tiles = (uint32_t)total_tiles - full_tiles;
  auto xcd_range = [&](uint32_t n, uint32_t& first, uint32_t& count) {
    const uint32_t tq = n >> 3, tr = n & 7u;
    first = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
    count = tq + (xcd < tr ? 1u : 0u);
  };
  uint32_t full_first, full_count, tail_first, tail_count;
  xcd_range(full_tiles, full_first, full_count);
  xcd_range(tail_tiles * (uint32_t)a.split, tail_first, tail_count);
  const int n_full = j_in_xcd < full_count ? (int)((full_count - j_in_xcd + blocks_in_xcd - 1) / blocks_in_xcd) : 0;
  const int n_tail = j_in_xcd < tail_count ? (int)((tail_count - j_in_xcd + blocks_in_xcd - 1) / blocks_in_xcd) : 0;  // <= 1 when split > 1
  const int my_tiles = n_full + n_tail;
  if (my_tiles == 0) return;
  const int ksuper = a.K / WL_BK;  // >= 2 * split (checked by the launcher)
  [[maybe_unused]] uint16_t* const silu_table = reinterpret_cast<uint16_t*>(lds + 2 * WL_SLOT);  // MLP: behind the two slots
  if constexpr (MLP) silu_table_fill(silu_table, (uint32_t)tid, 512u);  // published by the first tile's barriers
  // unit `it` of this block: tile origin, super-steps [k0, k1) of the contraction, tail-tile number (ticket / slab index; -1 for a
  // whole tile) and slice
  auto tile_origin = [&](int it, int& tm0, int& tn0, int& k0, int& k1, int& tile_no, int& slice, int& seg) {
    it = it < my_tiles ? it : my_tiles - 1;  // streams running past the block's last unit re-read it (never used)
    uint32_t tile_id;
    if (it < n_full) {
      tile_id = full_first + j_in_xcd + (uint32_t)it * blocks_in_xcd;
      tile_no = -1; slice = 0; k0 = 0; k1 = ksuper;
    } else {
      const uint32_t unit_id = tail_first + j_in_xcd + (uint32_t)(it - n_full) * blocks_in_xcd;
      const uint32_t sl = unit_id / tail_tiles, tail_no = unit_id - sl * tail_tiles;
      tile_id = full_tiles + tail_no;
      tile_no = a.split > 1 ? (int)tail_no : -1; slice = (int)sl;
      k0 = (int)(sl * (uint32_t)ksuper / (uint32_t)a.split);
      k1 = (int)((sl + 1u) * (uint32_t)ksuper / (uint32_t)a.split);
    }
    const uint32_t gm = (uint32_t)a.group_m;
    // the operand that a group re-reads in full should be the SMALLER one: it is what has to stay in the 256 MiB Infinity
    // Cache between groups while the other streams through once (launcher's choice, `group_cols`)
    const uint32_t across = a.group_cols ? (uint32_t)a.tiles_m : (uint32_t)a.tiles_n;   // tiles of a group along the re-read operand
    const uint32_t along = a.group_cols ? (uint32_t)a.tiles_n : (uint32_t)a.tiles_m;    // the dimension groups are cut from
    const uint32_t per_group = gm * across;
    const uint32_t group = tile_id / per_group, in_group = tile_id - group * per_group;
    const uint32_t group_size = min(gm, along - group * gm);
    const uint32_t inner = group * gm + in_group % group_size, outer = in_group / group_size;
    tm0 = (int)(a.group_cols ? outer : inner) * WL_BM;
    int tn = (int)(a.group_cols ? inner : outer);
    seg = 0;
    if constexpr (!MLP) {  // which weight matrix the column tile belongs to, and the tile's origin inside it
      if (tn >= a.seg_tile[1]) { seg = 2; tn -= a.seg_tile[1]; }
      else if (tn >= a.seg_tile[0]) { seg = 1; tn -= a.seg_tile[0]; }
    }
    tn0 = tn * BN_OUT;
  };
  // the matrix a tile works on (wave-uniform selects: no dynamic indexing of the kernel arguments)
  auto seg_codes = [&](int seg) { return seg == 0 ? a.w : seg == 1 ? a.seg_w[0] : a.seg_w[1]; };
  auto seg_scales = [&](int seg) { return seg == 0 ? a.w_scale : seg == 1 ? a.seg_scale[0] : a.seg_scale[1]; };
  auto seg_offsets = [&](int seg) { return seg == 0 ? a.w_offset : seg == 1 ? a.seg_offset[0] : a.seg_offset[1]; };
  auto seg_rows = [&](int seg) { return seg == 0 ? a.seg_n[0] : seg == 1 ? a.seg_n[1] : a.seg_n[2]; };
  // first byte of row `row0` of a matrix with `row_bytes` per row, kept in SGPRs (see ffq_linear.hip::row_base)
  auto row_base = [&](const uint8_t* base, int row0, uint32_t row_bytes) {
    const uint64_t off = (uint64_t)(uint32_t)row0 * (uint64_t)row_bytes;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)off), hi = __builtin_amdgcn_readfirstlane((uint32_t)(off >> 32));
    return base + (((uint64_t)hi << 32) | lo);
  };

  // ---- operand streams. Element e = (tile, super-step): the block computes e, LDS-DMA fetches the images of e + 1, the
  // conversion handles the codes of e + 1, the code loads run at e + 2.
  const uint32_t x_row_bytes = (uint32_t)a.K * 2u;
  const uint32_t w_row_bytes = BKIND == WL_B_BF16 ? (uint32_t)a.K * 2u : BKIND == WL_B_I8 ? (uint32_t)a.K : (uint32_t)a.K / 2u;
  const int d_row = lane >> 3;
  uint32_t a_voff[4];
  const uint8_t* a_base = a.x;
  [[maybe_unused]] uint32_t b_voff[4];          // WL_B_BF16: the B image by LDS-DMA, as A
  [[maybe_unused]] const uint8_t* b_base[4] = {a.w, a.w, a.w, a.w};  // per piece (MLP: gate or up matrix)
  auto set_image_sources = [&](int tm0, int tn0, int seg) {
    a_base = row_base(a.x, tm0, x_row_bytes);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int row = (wave * 4 + c) * 8 + d_row;
      const int d_slot = (lane & 7) ^ ((row >> 1) & 7);
      const int ra = tm0 + row < a.M ? row : a.M - 1 - tm0;  // rows past the edge re-read the last row and are never stored
      a_voff[c] = (uint32_t)ra * x_row_bytes + d_slot * 16;
      if constexpr (!CODES) {
        if constexpr (MLP) {
          // tile rows [64 q, 64 q + 32) are gate rows tn0 + 32 q + (0..31), the next 32 the same rows of up; row & 32 is the
          // same for all lanes of a piece (8 rows per piece). N % 128 == 0: always inside
          const int rb = (row >> 6) * 32 + (row & 31);
          b_base[c] = row_base((((wave * 4 + c) * 8) & 32) ? a.w2 : a.w, tn0, w_row_bytes);
          b_voff[c] = (uint32_t)rb * w_row_bytes + d_slot * 16;
        } else {
          const int rows = seg_rows(seg);
          const int rb = tn0 + row < rows ? row : rows - 1 - tn0;
          b_base[c] = row_base(seg_codes(seg), tn0, w_row_bytes);
          b_voff[c] = (uint32_t)rb * w_row_bytes + d_slot * 16;
        }
      }
    }
  };
  auto issue_a = [&](int ks, int slot, int c0) {
    uint8_t* base = lds + slot * WL_SLOT;
#pragma unroll
    for (int c = c0; c < c0 + 2; ++c) {
      asm volatile("" : "+v"(a_voff[c]));  // keeps the saddr form in every unrolled body (ffq_linear.hip)
      __builtin_amdgcn_global_load_lds((wl_gbl_t*)((a_base + ks * 128) + a_voff[c]), (wl_lds_t*)(base + (wave * 4 + c) * 1024), 16, 0, 0);
    }
  };
  auto issue_b = [&](int ks, int slot, int c0) {
    if constexpr (!CODES) {
      uint8_t* base = lds + slot * WL_SLOT + WL_IMAGE;
#pragma unroll
      for (int c = c0; c < c0 + 2; ++c) {
        asm volatile("" : "+v"(b_voff[c]));
        __builtin_amdgcn_global_load_lds((wl_gbl_t*)((b_base[c] + ks * 128) + b_voff[c]), (wl_lds_t*)(base + (wave * 4 + c) * 1024), 16, 0, 0);
      }
    }
  };

  // ---- the code stream (CODES). Instruction j of a lane covers row 32 w + 16 j + lane / 4 of the tile, piece p = lane % 4:
  // the 16 codes [16 p, 16 p + 16) of that row's 64-code slab — four lanes fetch one row's 64 contiguous bytes.
  [[maybe_unused]] const uint32_t c_piece = (uint32_t)lane & 3u;
  [[maybe_unused]] uint32_t c_voff[2];            // byte offset of the lane's row inside the tile's rows
  [[maybe_unused]] uint32_t p_voff[2];            // byte offset of the lane's row inside the tile's parameter rows
  [[maybe_unused]] const uint8_t* c_base = a.w;   // first code byte of the tile's first row
  [[maybe_unused]] const float* ps_base = a.w_scale;
  [[maybe_unused]] const float* po_base = a.w_offset;
  [[maybe_unused]] u32x4 raw[2];                  // the 16 code bytes of each row, in flight / waiting for their conversion
  [[maybe_unused]] float sc[2] = {1.0f, 1.0f}, ro[2] = {0.0f, 0.0f};
  [[maybe_unused]] uint32_t nib_shift = 0;        // WL_B_I4: 0 = the low nibbles of the bytes, 4 = the high ones (stream state)
  [[maybe_unused]] int grp = 0, grp_phase = 0;    // GROUPED: parameter group of the code stream's super-step, and the step inside it
  auto set_code_sources = [&](int tn0, int seg) {
    if constexpr (CODES) {
      // MLP: a wave's 32 tile rows are all gate rows (even waves) or all up rows (odd waves) of the tile's 128 output columns
      const bool second = MLP && (wave & 1);
      c_base = row_base(second ? a.w2 : seg_codes(seg), tn0, w_row_bytes);
      const uint32_t param_row = a.per_row ? (uint32_t)a.groups : 0u;
      ps_base = reinterpret_cast<const float*>(row_base(reinterpret_cast<const uint8_t*>(second ? a.w_scale2 : seg_scales(seg)), tn0, param_row * 4u));
      if constexpr (OFFSET) po_base = reinterpret_cast<const float*>(row_base(reinterpret_cast<const uint8_t*>(second ? a.w_offset2 : seg_offsets(seg)), tn0, param_row * 4u));
      const int rows = MLP ? a.N : seg_rows(seg);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int trow = wave * 32 + j * 16 + (lane >> 2);
        const int row = MLP ? (trow >> 6) * 32 + (trow & 31) : trow;
        const int rb = tn0 + row < rows ? row : rows - 1 - tn0;
        c_voff[j] = (uint32_t)rb * w_row_bytes;
        p_voff[j] = (uint32_t)rb * param_row * 4u;
      }
    }
  };
  // requests the codes (and, where they change, the parameters) of super-step `ks` of the tile the code sources point at
  auto load_codes = [&](int ks, bool new_tile) {
    if constexpr (CODES) {
      uint32_t byte0;  // first byte of the lane's piece inside its row
      if constexpr (BKIND == WL_B_I8) {
        byte0 = (uint32_t)ks * 64u + c_piece * 16u;
      } else {
        // codes k0 .. k0 + 15 of the row live in ONE half of ONE packing block (block >= 32): byte j of a block holds code j
        // in its low and code j + block / 2 in its high nibble (ffq_pack_int4, _packing.py:44-53)
        const uint32_t k0 = (uint32_t)ks * 64u + c_piece * 16u, lb = (uint32_t)a.pack_shift;
        const uint32_t within = k0 & ((1u << lb) - 1u);
        nib_shift = (within >> (lb - 1u)) * 4u;
        byte0 = ((k0 >> lb) << (lb - 1u)) + (within & ((1u << (lb - 1u)) - 1u));
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const uint32_t vo = c_voff[j] + byte0;
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(raw[j]) : "v"(vo), "s"(c_base) : "memory");
      }
      // the stream visits the super-steps of a unit in order (k0, k0 + 1, ...): group bookkeeping by counting, one division per unit
      bool new_params = new_tile;
      if constexpr (GROUPED) {
        if (new_tile) { grp = ks / a.steps_per_group; grp_phase = ks - grp * a.steps_per_group; }  // a unit may start inside the K range (split-K)
        else if (++grp_phase == a.steps_per_group) { grp_phase = 0; ++grp; new_params = true; }
      }
      if (new_params) {  // wave-uniform
        const uint32_t g4 = GROUPED ? (uint32_t)grp * 4u : 0u;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const uint32_t vo = p_voff[j] + g4;
          asm volatile("global_load_dword %0, %1, %2" : "=v"(sc[j]) : "v"(vo), "s"(ps_base) : "memory");
          if constexpr (OFFSET) asm volatile("global_load_dword %0, %1, %2" : "=v"(ro[j]) : "v"(vo), "s"(po_base) : "memory");
        }
      }
    }
  };
  // everything requested so far has landed: the registers of the code stream are re-defined HERE for the compiler (the
  // loads above are invisible to its wait insertion)
  auto wait_all_vmem = [&]() {
    if constexpr (CODES) {
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(raw[0]), "+v"(raw[1]), "+v"(sc[0]), "+v"(sc[1]), "+v"(ro[0]), "+v"(ro[1]) : : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  };
  // row j's 16 codes -> 16 bf16 of A2's value -> the B image of `slot` (logical 16-byte slots 2 p and 2 p + 1 of the row)
  [[maybe_unused]] const uint32_t c_row0 = (uint32_t)(wave * 32 + (lane >> 2));
  auto convert_row = [&](int j, int slot) {
    if constexpr (CODES) {
      const uint32_t row = c_row0 + 16u * (uint32_t)j;
      uint32_t w[4] = {raw[j].x, raw[j].y, raw[j].z, raw[j].w};
      float s = sc[j], c = OFFSET ? rne(ro[j]) : 0.0f;
      if constexpr (BKIND == WL_B_I4) {
        // nibble n = code + 8 -> (n ^ 8) << 4 in the byte's high half = 16 * code as a signed byte; (16 q + 16 o) * (s / 16)
        // is (q + o) * s with the same single rounding wherever s / 16 is exact — on lanes with a tiny scale the bytes are
        // taken down to q itself (arithmetic >> 4 of every byte) and s stays
        const bool tiny = __builtin_fabsf(s) < 0x1p-120f && s != 0.0f;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          uint32_t b = (((w[d] >> nib_shift) << 4) & 0xF0F0F0F0u) ^ 0x80808080u;
          if (__builtin_expect(tiny, 0)) {
            const uint32_t b0 = (uint32_t)(((int32_t)(b << 24)) >> 28) & 0xFFu, b1 = (uint32_t)(((int32_t)(b << 16)) >> 28) & 0xFFu;
            const uint32_t b2 = (uint32_t)(((int32_t)(b << 8)) >> 28) & 0xFFu, b3 = (uint32_t)(((int32_t)b) >> 28) & 0xFFu;
            b = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
          }
          w[d] = b;
        }
        if (!tiny) { s = s * 0.0625f; c = c * 16.0f; }
      }
      uint32_t o[8];
#pragma unroll
      for (int d = 0; d < 4; ++d) dequantize4<OFFSET>(w[d], s, c, o[2 * d], o[2 * d + 1]);
      uint8_t* image = lds + slot * WL_SLOT + WL_IMAGE + row * 128u;
      const uint32_t sw = row & 7u;
      *reinterpret_cast<u32x4*>(image + (((2u * c_piece) ^ sw) << 4)) = u32x4{o[0], o[1], o[2], o[3]};
      *reinterpret_cast<u32x4*>(image + (((2u * c_piece + 1u) ^ sw) << 4)) = u32x4{o[4], o[5], o[6], o[7]};
    }
  };

  // ---- fragment byte offsets inside a slot: [row tile][32-k chunk]. A image (LDS-DMA): slot ^= (row >> 1) & 7; B image:
  // the same when it arrives by LDS-DMA, slot ^= row & 7 when the conversion writes it (conflict-free 16-byte stores)
  const uint32_t r16 = lane & 15, g4 = lane >> 4;
  // both swizzles depend on the row only through r16 (row tiles are 16 rows apart, the terms are taken mod 8 of row / 2 or
  // row): one register per k-chunk and operand, the row tile goes into the instruction's offset field (mi * 2048 bytes)
  uint32_t a_off[2], b_off[2];
  {
    const uint32_t arow = wm * 128 + r16, brow = wn * 64 + r16;
    const uint32_t bsw = CODES ? (brow & 7u) : ((brow >> 1) & 7u);
#pragma unroll
    for (int kq = 0; kq < 2; ++kq) {
      a_off[kq] = arow * 128 + (((kq * 4 + g4) ^ ((arow >> 1) & 7u)) << 4);
      b_off[kq] = WL_IMAGE + brow * 128 + (((kq * 4 + g4) ^ bsw) << 4);
    }
  }

  wl_v4f acc[8][4];
  wl_v4i fa[4], fb[4];
  auto read_frags = [&](const uint8_t* st, int phase) {  // phase 0..3 of a super-step (compile-time after unrolling)
    const int kq = phase >> 1, mh = phase & 1;
    if (mh == 0) {
#pragma unroll
      for (int nj = 0; nj < 4; ++nj) fb[nj] = *reinterpret_cast<const wl_v4i*>(st + b_off[kq] + nj * 2048);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) fa[q] = *reinterpret_cast<const wl_v4i*>(st + a_off[kq] + (4 * mh + q) * 2048);
  };
  auto mfma_row = [&](int mh, int q) {
#pragma unroll
    for (int n_ = 0; n_ < 4; ++n_) {
      const int nj = (q & 1) ? 3 - n_ : n_;  // snake order: every MFMA shares one operand with its predecessor
      if (mh == 0) acc[q][nj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(wl_v8bf, fb[nj]), __builtin_bit_cast(wl_v8bf, fa[q]), acc[q][nj], 0, 0, 0);
      else acc[4 + q][nj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(wl_v8bf, fb[nj]), __builtin_bit_cast(wl_v8bf, fa[q]), acc[4 + q][nj], 0, 0, 0);
    }
  };
  // An MFMA cluster: 16 MFMAs under s_setprio 1. `vmem` (the register loads of the code stream) is issued behind the first four;
  // `work` (the conversion of one row's codes: ~40 VALU + 2 ds_write_b128) is interleaved with all sixteen — a wave hides about
  // 2.5 issue slots under a 16-cycle MFMA; in one lump the conversion runs with the matrix pipe idle (the partner wave only reads
  // LDS meanwhile). The pattern below must name ONLY instruction kinds the cluster contains: a group the scheduler cannot fill
  // voids every group behind it — round 3 moved the LDS-DMA pieces out of the clusters and left their group in, and the
  // compiler emitted the conversion as one 30-instruction lump (found in round 4 by reading the ISA; tools/asm_cluster_check.py
  // now fails the build's CPU test when a cluster carries more than WL_MAX_VALU_RUN VALU instructions in a row).
  auto cluster = [&](int mh, auto vmem, auto work, auto has_work) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    if constexpr (decltype(has_work)::value) {
      mfma_row(mh, 0);
      mfma_row(mh, 1);
      mfma_row(mh, 2);
      mfma_row(mh, 3);
      vmem();
      work();
#if FFQ_WL_CLUSTER_MODE == 1
#pragma unroll
      for (int g = 0; g < 15; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);  // three VALU
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);    // whatever VALU is left (address arithmetic of the stores)
      __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);    // the two LDS stores last
#elif FFQ_WL_CLUSTER_MODE == 2
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x002, 16, 0);
      __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
#endif
    } else {
      mfma_row(mh, 0);
      __builtin_amdgcn_sched_barrier(0);
      vmem();
      __builtin_amdgcn_sched_barrier(0);
      mfma_row(mh, 1);
      mfma_row(mh, 2);
      mfma_row(mh, 3);
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(0);
  };
  // ---- free-running form (FFQ_WL_LOOP_MODE 1, bf16 image only): fragments double-buffered in registers, fetched one phase ahead
  constexpr bool FREE = !CODES && FFQ_WL_LOOP_MODE == 1;
  // ---- whole-k-chunk phases (FFQ_WL_LOOP_MODE 2, bf16 image only): a phase = one 32-deep k-chunk x ALL eight row tiles of the
  // wave: 12 fragment reads + 4 LDS-DMA pieces in the LOAD segment, 32 MFMAs in the cluster, 4 barriers per super-step instead of 8
  constexpr bool WIDE = !CODES && FFQ_WL_LOOP_MODE == 2;
  [[maybe_unused]] wl_v4i fa8[8];
  auto read_frags_q = [&](const uint8_t* st, int kq) {
#pragma unroll
    for (int nj = 0; nj < 4; ++nj) fb[nj] = *reinterpret_cast<const wl_v4i*>(st + b_off[kq] + nj * 2048);
#pragma unroll
    for (int q = 0; q < 8; ++q) fa8[q] = *reinterpret_cast<const wl_v4i*>(st + a_off[kq] + q * 2048);
  };
  auto cluster_q = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
      for (int n_ = 0; n_ < 4; ++n_) {
        const int nj = (q & 1) ? 3 - n_ : n_;
        acc[q][nj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(wl_v8bf, fb[nj]), __builtin_bit_cast(wl_v8bf, fa8[q]), acc[q][nj], 0, 0, 0);
      }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(0);
  };
  [[maybe_unused]] wl_v4i fa2[2][4], fb2[2][4];
  auto read_frags2 = [&](const uint8_t* st, int phase) {  // into the buffers phase `phase` computes from: fa2[phase & 1], fb2[(phase >> 1) & 1]
    const int kq = phase >> 1, mh = phase & 1;
    if (mh == 0) {
#pragma unroll
      for (int nj = 0; nj < 4; ++nj) fb2[kq & 1][nj] = *reinterpret_cast<const wl_v4i*>(st + b_off[kq] + nj * 2048);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) fa2[phase & 1][q] = *reinterpret_cast<const wl_v4i*>(st + a_off[kq] + (4 * mh + q) * 2048);
  };
  auto cluster2 = [&](int phase) {
    const int kq = phase >> 1, mh = phase & 1;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int n_ = 0; n_ < 4; ++n_) {
        const int nj = (q & 1) ? 3 - n_ : n_;
        acc[4 * mh + q][nj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(wl_v8bf, fb2[kq & 1][nj]), __builtin_bit_cast(wl_v8bf, fa2[phase & 1][q]),
                                                                  acc[4 * mh + q][nj], 0, 0, 0);
      }
  };
  constexpr std::integral_constant<bool, CODES> kConverts{};
  constexpr std::false_type kNoWork{};

  int m0 = 0, n0 = 0, k0 = 0, k1 = 0, tile_no = 0, slice = 0, seg = 0;
  tile_origin(0, m0, n0, k0, k1, tile_no, slice, seg);
  set_image_sources(m0, n0, seg);
  // ---- prologue: element 0 staged entirely, the codes of element 1 requested
  if constexpr (CODES) {
    set_code_sources(n0, seg);
    load_codes(k0, true);
    wait_all_vmem();
    convert_row(0, 0);
    convert_row(1, 0);
    load_codes(k0 + 1, false);
  }
  issue_a(k0, 0, 0); issue_a(k0, 0, 2); issue_b(k0, 0, 0); issue_b(k0, 0, 2);
  wait_all_vmem();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if constexpr (FREE) read_frags2(lds, 0);

  int slot = 0;  // slot of the super-step about to be computed
  for (int it = 0; it < my_tiles; ++it) {
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
      for (int nj = 0; nj < 4; ++nj)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[mi][nj][e] = 0.0f;
    int nm0, nn0, nk0, nk1, ntile_no, nslice, nseg;
    tile_origin(it + 1, nm0, nn0, nk0, nk1, ntile_no, nslice, nseg);
    if (!FREE && wm == 1) __builtin_amdgcn_s_barrier();  // the upper group runs one interval behind
    for (int ks = k0; ks < k1; ++ks) {
      const uint8_t* st = lds + slot * WL_SLOT;
      // images / conversion: element e + 1; code loads: element e + 2
      const bool last = ks == k1 - 1;
      const int fetch = last ? nk0 : ks + 1;
      if (last) set_image_sources(nm0, nn0, nseg);
      if constexpr (FREE) {
        // every wave on its own: the fragments of phase p + 1 are requested ahead of the MFMAs of phase p (registers for both), the
        // pieces of the next super-step go out at the head of phases 0 and 1, and ONE barrier per super-step — ahead of phase 3:
        // this wave's pieces have landed (vmcnt) and its reads of this slot are done (lgkmcnt) — publishes the other slot for
        // reading and this one for refilling
        issue_a(fetch, slot ^ 1, 0); issue_b(fetch, slot ^ 1, 0);
        read_frags2(st, 1);
        cluster2(0);
        issue_a(fetch, slot ^ 1, 2); issue_b(fetch, slot ^ 1, 2);
        read_frags2(st, 2);
        cluster2(1);
        read_frags2(st, 3);
        cluster2(2);
        wait_all_vmem();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        read_frags2(lds + (slot ^ 1) * WL_SLOT, 0);
        cluster2(3);
        slot ^= 1;
        continue;
      }
      if constexpr (WIDE) {
        // all eight pieces go out in the first LOAD segment and are waited for in the second one, ahead of its barrier: the other
        // group's first read of the new slot lies behind a barrier this wave passes only after its wait (RAW: wait -> barrier -> read)
        read_frags_q(st, 0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        issue_a(fetch, slot ^ 1, 0); issue_b(fetch, slot ^ 1, 0); issue_a(fetch, slot ^ 1, 2); issue_b(fetch, slot ^ 1, 2);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        cluster_q();
        __builtin_amdgcn_s_barrier();
        read_frags_q(st, 1);
        wait_all_vmem();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        cluster_q();
        __builtin_amdgcn_s_barrier();
        slot ^= 1;
        continue;
      }
      int code_ks = ks + 2;
      bool code_new_tile = false;
      if (code_ks >= k1) {
        code_ks = nk0 + (code_ks - k1);
        if (ks == k1 - 2) { set_code_sources(nn0, nseg); code_new_tile = true; }
      }
      // The LDS-DMA pieces are issued by the group that is NOT computing: behind its fragment reads (lgkmcnt(0): the LDS is no
      // longer serving them) and ahead of the barrier, in the first two LOAD segments of the super-step. An LDS-DMA instruction
      // blocks its wave's instruction stream for 60+ cycles; inside a cluster that is the matrix pipe running dry behind every
      // piece (round 3, A/B on one box: two-pass layer mix 1.35 -> 1.41 PFLOP/s; 3 + 3 + 2 pieces over three segments 1.39-1.40,
      // all in the first 1.35, none in the first 1.31). WAR on the target slot: its last readers (the other group's fragment reads
      // of the previous super-step) were issued a whole barrier interval earlier than these pieces, which themselves follow this
      // wave's own completed reads.
      read_frags(st, 0);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      issue_a(fetch, slot ^ 1, 0); issue_b(fetch, slot ^ 1, 0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      cluster(0, [] {}, [&] { convert_row(0, slot ^ 1); }, kConverts);
      __builtin_amdgcn_s_barrier();
      read_frags(st, 1);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      issue_a(fetch, slot ^ 1, 2); issue_b(fetch, slot ^ 1, 2);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      cluster(1, [] {}, [&] { convert_row(1, slot ^ 1); }, kConverts);
      __builtin_amdgcn_s_barrier();
      read_frags(st, 2);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      cluster(0, [&] { load_codes(code_ks, code_new_tile); }, [] {}, kNoWork);
      __builtin_amdgcn_s_barrier();
      read_frags(st, 3);
      wait_all_vmem();  // the fetched images and codes landed (and older epilogue stores); the B image's ds_writes retired at the last cluster's lgkmcnt(0)
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      cluster(1, [] {}, [] {}, kNoWork);
      __builtin_amdgcn_s_barrier();
      slot ^= 1;
    }
    if (!FREE && wm == 0) __builtin_amdgcn_s_barrier();  // same number of barriers for both groups
    // ---- epilogue in the A image of the slot just consumed. The weight fragment is the MFMA's first operand: lane l holds,
    // for tile (mi, nj), register t: row m = 16 mi + l % 16, column n = 16 nj + 4 (l / 16) + t — four consecutive output
    // columns of one row.
    __syncthreads();
    // ---- split-K: the `split` units of a tile normally run CONCURRENTLY (the launcher admits a split only while all units fit the
    // chip in one round) and exchange partial sums all-to-all. A wave's accumulators are four PIECES (row-tile pairs i = 0..3: the
    // rounds of the epilogue below); piece (wave, i) is FINISHED by the unit `(4 wave + i) % split`: every other unit stores its
    // partial piece to its own slab — register layout, [piece][8][lane] x 16 B, one dense KiB per wave instruction, WRITE-THROUGH
    // (sc1: a release fence would write back the whole L2, CDNA guide "publish-large") — counts itself in, waits until every unit of
    // the tile is in, and every wave adds the peers' partials of ITS pieces to the partial it kept, the owner's own first, then the
    // others in slice order: each piece has one fixed summation order — results do not depend on timing.
    // Nobody depends on that wait (round 5; ADVICE r4: a peer that cannot become resident — another kernel, another process, a CU
    // mask holding its CU — made the old form spin for minutes and trap). The tile's state is ONE 64-bit word
    // [abandoned-slice mask : 32 | arrived : 8 | left : 8]: a unit whose wait runs out publishes the pieces it OWNS as well, sets
    // its mask bit and exits, which frees its CU; if the atomic OR still finds fewer than `split` arrivals, the unit that arrives
    // LAST is guaranteed to see the bit (its own arrival RMW comes later in the word's modification order) and finishes the
    // abandoned unit's pieces from the slabs in that piece's canonical order — the same bits as the symmetric exchange; if the
    // OR finds everybody arrived, the unit simply carries on. The last unit to leave zeroes the word for the next launch.
    uint32_t own = 0xFu;  // bit i: this wave finishes piece i (wave-uniform)
    if (tile_no >= 0) {
      const int S = a.split;
      const int rows_left = a.M - (m0 + wm * 128);
      const int mi_cnt = rows_left <= 0 ? 0 : rows_left >= 128 ? 8 : (rows_left + 15) >> 4;  // wave-uniform
      own = 0;
      uint8_t* const tile_slabs = reinterpret_cast<uint8_t*>(a.slabs) + (size_t)tile_no * (size_t)S * WL_UNIT_SLAB;
      const auto mine = __builtin_amdgcn_make_buffer_rsrc(tile_slabs + (size_t)slice * WL_UNIT_SLAB, 0, (int)WL_UNIT_SLAB, 0x00020000);
      auto publish_piece = [&](int i) {
#pragma unroll
        for (int q = 0; q < 8; ++q)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(wl_v4u, acc[2 * i + (q >> 2)][q & 3]), mine, (((wave * 4 + i) * 8 + q) * 64 + lane) * 16, 0, /*sc1*/ 16);
      };
      // piece i of this wave <- the S partials in the slabs: `first` (its owner), then the others in slice order; `keep` = the
      // owner's partial is already in the registers (the owner itself is summing)
      auto gather_piece = [&](int i, int first, bool keep) {
        const uint32_t piece_off = (uint32_t)(((wave * 4 + i) * 8) * 64 + lane) * 16u;
        for (int step = keep ? 1 : 0; step < S; ++step) {
          const int sp = step == 0 ? first : (step - 1 < first ? step - 1 : step);  // first, 0, 1, .., first - 1, first + 1, ..
          const auto peer = __builtin_amdgcn_make_buffer_rsrc(tile_slabs + (size_t)sp * WL_UNIT_SLAB, 0, (int)WL_UNIT_SLAB, 0x00020000);
          wl_v4u got[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) got[q] = __builtin_amdgcn_raw_buffer_load_b128(peer, piece_off + q * 1024, 0, /*sc1*/ 16);
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const wl_v4f g = __builtin_bit_cast(wl_v4f, got[q]);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[2 * i + (q >> 2)][q & 3][e] = step == 0 ? g[e] : acc[2 * i + (q >> 2)][q & 3][e] + g[e];
          }
        }
      };
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (2 * i >= mi_cnt) continue;
        if ((wave * 4 + i) % S == slice) { own |= 1u << i; continue; }
        publish_piece(i);
      }
      // every storing wave drains its write-through stores, the block meets, ONE lane counts the unit in and polls (one poller
      // per block: many pollers on one word cost chip bandwidth, CDNA guide "polling-cost"). The partials are read with sc1 loads,
      // which bypass this CU's L1 (the L2 cannot hold these lines: it is invalidated at kernel start and every slab line is
      // written once and read once per launch).
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      unsigned long long* const state = reinterpret_cast<unsigned long long*>(a.tickets + 2 * tile_no);
      constexpr unsigned long long kArrive = 1ull << 32;  // (the `left` field: 1ull << 40, below)
      // two words at the end of the slot just consumed (the epilogue's regions end below 40 KiB of it; the other slot may still receive
      // the trailing pieces of the DMA stream): [0]: 1 = everybody is in, 3 = the wait ran out; [1]: abandoned slices seen by the last arriver
      int* const verdict = reinterpret_cast<int*>(lds + (slot ^ 1) * WL_SLOT + WL_SLOT - 16);
      if (tid == 0) {
        const unsigned long long old = __hip_atomic_fetch_add(state, kArrive, FFQ_TICKET_ORDER, __HIP_MEMORY_SCOPE_AGENT);
        int v = 1;
        uint32_t seen = 0;
        if ((int)((old >> 32) & 0xFFu) + 1 == S) {
          seen = (uint32_t)old;  // the last to arrive: whoever gave up is known now, nobody can give up later
        } else {
          // ~0.25 ms of polling (an L2 round trip + s_sleep per poll): two orders of magnitude beyond a healthy exchange
          const uint32_t budget = a.abandon_test ? ((slice & 1) ? 0u : (1u << 20)) : 256u;
          uint32_t spins = 0;
          while ((int)((__hip_atomic_load(state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32) & 0xFFu) < S) {
            if (spins++ >= budget) { v = 3; break; }
            __builtin_amdgcn_s_sleep(8);
          }
          ffq_ticket_acquire();  // the polls themselves stay relaxed: one acquire once everybody is in (or the wait ran out)
        }
        verdict[0] = v;
        verdict[1] = (int)seen;
      }
      __syncthreads();
      int v = verdict[0];
      const uint32_t seen = (uint32_t)verdict[1];
      __syncthreads();  // verdict is re-written below
      if (v == 3) {  // give up: the owned pieces go to the slab as well, then the mask bit
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if ((own >> i) & 1u) publish_piece(i);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
          const unsigned long long old = __hip_atomic_fetch_or(state, 1ull << slice, FFQ_TICKET_ORDER, __HIP_MEMORY_SCOPE_AGENT);
          verdict[0] = (int)((old >> 32) & 0xFFu) == S ? 1 : 2;  // everybody came in meanwhile: nobody will cover for this unit, carry on
        }
        __syncthreads();
        v = verdict[0];
        if (v == 2) own = 0;  // abandoned: the last arriver finishes this unit's pieces; nothing left to do but to leave
      }
      if (v == 1) {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if ((own >> i) & 1u) gather_piece(i, slice, true);
        if (seen) {  // the last arriver covers for the units that gave up
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int owner = (wave * 4 + i) % S;
            if (2 * i >= mi_cnt || owner == slice || !((seen >> owner) & 1u)) continue;
            gather_piece(i, owner, false);
            own |= 1u << i;
          }
        }
      }
    }
    {
      TOut* out = static_cast<TOut*>(MLP || seg == 0 ? a.out : seg == 1 ? a.seg_out[0] : a.seg_out[1]);
      const int out_n = MLP ? a.N : seg_rows(seg);           // columns of this tile's output tensor
      constexpr int COLS = MLP ? 32 : 64;                     // output columns per wave
      constexpr int NJ = COLS / 16;                           // column tiles that leave the wave
      constexpr int PITCH = COLS * (int)sizeof(TOut) + 16;
      constexpr int SLAB = sizeof(TOut) == 2 ? 32 : 16;       // rows per round: the eight waves' slabs share one 64 KiB slot
      constexpr int WAVE_BYTES = SLAB * PITCH + 256;          // one slab + the wave's bias values
      static_assert(8 * WAVE_BYTES <= WL_SLOT, "the epilogue scratch must fit the consumed slot");
      uint8_t* region = lds + (slot ^ 1) * WL_SLOT + wave * WAVE_BYTES;
      float* bias_lds = reinterpret_cast<float*>(region + SLAB * PITCH);
      const int wave_n0 = n0 + wn * COLS, wave_m0 = m0 + wm * 128;
      const bool full = wave_n0 + COLS <= out_n && (out_n * (int)sizeof(TOut)) % 16 == 0;
      const bool has_bias = !MLP && a.bias != nullptr;
      if (lane < COLS) {
        const int n = wave_n0 + lane;
        bias_lds[lane] = has_bias ? (float)load_any(a.bias, a.bias_dt, n < out_n ? n : out_n - 1) : 0.0f;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private: only this wave's own writes
      if constexpr (MLP) {
        // z = bf16(silu(bf16(gate))) * bf16(up), rounded to bf16: the accumulator tiles nj = 0, 1 take the product (as the float of
        // that bf16 value), tiles 2, 3 (up) are spent. silu through the LDS table of ffq_silu.h (all 65536 bf16 patterns checked
        // against ATen), the reads of a row tile issued back to back, one wave-uniform branch for values outside its window.
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
          if (!((own >> (mi >> 1)) & 1u)) continue;  // split-K: another unit finishes this piece
          uint32_t wg[2][2], ws[2][2], bad = 0;
#pragma unroll
          for (int nj = 0; nj < 2; ++nj)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              wg[nj][h] = pack2<bf16_t>(acc[mi][nj][2 * h], acc[mi][nj][2 * h + 1]);
              ws[nj][h] = silu_pair_lookup(wg[nj][h], silu_table, bad);
            }
          if (__builtin_expect(silu_any_outside(bad), 0)) {
#pragma unroll
            for (int nj = 0; nj < 2; ++nj)
#pragma unroll
              for (int h = 0; h < 2; ++h) ws[nj][h] = silu_pair_patch(wg[nj][h], ws[nj][h]);
          }
#pragma unroll
          for (int nj = 0; nj < 2; ++nj)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const uint32_t wu = pack2<bf16_t>(acc[mi][nj + 2][2 * h], acc[mi][nj + 2][2 * h + 1]);
              const float a0 = __builtin_bit_cast(float, ws[nj][h] << 16), a1 = __builtin_bit_cast(float, ws[nj][h] & 0xFFFF0000u);
              const float u0 = __builtin_bit_cast(float, wu << 16), u1 = __builtin_bit_cast(float, wu & 0xFFFF0000u);
              acc[mi][nj][2 * h] = a0 * u0;          // rounded to bf16 by the slab write below: one rounding, as the eager multiply
              acc[mi][nj][2 * h + 1] = a1 * u1;
            }
        }
      }
#pragma unroll
      for (int i = 0; i < 128 / SLAB; ++i) {
        if (!((own >> (i * SLAB / 32)) & 1u)) continue;  // split-K: another unit finishes this piece (wave-uniform)
#pragma unroll
        for (int hh = 0; hh < SLAB / 16; ++hh) {
          const int mi = (SLAB / 16) * i + hh;
#pragma unroll
          for (int nj = 0; nj < NJ; ++nj) {
            const int nb = nj * 16 + 4 * g4;
            const wl_v4f b4 = *reinterpret_cast<const wl_v4f*>(bias_lds + nb);
            float y[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) y[t] = has_bias ? acc[mi][nj][t] + b4[t] : acc[mi][nj][t];
            if constexpr (sizeof(TOut) == 2) {
              u32x2 pk;
              pk.x = pack2<TOut>(y[0], y[1]);
              pk.y = pack2<TOut>(y[2], y[3]);
              *reinterpret_cast<u32x2*>(region + (16 * hh + r16) * PITCH + nb * 2) = pk;
            } else {
              u32x4 pk;
              pk.x = __builtin_bit_cast(uint32_t, y[0]); pk.y = __builtin_bit_cast(uint32_t, y[1]);
              pk.z = __builtin_bit_cast(uint32_t, y[2]); pk.w = __builtin_bit_cast(uint32_t, y[3]);
              *reinterpret_cast<u32x4*>(region + (16 * hh + r16) * PITCH + nb * 4) = pk;
            }
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (full) {
          constexpr int SEGS = COLS * (int)sizeof(TOut) / 16;  // 16-byte segments per row: 8 (bf16) / 16 (f32); 4 in MLP mode
#pragma unroll
          for (int t = 0; t < SLAB * SEGS / 64; ++t) {
            const int c = lane + 64 * t;
            const int row = c / SEGS, sg = c % SEGS;
            const int mm = wave_m0 + i * SLAB + row;
            const u32x4 v = *reinterpret_cast<const u32x4*>(region + row * PITCH + sg * 16);
            // non-temporal, as in ffq_linear.hip: the output must not push the operand panels out of L2
            if (mm < a.M) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(reinterpret_cast<uint8_t*>(out) + ((size_t)mm * out_n + wave_n0) * sizeof(TOut) + sg * 16));
          }
        } else {  // ragged right edge / unaligned rows: element stores (correctness path)
          for (int c = lane; c < SLAB * COLS; c += 64) {
            const int row = c / COLS, col = c % COLS;
            const int mm = wave_m0 + i * SLAB + row;
            if (mm < a.M && wave_n0 + col < out_n)
              out[(size_t)mm * out_n + wave_n0 + col] = *reinterpret_cast<const TOut*>(region + row * PITCH + col * sizeof(TOut));
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the slab is re-written by the next i
      }
    }
    __syncthreads();  // the scratch slot is the next tile's staging target
    if (tile_no >= 0 && tid == 0) {  // every wave of this block is done with the slabs (the barrier above): count the unit out
      unsigned long long* const state = reinterpret_cast<unsigned long long*>(a.tickets + 2 * tile_no);
      const unsigned long long old = __hip_atomic_fetch_add(state, 1ull << 40, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((int)((old >> 40) & 0xFFu) == a.split - 1) __hip_atomic_store(state, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // zero for the next launch
    }
    m0 = nm0; n0 = nn0; k0 = nk0; k1 = nk1; tile_no = ntile_no; slice = nslice; seg = nseg;
  }
  wait_all_vmem();  // the trailing requests of the streams must not outlive the block's LDS / registers
}

// =====================================================================================================================================
// The bf16-image form with ONE wavefront per SIMD (round 4; tools/probes/gemm4w_probe.hip is the experiment this came out of):
// 4 waves x 128 x 128 accumulators — 256 registers per lane, pinned to AGPRs as "+a" operands of inline-assembly MFMAs (with the
// builtin hipcc rotates them through VGPRs: ~590 v_accvgpr moves per two super-steps) — so a k-half of 64 MFMAs needs 16 fragment
// reads (0.25 per MFMA against 0.375 with 128 x 64 per wave), hand-placed between the MFMAs ahead of their use; both operand images by
// LDS-DMA into the two 64 KiB slots. The K-loop is `asm volatile` in source order: 256 MFMAs + 64 ds_read_b128 + 32 LDS-DMA pieces +
// scalar bookkeeping per two super-steps and nothing else. It runs ACROSS tile boundaries: the last two super-steps of a tile fetch the
// first two of the next and read its first fragments, and the epilogue in between goes through wave-private rows of LDS behind the
// slots while those images land. Same MFMA instruction, same k order as wq_gemm256_kernel: bit-identical results.
//
// Round 6 — the A operand runs HALF A STEP AHEAD of the B operand (super_step below). Rounds 4-5 refilled a slot as a whole: all 16
// pieces of a wave in the second k-half, none in the first. A piece occupies the CU's texture path for 16 cycles (1 KiB at 64 B/clk):
// four waves x 16 pieces are 1024 cycles of it inside a k-half of 1024 MFMA cycles, and a wave that cannot issue its piece cannot
// issue its next MFMA either — the "60+ cycles per LDS-DMA issue" of rounds 2-5 was this queue, not the instruction. With A's fetch,
// fragment reads and refill shifted half a step against B's, each IMAGE of a slot is refilled as soon as its last reader is done and
// every k-half issues 8 pieces; A(k+2) gets 1 - 1.5 super-steps to land (counted `vmcnt(8)` waits), B(k+2) 0.5 - 1 as before. Same
// tiles, same k order, same bits (tests/test_gemm_gpu.py). A/B of two builds, two interleaved rounds (profiles/r06_w4_half_ab.txt,
// 16 k tokens x 4096 columns): K = 4096 401 -> 375 us, 12288 1154 -> 1074, 14336 (down_proj) 1373 -> 1253, 16384 1690 -> 1435; the
// distance to the vendor's GEMM on the dequantized weight 8 - 23 % -> 2.5 - 5.5 %. (Earlier placements of the 16 pieces INSIDE the second
// k-half — back to back, four MFMAs apart (+1.9 %), an L2 warm-up 2 - 4 steps ahead (-3..-6 %: more texture work) — are in
// profiles/r06_w4_spread_ab.txt / r06_w4_ahead_ab.txt and docs/experiments.md.)
// Taken for plain launches of whole tiles (M and every weight matrix a multiple of 256 rows) without a split tail, without a bias
// and with an even number of super-steps (wq_dispatch); bf16 output.
// The gate+up+SiLU*up launch runs here as well (template parameter MLP, whole tiles of 256 rows x 128 output columns): round 4 had built
// that mode on the slot-at-a-time schedule, measured it 3-5 % SLOWER than the 8-wave kernel's (2.83 against 2.75 ms at 16 k tokens) and
// removed it; on the half-step-ahead schedule it is 6-8 % FASTER (profiles/r06_w4_mlp_ab.txt: 2.74-2.79 -> 2.59 ms at 16 k tokens,
// 0.75 -> 0.70 at 4 k; two launches of the vendor's GEMM + SiLU*up: 2.71 / 0.735), bit-equal to the 8-wave kernel and to the
// composition of its parts.
constexpr int W4_PITCH_PLAIN = 128 * 2 + 16;  // a staged row of a wave: 128 bf16 + pad

// fn(integral_constant<int, 0>), fn(<1>), ...: a loop whose index is a constant expression in every body (hipcc gives up fully unrolling
// a 64-trip loop around a 24-case switch; with constant indices no switch is needed)
template <typename F, int... I>
__device__ __forceinline__ void w4_each(F&& fn, std::integer_sequence<int, I...>) { (fn(std::integral_constant<int, I>{}), ...); }
// MLP (round 6): gate_proj + up_proj + SiLU * up in this kernel — the B image holds, per wave column (wn), 64 gate_proj rows and the same
// 64 up_proj rows (a wave's column tiles nj = 0..3 are gate, nj + 4 up of the SAME 64 output columns), the output tile is 256 x 128
// bf16(silu(bf16(gate))) * bf16(up) with the roundings of wq_gemm256_kernel's MLP mode (the reference's quantized_llama/mlp.py:30-40):
// same MFMA instruction, same k order, same epilogue arithmetic — bit-equal to that kernel, which keeps ragged shapes and splits.
constexpr int W4_PITCH_MLP = 64 * 2 + 16;  // a staged row of a wave in MLP mode: 64 bf16 + pad (the SiLU table needs 16 KiB of the LDS)
template <bool MLP>
__global__ __launch_bounds__(256, 1) void wq_gemm4w_kernel(WLinearArgs a, int total_tiles) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // the tile walk of wq_gemm256_kernel without a split tail: XCD x owns a contiguous range of the grouped order
  const uint32_t nblk = gridDim.x;
  const uint32_t xcd = blockIdx.x & 7u, j_in_xcd = blockIdx.x >> 3;
  const uint32_t blocks_in_xcd = (nblk >> 3) + (xcd < (nblk & 7u) ? 1u : 0u);
  uint32_t first, count;
  {
    const uint32_t tq = (uint32_t)total_tiles >> 3, tr = (uint32_t)total_tiles & 7u;
    first = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
    count = tq + (xcd < tr ? 1u : 0u);
  }
  const int my_tiles = j_in_xcd < count ? (int)((count - j_in_xcd + blocks_in_xcd - 1) / blocks_in_xcd) : 0;
  if (my_tiles == 0) return;
  const int ksuper = a.K / WL_BK;  // even (wq_dispatch)
  constexpr int STAGE_PITCH = MLP ? W4_PITCH_MLP : W4_PITCH_PLAIN;
  uint8_t* const stage = lds + 2 * WL_SLOT + wave * (16 * STAGE_PITCH);
  [[maybe_unused]] uint16_t* const silu_table = reinterpret_cast<uint16_t*>(lds + 2 * WL_SLOT + 4 * (16 * STAGE_PITCH));
  if constexpr (MLP) silu_table_fill(silu_table, (uint32_t)tid, 256u);  // published by the prologue's barriers
  // A contraction depth that is a large power of two puts the same depth of every row of every tile on the same few memory channels,
  // and all 256 CUs walk the depth in step: XCD x starts x/8 of the way in and wraps (round 6, A/B of two builds, two rounds,
  // profiles/r06_krot_ab.txt: K = 8192 848 -> 743 us at 16 k tokens x 4096 columns, 16384 +2 %; other depths +-1.5 % or worse — off there).
  // The fp32 sum of a tile is then taken in that rotated order: a function of the shape and the grid, like the rest of the plan.
  const uint32_t k_rot = (a.K >= 8192 && (a.K & (a.K - 1)) == 0) ? xcd * (uint32_t)ksuper / 8u : 0u;

  auto tile_origin = [&](int it, int& tm0, int& tn0, int& seg) {
    it = it < my_tiles ? it : my_tiles - 1;  // the stream running past the block's last tile re-reads it (never used)
    const uint32_t tile_id = first + j_in_xcd + (uint32_t)it * blocks_in_xcd;
    const uint32_t gm = (uint32_t)a.group_m;
    const uint32_t across = a.group_cols ? (uint32_t)a.tiles_m : (uint32_t)a.tiles_n;
    const uint32_t along = a.group_cols ? (uint32_t)a.tiles_n : (uint32_t)a.tiles_m;
    const uint32_t per_group = gm * across;
    const uint32_t group = tile_id / per_group, in_group = tile_id - group * per_group;
    const uint32_t group_size = min(gm, along - group * gm);
    const uint32_t inner = group * gm + in_group % group_size, outer = in_group / group_size;
    tm0 = (int)(a.group_cols ? outer : inner) * WL_BM;
    int tn = (int)(a.group_cols ? inner : outer);
    seg = 0;  // which weight matrix the column tile belongs to, and the tile's origin inside it
    if constexpr (MLP) {
      tn0 = tn * 128;  // 128 output columns per tile: 128 gate + 128 up rows in the B image
    } else {
      if (tn >= a.seg_tile[1]) { seg = 2; tn -= a.seg_tile[1]; }
      else if (tn >= a.seg_tile[0]) { seg = 1; tn -= a.seg_tile[0]; }
      tn0 = tn * WL_BN;
    }
  };
  auto seg_codes = [&](int seg) { return seg == 0 ? a.w : seg == 1 ? a.seg_w[0] : a.seg_w[1]; };
  auto seg_rows = [&](int seg) { return seg == 0 ? a.seg_n[0] : seg == 1 ? a.seg_n[1] : a.seg_n[2]; };
  auto row_base = [&](const uint8_t* base, int row0, uint32_t row_bytes) {
    const uint64_t off = (uint64_t)(uint32_t)row0 * (uint64_t)row_bytes;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)off), hi = __builtin_amdgcn_readfirstlane((uint32_t)(off >> 32));
    return base + (((uint64_t)hi << 32) | lo);
  };

  // ---- LDS-DMA sources: piece c (0..7) of wave w covers image rows (8 w + c) * 8 + lane / 8; the lane's 16-byte slot is
  // swizzled on the SOURCE address (slot ^ (row / 2) % 8): the linear LDS write lands in the swizzled image. Buffer addressing
  // (`buffer_load_dwordx4 ... offen lds`): one descriptor per operand and tile (base = the tile's first row, extent = what is
  // left of the matrix), ONE per-lane offset for the even and one for the odd pieces (the swizzle term (row / 2) % 8 contains
  // bit 0 of the piece number), the piece and the super-step in the scalar offset — no vector arithmetic in the K-loop, no
  // per-piece registers (the vendor kernel's addressing). The scalar offset is outside the hardware's range check: that is fine for
  // the WEIGHT image (whole tiles only, wq_takes_4w: every row a piece names exists). The ACTIVATION pieces carry their row in the
  // per-lane offset instead (eight registers, round 6): rows past a ragged last row tile fail the range check and arrive as zeros,
  // so any token count takes this kernel; the epilogue stores the rows that exist.
  const uint32_t row_bytes = (uint32_t)a.K * 2u;
  const int d_row = lane >> 3;
  uint32_t d_voff[2];
#pragma unroll
  for (int odd = 0; odd < 2; ++odd) {
    const int row = (wave * 8 + odd) * 8 + d_row;  // piece `odd` of this wave; pieces c and c + 2 differ by 16 rows: same swizzle
    const int d_slot = (lane & 7) ^ ((row >> 1) & 7);
    d_voff[odd] = (uint32_t)(wave * 64 + d_row) * row_bytes + d_slot * 16;
  }
  uint32_t a_voff[8];  // activation piece c: the lane's row INSIDE the descriptor's range check
#pragma unroll
  for (int c = 0; c < 8; ++c) a_voff[c] = d_voff[c & 1] + (uint32_t)c * 8u * row_bytes;
  // MLP: image rows 64 w .. 64 w + 63 of the B image are rows 64 (w / 2) .. of gate_proj (even waves) or up_proj (odd waves) counted
  // from the tile's first output column: a wave fetches from ONE matrix, its descriptor points at that matrix, and the lane offset
  // names the matrix row (the swizzle term stays the image row's)
  [[maybe_unused]] uint32_t b_voff[2] = {0u, 0u};
  if constexpr (MLP) {
#pragma unroll
    for (int odd = 0; odd < 2; ++odd) {
      const int row = (wave * 8 + odd) * 8 + d_row;
      const int d_slot = (lane & 7) ^ ((row >> 1) & 7);
      b_voff[odd] = (uint32_t)((wave >> 1) * 64 + d_row) * row_bytes + d_slot * 16;
    }
  }
  __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, 0, 0x00020000);
  __amdgpu_buffer_rsrc_t b_rsrc = a_rsrc;
  auto extent = [&](int rows_left) {  // bytes from the tile's first row to the end of the matrix, as a descriptor extent
    const uint64_t bytes = (uint64_t)(uint32_t)(rows_left > 0 ? rows_left : 0) * (uint64_t)row_bytes;
    return (int)(bytes < 0xFFFFFFFFull ? (uint32_t)bytes : 0xFFFFFFFFu);
  };
  auto set_image_sources = [&](int tm0, int tn0, int seg) {
    a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)row_base(a.x, tm0, row_bytes), 0, extent(a.M - tm0), 0x00020000);
    if constexpr (MLP) b_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)row_base((wave & 1) ? a.w2 : a.w, tn0, row_bytes), 0, extent(a.N - tn0), 0x00020000);
    else b_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)row_base(seg_codes(seg), tn0, row_bytes), 0, extent(seg_rows(seg) - tn0), 0x00020000);
  };
  auto issue_one = [&](int ks, int slot, int c, int which) {  // piece c of the A (0) or B (1) image of super-step ks
    uint8_t* base = lds + slot * WL_SLOT;
    ks += (int)k_rot;  // (scalar: two instructions per piece)
    ks = ks >= ksuper ? ks - ksuper : ks;
    const uint32_t soff = (uint32_t)c * 8u * row_bytes + (uint32_t)ks * 128u;
    if (which == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (wl_lds_t*)(base + (wave * 8 + c) * 1024), 16, a_voff[c], (uint32_t)ks * 128u, 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rsrc, (wl_lds_t*)(base + WL_IMAGE + (wave * 8 + c) * 1024), 16, MLP ? b_voff[c & 1] : d_voff[c & 1], soff, 0, 0);
  };

  // ---- fragment addresses: lane (r16, g4) reads 8 bf16 of row r16 of a 16-row tile, logical slot kq * 4 + g4; one register per
  // (slot, k-half, operand), the row tile in the instruction's offset field (t * 2048)
  const uint32_t r16 = lane & 15, g4 = lane >> 4;
  uint32_t a_off[2][2], b_off[2][2];
  {
    const uint32_t arow = wm * 128 + r16, brow = wn * 128 + r16;
#pragma unroll
    for (int sl = 0; sl < 2; ++sl)
#pragma unroll
      for (int kq = 0; kq < 2; ++kq) {
        a_off[sl][kq] = sl * WL_SLOT + arow * 128 + (((kq * 4 + g4) ^ ((arow >> 1) & 7u)) << 4);
        b_off[sl][kq] = sl * WL_SLOT + WL_IMAGE + brow * 128 + (((kq * 4 + g4) ^ ((brow >> 1) & 7u)) << 4);
      }
  }
  wl_v4i fX[8];      // A fragments of the first k-half of the current super-step (re-loaded for the next one during the second k-half)
  wl_v4i fY[2][8];   // A fragments of the second k-half, by super-step parity
  wl_v4i fB[2][8];   // B fragments [k-half][column tile]
  wl_v4f acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = wl_v4f{0.f, 0.f, 0.f, 0.f};
#define FFQ_W4_MFMA(ACC, B, A) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(ACC) : "v"(B), "v"(A))
#define FFQ_W4_READ(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF))
  // one fragment: 16-row tile t of the image whose lane address is `off`
  auto read_tile = [&](wl_v4i (&dst)[8], uint32_t off, int t) __attribute__((always_inline)) {
    switch (t) {
      case 0: FFQ_W4_READ(dst[0], off, 0 * 2048); break;
      case 1: FFQ_W4_READ(dst[1], off, 1 * 2048); break;
      case 2: FFQ_W4_READ(dst[2], off, 2 * 2048); break;
      case 3: FFQ_W4_READ(dst[3], off, 3 * 2048); break;
      case 4: FFQ_W4_READ(dst[4], off, 4 * 2048); break;
      case 5: FFQ_W4_READ(dst[5], off, 5 * 2048); break;
      case 6: FFQ_W4_READ(dst[6], off, 6 * 2048); break;
      default: FFQ_W4_READ(dst[7], off, 7 * 2048); break;
    }
  };
  // one k-half: 64 MFMAs on (fa, fb) in snake order with `reads(<i>)` behind MFMA i and LDS-DMA piece c behind MFMA 8 c + 3
  auto half = [&](const wl_v4i (&fa)[8], const wl_v4i (&fb)[8], auto reads, auto dma) __attribute__((always_inline)) {
    w4_each([&acc, &fa, &fb, &reads, &dma](auto ic) __attribute__((always_inline)) {
      constexpr int i = decltype(ic)::value, mi = i >> 3, n_ = i & 7, nj = (mi & 1) ? 7 - n_ : n_;
      FFQ_W4_MFMA(acc[mi][nj], fb[nj], fa[mi]);
      reads(ic);
      if constexpr ((i & 7) == 3) dma(i >> 3);
    }, std::make_integer_sequence<int, 64>{});
  };
  int m0 = 0, n0 = 0, seg = 0;
  tile_origin(0, m0, n0, seg);
  set_image_sources(m0, n0, seg);
#pragma unroll
  for (int c = 0; c < 8; ++c) issue_one(0, 0, c, 0);
#pragma unroll
  for (int c = 0; c < 8; ++c) issue_one(0, 0, c, 1);
#pragma unroll
  for (int c = 0; c < 8; ++c) issue_one(1, 1, c, 0);
#pragma unroll
  for (int c = 0; c < 8; ++c) issue_one(1, 1, c, 1);
  asm volatile("s_waitcnt vmcnt(16)" ::: "memory");  // super-step 0 has landed (this wave's pieces), super-step 1 may still fly
  __syncthreads();
#pragma unroll
  for (int t = 0; t < 8; ++t) read_tile(fB[0], b_off[0][0], t);
#pragma unroll
  for (int t = 0; t < 8; ++t) read_tile(fX, a_off[0][0], t);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int t = 0; t < 8; ++t) read_tile(fY[0], a_off[0][1], t);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();  // everybody has read A(0): super-step 0 refills that image
#pragma unroll 1
  for (int it = 0; it < my_tiles; ++it) {
    int nm0, nn0, nseg;
    tile_origin(it + 1, nm0, nn0, nseg);
    // Super-step ks from slot `cur` with the A operand HALF A STEP AHEAD of the B operand: at entry the registers hold A(ks) for both
    // k-halves and B(ks, k-half 0). Each image of a slot is refilled as soon as its last reader is done, so the LDS-DMA stream is spread
    // over the whole super-step — 8 pieces per wave and k-half — where the slot-at-a-time schedule issues all 16 in the second k-half:
    // a piece occupies the CU's texture path for 16 cycles (1 KiB at 64 B/clk), four waves x 16 pieces are 1024 cycles of it inside a
    // k-half of 1024 MFMA cycles, and a wave that cannot issue its piece cannot issue its next MFMA either.
    //   first k-half   MFMAs A(ks,0) x B(ks,0) | read B(ks,1) from `cur` | DMA A(kn) -> the A image of `cur` (last readers: the second
    //                  k-half of step ks-1, behind that step's closing barrier)
    //   barrier        vmcnt(8): everything but the 8 A pieces just issued has landed — A(ks+1), B(ks+1)
    //   second k-half  MFMAs A(ks,1) x B(ks,1) | read B(ks+1,0), A(ks+1,0), A(ks+1,1) from `nxt` | DMA B(kn) -> the B image of `cur`
    //                  (last readers: the first k-half, behind the barrier above)
    //   barrier        the A image of `nxt` has been read in full (no memory wait: nothing new is needed yet)
    // A(kn) has 1 - 1.5 super-steps to land, B(kn) 0.5 - 1. Loads return in order, so "at most 8 outstanding" means the 8 newest
    // whatever stores of an epilogue are still in flight.
    auto super_step = [&](int kn, auto cur_c) __attribute__((always_inline)) {
      constexpr int cur = decltype(cur_c)::value, nxt = cur ^ 1;
      half(fX, fB[0],
           [&fB, &b_off](auto ic) __attribute__((always_inline)) {  // 8 reads behind MFMAs 1, 4, .. 22
             // (explicit captures: clang does not capture what a generic lambda names only as an asm operand; the casts keep -Wall quiet)
             (void)fB; (void)b_off;
             constexpr int i = decltype(ic)::value;
             if constexpr (i % 3 == 1 && i / 3 < 8) FFQ_W4_READ(fB[1][i / 3], b_off[cur][1], (i / 3) * 2048);
           },
           [&](int c) __attribute__((always_inline)) { issue_one(kn, cur, c, 0); });
      asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      half(fY[cur], fB[1],
           [&fB, &fX, &fY, &a_off, &b_off](auto ic) __attribute__((always_inline)) {  // 24 reads behind MFMAs 1, 3, .. 47
             (void)fB; (void)fX; (void)fY; (void)a_off; (void)b_off;
             constexpr int i = decltype(ic)::value, r = i >> 1;
             if constexpr ((i & 1) == 1 && r < 8) FFQ_W4_READ(fB[0][r], b_off[nxt][0], r * 2048);
             else if constexpr ((i & 1) == 1 && r < 16) FFQ_W4_READ(fX[r - 8], a_off[nxt][0], (r - 8) * 2048);
             else if constexpr ((i & 1) == 1 && r < 24) FFQ_W4_READ(fY[nxt][r - 16], a_off[nxt][1], (r - 16) * 2048);
           },
           [&](int c) __attribute__((always_inline)) { issue_one(kn, cur, c, 1); });
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    };
    // ONE loop body for every pair of super-steps (a separate copy for the tile's last pair got its own register allocation from
    // hipcc, 256 VGPRs and a spilled fragment): the last pair fetches super-steps 0 and 1 of the NEXT tile and reads its first fragments
#pragma unroll 1
    for (int ks = 0; ks < ksuper; ks += 2) {
      int k0 = ks + 2, k1 = ks + 3;
      if (ks == ksuper - 2) {
        set_image_sources(nm0, nn0, nseg);
        k0 = 0; k1 = 1;
      }
      super_step(k0, std::integral_constant<int, 0>{});
      super_step(k1, std::integral_constant<int, 1>{});
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // the last MFMAs' results before the accumulators are read

    if constexpr (MLP) {
      // ---- MLP epilogue: z = bf16(silu(bf16(gate))) * bf16(up) of the wave's 128 rows x 64 output columns, 16 rows at a time through its
      // own staging rows (no block barrier). silu through the LDS table of ffq_silu.h: the 16 reads of a row tile issued back to back,
      // one wave-uniform branch for values outside its window (wq_gemm256_kernel's MLP epilogue, operation for operation).
      bf16_t* out = static_cast<bf16_t*>(a.out);
      const int out_n = a.N;
      constexpr int PITCH = W4_PITCH_MLP;
      const int wave_n0 = n0 + wn * 64, wave_m0 = m0 + wm * 128;
      uint32_t lane = threadIdx.x & 63u;  // (formed here, from an opaque copy: see the plain epilogue)
      asm volatile("" : "+v"(lane));
      const uint32_t r16 = lane & 15, g4 = lane >> 4;
#pragma unroll
      for (int mi = 0; mi < 8; ++mi) {
        uint32_t wg[4][2], ws[4][2], bad = 0;
#pragma unroll
        for (int nj = 0; nj < 4; ++nj)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            wg[nj][h] = pack2<bf16_t>(acc[mi][nj][2 * h], acc[mi][nj][2 * h + 1]);
            ws[nj][h] = silu_pair_lookup(wg[nj][h], silu_table, bad);
          }
        if (__builtin_expect(silu_any_outside(bad), 0)) {
#pragma unroll
          for (int nj = 0; nj < 4; ++nj)
#pragma unroll
            for (int h = 0; h < 2; ++h) ws[nj][h] = silu_pair_patch(wg[nj][h], ws[nj][h]);
        }
#pragma unroll
        for (int nj = 0; nj < 4; ++nj) {
          uint32_t z[2];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const uint32_t wu = pack2<bf16_t>(acc[mi][nj + 4][2 * h], acc[mi][nj + 4][2 * h + 1]);
            const float a0 = __builtin_bit_cast(float, ws[nj][h] << 16), a1 = __builtin_bit_cast(float, ws[nj][h] & 0xFFFF0000u);
            const float u0 = __builtin_bit_cast(float, wu << 16), u1 = __builtin_bit_cast(float, wu & 0xFFFF0000u);
            z[h] = pack2<bf16_t>(a0 * u0, a1 * u1);  // one rounding, as the eager multiply
          }
          u32x2 pk;
          pk.x = z[0]; pk.y = z[1];
          *reinterpret_cast<u32x2*>(stage + r16 * PITCH + (nj * 16 + 4 * g4) * 2) = pk;
          acc[mi][nj] = wl_v4f{0.f, 0.f, 0.f, 0.f};
          acc[mi][nj + 4] = wl_v4f{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int c = lane + 64 * t;
          const int row = c >> 3, sg = c & 7;  // 8 segments of 16 bytes per 64-column row
          const int mm = wave_m0 + mi * 16 + row;
          const u32x4 v = *reinterpret_cast<const u32x4*>(stage + row * PITCH + sg * 16);
          if (mm < a.M) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(reinterpret_cast<uint8_t*>(out) + ((size_t)mm * out_n + wave_n0) * 2 + sg * 16));
        }
      }
    } else {
    // ---- epilogue: 16 rows of the wave at a time through its own staging rows (no block barrier: the slots belong to the next
      // tile's images already). Lane l holds, for tile (mi, nj), register t: row m = 16 mi + l % 16, column n = 16 nj + 4 (l / 16) + t.
      // Whole tiles (wq_takes_4w): every row and column exists, rows are 16-byte aligned.
      {
        bf16_t* out = static_cast<bf16_t*>(seg == 0 ? a.out : seg == 1 ? a.seg_out[0] : a.seg_out[1]);
        const int out_n = seg_rows(seg);
        constexpr int PITCH = W4_PITCH_PLAIN;
        const int wave_n0 = n0 + wn * 128, wave_m0 = m0 + wm * 128;
        // (the lane's staging and store addresses are formed HERE, from an opaque copy of the lane id: hoisted above the tile loop they sit
        // in ~20 registers across a K-loop that has 192 fragment registers live)
        uint32_t lane = threadIdx.x & 63u;
        asm volatile("" : "+v"(lane));
        const uint32_t r16 = lane & 15, g4 = lane >> 4;
  #pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
  #pragma unroll
          for (int nj = 0; nj < 8; ++nj) {
            const int nb = nj * 16 + 4 * g4;
            u32x2 pk;
            pk.x = pack2<bf16_t>(acc[mi][nj][0], acc[mi][nj][1]);
            pk.y = pack2<bf16_t>(acc[mi][nj][2], acc[mi][nj][3]);
            *reinterpret_cast<u32x2*>(stage + r16 * PITCH + nb * 2) = pk;
            acc[mi][nj] = wl_v4f{0.f, 0.f, 0.f, 0.f};
          }
          // (no wait between the writes and the reads below, nor before the next row tile's writes: a wave's LDS instructions
          // execute in order, and the rows are this wave's own)
  #pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int c = lane + 64 * t;
            const int row = c >> 4, sg = c & 15;  // 16 segments of 16 bytes per 128-column row
            const int mm = wave_m0 + mi * 16 + row;
            const u32x4 v = *reinterpret_cast<const u32x4*>(stage + row * PITCH + sg * 16);
            // non-temporal, as in ffq_linear.hip: the output must not push the operand panels out of L2
            if (mm < a.M) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(reinterpret_cast<uint8_t*>(out) + ((size_t)mm * out_n + wave_n0) * 2 + sg * 16));
          }
        }
      }
    }
    m0 = nm0; n0 = nn0; seg = nseg;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the trailing requests of the stream must not outlive the block's LDS
#undef FFQ_W4_MFMA
#undef FFQ_W4_READ
}

}  // namespace ffq

using namespace ffq;

// 1 if ffq_linear_wq covers the problem with the MFMA kernel, 0 if the caller has to dequantize and use a float GEMM.
// `w_dt`: FFQ_I8 = one code per byte (pack_block = 0); FFQ_U8 = packed 4-bit codes as ffq_pack_int4 writes them with a
// power-of-two `pack_block` >= 32.
extern "C" int ffq_linear_wq_supported(int x_dt, int w_dt, int out_dt, int64_t M, int64_t N, int64_t K, int64_t group, int64_t pack_block) {
  if (x_dt != FFQ_BF16 || !(out_dt == FFQ_BF16 || out_dt == FFQ_F32)) return 0;
  if (M <= 0 || N <= 0 || K < 2 * WL_BK || K % WL_BK != 0) return 0;
  if (group <= 0 || K % group != 0) return 0;
  if (group != K && group % WL_BK != 0) return 0;
  if (M > INT32_MAX || N > INT32_MAX || K > INT32_MAX || (uint64_t)256 * (uint64_t)K * 2u + 128u >= (1ull << 32)) return 0;
  if (w_dt == FFQ_I8) return pack_block == 0;
  if (w_dt == FFQ_U8) return pack_block >= 32 && (pack_block & (pack_block - 1)) == 0 && K % pack_block == 0;
  return 0;
}

// ---- the launch plan: K slices per tile ("split-K") when a launch has fewer tiles than the chip has CUs ----------------------------
// The kernel is one persistent block per CU on 256 x 256 output tiles; 2048 tokens x a 4096-wide projection are 128 tiles, a k/v
// projection 32 (round 3 sent everything below 4096 tokens to the vendor's GEMM for that reason). A tile's K range is cut
// into `split` slices, each (tile, slice) is a work unit on its own CU, and the units of a tile exchange their fp32 partial sums
// (kernel epilogue). All units must be resident at once (they wait for each other), so a split is admitted only while
// tiles * split <= CUs of the device — one round, one unit per block. The choice is a pure function of (M, N, K, mode, CU count):
// the fp32 summation order, hence the result's last bit, depends on it. Cost model in units of one 64-deep super-step
// (tools/wq_split_sweep.py): steps(S) + exchange, the exchange ~ a fixed synchronisation cost + the slab traffic of a unit,
// which scales with the rows of the tile that exist.
int ffq::wq_cus() {
  static int cached[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  int n = __atomic_load_n(&cached[dev], __ATOMIC_RELAXED);
  if (n == 0) {
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    __atomic_store_n(&cached[dev], n, __ATOMIC_RELAXED);
  }
  return n;
}

static int64_t wq_tiles(int64_t M, int64_t N, bool mlp) {
  return ((M + WL_BM - 1) / WL_BM) * (mlp ? N / 128 : (N + WL_BN - 1) / WL_BN);
}

// the tiles of the last, partly filled round of the persistent walk (all tiles when there are fewer than CUs): the ones a split
// applies to. Whole rounds before it keep every CU busy without any exchange.
static int64_t wq_tail_tiles(int64_t M, int64_t N, bool mlp) {
  const int64_t tiles = wq_tiles(M, N, mlp);
  return tiles > 0 ? tiles % wq_cus() : 0;
}

static int wq_max_split(int64_t M, int64_t N, int64_t K, bool mlp) {
  const int64_t tail = wq_tail_tiles(M, N, mlp);
  if (tail <= 0) return 1;
  const int64_t by_cus = wq_cus() / tail, by_k = (K / WL_BK) / 2;
  const int64_t most = by_cus < by_k ? by_cus : by_k;
  return most < 1 ? 1 : most > 32 ? 32 : (int)most;
}

static int wq_split(int64_t M, int64_t N, int64_t K, bool mlp) {
  const int64_t ksuper = K / WL_BK;
  const int most = wq_max_split(M, N, K, mlp);
  const double rows = (double)(M < WL_BM ? M : WL_BM) / WL_BM;
  double best_cost = 0.0;
  int best = 1;
  for (int S = 1; S <= most; ++S) {
    if (S > 1 && ksuper / S < 4) break;  // the code stream looks two super-steps ahead: keep slices comfortably longer
    const double steps = (double)((ksuper + S - 1) / S);
    const double cost = steps + (S > 1 ? 6.0 + 6.0 * rows * (double)(S - 1) / S + 0.25 * S : 0.0);
    if (S == 1 || cost < best_cost * 0.97) { best_cost = cost; best = S; }  // a larger split has to pay for itself
  }
  return best;
}

// Which form a plain launch of M rows takes (ffq_wq.h): the skinny form up to FFQ_MID_MIN_M - 1 rows, the 128-column tiles of
// ffq_wmid.hip up to 512, the 256-row tiles beyond (and for every gate+up+SiLU*up launch, and under the test hook). The skinny form
// declines some storage forms (packing blocks other than 128, K % 128 != 0): such a launch takes the 128-column tiles, which cover
// everything ffq_linear_wq_supported() admits — so the scratch figures below answer for BOTH wherever the skinny form is preferred
// (a GGUF-packed decode step got the skinny plan's scratch, often none, and ran the 256-row tiles without a split).
enum { WQ_FORM_TILES = 0, WQ_FORM_SKINNY = 1, WQ_FORM_MID = 2 };
// the skinny form up to 16 rows, and up to 32 for the narrow projections (q/o at 32 rows 14.8 us against 16.6 on the 128-column tiles, k/v
// 14.8 against 15.5, down_proj 27.1 against 32.5; gate/up the other way round: 32.9 against 28.4 — profiles/r06_wq_rows_sweep.txt)
static bool wq_prefers_skinny(int64_t M, int64_t N, int64_t K) {
  return (M < FFQ_MID_MIN_M || (M <= 32 && N < 8192)) && wq_skinny_tickets(M, 128, K) > 0;
}
static int wq_plan_form(int64_t M, int64_t N, int64_t K, int mlp) {
  if (mlp || generic_kernels_forced()) return WQ_FORM_TILES;
  if (wq_prefers_skinny(M, N, K)) return WQ_FORM_SKINNY;
  if (wq_mid_prefers(M, N, K)) return WQ_FORM_MID;
  return wq_skinny_tickets(M, 128, K) > 0 ? WQ_FORM_SKINNY : WQ_FORM_TILES;
}

extern "C" int64_t ffq_linear_wq_split(int64_t M, int64_t N, int64_t K, int mlp) {
  if (M <= 0 || N <= 0 || K < 2 * WL_BK) return 1;
  switch (wq_plan_form(M, N, K, mlp)) {
    case WQ_FORM_SKINNY: return wq_skinny_split(M, N, K);
    case WQ_FORM_MID: return wq_mid_split(M, N, K);
    default: return wq_split(M, N, K, mlp != 0);
  }
}

// int32 counters the split-K exchange needs (0 = none): zero before the first launch that uses them, left zero by every launch — a
// caller keeps ONE zeroed buffer per stream and never touches it. An upper bound over the forms the launch can take.
extern "C" int64_t ffq_linear_wq_tickets(int64_t M, int64_t N, int64_t K, int mlp) {
  if (M <= 0 || N <= 0 || K < 2 * WL_BK) return 0;
  int64_t n = 2 * wq_tail_tiles(M, N, mlp != 0);
  if (!mlp) {
    const int64_t sk = wq_skinny_tickets(M, N, K), md = wq_mid_tickets(M, N, K);
    n = n > sk ? n : sk;
    n = n > md ? n : md;
  }
  return n;
}

// bytes of partial-sum slabs a launch with `split` K slices needs at the front of its workspace (0: none). `split` = the value of
// ffq_linear_wq_split (the plan) also covers the plan of the form that takes over where the preferred one declines the storage.
extern "C" size_t ffq_linear_wq_slab_bytes(int64_t M, int64_t N, int64_t K, int mlp, int64_t split) {
  if (M <= 0 || N <= 0 || K < 2 * WL_BK) return 0;
  switch (wq_plan_form(M, N, K, mlp)) {
    case WQ_FORM_SKINNY: {
      size_t bytes = wq_skinny_slab_bytes(M, N, K, split);
      if (wq_mid_prefers(M, N, K)) {
        const size_t md = wq_mid_slab_bytes(M, N, K, split == wq_skinny_split(M, N, K) ? wq_mid_split(M, N, K) : split);
        bytes = bytes > md ? bytes : md;
      }
      return bytes;
    }
    case WQ_FORM_MID: return wq_mid_slab_bytes(M, N, K, split);
    default: return split <= 1 ? 0 : (size_t)wq_tail_tiles(M, N, mlp != 0) * (size_t)split * WL_UNIT_SLAB;
  }
}

static size_t wq_slab_bytes(int64_t M, int64_t N, int split, bool mlp) {
  return split > 1 ? (size_t)wq_tail_tiles(M, N, mlp) * (size_t)split * WL_UNIT_SLAB : 0;
}

// Where the two-pass form (A2 of the whole weight into a bf16 image, then the one-wave-per-SIMD GEMM) is the plan of a plain launch: from
// 4096 tokens on, and from 1536 on where the 256 x 256 tiles make at least 144 — A2 costs N K 3 bytes whatever M is (26 % of the GEMM
// at 1536 tokens), the one-pass loop is ~25 % slower than the image loop, and below ~half a round of tiles the one-pass form's K split
// is what fills the chip. Round 6 (profiles/r06_wq_twopass_sweep.txt, 8B shapes): q/k/v as one launch (N = 6144) at 1536 / 2048 / 3072
// tokens 0.85 / 0.89 / 0.97 of the one-pass time, o_proj / down_proj at 3072 0.81 / 0.79; at 2048 (128 tiles) down_proj is 1.09: one-pass.
static bool wq_two_pass_planned(int64_t M, int64_t N) {
  if (M >= WL_TWO_PASS_MIN_TOKENS) return true;
  return M >= 1536 && ((M + WL_BM - 1) / WL_BM) * ((N + WL_BN - 1) / WL_BN) >= 144;
}

// workspace: [split-K slabs of the library's plan | bf16 image(s) of the two-pass form (A2 of the whole weight, then the GEMM on
// that image): N * K * 2 bytes where wq_two_pass_planned]. The caller may pass less (or NULL): the launch then runs without the part
// that does not fit (no split / conversion inside the GEMM) — never fails for lack of scratch.
extern "C" size_t ffq_linear_wq_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  if (M <= 0 || N <= 0 || K < 2 * WL_BK) return 0;
  if (wq_plan_form(M, N, K, 0) != WQ_FORM_TILES) return ffq_linear_wq_slab_bytes(M, N, K, 0, ffq_linear_wq_split(M, N, K, 0));
  return wq_slab_bytes(M, N, wq_split(M, N, K, false), false) + (wq_two_pass_planned(M, N) ? (size_t)N * (size_t)K * 2u : 0);
}

// resolves (requested split, scratch on offer) into what the launch uses; returns the bytes the slabs take at the front of `workspace`
static size_t wq_resolve_split(WLinearArgs& a, int64_t split_request, bool mlp, void* workspace, size_t workspace_bytes, int32_t* tickets, int* rc) {
  *rc = FFQ_OK;
  const int most = wq_max_split(a.M, a.N, a.K, mlp);
  if (split_request > most) {
    *rc = fail(FFQ_ERR_ARG, "weight-only linear: split %lld exceeds %d (the units of the last round must be resident at once: tail tiles * split <= CUs, K / 64 >= 2 * split)", (long long)split_request, most);
    return 0;
  }
  int split = split_request > 0 ? (int)split_request : wq_split(a.M, a.N, a.K, mlp);
  size_t slab = wq_slab_bytes(a.M, a.N, split, mlp);
  if (split > 1 && (!tickets || !workspace || workspace_bytes < slab || !aligned16(workspace))) {
    if (split_request > 1) { *rc = fail(FFQ_ERR_ARG, "weight-only linear: split %d needs %zu bytes of workspace and a ticket buffer", split, slab); return 0; }
    split = 1; slab = 0;  // the plan is a preference: without scratch every tile is one unit
  }
  const int64_t tiles = wq_tiles(a.M, a.N, mlp);
  a.split = split;
  a.full_tiles = (int)(split > 1 ? tiles - wq_tail_tiles(a.M, a.N, mlp) : tiles);
  a.slabs = split > 1 ? static_cast<float*>(workspace) : nullptr;
  a.tickets = split > 1 ? tickets : nullptr;
  a.abandon_test = splitk_abandon_forced() ? 1 : 0;
  return slab;
}

template <int BKIND, bool GROUPED, bool OFFSET, typename TOut, bool MLP = false>
static void wq_launch(const WLinearArgs& a, hipStream_t s) {
  const int total = a.tiles_m * a.tiles_n;
  const int64_t units = (int64_t)a.full_tiles + (int64_t)(total - a.full_tiles) * a.split;  // the split tail: <= CUs units (wq_max_split)
  const int cus = wq_cus();
  const unsigned grid = (unsigned)(units < cus ? units : cus);
  const size_t lds_bytes = (size_t)2 * WL_SLOT + (MLP ? kSiluBytes : 0);
  static uint64_t attr_set = 0;
  ensure_dynamic_lds(&attr_set, reinterpret_cast<const void*>(&wq_gemm256_kernel<BKIND, GROUPED, OFFSET, TOut, MLP>), (int)lds_bytes);
  wq_gemm256_kernel<BKIND, GROUPED, OFFSET, TOut, MLP><<<grid, 512, lds_bytes, s>>>(a, total);
}

// the one-wave-per-SIMD kernel: plain launches, whole tiles only (no split tail), an even number of super-steps, bf16 output
template <bool MLP>
static void wq_launch4w(const WLinearArgs& a, hipStream_t s) {
  const int total = a.tiles_m * a.tiles_n;
  const int cus = wq_cus();
  const unsigned grid = (unsigned)(total < cus ? total : cus);
  const size_t lds_bytes = (size_t)2 * WL_SLOT + (size_t)4 * 16 * (MLP ? W4_PITCH_MLP : W4_PITCH_PLAIN) + (MLP ? kSiluBytes : 0);
  static uint64_t attr_set = 0;
  ensure_dynamic_lds(&attr_set, reinterpret_cast<const void*>(&wq_gemm4w_kernel<MLP>), (int)lds_bytes);
  wq_gemm4w_kernel<MLP><<<grid, 256, lds_bytes, s>>>(a, total);
}

static bool wq_takes_4w(const WLinearArgs& a, bool mlp = false) {
#ifdef FFQ_WL_NO_4W  // A/B builds (tools/build_variant.sh)
  return false;
#else
#ifdef FFQ_WL_NO_4W_MLP  // A/B builds: the gate + up + SiLU*up launch on the 8-wave kernel
  if (mlp) return false;
#endif
  if (generic_kernels_forced()) return false;  // tests: the 8-wave kernel on the same operands (ffq_force_generic_kernels)
  // whole tiles of the WEIGHT only: its pieces' row offsets travel in the buffer instruction's SCALAR offset, which the hardware's range
  // check does not see (only the per-lane offset is compared with the descriptor's extent) — rows past a ragged edge would be read from
  // beyond the tensor. The activation pieces carry their rows in the per-lane offset: any token count. Ragged weight matrices take the
  // 8-wave kernel, which clamps its source rows.
  if (mlp) {
    if (a.N % 128 != 0) return false;  // tiles of 128 output columns (128 gate + 128 up rows)
  } else {
    for (int i = 0; i < 3; ++i)
      if (a.seg_n[i] % WL_BN != 0) return false;
  }
  return a.split == 1 && a.out_dt == FFQ_BF16 && a.bias == nullptr && (a.K / WL_BK) % 2 == 0 && a.K >= 4 * WL_BK;
#endif
}

template <int BKIND, bool MLP = false>
static void wq_dispatch(const WLinearArgs& a, bool grouped, bool offset, hipStream_t s) {
  if constexpr (BKIND == WL_B_BF16) {
    if (wq_takes_4w(a, MLP)) wq_launch4w<MLP>(a, s);
    else if (MLP || a.out_dt == FFQ_BF16) wq_launch<BKIND, false, false, bf16_t, MLP>(a, s);
    else if constexpr (!MLP) wq_launch<BKIND, false, false, float, false>(a, s);
  } else {
#define FFQ_WL_T(G, O) do { if (MLP || a.out_dt == FFQ_BF16) wq_launch<BKIND, G, O, bf16_t, MLP>(a, s); else if constexpr (!MLP) wq_launch<BKIND, G, O, float, false>(a, s); } while (0)
    if (grouped) { if (offset) FFQ_WL_T(true, true); else FFQ_WL_T(true, false); }
    else { if (offset) FFQ_WL_T(false, true); else FFQ_WL_T(false, false); }
#undef FFQ_WL_T
  }
}

// `count` weight matrices side by side along N on the same activations (count == 1: ffq_linear_wq)
static int wq_linear_impl(const void* x, int x_dt, int count, const void* const* w_codes, int w_dt, int64_t pack_block, const float* const* w_scale,
                          const float* const* w_offset, int per_row, int64_t group, const void* bias, int bias_dt, void* const* outs,
                          int out_dt, int64_t M, const int64_t* Ns, int64_t K, void* workspace, size_t workspace_bytes, int32_t* tickets,
                          int64_t split, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  int64_t N = 0;
  for (int i = 0; i < count; ++i) N += Ns[i];
  const int64_t groups = K / group;

  WLinearArgs a;
  a.x = static_cast<const uint8_t*>(x);
  a.w = static_cast<const uint8_t*>(w_codes[0]);
  a.w_scale = w_scale[0]; a.w_offset = w_offset[0];
  a.w2 = nullptr; a.w_scale2 = nullptr; a.w_offset2 = nullptr;
  a.bias = bias; a.bias_dt = bias_dt;
  a.out = outs[0]; a.out_dt = out_dt;
  a.M = (int)M; a.N = (int)N; a.K = (int)K;
  a.seg_n[0] = (int)Ns[0]; a.seg_n[1] = a.seg_n[2] = 0;
  a.seg_tile[0] = a.seg_tile[1] = INT32_MAX;
  for (int i = 0; i < 2; ++i) { a.seg_w[i] = nullptr; a.seg_scale[i] = nullptr; a.seg_offset[i] = nullptr; a.seg_out[i] = nullptr; }
  int64_t tile_edge = 0;
  for (int i = 1; i < count; ++i) {
    tile_edge += Ns[i - 1] / WL_BN;
    a.seg_tile[i - 1] = (int)tile_edge;
    a.seg_w[i - 1] = static_cast<const uint8_t*>(w_codes[i]);
    a.seg_scale[i - 1] = w_scale[i]; a.seg_offset[i - 1] = w_offset[i];
    a.seg_out[i - 1] = outs[i];
    a.seg_n[i] = (int)Ns[i];
  }
  a.groups = (int)groups;
  a.steps_per_group = (int)(group / WL_BK);
  a.per_row = per_row;
  a.pack_shift = 0;
  for (int64_t b = pack_block; b > 1; b >>= 1) ++a.pack_shift;
  a.tiles_m = (int)((M + WL_BM - 1) / WL_BM);
  a.tiles_n = (int)((N + WL_BN - 1) / WL_BN);
  a.group_m = K >= 4096 ? 4 : WL_GROUP_M;  // row tiles whose A panels (256 x 2 K bytes each) a group's column tiles share in their XCD's L2
  a.group_cols = 0;
#ifdef FFQ_EXPERIMENTS  // tuning builds only (tools/): the shipped library reads no environment
  if (const char* gm = getenv("FFQ_WQ_GROUP_M")) a.group_m = atoi(gm);
  if (const char* gc = getenv("FFQ_WQ_GROUP_COLS")) a.group_cols = atoi(gc);
#endif
  const bool grouped = groups > 1, offset = w_offset[0] != nullptr;
  // few rows: the contraction is a stream over the codes, bounded by HBM — 16-row MFMA tiles, no padding to 256 rows (ffq_wskinny.hip)
  if (wq_prefers_skinny(M, N, K) && wq_skinny_applies(a, pack_block)) return wq_skinny_launch(a, w_dt, pack_block, group, split, workspace, workspace_bytes, tickets, s);
  // a few hundred rows (and the storage forms the skinny kernel declines): 128-column tiles, codes converted once per block (ffq_wmid.hip)
  if (wq_mid_prefers(M, N, K) && wq_mid_applies(a)) return wq_mid_launch(a, w_dt, group, split, workspace, workspace_bytes, tickets, s);
  int rc_split;
  const size_t slab_bytes = wq_resolve_split(a, split, false, workspace, workspace_bytes, tickets, &rc_split);
  if (rc_split != FFQ_OK) return rc_split;
  workspace = workspace ? static_cast<uint8_t*>(workspace) + slab_bytes : nullptr;  // the image(s) (if any) lie behind the slabs
  workspace_bytes = workspace_bytes > slab_bytes ? workspace_bytes - slab_bytes : 0;
  const size_t image_bytes = (size_t)N * (size_t)K * 2u;
  if (workspace && workspace_bytes >= image_bytes && aligned16(workspace)) {  // the caller offered the image's scratch (ffq_linear_wq_workspace_bytes: from 4096 tokens on)
    // two-pass form: A2 of the whole weight once (3 or 2.5 B/elem, ~2 % of the GEMM at 16 k tokens), then the GEMM with both
    // operands by LDS-DMA — no conversion work per row tile
    uint8_t* image = static_cast<uint8_t*>(workspace);
    int rc = FFQ_OK;
    const uint8_t* images[3] = {nullptr, nullptr, nullptr};
    for (int i = 0; i < count && rc == FFQ_OK; ++i) {
      ffq_tiling t;
      t.ndim = 2;
      t.shape[0] = Ns[i]; t.shape[1] = K;
      t.tile[0] = per_row ? 1 : Ns[i]; t.tile[1] = per_row ? group : K;
      const int64_t numel = per_row ? Ns[i] * groups : 1;
      if (w_dt == FFQ_U8)
        rc = ffq_unpack_dequantize_int4(static_cast<const uint8_t*>(w_codes[i]), w_scale[i], numel, w_offset[i], w_offset[i] ? numel : 0, &t, pack_block, image, FFQ_BF16, stream);
      else
        rc = ffq_dequantize_by_tile(w_codes[i], FFQ_I8, w_scale[i], FFQ_F32, numel, w_offset[i], FFQ_F32, w_offset[i] ? numel : 0, &t, image, FFQ_BF16, stream);
      images[i] = image;
      image += (size_t)Ns[i] * (size_t)K * 2u;
    }
    if (rc == FFQ_OK) {
      a.w = images[0];
      for (int i = 1; i < count; ++i) a.seg_w[i - 1] = images[i];
      // Activations that do not fit the 256 MiB Infinity Cache (down_proj at 16 k tokens: 470 MB) must come from HBM ONCE: groups of 8
      // COLUMN tiles x all row tiles make the XCDs stream the same row panels at the same time (one HBM read, the other XCDs hit the
      // Infinity Cache) and keep their column panels resident, where row groups give every XCD its own rows and read them once per
      // batch of column tiles (round-4 A/B on one box, one-wave-per-SIMD kernel: down_proj 1.36 -> 1.42 PFLOP/s; q/o ±0, gate/up -1.3 %)
      if (count == 1 && (size_t)M * (size_t)K * 2u > ((size_t)200 << 20) && wq_takes_4w(a)) { a.group_cols = 1; a.group_m = 8; }
      wq_dispatch<WL_B_BF16>(a, false, false, s);
      return check_launch("wq_gemm256_kernel (bf16 image)");
    }
    if (rc != FFQ_ERR_DTYPE) return rc;  // a tiling the stand-alone dequantize kernels decline: the one-pass kernel below covers it
  }
  if (w_dt == FFQ_U8) wq_dispatch<WL_B_I4>(a, grouped, offset, s);
  else wq_dispatch<WL_B_I8>(a, grouped, offset, s);
  return check_launch("wq_gemm256_kernel");
}

extern "C" int ffq_linear_wq(const void* x, int x_dt, const void* w_codes, int w_dt, int64_t pack_block, const float* w_scale,
                             const float* w_offset, int64_t scale_numel, int64_t group, const void* bias, int bias_dt, void* out,
                             int out_dt, int64_t M, int64_t N, int64_t K, void* workspace, size_t workspace_bytes, int32_t* tickets,
                             int64_t split, void* stream) {
  if (M < 0 || N < 0 || K < 0 || split < 0) return fail(FFQ_ERR_ARG, "negative extent");
  if (M == 0 || N == 0) return FFQ_OK;
  if (!x || !w_codes || !w_scale || !out) return fail(FFQ_ERR_ARG, "NULL buffer");
  if (!ffq_linear_wq_supported(x_dt, w_dt, out_dt, M, N, K, group, pack_block))
    return fail(FFQ_ERR_DTYPE, "weight-only linear: needs bf16 activations, int8-container or packed 4-bit codes, bf16 / f32 output, K %% 64 == 0, K >= 128 and groups of a multiple of 64 input channels");
  if (!aligned16(x) || !aligned16(w_codes) || !aligned16(out)) return fail(FFQ_ERR_DTYPE, "weight-only linear needs 16-byte aligned buffers");
  if (bias && !dt_valid(bias_dt)) return fail(FFQ_ERR_ARG, "bad bias dtype");
  const int64_t groups = K / group;
  if (!(scale_numel == 1 || scale_numel == N * groups))
    return fail(FFQ_ERR_PARAM_NUMEL, "weight-only linear: %lld parameters for %lld x %lld tiles", (long long)scale_numel, (long long)N, (long long)groups);
  if (scale_numel == 1 && groups != 1) return fail(FFQ_ERR_PARAM_NUMEL, "one parameter pair needs group == K");
  return wq_linear_impl(x, x_dt, 1, &w_codes, w_dt, pack_block, &w_scale, &w_offset, scale_numel != 1, group, bias, bias_dt, &out, out_dt, M, &N, K,
                        workspace, workspace_bytes, tickets, split, stream);
}

// Two or three weight matrices on the SAME activations in one launch (q_proj / k_proj / v_proj of an attention block: three
// QuantizedLinear modules reading one hidden state, nn/linear.py:32-39 three times): outs[i] = F.linear(x, dequantize(w_i)) exactly as
// ffq_linear_wq computes it for the concatenated weight — one tile walk over all the column tiles, so a k/v projection (1024
// columns = 4 tiles per row tile) no longer runs alone on a corner of the chip. All matrices share dtype, packing, group size,
// granularity kind (`per_row`: one pair per (row, group), else one pair per matrix) and the presence of offsets; every matrix but
// the last has a multiple of 256 rows. Workspace / tickets / split as ffq_linear_wq with N = the sum of the rows.
extern "C" int ffq_linear_wq_multi(const void* x, int x_dt, int count, const void* const* w_codes, int w_dt, int64_t pack_block,
                                   const float* const* w_scale, const float* const* w_offset, int per_row, int64_t group, void* const* outs,
                                   int out_dt, int64_t M, const int64_t* Ns, int64_t K, void* workspace, size_t workspace_bytes, int32_t* tickets,
                                   int64_t split, void* stream) {
  if (count < 1 || count > 3 || !w_codes || !w_scale || !w_offset || !outs || !Ns) return fail(FFQ_ERR_ARG, "1 to 3 weight matrices");
  if (M < 0 || K < 0 || split < 0) return fail(FFQ_ERR_ARG, "negative extent");
  int64_t N = 0;
  for (int i = 0; i < count; ++i) {
    if (Ns[i] <= 0) return fail(FFQ_ERR_ARG, "empty weight matrix");
    if (i + 1 < count && Ns[i] % WL_BN != 0) return fail(FFQ_ERR_DTYPE, "every weight matrix but the last needs a multiple of 256 rows");
    if (!w_codes[i] || !w_scale[i] || !outs[i]) return fail(FFQ_ERR_ARG, "NULL buffer");
    if ((w_offset[i] == nullptr) != (w_offset[0] == nullptr)) return fail(FFQ_ERR_ARG, "offsets for all weight matrices or for none");
    if (!aligned16(w_codes[i]) || !aligned16(outs[i])) return fail(FFQ_ERR_DTYPE, "weight-only linear needs 16-byte aligned buffers");
    N += Ns[i];
  }
  if (M == 0) return FFQ_OK;
  if (!x || !aligned16(x)) return fail(FFQ_ERR_ARG, "NULL or misaligned activations");
  if (!ffq_linear_wq_supported(x_dt, w_dt, out_dt, M, N, K, group, pack_block))
    return fail(FFQ_ERR_DTYPE, "weight-only linear: needs bf16 activations, int8-container or packed 4-bit codes, bf16 / f32 output, K %% 64 == 0, K >= 128 and groups of a multiple of 64 input channels");
  if (!per_row && group != K) return fail(FFQ_ERR_PARAM_NUMEL, "one parameter pair per matrix needs group == K");
  return wq_linear_impl(x, x_dt, count, w_codes, w_dt, pack_block, w_scale, w_offset, per_row, group, nullptr, 0, outs, out_dt, M, Ns, K, workspace,
                        workspace_bytes, tickets, split, stream);
}

// ---- gate_proj + up_proj + SiLU * up of a weight-only quantized MLP in one launch ---------------------------------------------------
// workspace: [split-K slabs of the library's plan | the bf16 images of BOTH matrices (two-pass form, from 4096 tokens on)]
extern "C" size_t ffq_mlp_gate_up_wq_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  if (M <= 0 || N <= 0 || N % 128 != 0 || K < 2 * WL_BK) return 0;
  return wq_slab_bytes(M, N, wq_split(M, N, K, true), true) + (M >= WL_TWO_PASS_MIN_TOKENS ? (size_t)2 * (size_t)N * (size_t)K * 2u : 0);
}

extern "C" int ffq_mlp_gate_up_wq(const void* x, int x_dt, const void* gate_codes, const void* up_codes, int w_dt, int64_t pack_block,
                                  const float* gate_scale, const float* gate_offset, const float* up_scale, const float* up_offset,
                                  int64_t scale_numel, int64_t group, void* out, int64_t M, int64_t N, int64_t K, void* workspace,
                                  size_t workspace_bytes, int32_t* tickets, int64_t split, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (M < 0 || N < 0 || K < 0 || split < 0) return fail(FFQ_ERR_ARG, "negative extent");
  if (M == 0 || N == 0) return FFQ_OK;
  if (!x || !gate_codes || !up_codes || !gate_scale || !up_scale || !out) return fail(FFQ_ERR_ARG, "NULL buffer");
  if ((gate_offset == nullptr) != (up_offset == nullptr)) return fail(FFQ_ERR_ARG, "gate and up need offsets both or neither");
  if (N % 128 != 0 || !ffq_linear_wq_supported(x_dt, w_dt, FFQ_BF16, M, N, K, group, pack_block))
    return fail(FFQ_ERR_DTYPE, "fused weight-only gate/up: needs N %% 128 == 0 and what ffq_linear_wq needs");
  if (!aligned16(x) || !aligned16(gate_codes) || !aligned16(up_codes) || !aligned16(out)) return fail(FFQ_ERR_DTYPE, "weight-only linear needs 16-byte aligned buffers");
  const int64_t groups = K / group;
  if (!(scale_numel == 1 || scale_numel == N * groups))
    return fail(FFQ_ERR_PARAM_NUMEL, "weight-only linear: %lld parameters for %lld x %lld tiles", (long long)scale_numel, (long long)N, (long long)groups);
  if (scale_numel == 1 && groups != 1) return fail(FFQ_ERR_PARAM_NUMEL, "one parameter pair needs group == K");

  WLinearArgs a;
  a.x = static_cast<const uint8_t*>(x);
  a.w = static_cast<const uint8_t*>(gate_codes); a.w2 = static_cast<const uint8_t*>(up_codes);
  a.w_scale = gate_scale; a.w_offset = gate_offset;
  a.w_scale2 = up_scale; a.w_offset2 = up_offset;
  a.bias = nullptr; a.bias_dt = 0;
  a.out = out; a.out_dt = FFQ_BF16;
  a.M = (int)M; a.N = (int)N; a.K = (int)K;
  a.seg_n[0] = (int)N; a.seg_n[1] = a.seg_n[2] = 0;
  a.seg_tile[0] = a.seg_tile[1] = INT32_MAX;
  for (int i = 0; i < 2; ++i) { a.seg_w[i] = nullptr; a.seg_scale[i] = nullptr; a.seg_offset[i] = nullptr; a.seg_out[i] = nullptr; }
  a.groups = (int)groups;
  a.steps_per_group = (int)(group / WL_BK);
  a.per_row = scale_numel != 1;
  a.pack_shift = 0;
  for (int64_t b = pack_block; b > 1; b >>= 1) ++a.pack_shift;
  a.tiles_m = (int)((M + WL_BM - 1) / WL_BM);
  a.tiles_n = (int)(N / 128);
  a.group_m = K >= 4096 ? 4 : WL_GROUP_M;
  // both weight images together (2 N K x 2 B) against the activations (M K x 2 B): re-read the smaller one per group
  a.group_cols = 2 * N > M ? 1 : 0;
#ifdef FFQ_EXPERIMENTS
  if (const char* gm = getenv("FFQ_WQ_GROUP_M")) a.group_m = atoi(gm);
  if (const char* gc = getenv("FFQ_WQ_GROUP_COLS")) a.group_cols = atoi(gc);
#endif
  const bool grouped = groups > 1, offset = gate_offset != nullptr;
  int rc_split;
  const size_t slab_bytes = wq_resolve_split(a, split, true, workspace, workspace_bytes, tickets, &rc_split);
  if (rc_split != FFQ_OK) return rc_split;
  workspace = workspace ? static_cast<uint8_t*>(workspace) + slab_bytes : nullptr;
  workspace_bytes = workspace_bytes > slab_bytes ? workspace_bytes - slab_bytes : 0;

  const size_t image_bytes = (size_t)N * (size_t)K * 2u;
  if (workspace && workspace_bytes >= 2 * image_bytes && aligned16(workspace)) {
    ffq_tiling t;
    t.ndim = 2;
    t.shape[0] = N; t.shape[1] = K;
    t.tile[0] = scale_numel == 1 ? N : 1; t.tile[1] = scale_numel == 1 ? K : group;
    uint8_t* image = static_cast<uint8_t*>(workspace);
    int rc = FFQ_OK;
    for (int which = 0; which < 2 && rc == FFQ_OK; ++which) {
      const void* codes = which ? up_codes : gate_codes;
      const float* sc = which ? up_scale : gate_scale;
      const float* of = which ? up_offset : gate_offset;
      if (w_dt == FFQ_U8)
        rc = ffq_unpack_dequantize_int4(static_cast<const uint8_t*>(codes), sc, scale_numel, of, of ? scale_numel : 0, &t, pack_block, image + which * image_bytes, FFQ_BF16, stream);
      else
        rc = ffq_dequantize_by_tile(codes, FFQ_I8, sc, FFQ_F32, scale_numel, of, FFQ_F32, of ? scale_numel : 0, &t, image + which * image_bytes, FFQ_BF16, stream);
    }
    if (rc == FFQ_OK) {
      a.w = image; a.w2 = image + image_bytes;
      wq_dispatch<WL_B_BF16, true>(a, false, false, s);
      return check_launch("wq_gemm256_kernel (mlp mode, bf16 images)");
    }
    if (rc != FFQ_ERR_DTYPE) return rc;
  }
  if (w_dt == FFQ_U8) wq_dispatch<WL_B_I4, true>(a, grouped, offset, s);
  else wq_dispatch<WL_B_I8, true>(a, grouped, offset, s);
  return check_launch("wq_gemm256_kernel (mlp mode)");
}

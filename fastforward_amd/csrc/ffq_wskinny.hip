// ffq_wskinny.hip — the weight-only quantized linear for FEW token rows (1 <= M <= 128): decode steps and short prompts.
//
// Same contract as ffq_wlinear.hip (reference _gen/fallback.py:86-112: y = F.linear(x, dequantize(w)); the B operand of the bf16
// MFMA is bit for bit A2's bf16 value, fp32 accumulation, only the summation order is this kernel's own), different regime: with a
// handful of rows the contraction is a STREAM over the weight codes — 16.8 MB for a 4096 x 4096 projection, 58.7 MB for
// gate / up / down of Llama-3-8B — and the roofline is HBM, not the matrix cores. The 256 x 256-tile kernel pads such a problem
// to 256 rows and ran its codes at 7-17 % of the HBM rate (profiles/r04_wq_split_sweep.txt); here
//   * a block is 8 waves x 16 weight rows = 128 output columns over one slice of K; a wave's lanes (r = lane % 16, g = lane / 16)
//     load 16 code bytes of row r straight into registers (one instruction: 16 rows x 64 contiguous bytes; nothing of the
//     weight goes through LDS), convert them with A2's arithmetic (dequantize4 of ffq_wq.h — the function the big kernel uses)
//     and feed them to v_mfma_f32_16x16x32_bf16 as its first operand: bytes 0-7 are one MFMA's k-slice of the lane, bytes 8-15
//     the next one's (a contraction is invariant under any permutation of k applied to both operands, so the activations are
//     read in the same order and nothing is transposed). Packed nibbles (0.5 B per weight) are consumed as stored: a lane's
//     16 bytes are 16 codes of the low half of a packing block and 16 of the high half (export/stages/gguf/_packing.py:44-53);
//   * the activations of the block's k range ([16 MT rows] x KC columns, rows beyond M zero) are staged through LDS once per
//     chunk, double-buffered, and shared by the eight waves: HBM sees every weight byte once, L2 sees the activations
//     N / 128 times in total (BN = 128 >= 2 M keeps that below the weight bytes up to 64 rows);
//   * all code loads of the next two chunks are in flight while a chunk is computed (32 registers): a CU has 64 KiB on the way;
//   * K is cut into S slices across blocks so that the launch has at least one block per CU; a wave leaves its 16-column strip
//     of partial sums in a write-through slab, takes a ticket for that strip, and the LAST wave to arrive — whoever it is —
//     adds the S partials in slice order and writes the output: no block ever waits for another (the exchange of
//     ffq_wlinear.hip spins on co-resident peers), the summation order is a function of the plan alone (bit-reproducible), the
//     ticket words are zero before and after. Tickets are agent-scope RMWs; data goes out with sc1 write-through stores that are
//     drained (vmcnt(0)) before the ticket and comes back with sc1 loads, which bypass the reading CU's L1 and miss its L2 (a slab
//     line is written once and read once per launch; the L2 holds nothing of other XCDs' writes across a kernel boundary).
// Covered: everything ffq_linear_wq_supported() admits with M <= 128, plain launches (one to three weight matrices on the same
// activations), int8 containers and packed nibbles with packing block 128, per-tensor / per-channel / per-group
// parameters, bias, bf16 / f32 output. The gate+up+SiLU*up launch keeps the 256-row-tile kernel.
#include "ffq_wq.h"

#include <math.h>

#include <type_traits>

namespace ffq {

constexpr int SK_BN = 128;     // weight rows (output columns) per block
constexpr int SK_MAX_M = 128;  // token rows the skinny form covers
#ifndef FFQ_SK_ROWS_MAX_M
#define FFQ_SK_ROWS_MAX_M 32    // rows form up to this many token rows ... (A/B hooks: tools/build_variant.sh)
#endif
#ifndef FFQ_SK_ROWS_MAX_K
#define FFQ_SK_ROWS_MAX_K 4096  // ... of contractions up to this deep (16 rows: any depth)
#endif
#ifndef FFQ_SK_NT2_MIN_M
#define FFQ_SK_NT2_MIN_M 5      // rows form: two weight tiles per block from this many token rows
#endif
constexpr int SK_WAVES = 8;

struct SkinnyArgs {
  const uint8_t* x;
  const uint8_t* w[3]; const float* scale[3]; const float* offset[3]; void* out[3];
  int seg_n[3];          // rows of each weight matrix (0: absent)
  int seg_block[3];      // first n-block of matrices 1 and 2 (seg_block[0] = 0); INT32_MAX: absent
  const void* bias; int bias_dt;
  int out_dt;
  int M, K;
  int n_blocks;          // n-blocks over all matrices
  int groups;            // parameters per row along K (1: per channel / per tensor)
  FastDiv group_div;     // codes per group, as an exact magic-number divider
  int per_row;
  int pack_shift;        // WL_B_I4: log2(packing block)
  int chunks;            // K / KC
  int S;                 // K slices across blocks
  float* slabs;          // [n_block][S][wave][MT][64 lanes] x 16 B
  int* tickets;          // [n_block][wave]
};

template <int MT> struct SkinnyShape {
  static constexpr int KC = MT <= 4 ? 256 : 128;        // k values per chunk
  static constexpr int PITCH = KC * 2 + 16;             // bytes per activation row in LDS: conflict-free ds_read_b128
  static constexpr int STAGE = 16 * MT * PITCH;
  static constexpr int PIECES = MT * KC / 256;          // 16-byte activation pieces per thread and chunk (512 threads)
};

template <int BKIND, bool GROUPED, bool OFFSET, int MT>
__global__ __launch_bounds__(512, 2) void wq_skinny_kernel(SkinnyArgs a) {
  using Shape = SkinnyShape<MT>;
  constexpr int KC = Shape::KC, PITCH = Shape::PITCH;
  constexpr int ROW_BYTES_PER_CHUNK = BKIND == WL_B_I8 ? KC : KC / 2;  // code bytes of one weight row per chunk
  constexpr int NL = ROW_BYTES_PER_CHUNK / 64;                         // 16-byte loads per lane and chunk
  constexpr int STEPS = BKIND == WL_B_I8 ? 2 : 4;                      // MFMA k-steps one load feeds
  constexpr int DEPTH = 2;                                             // chunks of code loads in flight
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g4 = lane >> 4;
  const int nb = (int)blockIdx.x / a.S, slice = (int)blockIdx.x - nb * a.S;
  // the matrix this n-block belongs to (block-uniform selects: no dynamic indexing of the kernel arguments)
  const int seg = nb >= a.seg_block[2] ? 2 : nb >= a.seg_block[1] ? 1 : 0;
  const uint8_t* const w_base = seg == 0 ? a.w[0] : seg == 1 ? a.w[1] : a.w[2];
  const float* const s_base = seg == 0 ? a.scale[0] : seg == 1 ? a.scale[1] : a.scale[2];
  const float* const o_base = seg == 0 ? a.offset[0] : seg == 1 ? a.offset[1] : a.offset[2];
  const int rows = seg == 0 ? a.seg_n[0] : seg == 1 ? a.seg_n[1] : a.seg_n[2];
  const int n0 = (nb - (seg == 0 ? 0 : seg == 1 ? a.seg_block[1] : a.seg_block[2])) * SK_BN;  // first row of the block inside its matrix
  int row = n0 + wave * 16 + r16;
  row = row < rows ? row : rows - 1;  // rows past the edge re-read the last row and are never stored
  const uint32_t w_row_bytes = BKIND == WL_B_I8 ? (uint32_t)a.K : (uint32_t)a.K / 2u;
  const uint8_t* const w_row = w_base + (size_t)row * w_row_bytes;
  const size_t p_row = a.per_row ? (size_t)row * (size_t)a.groups : 0;

  const int c_begin = (int)((int64_t)slice * a.chunks / a.S), c_end = (int)((int64_t)(slice + 1) * a.chunks / a.S);

  // ---- the code stream: chunk c -> NL loads of 16 bytes per lane (+ the parameters of their groups)
  u32x4 raw[DEPTH][NL];
  [[maybe_unused]] float sc[DEPTH][NL][2], ro[DEPTH][NL][2];  // GROUPED: per load; [1] = the high nibbles' group (WL_B_I4)
  float s_row = 1.0f, o_row = 0.0f;
  if constexpr (!GROUPED) {
    s_row = s_base[p_row];
    if constexpr (OFFSET) o_row = rne(o_base[p_row]);
  }
  // first code (k) of load j's low / only half inside the chunk, for this lane
  auto k_of = [&](int j, int half) -> int {
    if constexpr (BKIND == WL_B_I8) {
      return 64 * j + 16 * g4;
    } else {
      const int p = 64 * j + 16 * g4, hb = 1 << (a.pack_shift - 1);  // packed byte inside the chunk's row segment; bytes per packing block
      return ((p >> (a.pack_shift - 1)) << a.pack_shift) + (p & (hb - 1)) + half * hb;
    }
  };
  auto load_chunk = [&](int c, int d) {
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      raw[d][j] = *reinterpret_cast<const u32x4*>(w_row + (size_t)c * ROW_BYTES_PER_CHUNK + 64 * j + 16 * g4);
      if constexpr (GROUPED) {
#pragma unroll
        for (int h = 0; h < (BKIND == WL_B_I4 ? 2 : 1); ++h) {
          const size_t gi = p_row + (size_t)fdiv((uint32_t)(c * KC + k_of(j, h)), a.group_div);
          sc[d][j][h] = s_base[gi];
          if constexpr (OFFSET) ro[d][j][h] = rne(o_base[gi]);
        }
      }
    }
  };

  // ---- the activation stream: chunk c -> PIECES 16-byte pieces per thread, register-staged into LDS buffer c & 1
  constexpr int SEGS = KC / 8;  // 16-byte pieces per activation row and chunk
  u32x4 xr[Shape::PIECES];
  auto load_x = [&](int c) {
#pragma unroll
    for (int i = 0; i < Shape::PIECES; ++i) {
      const int idx = tid + 512 * i, m = idx / SEGS, sg = idx % SEGS;
      xr[i] = u32x4{0u, 0u, 0u, 0u};
      if (m < a.M) xr[i] = *reinterpret_cast<const u32x4*>(a.x + ((size_t)m * (size_t)a.K + (size_t)c * KC) * 2u + sg * 16);
    }
  };
  auto store_x = [&](int c) {
    uint8_t* buf = lds + (c & 1) * Shape::STAGE;
#pragma unroll
    for (int i = 0; i < Shape::PIECES; ++i) {
      const int idx = tid + 512 * i, m = idx / SEGS, sg = idx % SEGS;
      *reinterpret_cast<u32x4*>(buf + m * PITCH + sg * 16) = xr[i];
    }
  };

  wl_v4f acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) acc[mt] = wl_v4f{0.0f, 0.0f, 0.0f, 0.0f};

  if (c_begin < c_end) {
    load_x(c_begin);
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
      if (c_begin + d < c_end) load_chunk(c_begin + d, d);
    store_x(c_begin);
  }
  for (int c = c_begin; c < c_end; ++c) {
    const int d = (c - c_begin) & (DEPTH - 1);
    __syncthreads();  // activations of chunk c visible; everybody is done reading the buffer chunk c + 1 goes to
    if (c + 1 < c_end) load_x(c + 1);
    const uint8_t* xbuf = lds + (c & 1) * Shape::STAGE + r16 * PITCH;
    // (DEPTH == 2: the compiler resolves raw[d] with d in {0, 1} by predication of two unrolled bodies)
#pragma unroll
    for (int dd = 0; dd < DEPTH; ++dd) {
      if (dd != d) continue;
#pragma unroll
      for (int j = 0; j < NL; ++j) {
        const uint32_t wsrc[4] = {raw[dd][j].x, raw[dd][j].y, raw[dd][j].z, raw[dd][j].w};
#pragma unroll
        for (int h = 0; h < (BKIND == WL_B_I4 ? 2 : 1); ++h) {
          float s = GROUPED ? sc[dd][j][h] : s_row;
          float co = OFFSET ? (GROUPED ? ro[dd][j][h] : o_row) : 0.0f;
          uint32_t wv[4];
          if constexpr (BKIND == WL_B_I4) {
            // nibble n = code + 8 -> (n ^ 8) << 4 in the byte's high half = 16 * code as a signed byte, and (16 q + 16 o) * (s / 16) is
            // (q + o) * s with the same single rounding wherever s / 16 is exact; a tiny scale takes the codes themselves (ffq_wlinear.hip)
            const bool tiny = __builtin_fabsf(s) < 0x1p-120f && s != 0.0f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              uint32_t b = (((wsrc[q] >> (4 * h)) << 4) & 0xF0F0F0F0u) ^ 0x80808080u;
              if (__builtin_expect(tiny, 0)) {
                const uint32_t b0 = (uint32_t)(((int32_t)(b << 24)) >> 28) & 0xFFu, b1 = (uint32_t)(((int32_t)(b << 16)) >> 28) & 0xFFu;
                const uint32_t b2 = (uint32_t)(((int32_t)(b << 8)) >> 28) & 0xFFu, b3 = (uint32_t)(((int32_t)b) >> 28) & 0xFFu;
                b = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
              }
              wv[q] = b;
            }
            if (!tiny) { s = s * 0.0625f; co = co * 16.0f; }
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) wv[q] = wsrc[q];
          }
          uint32_t o8[8];
#pragma unroll
          for (int q = 0; q < 4; ++q) dequantize4<OFFSET>(wv[q], s, co, o8[2 * q], o8[2 * q + 1]);
          const int k0 = k_of(j, h);
#pragma unroll
          for (int st = 0; st < 2; ++st) {  // bytes 0-7 and 8-15 of the piece: two MFMA k-slices of this lane
            const wl_v4i wf = {(int)o8[4 * st], (int)o8[4 * st + 1], (int)o8[4 * st + 2], (int)o8[4 * st + 3]};
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
              const wl_v4i xf = *reinterpret_cast<const wl_v4i*>(xbuf + mt * 16 * PITCH + (k0 + 8 * st) * 2);
              acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(wl_v8bf, wf), __builtin_bit_cast(wl_v8bf, xf), acc[mt], 0, 0, 0);
            }
          }
        }
      }
      if (c + DEPTH < c_end) load_chunk(c + DEPTH, dd);  // the registers just consumed take the chunk after next
    }
    if (c + 1 < c_end) store_x(c + 1);
  }
  (void)STEPS;

  // ---- a wave's result: acc[mt][t] = y[m = 16 mt + r16][n = n0 + 16 wave + 4 g4 + t] (partial over this block's k slice)
  if (a.S > 1) {
    const size_t unit_bytes = (size_t)MT * 1024;
    uint8_t* const strip = reinterpret_cast<uint8_t*>(a.slabs) + ((size_t)nb * a.S * SK_WAVES + wave) * unit_bytes;  // slice 0 of this strip
    const size_t slice_stride = (size_t)SK_WAVES * unit_bytes;
    {
      const auto mine = __builtin_amdgcn_make_buffer_rsrc(strip + (size_t)slice * slice_stride, 0, (int)unit_bytes, 0x00020000);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(wl_v4u, acc[mt]), mine, (mt * 64 + lane) * 16, 0, /*sc1*/ 16);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the write-through stores have left before the ticket is taken
    int t = 0;
    if (lane == 0) t = __hip_atomic_fetch_add(a.tickets + nb * SK_WAVES + wave, 1, FFQ_TICKET_ORDER, __HIP_MEMORY_SCOPE_AGENT);
    t = __builtin_amdgcn_readfirstlane(t);
    if (t != a.S - 1) return;  // somebody else finishes this strip
    if (lane == 0) __hip_atomic_store(a.tickets + nb * SK_WAVES + wave, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // zero again for the next launch
    asm volatile("" ::: "memory");  // the partials are read after the ticket said everybody has written
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = wl_v4f{0.0f, 0.0f, 0.0f, 0.0f};
    for (int sl = 0; sl < a.S; ++sl) {  // slice order, whoever reduces: the sum is a function of the plan alone
      const auto peer = __builtin_amdgcn_make_buffer_rsrc(strip + (size_t)sl * slice_stride, 0, (int)unit_bytes, 0x00020000);
      wl_v4u got[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) got[mt] = __builtin_amdgcn_raw_buffer_load_b128(peer, (mt * 64 + lane) * 16, 0, /*sc1*/ 16);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const wl_v4f g = __builtin_bit_cast(wl_v4f, got[mt]);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[mt][e] = acc[mt][e] + g[e];
      }
    }
  }
  // ---- epilogue: bias, cast, 4 consecutive columns per lane and row
  const int ncol = n0 + wave * 16 + 4 * g4;
  if (ncol >= rows) return;
  void* const out = seg == 0 ? a.out[0] : seg == 1 ? a.out[1] : a.out[2];
  float b4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  if (a.bias) {
#pragma unroll
    for (int e = 0; e < 4; ++e) b4[e] = ncol + e < rows ? (float)load_any(a.bias, a.bias_dt, ncol + e) : 0.0f;
  }
  const bool whole = ncol + 4 <= rows && (rows & 3) == 0;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = 16 * mt + r16;
    if (m >= a.M) continue;
    float y[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) y[e] = a.bias ? acc[mt][e] + b4[e] : acc[mt][e];
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

// ---- the rows form: NT 16-row weight tiles per BLOCK, the K range cut across its eight WAVES -----------------------------------------
// With few rows the activations are small (32 KiB per 16 rows of a 4096-deep contraction, L2-resident for everybody), so nothing
// has to be shared through LDS and nothing has to be exchanged between blocks: a block owns 16 NT output columns for the whole
// contraction (256 blocks for a 4096-wide projection at NT = 1, 448 for gate / up at NT = 2), every wave contracts an eighth of
// the K range with both operands loaded straight into the MFMA's register layout (weights: 16 rows x 64 B per instruction;
// activations: lane (m, g) reads the 32 bytes of row 16 mt + m it multiplies — rows >= M re-read row M - 1: a column of the
// MFMA's result that is never stored — and every activation fragment serves the block's NT weight tiles), the eight waves'
// partial tiles meet in LDS and wave t adds tile t of the NT x MT in wave order. The dependent chain of a launch is one memory
// round trip + one barrier, where the 128-column form pays a write-through, a ticket and a read-back on top; the price is
// activation traffic from L2 — N / (16 NT) x 16 MT x K x 2 bytes — which bounds the form (sk_rows_form below, measured).
// Matrices with fewer than CUs / 4 blocks also cut K across `S` blocks and finish through the tickets.
template <int BKIND, bool GROUPED, bool OFFSET, int MT, int NT>
__global__ __launch_bounds__(512, 2) void wq_skinny_rows_kernel(SkinnyArgs a) {
  // The K range is walked in UNITS of 64 k: one 16-byte piece of int8 codes per lane and weight tile, or one nibble half of a
  // 16-byte piece of packed codes (two units share a load); the activation fragments of XD units are in flight per wave.
  // Partitions (slices across blocks, ranges of the eight waves) are taken in PAIRS of units = 128 k, a packing block of nibbles:
  // both storage forms contract the same k values in the same MFMA steps of the same wave and give the same bits.
  constexpr int XD = MT == 4 ? 2 : 4;                      // units in flight (8 MT registers of activation fragments each)
  constexpr int WD = BKIND == WL_B_I8 ? XD : XD / 2;       // weight pieces in flight per weight tile
  constexpr int TILES = NT * MT;
  static_assert(TILES <= SK_WAVES, "one wave finishes one tile");
  __shared__ __attribute__((aligned(16))) uint8_t red[SK_WAVES * TILES * 1024];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g4 = lane >> 4;
  const int nb = (int)blockIdx.x / a.S, slice = (int)blockIdx.x - nb * a.S;
  const int seg = nb >= a.seg_block[2] ? 2 : nb >= a.seg_block[1] ? 1 : 0;
  const uint8_t* const w_base = seg == 0 ? a.w[0] : seg == 1 ? a.w[1] : a.w[2];
  const float* const s_base = seg == 0 ? a.scale[0] : seg == 1 ? a.scale[1] : a.scale[2];
  const float* const o_base = seg == 0 ? a.offset[0] : seg == 1 ? a.offset[1] : a.offset[2];
  const int rows = seg == 0 ? a.seg_n[0] : seg == 1 ? a.seg_n[1] : a.seg_n[2];
  const int n0 = (nb - (seg == 0 ? 0 : seg == 1 ? a.seg_block[1] : a.seg_block[2])) * 16 * NT;
  const uint32_t w_row_bytes = BKIND == WL_B_I8 ? (uint32_t)a.K : (uint32_t)a.K / 2u;
  const uint8_t* w_row[NT];
  size_t p_row[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    int row = n0 + 16 * nt + r16;
    row = row < rows ? row : rows - 1;  // rows past the edge re-read the last row and are never stored
    w_row[nt] = w_base + (size_t)row * w_row_bytes + 16 * g4;
    p_row[nt] = a.per_row ? (size_t)row * (size_t)a.groups : 0;
  }
  const uint8_t* x_row[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = 16 * mt + r16;
    x_row[mt] = a.x + ((size_t)(m < a.M ? m : a.M - 1) * (size_t)a.K + 16u * g4) * 2u;
  }

  // pairs of units (128 k) of this block's K slice, then of this wave; units [u_begin, u_end)
  const int b_begin = (int)((int64_t)slice * a.chunks / a.S), b_end = (int)((int64_t)(slice + 1) * a.chunks / a.S);
  const int u_begin = 2 * (b_begin + (int)((int64_t)wave * (b_end - b_begin) / SK_WAVES)), u_end = 2 * (b_begin + (int)((int64_t)(wave + 1) * (b_end - b_begin) / SK_WAVES));

  u32x4 raw[WD][NT], xf[XD][2][MT];
  [[maybe_unused]] float sc[XD][NT], ro[XD][NT];
  float s_row[NT], o_row[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    s_row[nt] = 1.0f; o_row[nt] = 0.0f;
    if constexpr (!GROUPED) {
      s_row[nt] = s_base[p_row[nt]];
      if constexpr (OFFSET) o_row[nt] = rne(o_base[p_row[nt]]);
    }
  }
  // unit v into slot k (compile-time): this lane's k values are 64 v + 16 g4 + (0..15)
  auto load_unit = [&](int v, auto kc) {
    constexpr int k = decltype(kc)::value;
    if constexpr (BKIND == WL_B_I8) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) raw[k][nt] = *reinterpret_cast<const u32x4*>(w_row[nt] + (size_t)v * 64);
    } else if constexpr ((k & 1) == 0) {  // the pair's packed bytes: low nibbles = unit v, high nibbles = unit v + 1 (packing block 128)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) raw[k >> 1][nt] = *reinterpret_cast<const u32x4*>(w_row[nt] + (size_t)(v >> 1) * 64);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      xf[k][0][mt] = *reinterpret_cast<const u32x4*>(x_row[mt] + (size_t)v * 128);
      xf[k][1][mt] = *reinterpret_cast<const u32x4*>(x_row[mt] + (size_t)v * 128 + 16);
    }
    if constexpr (GROUPED) {
      const uint32_t grp = fdiv((uint32_t)v * 64u, a.group_div);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        sc[k][nt] = s_base[p_row[nt] + grp];
        if constexpr (OFFSET) ro[k][nt] = rne(o_base[p_row[nt] + grp]);
      }
    }
  };

  wl_v4f acc[NT][MT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = wl_v4f{0.0f, 0.0f, 0.0f, 0.0f};
  auto compute_unit = [&](auto kc) {
    constexpr int k = decltype(kc)::value;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      u32x4 piece;
      if constexpr (BKIND == WL_B_I8) piece = raw[k][nt]; else piece = raw[k >> 1][nt];
      const uint32_t wsrc[4] = {piece.x, piece.y, piece.z, piece.w};
      float s = GROUPED ? sc[k][nt] : s_row[nt];
      float co = OFFSET ? (GROUPED ? ro[k][nt] : o_row[nt]) : 0.0f;
      uint32_t wv[4];
      if constexpr (BKIND == WL_B_I4) {
        // nibble n = code + 8 -> (n ^ 8) << 4 in the byte's high half = 16 * code as a signed byte, and (16 q + 16 o) * (s / 16) is
        // (q + o) * s with the same single rounding wherever s / 16 is exact; a tiny scale takes the codes themselves (ffq_wlinear.hip)
        const bool tiny = __builtin_fabsf(s) < 0x1p-120f && s != 0.0f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          uint32_t b = (((wsrc[q] >> (4 * (k & 1))) << 4) & 0xF0F0F0F0u) ^ 0x80808080u;
          if (__builtin_expect(tiny, 0)) {
            const uint32_t b0 = (uint32_t)(((int32_t)(b << 24)) >> 28) & 0xFFu, b1 = (uint32_t)(((int32_t)(b << 16)) >> 28) & 0xFFu;
            const uint32_t b2 = (uint32_t)(((int32_t)(b << 8)) >> 28) & 0xFFu, b3 = (uint32_t)(((int32_t)b) >> 28) & 0xFFu;
            b = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
          }
          wv[q] = b;
        }
        if (!tiny) { s = s * 0.0625f; co = co * 16.0f; }
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) wv[q] = wsrc[q];
      }
      uint32_t o8[8];
#pragma unroll
      for (int q = 0; q < 4; ++q) dequantize4<OFFSET>(wv[q], s, co, o8[2 * q], o8[2 * q + 1]);
#pragma unroll
      for (int st = 0; st < 2; ++st) {  // bytes 0-7 and 8-15 of the piece: two MFMA k-slices of this lane
        const wl_v4i wf = {(int)o8[4 * st], (int)o8[4 * st + 1], (int)o8[4 * st + 2], (int)o8[4 * st + 3]};
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(wl_v8bf, wf), __builtin_bit_cast(wl_v8bf, xf[k][st][mt]), acc[nt][mt], 0, 0, 0);
      }
    }
  };
  auto each_slot = [&](auto&& fn) {
    fn(std::integral_constant<int, 0>{});
    fn(std::integral_constant<int, 1>{});
    if constexpr (XD == 4) {
      fn(std::integral_constant<int, 2>{});
      fn(std::integral_constant<int, 3>{});
    }
  };
  each_slot([&](auto kc) { if (u_begin + decltype(kc)::value < u_end) load_unit(u_begin + decltype(kc)::value, kc); });
  for (int u = u_begin; u < u_end; u += XD) {  // (u_begin, u_end and XD are even: a pair of units never straddles two rounds)
    each_slot([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      if (u + k < u_end) {
        compute_unit(kc);
        // the slot just consumed takes the unit a round ahead; a packed pair's piece is consumed by its SECOND unit
        if (u + k + XD < u_end) {
          if constexpr (BKIND == WL_B_I8 || (k & 1) == 1) {
            if constexpr (BKIND == WL_B_I4) load_unit(u + k - 1 + XD, std::integral_constant<int, k - 1>{});
            load_unit(u + k + XD, kc);
          }
        }
      }
    });
  }
  // ---- the eight waves' partial tiles -> wave t finishes tile t = (nt, mt), adding the waves in wave order
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) *reinterpret_cast<wl_v4f*>(red + ((wave * TILES + nt * MT + mt) * 64 + lane) * 16) = acc[nt][mt];
  __syncthreads();
  if (wave >= TILES) return;
  const int tile = wave, tnt = tile / MT, tmt = tile - tnt * MT;
  wl_v4f sum = *reinterpret_cast<const wl_v4f*>(red + ((0 * TILES + tile) * 64 + lane) * 16);
#pragma unroll
  for (int w = 1; w < SK_WAVES; ++w) {
    const wl_v4f p = *reinterpret_cast<const wl_v4f*>(red + ((w * TILES + tile) * 64 + lane) * 16);
#pragma unroll
    for (int e = 0; e < 4; ++e) sum[e] = sum[e] + p[e];
  }
  if (a.S > 1) {  // K also cut across blocks: the ticketed exchange of wq_skinny_kernel with one strip per (n-block, tile)
    uint8_t* const strip = reinterpret_cast<uint8_t*>(a.slabs) + ((size_t)nb * a.S * TILES + tile) * 1024;
    const size_t slice_stride = (size_t)TILES * 1024;
    {
      const auto mine = __builtin_amdgcn_make_buffer_rsrc(strip + (size_t)slice * slice_stride, 0, 1024, 0x00020000);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(wl_v4u, sum), mine, lane * 16, 0, /*sc1*/ 16);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int t = 0;
    if (lane == 0) t = __hip_atomic_fetch_add(a.tickets + nb * TILES + tile, 1, FFQ_TICKET_ORDER, __HIP_MEMORY_SCOPE_AGENT);
    t = __builtin_amdgcn_readfirstlane(t);
    if (t != a.S - 1) return;
    if (lane == 0) __hip_atomic_store(a.tickets + nb * TILES + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("" ::: "memory");
    sum = wl_v4f{0.0f, 0.0f, 0.0f, 0.0f};
    for (int sl = 0; sl < a.S; ++sl) {
      const auto peer = __builtin_amdgcn_make_buffer_rsrc(strip + (size_t)sl * slice_stride, 0, 1024, 0x00020000);
      const wl_v4f g = __builtin_bit_cast(wl_v4f, __builtin_amdgcn_raw_buffer_load_b128(peer, lane * 16, 0, /*sc1*/ 16));
#pragma unroll
      for (int e = 0; e < 4; ++e) sum[e] = sum[e] + g[e];
    }
  }
  const int ncol = n0 + 16 * tnt + 4 * g4, m = 16 * tmt + r16;
  if (ncol >= rows || m >= a.M) return;
  void* const out = seg == 0 ? a.out[0] : seg == 1 ? a.out[1] : a.out[2];
  float y[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) y[e] = a.bias ? sum[e] + (ncol + e < rows ? (float)load_any(a.bias, a.bias_dt, ncol + e) : 0.0f) : sum[e];
  const bool whole = ncol + 4 <= rows && (rows & 3) == 0;
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

// ---- the plan ---------------------------------------------------------------------------------------------------------------------
// Both storage forms of one weight take the same kernel and the same partition of K (pairs of 64-k units = 128 k): that is what makes
// them agree bit for bit, so nothing below depends on the container.
static int sk_mt(int64_t M) { return M <= 16 ? 1 : M <= 32 ? 2 : M <= 64 ? 4 : 8; }
static int sk_cols_kc(int64_t M) { return sk_mt(M) <= 4 ? 256 : 128; }  // K chunk of the 128-column form
// The rows form (a block owns its columns for the whole contraction) re-reads the activations from L2 once per 16 NT columns —
// N / (16 NT) x 16 MT x K x 2 bytes: it wins outright up to 16 rows and, for contractions up to 4096 deep, up to 32 (q/o 13.2 vs
// 17.3 us, k/v 12.9 vs 14.6; down_proj's 14336-deep rows would be 36 vs 27 us) — and covers the shapes the 128-column form's chunk
// does not divide. The form is a function of (M, K) alone: one launch of several matrices and one launch each agree bit for bit.
static bool sk_rows_form(int64_t M, int64_t K) {
  return M <= 16 || (M <= FFQ_SK_ROWS_MAX_M && K <= FFQ_SK_ROWS_MAX_K) || (M <= 64 && K % sk_cols_kc(M) != 0);
}
constexpr int SK_KC = 128;  // rows form: a pair of 64-k units
// weight tiles of 16 rows per block of the rows form: two where every matrix of the launch keeps >= CUs blocks (gate / up) and the
// rows are distinct enough to be worth sharing (a single token's row is one L2 line for all sixteen lanes of a fragment)
static int sk_nt(int64_t M, int64_t K, int64_t n_min) {
  if (!sk_rows_form(M, K) || sk_mt(M) > 2 || M < FFQ_SK_NT2_MIN_M) return 1;
  return n_min / 32 >= wq_cus() && n_min % 32 == 0 ? 2 : 1;
}
static int sk_bn(int64_t M, int64_t K, int64_t n_min) { return sk_rows_form(M, K) ? 16 * sk_nt(M, K, n_min) : SK_BN; }
static int64_t sk_n_blocks(int64_t M, int64_t K, int64_t N, int64_t n_min) { const int bn = sk_bn(M, K, n_min); return (N + bn - 1) / bn; }
// ticket words = partial strips per n-block, and the bytes of one strip
static int64_t sk_strips(int64_t M, int64_t K, int64_t n_min) { return sk_rows_form(M, K) ? sk_nt(M, K, n_min) * sk_mt(M) : SK_WAVES; }
static int64_t sk_strip_bytes(int64_t M, int64_t K) { return sk_rows_form(M, K) ? 1024 : (int64_t)sk_mt(M) * 1024; }

static bool sk_shape_ok(int64_t M, int64_t K) { return M >= 1 && M <= SK_MAX_M && K % SK_KC == 0; }

static int sk_split_for(int64_t M, int64_t N, int64_t n_min, int64_t K) {
  if (!sk_shape_ok(M, K)) return 1;
  const int64_t blocks = sk_n_blocks(M, K, N, n_min), cus = wq_cus();
  int64_t S;
  if (sk_rows_form(M, K)) {
    // a block's eight waves already share the K range: 64 blocks (k / v) are quicker whole than as 128 halves that meet through a
    // ticket (6.6 vs 8.9 us at one token); cut K across blocks only below a quarter of the chip
    const int64_t chunks = K / SK_KC;
    S = blocks >= cus / 4 ? 1 : (cus / 4 + blocks - 1) / blocks;
    if (S > chunks / SK_WAVES) S = chunks / SK_WAVES;  // at least one pair of units per wave
  } else {
    const int64_t chunks = K / sk_cols_kc(M);
    S = (cus + blocks - 1) / blocks;  // at least one block per CU ...
    while (S < chunks / 2 && chunks % S != 0) ++S;  // ... of equally many chunks (three slices of 16 chunks: 29.1 us, four: 25.8) ...
    if (S > chunks / 2) S = chunks / 2;  // ... and at least two (one in flight behind the one being contracted)
  }
  if (S > 64) S = 64;
  return S < 1 ? 1 : (int)S;
}

// (the plan queries of the C ABI see M, N, K only: one matrix)
int wq_skinny_split(int64_t M, int64_t N, int64_t K) { return sk_split_for(M, N, N, K); }

// ticket words / slab bytes: upper bounds over the forms a launch of this (M, N, K) can take (one or two weight tiles per block, one
// to three matrices, each rounded up to whole blocks)
int64_t wq_skinny_tickets(int64_t M, int64_t N, int64_t K) {
  if (!sk_shape_ok(M, K)) return 0;
  return ((N + 15) / 16 + 6) * SK_WAVES;
}

size_t wq_skinny_slab_bytes(int64_t M, int64_t N, int64_t K, int64_t split) {
  if (!sk_shape_ok(M, K) || split <= 1) return 0;
  return (size_t)((N + 15) / 16 + 24) * (size_t)split * (size_t)sk_mt(M) * 1024u;
}

bool wq_skinny_applies(const WLinearArgs& a, int64_t pack_block) {
  if (generic_kernels_forced()) return false;  // tests: the 256-row-tile kernel on the same operands (ffq_force_generic_kernels)
  // packed nibbles: only the packing block whose low / high halves are the 64-code units of the int8 walk (128: BASELINE config 4's
  // group size) — then lane (r, g) multiplies the same k values in the same MFMA steps as with an int8 container and every storage
  // form gives the same bits; other blocks (GGUF's 32, 64, 256) keep the 256-row-tile kernel, which converts into a
  // k-ordered LDS image
  if (pack_block != 0 && pack_block != 128) return false;
  if (!sk_shape_ok(a.M, a.K)) return false;
  if (a.groups > 1 && (a.K / a.groups) % 64 != 0) return false;
  return true;
}

template <int BKIND, bool GROUPED, bool OFFSET>
static void sk_launch_cols(const SkinnyArgs& s, int mt, unsigned grid, hipStream_t stream) {
#define FFQ_SK(MT)                                                                                                            \
  do {                                                                                                                        \
    static uint64_t attr_set = 0;                                                                                             \
    const int lds_bytes = 2 * SkinnyShape<MT>::STAGE;                                                                         \
    ensure_dynamic_lds(&attr_set, reinterpret_cast<const void*>(&wq_skinny_kernel<BKIND, GROUPED, OFFSET, MT>), lds_bytes);   \
    wq_skinny_kernel<BKIND, GROUPED, OFFSET, MT><<<grid, 512, lds_bytes, stream>>>(s);                                         \
  } while (0)
  switch (mt) {
    case 2: FFQ_SK(2); break;
    case 4: FFQ_SK(4); break;
    default: FFQ_SK(8); break;
  }
#undef FFQ_SK
}

template <int BKIND, bool GROUPED, bool OFFSET>
static void sk_launch_rows(const SkinnyArgs& s, int mt, int nt, unsigned grid, hipStream_t stream) {
#define FFQ_SKR(MT, NT) wq_skinny_rows_kernel<BKIND, GROUPED, OFFSET, MT, NT><<<grid, 512, 0, stream>>>(s)
  switch (mt) {
    case 1: if (nt == 2) FFQ_SKR(1, 2); else FFQ_SKR(1, 1); break;
    case 2: if (nt == 2) FFQ_SKR(2, 2); else FFQ_SKR(2, 1); break;
    default: FFQ_SKR(4, 1); break;
  }
#undef FFQ_SKR
}

int wq_skinny_launch(const WLinearArgs& a, int w_dt, int64_t pack_block, int64_t group, int64_t split, void* workspace, size_t workspace_bytes,
                     int32_t* tickets, hipStream_t stream) {
  (void)pack_block;
  SkinnyArgs s;
  s.x = a.x;
  s.w[0] = a.w; s.scale[0] = a.w_scale; s.offset[0] = a.w_offset; s.out[0] = a.out;
  int64_t N = 0, blocks = 0, n_min = INT64_MAX;
  for (int i = 0; i < 3; ++i)
    if (a.seg_n[i] > 0 && a.seg_n[i] < n_min) n_min = a.seg_n[i];
  for (int i = 0; i < 3; ++i) {
    s.seg_n[i] = a.seg_n[i];
    s.seg_block[i] = i == 0 ? 0 : (a.seg_n[i] > 0 ? (int)blocks : INT32_MAX);
    if (i > 0) { s.w[i] = a.seg_w[i - 1]; s.scale[i] = a.seg_scale[i - 1]; s.offset[i] = a.seg_offset[i - 1]; s.out[i] = a.seg_out[i - 1]; }
    N += a.seg_n[i];
    blocks += sk_n_blocks(a.M, a.K, a.seg_n[i], n_min);
  }
  s.bias = a.bias; s.bias_dt = a.bias_dt; s.out_dt = a.out_dt;
  s.M = a.M; s.K = a.K;
  s.n_blocks = (int)blocks;
  s.groups = a.groups; s.group_div = make_fastdiv((uint32_t)group); s.per_row = a.per_row; s.pack_shift = a.pack_shift;
  const bool rows_form = sk_rows_form(a.M, a.K);
  const int mt = sk_mt(a.M), nt = sk_nt(a.M, a.K, n_min), kc = rows_form ? SK_KC : sk_cols_kc(a.M);
  s.chunks = (int)(a.K / kc);
  int64_t S = split > 0 ? split : sk_split_for(a.M, N, n_min, a.K);
  if (S > s.chunks) {
    if (split > 0) return fail(FFQ_ERR_ARG, "weight-only linear (skinny form): split %lld exceeds the %d chunks of %d along K", (long long)split, s.chunks, kc);
    S = s.chunks;
  }
  const size_t slab = S > 1 ? (size_t)blocks * (size_t)S * (size_t)sk_strips(a.M, a.K, n_min) * (size_t)sk_strip_bytes(a.M, a.K) : 0;
  if (S > 1 && (!tickets || !workspace || workspace_bytes < slab || !aligned16(workspace))) {
    if (split > 1) return fail(FFQ_ERR_ARG, "weight-only linear (skinny form): split %lld needs %zu bytes of workspace and a ticket buffer", (long long)S, slab);
    S = 1;  // the plan is a preference: without scratch every block walks the whole K range
  }
  s.S = (int)S;
  s.slabs = S > 1 ? static_cast<float*>(workspace) : nullptr;
  s.tickets = S > 1 ? tickets : nullptr;
  const unsigned grid = (unsigned)(blocks * S);
  const bool grouped = a.groups > 1, offset = a.w_offset != nullptr;
  if (rows_form) {
#define FFQ_SKR_T(BK) do { if (grouped) { if (offset) sk_launch_rows<BK, true, true>(s, mt, nt, grid, stream); else sk_launch_rows<BK, true, false>(s, mt, nt, grid, stream); } else { if (offset) sk_launch_rows<BK, false, true>(s, mt, nt, grid, stream); else sk_launch_rows<BK, false, false>(s, mt, nt, grid, stream); } } while (0)
    if (w_dt == FFQ_U8) FFQ_SKR_T(WL_B_I4); else FFQ_SKR_T(WL_B_I8);
#undef FFQ_SKR_T
    return check_launch("wq_skinny_rows_kernel");
  }
#define FFQ_SK_T(BK)                                                                              \
  do {                                                                                            \
    if (grouped) { if (offset) sk_launch_cols<BK, true, true>(s, mt, grid, stream); else sk_launch_cols<BK, true, false>(s, mt, grid, stream); } \
    else { if (offset) sk_launch_cols<BK, false, true>(s, mt, grid, stream); else sk_launch_cols<BK, false, false>(s, mt, grid, stream); }        \
  } while (0)
  if (w_dt == FFQ_U8) FFQ_SK_T(WL_B_I4); else FFQ_SK_T(WL_B_I8);
#undef FFQ_SK_T
  return check_launch("wq_skinny_kernel");
}

}  // namespace ffq

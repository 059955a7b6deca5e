/*
 * ffq.h — C ABI of the MI355X-native fake-quantization backend.
 *
 * This is the drop-in boundary for ONE hot path of Qualcomm-AI-research/fastforward:
 * affine quantize / dequantize by tile, the RunningMinMax reduction, range -> (scale, offset),
 * sub-byte packing, and the W8A8 linear. The reference exposes this path as four torch custom
 * ops in the `fastforward::` namespace plus one dispatcher hook; each entry point below names
 * the reference interface (file:line under the reference's `src/fastforward/`) it replaces.
 *
 * Two libraries export exactly this ABI:
 *   - libffq_hip.so    (fastforward_amd/csrc, hand-written HIP for gfx950) — pointers are DEVICE
 *                      pointers, `stream` is a hipStream_t. This is the product.
 *   - libffq_oracle.so (oracle/, plain C)                                  — pointers are HOST
 *                      pointers, `stream` is ignored. This is test infrastructure only.
 *
 * Conventions
 *   - All tensors are dense, row-major ("contiguous"), described by dtype tags + ffq_tiling.
 *   - Functional: inputs are never written, outputs are caller-allocated, same numel as `data`.
 *   - Nothing here allocates, synchronises or reads back: every call is a pure enqueue on
 *     `stream` and is therefore legal inside hipGraph capture. Scratch memory is passed in by the
 *     caller (`workspace`), sized by the matching *_workspace_bytes() query.
 *   - Every function returns an ffq_status; ffq_last_error() gives a thread-local message.
 *   - Parameters (scale/offset) are indexed by tile in the row order defined by the reference's
 *     `tiles_to_rows` (quantization/tiled_tensor.py:71-98): tile-grid index row-major.
 */
#ifndef FFQ_H
#define FFQ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FFQ_MAX_DIMS 8
#define FFQ_ABI_VERSION 9

typedef enum ffq_status {
  FFQ_OK = 0,
  FFQ_ERR_TILE_RANK = 1,   /* -> ValueError, tiled_tensor.py:24-29 (rank mismatch)              */
  FFQ_ERR_TILE_DIVIDE = 2, /* -> ValueError, tiled_tensor.py:31-42 (tile does not divide shape) */
  FFQ_ERR_PARAM_NUMEL = 3, /* -> RuntimeError, broadcast failure of scale[:, None]
                                 (_quantizer_impl.py:161; tests/nn/test_linear_quantizer.py:168-186) */
  FFQ_ERR_PRECISION = 4,   /* -> RuntimeError, _quantizer_impl.py:165-167                       */
  FFQ_ERR_EMPTY = 5,       /* -> QuantizationError, _quantizer_impl.py:259-264                  */
  FFQ_ERR_DTYPE = 6,       /* -> NotImplementedError: dtype combination not built               */
  FFQ_ERR_ARG = 7,         /* -> ValueError: null pointer, bad enum, bad size                   */
  FFQ_ERR_WORKSPACE = 8,   /* -> RuntimeError: workspace too small                              */
  FFQ_ERR_LAUNCH = 9,      /* -> RuntimeError: HIP runtime reported an error                    */
  FFQ_ERR_PARAM_ROWS = 10  /* -> ValueError: one tile but several parameters; the eager chain
                                 broadcasts rows [1,L] against scale [P,1] and rows_to_tiles then
                                 rejects the [P,L] result (tiled_tensor.py:128-131)             */
} ffq_status;

/* Element types. Values are part of the ABI. */
typedef enum ffq_dtype {
  FFQ_F32 = 0,
  FFQ_BF16 = 1,
  FFQ_F16 = 2,
  FFQ_F64 = 3,
  FFQ_I8 = 4,
  FFQ_I16 = 5,
  FFQ_I32 = 6,
  FFQ_I64 = 7,
  FFQ_U8 = 8
} ffq_dtype;

/*
 * Tile layout: `shape` is the data shape, `tile` the parameter-sharing tile
 * (Granularity.tile_size, quantization/granularity.py:52-62). tile[i] must divide shape[i].
 * ndim == 0 denotes a scalar (one tile of one element).
 */
typedef struct ffq_tiling {
  int32_t ndim;
  int64_t shape[FFQ_MAX_DIMS];
  int64_t tile[FFQ_MAX_DIMS];
} ffq_tiling;

/* Bits written into the int32 `status_flags` word by ffq_minmax_by_tile. */
#define FFQ_FLAG_INF 1 /* a per-tile min or max is +-Inf (range_setting/minmax.py:233-234) */
#define FFQ_FLAG_NAN 2 /* a per-tile min or max is NaN (informational)                    */

int ffq_abi_version(void);
const char* ffq_last_error(void);
/* "hip:gfx950" for the product library, "oracle:c" for the oracle. */
const char* ffq_backend_name(void);

/* Number of tiles (= number of scale/offset entries); negative ffq_status on a bad tiling. */
int64_t ffq_num_tiles(const ffq_tiling* tiling);

/* can_support_bitwidth, _quantizer_impl.py:44-75. Returns 1/0. */
int ffq_can_support_bitwidth(int dtype, double num_bits);

/* torch.result_type for two dimensioned tensors (the promotion the eager chain performs). */
int ffq_promote_types(int a, int b);

/*
 * A1 — fastforward::quantize_by_tile, _quantizer_impl.py:144-169 (called from
 * affine/_autograd.py:86).
 *   q = cast<out_dt>( clamp( round_half_even( x / s_t - round_half_even(o_t) ), -2^(b-1), 2^(b-1)-1 ) )
 * Arithmetic follows the eager chain's dtype promotion: the division is evaluated and rounded in
 * result_type(data, scale), the subtraction in result_type(that, offset). `offset == NULL` means
 * zeros_like(scale) (_infer_offset, :140-141). scale_numel / offset_numel must equal the number
 * of tiles, or 1 (the eager chain broadcasts `scale[:, None]` over the rows).
 */
int ffq_quantize_by_tile(const void* data, int data_dt, const void* scale, int scale_dt,
                         int64_t scale_numel, const void* offset, int offset_dt,
                         int64_t offset_numel, const ffq_tiling* tiling, double num_bits, void* out,
                         int out_dt, void* stream);

/*
 * A2 — fastforward::dequantize_by_tile, _quantizer_impl.py:172-190 (affine/_autograd.py:148).
 *   x^ = cast<out_dt>( (q + round_half_even(o_t)) * s_t )
 * The add is rounded in result_type(data, offset), the multiply in result_type(that, scale).
 */
int ffq_dequantize_by_tile(const void* data, int data_dt, const void* scale, int scale_dt,
                           int64_t scale_numel, const void* offset, int offset_dt,
                           int64_t offset_numel, const ffq_tiling* tiling, void* out, int out_dt,
                           void* stream);

/* The dtype dequantize_by_tile produces when output_dtype is None (_quantizer_impl.py:187-189). */
int ffq_dequantize_result_dtype(int data_dt, int scale_dt, int offset_dt, int has_offset);

/*
 * A4 — the reduction inside RunningMinMaxEstimator.estimate_step, range_setting/minmax.py:227-237
 * (also the first half of quantize_dynamic_by_tile, _quantizer_impl.py:257-258).
 * Per-tile min and max of `data`, NaN-propagating like torch.min/torch.max, written in the DATA
 * dtype (the reference keeps them in data dtype, minmax.py:209-213). With accumulate != 0 the
 * result is merged into the existing contents (running min / running max, :236-237).
 * `status_flags` (nullable, one int32) is OR-ed with FFQ_FLAG_* for THIS batch's per-tile
 * values, so the caller can raise NotImplementedError("Infinite") without a sync per step.
 * `ticket` (nullable, ABI 7): one int32 that is ZERO before the first call and that every call leaves zero (one word per
 * stream). With it a per-tensor reduction is ONE launch — every block publishes its partial result, the last block to
 * arrive finishes — instead of a reduction launch plus a one-block finalize launch.
 */
size_t ffq_minmax_workspace_bytes(const ffq_tiling* tiling, int data_dt);
int ffq_minmax_by_tile(const void* data, int data_dt, const ffq_tiling* tiling, void* min_inout,
                       void* max_inout, int accumulate, int32_t* status_flags, void* workspace,
                       size_t workspace_bytes, int32_t* ticket, void* stream);

/*
 * One RunningMinMaxEstimator.estimate_step on the device (range_setting/minmax.py:215-239): A4 merged into the running
 * extrema in place, then the range setter (nn/linear_quantizer.py:350-357) = A5 (affine/range.py:54-122) of the merged range
 * written straight into the quantizer's scale / offset tensors — the same values ffq_minmax_by_tile(accumulate = 1) followed
 * by ffq_parameters_for_range produce, bit for bit. With `ticket` a per-tensor quantizer takes ONE launch for the reduction,
 * the merge, the status flags and both parameters (three launches before round 4; 448 quantizers x 64 steps per calibration).
 */
int ffq_running_minmax_step(const void* data, int data_dt, const ffq_tiling* tiling, void* min_inout, void* max_inout,
                            int32_t* status_flags, double num_bits, int symmetric, int allow_one_sided, void* scale_out,
                            int scale_dt, void* offset_out, int offset_dt, void* workspace, size_t workspace_bytes,
                            int32_t* ticket, void* stream);

/*
 * The estimator step above AND the quantizer's own forward on the same data in ONE pass (2 B read + 1 B written per bf16 element
 * into an int8 container, instead of A4's read + A1's read and write): what ``estimate_ranges`` runs per quantizer call —
 * RunningMinMax.estimate_step, the range setter, then the quantizer's forward (range_setting/common.py:218-238) — for tilings whose
 * tiles are contiguous runs of at most 16384 / 8192 elements (per-channel(0) weights, per-token activations, group-128 weights;
 * more than one tile). Running min / max (data dtype) merged in place, status flags of THIS batch OR-ed in, scale / offset
 * (fp32, one per tile; the offset keeps its fraction as ffq_parameters_for_range leaves it) written, codes = A1 of `data` with
 * them. symmetric && allow_one_sided needs `ticket` (two int32, ZERO before the first call, left zero: the guess / settle launches
 * of ffq_quantize_dynamic_by_tile, judged on the MERGED minima). Same values as the two calls, bit for bit.
 * Anything else returns FFQ_ERR_DTYPE before touching a buffer: take ffq_running_minmax_step, then ffq_quantize_by_tile.
 */
int ffq_running_minmax_quantize(const void* data, int data_dt, const ffq_tiling* tiling, void* min_inout, void* max_inout,
                                int32_t* status_flags, double num_bits, int symmetric, int allow_one_sided, float* scale_out,
                                float* offset_out, void* out, int out_dt, int32_t* ticket, void* stream);

/*
 * A5 — parameters_for_range, quantization/affine/range.py:54-122, fused with the copy performed
 * by the LinearQuantizer.quantization_range setter, nn/linear_quantizer.py:350-357.
 * min/max (dtype `range_dt`, `ntiles` entries) are cast to fp32 (:90); the one-sided test is
 * GLOBAL over all tiles (:100) and is decided on the device. scale_out / offset_out are written
 * in their own dtypes. When the symmetric two-sided branch is taken (reference returns
 * offset=None) and offset_out != NULL it is filled with 0 (linear_quantizer.py:353-357).
 * `workspace` (ABI 8, nullable): ffq_parameters_for_range_workspace_bytes() bytes of scratch, contents irrelevant. Above 8192
 * tiles (group-128 weights: 458,752 for a 14336 x 4096 projection) the tiles are spread over the chip; the global one-sided
 * test then costs a first short launch that leaves one minimum per block in the scratch. Without scratch one block does
 * everything, as below 8192 tiles (same values either way).
 */
size_t ffq_parameters_for_range_workspace_bytes(int64_t ntiles, int symmetric, int allow_one_sided);
int ffq_parameters_for_range(const void* min_range, const void* max_range, int range_dt,
                             int64_t ntiles, double num_bits, int symmetric, int allow_one_sided,
                             void* scale_out, int scale_dt, void* offset_out, int offset_dt,
                             void* workspace, size_t workspace_bytes, void* stream);

/*
 * A3 — fastforward::quantize_dynamic_by_tile, _quantizer_impl.py:243-285 (affine/_autograd.py:121).
 * = A4 (fresh min/max) + A5 + round(offset) + A1, returning (q, scale, offset) with fp32 params.
 * Empty input -> FFQ_ERR_EMPTY.
 * ONE launch (ABI 8; 2 B read + 1 B written per bf16 element into an int8 container) when every tile is a contiguous run of at
 * most 16384 (1-byte containers) / 8192 elements — per-token and per-channel(0) activations, group-128 weights — and the
 * parameters of a tile depend on that tile alone: asymmetric, or symmetric with allow_one_sided == 0. The one-sided test of
 * range.py:100 (symmetric && allow_one_sided: is the smallest minimum of ALL tiles >= 0?) is the only cross-tile dependency of the
 * op; with `ticket` those calls take the same kernel twice: every tile whose own minimum is negative answers the question by
 * itself and is finished in the first launch, a tile that cannot know leaves its range behind and is counted, and the second
 * launch returns at once when nothing was left open (data with both signs in every tile: 3 B/elem) or finishes the open tiles
 * with the verdict (all of them when the data are non-negative: 5 B/elem, what the composed form moves).
 * Everything else is A4 -> A5 -> A1 enqueued back to back through `workspace`; a per-tensor call with `ticket` folds A5 into the
 * reduction's last block (two launches instead of three). Same values on every route.
 * `ticket` (nullable): TWO int32, ZERO before the first call, left zero by every call: one pair per stream.
 */
size_t ffq_quantize_dynamic_workspace_bytes(const ffq_tiling* tiling, int data_dt);
int ffq_quantize_dynamic_by_tile(const void* data, int data_dt, const ffq_tiling* tiling,
                                 double num_bits, int symmetric, int allow_one_sided, void* out,
                                 int out_dt, float* scale_out, float* offset_out, void* workspace,
                                 size_t workspace_bytes, int32_t* ticket, void* stream);

/*
 * A7 — sub-byte storage. The reference keeps 4-bit codes unpacked; its only packing convention is
 * GGUF Q4_0 (export/stages/gguf/_packing.py:44-53): within each block of `block` codes,
 *   byte[j] = (code[j] + 8) | ((code[j + block/2] + 8) << 4),  j in [0, block/2).
 * `codes` holds integer-valued elements in [-8, 7] in any supported dtype; numel % block == 0,
 * block even. unpack(pack(q)) == q exactly.
 */
int ffq_pack_int4(const void* codes, int codes_dt, int64_t numel, int64_t block, uint8_t* packed,
                  void* stream);
int ffq_unpack_int4(const uint8_t* packed, int64_t numel, int64_t block, void* codes_out,
                    int codes_dt, void* stream);

/*
 * A6 — the quantized linear that replaces `fallback.linear`, _gen/fallback.py:77-112 (reached via
 * ff.nn.functional.linear, _gen/operators.py:79-106, from QuantizedLinear.forward,
 * nn/linear.py:32-39). Integer codes in, real-valued output out:
 *   y[m,n] = sx[m'] * sw[n'] * sum_k (xq[m,k] + ox[m']) * (wq[n,k] + ow[n'])  (+ bias[n])
 * with ox/ow = round_half_even(offset) (A2), m' = m if x_per_row else 0, n' likewise.
 * The contraction runs on int8 MFMA with int32 accumulation; the zero-point terms use row sums.
 * If out_scale != NULL the result is re-quantized in the same launch (the `output_quantizer` of fallback.py:110-111,
 * per-tensor A1): y is rounded once to `y_dt` — the dtype the float GEMM of the reference returns, i.e. the input's
 * dequantize dtype (nn/linear.py:32-39) — and codes = clamp(rne(y / out_scale - rne(out_offset))) go to `out` in the
 * container `out_dt`; bit-identical to ffq_quantize_by_tile on the tensor the launch without out_scale writes. Else `out`
 * holds y in out_dt (bf16/f16/f32) and y_dt is ignored.
 * `w_rowsum` (nullable): sum_k wq[n, k] when the caller already has it (ffq_quantize_rows_rowsum) — one reduction launch
 * fewer, same result. A weight offset whose rounded entries are all zero (the offset BUFFER of a symmetric quantizer,
 * nn/linear_quantizer.py:164-170) is detected on the device and costs nothing but a 1-block check.
 * Tolerance vs the reference's bf16 eager path is stated in tests/parity_cases.py::check_linear (G6) and
 * tests/test_parity_gpu.py::test_w8a8_linear_*; exact-integer checks at full BASELINE sizes: tests/test_fullsize_gpu.py.
 * Workspace: ffq_linear_w8a8_workspace_bytes(M, N, K) bytes cover every launch. It may be NULL / 0 when w_offset == NULL and either
 * x_offset == NULL, or w_rowsum != NULL, or ffq_linear_w8a8_takes_earlier(M, N, K) == 0 (below the persistent kernel's shape class
 * the tile kernel sums its own weight rows); a launch that needs it and gets less returns FFQ_ERR_WORKSPACE before touching a buffer.
 */
size_t ffq_linear_w8a8_workspace_bytes(int64_t M, int64_t N, int64_t K);
int ffq_linear_w8a8(const int8_t* xq, const int8_t* wq, const int32_t* w_rowsum, const float* x_scale,
                    const float* x_offset, int x_per_row, const float* w_scale, const float* w_offset, int w_per_row,
                    const void* bias, int bias_dt, void* out, int out_dt, const float* out_scale,
                    const float* out_offset, double out_num_bits, int y_dt, int64_t M, int64_t N, int64_t K,
                    void* workspace, size_t workspace_bytes, void* stream);

/*
 * Two or three W8A8 linears on the SAME activation codes in one launch (ABI 9) — q_proj / k_proj / v_proj of an attention block: three
 * QuantizedLinear modules reading one quantized hidden state (reference nn/linear.py:32-39 three times over _gen/fallback.py:77-112).
 * The weight codes of the `count` matrices lie matrix after matrix in ONE [N, K] run (N = the sum of Ns), their scales (one per weight
 * row) and — optionally — their int32 row sums in ONE [N] run each; outs[i] is a separate [M, Ns[i]] tensor. outs[i] equals what
 * ffq_linear_w8a8 returns for matrix i alone, bit for bit (the same tiles, accumulators and epilogue; only the tile walk covers all
 * column tiles). Every matrix but the last needs a multiple of 256 rows; activation parameters per tensor or per row; no weight
 * offsets, bias or output quantizer. Problems the persistent kernel does not take (fewer than 64 tiles of 256 x 256, K % 128 != 0)
 * return FFQ_ERR_DTYPE before touching a buffer: launch the matrices one by one. Workspace: ffq_linear_w8a8_workspace_bytes(M, N, K).
 */
int ffq_linear_w8a8_multi(const int8_t* xq, const int8_t* wq, const int32_t* w_rowsum, const float* x_scale, const float* x_offset,
                          int x_per_row, const float* w_scale, int count, void* const* outs, int out_dt, int64_t M, const int64_t* Ns,
                          int64_t K, void* workspace, size_t workspace_bytes, void* stream);

/*
 * The second linear of a gated MLP on int8 codes: out = bf16(silu(gate)) * bf16(y), y = ffq_linear_w8a8(...) in bf16 —
 * silu(gate_proj(x)) * up_proj(x) of the reference's Llama MLP (docs/examples/doc_helpers/quantized_llama/mlp.py:36-38) formed in
 * up_proj's epilogue from gate_proj's stored bf16 result `gate` [M, N]: the up tensor and the SiLU * up pass (two reads, one write
 * of M x N bf16) never exist. Same values as the three-step chain (silu rounded to bf16, then one rounding of the product).
 * For launches whose down_proj input quantizer is not yet known (range estimation: ffq_mlp_gate_up_w8a8 needs its parameters).
 * bf16 output, no bias; other arguments as ffq_linear_w8a8. Covered: N % 64 == 0, K % 128 == 0, >= 64 output tiles of
 * 256 x 256; anything else returns FFQ_ERR_DTYPE and the caller takes ffq_linear_w8a8 + ffq_silu_mul_quantize.
 * `extrema_words` / `extrema_pair` (both or neither): the launch also leaves [min, max] of the product (bf16, NaN-propagating:
 * what ffq_minmax_by_tile over `out` as one tile returns) in `extrema_pair` — the first step of the next quantizer's RunningMinMax
 * estimator without a pass over the tensor. `extrema_words`: four uint32 holding {0xFFFFFFFF, 0, 0, 0} before the first launch
 * that uses them; every launch leaves them in that state (one buffer per stream).
 */
int ffq_linear_w8a8_gated(const int8_t* xq, const int8_t* wq, const int32_t* w_rowsum, const float* x_scale,
                          const float* x_offset, int x_per_row, const float* w_scale, const float* w_offset,
                          int w_per_row, const void* gate, void* out, int64_t M, int64_t N, int64_t K,
                          void* workspace, size_t workspace_bytes, uint32_t* extrema_words, void* extrema_pair, void* stream);

/*
 * bmm on int8 codes — the quantized-operand pattern of _gen/fallback.py:699-798 (dequantize both operands, the float op, the output
 * quantizer) for `batch` independent products out[b] = xq[b] [M, K] x wq[b]^T [N, K] with ONE parameter pair per operand (per-tensor
 * quantizers: the batch shares them), as ONE launch; arithmetic, epilogue and the optional fused output quantizer exactly as
 * ffq_linear_w8a8 (per matrix pair: bit-identical to `batch` calls of it). K % 16 == 0. Workspace: needed with w_offset != NULL only.
 */
size_t ffq_bmm_w8a8_workspace_bytes(int64_t batch, int64_t M, int64_t N, int64_t K);
int ffq_bmm_w8a8(const int8_t* xq, const int8_t* wq, const float* x_scale, const float* x_offset, const float* w_scale,
                 const float* w_offset, void* out, int out_dt, const float* out_scale, const float* out_offset,
                 double out_num_bits, int y_dt, int64_t batch, int64_t M, int64_t N, int64_t K, void* workspace,
                 size_t workspace_bytes, void* stream);

/*
 * A6 (weight-only) — the branch of `fallback.linear` taken by a quantized WEIGHT and a plain, non-quantized INPUT
 * (_gen/fallback.py:77-112 with strict_quantization off, :86-100): y = F.linear(x, weight.dequantize(), bias).
 * `x` is [M, K] bf16. `w_codes` holds the weight's integer codes (any num_bits <= 8) either one per byte
 * (`w_dt` = FFQ_I8, [N, K], `pack_block` = 0) or, for 4-bit codes, PACKED two per byte exactly as ffq_pack_int4 writes them
 * (`w_dt` = FFQ_U8, [N, K / 2], `pack_block` = the packing block: a power of two >= 32 dividing K — the GGUF Q4_0 nibble
 * order of export/stages/gguf/_packing.py:44-53 applied along each row; BASELINE config 4 packs with its group size 128).
 * `w_scale` / `w_offset` (nullable) hold `scale_numel` = 1 or N * (K / group) fp32 entries in tiles_to_rows order
 * ([N, K / group] row-major): group == K is per-tensor / PerChannel(0), group < K is PerBlock(block_dims=1,
 * block_sizes=group, per_channel_dims=0) (quantization/granularity.py:159-216; config 4: group 128).
 * The codes are dequantized with A2's arithmetic, (float(q) + round_half_even(o)) * s in fp32 rounded once to bf16 — the B
 * operand of the bf16 MFMA is bit for bit the reference's dequantized weight, accumulation is fp32; only the summation order
 * differs from F.linear. One launch converts the codes once per 256-row tile on their way into LDS; with enough `workspace`
 * behind the slabs (below) and from 4096 tokens on (from 1536 where the launch has at least 144 tiles of 256 x 256: what
 * ffq_linear_wq_workspace_bytes() sizes the scratch for) the weight is dequantized once per CALL by A2 into it and the GEMM streams
 * the bf16 image — same operands, same k order: bit-identical, faster at large M (whole-tile launches of that form with a bf16
 * output, no bias and K % 128 == 0 run a one-wave-per-SIMD kernel, 4 waves x 128 x 128 accumulators; also bit-identical).
 * Split-K (ABI 7): the kernel is one persistent block per CU on 256 x 256 output tiles; the tiles of the LAST, partly filled
 * round of that walk (all tiles when there are fewer than CUs: 2048 tokens x a 4096-wide projection are 128) have their K range
 * cut into ffq_linear_wq_split() slices, one work unit each on its own CU; the units of a tile exchange fp32 partial sums
 * through `workspace` and each finishes a fixed share of the tile in a fixed summation order — no token count is left to a
 * vendor GEMM. The split is a pure function of (M, N, K, CU count): results are reproducible call to call.
 *   workspace  [ffq_linear_wq_workspace_bytes()]: split-K slabs first, the two-pass image behind them. May be smaller or NULL:
 *              the launch then runs without the part that does not fit (never fails for lack of scratch).
 *   tickets    ffq_linear_wq_tickets() int32 counters, ZERO before the first launch that uses the buffer; every launch
 *              leaves them zero (keep one buffer per stream). NULL: no split.
 *   split      0 = the library's plan; >= 1 forces that many slices (tests, tuning): ffq_linear_wq_slab_bytes(..., split) bytes
 *              of workspace; FFQ_ERR_ARG if that scratch is missing or the units of the last round would not all fit the chip
 *              at once (they wait for each other: tail tiles * split <= CUs).
 * Few rows (ABI 8, M <= 128 with K % 256 == 0, or K % 128 == 0 for M <= 16 and M > 64; int8 containers and nibbles packed with block
 * 128): the contraction is a STREAM over the codes and takes its own kernel (csrc/ffq_wskinny.hip) — 16-row MFMA tiles, the codes
 * straight from HBM into the MFMA's register layout, no padding to 256 rows. Up to 16 rows a block owns 16 output columns for the
 * whole contraction (its eight waves share K); above that a block owns 128 columns of one K slice and the slices meet through
 * `workspace` / `tickets`: every wave takes a ticket for its strip and the LAST one to arrive adds the partial sums in slice order —
 * nobody waits for anybody. The plan queries below describe whichever kernel the launch takes (same arguments, same meaning).
 * Summation order: the fp32 order of a contraction — hence the last bit of an output — is a function of (M, N, K, CU count, split,
 * kernel): bit-reproducible launch to launch, equal across the storage forms that share a kernel, NOT equal across batch sizes or
 * across the kernels (every route stays within one output rounding of the float64 product of the same operands).
 * ffq_linear_wq_supported() == 0 (K % 64 != 0, K < 128, other dtypes, group % 64 != 0): the caller dequantizes (A2) and
 * runs a float GEMM, as the reference does.
 */
int ffq_linear_wq_supported(int x_dt, int w_dt, int out_dt, int64_t M, int64_t N, int64_t K, int64_t group, int64_t pack_block);
int64_t ffq_linear_wq_split(int64_t M, int64_t N, int64_t K, int mlp);   /* mlp != 0: the plan of ffq_mlp_gate_up_wq */
int64_t ffq_linear_wq_tickets(int64_t M, int64_t N, int64_t K, int mlp);
size_t ffq_linear_wq_slab_bytes(int64_t M, int64_t N, int64_t K, int mlp, int64_t split);
size_t ffq_linear_wq_workspace_bytes(int64_t M, int64_t N, int64_t K);
int ffq_linear_wq(const void* x, int x_dt, const void* w_codes, int w_dt, int64_t pack_block, const float* w_scale,
                  const float* w_offset, int64_t scale_numel, int64_t group, const void* bias, int bias_dt, void* out,
                  int out_dt, int64_t M, int64_t N, int64_t K, void* workspace, size_t workspace_bytes, int32_t* tickets,
                  int64_t split, void* stream);

/*
 * Two or three weight matrices on the SAME activations in one launch — q_proj / k_proj / v_proj of an attention block are three
 * QuantizedLinear modules reading one hidden state (nn/linear.py:32-39 three times; docs/examples/doc_helpers/quantized_llama/
 * attention.py:45-60): outs[i] = F.linear(x, dequantize(w_i)), the value ffq_linear_wq gives for each matrix (same operands; the
 * summation order is that of ONE launch over the concatenated columns), written to `count` separate [M, Ns[i]] tensors. One tile
 * walk covers all column tiles, so a 1024-column k/v projection no longer runs alone on a corner of the chip. All matrices share
 * dtype, packing, `group`, the granularity kind (`per_row` != 0: Ns[i] * K / group parameter pairs each; 0: one pair each,
 * group == K) and the presence of offsets; every matrix but the last has a multiple of 256 rows. workspace / tickets / split as
 * ffq_linear_wq with N = sum(Ns) (ffq_linear_wq_workspace_bytes(M, N, K), ...).
 */
int ffq_linear_wq_multi(const void* x, int x_dt, int count, const void* const* w_codes, int w_dt, int64_t pack_block,
                        const float* const* w_scale, const float* const* w_offset, int per_row, int64_t group, void* const* outs,
                        int out_dt, int64_t M, const int64_t* Ns, int64_t K, void* workspace, size_t workspace_bytes, int32_t* tickets,
                        int64_t split, void* stream);

/*
 * The MLP front half of a WEIGHT-ONLY quantized Llama (docs/examples/doc_helpers/quantized_llama/mlp.py:30-40 with plain bf16
 * activations: BASELINE configs 2 and 4) in one launch: gate_proj and up_proj as ffq_linear_wq computes them (same operands, same
 * summation order), then bf16(silu(bf16(gate))) * bf16(up) rounded to bf16 —
 *   out == product of ffq_silu_mul_quantize(ffq_linear_wq(x, gate), ffq_linear_wq(x, up))    bit for bit,
 * without the two [M, N] bf16 projections ever visiting HBM. Both weights share dtype, packing, granularity (`scale_numel`,
 * `group`); offsets both or neither; N % 128 == 0; otherwise as ffq_linear_wq. `out` is [M, N] bf16. `workspace`
 * (ffq_mlp_gate_up_wq_workspace_bytes(): split-K slabs, then both bf16 images of the two-pass form from 4096 tokens on),
 * `tickets` (ffq_linear_wq_tickets(..., 1)) and `split` as in ffq_linear_wq; the equality above holds bit for bit when both
 * sides run the same split (ffq_linear_wq_split(M, N, K, 1) vs (..., 0)), else to one rounding of the fp32 sums.
 */
size_t ffq_mlp_gate_up_wq_workspace_bytes(int64_t M, int64_t N, int64_t K);
int ffq_mlp_gate_up_wq(const void* x, int x_dt, const void* gate_codes, const void* up_codes, int w_dt, int64_t pack_block,
                       const float* gate_scale, const float* gate_offset, const float* up_scale, const float* up_offset,
                       int64_t scale_numel, int64_t group, void* out, int64_t M, int64_t N, int64_t K, void* workspace,
                       size_t workspace_bytes, int32_t* tickets, int64_t split, void* stream);

/*
 * Producer-fused A1 (ABI version 2). In the reference's quantized Llama helpers
 * (docs/examples/doc_helpers/quantized_llama/) every quantized linear quantizes its own input
 * (nn/linear.py:33), and those inputs leave three elementwise producers that run as eager ATen
 * chains. Each entry point below is one pass: the bf16 value the eager chain would have stored is
 * formed with the same per-op roundings, optionally written, and quantized (A1, static per-tensor
 * parameters, int8 container) for up to FFQ_MAX_FANOUT consumers. Quantizers holding equal
 * parameters get equal codes, as they would from separate A1 calls.
 */
#define FFQ_MAX_FANOUT 3
typedef struct ffq_fanout {
  int32_t count;                          /* 0..FFQ_MAX_FANOUT quantizers fed by this producer     */
  double num_bits;                        /* shared bit-width (1..8, integral)                      */
  const float* scale[FFQ_MAX_FANOUT];     /* one fp32 scale each (per-tensor)                       */
  const float* offset[FFQ_MAX_FANOUT];    /* nullable: zeros                                        */
  int8_t* codes[FFQ_MAX_FANOUT];          /* outputs, same numel as the producer's result           */
} ffq_fanout;

/*
 * RMSNorm behind a residual add — rms_norm.py:17-35 after decoder.py:60-90:
 *   sum  = x + delta                     (bf16; skipped when delta == NULL, then sum = x)
 *   h    = float(sum); h = h * rsqrt(mean(h^2, -1) + eps)          (fp32)
 *   z    = weight * bf16(h)              (bf16)
 *   codes_j = A1(z; scale_j, offset_j)
 * x, delta, sum_out, norm_out: [rows, cols] of `dt` (bf16 only); weight: [cols]. sum_out and
 * norm_out are nullable (sum_out may alias x). The fp32 summation order of mean(h^2) is the
 * kernel's own, so z can differ from the eager chain by one bf16 ulp on rare elements; the codes are
 * exactly A1 of the z this call produces. cols % 16 == 0, cols <= 8192.
 */
int ffq_add_rmsnorm_quantize(const void* x, const void* delta, void* sum_out, const void* weight,
                             int dt, int64_t rows, int64_t cols, double eps, void* norm_out,
                             const ffq_fanout* fan, void* stream);

/*
 * SiLU(gate) * up — mlp.py:30-40:  z = bf16(silu(float(gate))) * up  (bf16), codes = A1(z).
 * silu(v) = v / (1 + exp(-v)) in fp32, as ATen evaluates it. product_out is nullable. numel % 16 == 0.
 */
int ffq_silu_mul_quantize(const void* gate, const void* up, int dt, int64_t numel, void* product_out,
                          const ffq_fanout* fan, void* stream);

/*
 * Rotary position embedding in place — attention.py:20-41 (apply_rotary_pos_emb):
 *   out = bf16(bf16(v * cos) + bf16(rotate_half(v) * sin)),  rotate_half(v) = cat(-v[D/2:], v[:D/2])
 * q: [tokens, q_heads, head_dim], k: [tokens, k_heads, head_dim] as they leave the projections;
 * cos/sin: [seq_len, head_dim]; the position of token t is t % seq_len. head_dim % 16 == 0.
 */
int ffq_rope_inplace(void* q, int64_t q_heads, void* k, int64_t k_heads, int dt, int64_t tokens,
                     int64_t seq_len, int64_t head_dim, const void* cos_table, const void* sin_table,
                     void* stream);

/*
 * A8 — fastforward::quantize_by_tile_backward, _quantizer_impl.py:193-237 (affine/_autograd.py:99): the
 * straight-through / LSQ gradients of quantize -> dequantize.
 *   u = x / s_t - round(o_t);  q = round(u);  clip = (q < lo) | (q > hi)
 *   dinput  = clip ? 0 : g                                   (dtype of data / output_grad, `dt`)
 *   doffset = sum over the tile of (clip ? s_t * g : 0)      (fp32, one per tile; only when offset != NULL)
 *   dscale  = sum over the tile of (clip ? (q < lo ? lo : hi) + round(o_t) : q - u) * g   (fp32, one per tile)
 * scale / offset are fp32 with one entry per tile. The elementwise part is exact; the per-tile sums are
 * fp32 sums in this implementation's own (fixed, deterministic) order. Tilings covered: one tile
 * (per-tensor) and contiguous-run tiles (per-channel on dim 0, per-block along the last dim,
 * per-token) with numel % 8 == 0; anything else returns FFQ_ERR_DTYPE. ffq_quantize_backward_workspace_bytes is 0 for the problems
 * that run as one launch (up to 65536 elements in >= 4 tiles of <= 4096 elements, and every tiling of the by-tile kernel).
 */
size_t ffq_quantize_backward_workspace_bytes(const ffq_tiling* tiling);
int ffq_quantize_by_tile_backward(const void* data, const void* output_grad, int dt, const float* scale,
                                  int64_t scale_numel, const float* offset, int64_t offset_numel,
                                  const ffq_tiling* tiling, double num_bits, void* dinput, float* dscale,
                                  float* doffset, void* workspace, size_t workspace_bytes, void* stream);

/*
 * A1 + A7 and A7 + A2 in one pass each (4-bit, the W4 PerBlock(128) configuration): the codes never make
 * an HBM round trip. packed == ffq_pack_int4(ffq_quantize_by_tile(data, num_bits = 4, int8 container)),
 * out == ffq_dequantize_by_tile(ffq_unpack_int4(packed)) exactly. fp32 parameters; per-tensor or
 * contiguous-run tiles with run % 16 == 0; block % 32 == 0; anything else returns FFQ_ERR_DTYPE and the
 * caller composes the two-step path.
 */
int ffq_quantize_pack_int4(const void* data, int data_dt, const float* scale, int64_t scale_numel,
                           const float* offset, int64_t offset_numel, const ffq_tiling* tiling, int64_t block,
                           uint8_t* packed, void* stream);
int ffq_unpack_dequantize_int4(const uint8_t* packed, const float* scale, int64_t scale_numel, const float* offset,
                               int64_t offset_numel, const ffq_tiling* tiling, int64_t block, void* out, int out_dt,
                               void* stream);

/*
 * Inner loop of the min-error (MSE grid) range estimator — _MinAvgErrorGridEstimator.estimate_step,
 * range_setting/min_error.py:218-231 with error_fn = mse_error (:62-72): for each of `ncand` candidate
 * parameter sets (scales / offsets: [ncand, ntiles] fp32, offsets nullable) the SUM over every tile of
 *   ( cast<dt>( cast<dt>( dequantize(quantize(x)) ) - x ) )^2   (each step rounded to `dt` as the eager chain does)
 * is written to (accumulate == 0) or added to (accumulate != 0) err[ncand, ntiles] (fp32). The caller divides
 * by the tile size for the mean. One pass over the data for all candidates; the per-tile sums are fp32 sums in
 * a fixed, implementation-defined order. Tilings covered: one tile, and contiguous runs of 8 * 2^k <= 512 or a
 * multiple of 2048 elements; anything else returns FFQ_ERR_DTYPE and the caller loops over A1 / A2.
 */
size_t ffq_grid_sqerror_workspace_bytes(const ffq_tiling* tiling, int64_t ncand);
int ffq_grid_sqerror_by_tile(const void* data, int dt, const float* scales, const float* offsets, int64_t ncand,
                             const ffq_tiling* tiling, double num_bits, float* err, int accumulate, void* workspace,
                             size_t workspace_bytes, void* stream);

/*
 * The MLP front half of the reference's quantized Llama (docs/examples/doc_helpers/quantized_llama/mlp.py:30-40)
 * in ONE launch: gate_proj and up_proj (two A6 linears on the same activation codes), SiLU(gate) * up, and the
 * input quantizer of down_proj (nn/linear.py:33):
 *   codes = A1( bf16(silu(bf16(gate_linear(x)))) * bf16(up_linear(x)) ;  out_scale, out_offset )
 * == ffq_silu_mul_quantize(ffq_linear_w8a8(x, gate -> bf16), ffq_linear_w8a8(x, up -> bf16)) exactly, without the two
 * bf16 projections ever visiting HBM. x: per-tensor (scale, offset); weights: per-output-channel scales, no offset
 * (symmetric); N % 128 == 0, K % 128 == 0, K >= 256. gate_rowsum / up_rowsum (nullable, both or neither): the row
 * sums of the weight codes when the caller has them.
 */
size_t ffq_mlp_gate_up_w8a8_workspace_bytes(int64_t M, int64_t N, int64_t K);
int ffq_mlp_gate_up_w8a8(const int8_t* xq, const int8_t* gate_wq, const int8_t* up_wq, const int32_t* gate_rowsum,
                         const int32_t* up_rowsum, const float* x_scale, const float* x_offset,
                         const float* gate_w_scale, const float* up_w_scale, int8_t* codes_out, const float* out_scale,
                         const float* out_offset, double out_num_bits, int64_t M, int64_t N, int64_t K,
                         void* workspace, size_t workspace_bytes, void* stream);

/*
 * The gated MLP up to its product while gate_proj's and up_proj's input quantizers are being calibrated (range estimation:
 * `ff.estimate_ranges` around QuantizedLlamaMLP.forward, mlp.py:30-40): product_out [M, N] bf16 = bf16(silu(bf16 gate)) * bf16(up)
 * with gate = linear(xq_gate; gate weights), up = linear(xq_up; up weights) — two activation code tensors, one per input
 * quantizer (nn/linear.py:33), per-tensor parameters; weights per output channel with nullable offset buffers. The one-launch
 * mode above needs ONE set of activation codes and offset-free weights: whether the two quantizers hold equal parameters (they
 * do whenever both have seen the same data) and whether the offset buffers are all zero is decided ON THE DEVICE, and both routes
 * are enqueued with that flag as their predicate — one launch (gate + up + SiLU * up, the product left unquantized), or
 * gate_proj's linear into `gate_scratch` [M, N] bf16 followed by ffq_linear_w8a8_gated. Same values either way (the product of
 * the two-tensor chain); no host read. `extrema_words` / `extrema_pair` as in ffq_linear_w8a8_gated.
 * Shapes: N % 128 == 0, K % 128 == 0, K >= 256, >= 64 output tiles of 256 x 256; else FFQ_ERR_DTYPE.
 */
size_t ffq_mlp_gate_up_w8a8_estimating_workspace_bytes(int64_t M, int64_t N, int64_t K);
int ffq_mlp_gate_up_w8a8_estimating(const int8_t* xq_gate, const int8_t* xq_up, const int8_t* gate_wq, const int8_t* up_wq,
                                    const float* x_scale_gate, const float* x_offset_gate, const float* x_scale_up,
                                    const float* x_offset_up, const float* gate_w_scale, const float* gate_w_offset,
                                    const float* up_w_scale, const float* up_w_offset, void* gate_scratch, void* product_out,
                                    int64_t M, int64_t N, int64_t K, void* workspace, size_t workspace_bytes,
                                    uint32_t* extrema_words, void* extrema_pair, void* stream);

/*
 * Sibling quantizers while their ranges are being estimated (ABI version 8). q_proj / k_proj / v_proj (gate_proj / up_proj) each
 * quantize the SAME hidden state with their own input quantizer (nn/linear.py:32-39; after every estimator step the parameters
 * are new, range_setting/common.py:218-238), and A1's codes are a function of the scale's bits and the ROUNDED offset
 * (_quantizer_impl.py:140-141). Whether a later quantizer's pair equals an earlier one's is decided on the device:
 *   ffq_quantize_by_tile_unless_same   A1 of a per-tensor quantizer (`scale`, nullable `offset`: one fp32 each) into an int8
 *       container — unless (bits of scale == bits of earlier_scale) and (round_half_even(offset) == round_half_even(earlier_offset)):
 *       then nothing is read and `out` keeps what it held. Written codes are ffq_quantize_by_tile's. Whole 16-element chunks of
 *       f32 / bf16 / f16 data, 16-byte aligned; else FFQ_ERR_DTYPE before anything is touched.
 *   ffq_linear_w8a8_earlier            ffq_linear_w8a8 (per-tensor activation parameters, no bias, real-valued output) whose
 *       activation codes come from such a launch: it reads `earlier_xq` where the same comparison holds (they ARE this linear's
 *       codes then, written or not) and `xq` where it does not — row sums of the activation codes included. Shapes of
 *       ffq_linear_w8a8_takes_earlier(M, N, K) == 1 only (else FFQ_ERR_DTYPE); workspace of ffq_linear_w8a8_workspace_bytes always.
 * ffq_mlp_gate_up_w8a8_estimating treats `xq_up` the same way with gate_proj's codes and parameters as the earlier ones.
 * Same values as every quantizer quantizing and every linear reading its own codes; no host read.
 */
int ffq_quantize_by_tile_unless_same(const void* data, int data_dt, const float* scale, const float* offset, int64_t numel,
                                     double num_bits, const float* earlier_scale, const float* earlier_offset, int8_t* out,
                                     void* stream);
int ffq_linear_w8a8_takes_earlier(int64_t M, int64_t N, int64_t K);
int ffq_linear_w8a8_earlier(const int8_t* xq, const int8_t* earlier_xq, const float* earlier_scale, const float* earlier_offset,
                            const int8_t* wq, const int32_t* w_rowsum, const float* x_scale, const float* x_offset,
                            const float* w_scale, const float* w_offset, int w_per_row, void* out, int out_dt, int64_t M,
                            int64_t N, int64_t K, void* workspace, size_t workspace_bytes, void* stream);

/*
 * GGUF block-32 records — pack_q4_0_blocks / pack_q8_0_blocks, export/stages/gguf/_packing.py:23-72: `codes` is
 * [nblocks, 32] int8 in FastForward's signed convention, `scales` [nblocks] fp32 (positive). format 4 -> Q4_0:
 * 18 bytes per block = fp16(scale) then byte[j] = (code[j] + 8) | (code[j + 16] + 8) << 4 (nibbles clamped to
 * [0, 15]); format 8 -> Q8_0: 34 bytes per block = fp16(scale) then the 32 codes clipped to [-127, 127].
 * `out` holds nblocks * 18 (34) bytes that llama.cpp dequantizes as d * (qs - 8) (d * qs).
 */
int ffq_pack_gguf_blocks(const int8_t* codes, const float* scales, int64_t nblocks, int format, uint8_t* out, void* stream);

/*
 * The inner loop of GPTQ for one block of columns — gptq(), quantization/gptq.py:101-131, with the per-row
 * quantize-dequantize of column_quantizer (:149-235) for PerTensor / PerChannel(0) weight quantizers:
 *   for j in 0 .. block_cols:   q = dequantize(quantize(w[:, col0 + j]));  e = (w[:, col0 + j] - q) / hinv[col0 + j, col0 + j]
 *                               quantized[:, col0 + j] = q;  errors[:, col0 + j] = e
 *                               w[:, col0 + k] -= e * hinv[col0 + j, col0 + k]     for j < k < block_cols (block-local copy)
 * All matrices fp32: weights / quantized / errors [rows, row_stride], hinv [n, hinv_stride] (the upper Cholesky factor of
 * the inverse Hessian). `weights` is read only (the reference updates a clone of the block); the caller applies the
 * trailing update weights[:, col0 + block_cols:] -= errors[:, block] @ hinv[block, col0 + block_cols:] (:133) as a GEMM.
 * block_cols <= 128; scale / offset: one per row or one in total.
 */
int ffq_gptq_block(float* weights, float* quantized, float* errors, int64_t rows, int64_t row_stride,
                   int64_t col0, int64_t block_cols, const float* hinv, int64_t hinv_stride, const float* scale,
                   int64_t scale_numel, const float* offset, int64_t offset_numel, double num_bits, void* stream);

/*
 * Weight codes and their row sums in one pass (ABI version 3). A6's zero-point term needs sum_k wq[n, k] for every weight row
 * whenever the activation quantizer has an offset; with the reference's semantics the weight is re-quantized on every
 * forward (nn/linear.py:34), so the sums change with it. ffq_quantize_rows_rowsum is A1 for a [rows, cols] weight with one
 * (scale, offset) per row (PerChannel(0), `offset` nullable), int8 container, that also ADDS sum_k codes[r, k] to rowsum[r]
 * (exact integers: int32 atomics; the caller zeroes `rowsum` first — one fill for all the weights of a forward); codes are bit-identical to ffq_quantize_by_tile. bf16 data, cols % 1024 == 0, anything
 * else returns FFQ_ERR_DTYPE and the caller takes ffq_quantize_by_tile. The two GEMM entry points take such sums (nullable:
 * NULL = compute them from the codes) instead of launching their own reduction; results are identical.
 */
int ffq_quantize_rows_rowsum(const void* data, int data_dt, const float* scale, const float* offset, int64_t rows,
                             int64_t cols, double num_bits, int8_t* codes, int32_t* rowsum, void* stream);

/*
 * A1 of several row-quantized weights in ONE launch: member i is a [rows[i], cols[i]] bf16 matrix with one (scale, offset) per
 * row (PerChannel(0); `offset[i]` nullable), codes into an int8 container — bit-identical to ffq_quantize_by_tile on each member.
 * The reference re-quantizes every linear's weight on every forward (nn/linear.py:34): a decoder layer's seven weights are one
 * call here instead of seven launches, two of which (k_proj / v_proj) are too short to reach the streaming rate on their own.
 * rows[i] * cols[i] must be a multiple of 4096 and cols[i] of 16, else FFQ_ERR_DTYPE (the caller quantizes member by member).
 * `rowsum[i]` (ABI 7; for every member or for none; cols[i] % 1024 == 0): += sum_k codes[row, k] into int32 entries that are ZERO
 * on entry — what ffq_linear_w8a8 / ffq_mlp_gate_up_w8a8 take as `w_rowsum` instead of launching their own reduction per linear.
 */
#define FFQ_MAX_BATCH 8
typedef struct {
  int32_t count;
  double num_bits;
  const void* data[FFQ_MAX_BATCH];
  const float* scale[FFQ_MAX_BATCH];
  const float* offset[FFQ_MAX_BATCH];
  int8_t* codes[FFQ_MAX_BATCH];
  int64_t rows[FFQ_MAX_BATCH];
  int64_t cols[FFQ_MAX_BATCH];
  int32_t* rowsum[FFQ_MAX_BATCH];
} ffq_rows_batch;
int ffq_quantize_rows_batch(const ffq_rows_batch* batch, int data_dt, void* stream);

/*
 * The attention between q/k/v_proj and o_proj of the reference's quantized Llama —
 * docs/examples/doc_helpers/quantized_llama/attention.py:45-92 (repeat_kv, matmul, * scaling, causal mask, fp32
 * softmax, cast, matmul; the attn_weights / attn_probs / attn_output quantizers are stubs in the recipe) — with the
 * input quantizer of o_proj (nn/linear.py:33 -> A1, static per-tensor) applied to the context in the same launch:
 *   ctx[b, s, h, :] = softmax_j( q[b, s, h, :] . k[b, j, h / (q_heads / kv_heads), :] * softmax_scale ;  j <= s if causal ) @ v
 *   codes = A1(ctx; out_scale, out_offset, out_num_bits)          (bit-identical to ffq_quantize_by_tile on this ctx)
 * q: [batch, seq_len, q_heads, head_dim], k / v: [batch, seq_len, kv_heads, head_dim] as they leave the projections
 * (rotary embedding already applied), ctx_out / codes_out: [batch, seq_len, q_heads * head_dim]; either output is
 * nullable. bf16 operands, fp32 accumulation and softmax (one pass, online softmax; no seq x seq matrix in HBM), so
 * ctx agrees with the eager chain to bf16 rounding (tolerance in tests/parity_cases.py::check_attention).
 * head_dim == 128, seq_len % 64 == 0, bf16 only: anything else returns FFQ_ERR_DTYPE.
 * `q_cos` / `q_sin` (both or neither; ABI 8): the rotary tables [seq_len, head_dim] of ffq_rope_inplace — q then arrives
 * UN-rotated and is rotated as the kernel loads it (the same arithmetic: equal to ffq_rope_inplace on q followed by this call
 * without tables, bit for bit); k is rotated by the caller (ffq_rope_inplace with q_heads = 0).
 */
int ffq_attention(const void* q, const void* k, const void* v, int dt, int64_t batch, int64_t seq_len,
                  int64_t q_heads, int64_t kv_heads, int64_t head_dim, double softmax_scale, int causal,
                  void* ctx_out, int8_t* codes_out, const float* out_scale, const float* out_offset,
                  double out_num_bits, const void* q_cos, const void* q_sin, void* stream);

/*
 * Test hook. The streaming kernels (fp32 parameters, one 16-byte chunk per lane, Markstein division) and the generic kernels
 * (any dtype mix / tiling, one element per lane, the compiler's IEEE division, every eager rounding reproduced) must agree
 * wherever both apply; ffq_force_generic_kernels(1) routes A1 / A2 / A4 to the generic family until it is called with 0 — and the
 * bf16-image form of ffq_linear_wq / ffq_mlp_gate_up_wq to its 8-wave kernel instead of the one-wave-per-SIMD one (bit-equal results),
 * and plain launches of up to 512 rows to the 256-row tiles. Bit 1 (2): odd K slices of a split tile abandon their wait at once.
 * Bit 2 (4): the 128-column tiles (up to 512 rows) take their register-staged kernel instead of the LDS-DMA one (bit-equal results).
 * Returns the previous setting. Process-wide; the library reads no environment variables.
 */
int ffq_force_generic_kernels(int on);

#ifdef __cplusplus
}
#endif
#endif /* FFQ_H */

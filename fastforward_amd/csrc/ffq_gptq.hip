// ffq_gptq.hip — the inner loop of GPTQ for one block of columns.
//
// Reference: gptq(), src/fastforward/quantization/gptq.py:101-136 — for each of the (up to 128) columns of a block:
//   q_j   = dequantize(quantize(w_j))                       (column_quantizer, :149-235: one scale/offset per ROW)
//   e_j   = (w_j - q_j) / Hinv[j, j]
//   w_k  -= e_j * Hinv[j, k]   for the block's remaining columns k > j
// run as ~5 eager launches per column (640 per block, 20 k per 4096-column weight) on [rows] vectors. Rows are
// independent, so one lane owns one row: its block of weights lives in registers (the column loop is fully
// unrolled), the Hinv block lives in LDS and every read of it is a wave-wide broadcast. Each arithmetic step is the
// fp32 operation the eager chain performs (the [rows,1] @ [1,n] update is one multiply and one subtract per element).
#include "ffq_common.h"
#include "ffq_vec.h"

#include <math.h>

namespace ffq {

constexpr int kGptqBlock = 128;

struct GptqArgs {
  float* weights;            // [rows, row_stride], the block starts at column col0; updated columns are NOT written back
  float* quantized;          // [rows, row_stride] out: columns col0 .. col0 + bs
  float* errors;             // [rows, row_stride] out: columns col0 .. col0 + bs
  const float* hinv;         // [n, n] upper Cholesky factor of the inverse Hessian; the block is hinv[col0:, col0:]
  const float* scale;        // [rows] or [1]
  const float* offset;       // nullable, [rows] or [1]
  int rows, bs, col0;
  int64_t row_stride, hinv_stride;
  int scale_stride, offset_stride;
  float lo, hi;
};

__global__ __launch_bounds__(kBlock) void gptq_block_kernel(GptqArgs a) {
  __shared__ float h[kGptqBlock * kGptqBlock];  // 64 KiB: row j holds Hinv[col0 + j, col0 + (0..127)]
  for (int idx = threadIdx.x; idx < a.bs * kGptqBlock; idx += kBlock) {
    const int j = idx / kGptqBlock, k = idx - j * kGptqBlock;
    h[idx] = k < a.bs ? a.hinv[(size_t)(a.col0 + j) * a.hinv_stride + a.col0 + k] : 0.0f;
  }
  __syncthreads();
  const int row = blockIdx.x * kBlock + threadIdx.x;
  if (row >= a.rows) return;
  float* wrow = a.weights + (size_t)row * a.row_stride + a.col0;
  float w[kGptqBlock];
#pragma unroll
  for (int k = 0; k < kGptqBlock; ++k) w[k] = k < a.bs ? wrow[k] : 0.0f;
  const float s = a.scale[row * a.scale_stride];
  const float o = a.offset ? rne(a.offset[row * a.offset_stride]) : 0.0f;
  float* qrow = a.quantized + (size_t)row * a.row_stride + a.col0;
  float* erow = a.errors + (size_t)row * a.row_stride + a.col0;
#pragma unroll
  for (int j = 0; j < kGptqBlock; ++j) {
    if (j < a.bs) {  // block-uniform
      const float x = w[j];
      float q = rne(x / s - o);                    // quantize: _quantizer_impl.py:161-162
      q = clamp_nan(q, a.lo, a.hi);
      const float dq = (q + o) * s;                // dequantize: :186
      const float e = (x - dq) / h[j * kGptqBlock + j];
      qrow[j] = dq;
      erow[j] = e;
#pragma unroll
      for (int k = j + 1; k < kGptqBlock; ++k) w[k] = w[k] - e * h[j * kGptqBlock + k];  // columns >= bs carry zeros
    }
  }
}

}  // namespace ffq

using namespace ffq;

extern "C" int ffq_gptq_block(float* weights, float* quantized, float* errors, int64_t rows, int64_t row_stride,
                              int64_t col0, int64_t block_cols, const float* hinv, int64_t hinv_stride, const float* scale,
                              int64_t scale_numel, const float* offset, int64_t offset_numel, double num_bits, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (rows < 0 || block_cols < 0 || col0 < 0) return fail(FFQ_ERR_ARG, "negative extent");
  if (block_cols > kGptqBlock) return fail(FFQ_ERR_DTYPE, "GPTQ block kernel handles at most %d columns per block", kGptqBlock);
  if (rows == 0 || block_cols == 0) return FFQ_OK;
  if (!weights || !quantized || !errors || !hinv || !scale) return fail(FFQ_ERR_ARG, "NULL buffer");
  if ((scale_numel != 1 && scale_numel != rows) || (offset && offset_numel != 1 && offset_numel != rows))
    return fail(FFQ_ERR_PARAM_NUMEL, "GPTQ block kernel takes one scale / offset per row (or one in total)");
  if (rows >= ((int64_t)1 << 31) || col0 + block_cols > row_stride || col0 + block_cols > hinv_stride) return fail(FFQ_ERR_ARG, "block outside the matrix");
  GptqArgs a;
  a.weights = weights; a.quantized = quantized; a.errors = errors; a.hinv = hinv; a.scale = scale; a.offset = offset;
  a.rows = (int)rows; a.bs = (int)block_cols; a.col0 = (int)col0;
  a.row_stride = row_stride; a.hinv_stride = hinv_stride;
  a.scale_stride = scale_numel == 1 ? 0 : 1;
  a.offset_stride = offset_numel == 1 ? 0 : 1;
  const double lo = -pow(2.0, num_bits - 1.0);
  a.lo = (float)lo; a.hi = (float)(-lo - 1.0);
  gptq_block_kernel<<<(unsigned)((rows + kBlock - 1) / kBlock), kBlock, 0, s>>>(a);
  return check_launch("gptq_block_kernel");
}

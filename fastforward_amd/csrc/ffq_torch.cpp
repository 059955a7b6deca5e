// libffq_torch.so — the operators of the reference's registry (fastforward::quantize_by_tile, dequantize_by_tile,
// quantize_dynamic_by_tile, quantize_by_tile_backward; reference quantization/_quantizer_impl.py:144-285) and the hot entry
// points behind its dispatcher and range estimator (the quantized linear / bmm of _gen/fallback.py:77-112, 699-798; one
// RunningMinMax step, range_setting/minmax.py:215-239) registered for the HIP dispatch key in C++, straight on top of the C ABI
// of libffq_hip.so (include/ffq.h). Round 4 had the two static operators here; round 5 adds the rest: the dynamic operator
// raises the package's own QuantizationError through the CPython API (:259-264), the backward carries its tensor-op
// composite for the tilings the kernel does not cover as ATen calls.
//
// The schemas are defined in fastforward_amd/ops.py (torch.library "fastforward_amd"); this file only adds the device
// kernels, so `torch.ops.fastforward_amd.*` on a HIP tensor goes dispatcher -> this file -> ffq_* without entering the Python
// interpreter or ctypes. Host tensors never reach these functions (the key is the device key); there is no CPU path here.
// PyTorch is plumbing: output allocation, the current stream and the device guard. Everything else is the C ABI.
//
// Host C++ only (g++): no kernels in this file.
#include <Python.h>

#include <ATen/ATen.h>
#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPGraphsC10Utils.h>
#include <c10/hip/HIPStream.h>
#include <torch/csrc/Exceptions.h>
#include <torch/library.h>

#include <map>
#include <mutex>
#include <optional>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/ffq.h"

namespace {

int tag_of(at::ScalarType t) {
  switch (t) {
    case at::kFloat: return FFQ_F32;
    case at::kBFloat16: return FFQ_BF16;
    case at::kHalf: return FFQ_F16;
    case at::kDouble: return FFQ_F64;
    case at::kChar: return FFQ_I8;
    case at::kShort: return FFQ_I16;
    case at::kInt: return FFQ_I32;
    case at::kLong: return FFQ_I64;
    case at::kByte: return FFQ_U8;
    default:
      TORCH_CHECK_NOT_IMPLEMENTED(false, "fastforward_amd: dtype ", c10::toString(t), " is not supported by the HIP backend");
  }
}

at::ScalarType dtype_of(int tag) {
  switch (tag) {
    case FFQ_F32: return at::kFloat;
    case FFQ_BF16: return at::kBFloat16;
    case FFQ_F16: return at::kHalf;
    case FFQ_F64: return at::kDouble;
    case FFQ_I8: return at::kChar;
    case FFQ_I16: return at::kShort;
    case FFQ_I32: return at::kInt;
    case FFQ_I64: return at::kLong;
    case FFQ_U8: return at::kByte;
    default: TORCH_CHECK(false, "fastforward_amd: unknown dtype tag ", tag);
  }
}

// fastforward_amd.exceptions.QuantizationError (reference exceptions.py:5; raised by the dynamic operator for an empty input,
// _quantizer_impl.py:259-264) set as the pending Python error; the dispatcher's Python binding re-raises it as it is. Outside
// an interpreter (a C++ caller of the operator registry) the error is a c10::Error with the same text.
bool g_reference_errors = false;

[[noreturn]] void raise_quantization_error(const std::string& message) {
  if (Py_IsInitialized()) {
    PyGILState_STATE gil = PyGILState_Ensure();
    PyObject* module = PyImport_ImportModule("fastforward_amd.exceptions");
    PyObject* type = module ? PyObject_GetAttrString(module, "QuantizationError") : nullptr;
    if (g_reference_errors) {  // installed under the reference's operator names: the reference's own exception class
      PyObject* theirs = PyImport_ImportModule("fastforward.exceptions");
      PyObject* their_type = theirs ? PyObject_GetAttrString(theirs, "QuantizationError") : nullptr;
      if (their_type) { Py_XDECREF(type); type = their_type; } else { PyErr_Clear(); }
      Py_XDECREF(theirs);
    }
    if (type) {
      PyErr_SetString(type, message.c_str());
      Py_DECREF(type);
      Py_XDECREF(module);
      PyGILState_Release(gil);
      throw python_error();
    }
    PyErr_Clear();
    Py_XDECREF(module);
    PyGILState_Release(gil);
  }
  TORCH_CHECK(false, message);
}

// The exception the reference raises for each status (the comments beside ffq_status in include/ffq.h; the same table as
// fastforward_amd/_cabi.py).
void check(int status) {
  if (status == FFQ_OK) return;
  const char* text = ffq_last_error();
  std::string message = (text && *text) ? std::string(text) : "ffq status " + std::to_string(status);
  switch (status) {
    case FFQ_ERR_EMPTY: raise_quantization_error(message);
    case FFQ_ERR_TILE_RANK:
    case FFQ_ERR_TILE_DIVIDE:
    case FFQ_ERR_ARG:
    case FFQ_ERR_PARAM_ROWS: TORCH_CHECK_VALUE(false, message);
    case FFQ_ERR_DTYPE: TORCH_CHECK_NOT_IMPLEMENTED(false, message);
    default: TORCH_CHECK(false, message);
  }
}

ffq_tiling tiling_of(const at::Tensor& data, at::IntArrayRef tile) {
  // check_tile_compatibility, quantization/tiled_tensor.py:24-29
  TORCH_CHECK_VALUE(data.dim() == static_cast<int64_t>(tile.size()), "Input dimensionality must match tile_size dimensionality got ",
                    data.dim(), " and ", tile.size());
  TORCH_CHECK_NOT_IMPLEMENTED(data.dim() <= FFQ_MAX_DIMS, "tensors of rank > ", FFQ_MAX_DIMS, " are not supported");
  ffq_tiling t{};
  t.ndim = static_cast<int32_t>(data.dim());
  for (int64_t i = 0; i < data.dim(); ++i) {
    t.shape[i] = data.size(i);
    t.tile[i] = tile[i];
  }
  return t;
}

void same_device(const at::Tensor& data, const at::Tensor& other) {
  TORCH_CHECK(other.device() == data.device(), "Expected all tensors to be on the same device, but found at least two devices, ",
              data.device(), " and ", other.device(), "!");
}

at::Tensor flat(const at::Tensor& t) { return t.is_contiguous() ? t : t.reshape({-1}).contiguous(); }  // (pointer + numel() are what is read)

void* stream_on(const at::Tensor& data) { return c10::hip::getCurrentHIPStream(data.device().index()).stream(); }

// A1 — fastforward::quantize_by_tile (_quantizer_impl.py:144-169)
at::Tensor quantize_by_tile(const at::Tensor& data, const at::Tensor& scale, at::IntArrayRef tile_size, double num_bits,
                            std::optional<at::ScalarType> output_dtype, const std::optional<at::Tensor>& offset) {
  const bool has_offset = offset.has_value() && offset->defined();
  same_device(data, scale);
  if (has_offset) same_device(data, *offset);
  c10::DeviceGuard guard(data.device());
  at::Tensor data_c = data.contiguous(), scale_c = flat(scale), offset_c = has_offset ? flat(*offset) : at::Tensor();
  ffq_tiling tiling = tiling_of(data_c, tile_size);
  const int data_dt = tag_of(data_c.scalar_type()), scale_dt = tag_of(scale_c.scalar_type());
  const int offset_dt = has_offset ? tag_of(offset_c.scalar_type()) : 0;
  at::ScalarType out_type;
  if (output_dtype.has_value()) {
    out_type = *output_dtype;
  } else {
    // `output_dtype or result.dtype`: the dtype the eager chain ends in (:164)
    int div = ffq_promote_types(data_dt, scale_dt);
    if (div != FFQ_F32 && div != FFQ_BF16 && div != FFQ_F16 && div != FFQ_F64) div = FFQ_F32;
    out_type = dtype_of(ffq_promote_types(div, has_offset ? offset_dt : scale_dt));
  }
  const int out_dt = tag_of(out_type);
  at::Tensor out = at::empty(data_c.sizes(), data_c.options().dtype(out_type));
  check(ffq_quantize_by_tile(data_c.data_ptr(), data_dt, scale_c.data_ptr(), scale_dt, scale_c.numel(),
                             has_offset ? offset_c.data_ptr() : nullptr, offset_dt, has_offset ? offset_c.numel() : 0, &tiling,
                             num_bits, out.data_ptr(), out_dt, stream_on(data_c)));
  return out;
}

// A2 — fastforward::dequantize_by_tile (_quantizer_impl.py:172-190)
at::Tensor dequantize_by_tile(const at::Tensor& data, const at::Tensor& scale, at::IntArrayRef tile_size,
                              const std::optional<at::Tensor>& offset, std::optional<at::ScalarType> output_dtype) {
  const bool has_offset = offset.has_value() && offset->defined();
  same_device(data, scale);
  if (has_offset) same_device(data, *offset);
  c10::DeviceGuard guard(data.device());
  at::Tensor data_c = data.contiguous(), scale_c = flat(scale), offset_c = has_offset ? flat(*offset) : at::Tensor();
  ffq_tiling tiling = tiling_of(data_c, tile_size);
  const int data_dt = tag_of(data_c.scalar_type()), scale_dt = tag_of(scale_c.scalar_type());
  const int offset_dt = has_offset ? tag_of(offset_c.scalar_type()) : 0;
  const at::ScalarType out_type =
      output_dtype.has_value() ? *output_dtype : dtype_of(ffq_dequantize_result_dtype(data_dt, scale_dt, offset_dt, has_offset ? 1 : 0));
  const int out_dt = tag_of(out_type);
  at::Tensor out = at::empty(data_c.sizes(), data_c.options().dtype(out_type));
  check(ffq_dequantize_by_tile(data_c.data_ptr(), data_dt, scale_c.data_ptr(), scale_dt, scale_c.numel(),
                               has_offset ? offset_c.data_ptr() : nullptr, offset_dt, has_offset ? offset_c.numel() : 0, &tiling,
                               out.data_ptr(), out_dt, stream_on(data_c)));
  return out;
}


at::Tensor workspace(size_t nbytes, const at::Tensor& like) {
  return nbytes ? at::empty({static_cast<int64_t>(nbytes)}, like.options().dtype(at::kByte)) : at::Tensor();
}
void* ptr(const at::Tensor& t) { return t.defined() ? t.data_ptr() : nullptr; }
at::Tensor opt(const std::optional<at::Tensor>& t) { return t.has_value() && t->defined() ? *t : at::Tensor(); }
// fp32 parameters as the C ABI reads them: a dense run of floats (only the pointer and numel() are used, so a contiguous fp32
// tensor of any shape passes as it is: the common case costs no operator call)
at::Tensor f32_flat(const at::Tensor& t) {
  if (!t.defined() || (t.scalar_type() == at::kFloat && t.is_contiguous())) return t;
  return t.reshape({-1}).to(at::kFloat).contiguous();
}

// The ticket words of the one-launch reductions and the split-K exchanges (include/ffq.h: zero before the first launch, left zero
// by every launch): one buffer per (kind, device, stream) for eager launches, which are serialised on that stream; a launch
// that is being captured into a hipGraph gets words of its own (allocated from the graph's pool, zeroed by a node of the capture).
at::Tensor tickets(int64_t count, const at::Tensor& like, void* stream, int kind) {
  if (count <= 0) return at::Tensor();
  if (c10::hip::currentStreamCaptureStatusMayInitCtx() != c10::hip::CaptureStatus::None)
    return at::zeros({count}, like.options().dtype(at::kInt));
  static std::mutex guard;
  static std::map<std::tuple<int, int, void*>, at::Tensor> cache;
  std::lock_guard<std::mutex> lock(guard);
  at::Tensor& have = cache[std::make_tuple(kind, static_cast<int>(like.device().index()), stream)];
  if (!have.defined() || have.numel() < count) have = at::zeros({std::max<int64_t>(count, 4096)}, like.options().dtype(at::kInt));
  return have;
}

int64_t num_tiles(const ffq_tiling& t) {
  const int64_t n = ffq_num_tiles(&t);
  if (n < 0) check(static_cast<int>(-n));
  return n;
}

// A3 — fastforward::quantize_dynamic_by_tile (_quantizer_impl.py:243-285)
std::tuple<at::Tensor, at::Tensor, at::Tensor> quantize_dynamic_by_tile(const at::Tensor& data, at::IntArrayRef tile_size, double num_bits,
                                                                        bool symmetric, bool allow_one_sided,
                                                                        std::optional<at::ScalarType> output_dtype) {
  c10::DeviceGuard guard(data.device());
  at::Tensor data_c = data.contiguous();
  ffq_tiling tiling = tiling_of(data_c, tile_size);
  const int64_t ntiles = num_tiles(tiling);
  const at::ScalarType out_type = output_dtype.has_value() ? *output_dtype
                                  : (data_c.scalar_type() == at::kFloat || data_c.scalar_type() == at::kDouble) ? data_c.scalar_type() : at::kFloat;
  const int data_dt = tag_of(data_c.scalar_type()), out_dt = tag_of(out_type);
  at::Tensor out = at::empty(data_c.sizes(), data_c.options().dtype(out_type));
  at::Tensor scale = at::empty({ntiles}, data_c.options().dtype(at::kFloat)), offset = at::empty({ntiles}, data_c.options().dtype(at::kFloat));
  const size_t nbytes = ffq_quantize_dynamic_workspace_bytes(&tiling, data_dt);
  at::Tensor ws = workspace(nbytes, data_c);
  void* stream = stream_on(data_c);
  // per-tensor: A5 in the reduction's last block; symmetric with the one-sided fallback: the two words of the guess / settle pair
  at::Tensor ticket = (ntiles == 1 || (symmetric && allow_one_sided)) ? tickets(2, data_c, stream, /*minmax*/ 1) : at::Tensor();
  check(ffq_quantize_dynamic_by_tile(data_c.data_ptr(), data_dt, &tiling, num_bits, symmetric ? 1 : 0, allow_one_sided ? 1 : 0, out.data_ptr(), out_dt,
                                     static_cast<float*>(scale.data_ptr()), static_cast<float*>(offset.data_ptr()), ptr(ws), nbytes,
                                     static_cast<int32_t*>(ptr(ticket)), stream));
  return {out, scale, offset};
}

// tiles_to_rows / rows_to_tiles (quantization/tiled_tensor.py:71-144) as views + one copy: [grid..., tile...] -> [ntiles, tile numel]
at::Tensor tiles_to_rows(const at::Tensor& x, at::IntArrayRef tile) {
  const int64_t nd = x.dim();
  std::vector<int64_t> split, perm, grid_then_tile;
  int64_t ntiles = 1, tile_numel = 1;
  for (int64_t i = 0; i < nd; ++i) {
    split.push_back(x.size(i) / tile[i]);
    split.push_back(tile[i]);
    ntiles *= x.size(i) / tile[i];
    tile_numel *= tile[i];
  }
  for (int64_t i = 0; i < nd; ++i) perm.push_back(2 * i);
  for (int64_t i = 0; i < nd; ++i) perm.push_back(2 * i + 1);
  return x.reshape(split).permute(perm).reshape({ntiles, tile_numel});
}
at::Tensor rows_to_tiles(const at::Tensor& rows, at::IntArrayRef shape, at::IntArrayRef tile) {
  const int64_t nd = static_cast<int64_t>(shape.size());
  std::vector<int64_t> grid_tile, perm(2 * nd);
  for (int64_t i = 0; i < nd; ++i) grid_tile.push_back(shape[i] / tile[i]);
  for (int64_t i = 0; i < nd; ++i) grid_tile.push_back(tile[i]);
  for (int64_t i = 0; i < nd; ++i) { perm[2 * i] = i; perm[2 * i + 1] = nd + i; }
  return rows.reshape(grid_tile).permute(perm).reshape(shape);
}

// A8 — fastforward::quantize_by_tile_backward (_quantizer_impl.py:193-237): the HIP kernel where it applies (one dtype for data and
// gradient, fp32 parameters, per-tensor / contiguous-run / by-tile tilings), else the same formulas op for op on the device
std::vector<at::Tensor> quantize_by_tile_backward(const at::Tensor& data, const at::Tensor& output_grad, const at::Tensor& scale,
                                                  at::IntArrayRef tile_size, double num_bits, const std::optional<at::Tensor>& offset) {
  const bool has_offset = offset.has_value() && offset->defined();
  same_device(data, output_grad);
  same_device(data, scale);
  if (has_offset) same_device(data, *offset);
  c10::DeviceGuard guard(data.device());
  const at::ScalarType dt = data.scalar_type();
  const bool fast = dt == output_grad.scalar_type() && (dt == at::kFloat || dt == at::kBFloat16 || dt == at::kHalf) && scale.scalar_type() == at::kFloat &&
                    (!has_offset || offset->scalar_type() == at::kFloat) && data.sizes() == output_grad.sizes();
  if (fast) {
    at::Tensor data_c = data.contiguous(), grad_c = output_grad.contiguous();
    at::Tensor scale_c = flat(scale), offset_c = has_offset ? flat(*offset) : at::Tensor();
    ffq_tiling tiling = tiling_of(data_c, tile_size);
    const int64_t ntiles = num_tiles(tiling);
    at::Tensor dinput = at::empty_like(data_c), dscale = at::empty({ntiles}, scale_c.options());
    at::Tensor doffset = has_offset ? at::empty({ntiles}, scale_c.options()) : at::Tensor();
    const size_t nbytes = ffq_quantize_backward_workspace_bytes(&tiling);
    at::Tensor ws = workspace(nbytes, data_c);
    const int status = ffq_quantize_by_tile_backward(data_c.data_ptr(), grad_c.data_ptr(), tag_of(dt), static_cast<const float*>(scale_c.data_ptr()),
                                                     scale_c.numel(), static_cast<const float*>(ptr(offset_c)), has_offset ? offset_c.numel() : 0, &tiling,
                                                     num_bits, dinput.data_ptr(), static_cast<float*>(dscale.data_ptr()),
                                                     static_cast<float*>(ptr(doffset)), ptr(ws), nbytes, stream_on(data_c));
    if (status == FFQ_OK) return {dinput, dscale.reshape(scale.sizes()), has_offset ? doffset.reshape(scale.sizes()) : at::empty({0}, scale.options())};
    if (status != FFQ_ERR_DTYPE) check(status);  // FFQ_ERR_DTYPE: a tiling the kernel does not cover
  }
  // the composite (reference :203-237 op for op): strided channels with half-precision parameters and the like
  tiling_of(data, tile_size);
  const at::Tensor s = scale.reshape({-1});
  const at::Tensor o = has_offset ? at::round(offset->reshape({-1})) : at::zeros_like(s);
  const double lo = -std::pow(2.0, num_bits - 1.0), hi = -lo - 1.0;
  const at::Tensor rows = tiles_to_rows(data, tile_size), grows = tiles_to_rows(output_grad, tile_size);
  const at::Tensor u = rows / s.unsqueeze(1) - o.unsqueeze(1);
  const at::Tensor q = at::round(u);
  const at::Tensor below = q < lo, above = q > hi;
  const at::Tensor clipped = at::logical_or(below, above);
  at::Tensor dinput = rows_to_tiles(at::where(clipped, at::zeros_like(grows), grows), data.sizes(), tile_size);
  at::Tensor doffset = has_offset ? at::where(clipped, s.unsqueeze(1) * grows, at::zeros_like(s.unsqueeze(1) * grows)).sum(1).reshape(scale.sizes())
                                  : at::empty({0}, scale.options());
  const at::Tensor bound = at::where(below, at::full({1}, lo, s.options()), at::full({1}, hi, s.options())) + o.unsqueeze(1).to(s.scalar_type());
  const at::Tensor dscale = at::where(clipped, bound, (q - u).to(s.scalar_type())) * grows;
  return {dinput, dscale.sum(1).reshape(scale.sizes()), doffset};
}

// One RunningMinMaxEstimator.estimate_step without leaving the device (range_setting/minmax.py:215-239 + the range setter
// nn/linear_quantizer.py:350-357): ffq_running_minmax_step on the estimator's running extrema and the quantizer's parameters
void running_minmax_step(const at::Tensor& data, at::IntArrayRef tile_size, at::Tensor running_min, at::Tensor running_max,
                         const std::optional<at::Tensor>& status_flags, double num_bits, bool symmetric, bool allow_one_sided, at::Tensor scale_out,
                         const std::optional<at::Tensor>& offset_out) {
  c10::DeviceGuard guard(data.device());
  at::Tensor data_c = data.contiguous(), flags = opt(status_flags), offset = opt(offset_out);
  for (const at::Tensor* t : {&running_min, &running_max, &scale_out}) same_device(data_c, *t);
  if (flags.defined()) same_device(data_c, flags);
  if (offset.defined()) same_device(data_c, offset);
  ffq_tiling tiling = tiling_of(data_c, tile_size);
  const int64_t ntiles = num_tiles(tiling);
  TORCH_CHECK(running_min.numel() == ntiles && running_max.numel() == ntiles && running_min.scalar_type() == data_c.scalar_type() &&
                  running_max.scalar_type() == data_c.scalar_type() && running_min.is_contiguous() && running_max.is_contiguous(),
              "running min/max must hold ", ntiles, " contiguous values of dtype ", c10::toString(data_c.scalar_type()));
  TORCH_CHECK(scale_out.numel() == ntiles && scale_out.is_contiguous() && (!offset.defined() || (offset.numel() == ntiles && offset.is_contiguous())),
              "scale / offset must hold ", ntiles, " contiguous values");
  const int data_dt = tag_of(data_c.scalar_type());
  const size_t nbytes = ffq_minmax_workspace_bytes(&tiling, data_dt);
  at::Tensor ws = workspace(nbytes, data_c);
  void* stream = stream_on(data_c);
  at::Tensor ticket = ntiles == 1 ? tickets(1, data_c, stream, /*minmax*/ 1) : at::Tensor();
  check(ffq_running_minmax_step(data_c.data_ptr(), data_dt, &tiling, running_min.data_ptr(), running_max.data_ptr(), static_cast<int32_t*>(ptr(flags)),
                                num_bits, symmetric ? 1 : 0, allow_one_sided ? 1 : 0, scale_out.data_ptr(), tag_of(scale_out.scalar_type()), ptr(offset),
                                offset.defined() ? tag_of(offset.scalar_type()) : 0, ptr(ws), nbytes, static_cast<int32_t*>(ptr(ticket)), stream));
}

// A6 — the quantized linear behind the dispatcher (_gen/fallback.py:77-112): int8 codes in, real-valued (or re-quantized) output out
at::Tensor linear_w8a8(const at::Tensor& x_codes, const at::Tensor& w_codes, const at::Tensor& x_scale, const std::optional<at::Tensor>& x_offset,
                       const at::Tensor& w_scale, const std::optional<at::Tensor>& w_offset, const std::optional<at::Tensor>& bias,
                       at::ScalarType out_dtype, const std::optional<at::Tensor>& out_scale, const std::optional<at::Tensor>& out_offset,
                       double out_num_bits, const std::optional<at::Tensor>& w_rowsum, std::optional<at::ScalarType> requant_from) {
  TORCH_CHECK_TYPE(x_codes.scalar_type() == at::kChar && w_codes.scalar_type() == at::kChar, "linear_w8a8 expects int8 codes");
  c10::DeviceGuard guard(x_codes.device());
  at::Tensor xc = x_codes.contiguous(), wc = w_codes.contiguous();
  const int64_t K = xc.dim() ? xc.size(-1) : 0, N = wc.dim() ? wc.size(0) : 0, M = K ? xc.numel() / K : 0;
  TORCH_CHECK(wc.dim() == 2 && wc.size(1) == K, "mat1 and mat2 shapes cannot be multiplied (", M, "x", K, " and ", wc.sizes(), "^T)");
  at::Tensor xs = f32_flat(x_scale), xo = f32_flat(opt(x_offset)), ws_ = f32_flat(w_scale), wo = f32_flat(opt(w_offset));
  at::Tensor os_ = f32_flat(opt(out_scale)), oo = f32_flat(opt(out_offset)), bias_c = opt(bias), rowsum = opt(w_rowsum);
  if (bias_c.defined()) bias_c = bias_c.contiguous();
  for (const at::Tensor* t : {&wc, &xs, &xo, &ws_, &wo, &os_, &oo, &bias_c, &rowsum})
    if (t->defined()) same_device(xc, *t);
  const int x_per_row = xs.numel() != 1, w_per_row = ws_.numel() != 1;
  TORCH_CHECK(!x_per_row || xs.numel() == M, "activation scale must have 1 or ", M, " entries, got ", xs.numel());
  TORCH_CHECK(!w_per_row || ws_.numel() == N, "weight scale must have 1 or ", N, " entries, got ", ws_.numel());
  TORCH_CHECK(!rowsum.defined() || (rowsum.scalar_type() == at::kInt && rowsum.numel() == N && rowsum.is_contiguous()),
              "w_rowsum must be a contiguous int32 tensor with ", N, " entries on the codes' device");
  std::vector<int64_t> shape(xc.sizes().begin(), xc.sizes().end());
  if (!shape.empty()) shape.back() = N;
  at::Tensor out = at::empty(shape, xc.options().dtype(out_dtype));
  // (include/ffq.h: no workspace without weight offsets when the weight row sums come with the call, are not needed, or the shape is
  // below the persistent kernel's class — an eager linear of a small model is one allocation less)
  const bool needs_ws = wo.defined() || (xo.defined() && !rowsum.defined() && ffq_linear_w8a8_takes_earlier(M, N, K));
  const size_t nbytes = needs_ws ? ffq_linear_w8a8_workspace_bytes(M, N, K) : 0;
  at::Tensor ws = workspace(nbytes, xc);
  const int y_dt = os_.defined() ? tag_of(requant_from.value_or(at::kBFloat16)) : 0;
  check(ffq_linear_w8a8(static_cast<const int8_t*>(xc.data_ptr()), static_cast<const int8_t*>(wc.data_ptr()), static_cast<const int32_t*>(ptr(rowsum)),
                        static_cast<const float*>(xs.data_ptr()), static_cast<const float*>(ptr(xo)), x_per_row, static_cast<const float*>(ws_.data_ptr()),
                        static_cast<const float*>(ptr(wo)), w_per_row, ptr(bias_c), bias_c.defined() ? tag_of(bias_c.scalar_type()) : 0, out.data_ptr(),
                        tag_of(out_dtype), static_cast<const float*>(ptr(os_)), static_cast<const float*>(ptr(oo)), out_num_bits, y_dt, M, N, K, ptr(ws), nbytes,
                        stream_on(xc)));
  return out;
}

// bmm on int8 codes in one launch (_gen/fallback.py:699-798): [B, M, K] x [B, N, K]^T with one parameter pair per operand
at::Tensor bmm_w8a8(const at::Tensor& x_codes, const at::Tensor& w_codes, const at::Tensor& x_scale, const std::optional<at::Tensor>& x_offset,
                    const at::Tensor& w_scale, const std::optional<at::Tensor>& w_offset, at::ScalarType out_dtype,
                    const std::optional<at::Tensor>& out_scale, const std::optional<at::Tensor>& out_offset, double out_num_bits,
                    std::optional<at::ScalarType> requant_from) {
  TORCH_CHECK_TYPE(x_codes.scalar_type() == at::kChar && w_codes.scalar_type() == at::kChar && x_codes.dim() == 3 && w_codes.dim() == 3,
                   "bmm_w8a8 expects int8 codes of shape [B, M, K] and [B, N, K]");
  c10::DeviceGuard guard(x_codes.device());
  at::Tensor xc = x_codes.contiguous(), wc = w_codes.contiguous();
  const int64_t B = xc.size(0), M = xc.size(1), K = xc.size(2), N = wc.size(1);
  TORCH_CHECK(wc.size(0) == B && wc.size(2) == K, "batch1 and batch2 shapes cannot be multiplied (", xc.sizes(), " and ", wc.sizes(), "^T)");
  at::Tensor xs = f32_flat(x_scale), xo = f32_flat(opt(x_offset)), ws_ = f32_flat(w_scale), wo = f32_flat(opt(w_offset));
  at::Tensor os_ = f32_flat(opt(out_scale)), oo = f32_flat(opt(out_offset));
  TORCH_CHECK(xs.numel() == 1 && ws_.numel() == 1, "bmm_w8a8 takes per-tensor parameters (one scale per operand)");
  for (const at::Tensor* t : {&wc, &xs, &xo, &ws_, &wo, &os_, &oo})
    if (t->defined()) same_device(xc, *t);
  at::Tensor out = at::empty({B, M, N}, xc.options().dtype(out_dtype));
  const size_t nbytes = wo.defined() ? ffq_bmm_w8a8_workspace_bytes(B, M, N, K) : 0;  // (the activation row sums of a launch with weight offsets)
  at::Tensor ws = workspace(nbytes, xc);
  const int y_dt = os_.defined() ? tag_of(requant_from.value_or(at::kBFloat16)) : 0;
  check(ffq_bmm_w8a8(static_cast<const int8_t*>(xc.data_ptr()), static_cast<const int8_t*>(wc.data_ptr()), static_cast<const float*>(xs.data_ptr()),
                     static_cast<const float*>(ptr(xo)), static_cast<const float*>(ws_.data_ptr()), static_cast<const float*>(ptr(wo)), out.data_ptr(),
                     tag_of(out_dtype), static_cast<const float*>(ptr(os_)), static_cast<const float*>(ptr(oo)), out_num_bits, y_dt, B, M, N, K, ptr(ws), nbytes,
                     stream_on(xc)));
  return out;
}

// A6, weight-only (_gen/fallback.py:86-112: quantized weight, plain input): x [..., K] bf16 x integer codes / packed nibbles.
// `covered` tensors only (ffq_linear_wq_supported: the Python wrapper asks first and otherwise lets the caller dequantize).
at::Tensor linear_wq(const at::Tensor& x, const at::Tensor& w_codes, const at::Tensor& w_scale, const std::optional<at::Tensor>& w_offset, int64_t group,
                     const std::optional<at::Tensor>& bias, at::ScalarType out_dtype, int64_t pack_block, int64_t two_pass, int64_t split) {
  c10::DeviceGuard guard(x.device());
  at::Tensor xc = x.contiguous(), wc = w_codes.contiguous();
  const int64_t K = xc.dim() ? xc.size(-1) : 0;
  TORCH_CHECK(K > 0, "linear_wq: empty contraction");
  const int64_t N = pack_block > 0 ? wc.numel() * 2 / K : wc.size(0), M = xc.numel() / K;
  TORCH_CHECK(ffq_linear_wq_supported(tag_of(xc.scalar_type()), tag_of(wc.scalar_type()), tag_of(out_dtype), M, N, K, group, pack_block),
              "linear_wq: the weight-code GEMM does not cover this problem (ask ffq_linear_wq_supported first)");
  at::Tensor sc = f32_flat(w_scale), of = f32_flat(opt(w_offset)), bias_c = opt(bias);
  TORCH_CHECK(!of.defined() || of.numel() == sc.numel(), "scale has ", sc.numel(), " entries, offset ", of.numel());
  if (bias_c.defined()) bias_c = bias_c.contiguous();
  for (const at::Tensor* t : {&wc, &sc, &of, &bias_c})
    if (t->defined()) same_device(xc, *t);
  std::vector<int64_t> shape(xc.sizes().begin(), xc.sizes().end());
  shape.back() = N;
  at::Tensor out = at::empty(shape, xc.options().dtype(out_dtype));
  void* stream = stream_on(xc);
  // (workspace bytes, ticket words) of the launch: the split-K slabs of the plan (or of a forced split) at the front, the bf16 image
  // of the two-pass form behind them (two_pass: -1 = the library's rule, 0 = never, 1 = offer the scratch whatever M)
  const int64_t n_tickets = ffq_linear_wq_tickets(M, N, K, 0), plan = ffq_linear_wq_split(M, N, K, 0);
  const int64_t use = std::max<int64_t>(1, split > 0 ? split : plan);
  const size_t slabs = ffq_linear_wq_slab_bytes(M, N, K, 0, use);
  size_t image = 0;
  if (two_pass > 0) image = static_cast<size_t>(N) * K * 2;
  else if (two_pass < 0) image = ffq_linear_wq_workspace_bytes(M, N, K) - ffq_linear_wq_slab_bytes(M, N, K, 0, plan);
  const size_t nbytes = slabs + image;
  at::Tensor ws = workspace(nbytes, xc);
  // (tickets whenever slabs are offered: where the preferred form declines the weight's storage, the form that takes over has its own plan)
  at::Tensor tk = (use > 1 || slabs > 0) && n_tickets > 0 ? tickets(n_tickets, xc, stream, /*wq*/ 0) : at::Tensor();
  check(ffq_linear_wq(xc.data_ptr(), tag_of(xc.scalar_type()), wc.data_ptr(), tag_of(wc.scalar_type()), pack_block, static_cast<const float*>(sc.data_ptr()),
                      static_cast<const float*>(ptr(of)), sc.numel(), group, ptr(bias_c), bias_c.defined() ? tag_of(bias_c.scalar_type()) : 0, out.data_ptr(),
                      tag_of(out_dtype), M, N, K, ptr(ws), nbytes, static_cast<int32_t*>(ptr(tk)), split, stream));
  return out;
}

}  // namespace

// PyTorch-ROCm names the HIP dispatch key "CUDA".
TORCH_LIBRARY_IMPL(fastforward_amd, CUDA, m) {
  m.impl("quantize_by_tile", &quantize_by_tile);
  m.impl("dequantize_by_tile", &dequantize_by_tile);
  m.impl("quantize_dynamic_by_tile", &quantize_dynamic_by_tile);
  m.impl("quantize_by_tile_backward", &quantize_by_tile_backward);
  m.impl("running_minmax_step", &running_minmax_step);
  m.impl("linear_w8a8", &linear_w8a8);
  m.impl("bmm_w8a8", &bmm_w8a8);
  m.impl("linear_wq", &linear_wq);
}

// The same four device kernels under the REFERENCE's operator names (fastforward::*, defined by its own
// torch.library.custom_op calls, quantization/_quantizer_impl.py:127-134; identical schemas): fastforward_amd.adapter.install()
// calls this once, after which an unmodified FastForward program's operators run dispatcher -> C++ -> C ABI on HIP tensors.
// Opt-in on purpose: importing this package never changes what the reference's operators do. Returns the number attached.
extern "C" int ffq_torch_install_reference_kernels() {
  static torch::Library* reference = nullptr;
  if (reference) return 0;
  g_reference_errors = true;
  reference = new torch::Library(torch::Library::IMPL, "fastforward", std::make_optional(c10::DispatchKey::CUDA), __FILE__, __LINE__);
  reference->impl("quantize_by_tile", &quantize_by_tile);
  reference->impl("dequantize_by_tile", &dequantize_by_tile);
  reference->impl("quantize_dynamic_by_tile", &quantize_dynamic_by_tile);
  reference->impl("quantize_by_tile_backward", &quantize_by_tile_backward);
  return 4;
}

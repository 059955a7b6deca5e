// libffq_torch.so — the two static operators of the reference's registry (fastforward::quantize_by_tile and
// fastforward::dequantize_by_tile; reference ops/_quantizer_impl.py:144-190) registered for the HIP dispatch key in C++,
// straight on top of the C ABI of libffq_hip.so (include/ffq.h). The dynamic operator and the backward stay in Python
// (fastforward_amd/ops.py): the first raises a Python exception type (QuantizationError, :259-264) this layer cannot
// construct, the second has a tensor-op composite for the tilings its kernel does not cover.
//
// The schemas are defined in fastforward_amd/ops.py (torch.library "fastforward_amd"); this file only adds the device
// kernels, so `torch.ops.fastforward_amd.*` on a HIP tensor goes dispatcher -> this file -> ffq_* without entering the Python
// interpreter or ctypes. Host tensors never reach these functions (the key is the device key); there is no CPU path here.
// PyTorch is plumbing: output allocation, the current stream and the device guard. Everything else is the C ABI.
//
// Host C++ only (g++): no kernels in this file.
#include <ATen/ATen.h>
#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include <optional>
#include <string>

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

// The exception the reference raises for each status (the comments beside ffq_status in include/ffq.h; the same table as
// fastforward_amd/_cabi.py; neither operator of this file returns FFQ_ERR_EMPTY).
void check(int status) {
  if (status == FFQ_OK) return;
  const char* text = ffq_last_error();
  std::string message = (text && *text) ? std::string(text) : "ffq status " + std::to_string(status);
  switch (status) {
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

at::Tensor flat(const at::Tensor& t) { return t.reshape({-1}).contiguous(); }

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

}  // namespace

// PyTorch-ROCm names the HIP dispatch key "CUDA".
TORCH_LIBRARY_IMPL(fastforward_amd, CUDA, m) {
  m.impl("quantize_by_tile", &quantize_by_tile);
  m.impl("dequantize_by_tile", &dequantize_by_tile);
}

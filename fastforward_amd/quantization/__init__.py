from fastforward_amd.quantization import affine as affine
from fastforward_amd.quantization import granularity as granularity
from fastforward_amd.quantization import tiled_tensor as tiled_tensor
from fastforward_amd.quantization.function import QuantizationContext as QuantizationContext
from fastforward_amd.quantization.function import QuantizationFunction as QuantizationFunction
from fastforward_amd.quantization.function import QuantizationParameters as QuantizationParameters
from fastforward_amd.quantization.fuse import find_weight_quantizers as find_weight_quantizers  # noqa: E402
from fastforward_amd.quantization.fuse import fuse_qdq_weights as fuse_qdq_weights  # noqa: E402
from fastforward_amd.quantization.freeze import freeze_parameters as freeze_parameters  # noqa: E402

"""fastforward_amd — MI355X-native backend for FastForward's affine fake-quantization hot path.

The package mirrors the part of ``fastforward``'s Python surface that the hot path runs through
(same names, argument meaning and error behaviour) and routes all arithmetic to
hand-written HIP kernels for gfx950 behind the C ABI of ``include/ffq.h``. There is no CPU
implementation: operators raise ``BackendError`` for host tensors or when ``libffq_hip.so`` is missing.
"""

from fastforward_amd import dispatcher as dispatcher
from fastforward_amd import exceptions as exceptions
from fastforward_amd import flags as flags
from fastforward_amd.flags import export_mode as export_mode
from fastforward_amd.flags import get_export_mode as get_export_mode
from fastforward_amd.flags import get_strict_quantization as get_strict_quantization
from fastforward_amd.flags import set_export_mode as set_export_mode
from fastforward_amd.flags import set_strict_quantization as set_strict_quantization
from fastforward_amd.flags import strict_quantization as strict_quantization
from fastforward_amd.quantized_tensor import QuantizedTensor as QuantizedTensor

from fastforward_amd import ops as ops  # isort: skip
from fastforward_amd import quantization as quantization  # isort: skip
from fastforward_amd.quantization.granularity import PerBlock as PerBlock  # isort: skip
from fastforward_amd.quantization.granularity import PerChannel as PerChannel  # isort: skip
from fastforward_amd.quantization.granularity import PerTensor as PerTensor  # isort: skip
from fastforward_amd.quantization.granularity import PerTile as PerTile  # isort: skip
import fastforward_amd.quantization._linear_quantized_ops  # noqa: E402,F401  isort: skip
from fastforward_amd import nn as nn  # isort: skip
from fastforward_amd import range_setting as range_setting  # isort: skip
from fastforward_amd import overrides as overrides  # isort: skip
from fastforward_amd.nn.quantized_module import quantize_model as quantize_model  # isort: skip
from fastforward_amd.overrides import disable_quantization as disable_quantization  # isort: skip
from fastforward_amd.overrides import enable_quantization as enable_quantization  # isort: skip
from fastforward_amd.range_setting import estimate_ranges as estimate_ranges  # isort: skip
import fastforward_amd.fused_linear  # noqa: E402,F401  isort: skip

__version__ = "0.1.0"

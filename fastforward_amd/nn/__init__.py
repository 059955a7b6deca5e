from fastforward_amd.nn.quantizer import Quantizer as Quantizer
from fastforward_amd.nn.quantizer import QuantizerMetadata as QuantizerMetadata
from fastforward_amd.nn.quantizer import QuantizerStub as QuantizerStub
from fastforward_amd.nn.quantizer import Tag as Tag
from fastforward_amd.nn.quantizer import default_tags as default_tags

from fastforward_amd.nn.dynamic_linear_quantizer import DynamicLinearQuantizer as DynamicLinearQuantizer  # isort: skip
from fastforward_amd.nn.linear_quantizer import LinearQuantizer as LinearQuantizer  # isort: skip
from fastforward_amd.nn.quantized_module import QuantizedModule as QuantizedModule  # isort: skip
from fastforward_amd.nn.quantized_module import SKIP_QUANTIZATION as SKIP_QUANTIZATION  # isort: skip
from fastforward_amd.nn.quantized_module import named_quantizers as named_quantizers  # isort: skip
from fastforward_amd.nn.quantized_module import quantize_model as quantize_model  # isort: skip
from fastforward_amd.nn.quantized_module import quantized_module_map as quantized_module_map  # isort: skip
from fastforward_amd.nn.quantized_module import surrogate_quantized_modules as surrogate_quantized_modules  # isort: skip
from fastforward_amd.nn import functional as functional  # isort: skip
from fastforward_amd.nn.linear import QuantizedLinear as QuantizedLinear  # isort: skip
from fastforward_amd.nn.container import QuantizedModuleDict as QuantizedModuleDict  # isort: skip
from fastforward_amd.nn.container import QuantizedModuleList as QuantizedModuleList  # isort: skip
from fastforward_amd.nn.container import QuantizedParameterDict as QuantizedParameterDict  # isort: skip
from fastforward_amd.nn.container import QuantizedParameterList as QuantizedParameterList  # isort: skip
from fastforward_amd.nn.container import QuantizedSequential as QuantizedSequential  # isort: skip

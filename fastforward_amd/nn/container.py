"""Containers convert to themselves: only their children matter (reference: src/fastforward/nn/container.py)."""

import torch

from fastforward_amd.nn.quantized_module import QuantizedModule


class QuantizedSequential(QuantizedModule, torch.nn.Sequential):
    pass


class QuantizedModuleList(QuantizedModule, torch.nn.ModuleList):
    pass


class QuantizedModuleDict(QuantizedModule, torch.nn.ModuleDict):
    pass


class QuantizedParameterList(QuantizedModule, torch.nn.ParameterList):
    pass


class QuantizedParameterDict(QuantizedModule, torch.nn.ParameterDict):
    pass

"""Quantized ``torch.nn.Linear`` (reference: src/fastforward/nn/linear.py:12-39).

Four quantizer slots with the reference's names and tags — ``input_quantizer`` (activation/input),
``weight_quantizer`` (parameter/weight), ``bias_quantizer`` (parameter/bias), ``output_quantizer``
(activation/output) — and the reference's forward: quantize input, RE-quantize the weight on every
call (:34), functional ``linear`` with the output quantizer.
"""

from __future__ import annotations

import torch

from fastforward_amd.nn.functional import linear
from fastforward_amd.nn.quantized_module import QuantizedModule
from fastforward_amd.nn.quantizer import Quantizer, QuantizerStub


class QuantizedLinear(QuantizedModule, torch.nn.Linear):
    weight_quantizer: Quantizer
    bias_quantizer: Quantizer | None
    input_quantizer: Quantizer
    output_quantizer: Quantizer

    def __init_quantization__(self) -> None:
        super().__init_quantization__()
        self.input_quantizer = QuantizerStub(input_quantizer=True)
        self.weight_quantizer = QuantizerStub(weight_quantizer=True, shape=self.weight.shape)
        if self.bias is not None:
            self.bias_quantizer = QuantizerStub(bias_quantizer=True, shape=self.bias.shape)
        else:
            self.register_quantizer("bias_quantizer", None)
        self.output_quantizer = QuantizerStub(output_quantizer=True)

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        input = self.input_quantizer(input)
        weight = self.weight_quantizer(self.weight)
        bias = self.bias
        if bias is not None and self.bias_quantizer is not None:
            bias = self.bias_quantizer(bias)
        return linear(input, weight, bias, output_quantizer=self.output_quantizer)

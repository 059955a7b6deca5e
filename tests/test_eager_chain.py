"""oracle/eager_chain.py (the CPU baseline bench.py times) reproduces the reference's fixtures."""

import sys

import torch

from conftest import ROOT, golden
from helpers import case_input, same_with_nan

sys.path.insert(0, str(ROOT / "oracle"))
import eager_chain  # noqa: E402


def test_eager_chain_matches_golden_sweeps():
    n = 0
    for c in golden("g3_sweeps.pt"):
        if "sweep3d" not in c["name"]:
            continue
        x = case_input(c)
        q = eager_chain.quantize(x, c["scale"], c["tile"], c["num_bits"], None, c["offset"])
        assert same_with_nan(q.to(torch.int8), c["codes"]), c["name"]
        d = eager_chain.dequantize(q, c["scale"], c["tile"], c["offset"], x.dtype)
        assert same_with_nan(d, c["dequantized"]), c["name"]
        n += 1
    assert n > 100


def test_eager_chain_ranges_and_linear():
    for c in golden("g4_ranges.pt"):
        s, o = eager_chain.parameters_for_range(c["min"], c["max"], c["num_bits"], c["symmetric"], c["allow_one_sided"])
        assert same_with_nan(s, c["scale"]) and (o is None) == (c["offset"] is None)
        if o is not None:
            assert same_with_nan(o, c["offset"])
    for c in golden("g6_linear.pt"):
        y = eager_chain.linear_w8a8(c["x"], c["weight"], c["x_scale"], c["x_offset"], c["w_scale"], c["w_offset"], 8, c["bias"])
        assert torch.equal(y, c["y"])  # same ops on the same machine: identical

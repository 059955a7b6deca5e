"""Pins the C oracle to the reference: every fixture produced by the reference must be reproduced.

CPU only. The fixtures come from tests/golden/gen_golden.py (the reference imported and run in the
build container) and include the known-answer vectors of the reference's own test-suite.
"""

import pytest

import parity_cases


@pytest.fixture(autouse=True)
def _backend(oracle_backend):
    yield


def test_known_answer_vectors():
    parity_cases.check_known_answers("cpu")


def test_ties_clamps_nan_inf_negative_zero():
    parity_cases.check_edges("cpu")


def test_random_sweeps_all_granularities():
    parity_cases.check_sweeps("cpu")


def test_mixed_dtype_sweep():
    parity_cases.check_dtype_sweep("cpu")


def test_parameters_for_range():
    parity_cases.check_ranges("cpu")


@pytest.mark.parametrize("sync_free", [False, True])
def test_running_minmax_trajectories(sync_free):
    parity_cases.check_running_minmax("cpu", sync_free=sync_free)


def test_int4_codes_and_q4_0_nibble_order():
    parity_cases.check_int4("cpu")


def test_quantized_linear_w8a8():
    parity_cases.check_linear("cpu")


def test_fused_producers_rmsnorm_silu_rope():
    parity_cases.check_producers("cpu")


def test_mm_matmul_bmm_through_the_dispatcher():
    parity_cases.check_matmul_family("cpu")


def test_weight_codes_with_row_sums():
    parity_cases.check_rowsum_fusion("cpu")


def test_attention_chain_and_fused_output_quantizer():
    parity_cases.check_attention("cpu", exact_chain=True)


def test_quantize_by_tile_backward():
    parity_cases.check_backward("cpu")


def test_mse_grid_range_estimator():
    parity_cases.check_mse_grid("cpu")


def test_gguf_block_writers():
    parity_cases.check_gguf_blocks("cpu")


def test_gptq_matches_the_reference_bit_for_bit():
    parity_cases.check_gptq("cpu", exact=True)


def test_weight_only_linear_matches_the_reference():
    parity_cases.check_weight_only_linear("cpu")


def test_smoothed_minmax_trajectories():
    parity_cases.check_smoothed_minmax("cpu")


def test_large_linear_fixture():
    parity_cases.check_linear_large("cpu")

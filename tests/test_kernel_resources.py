"""What hipcc actually emitted for the persistent GEMM kernels (CPU tests: no device needed).

* `tools/kernel_resources.py` reads the AMDGPU metadata of the code objects inside the SHIPPED library: no instantiation of the
  int8 GEMM may spill a vector register (round 3 shipped 35-87 spilled VGPRs in every WOFF / REQUANT form), nor may the bf16-image
  form of the weight-only GEMM.
* `tools/asm_cluster_check.py` compiles the two GEMM sources to assembly and checks the K-loops themselves: no scratch access
  between the first and the last MFMA of any persistent kernel (the code-converting forms of the weight-only GEMM keep a few
  spills in their cold prologue / split-K epilogue, never in the loop).
"""

import pathlib
import sys

import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))

import asm_cluster_check  # noqa: E402
import kernel_resources  # noqa: E402


def _kernels(needle: str):
    if kernel_resources.readelf() is None or not kernel_resources.DEFAULT_LIBRARY.exists():
        pytest.skip("llvm-readelf or the built library is missing")
    rows = [k for k in kernel_resources.kernel_resources() if needle in str(k["name"])]
    assert rows, f"no {needle} kernel in {kernel_resources.DEFAULT_LIBRARY}"
    return rows


def test_the_int8_gemm_spills_nothing():
    rows = _kernels("w8a8_gemm256fq_kernel")
    assert len(rows) >= 13  # 4 containers x REQUANT x WOFF (- the float / WOFF pair) + the MLP mode
    spilled = {str(k["name"]): k["vgpr_spill_count"] for k in rows if k["vgpr_spill_count"] or k["private_segment_fixed_size"]}
    assert not spilled, spilled
    assert all(k["vgpr_count"] <= 256 and k["agpr_count"] == 0 for k in rows)  # two waves per SIMD: 256 registers each


def test_the_bf16_image_form_of_the_weight_only_gemm_spills_nothing():
    rows = [k for k in _kernels("wq_gemm256_kernel") if "wq_gemm256_kernelILi0E" in str(k["name"])]
    assert len(rows) == 3  # bf16 / fp32 output, MLP mode
    assert all(k["vgpr_spill_count"] == 0 and k["private_segment_fixed_size"] == 0 for k in rows), rows


@pytest.mark.parametrize("source,needle", [("csrc/ffq_linear.hip", "w8a8_gemm256fq_kernel"), ("csrc/ffq_wlinear.hip", "wq_gemm256_kernel")])
def test_no_scratch_access_inside_a_k_loop(source, needle):
    if not pathlib.Path("/opt/rocm/bin/hipcc").exists():
        pytest.skip("hipcc is missing")
    text = asm_cluster_check.build_assembly(ROOT / "fastforward_amd" / source)
    seen = 0
    for name, clusters, scratch in asm_cluster_check.clusters_of(text, needle):
        seen += 1
        assert scratch == 0, f"{name}: {scratch} scratch instructions between the first and the last MFMA"
        assert clusters, name
    assert seen >= 10


def test_the_one_wave_per_simd_k_loop_is_what_was_written():
    """wq_gemm4w_kernel's K-loop is inline assembly in source order; what hipcc may add is scalar bookkeeping and address moves.
    Its inner loop (two super-steps) must hold exactly 256 MFMAs, 64 fragment reads and 32 LDS-DMA pieces (buffer_load ... lds), four barriers
    (round 6: the A operand half a step ahead of B, a barrier per k-half) behind two counted `vmcnt(8)` and two lgkm-only waits, no scratch
    access, no accumulator traffic between AGPRs and VGPRs, no other vector instruction except the handful that switch the image
    descriptors to the next tile (the tile's last pair of super-steps is the same loop body), and the kernel must own all 256 AGPRs
    (the accumulators pinned there) without a spill."""
    import re

    if not pathlib.Path("/opt/rocm/bin/hipcc").exists():
        pytest.skip("hipcc is missing")
    text = asm_cluster_check.build_assembly(ROOT / "fastforward_amd" / "csrc/ffq_wlinear.hip")
    lines = text.split("\n")
    starts = [i for i, l in enumerate(lines) if re.match(r"^_ZN3ffq16wq_gemm4w_kernel\w*:", l)]
    assert len(starts) == 2  # plain, and gate + up + SiLU*up (round 6)
    for start in starts:
        _check_one_wave_per_simd_loop(lines, start)
    rows = [k for k in _kernels("wq_gemm4w_kernel")]
    assert len(rows) == 2 and all(k["agpr_count"] == 256 and k["vgpr_spill_count"] == 0 and k["private_segment_fixed_size"] == 0 for k in rows), rows


def _check_one_wave_per_simd_loop(lines, start):
    import re

    end = next(i for i in range(start, len(lines)) if ".end_amdhsa_kernel" in lines[i])
    # the K-loop = the innermost loop around the MFMAs: its header is the last loop-header label ahead of the first MFMA, its end the
    # first branch behind the last one
    mfma = [i for i in range(start, end) if "v_mfma" in lines[i]]
    header = next(j for j in range(mfma[0], start, -1) if re.match(r"^\.LBB\d+_\d+:", lines[j]) and "Loop Header" in " ".join(lines[j:j + 3]))
    back = next(i for i in range(mfma[-1], end) if "s_cbranch" in lines[i])
    ops = [l.split()[0] for l in (x.strip() for x in lines[header:back]) if l and not l.startswith((";", ".", "//"))]
    count = lambda prefix: sum(1 for o in ops if o.startswith(prefix))  # noqa: E731
    assert count("v_mfma_f32_16x16x32_bf16") == 256 and count("ds_read_b128") == 64 and count("buffer_load_dwordx4") == 32
    assert count("scratch_") == 0 and count("v_accvgpr") == 0 and count("s_barrier") == 4
    body = "\n".join(lines[header:back])
    assert len(re.findall(r"s_waitcnt vmcnt\(8\) lgkmcnt\(0\)", body)) == 2 and "vmcnt(0)" not in body
    assert count("v_") - count("v_mfma") <= 4  # buffer addressing: the piece and the super-step are in the scalar offset


def test_the_128_column_tile_kernels_keep_everything_in_registers_at_two_blocks_per_cu():
    """wq_mid_kernel (csrc/ffq_wmid.hip): 16 instantiations, none with scratch memory (round 6 found hipcc keeping the by-value argument
    struct and the register rings in scratch for the nibble forms until the stream lambdas were force-inlined), all within the 256
    registers that two blocks per CU leave a wave."""
    rows = _kernels("wq_mid_kernel")
    assert len(rows) == 16
    bad = {str(k["name"]): (k["vgpr_count"], k["vgpr_spill_count"], k["private_segment_fixed_size"]) for k in rows
           if k["vgpr_spill_count"] or k["private_segment_fixed_size"] or k["vgpr_count"] + k["agpr_count"] > 256}
    assert not bad, bad


def test_the_lds_dma_form_of_the_128_column_tiles_is_the_loop_that_was_written():
    """wq_mid_dma_kernel (csrc/ffq_wmid.hip): no spills, no scratch; its K-loop (int8 per-channel weights, BM = 128, 8 waves of 16 columns) holds
    ONE counted `s_waitcnt vmcnt(N)` with N = the wave's LDS-DMA requests of the ring's RING - 2 younger stages (1 x 3), one raw barrier, the
    step's 3 LDS-DMA requests (2 activation pieces + the wave's own 16 code rows), 2 x (1 + 8) fragment reads in inline assembly and 16 MFMAs
    — and no compiler-inserted `vmcnt(0)`."""
    import re

    rows = _kernels("wq_mid_dma_kernel")
    assert len(rows) == 16
    assert all(k["vgpr_spill_count"] == 0 and k["private_segment_fixed_size"] == 0 and k["vgpr_count"] <= 256 for k in rows), rows
    if not pathlib.Path("/opt/rocm/bin/hipcc").exists():
        pytest.skip("hipcc is missing")
    text = asm_cluster_check.build_assembly(ROOT / "fastforward_amd" / "csrc/ffq_wmid.hip")
    lines = text.split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN3ffq17wq_mid_dma_kernelILi1ELb0ELb0ELi128EEEvNS_7MidArgsE:", l))
    end = next(i for i in range(start, len(lines)) if ".end_amdhsa_kernel" in lines[i])
    header = next(i for i in range(start, end) if "Inner Loop Header" in lines[i])
    label = next(lines[j].split(":")[0] for j in range(header, start, -1) if re.match(r"^\.LBB\d+_\d+:", lines[j]))
    back = next(i for i in range(header, end) if re.search(r"s_cbranch_\w+ " + re.escape(label) + r"\b", lines[i]))
    body = [x.strip() for x in lines[header:back] if x.strip() and not x.strip().startswith((";", ".", "//"))]
    count = lambda prefix: sum(1 for o in body if o.startswith(prefix))  # noqa: E731
    assert count("v_mfma_f32_16x16x32_bf16") == 16 and count("global_load_lds_dwordx4") == 3 and count("s_barrier") == 1
    assert count("ds_read_b128") == 16 and count("ds_read_b64") == 2 and count("scratch_") == 0
    waits = [o for o in body if o.startswith("s_waitcnt") and "vmcnt" in o]
    assert waits == ["s_waitcnt vmcnt(3)"], waits

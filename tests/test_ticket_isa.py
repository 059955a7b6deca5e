"""The cross-block hand-overs rest on an instruction order, not on the HIP memory model (ffq_common.h: ACQ_REL tickets measured 2-5 x
slower, profiles/r06_ticket_order_ab.txt) — so the order is pinned here, on the ISA hipcc emits (CPU test: no device needed):

* every slab store / load of the ticketed exchanges is a buffer instruction with the `sc1` bit (write-through past the XCD's L2 /
  a load that bypasses the reading CU's L1);
* between a kernel's last slab store and its ticket read-modify-write stands `s_waitcnt vmcnt(0)` with no vector-memory store behind it
  (the partial sums have left the CU before anybody can observe the ticket);
* the ticket is an agent-scope returning atomic (`sc0`), i.e. performed at L2, where the peers' write-through data already is.
A compiler change that reorders or drops any of these fails this test instead of corrupting a split-K sum once in a million launches.
"""

import pathlib
import re
import sys

import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))

import asm_cluster_check  # noqa: E402

STORE = re.compile(r"^(buffer_store|global_store|flat_store)_")
TICKET = re.compile(r"^global_atomic_(add|or)(_x2)?\s")


_ASSEMBLY: dict[str, str] = {}  # one compile per source


def _kernels(source: str, needle: str):
    if not pathlib.Path("/opt/rocm/bin/hipcc").exists():
        pytest.skip("hipcc is missing")
    if source not in _ASSEMBLY:
        _ASSEMBLY[source] = asm_cluster_check.build_assembly(ROOT / "fastforward_amd" / "csrc" / source)
    text = _ASSEMBLY[source]
    found = 0
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)\n\s*\.end_amdhsa_kernel", text, re.S | re.M):
        if needle not in m.group(1):
            continue
        found += 1
        code = [l.strip() for l in m.group(2).split("\n")]
        yield m.group(1), [l for l in code if l and not l.startswith((";", ".", "//")) and not l.endswith(":")]
    assert found, f"no {needle} in {source}"


@pytest.mark.parametrize("source,needle,instantiations", [("ffq_wskinny.hip", "wq_skinny_kernel", 12), ("ffq_wskinny.hip", "wq_skinny_rows_kernel", 20),
                                                         ("ffq_wmid.hip", "wq_mid_kernel", 16), ("ffq_wmid.hip", "wq_mid_dma_kernel", 16)])
def test_slabs_are_write_through_and_drained_before_the_ticket(source, needle, instantiations):
    seen = 0
    for name, code in _kernels(source, needle):
        seen += 1
        tickets = [i for i, l in enumerate(code) if TICKET.match(l)]
        assert tickets, f"{name}: no ticket read-modify-write"
        for i in tickets:
            assert " sc0" in code[i], f"{name}: the ticket is not a returning (L2) atomic: {code[i]}"
            # walk back to the nearest vector-memory store: a vmcnt(0) wait must stand between it and the ticket
            j = next((k for k in range(i - 1, -1, -1) if STORE.match(code[k])), None)
            assert j is not None, f"{name}: no slab store ahead of the ticket"
            assert any(re.match(r"^s_waitcnt\b.*vmcnt\(0\)", l) for l in code[j + 1:i]), f"{name}: no s_waitcnt vmcnt(0) between {code[j]} and {code[i]}"
        first = tickets[0]
        # the exchange's data path: every 16-byte buffer store ahead of the first ticket and every 16-byte buffer load behind it is sc1
        slab_stores = [l for l in code[:first] if l.startswith("buffer_store_dwordx4")]
        slab_loads = [l for l in code[first:] if l.startswith("buffer_load_dwordx4")]
        assert slab_stores and slab_loads, name
        assert all(" sc1" in l for l in slab_stores), f"{name}: a slab store without sc1: {[l for l in slab_stores if ' sc1' not in l][:2]}"
        assert all(" sc1" in l for l in slab_loads), f"{name}: a slab load without sc1: {[l for l in slab_loads if ' sc1' not in l][:2]}"
    assert seen >= instantiations, (needle, seen)

"""gpurun_out/r05/int8_ceiling_raw.txt (tools/pmc_int8_ceiling.sh) -> profiles/r05_int8_ceiling.md.
usage: python tools/int8_ceiling_md.py RAW OUT.md"""
import re, sys, collections

raw, out = sys.argv[1], sys.argv[2]
rows = collections.defaultdict(dict)   # (label, kernel) -> {counter: value, "us": [...]}
label = None
tail = []
for line in open(raw):
    line = line.rstrip("\n")
    if line.startswith("== "):
        label = line[3:].split(" | ")[0]
    elif line.startswith("  ") and ": launches" in line:
        kernel, rest = line.strip().split(": launches", 1)
        key = (label, kernel)
        us = float(re.search(r"avg_us ([\d.]+)", rest).group(1))
        rows[key].setdefault("us", []).append(us)
        for c, v in re.findall(r"\| (\w+) (\d+)", rest):
            rows[key][c] = float(v)
    elif "TOP/s" in line:
        tail.append(line)
SIMDS, XCDS = 1024, 8


def derived(r, ops):
    if "GRBM_GUI_ACTIVE" not in r:
        return None
    # the counters' pass (first list entry is the pass that carried GRBM_GUI_ACTIVE: passes are listed in the script's order)
    us = r["us"][0]
    cyc = r["GRBM_GUI_ACTIVE"] / XCDS
    clock = cyc / us / 1e3            # GHz
    busy = r["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * SIMDS)
    return us, clock, busy, busy * clock, ops / us / 1e6 if ops else None


md = ["# r05 — where the int8 GEMM stands against what the matrix pipe sustains on toggling operands", "",
      "`tools/pmc_int8_ceiling.sh` on one MI355X: separate `rocprofv3 --kernel-trace --pmc` passes (never combined with other trace domains), 4 launches per pass, the first dropped. "
      "`SQ_VALU_MFMA_BUSY_CYCLES` counts the matrix pipe's work cycles (identical for equal MAC counts: 939,524,096 for both 1.92 PFLOP shapes, whoever runs them), "
      "`GRBM_GUI_ACTIVE` the GPU-active cycles summed over the 8 XCDs; **busy = MFMA cycles / (GUI_ACTIVE / 8 x 1024 SIMDs)**, **clock = GUI_ACTIVE / 8 / duration**. "
      "The chip is power-limited in all of these launches (2.4 GHz nominal), so busy x clock — matrix-pipe work per second — is the figure of merit; durations under counter collection are 5-15 % longer than unprofiled ones.", "",
      "| launch | duration us | clock GHz | MFMA busy | busy x clock | POP/s in this pass | vs toggling probe |", "|---|---:|---:|---:|---:|---:|---:|"]
probe_ref = None
for key, r in rows.items():
    if key[0] == "probe" and key[1].startswith("probe<1, 1>") and "GRBM_GUI_ACTIVE" in r:
        probe_ref = derived(r, None)[3]
shapes = {"16384 14336 4096": "gate/up shape [16384 x 14336 x 4096]", "16384 4096 14336": "down_proj shape [16384 x 4096 x 14336]"}
names = {"probe<0, 0>": "probe 32x32x32, constant operands", "probe<1, 0>": "probe 16x16x64, constant operands", "probe<0, 1>": "probe 32x32x32, random bytes (toggling)",
         "probe<1, 1>": "probe 16x16x64, random bytes (toggling) — the reference", "probe<0, 2>": "probe 32x32x32, random small codes", "probe<1, 2>": "probe 16x16x64, random small codes"}
for (label, kernel), r in rows.items():
    parts = label.split(" ", 1)
    ops = 2.0 * 16384 * 14336 * 4096 if parts[0] in ("ours", "vendor") else 16 * 2.0 * 16 * 16 * 64 * 40000 * 512 * 4 if kernel.startswith("probe<1") else 8 * 2.0 * 32 ** 3 * 40000 * 512 * 4
    d = derived(r, ops)
    if d is None:
        continue
    who = {"ours": "`w8a8_gemm256fq_kernel`", "vendor": "vendor `Cijk_..I8II..` (torch._int_mm)"}.get(parts[0])
    name = f"{who}, {shapes.get(parts[1], parts[1])}" if who else names.get(kernel.split("(")[0], kernel)
    md.append(f"| {name} | {d[0]:.1f} | {d[1]:.2f} | {d[2]:.3f} | {d[3]:.3f} | {d[4] / 1e3:.2f} | {d[3] / probe_ref:.2f} |")
md += ["", "Memory side of the same launches (per launch; `FETCH_SIZE` / `WRITE_SIZE` in KiB-units of the counter, x 1 / x 1 as listed; L1 -> L2 read latency = `TCP_TCC_READ_REQ_LATENCY_sum / TCP_TCC_READ_REQ_sum` in cycles):", "",
       "| launch | L2 requests | L2 hit rate | L1->L2 read requests | latency per read request (cycles) | FETCH_SIZE | WRITE_SIZE | wave cycles waiting for an instruction / wave cycles |", "|---|---:|---:|---:|---:|---:|---:|---:|"]
for (label, kernel), r in rows.items():
    parts = label.split(" ", 1)
    if parts[0] not in ("ours", "vendor") or "TCC_REQ_sum" not in r:
        continue
    who = {"ours": "`w8a8_gemm256fq_kernel`", "vendor": "vendor int8 GEMM"}[parts[0]]
    md.append(f"| {who}, {shapes[parts[1]]} | {r['TCC_REQ_sum'] / 1e6:.1f} M | {r['TCC_HIT_sum'] / max(1.0, r['TCC_REQ_sum']):.3f} | {r.get('TCP_TCC_READ_REQ_sum', 0) / 1e6:.1f} M | "
              f"{r.get('TCP_TCC_READ_REQ_LATENCY_sum', 0) / max(1.0, r.get('TCP_TCC_READ_REQ_sum', 1)):.0f} | {r.get('FETCH_SIZE', 0):.0f} | {r.get('WRITE_SIZE', 0):.0f} | "
              f"{r.get('SQ_WAIT_INST_ANY', 0) / max(1.0, r.get('SQ_WAVE_CYCLES', 1)):.3f} |")
md += ["", "Reading. (1) The shipped kernel does 0.70-0.71 of the matrix-pipe work per second that back-to-back MFMAs on toggling operands do (unprofiled: 2.75 against 3.38 POP/s = 0.81 "
       "on down_proj) — NOT the 0.90 that would call it a ceiling; the vendor's hand-scheduled assembly stands at 0.62-0.70 on the same box and operands. "
       "(2) Both are power-limited, and differently: the vendor's loop keeps the pipe busier (0.77 against 0.71 on down_proj) at a LOWER clock (1.47 against 1.62 GHz) for the same product; "
       "the probe, which moves no data at all, reaches 0.88 busy at 1.84 GHz. What separates a GEMM from the probe is therefore the energy of the operand traffic per MAC (LDS fragment reads, "
       "the LDS-DMA stream, L2 / fabric), not issue slots: raising busy lowers the clock. (3) Memory side: the same L1->L2 request count as the vendor's kernel and the same hit rate (0.77-0.79), but "
       "1.3-1.4 x its latency per request (335-361 against 243-283 cycles: eight waves issue their DMA pieces in two bursts per super-step where the vendor's four waves spread them), and waves "
       "wait for an instruction 39-42 % of their cycles against 55 % (two waves per SIMD against one). (4) Tried on top of this in round 5, A/B of two builds on one box: column groups of the tile walk "
       "for down_proj (the XCDs of a grid row streaming the same activation panels, worth +5 % on the bf16 kernel): -4.5 % (0.732 against 0.699 ms); row sums from the weight-quantization launches "
       "instead of a rowsum_i8 launch before each GEMM: the gate+up launch 11.6 % SLOWER (1584 against 1420 us, `r05_rowsum_ab.md`) — the reduction launch is what brings the weight codes into the Infinity "
       "Cache, and a GEMM whose weight panels miss it waits on HBM latency: operand residency in the 256 MiB cache is worth more than any schedule change measured so far.",
       "", "The probe's own timing (no profiler):", "", "```"] + tail + ["```", ""]
open(out, "w").write("\n".join(md))
print("\n".join(md))

"""Where a kernel's scratch (spill) instructions sit relative to its MFMA loop, from a hipcc -S device assembly file.
usage: python tools/asm_scratch_map.py file.s [name-filter]"""
import re, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)\n\s*\.end_amdhsa_kernel', txt, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if flt not in name:
        continue
    lines = body.split('\n')
    scr = [i for i, l in enumerate(lines) if re.search(r'\bscratch_(load|store)', l)]
    mf = [i for i, l in enumerate(lines) if 'v_mfma' in l]
    if not mf:
        continue
    # the K-loop: the densest MFMA region = between the first and last MFMA
    inloop = [i for i in scr if mf[0] <= i <= mf[-1]]
    rl = [i for i, l in enumerate(lines) if re.search(r'v_(readlane|writelane)', l) and mf[0] <= i <= mf[-1]]
    print(f"{name[:70]:70s} scratch ops {len(scr):4d}, inside the MFMA span {len(inloop):4d}, lane moves inside {len(rl):4d} (span {mf[0]}..{mf[-1]} of {len(lines)} lines, {len(mf)} MFMAs)")

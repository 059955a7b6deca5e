"""Per-kernel register / scratch usage from a hipcc --save-temps assembly file (the .amdhsa metadata at its end).
usage: python tools/kernel_resources.py file.s [name-filter]"""
import re, subprocess, sys

text = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for block in re.findall(r"- \.agpr_count:.*?(?=\n  - \.agpr_count:|\namdhsa\.target|\Z)", text, re.S):
    g = lambda k: re.search(rf"\.{k}:\s+(\S+)", block)
    name = g("name").group(1)
    if flt not in name:
        continue
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = dem.replace("ffq::", "").replace("(ffq::LinearArgs, int)", "").replace("void ", "")
    print(f"{dem[:90]:90s} vgpr {g('vgpr_count').group(1):>4s} agpr {g('agpr_count').group(1):>4s} sgpr {g('sgpr_count').group(1):>4s} scratch {g('private_segment_fixed_size').group(1):>5s} lds {g('group_segment_fixed_size').group(1)}")

import re,sys
from collections import Counter
L=open(sys.argv[1] if len(sys.argv)>1 else "_tmp/gemm4w_probe-hip-amdgcn-amd-amdhsa-gfx950.s").read().split("\n")
hdr=[i for i,l in enumerate(L) if "Inner Loop Header" in l or "Loop Header" in l]
for h in hdr:
    lab=L[h].split(":")[0].strip()
    end=next(i for i in range(h,len(L)) if re.search(r"s_cbranch_\w+ "+re.escape(lab)+r"\b",L[i]))
    seg=[l.strip() for l in L[h+1:end]]
    seg=[l for l in seg if l and not l.startswith((";",".","//"))]
    c=Counter(x.split()[0] for x in seg)
    print(lab,len(seg),"instructions:",dict(c.most_common(14)))
    def cat(x):
        op=x.split()[0]
        if op.startswith("v_mfma"): return "M"
        if op.startswith("ds_read") or op.startswith("ds_load"): return "r"
        if op.startswith("global_load"): return "G"
        if op.startswith("scratch"): return "S"
        if op.startswith("v_accvgpr"): return "a"
        if op.startswith("s_waitcnt"): return "w"
        if op.startswith("s_barrier"): return "B"
        if op.startswith("v_"): return "v"
        if op.startswith("s_"): return "s"
        return "?"
    print("".join(cat(x) for x in seg))
for k in (".vgpr_count",".agpr_count",".vgpr_spill_count",".private_segment_fixed_size:"):
    print([l.strip() for l in L if k in l][:1])

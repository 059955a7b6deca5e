"""Mean of every counter per kernel from the csvs tools/pmc_gemm.sh wrote: python tools/pmc_table.py gpurun_out/pmc_gemm/TAG_*.csv"""
import collections, csv, sys
for path in sys.argv[1:]:
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(path)
    for k, v in agg.items():
        print(f"  {k:28s} mean {sum(v)/len(v):16.1f}  n={len(v)}")

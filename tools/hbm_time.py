"""The HBM-bound kernels of bench.py's `hbm_kernels` table on [14336, 4096] (FFQ_LIB selects a variant build: tools/build_variant.sh).
usage: python tools/hbm_time.py [substring-filter]"""
import os, pathlib, sys
import torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from fastforward_amd import _native
if os.environ.get("FFQ_LIB"):
    from fastforward_amd._cabi import FFQLibrary
    _native._LIB = FFQLibrary(os.environ["FFQ_LIB"])
import bench
flt = sys.argv[1] if len(sys.argv) > 1 else ""
for row in bench.hbm_kernels(torch.device("cuda", 0)):
    if flt in row["op"]:
        print(f"{row['op'][:70]:70s} {row['ms'] * 1e3:8.2f} us  {row['achieved']:8.1f} GB/s  frac {row['frac']:.4f}", flush=True)

"""bench.py on another build of the library (same-box A/B): FFQ_LIB=tools/_exp/libffq_x.so python tools/bench_with_lib.py [bench args]"""
import os, pathlib, runpy, sys
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from fastforward_amd import _native
from fastforward_amd._cabi import FFQLibrary
_native._LIB = FFQLibrary(os.environ["FFQ_LIB"])
sys.argv = [str(ROOT / "bench.py")] + sys.argv[1:]
runpy.run_path(str(ROOT / "bench.py"), run_name="__main__")

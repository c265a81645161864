"""Host-side cost of the row-sharded step: cProfile of bench.py restricted to the step's own call tree.
usage (GPU box): DCTR_FORCE_SHARDED=1 python tools/host_profile.py [steps]"""
import cProfile
import os
import pstats
import runpy
import sys

steps = sys.argv[1] if len(sys.argv) > 1 else "300"
sys.argv = ["bench.py", "--steps", steps, "--warmup", "20", "--no-cpu-baseline"]
here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path(os.path.join(here, "bench.py"), run_name="__main__")
finally:
    pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(r"distributed\.py|distributed_c10d|capi\.py|streams\.py|ctypes|engine\.py", 40)
st.sort_stats("tottime").print_stats(25)

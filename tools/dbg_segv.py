import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import faulthandler; faulthandler.enable()
import test_gpu_tracking as t
print(t.run_streams(640, 480, 8, 0, 6, t.SEEDS, "resync"), flush=True)

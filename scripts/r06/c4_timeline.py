"""C4 rig (512 envs, FEM on a side stream) for a kernel timeline: 24 warm-up + 21 timed steps.  Run under
`rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_c4 -- python3 scripts/r06/c4_timeline.py`; scripts/r06/timeline_of.py turns the trace into one
steady step, kernel by kernel."""
import sys
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch, bench
from tacex_amd.uipc.gelpad_scene import FemGelpad
dev = "cuda:0"
fem = FemGelpad(512, dev, max_newton_iter=64, side_stream=True)
rig = bench.Rig(512, 240, 320, 1, False, dev, 1, 0, fem=fem)
fem.ms_log = []
el = rig.timed(21, 24)
print(f"C4: {el / 21 * 1e3:.3f} ms per step")

"""C4 / rolling / C5 sweep entries only (side stream A/B): python scripts/c4_quick.py"""
import sys, json, types
sys.path.insert(0, '/root/repo')
import torch, bench
from tacex_amd.uipc.gelpad_scene import FemGelpad
dev = "cuda:0"
def one(label, B, H, W, fem, steps=42):
    rig = bench.Rig(B, H, W, 1, False, dev, 1, 0, fem=fem)
    fem.ms_log = []
    el = rig.timed(steps, 24)
    ms = fem.ms_log[-steps:]
    print(f"{label}: {el / steps * 1e3:.3f} ms per step = {B * steps / el / 1e3:.1f} K frames/s; fem mean {sum(ms) / len(ms):.3f} ms", flush=True)
    del rig
    torch.cuda.empty_cache()
for side in (True, False, True, False):
    one(f"C4 side_stream={side}", 512, 240, 320, FemGelpad(512, dev, max_newton_iter=64, side_stream=side))
one("rolling side", 512, 240, 320, FemGelpad(512, dev, max_newton_iter=64, motion="rolling", side_stream=True))
one("C5 side", 1024, 480, 640, FemGelpad(1024, dev, max_newton_iter=64, side_stream=True), steps=21)
one("C5 one stream", 1024, 480, 640, FemGelpad(1024, dev, max_newton_iter=64, side_stream=False), steps=21)

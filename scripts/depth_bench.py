"""Times the depth -> height map + contact rows / columns pass (frame_rows_kernel) alone: us per launch, TB/s of its 8 B/px."""
import sys
import torch
from tacex_amd import _lib

lib = _lib.load_library()
dev = torch.device("cuda:0")
for (B, H, W) in ((2048, 240, 320), (1024, 480, 640)):
    g = torch.Generator(device=dev); g.manual_seed(1)
    depth = 0.024 + 0.005 * torch.rand((B, H, W), device=dev, generator=g)
    hm = torch.empty_like(depth); fmin = torch.empty(B, device=dev); ind = torch.empty(B, device=dev)
    rows = torch.empty((B, 4), dtype=torch.int32, device=dev)
    st = _lib.current_stream_handle(dev)
    def run():
        rc = lib.tacex_height_map_from_depth(_lib.ptr(depth), 0.024, 0.029, 0.0045, 0.024, _lib.ptr(hm), _lib.ptr(fmin), _lib.ptr(ind), 0,
                                             _lib.ptr(rows), B, H, W, st)
        _lib.check(rc, "tacex_height_map_from_depth")
    for _ in range(5): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = []
    for rep in range(5):
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / 20 * 1e3)
    us = sorted(best)[len(best) // 2]
    print(f"depth pass {B}x{H}x{W}: {us:.1f} us per launch (min {min(best):.1f}), {B * H * W * 8 / us / 1e6:.2f} TB/s", flush=True)

#!/usr/bin/env python3
"""Golden vectors for the FOTS marker image (container only; reads /root/reference at run time).

`FOTSMarkerSimulator.draw_markers` (fots_marker_sim.py:346-384) is plain NumPy + math once the patch table exists; the
module itself imports cv2 / IsaacLab and cannot be imported.  The method's source is extracted from the file where it lies
with `ast` and executed with a stub `self` holding a SEEDED SYNTHETIC patch table (oracle.fots_oracle.synthetic_patch_table -
the reference draws its table with cv2, which this image lacks; the table is an input of the stamping, not part of it).
Inputs and the reference's output images are stored in tests/golden/fots_marker_image.npz.
"""
import ast
import math
import sys
import textwrap
import types
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
sys.path.insert(0, str(REPO))
FS = Path("/root/reference/source/tacex/tacex/simulation_approaches/fots/fots_marker_sim.py")


def marker_sets(rs, W, H):
    """(N, M, 2) marker positions: the GelSight Mini grid displaced smoothly, plus the awkward cases."""
    xi = np.linspace(15, W - 15, 11, dtype=int)
    yi = np.linspace(26, H - 26, 9, dtype=int)
    gx, gy = np.meshgrid(xi, yi)
    base = np.stack((gx.reshape(-1), gy.reshape(-1)), -1).astype(np.float32)
    sets = [base.copy()]
    a = base + rs.uniform(-6, 6, base.shape).astype(np.float32)            # every sub-pixel phase
    sets.append(a)
    b = base.copy()
    b[10:20] = b[0:10] + rs.uniform(-3, 3, (10, 2)).astype(np.float32)     # overlapping stamps: order matters
    b[30] = (-3.2, 5.7); b[31] = (W + 2.4, 40.1); b[32] = (100.3, -7.9)    # inside the 12-pixel margin
    b[33] = (-20.0, 50.0); b[34] = (W + 12.0, 10.0); b[35] = (50.0, H + 40.0)  # outside the canvas: skipped
    b[36] = (0.0, 0.0); b[37] = (W - 1.0, H - 1.0); b[38] = (-6.5, -6.5); b[39] = (W + 5.49, H + 5.49)
    sets.append(b)
    sets.append(base * 0 + np.array([W / 2, H / 2], np.float32) + rs.uniform(-8, 8, base.shape).astype(np.float32))  # a heap
    return np.stack(sets).astype(np.float32)


def main():
    from oracle.fots_oracle import synthetic_patch_table

    txt = FS.read_text()
    src = None
    for node in ast.walk(ast.parse(txt)):
        if isinstance(node, ast.FunctionDef) and node.name == "draw_markers":
            src = ast.get_source_segment(txt, node)
    assert src, "draw_markers not found in the reference"
    ns = {"np": np, "math": math}
    exec(textwrap.dedent(src), ns)
    stub = types.SimpleNamespace(patch_array_dict=synthetic_patch_table(seed=5))
    rs = np.random.RandomState(17)
    out = {}
    for (W, H) in ((320, 240), (640, 480)):
        uv = marker_sets(rs, W, H)
        if (W, H) == (640, 480):
            uv = uv[1:3]
        imgs = np.stack([ns["draw_markers"](stub, uv[k].astype(np.float32), 3, W, H) for k in range(len(uv))])
        imgs4 = np.stack([ns["draw_markers"](stub, uv[k].astype(np.float32), 4.2, W, H) for k in range(len(uv))])
        out[f"uv_{H}x{W}"] = uv
        out[f"img_{H}x{W}"] = imgs
        out[f"img_size42_{H}x{W}"] = imgs4
        assert imgs.dtype == np.uint8 and imgs.shape == (len(uv), H, W)
    np.savez_compressed(HERE / "fots_marker_image.npz", table_seed=5, **out)
    print((HERE / "fots_marker_image.npz").stat().st_size // 1024, "KiB")


if __name__ == "__main__":
    main()

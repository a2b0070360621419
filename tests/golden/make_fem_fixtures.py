#!/usr/bin/env python3
"""Re-save the reference's real Gmsh tet meshes (DATA files under
/root/reference/source/tacex_uipc/examples/libuipc-samples/tet_meshes/*.msh) as compact .npz fixtures.
Container only (reads /root/reference at run time)."""
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent.parent))
from oracle.fem_oracle import load_msh  # noqa: E402

SRC = Path("/root/reference/source/tacex_uipc/examples/libuipc-samples/tet_meshes")
out = {}
for name in ("tet", "cube", "simple_axle", "link", "cylinder_hole"):
    pts, tets = load_msh(SRC / f"{name}.msh")
    out[f"{name}_points"] = pts
    out[f"{name}_tets"] = tets
    print(name, pts.shape, tets.shape)
np.savez_compressed(HERE / "fem_meshes.npz", **out)

#!/usr/bin/env python3
"""Golden values for the FEM wrapper rows a18 / a20 (container only; reads /root/reference at run time).

The reference's tacex_uipc modules import libuipc / IsaacLab / omni and cannot be imported, but what they DEFINE on the path is
plain Python that `ast` can lift out of the files where they lie:

  * the default values of `UipcSimCfg` (uipc_sim.py:32-131, nested Newton / LinearSystem / LineSearch / Contact),
    `UipcObjectCfg` + `StableNeoHookeanCfg` (uipc_object.py:54-90) and `UipcIsaacAttachmentsCfg` (uipc_attachments.py:33-66):
    the class bodies are executed with a no-op `configclass` decorator and the annotated fields read back;
  * `UipcIsaacAttachments.compute_attachment_data` (uipc_attachments.py:247-346): the function body is executed with stand-ins
    for the three services it CALLS and that do not exist here - the PhysX sphere sweep (answered by an analytic box collider),
    the USD world transform of the rigid prim (a fixed pose) and IsaacLab's `quat_apply_inverse` (restated from IsaacLab's
    public formula) - so what is pinned is the reference's own control flow and arithmetic around them: which vertices are kept,
    in which order, `offset = quat_apply_inverse(q, v - obj_pos)` in float32, and the layout of what is returned.

Writes tests/golden/uipc_cfg.npz (numbers and strings as 0-d / 1-d arrays; no source text).

SECURITY: the lifted class bodies / function body are EXECUTED.  /root/reference is untrusted public content: `assert_reviewable` refuses
imports, dunder access and process / file / introspection builtins before anything runs; run this in the sandboxed container only."""
import ast
import sys
import textwrap
import types
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
sys.path.insert(0, str(REPO))
REF = Path("/root/reference/source/tacex_uipc/tacex_uipc")
US, UO, UA = REF / "sim/uipc_sim.py", REF / "objects/uipc_object.py", REF / "sim/uipc_attachments.py"


FORBIDDEN_NAMES = {"exec", "eval", "compile", "open", "__import__", "input", "breakpoint", "globals", "locals", "vars", "getattr", "setattr",
                   "delattr", "os", "sys", "subprocess", "shutil", "socket", "importlib", "ctypes", "builtins", "exit", "quit"}


def assert_reviewable(src: str, what: str):
    """The reference is untrusted public content and the lifted source is EXECUTED: refuse anything a cfg class body or
    the attachment arithmetic has no business doing - imports, dunder access, process / file / introspection builtins.  Run this
    script in the sandboxed build container only."""
    for node in ast.walk(ast.parse(src)):
        if isinstance(node, (ast.Import, ast.ImportFrom, ast.Global, ast.Nonlocal, ast.AsyncFunctionDef, ast.Await, ast.Yield, ast.YieldFrom)):
            raise RuntimeError(f"{what}: refusing to execute source containing {type(node).__name__} (line {getattr(node, 'lineno', '?')})")
        if isinstance(node, ast.Name) and node.id in FORBIDDEN_NAMES:
            raise RuntimeError(f"{what}: refusing to execute source that names `{node.id}` (line {node.lineno})")
        if isinstance(node, ast.Attribute) and node.attr.startswith("__"):
            raise RuntimeError(f"{what}: refusing to execute source with dunder attribute `{node.attr}` (line {node.lineno})")


def class_source(path: Path, name: str) -> str:
    src = path.read_text()
    for node in ast.walk(ast.parse(src)):
        if isinstance(node, ast.ClassDef) and node.name == name:
            seg = ast.get_source_segment(src, node)
            assert_reviewable(seg, f"{path.name}:{name}")
            return seg
    raise KeyError(name)


def function_source(path: Path, cls: str, name: str) -> str:
    src = path.read_text()
    for node in ast.walk(ast.parse(src)):
        if isinstance(node, ast.ClassDef) and node.name == cls:
            for f in node.body:
                if isinstance(f, ast.FunctionDef) and f.name == name:
                    seg = textwrap.dedent("\n".join(src.splitlines()[f.lineno - 1:f.end_lineno]))
                    assert_reviewable(seg, f"{path.name}:{cls}.{name}")
                    return seg
    raise KeyError(name)


def defaults(cls, prefix=""):
    """annotated class attributes (and those of nested cfg classes), flattened to 'Outer.Inner.field' -> value"""
    out = {}
    for k, v in vars(cls).items():
        if k.startswith("_"):
            continue
        if isinstance(v, type):
            out.update(defaults(v, prefix + k + "."))
        elif not callable(v) and not isinstance(v, (staticmethod, classmethod, property)):
            if hasattr(v, "__class__") and v.__class__.__module__ == "__ref__" and not isinstance(v, (int, float, str, bool, tuple)):
                continue  # an instance of a nested cfg (newton: Newton = Newton()): its fields come from the nested class
            out[prefix + k] = v
    return out


def main():
    ns = {"__name__": "__ref__", "configclass": lambda c: c, "pathlib": __import__("pathlib"), "AssetBaseCfg": object, "TetMeshCfg": object,
          "UipcIsaacAttachmentsCfg": object}
    exec("from __future__ import annotations\n" + class_source(US, "UipcSimCfg"), ns)
    exec("from __future__ import annotations\n" + class_source(UO, "UipcObjectCfg"), ns)
    exec("from __future__ import annotations\n" + class_source(UA, "UipcIsaacAttachmentsCfg"), ns)
    gold = {}
    for cname in ("UipcSimCfg", "UipcObjectCfg", "UipcIsaacAttachmentsCfg"):
        for k, v in defaults(ns[cname]).items():
            if k.endswith("workspace"):
                continue  # the current working directory of whoever imports the reference
            if v is None:
                v = "None"
            gold[f"cfg/{cname}.{k}"] = np.asarray(v)

    # ---- compute_attachment_data with stand-ins for PhysX / USD / IsaacLab ----
    from tacex_amd.uipc.uipc_object import gelpad_box_mesh

    P, T = gelpad_box_mesh(8, 10, 4)
    size = P.max(0) - P.min(0)
    # the sensor case: a box hugging the back face, rotated about z and shifted (so that the inverse rotation matters)
    half = np.array([size[0] / 2 + 1e-6, size[1] / 2 + 1e-6, 0.001])
    ang = 0.3
    quat = np.array([np.cos(ang / 2), 0.0, 0.0, np.sin(ang / 2)])  # w, x, y, z
    Rz = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]])
    centre_local = np.array([size[0] / 2, size[1] / 2, -0.001])
    shift = np.array([0.05, -0.02, 0.3])
    pts = (P - centre_local) @ Rz.T + shift  # the gelpad placed in the world with the case at `shift`, orientation `quat`
    sphere_radius, max_dist = 5e-4, 1e-5

    def box_sd(p_world):
        loc = Rz.T @ (np.asarray(p_world, np.float64) - shift)
        d = np.abs(loc) - half
        return np.linalg.norm(np.maximum(d, 0.0)) + min(d.max(), 0.0)

    class _Query:  # get_physx_scene_query_interface().sweep_sphere_closest(radius, origin, dir, distance, bothSides)
        def sweep_sphere_closest(self, radius, origin, dir, distance, bothSides):
            o = np.asarray(origin, np.float64)
            hit = min(box_sd(o), box_sd(o + distance * np.asarray(dir, np.float64))) <= radius
            return {"hit": bool(hit), "collision": "/World/case/collisions" if hit else ""}

    class _Prim:
        def GetPath(self): return "/World/case"

    class _Quat:
        def GetReal(self): return float(quat[0])
        def GetImaginary(self): return [float(quat[1]), float(quat[2]), float(quat[3])]

    class _Rot:
        def GetQuaternion(self): return _Quat()

    class _Pose:
        def ExtractTranslation(self): return [float(v) for v in shift]
        def ExtractRotation(self): return _Rot()

    def quat_apply_inverse(q, v):  # IsaacLab isaaclab.utils.math.quat_apply_inverse (public formula): v - w t + xyz x t, t = 2 xyz x v
        xyz = q[:, 1:]
        t = torch.linalg.cross(xyz, v, dim=-1) * 2
        return v - q[:, 0:1] * t + torch.linalg.cross(xyz, t, dim=-1)

    real_tensor = torch.tensor
    torch_cpu = types.SimpleNamespace(tensor=lambda a, device=None: real_tensor(a))  # the reference hard-codes device="cuda:0"
    fns = {
        "np": np, "torch": torch_cpu,
        "get_physx_interface": lambda: types.SimpleNamespace(force_load_physics_from_usd=lambda: None),
        "get_physx_scene_query_interface": lambda: _Query(),
        "sim_utils": types.SimpleNamespace(find_matching_prims=lambda path: [_Prim()]),
        "omni": types.SimpleNamespace(usd=types.SimpleNamespace(get_world_transform_matrix=lambda prim: _Pose())),
        "math_utils": types.SimpleNamespace(quat_apply_inverse=quat_apply_inverse),
        "print": lambda *a, **k: None,
    }
    src = function_source(UA, "UipcIsaacAttachments", "compute_attachment_data").replace("@staticmethod\n", "")
    exec(src, fns)
    offsets, idx, prims, positions, obj_pos = fns["compute_attachment_data"]("/World/case", pts, T, sphere_radius=sphere_radius, max_dist=max_dist)
    gold.update({
        "att/tet_points": pts, "att/box_half": half, "att/rigid_pos": shift, "att/rigid_quat": quat, "att/sphere_radius": np.asarray(sphere_radius),
        "att/max_dist": np.asarray(max_dist), "att/offsets": np.asarray(offsets, np.float32), "att/idx": np.asarray(idx, np.int64),
        "att/positions": np.asarray(positions, np.float64), "att/obj_pos": np.asarray(obj_pos, np.float64),
    })
    np.savez_compressed(HERE / "uipc_cfg.npz", **gold)
    print(f"wrote {HERE / 'uipc_cfg.npz'}: {len(gold)} entries, {len(idx)} attachment points")
    for k in sorted(gold):
        if k.startswith("cfg/"):
            print(" ", k, "=", gold[k])


if __name__ == "__main__":
    main()

"""ctypes binding of libtacex_hip.so (the C ABI declared in include/tacex_hip.h).

The product path has NO CPU fallback: if the library is missing, cannot be loaded or no gfx950 device is
visible, the functions here raise - loudly - instead of computing anything on the host.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

from ._build import LIB, build_library

MAX_LEVELS = 8
ABI_VERSION = 15  # TACEX_ABI_VERSION of include/tacex_hip.h; bumped whenever a signature or struct layout changes
FLAG_NO_SHIFT = 1
FLAG_HAVE_FRAME_MIN = 2
FLAG_WITH_SHADOW = 4
FLAG_OBS_U8 = 8
FLAG_HAVE_FRAME_ROWS = 16

c_float_p = C.POINTER(C.c_float)
c_int32_p = C.POINTER(C.c_int32)


class TaximParams(C.Structure):
    _fields_ = [
        ("height", C.c_int32),
        ("width", C.c_int32),
        ("calib_height", C.c_int32),
        ("calib_width", C.c_int32),
        ("pixmm", C.c_float),
        ("num_bins", C.c_int32),
        ("contact_scale", C.c_float),
        ("n_levels", C.c_int32),
        ("ksize_w", C.c_int32 * MAX_LEVELS),
        ("ksize_h", C.c_int32 * MAX_LEVELS),
        ("taps_w", c_float_p * MAX_LEVELS),
        ("taps_h", c_float_p * MAX_LEVELS),
        ("poly", c_float_p),
        ("gel_map", c_float_p),
        ("background", c_float_p),
        ("feat_x", c_float_p),
        ("feat_y", c_float_p),
    ]


class ShadowParams(C.Structure):
    _fields_ = [
        ("num_directions", C.c_int32), ("num_fan_rays", C.c_int32), ("num_heights", C.c_int32), ("num_steps", C.c_int32),
        ("fan_angles", c_float_p), ("fan_cos", c_float_p), ("fan_sin", c_float_p), ("table", c_float_p),
        ("win_left", C.c_int32), ("win_right", C.c_int32), ("win_top", C.c_int32), ("win_bottom", C.c_int32),
        ("shadow_depth_0", C.c_float), ("height_precision", C.c_float), ("discretize_precision", C.c_float),
        ("step_x", C.c_float), ("step_y", C.c_float),
        ("blur_kw", C.c_int32), ("blur_kh", C.c_int32), ("blur_taps_w", c_float_p), ("blur_taps_h", c_float_p),
    ]


class FotsParams(C.Structure):
    _fields_ = [
        ("height", C.c_int32),
        ("width", C.c_int32),
        ("num_markers_row", C.c_int32),
        ("num_markers_col", C.c_int32),
        ("marker_x", c_int32_p),
        ("marker_y", c_int32_p),
        ("lamb", C.c_double * 3),
        ("mm2pix", C.c_float),
        ("shear_max", C.c_float),
        ("theta_max_deg", C.c_float),
    ]


class FemParams(C.Structure):
    _fields_ = [
        ("num_verts", C.c_int32),
        ("num_tets", C.c_int32),
        ("rest_positions", C.POINTER(C.c_double)),
        ("tets", c_int32_p),
        ("youngs", C.c_double),
        ("poisson", C.c_double),
        ("density", C.c_double),
        ("dt", C.c_double),
        ("gravity", C.c_double * 3),
        ("constraint_strength_ratio", C.c_double),
    ]


# name -> (restype, argtypes); mirrors include/tacex_hip.h one to one
_vp, _i, _u, _f, _d, _sz = C.c_void_p, C.c_int, C.c_uint, C.c_float, C.c_double, C.c_size_t
SIGNATURES = {
    "tacex_last_error": (C.c_char_p, []),
    "tacex_abi_version": (_i, []),
    "tacex_device_count": (_i, [C.POINTER(C.c_int)]),
    "tacex_device_arch": (_i, [_i, C.c_char_p, _sz]),
    "tacex_taxim_create": (_i, [_i, C.POINTER(TaximParams), C.POINTER(_vp)]),
    "tacex_taxim_destroy": (None, [_vp]),
    "tacex_taxim_workspace_bytes": (_sz, [_vp, _i]),
    "tacex_taxim_set_shadow": (_i, [_vp, C.POINTER(ShadowParams)]),
    "tacex_taxim_shadow_workspace_bytes": (_sz, [_vp, _i]),
    "tacex_taxim_shadow_rays": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "tacex_height_map_from_depth": (_i, [_vp, _d, _d, _f, _f, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "tacex_taxim_defer_height_map_from_depth": (_i, [_vp, _vp, _d, _d, _f, _f, _vp, _vp, _vp, _vp, _vp, _i]),
    "tacex_taxim_flush_deferred": (_i, [_vp, _vp]),
    "tacex_indentation_depth": (_i, [_vp, _f, _f, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "tacex_taxim_set_frame_rows": (_i, [_vp, _vp, _i]),
    "tacex_depth_from_mesh": (_i, [_vp, _vp, _i, _i, _vp, _vp, _f, _f, _f, _f, _f, _f, _vp, _vp, _i, _i, _i, _vp]),
    "tacex_taxim_render": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _u, _vp]),
    "tacex_taxim_render_obs": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _u, _vp]),
    "tacex_taxim_deform": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _u, _vp]),
    "tacex_taxim_shade": (_i, [_vp, _vp, _vp, _vp, _i, _vp]),
    "tacex_resize_bilinear_aa": (_i, [_vp, _i, _i, _vp, _i, _i, _i, _vp]),
    "tacex_resize_bilinear_aa_nhwc": (_i, [_vp, _i, _i, _vp, _i, _i, _i, _i, _vp, _vp]),
    "tacex_taxim_set_fused_tail": (_i, [_vp, _i]),
    "tacex_taxim_chunk_frames": (_i, [_vp, _i]),
    "tacex_taxim_set_profiling": (_i, [_vp, _i]),
    "tacex_taxim_read_profile": (_i, [_vp, _i, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "tacex_taxim_num_stages": (_i, [_vp]),
    "tacex_taxim_stage_name": (C.c_char_p, [_vp, _i]),
    "tacex_fots_create": (_i, [_i, C.POINTER(FotsParams), C.POINTER(_vp)]),
    "tacex_fots_destroy": (None, [_vp]),
    "tacex_fots_state_bytes": (_sz, [_i]),
    "tacex_fots_workspace_bytes": (_sz, [_i]),
    "tacex_fots_markers": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "tacex_height_map_from_indenters": (_i, [_vp, _f, _f, _f, _f, _f, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "tacex_fots_markers_partials": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "tacex_taxim_fots_partials_per_env": (_i, [_vp]),
    "tacex_taxim_set_fots_partials": (_i, [_vp, _vp, _i]),
    "tacex_taxim_set_fots_taps": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _i]),
    "tacex_fots_markers_compact": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "tacex_fots_marker_image": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "tacex_fem_create": (_i, [_i, C.POINTER(FemParams), C.POINTER(_vp)]),
    "tacex_fem_destroy": (None, [_vp]),
    "tacex_fem_workspace_bytes": (_sz, [_vp, _i]),
    "tacex_fem_element_terms": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "tacex_fem_energy": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "tacex_fem_gradient": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "tacex_fem_newton_step": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _d, _i, _vp]),
    "tacex_fem_step": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, C.POINTER(C.c_double), _i, _d, _i, _d, _i, _vp]),
    "tacex_fem_set_coarse_space": (_i, [_vp, _i, _vp, _vp, _vp]),
    "tacex_fem_set_chains": (_i, [_vp, _i, _vp, _vp]),
    "tacex_fem_set_indenter_mesh": (_i, [_vp, _i, _vp, _i, _vp]),
    "tacex_fem_contact_gaps": (_i, [_vp, _vp, _vp, _i, _vp]),
    "tacex_fem_newton_resident": (_i, [_vp]),
    "tacex_fem_set_friction_lag": (_i, [_vp, _i]),
    "tacex_fem_set_affine_body": (_i, [_vp, _i, _vp, _i, _vp, _d, _d, _vp, _i, _vp, _d, _d, _d, _i, _i]),
    "tacex_fem_ball_workspace_bytes": (_sz, [_vp, _i]),
    "tacex_fem_set_edge_edge": (_i, [_vp, _i]),
    "tacex_fem_set_line_search_refine": (_i, [_vp, _i]),
    "tacex_fem_ball_moments": (_i, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "tacex_fem_ball_terms": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "tacex_fem_ball_step": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, C.POINTER(C.c_double), _i, _d, _d, _i, _d, _i, _vp]),
    "tacex_fem_reset_envs": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "tacex_fem_set_friction": (_i, [_vp, _d, _d]),
    "tacex_fem_set_contact_following": (_i, [_vp, _i]),
    "tacex_fem_set_deterministic": (_i, [_vp, _i]),
    "tacex_fem_set_contact": (_i, [_vp, _vp, _d, _d, _vp]),
    "tacex_fem_set_newton_early_exit": (_i, [_vp, _vp, C.c_double]),
    "tacex_fem_set_attachment_targets": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "tacex_fem_marker_uv": (_i, [_vp, _vp, _vp, _d, _d, _d, _d, _vp, _i, _i, _i, _vp]),
    "tacex_fem_marker_flow": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _d, _d, _d, _d, _vp, _vp, _d, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
}

_lib = None
MISSING_SYMBOLS: list = []


class TacexHipError(RuntimeError):
    pass


def load_library(build_if_missing: bool = True) -> C.CDLL:
    """dlopen libtacex_hip.so (building it with hipcc if the in-tree .so is absent) and bind every symbol."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB.exists() and not build_if_missing:
        raise TacexHipError(f"{LIB} is missing - run `python -m tacex_amd._build` (needs hipcc)")
    if build_if_missing:
        # a sha256 stamp compare of the sources when the library is current; rebuilds a missing OR STALE one (a library
        # older than include/tacex_hip.h would be called with the wrong signatures)
        try:
            build_library()
        except Exception as e:
            if not LIB.exists():
                raise TacexHipError(f"cannot build {LIB}: {e}") from e
            raise TacexHipError(f"{LIB} is stale (sources changed) and rebuilding it failed: {e}") from e
    try:
        lib = C.CDLL(str(LIB))
    except OSError as e:  # e.g. libamdhip64 missing
        raise TacexHipError(f"cannot load {LIB}: {e}") from e
    missing = []
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            missing.append(name)
            continue
        fn.restype = res
        fn.argtypes = args
    global MISSING_SYMBOLS
    MISSING_SYMBOLS = missing
    if missing:
        raise TacexHipError(f"{LIB} lacks symbols declared in include/tacex_hip.h: {missing}")
    if lib.tacex_abi_version() != ABI_VERSION:
        raise TacexHipError(f"libtacex_hip.so ABI version {lib.tacex_abi_version()} != {ABI_VERSION} (include/tacex_hip.h)")
    _lib = lib
    return lib


def last_error() -> str:
    return load_library().tacex_last_error().decode(errors="replace")


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = last_error()
        if rc == 2:
            raise ValueError(f"{what}: {msg}")
        raise TacexHipError(f"{what}: {msg}")


def require_gpu(device_index: int = 0) -> str:
    """Fail loudly unless a HIP device is visible; returns its gcn arch name."""
    lib = load_library()
    n = C.c_int(0)
    rc = lib.tacex_device_count(C.byref(n))
    if rc != 0 or n.value <= device_index:
        raise TacexHipError(
            f"tacex_amd needs an AMD GPU (HIP device {device_index}); visible devices: {n.value} "
            f"({last_error()}). There is no CPU fallback for the tactile hot path."
        )
    buf = C.create_string_buffer(128)
    check(lib.tacex_device_arch(device_index, buf, 128), "tacex_device_arch")
    return buf.value.decode()


def ptr(t) -> int:
    """Raw device pointer of a torch tensor (or 0 for None)."""
    return 0 if t is None else t.data_ptr()


def current_stream_handle(device=None) -> int:
    import torch

    return torch.cuda.current_stream(device).cuda_stream

"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/tacex_hip.h declares
(no compute call is made - there is no GPU here)."""
import re

import pytest

from conftest import REPO


def _declared_functions():
    txt = (REPO / "include" / "tacex_hip.h").read_text()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(tacex_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from tacex_amd import _lib

    lib = _lib.load_library()
    assert _lib.MISSING_SYMBOLS == []
    declared = _declared_functions()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/tacex_hip.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature in tacex_amd/_lib.py"
    assert sorted(_lib.SIGNATURES) == declared, "ctypes table and header disagree"
    hdr = int(re.search(r"#define\s+TACEX_ABI_VERSION\s+(\d+)", (REPO / "include" / "tacex_hip.h").read_text()).group(1))
    assert lib.tacex_abi_version() == hdr == _lib.ABI_VERSION


def test_argument_validation_needs_no_gpu():
    """NULL / bad arguments are rejected before any HIP call, with a message in tacex_last_error()."""
    import ctypes as C

    from tacex_amd import _lib

    lib = _lib.load_library()
    h = C.c_void_p()
    assert lib.tacex_taxim_create(0, None, C.byref(h)) == 2
    assert b"null" in lib.tacex_last_error()
    p = _lib.TaximParams()
    p.n_levels = 99
    assert lib.tacex_taxim_create(0, C.byref(p), C.byref(h)) == 2
    assert b"n_levels" in lib.tacex_last_error()
    assert lib.tacex_taxim_render(None, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0) == 2
    assert lib.tacex_fots_markers(None, 0, 0, 0, 0, 0, 0, 0, 1, 0) == 2
    assert lib.tacex_fem_create(0, None, C.byref(h)) == 2
    assert lib.tacex_taxim_workspace_bytes(None, 4) == 0
    assert lib.tacex_fots_state_bytes(3) == 3 * 8 * 4


def test_no_gpu_fails_loudly(calib_dir):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from tacex_amd import _lib
    from tacex_amd.simulation_approaches.gpu_taxim.sim import Taxim, TaximHip

    with pytest.raises(_lib.TacexHipError):
        Taxim(calib_folder=calib_dir, device="cpu")  # the reference's CPU path is not shipped
    with pytest.raises((_lib.TacexHipError, RuntimeError, AssertionError)):
        TaximHip(calib_folder=calib_dir, device="cuda:0")  # no device -> loud failure, never a fallback


def test_product_never_imports_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    for f in (REPO / "tacex_amd").rglob("*.py"):
        src = f.read_text()
        assert "import oracle" not in src and "from oracle" not in src, f
    for f in (REPO / "tacex_amd" / "csrc").glob("*"):
        # native sources may CITE the oracle (which restatement a kernel follows) in comments; nothing may include, open or run it
        for ln in f.read_text().splitlines():
            if "oracle/" in ln:
                code = ln.split("//")[0]
                assert "oracle/" not in code or code.lstrip().startswith(("*", "/*")), (f.name, ln)

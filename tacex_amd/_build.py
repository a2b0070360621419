"""Builds libtacex_hip.so in-tree with hipcc for gfx950 (no torch involvement, no JIT cache)."""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
INCLUDE = PKG.parent / "include"
# TACEX_LIB_TAG=<name>: a second, separately stamped build (libtacex_hip.<name>.so, objects under build.<name>/) whose extra
# compiler flags are remembered in libtacex_hip.<name>.flags - kernel A/B variants are compiled in the build container, travel
# to the GPU box next to the product library and are selected there by the tag alone (scripts/ab_*.sh); unset = the product
_TAG = os.environ.get("TACEX_LIB_TAG", "")
LIB = PKG / (f"libtacex_hip.{_TAG}.so" if _TAG else "libtacex_hip.so")
STAMP = PKG / (LIB.name + ".stamp")
_FLAGS_FILE = PKG / f"libtacex_hip.{_TAG}.flags"
SOURCES = ["taxim_kernels.hip", "taxim_mfma.hip", "taxim_tail.hip", "taxim_stream.hip", "taxim_shadow.hip", "fots_kernels.hip", "fem_kernels.hip", "depth_raster.hip", "tacex_capi.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-Wno-unused-result"]
# A/B hook for kernel tuning macros, e.g. TACEX_EXTRA_HIPCC_FLAGS="-DTACEX_MFMA_CH=2" (part of the build digest)
_STREAM_DEFAULT = "-mllvm -amdgpu-sched-strategy=max-memory-clause"
_stream_flags = os.environ.get("TACEX_STREAM_HIPCC_FLAGS", _STREAM_DEFAULT)
if _TAG and "TACEX_EXTRA_HIPCC_FLAGS" not in os.environ and "TACEX_STREAM_HIPCC_FLAGS" not in os.environ and _FLAGS_FILE.exists():
    _side = _FLAGS_FILE.read_text().split("\n")  # line 1: flags of every file, line 2 (optional): flags of the streaming tail
    FLAGS += _side[0].split()
    if len(_side) > 1:
        _stream_flags = _side[1]
else:
    FLAGS += os.environ.get("TACEX_EXTRA_HIPCC_FLAGS", "").split()
# per-file flags (part of the digest): the streaming tail is scalar f32 FMA chains - SLP packing into v_pk_fma_f32 (half rate on
# gfx950, scripts/hip_probes/valu_rates.hip) only adds register shuffles and pushed the kernel into scratch
# TACEX_STREAM_HIPCC_FLAGS: A/B hook for flags of the streaming tail alone (scheduler strategies, profiles/r03_experiments.md section 11)
# Scheduler: -amdgpu-sched-strategy=max-memory-clause measured -3.7 % on the streaming tail (769 vs 798 us per 1024 frames; max-ilp -1.6 %)
FILE_FLAGS = {"taxim_stream.hip": ["-fno-slp-vectorize"] + _stream_flags.split(),
              # the rasteriser must round every product on its own (bit-equal to its NumPy restatement): with the global
              # -ffp-contract=fast the backend fuses multiply-adds even under `#pragma clang fp contract(off)`
              "depth_raster.hip": ["-ffp-contract=off"]}


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (need ROCm to build libtacex_hip.so)")


def _digest() -> str:
    h = hashlib.sha256()
    for f in sorted(list(CSRC.glob("*.hip")) + list(CSRC.glob("*.h")) + list(INCLUDE.glob("*.h"))):
        h.update(f.name.encode())
        h.update(f.read_bytes())
    h.update(" ".join(FLAGS).encode())
    h.update(repr(sorted(FILE_FLAGS.items())).encode())
    return h.hexdigest()


def build_library(force: bool = False, verbose: bool = False) -> Path:
    """Compile every HIP translation unit to an object (in parallel) and link the shared library."""
    # TACEX_LIB_FROZEN=1 with a tag: use the tagged library as it was built, whatever the sources say now - an A/B partner built
    # from an OLDER commit (git worktree) and carried to the GPU box beside the product library
    if _TAG and os.environ.get("TACEX_LIB_FROZEN") == "1" and LIB.exists():
        return LIB
    dig = _digest()
    if not force and LIB.exists() and STAMP.exists() and STAMP.read_text().strip() == dig:
        return LIB
    # several ranks of one node may get here at once (one process per GPU): one builds, the others wait and re-check
    import fcntl

    lock = open(PKG / ".build.lock", "w")
    fcntl.flock(lock, fcntl.LOCK_EX)
    try:
        if not force and LIB.exists() and STAMP.exists() and STAMP.read_text().strip() == dig:
            return LIB
        return _build_locked(dig, verbose)
    finally:
        fcntl.flock(lock, fcntl.LOCK_UN)
        lock.close()


def _mllvm_supported(hipcc: str, flags: list, objdir: Path) -> bool:
    """`-mllvm <opt>` names a hidden LLVM option: a hipcc / LLVM without it stops with 'Unknown command line argument' and the
    library would not build for the sake of a 3.7 % scheduling preference.  One empty-kernel compile decides."""
    probe = objdir / "_mllvm_probe.hip"
    probe.write_text("#include <hip/hip_runtime.h>\n__global__ void tacex_probe() {}\n")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-c", *flags, str(probe), "-o", str(objdir / "_mllvm_probe.o")],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    return r.returncode == 0


def _build_locked(dig: str, verbose: bool) -> Path:
    hipcc = _hipcc()
    objdir = PKG / (f"build.{_TAG}" if _TAG else "build")
    objdir.mkdir(exist_ok=True)
    file_flags = dict(FILE_FLAGS)
    sched = _stream_flags.split()
    if "-mllvm" in sched and "TACEX_STREAM_HIPCC_FLAGS" not in os.environ and not _mllvm_supported(hipcc, sched, objdir):
        # the DEFAULT scheduler preference is optional (an explicitly requested flag set is passed through and may fail loudly)
        file_flags["taxim_stream.hip"] = [f for f in file_flags["taxim_stream.hip"] if f not in sched]
        if verbose:
            print(f"hipcc does not know {' '.join(sched)}: building taxim_stream.hip without it")
    srcs = [CSRC / s for s in SOURCES if (CSRC / s).exists()]
    procs = []
    for s in srcs:
        obj = objdir / (s.stem + ".o")
        cmd = [hipcc, *FLAGS, *file_flags.get(s.name, []), f"-I{INCLUDE}", f"-I{CSRC}", "-c", str(s), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd))
        procs.append((s, obj, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    objs = []
    for s, obj, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {s.name}:\n{out}")
        objs.append(str(obj))
    tmp = LIB.with_suffix(".so.tmp")  # link beside, then rename: a reader never maps a half-written library
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", str(tmp)]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    os.replace(tmp, LIB)
    STAMP.write_text(dig)
    if _TAG:
        _FLAGS_FILE.write_text(os.environ.get("TACEX_EXTRA_HIPCC_FLAGS", "") + "\n" + _stream_flags)
    return LIB


if __name__ == "__main__":
    import sys

    print(build_library(force="--force" in sys.argv, verbose=True))

"""Builds libpylinac_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

``python -m pylinac_amd._build`` or ``__graft_entry__.build()``.  One object per .hip file,
compiled in parallel, linked into ``pylinac_amd/libpylinac_hip.so`` (git-ignored, but shipped to
the GPU box by gpurun).
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
BUILD = PKG.parent / "build" / "hip"
LIB = PKG / "libpylinac_hip.so"

# -ffp-contract=off is part of the CONTRACT, not a tuning flag: the kernels reproduce scipy's
# float64 operation order, an FMA contraction changes results ("exactness").
HIPCC_FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-ffp-contract=off",
    "-fno-fast-math",
    "-Wall",
    "-Wno-unused-function",
]


# per-file additions: the matrix-core Gaussian keeps its accumulators in VGPRs (gfx950's register file is unified; the
# default AGPR form costs one v_accvgpr_read per accumulator value the integer recombination touches)
EXTRA_FLAGS = {"gaussian_mm.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
               # edge_field asks for four waves per SIMD; its radius >= 6 instantiations settle for three and say so
               "edge_stream.hip": ["-Wno-pass-failed"]}


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC=...)")


def sources() -> list[Path]:
    return sorted(CSRC.glob("*.hip"))


def _stale(target: Path, deps: list[Path]) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(d.stat().st_mtime > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> Path:
    cc = hipcc()
    extra = os.environ.get("PL_EXTRA_HIPCC_FLAGS", "").split()
    # a variant build (e.g. -DPL_OTSU_VARIANT=1 compiles a stopwatch kernel that gives WRONG thresholds) never touches the
    # product library or its objects: objects and the linked library go to build/variants/<hash>/, and only a process that
    # sets PYLINAC_HIP_LIB to that path (see _lib.lib_path) loads it
    tag = hashlib.sha256(" ".join(extra).encode()).hexdigest()[:12] if extra else ""
    build_dir = BUILD if not extra else BUILD.parent / "variants" / tag
    target = LIB if not extra else build_dir / "libpylinac_hip.so"
    build_dir.mkdir(parents=True, exist_ok=True)
    headers = list(CSRC.glob("*.h")) + [PKG.parent / "include" / "pylinac_hip.h", Path(__file__)]
    jobs = []
    objs = []
    for src in sources():
        obj = build_dir / (src.stem + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            jobs.append([cc, *HIPCC_FLAGS, *EXTRA_FLAGS.get(src.name, []), *extra, "-c", str(src), "-o", str(obj)])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        return r

    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if force or jobs or _stale(target, objs):
        run([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(target), *map(str, objs)])
    return target


if __name__ == "__main__":
    lib = build(force="--force" in sys.argv, verbose=True)
    print(lib)

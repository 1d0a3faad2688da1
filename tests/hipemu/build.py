"""TEST INFRASTRUCTURE ONLY: compile csrc/*.hip for the CPU wave64 emulator (tests/hipemu/hip/hip_runtime.h).

    python tests/hipemu/build.py            # -> tests/hipemu/_build/libpylinac_emu.so

The product (pylinac_amd/) never loads this library; tests/test_emulated_kernels.py does, to check kernel logic
against the oracle where there is no GPU.  Only the files listed in SOURCES are built: the ones whose device code
is plain C++ plus barriers / wave intrinsics (no gfx950 inline assembly or clang vector builtins on the live path).
"""
from __future__ import annotations

import pathlib
import re
import subprocess
import sys

HERE = pathlib.Path(__file__).resolve().parent
ROOT = HERE.parent.parent
CSRC = ROOT / "pylinac_amd" / "csrc"
BUILD = HERE / "_build"
LIB = BUILD / "libpylinac_emu.so"

# the ROCm toolchain's clang++ as a plain x86 host compiler: it understands the clang vector extensions
# (ext_vector_type, __builtin_elementwise_*) that gaussian_rw.hip is written in; g++ does for everything else
_CLANG = pathlib.Path("/opt/rocm/lib/llvm/bin/clang++")
CXX = str(_CLANG) if _CLANG.exists() else "g++"

SOURCES = ["runtime.hip", "interp.hip", "gamma.hip", "roi.hip", "canny.hip", "elementwise.hip", "reduce.hip",
           "edge.hip", "circle.hip", "spectral.hip", "xim.hip", "planar.hip", "ccl.hip", "ct.hip", "features.hip", "peaks.hip",
           "hist_otsu.hip", "picketfence.hip", "median.hip", "gaussian.hip", "features_sweep.hip", "slice_regions.hip", "edge_stream.hip", "hill.hip", "dicom.hip", "ct_axis.hip"]
if CXX != "g++":
    SOURCES += ["gaussian_rw.hip", "gaussian_mm.hip", "edge_stream32.hip"]

_DYN = re.compile(r"extern\s+__shared__\s+(?:__attribute__\(\(aligned\(\d+\)\)\)\s+)?([\w\s]+?)\s+(\w+)\[\];")


_WAITCNT = re.compile(r'asm\s+volatile\s*\(\s*"s_waitcnt[^;]*;')
_MED3 = re.compile(r'asm\("v_med3_[ui]32[^;]*;')
_MINF64 = re.compile(r'asm\("v_min_f64[^;]*;')                           # operands a, b -> r: the NaN-ignoring minimum = fmin
_MAXF64 = re.compile(r'asm\("v_max_f64[^;]*;')
_PIN = re.compile(r'asm\s+volatile\s*\(\s*""[^;]*;')                      # empty asm: a register-allocation hint
_LDSABS = re.compile(r'__hip_atomic_fetch_add\(\(pl_lds_u32\*\)\(uintptr_t\)byte_addr[^;]*;')   # absolute LDS address -> offset into the emulator's block
_LDSBASE = re.compile(r'return \(unsigned\)reinterpret_cast<uintptr_t>\(lds_ptr\);')
_LDSTYPE = re.compile(r'typedef __attribute__\(\(address_space\(3\)\)\) unsigned pl_lds_u32;')
_CONSTAS = re.compile(r'#define PL_CONSTANT_AS __attribute__\(\(address_space\(4\)\)\)')   # scalar-load hint: plain pointer here
_OCC = re.compile(r'__attribute__\(\(amdgpu_waves_per_eu\([^)]*\)\)\)')   # occupancy target of a kernel


def _rewrite(text: str) -> str:
    # gfx950 inline assembly: waits are meaningless here; v_med3 (operands a, b, c -> r) is spelled out
    text = _WAITCNT.sub(";", text)
    text = _PIN.sub(";", text)
    text = _OCC.sub("", text)
    text = _CONSTAS.sub("#define PL_CONSTANT_AS", text)
    text = _LDSABS.sub("atomicAdd(reinterpret_cast<unsigned*>(static_cast<unsigned char*>(hipemu::dyn_lds()) + byte_addr), v);", text)
    text = _LDSBASE.sub("return (unsigned)(static_cast<const unsigned char*>(lds_ptr) - static_cast<const unsigned char*>(hipemu::dyn_lds()));", text)
    text = _LDSTYPE.sub("", text)
    text = _MED3.sub("r = std::max(std::min(a, b), std::min(std::max(a, b), c));", text)
    text = _MINF64.sub("r = std::fmin(a, b);", text)
    text = _MAXF64.sub("r = std::fmax(a, b);", text)
    # `extern __shared__ T name[];`  ->  a pointer to the emulator's dynamic-LDS buffer
    return _DYN.sub(lambda m: f"{m.group(1)}* {m.group(2)} = reinterpret_cast<{m.group(1)}*>(hipemu::dyn_lds());", text)


def build(sources=None, verbose: bool = False) -> pathlib.Path:
    sources = [s for s in (sources or SOURCES) if (CSRC / s).exists()]
    BUILD.mkdir(exist_ok=True)
    inputs = [CSRC / s for s in sources] + list(CSRC.glob("*.h")) + [HERE / "hipemu.cpp", HERE / "emu_stubs.cpp", HERE / "hip" / "hip_runtime.h",
                                            ROOT / "include" / "pylinac_hip.h", pathlib.Path(__file__)]
    if LIB.exists() and all(LIB.stat().st_mtime >= p.stat().st_mtime for p in inputs):
        return LIB
    (BUILD / "pl_common.h").write_text(f'#line 1 "{CSRC / "pl_common.h"}"\n' + _rewrite((CSRC / "pl_common.h").read_text())
                                       .replace('"../../include/pylinac_hip.h"', f'"{ROOT / "include" / "pylinac_hip.h"}"'))
    cpps = []
    for s in sources:
        out = BUILD / (pathlib.Path(s).stem + "_emu.cpp")
        out.write_text(f'#line 1 "{CSRC / s}"\n' + _rewrite((CSRC / s).read_text()))
        cpps.append(str(out))
    cmd = [CXX, "-std=c++17", "-O1", "-g", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
           "-Wno-attributes", "-Wno-unknown-pragmas", f"-I{HERE}", f"-I{BUILD}", f"-I{CSRC}", *cpps, str(HERE / "hipemu.cpp"),
           *([] if "gaussian_rw.hip" in sources else [str(HERE / "emu_stubs.cpp")]),
           "-o", str(LIB)]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(verbose=True))

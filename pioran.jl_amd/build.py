"""Builds libpioran_hip.so (gfx950) in-tree with hipcc.  `python pioran.jl_amd/build.py [-f]`."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
OBJ = PKG / "_obj"
LIB = PKG / "libpioran_hip.so"
SOURCES = ["celerite_scan.hip", "celerite_wide.hip", "celerite_block.hip", "celerite_tile.hip", "celerite_tp.hip", "celerite_predict.hip", "celerite_fallback.hip", "table.hip", "approx.hip", "dense.hip", "capi.hip"]
HEADERS = [CSRC / "common.h", CSRC / "window_common.h", CSRC / "ldl_steps.inc", PKG.parent / "include" / "pioran_hip.h"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# per-source additions.  celerite_tile.hip: keep the operands and results of the matrix instructions in VGPRs where the allocator has the choice — the
# kernels' vector work (rescaling the state, LDS copies, the adjoint's products) otherwise reaches them through v_accvgpr moves (2568 -> 1516 in the file)
EXTRA_FLAGS = {"celerite_tile.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found: libpioran_hip.so cannot be built")


def _stale(target: Path, deps) -> bool:
    if not target.exists():
        return True
    mt = target.stat().st_mtime
    return any(Path(d).stat().st_mtime > mt for d in deps)


def build(force: bool = False, verbose: bool = True) -> Path:
    OBJ.mkdir(exist_ok=True)
    cc = hipcc()
    jobs = []
    for s in SOURCES:
        src, obj = CSRC / s, OBJ / (Path(s).stem + ".o")
        if force or _stale(obj, [src, *HEADERS]):
            jobs.append([cc, *FLAGS, *EXTRA_FLAGS.get(s, []), "-c", str(src), "-o", str(obj)])

    def run(cmd):
        if verbose:
            print(" ".join(cmd[-4:]), flush=True)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    objs = [str(OBJ / (Path(s).stem + ".o")) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        tmp = PKG / f"libpioran_hip.{os.getpid()}.so"
        subprocess.run([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(tmp), *objs], check=True)
        os.replace(tmp, LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="-f" in sys.argv))

"""Build libinklayer_hip.so (gfx950 only) with hipcc.

`python -m inklayer_amd.build` cross-compiles every csrc/*.hip for MI355X and
links ONE C-ABI shared library in-tree (inklayer_amd/lib/), so it travels to
the GPU box with the repo snapshot.  No torch headers are involved: the library
only depends on the HIP runtime.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent
CSRC = ROOT / "csrc"
LIBDIR = ROOT / "lib"
OBJDIR = LIBDIR / "obj"
LIB = LIBDIR / "libinklayer_hip.so"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-mllvm", "-amdgpu-mfma-vgpr-form",   # MFMA C/D in arch VGPRs: no v_accvgpr copies around softmax

         "-Wall", "-Wno-unused-function", "-Wno-unused-variable"]


def _newest_header() -> float:
    hs = list(CSRC.glob("*.h")) + list((ROOT.parent / "include").glob("*.h"))
    return max(h.stat().st_mtime for h in hs)


# Kernels whose LDS fragment reads are volatile asm with hand-counted waits (ffn_fused.hip, proj_ln.hip): the compiler
# does not know that such a read completes later, so a SPILL of its destination register would store the register
# before the data has arrived (round 3: NaNs from a 32-byte spill).  Their build fails unless every kernel of the file
# reports zero scratch.
NO_SPILL = {"ffn_fused.hip", "proj_ln.hip"}


def _check_no_spill(src: Path, remarks: str) -> None:
    name = None
    for line in remarks.splitlines():
        if "Function Name:" in line:
            name = line.split("Function Name:")[1].split("[")[0].strip()
        elif "ScratchSize [bytes/lane]:" in line:
            n = int(line.split("ScratchSize [bytes/lane]:")[1].split("[")[0])
            if n != 0:
                raise RuntimeError(f"{src.name}: kernel {name} spills ({n} bytes of scratch per lane) - its asm LDS reads "
                                   f"are not spill-safe; reduce register pressure")


def _compile(src: Path, force: bool, hdr_time: float) -> Path:
    obj = OBJDIR / (src.stem + ".o")
    if (not force and obj.exists() and obj.stat().st_mtime > src.stat().st_mtime
            and obj.stat().st_mtime > hdr_time):
        return obj
    guard = src.name in NO_SPILL
    cmd = [HIPCC, *FLAGS, *(["-Rpass-analysis=kernel-resource-usage", "-fno-caret-diagnostics"] if guard else []), "-c", str(src), "-o", str(obj)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src.name}:\n{r.stdout}\n{r.stderr}")
    err = r.stderr
    if guard:
        try:
            _check_no_spill(src, err)
        except RuntimeError:
            obj.unlink(missing_ok=True)
            raise
        err = "\n".join(l for l in err.splitlines() if "kernel-resource-usage" not in l)
    if err.strip():
        sys.stderr.write(err)
    return obj


def build(force: bool = False, verbose: bool = True, ablation: bool = False) -> Path:
    """ablation=True: a SEPARATE library (lib/libinklayer_hip_ablation.so, -DINK_ABLATION) that also contains the
    measurement-only GEMM kernels (no-MFMA / no-epilogue / timeline variants); tools select it with
    INKLAYER_HIP_LIB.  The product library never contains them."""
    if ablation:
        return _build_ablation(verbose)
    OBJDIR.mkdir(parents=True, exist_ok=True)
    srcs = sorted(CSRC.glob("*.hip"))
    if not srcs:
        raise RuntimeError("no HIP sources found")
    hdr_time = _newest_header()
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force, hdr_time), srcs))
    if (force or not LIB.exists()
            or any(o.stat().st_mtime > LIB.stat().st_mtime for o in objs)):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB),
               *map(str, objs)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"[inklayer_amd.build] {LIB} ({LIB.stat().st_size >> 10} KiB, {len(srcs)} sources)")
    return LIB


def _build_ablation(verbose: bool) -> Path:
    odir = LIBDIR / "obj_ablation"
    odir.mkdir(parents=True, exist_ok=True)
    lib = LIBDIR / "libinklayer_hip_ablation.so"
    objs = []
    for src in sorted(CSRC.glob("*.hip")):
        obj = odir / (src.stem + ".o")
        r = subprocess.run([HIPCC, *FLAGS, "-DINK_ABLATION", "-c", str(src), "-o", str(obj)], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src.name}:\n{r.stdout}\n{r.stderr}")
        objs.append(obj)
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(lib), *map(str, objs)],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"[inklayer_amd.build] {lib} (ablation build)")
    return lib


if __name__ == "__main__":
    build(force="--force" in sys.argv, ablation="--ablation" in sys.argv)

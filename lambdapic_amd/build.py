"""Build the HIP library in-tree: lambdapic_amd/liblambdapic_amd.so (gfx950 only).

    python -m lambdapic_amd.build [--force] [--verbose]

hipcc cross-compiles without a GPU.  Objects are rebuilt when their source (or a header) is newer.
"""
from __future__ import annotations

import os
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
LIB = HERE / "liblambdapic_amd.so"
SOURCES = ["lpa_fields.hip", "lpa_particles.hip", "lpa_particles3d.hip", "lpa_sort.hip", "lpa_patches.hip",
           "lpa_rho.hip", "lpa_step.hip", "lpa_patches3d.hip", "lpa_comm.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-munsafe-fp-atomics",
         "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (Path(c).exists() or c == "hipcc"):
            return c
    raise RuntimeError("hipcc not found")


def build_variant(name: str, defines, extra_flags=()) -> Path:
    """diagnostic build with extra -D flags (and compiler flags) into lambdapic_amd/csrc/build/<name>.so (not the product)"""
    out = CSRC / "build" / f"liblambdapic_amd_{name}.so"
    out.parent.mkdir(exist_ok=True)
    cmd = [_hipcc(), *FLAGS, *extra_flags, *[f"-D{d}" for d in defines], "-shared", *[str(CSRC / s) for s in SOURCES],
           "-ldl", "-o", str(out)]
    subprocess.run(cmd, check=True)
    return out


VARIANTS_LIB = CSRC / "build" / "liblambdapic_amd_variants.so"


def build_variants(force: bool = False) -> Path:
    """the same library with the three measured-slower deposit paths of the 2-D tiled kernel compiled in
    (-DLPA_K1_VARIANTS=1: wave reduce-scatter, in-kernel re-seating, cooperative deposit); loaded by the tests that
    pin those paths (``_lib.use_variants()``) and by ``bench.py --order padded / --reseat``, never by the product"""
    srcs = [CSRC / s for s in SOURCES] + list(CSRC.glob("*.hpp")) + [HERE.parent / "include" / "lambdapic_amd.h"]
    if not force and VARIANTS_LIB.exists() and VARIANTS_LIB.stat().st_mtime >= max(f.stat().st_mtime for f in srcs):
        return VARIANTS_LIB
    return build_variant("variants", ["LPA_K1_VARIANTS=1"])


def build(force: bool = False, verbose: bool = False, extra_flags=()) -> Path:
    headers = list(CSRC.glob("*.hpp")) + [HERE.parent / "include" / "lambdapic_amd.h"]
    newest_hdr = max(h.stat().st_mtime for h in headers)
    objdir = CSRC / "build"
    objdir.mkdir(exist_ok=True)
    objs, rebuilt = [], False
    procs = []
    for s in SOURCES:
        src, obj = CSRC / s, objdir / (s + ".o")
        objs.append(obj)
        if force or not obj.exists() or obj.stat().st_mtime < max(src.stat().st_mtime, newest_hdr):
            cmd = [_hipcc(), *FLAGS, *extra_flags, "-c", str(src), "-o", str(obj)]
            if verbose:
                print(" ".join(cmd))
            procs.append((s, subprocess.Popen(cmd)))
            rebuilt = True
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {s}")
    if rebuilt or not LIB.exists():
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *map(str, objs), "-ldl", "-o", str(LIB)]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv or True))
    print(build_variants(force="--force" in sys.argv))

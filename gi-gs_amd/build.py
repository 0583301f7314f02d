"""Builds libgigs_hip.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

    python gi-gs_amd/build.py [--force] [--save-temps]

No torch headers are involved: the boundary is a plain C ABI (include/gigs_hip.h).  hipcc
cross-compiles without a GPU.  -ffp-contract=off is part of the numerical contract of the
library (see csrc/gigs_common.h), not an optimisation knob.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.environ.get("GIGS_OBJ", os.path.join(HERE, "build"))
LIB = os.environ.get("GIGS_LIB", os.path.join(HERE, "libgigs_hip.so"))
SOURCES = ["api.hip", "preprocess.hip", "binning.hip", "blend.hip", "gi.hip", "pbr.hip", "stage2.hip", "train_glue.hip", "knn.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = [
    "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-std=c++17",
    "-Wall", "-Wno-unused-function", "-Wno-unused-result", "-DNDEBUG",
] + os.environ.get("GIGS_EXTRA_FLAGS", "").split()


def _deps(src: str):
    yield os.path.join(CSRC, src)
    yield os.path.join(CSRC, "gigs_common.h")
    yield os.path.join(CSRC, "pixel_ops.h")
    yield os.path.join(HERE, "..", "include", "gigs_hip.h")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, save_temps: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    sources = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    objs, jobs = [], []
    for src in sources:
        obj = os.path.join(OBJ, src.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, _deps(src)):
            cmd = [HIPCC, *FLAGS, "-c", os.path.join(CSRC, src), "-o", obj]
            if save_temps:
                cmd += ["-save-temps=obj", "-Rpass-analysis=kernel-resource-usage"]
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        return r.stderr

    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        logs = list(ex.map(run, jobs))
    if save_temps:
        with open(os.path.join(OBJ, "resource_usage.txt"), "w") as f:
            f.write("\n".join(logs))
    if jobs or force or _stale(LIB, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, save_temps="--save-temps" in sys.argv))

#!/usr/bin/env python3
"""Build librbl.so (hand-written HIP for gfx950) in-tree with hipcc.

    python admm-for-rank-based-loss_amd/csrc/build.py [--force]

One hipcc -c per .hip source (in parallel), then one link.  The .so stays next to the
sources (git-ignored, but it travels to the GPU box with the snapshot)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["api.hip", "sweep.hip", "sweep_erm.hip", "elementwise.hip", "sort.hip", "pav.hip", "wstep.hip", "lasso_fs.hip", "gram.hip", "synth.hip", "eig.hip", "baselines.hip", "zband.hip"]
HEADERS = ["rbl_internal.h", "device_math.h", os.path.join("..", "..", "include", "rbl.h")]
OUT = os.path.join(HERE, "librbl.so")
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-ffp-contract=off"]


def _newer(src, dst):
    return (not os.path.exists(dst)) or os.path.getmtime(src) > os.path.getmtime(dst)


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(HERE, "_obj")
    os.makedirs(objdir, exist_ok=True)
    hdr_m = max(os.path.getmtime(os.path.join(HERE, h)) for h in HEADERS)
    jobs = []
    for src in SOURCES:
        s = os.path.join(HERE, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        if force or _newer(s, o) or hdr_m > os.path.getmtime(o):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        cmd = [hipcc] + FLAGS + os.environ.get("RBL_HIPCC_FLAGS", "").split() + ["-c", s, "-o", o]   # (extra flags: kernel experiments)
        r = subprocess.run(cmd, capture_output=True, text=True)
        return job, r

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            for (s, o), r in ex.map(cc, jobs):
                if verbose and r.stderr.strip():
                    sys.stderr.write(r.stderr)
                if r.returncode != 0:
                    raise RuntimeError(f"hipcc failed on {s}:\n{r.stderr}")
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or not os.path.exists(OUT):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))

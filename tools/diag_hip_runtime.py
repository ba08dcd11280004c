"""Diagnostic: can torch's HIP runtime and librbl's coexist in one process, in either
initialisation order?  (python tools/diag_hip_runtime.py lib_first|torch_first)"""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
order = sys.argv[1] if len(sys.argv) > 1 else "lib_first"
import numpy as np

if order == "lib_first":
    import admm_for_rank_based_loss_amd as rbl
    print("k_prox", rbl._lib.k_prox("hinge", np.ones(3), 1.0, np.zeros(3)))
    import torch
    print("torch avail", torch.cuda.is_available(), torch.cuda.device_count())
    try:
        torch.cuda.init()
        print("init ok", torch.zeros(3, device="cuda").sum().item())
    except Exception as e:
        print("ERR", e)
else:
    import torch
    print("torch avail", torch.cuda.is_available())
    torch.cuda.init()
    print("init ok", torch.zeros(3, device="cuda").sum().item())
    import admm_for_rank_based_loss_amd as rbl
    print("k_prox", rbl._lib.k_prox("hinge", np.ones(3), 1.0, np.zeros(3)))
with open("/proc/self/maps") as f:
    libs = sorted({l.split()[-1] for l in f if "hip" in l.lower() or "hsa" in l.lower()})
print("\n".join(libs))

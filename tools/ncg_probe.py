#!/usr/bin/env python3
"""The linear first phase of the smoothed-l1 w-step (csrc/wstep.hip: k_ncg_persist, phase A) emulated on the host with the
device's G, q, w, rho and t of an sADMM run at C2smooth's size: per selected ADMM iteration the number of coordinates
outside the assumed Huber pattern and the residual after every CG step, next to the inner iterations the device's
nonlinear CG took (run with RBL_NCG_ACTIVE=0 for that comparison).  This is the evidence behind the rules in the kernel
(round 3): crossings in the first steps come back, a wrong pattern shows by the 6th step.
    python tools/ncg_probe.py [rows]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import admm_for_rank_based_loss_amd as rbl
from admm_for_rank_based_loss_amd import _lib
from admm_for_rank_based_loss_amd.dist import _DevArray
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 6_000_000
s = rbl.Solver(rows, 1000, "erm", "binary_cross_entropy", reg=0.01, wstep=3, storage="f32", tol=0.0)
s.generate_synthetic(17); s.gram()
def view(which):
    p, c = s.buffer(which)
    return torch.as_tensor(_DevArray(p, c), device="cuda:0")
G = view(_lib.BUF_G).cpu().numpy().reshape(1000, 1000)
reg = 0.01
def hub_g(u,t): return np.where(np.abs(u)<=t, reg*u/(2*t), np.sign(u)*0.5*reg)
def hub_c(u,t): return np.where(np.abs(u)<=t, reg/(2*t), 0.0)
for it in range(60):
    s.phase_m(); s.phase_z(); s.phase_q()
    torch.cuda.synchronize()
    st0 = s.get_state(want_z=False, want_lam=False)
    q = view(_lib.BUF_Q).cpu().numpy()[:1000].copy(); w0 = st0["w"].copy(); rho = st0["rho"]; t = st0["smooth_t"]
    s.phase_w(); s.phase_dual(False); st = s.phase_finish()
    if it in (3, 6, 10, 16, 18, 20, 25, 30, 40, 50, 58):
        wstar = s.get_state(want_z=False, want_lam=False)["w"]
        gd = np.diag(G); w = w0.copy(); gw = G @ w
        c = hub_c(w, t); sg = np.where(c == 0, np.sign(w), 0.0)
        r = -(rho*(gw-q)+hub_g(w,t)); Minv = 1/(rho*gd+c); z = Minv*r; p = z.copy(); rz = r@z
        thr = 1e-13*max(np.max(rho*np.abs(q)), 0.5*reg); tr=[]
        for k in range(25):
            gp=G@p; Ap=rho*gp+c*p; al=rz/(p@Ap); w+=al*p; gw+=al*gp; r-=al*Ap; z=Minv*r; rz2=r@z; be=rz2/rz; p=z+be*p; rz=rz2
            quad=np.abs(w)<=t; bad=int(np.sum((quad!=(c!=0))|(~quad&(np.sign(w)!=sg))))
            tr.append((bad, float('%.1g'%np.max(np.abs(r)))))
            if np.max(np.abs(r))<=thr: break
        print("iter", it, "t %.3g rho %.3g device inner %d form %d | nquad %d | emulated phase A:" % (t, rho, st.inner_iters, st.wstep_form, int((np.abs(w0)<=t).sum())), tr[:12], "|w-w*| %.1e" % np.max(np.abs(w-wstar)))

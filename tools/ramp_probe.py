#!/usr/bin/env python3
"""What ramps when the sweep kernel starts at 4.6 ms and settles at 3.7 ms?  

    python tools/ramp_probe.py [--rows 6000000] [--out gpurun_out/ramp_probe.json]

Round 2's kernel trace (profiles/r02_C2_fused_kernel_stats.csv is its summary) shows k_sweep_erm falling
4.62 -> 3.70 ms over ~25 launches, and the SAME ramp starting again after a 43 ms pause of the device.  Two
explanations make different predictions:

  address translation / first touch of the 24 GB allocation   the ramp happens ONCE per allocation; idle
                                                             time does not bring it back
  clock / power state of the device                           every idle gap of some length brings it back;
                                                             the clocks the driver reports move with it

This probe runs C2's single-sweep iteration with a HIP-event pair around every launch and
  A  60 iterations back to back after set-up,
  B  idle gaps of 1 / 5 / 20 / 50 / 200 / 1000 ms (host sleeps, device idle) each followed by 30 iterations,
  C  the same gaps with the device kept busy by a compute-only kernel (the Gram MFMA kernel on a small handle:
     no HBM streaming) followed by 30 iterations,
while a thread samples the amdgpu sysfs clock tables (pp_dpm_sclk / mclk / fclk / socclk: the active level is
the line marked '*') and the hwmon power reading every 2 ms.  Output: per-launch milliseconds per section and the
clock / power samples, as JSON and as a table on stderr.
"""
import argparse
import glob
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class Sampler(threading.Thread):
    def __init__(self, period=0.002):
        super().__init__(daemon=True)
        self.period = period
        self.stop = False
        self.rows = []
        dev = None
        for c in sorted(glob.glob("/sys/class/drm/card*/device")):
            if os.path.exists(os.path.join(c, "pp_dpm_sclk")):
                dev = c
                break
        self.files = {}
        if dev:
            for k in ("pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk", "pp_dpm_socclk", "gpu_busy_percent", "mem_busy_percent"):
                p = os.path.join(dev, k)
                if os.access(p, os.R_OK):
                    self.files[k] = p
            for hw in glob.glob(os.path.join(dev, "hwmon", "hwmon*")):
                for k in ("power1_average", "power1_input", "freq1_input", "freq2_input"):
                    p = os.path.join(hw, k)
                    if os.access(p, os.R_OK):
                        self.files[k] = p
        self.dev = dev

    @staticmethod
    def _active(txt):
        for line in txt.splitlines():
            if "*" in line:
                return line.replace("*", "").strip()
        return txt.strip().replace("\n", " | ")[:60]

    def run(self):
        while not self.stop:
            t = time.perf_counter()
            row = {"t": t}
            for k, p in self.files.items():
                try:
                    with open(p) as f:
                        txt = f.read()
                    row[k] = self._active(txt) if k.startswith("pp_dpm") else txt.strip()
                except OSError:
                    pass
            self.rows.append(row)
            time.sleep(self.period)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=6_000_000)
    ap.add_argument("--cols", type=int, default=1000)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "ramp_probe.json"))
    a = ap.parse_args()
    import torch
    import admm_for_rank_based_loss_amd as rbl
    from admm_for_rank_based_loss_amd import _lib

    smp = Sampler()
    smp.start()
    marks = []

    def mark(name):
        marks.append((name, time.perf_counter()))

    mark("setup")
    s = rbl.Solver(a.rows, a.cols, "erm", "binary_cross_entropy", reg=0.01, wstep=1, storage="f32", tol=0.0)
    s.generate_synthetic(17)
    s.gram()
    s.profile_kernels(1)
    s.profile_sampling(1)
    # a small second handle whose Gram kernel is the compute-only filler of section C (fp64 MFMA, 400 MB of data)
    filler = rbl.Solver(100_000, a.cols, "erm", "binary_cross_entropy", reg=0.01, wstep=1, storage="f32", tol=0.0)
    filler.generate_synthetic(3)
    filler.gram()
    torch.cuda.synchronize()

    def run_iters(k):
        s.reset_kernel_times()
        t = []
        for _ in range(k):
            s.step(False)
            t.append(time.perf_counter())
        torch.cuda.synchronize()
        return [float(x) for x in s.kernel_samples(_lib.KERNEL_SWEEP_ERM)], t

    out = {"rows": a.rows, "cols": a.cols, "sections": [], "sysfs_device": smp.dev, "sysfs_files": sorted(smp.files)}
    mark("A")
    ms, t = run_iters(60)
    out["sections"].append({"name": "A: 60 iterations right after set-up", "kernel_ms": ms})
    for gap in (0.001, 0.005, 0.02, 0.05, 0.2, 1.0):
        mark("B gap %g" % gap)
        time.sleep(gap)
        ms, t = run_iters(30)
        out["sections"].append({"name": "B: idle %g ms, then 30 iterations" % (gap * 1e3), "gap_ms": gap * 1e3, "kernel_ms": ms})
    for gap in (0.02, 0.05, 0.2, 1.0):
        mark("C busy %g" % gap)
        t_end = time.perf_counter() + gap
        nfill = 0
        while time.perf_counter() < t_end:
            filler.gram_local()      # ~3 ms of fp64 MFMA per call (the call waits for its kernel: ~20 us of idle device between two)
            nfill += 1
        ms, t = run_iters(30)
        out["sections"].append({"name": "C: %g ms of compute-only kernels (no idle), then 30 iterations" % (gap * 1e3),
                                "gap_ms": gap * 1e3, "filler_launches": nfill, "kernel_ms": ms})
    mark("end")
    smp.stop = True
    smp.join(1.0)
    # clock samples: collapse to change points per section
    t0 = marks[0][1]
    changes = []
    prev = None
    for r in smp.rows:
        key = tuple((k, r.get(k)) for k in sorted(r) if k not in ("t", "power1_average", "power1_input", "gpu_busy_percent", "mem_busy_percent"))
        if key != prev:
            changes.append(dict(r, t=round((r["t"] - t0) * 1e3, 2)))
            prev = key
    out["marks_ms"] = [(n, round((t - t0) * 1e3, 2)) for n, t in marks]
    out["clock_changes"] = changes[:4000]
    pw = [(round((r["t"] - t0) * 1e3, 1), r.get("power1_average") or r.get("power1_input")) for r in smp.rows[::25]]
    out["power_samples"] = pw[:4000]
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(out, f)
    for sec in out["sections"]:
        k = sec["kernel_ms"]
        print("%-66s first %.3f  2nd %.3f  5th %.3f  10th %.3f  20th %.3f  last %.3f  min %.3f" % (
            sec["name"], k[0], k[1], k[4], k[9], k[19], k[-1], min(k)), file=sys.stderr)
    print("sysfs:", smp.dev, sorted(smp.files), "clock change points:", len(changes), file=sys.stderr)
    for c in changes[:60]:
        print(c, file=sys.stderr)
    s.close()
    filler.close()


if __name__ == "__main__":
    main()

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
    """samples EVERY amdgpu card of the host (the box shows all of them in sysfs, one is ours): the card whose
    gpu_busy_percent moves during the run is the one the probe ran on"""
    KEYS = ("pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk", "gpu_busy_percent")

    def __init__(self, period=0.004):
        super().__init__(daemon=True)
        self.period = period
        self.stop = False
        self.rows = []
        self.cards = {}
        for c in sorted(glob.glob("/sys/class/drm/card*/device")):
            if not os.path.exists(os.path.join(c, "pp_dpm_sclk")):
                continue
            files = {k: os.path.join(c, k) for k in self.KEYS if os.access(os.path.join(c, k), os.R_OK)}
            for hw in glob.glob(os.path.join(c, "hwmon", "hwmon*")):
                for k in ("power1_average", "power1_input"):
                    if os.access(os.path.join(hw, k), os.R_OK):
                        files["power"] = os.path.join(hw, k)
            self.cards[c.split("/")[-2]] = files

    @staticmethod
    def _active(txt):
        for line in txt.splitlines():
            if "*" in line:
                return line.replace("*", "").strip()
        return txt.strip().replace("\n", " | ")[:60]

    def run(self):
        while not self.stop:
            row = {"t": time.perf_counter()}
            for card, files in self.cards.items():
                for k, p in files.items():
                    try:
                        with open(p) as f:
                            txt = f.read()
                        row[card + ":" + k] = self._active(txt) if k.startswith("pp_dpm") else txt.strip()
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

    out = {"rows": a.rows, "cols": a.cols, "sections": []}
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
    # the card(s) that were busy at some point = ours; per section the range of each reported clock / power
    t0 = marks[0][1]
    busy_cards = sorted({k.split(":")[0] for r in smp.rows for k, v in r.items() if k.endswith("gpu_busy_percent") and v not in ("0", "")})
    out["cards_seen"] = sorted(smp.cards)
    out["cards_busy_during_run"] = busy_cards
    out["marks_ms"] = [(n, round((t - t0) * 1e3, 2)) for n, t in marks]
    sect = []
    for i in range(len(marks) - 1):
        lo, hi = marks[i][1], marks[i + 1][1]
        rows = [r for r in smp.rows if lo <= r["t"] < hi]
        summ = {"section": marks[i][0], "samples": len(rows)}
        for card in (busy_cards or sorted(smp.cards)[:1]):
            for k in ("pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk", "gpu_busy_percent", "power"):
                vals = sorted({r.get(card + ":" + k) for r in rows if r.get(card + ":" + k) is not None})
                summ[card + ":" + k] = vals if len(vals) <= 6 else [vals[0], "...", vals[-1], "%d distinct" % len(vals)]
        sect.append(summ)
    out["clock_summary_per_section"] = sect
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(out, f)
    for sec in out["sections"]:
        k = sec["kernel_ms"]
        print("%-66s first %.3f  2nd %.3f  5th %.3f  10th %.3f  20th %.3f  last %.3f  min %.3f" % (
            sec["name"], k[0], k[1], k[4], k[9], k[19], k[-1], min(k)), file=sys.stderr)
    print("cards in sysfs:", sorted(smp.cards), " busy during the run:", busy_cards, file=sys.stderr)
    for sm in sect:
        print(json.dumps(sm), file=sys.stderr)
    s.close()
    filler.close()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""HBM traffic per launch of the sweep kernels from two rocprofv3 PMC passes.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o f -- python3 bench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o w -- python3 bench.py ...
    python tools/traffic_from_pmc.py gpurun_out/pmc_fetch/f_counter_collection.csv \
                                     gpurun_out/pmc_write/w_counter_collection.csv rows cols storage tag

Units and corrections as MI355X_MICROARCH.md (HBM / rocprofv3 section) prescribes:
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of
a wide (16 B/lane) coalesced streaming read, so it is doubled; WRITE_SIZE is taken as is.
FETCH_SIZE and WRITE_SIZE do not fit one pass (3 + 2 of the 4 TCC slots): separate passes.
Writes profiles/traffic_latest.json (read by bench.py for roofline.traffic) and a per-kernel
summary profiles/<tag>_pmc_summary.csv."""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def mean_by_kernel(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


def main():
    fpath, wpath, rows, cols, storage, tag = sys.argv[1:7]
    fetch = mean_by_kernel(fpath, "FETCH_SIZE")
    write = mean_by_kernel(wpath, "WRITE_SIZE")
    names = sorted(set(fetch) | set(write), key=lambda k: -(fetch.get(k, (0, 0))[0]))
    out_rows = []
    kernels = {}
    for k in names:
        f, nf = fetch.get(k, (0.0, 0))
        w, nw = write.get(k, (0.0, 0))
        hbm = (2.0 * f + w) * 1024.0
        out_rows.append((k, nf, f, nw, w, hbm))
        short = None
        if "k_gemvt<" in k and "false" in k:
            short = "gemvt"
        elif "k_gemv<" in k:
            short = "gemv"
        elif "k_sweep_erm_wide<" in k:
            short = "sweep_erm"
        elif "k_sweep_erm<" in k:
            # k_sweep_erm<T, LOSS, P, R, S, WL, EXP, ONE>: EXP = 0 is the fused pass; 134 the v-only pass (timed in the
            # library's gemv slot), 93 the q-only pass (its gemvt slot)
            targs = k.split("k_sweep_erm<", 1)[1].split(">", 1)[0].split(",")
            exp = int(targs[6]) if len(targs) > 6 and targs[6].strip().lstrip("-").isdigit() else 0
            short = {0: "sweep_erm", 134: "sweep_v", 93: "sweep_q"}.get(exp)
        if short:
            kernels[short] = dict(kernel=k, launches=nf, fetch_size_kib=f, write_size_kib=w,
                                  hbm_bytes_per_launch=hbm)
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    with open(os.path.join(ROOT, "profiles", f"{tag}_pmc_summary.csv"), "w", newline="") as fh:
        wr = csv.writer(fh)
        wr.writerow(["kernel", "launches_fetch_pass", "mean_FETCH_SIZE_KiB", "launches_write_pass",
                     "mean_WRITE_SIZE_KiB", "hbm_bytes_per_launch=(2*FETCH+WRITE)*1024"])
        for r in out_rows[:20]:
            wr.writerow(r)
    with open(os.path.join(ROOT, "profiles", "traffic_latest.json"), "w") as fh:
        json.dump(dict(rows=int(rows), cols=int(cols), storage=storage, source=f"profiles/{tag}_pmc_summary.csv",
                       correction="hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024  (gfx950: FETCH_SIZE counts 1/2)",
                       kernels=kernels), fh, indent=1)
    for k, v in kernels.items():
        print(k, f"{v['hbm_bytes_per_launch']/1e9:.3f} GB per launch over {v['launches']} launches")


if __name__ == "__main__":
    main()

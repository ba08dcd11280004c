#!/usr/bin/env python3
"""Per-kernel LDS activity of the sort / PAV z-step from one rocprofv3 PMC pass.

    rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES \
              SQ_WAVE_CYCLES SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d DIR -o l -- python3 bench.py ...
    python tools/lds_from_pmc.py DIR/.../l_counter_collection.csv TAG

Writes profiles/<TAG>_lds_pmc_summary.csv: per kernel the mean counter values per launch, the
LDS size / VGPRs / workgroup size the dispatch was made with, and
    conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE   (share of LDS-array cycles lost to conflicts)
(MI355X_MICROARCH.md: BANK_CONFLICT = extra cycles, IDX_ACTIVE = all LDS-array cycles)."""
import collections
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COUNTERS = ["SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_LDS", "SQ_WAVES", "SQ_BUSY_CYCLES",
            "SQ_WAVE_CYCLES", "SQ_WAIT_INST_LDS"]


def main():
    path, tag = sys.argv[1:3]
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        meta[k] = (r.get("LDS_Block_Size", ""), r.get("VGPR_Count", ""), r.get("Workgroup_Size", ""),
                   r.get("Grid_Size", ""))
    rows = []
    for k, c in vals.items():
        mean = {n: (sum(c[n]) / len(c[n]) if c.get(n) else 0.0) for n in COUNTERS}
        launches = max(len(v) for v in c.values())
        frac = mean["SQ_LDS_BANK_CONFLICT"] / mean["SQ_LDS_IDX_ACTIVE"] if mean["SQ_LDS_IDX_ACTIVE"] else 0.0
        rows.append((mean["SQ_LDS_IDX_ACTIVE"], k, launches, meta[k], mean, frac))
    rows.sort(key=lambda r: -r[0])
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    out = os.path.join(ROOT, "profiles", f"{tag}_lds_pmc_summary.csv")
    with open(out, "w", newline="") as fh:
        wr = csv.writer(fh)
        wr.writerow(["kernel", "launches", "lds_block_bytes", "vgprs", "workgroup", "grid"] + ["mean_" + n for n in COUNTERS] +
                    ["conflict_frac"])
        for _, k, launches, m, mean, frac in rows[:30]:
            wr.writerow([k, launches, *m] + [round(mean[n], 1) for n in COUNTERS] + [round(frac, 4)])
    print(out)


if __name__ == "__main__":
    main()

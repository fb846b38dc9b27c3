#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes into profiles/<name>.json.

Usage: pmc_summarise.py FETCH_DIR WRITE_DIR OUT.json [--note TEXT] [--workload JSON]

Each DIR is the `-d` directory of one rocprofv3 counter pass over the same bench.py command
(separate passes: FETCH_SIZE and WRITE_SIZE do not fit one TCC pass).  Corrections follow
MI355X_MICROARCH.md's HBM section: counter unit = KiB; on gfx950 FETCH_SIZE reports half of the bytes of
a wide coalesced streaming read, so it is doubled; WRITE_SIZE is exact.  Infinity-Cache hits are counted
by FETCH_SIZE, so for kernels that re-read through L2/MALL the figure is an upper bound on HBM bytes.
bench.py reads the resulting file for `roofline.traffic`.  Averages are over a kernel's full-size launches (see collect()).
"""
import csv
import glob
import json
import re
import sys


def short(name: str) -> str:
    m = re.search(r"(k_[A-Za-z0-9_]+)(<[^(]*>)?\(", name)
    if not m:
        return name[:80]
    return m.group(1) + (m.group(2) or "")


ALL_LAUNCHES = False   # --all: keep every launch (a run that only holds one workload, e.g. the encoder at its fixed shape)


def collect(d: str, counter: str):
    """Per kernel: [sum of counter, launches, sum of ms] over its FULL-SIZE launches -- those lasting at least half as
    long as the kernel's longest launch.  One bench run launches a kernel on several workloads (the 10 M-row index of
    the headline, the 1 M-row indexes of the clustered extra, warm-ups of other shapes); the headline's launches are
    the long ones, and only they are comparable with its algorithmic bytes."""
    rows = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != counter:
                    continue
                k = short(row["Kernel_Name"])
                rows.setdefault(k, []).append((float(row["Counter_Value"]),
                                               (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-6))
    acc = {}
    for k, lst in rows.items():
        longest = max(ms for _, ms in lst)
        keep = [(v, ms) for v, ms in lst if ALL_LAUNCHES or ms >= 0.5 * longest]
        acc[k] = [sum(v for v, _ in keep), len(keep), sum(ms for _, ms in keep)]
    return acc


def main():
    global ALL_LAUNCHES
    ALL_LAUNCHES = "--all" in sys.argv
    fetch_dir, write_dir, out = sys.argv[1:4]
    note = ""
    if "--note" in sys.argv:
        note = sys.argv[sys.argv.index("--note") + 1]
    fe = collect(fetch_dir, "FETCH_SIZE")
    wr = collect(write_dir, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fe) | set(wr)):
        f = fe.get(k, [0.0, 0, 0.0])
        w = wr.get(k, [0.0, 0, 0.0])
        fk = f[0] / max(f[1], 1)
        wk = w[0] / max(w[1], 1)
        kernels[k] = {
            "launches": f[1] or w[1],
            "FETCH_SIZE_KiB_avg_per_launch": fk,
            "WRITE_SIZE_KiB_avg_per_launch": wk,
            "avg_ms_under_pmc": (f[2] / f[1]) if f[1] else None,
            "hbm_bytes_per_launch_corrected": 2.0 * fk * 1024.0 + wk * 1024.0,
        }
    workload = None
    if "--workload" in sys.argv:
        workload = json.loads(sys.argv[sys.argv.index("--workload") + 1])
    json.dump({"note": note, "workload": workload, "correction": "bytes = 2*FETCH_SIZE_KiB*1024 + WRITE_SIZE_KiB*1024", "kernels": kernels},
              open(out, "w"), indent=1)
    for k, v in kernels.items():
        print(f"{k:48s} n={v['launches']:5d} bytes/launch={v['hbm_bytes_per_launch_corrected']:.4g}")


if __name__ == "__main__":
    main()

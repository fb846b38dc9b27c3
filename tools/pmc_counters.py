#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (any SQ / GRBM / TCC counters) per kernel.

Usage: pmc_counters.py OUT.json DIR [DIR ...] [--note TEXT] [--cus N]

Each DIR is the `-d` directory of one counter pass over the same command.  For every kernel the counters
are averaged per launch.  Derived figures (MI355X_MICROARCH.md, "rocprofv3 PMC slots" / "DVFS give-back"):

  clock_GHz        GRBM_GUI_ACTIVE / 8 / duration      (the counter sums the 8 XCDs)
  mfma_util        SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * CUs * 4 SIMDs)
                   busy cycles of the matrix pipes over the pipe-cycles the launch had
  mfma_util_busy   the same over SQ_BUSY_CYCLES-equivalent CU time when GRBM is absent
  wait_frac        SQ_WAIT_ANY / SQ_WAVE_CYCLES         (waves parked on s_waitcnt / barrier)
  issue_stall_frac SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES
  active_frac      SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES
  lds_conflict     SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
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


def collect(d: str, acc: dict):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                k = short(row["Kernel_Name"])
                c = row["Counter_Name"]
                a = acc.setdefault(k, {}).setdefault(c, [0.0, 0, 0.0])
                a[0] += float(row["Counter_Value"])
                a[1] += 1
                a[2] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-9


def main():
    argv = sys.argv[1:]
    note, cus = "", 256
    if "--note" in argv:
        i = argv.index("--note")
        note = argv[i + 1]
        del argv[i:i + 2]
    if "--cus" in argv:
        i = argv.index("--cus")
        cus = int(argv[i + 1])
        del argv[i:i + 2]
    out, dirs = argv[0], argv[1:]
    acc = {}
    for d in dirs:
        collect(d, acc)
    kernels = {}
    for k, cs in sorted(acc.items()):
        e = {"launches": max(v[1] for v in cs.values())}
        avg = {c: v[0] / max(v[1], 1) for c, v in cs.items()}
        dur = {c: v[2] / max(v[1], 1) for c, v in cs.items()}
        e["counters_avg_per_launch"] = avg
        anyc = next(iter(dur))
        e["avg_ms_under_pmc"] = dur[anyc] * 1e3
        g = avg.get("GRBM_GUI_ACTIVE")
        if g:
            e["clock_GHz"] = g / 8.0 / dur["GRBM_GUI_ACTIVE"] / 1e9
        mb = avg.get("SQ_VALU_MFMA_BUSY_CYCLES")
        if mb is not None and g:
            e["mfma_util"] = mb / (g / 8.0 * cus * 4)
        bc = avg.get("SQ_BUSY_CYCLES")
        if mb is not None and bc:
            e["mfma_busy_over_sq_busy"] = mb / bc
        wc = avg.get("SQ_WAVE_CYCLES")
        if wc:
            for name, c in (("wait_frac", "SQ_WAIT_ANY"), ("issue_stall_frac", "SQ_WAIT_INST_ANY"),
                            ("active_frac", "SQ_ACTIVE_INST_ANY"), ("lds_issue_stall_frac", "SQ_WAIT_INST_LDS")):
                if c in avg:
                    e[name] = avg[c] / wc
        if avg.get("SQ_LDS_IDX_ACTIVE"):
            e["lds_conflict"] = avg.get("SQ_LDS_BANK_CONFLICT", 0.0) / avg["SQ_LDS_IDX_ACTIVE"]
        kernels[k] = e
    json.dump({"note": note, "cus": cus, "kernels": kernels}, open(out, "w"), indent=1)
    for k, v in kernels.items():
        line = f"{k:52s} n={v['launches']:4d} {v['avg_ms_under_pmc']:8.3f} ms"
        for f in ("clock_GHz", "mfma_util", "wait_frac", "issue_stall_frac", "active_frac", "lds_conflict"):
            if f in v:
                line += f" {f}={v[f]:.3f}"
        print(line)


if __name__ == "__main__":
    main()

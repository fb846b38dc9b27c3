#!/usr/bin/env python3
"""Kernel timeline of ONE search step from a rocprofv3 --kernel-trace database (rocpd sqlite):
start offset, duration, grid and kernel name of every launch between the last two k_coarse_init launches.
Usage: tools/step_timeline.py <results.db> [marker-kernel-substring]"""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    marker = sys.argv[2] if len(sys.argv) > 2 else "k_coarse_init"
    rows = list(db.execute("select name, start, end, grid_x, workgroup_x from kernels order by start"))
    idx = [i for i, r in enumerate(rows) if marker in r[0]]
    if len(idx) < 3:
        raise SystemExit(f"fewer than 3 launches of {marker}")
    a, b = idx[-3], idx[-2]
    t0 = rows[a][1]
    for r in rows[a:b + 1]:
        m = re.search(r"(k_\w+(<[^>]*>)?)", r[0])
        print(f"{(r[1] - t0) / 1e3:9.1f} us  dur {(r[2] - r[1]) / 1e3:8.1f} us  blocks {r[3] // max(r[4], 1):6d}  "
              f"{m.group(1) if m else r[0][:60]}")
    print(f"step: {(rows[b][1] - t0) / 1e3:.1f} us")


if __name__ == "__main__":
    main()

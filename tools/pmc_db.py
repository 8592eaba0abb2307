#!/usr/bin/env python3
"""Fold rocprofv3 counter passes that were written as rocpd sqlite databases (rocprofv3's default output) into a per-kernel table.

usage: pmc_db.py <dir-or-db> [<dir-or-db> ...]      (one pass per counter set: FETCH_SIZE, WRITE_SIZE, TCC_HIT_sum TCC_MISS_sum ...)
FETCH_SIZE / WRITE_SIZE are KB per dispatch.  On gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes for wide coalesced
reads (MI355X_MICROARCH.md, HBM section), so the "fetch x2" column doubles it; both counters sit on the L2's memory-side
(fabric) ports, i.e. Infinity-Cache hits are included -- they measure what leaves the XCD, not what reaches HBM.
"""
import collections
import glob
import os
import re
import sqlite3
import sys


def fold(path, acc):
    dbs = [path] if path.endswith(".db") else glob.glob(os.path.join(path, "**", "*.db"), recursive=True)
    for db in dbs:
        cur = sqlite3.connect(db).cursor()
        cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
        ki, ci, vi = cols.index("kernel_name"), cols.index("counter_name"), cols.index("value")
        for r in cur.execute("select * from counters_collection"):
            k = re.sub(r"^void ", "", re.sub(r"\(.*", "", r[ki]))
            acc[k][r[ci]].append(float(r[vi]))


def main():
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for p in sys.argv[1:]:
        fold(p, acc)
    mean = lambda x: sum(x) / len(x) if x else float("nan")
    print("%-46s %5s %11s %11s %8s" % ("kernel", "n", "fetch x2 GB", "write GB", "L2 hit"))
    for k in sorted(acc):
        c = acc[k]
        if k.startswith("__amd"):
            continue
        f, w = mean(c.get("FETCH_SIZE", [])), mean(c.get("WRITE_SIZE", []))
        h, m = mean(c.get("TCC_HIT_sum", [])), mean(c.get("TCC_MISS_sum", []))
        n = max(len(v) for v in c.values())
        print("%-46s %5d %11.3f %11.3f %8.3f" % (k[:46], n, 2 * f * 1024 / 1e9, w * 1024 / 1e9, h / (h + m) if h == h else float("nan")))


if __name__ == "__main__":
    main()

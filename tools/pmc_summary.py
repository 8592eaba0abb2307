#!/usr/bin/env python3
"""Fold two rocprofv3 counter passes (--pmc FETCH_SIZE, --pmc WRITE_SIZE; separate runs of the same command, as
MI355X_MICROARCH.md's HBM section prescribes) into the per-kernel HBM-traffic summary bench.py reads from profiles/.

usage: pmc_summary.py <fetch_dir> <write_dir> <out.json> "<command that was profiled>" frames frames_per_lane N E
gfx950 correction: FETCH_SIZE counts half of the coalesced streaming reads -> reads = 2 x FETCH_SIZE (KB); WRITE_SIZE is exact.
Only this repo's kernels (names starting q[kehip]_ after the return type / template prefix) are kept.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def fold(directory, counter):
    acc = defaultdict(lambda: [0, 0.0])
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            if row["Counter_Name"] != counter:
                continue
            m = re.search(r"\b(q[kehip]_[A-Za-z0-9_]+(?:<[^(]*>)?)\(", row["Kernel_Name"])
            if not m:
                continue
            a = acc[m.group(1)]
            a[0] += 1
            a[1] += float(row["Counter_Value"])
    return acc


def main():
    fetch_dir, write_dir, out, cmd, frames, fpl, N, E = sys.argv[1:9]
    f, w = fold(fetch_dir, "FETCH_SIZE"), fold(write_dir, "WRITE_SIZE")
    kernels = {}
    for name in sorted(set(f) | set(w)):
        nf, sf = f.get(name, [0, 0.0])
        nw, sw = w.get(name, [0, 0.0])
        fk, wk = (sf / nf if nf else 0.0), (sw / nw if nw else 0.0)
        kernels[name] = dict(launches=max(nf, nw), FETCH_SIZE_KB=fk, WRITE_SIZE_KB=wk, hbm_bytes_corrected=(2.0 * fk + wk) * 1024.0)
    json.dump(dict(source="rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes -- " + cmd,
                   units="KB per launch as reported by rocprofv3",
                   correction="gfx950: FETCH_SIZE reports 1/2 of coalesced streaming reads (MI355X_MICROARCH.md, HBM section), so reads = 2 x FETCH_SIZE; WRITE_SIZE is exact",
                   workload=dict(frames=int(frames), frames_per_lane=int(fpl), N=int(N), E=int(E)), kernels=kernels), open(out, "w"), indent=1)
    for k, v in kernels.items():
        print("%-60s x%-4d %10.1f MB per launch" % (k, v["launches"], v["hbm_bytes_corrected"] / 1e6))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""rocprofv3 kernel trace of tools/bench_engine_overlap.py -> per concurrency level (number of queues launching within +-0.5 ms):
mean duration of each decode-step kernel and mean gap (end -> next start) on its queue.  Answers: when engines overlap, do the
kernels get longer or the gaps between them?

usage: trace_concurrency.py <dir with *kernel_trace.csv>"""
import bisect
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*", "", name)
    m = re.search(r"(k_[a-z0-9_]+)", name)
    return m.group(1) if m else name[:32]


def main():
    rows = []
    for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r.get("Kernel_Name", "")), r.get("Queue_Id", "")))
    rows.sort()
    rows = [r for r in rows if r[2].startswith("k_dstep") or r[2] in ("k_resid_ln_rows", "k_ln_rows")]
    starts = [r[0] for r in rows]
    half = 500_000
    # concurrency level of a kernel = distinct queues with a launch inside +-0.5 ms (sampled)
    lvl = []
    for i, (s, e, k, q) in enumerate(rows):
        lo, hi = bisect.bisect_left(starts, s - half), bisect.bisect_right(starts, s + half)
        stepn = max(1, (hi - lo) // 48)
        lvl.append(len({rows[j][3] for j in range(lo, hi, stepn)}))
    dur = defaultdict(lambda: defaultdict(list))
    gap = defaultdict(lambda: defaultdict(list))
    last = {}
    for (s, e, k, q), L in zip(rows, lvl):
        dur[L][k].append(e - s)
        if q in last:
            pe, pk, pL = last[q]
            if 0 <= s - pe < 100_000 and pL == L:
                gap[L][pk + "->" + k].append(s - pe)
        last[q] = (e, k, L)
    for L in sorted(dur):
        n = sum(len(v) for v in dur[L].values())
        if n < 500:
            continue
        print(f"== {L} queue(s) active: {n} step kernels")
        tot_d = tot_g = 0.0
        for k, v in sorted(dur[L].items(), key=lambda kv: -sum(kv[1])):
            v.sort()
            print(f"   {k:22s} n={len(v):7d}  mean {sum(v)/len(v)/1e3:7.2f} us  median {v[len(v)//2]/1e3:7.2f}  p90 {v[int(len(v)*0.9)]/1e3:7.2f}")
            tot_d += sum(v)
        for k, v in sorted(gap[L].items(), key=lambda kv: -sum(kv[1])):
            if len(v) < 100:
                continue
            v.sort()
            print(f"   gap {k:40s} n={len(v):7d}  mean {sum(v)/len(v)/1e3:6.2f} us  median {v[len(v)//2]/1e3:6.2f}  p90 {v[int(len(v)*0.9)]/1e3:6.2f}")
            tot_g += sum(v)
        print(f"   kernel time {tot_d/1e6:.1f} ms, gap time {tot_g/1e6:.1f} ms  (gap share of a queue's timeline {100*tot_g/(tot_d+tot_g):.1f} %)")


if __name__ == "__main__":
    main()

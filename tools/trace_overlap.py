#!/usr/bin/env python3
"""Timeline view of a rocprofv3 kernel trace of bench.py: how much of the decode stage's wall time has a prefill kernel
running, how much a decode-step kernel, how much both, how much neither -- and the average number of kernels in flight.

usage: trace_overlap.py <dir with *kernel_trace.csv>     (run on the GPU box right after the trace: the CSV is not kept)
Only the graph-replayed (timed) step is looked at: the window from the first to the last k_dattn launch of the densest
half of the trace."""
import csv
import glob
import os
import re
import sys

PREFILL = ("k_linear", "k_attn", "k_ln_rows", "k_dembed", "k_dgemm_s<true, 3>", "k_gather_rows", "k_dargmax")
STEP = ("k_dattn", "k_dstep_attn_down", "k_dstep_qkv_up", "k_dgemm_s<true, 5>", "k_resid_ln_rows", "k_dstep_head")


def short(name):
    name = re.sub(r"\(.*", "", name)
    m = re.search(r"(k_[a-z0-9_]+)(?:<([^>]*)>)?", name)
    return (m.group(1) + (f"<{m.group(2)}>" if m.group(2) else "")) if m else name[:40]


def union(iv):
    iv.sort()
    tot, cur_s, cur_e = 0, None, None
    out = []
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                out.append((cur_s, cur_e)); tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        out.append((cur_s, cur_e)); tot += cur_e - cur_s
    return tot, out


def inter(a, b):
    i = j = 0; tot = 0
    while i < len(a) and j < len(b):
        s, e = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if s < e:
            tot += e - s
        if a[i][1] < b[j][1]:
            i += 1
        else:
            j += 1
    return tot


def main():
    rows = []
    for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r.get("Kernel_Name", "")), r.get("Queue_Id", "")))
    rows.sort()
    # decoder kernels only, inside the decode stage of the LAST step of the run (the eager event pass runs engines one at a
    # time, so take the window in which >= 3 queues launch k_dattn: the graph-replayed step)
    rows = [(a_, b_, (k[:k.index("<")] if k.startswith("k_dstep_attn_down<") else k), q) for a_, b_, k, q in rows]     # k_dstep_attn_down<NW> -> k_dstep_attn_down
    anchor = "k_dstep_attn_down" if any(k == "k_dstep_attn_down" for _, _, k, _ in rows) else "k_dattn"
    datt = [(s, e, q) for s, e, k, q in rows if k == anchor]
    t0, t1 = datt[0][0], datt[-1][1]
    # find the longest span where four distinct queues are active within any 2 ms window
    import bisect
    starts = [s for s, _, _ in datt]
    best = None
    step = max(1, len(datt) // 4000)
    win = 2_000_000
    flags = []
    for i in range(0, len(datt), step):
        j = bisect.bisect_right(starts, datt[i][0] + win)
        flags.append((datt[i][0], len({q for _, _, q in datt[i:j:max(1, (j - i) // 64)]})))
    span_s = next((t for t, n in flags if n >= 3), t0)
    span_e = next((t for t, n in reversed(flags) if n >= 3), t1)
    sel = [(s, e, k) for s, e, k, _ in rows if s >= span_s and e <= span_e]
    pre = [(s, e) for s, e, k in sel if any(k.startswith(p) for p in PREFILL) and k != "k_dgemm_s<true, 5>"]
    stp = [(s, e) for s, e, k in sel if k in STEP]
    wall = span_e - span_s
    up, ivp = union(pre)
    us, ivs = union(stp)
    both = inter(ivp, ivs)
    ua, _ = union(pre + stp)
    busy = sum(e - s for s, e in pre + stp)
    print(f"window {wall/1e6:.1f} ms, {len(sel)} kernels; average kernels in flight {busy/wall:.2f}")
    print(f"some prefill kernel running {100*up/wall:.1f} %   some step kernel running {100*us/wall:.1f} %   both {100*both/wall:.1f} %   neither {100*(wall-ua)/wall:.1f} %")
    print(f"prefill kernel time {sum(e-s for s,e in pre)/1e6:.1f} ms   step kernel time {sum(e-s for s,e in stp)/1e6:.1f} ms")
    # gaps between consecutive step kernels of one queue (end -> next start, host stalls > 50 us excluded): with four
    # engines submitting vs the window before/after in which a single queue runs (the event pass)
    def gaps(lo, hi):
        byq = {}
        for s_, e_, k, q in rows:
            if s_ >= lo and e_ <= hi and k in STEP:
                byq.setdefault(q, []).append((s_, e_, k))
        g, d = [], []
        for q, lst in byq.items():
            for (s0, e0, k0), (s1, e1, k1) in zip(lst[:-1], lst[1:]):
                if 0 <= s1 - e0 < 50_000:
                    g.append(s1 - e0)
                d.append(e0 - s0)
        g.sort(); d.sort()
        return (sum(g) / max(len(g), 1), g[len(g) // 2] if g else 0, sum(d) / max(len(d), 1), len(g))
    g4 = gaps(span_s, span_e)
    g1 = gaps(span_e, rows[-1][1])
    print(f"step-kernel gap on one queue: four engines mean {g4[0]/1e3:.2f} us (median {g4[1]/1e3:.2f}), mean kernel {g4[2]/1e3:.2f} us, n={g4[3]}")
    print(f"                              single engine (event pass) mean {g1[0]/1e3:.2f} us (median {g1[1]/1e3:.2f}), mean kernel {g1[2]/1e3:.2f} us, n={g1[3]}")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel trace + PMC passes) into per-kernel tables.

usage: summarize_profile.py <dir with prof_trace/ prof_fetch/ prof_write/>
Prints (a) per-kernel count / total / average duration from the kernel trace, (b) per-kernel HBM bytes per launch from
FETCH_SIZE / WRITE_SIZE with the gfx950 corrections of MI355X_MICROARCH.md (FETCH_SIZE counts 64 B per 128-B request
on wide coalesced streams -> x2; both counters are in KiB), and writes traffic.json next to it."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    # rocprofv3 leaves names with _Float16 pointer arguments mangled (its demangler does not know DF16_): _Z<len><name>... -> <name>; template instances of such
    # kernels (_Z<len><name>I...E) lose their arguments here, which only merges instances
    m = re.match(r"_Z(\d+)", name)
    if m:
        n = int(m.group(1)); st = m.end()
        name = name[st:st + n]
    name = re.sub(r"\(.*", "", name)
    m = re.search(r"(k_[a-z0-9_]+)(?:<([^>]*)>)?", name)
    if m:
        return m.group(1) + (f"<{m.group(2)}>" if m.group(2) else "")
    return name[:60]


def kernel_trace(d):
    rows = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                k = short(r.get("Kernel_Name", ""))
                dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
                rows[k][0] += 1
                rows[k][1] += dur
    return rows


def union_of(d, kernel_prefix):
    """(launches, sum of durations, union of the dispatch intervals) in seconds of one kernel over every queue of the trace: concurrent engines' launches overlap, and
    what the memory system delivers is bytes / the time during which ANY of them ran (bench.py: roofline.frac), not a per-launch figure"""
    iv = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if short(r.get("Kernel_Name", "")).startswith(kernel_prefix):
                    iv.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    if not iv:
        return 0, 0.0, 0.0
    iv.sort()
    tot = 0; cs, ce = iv[0]
    for a, b in iv[1:]:
        if a > ce:
            tot += ce - cs; cs, ce = a, b
        elif b > ce:
            ce = b
    tot += ce - cs
    return len(iv), sum(b - a for a, b in iv) * 1e-9, tot * 1e-9


def pmc(d, counter):
    rows = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r.get("Counter_Name") != counter:
                    continue
                k = short(r.get("Kernel_Name", ""))
                rows[k][0] += 1
                rows[k][1] += float(r["Counter_Value"])
    return rows


def pmc_steady(d, counter, kernel_prefix, skip_frac=0.5):
    """per-launch mean of `counter` over the LAST (1 - skip_frac) of the dispatches of one kernel, in dispatch order: the PMC passes decode the first 8 bars
    of every job, and from bar 4 on every prompt sits at generate()'s 512-token truncation -- the second half of the attention launches runs at the
    contexts of the full run's steady state (96 % of a job's bars)"""
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r.get("Counter_Name") == counter and short(r.get("Kernel_Name", "")).startswith(kernel_prefix):
                    rows.append((int(r.get("Dispatch_Id", len(rows))), float(r["Counter_Value"])))
    rows.sort()
    tail = rows[int(len(rows) * skip_frac):]
    return (len(tail), sum(v for _, v in tail) / max(1, len(tail)))


def main():
    root = sys.argv[1]
    kt = kernel_trace(os.path.join(root, "prof_trace"))
    tot = sum(v[1] for v in kt.values()) or 1.0
    print("== kernel trace (prof_trace): kernel, launches, total_ms, avg_us, share")
    for k, (n, us) in sorted(kt.items(), key=lambda kv: -kv[1][1])[:30]:
        print(f"{k:40s} {n:9d} {us/1e3:12.3f} {us/n:10.2f} {100*us/tot:6.2f}%")
    try:
        n_u, sum_u, uni_u = union_of(os.path.join(root, "prof_trace"), "k_dstep_attn_down<8, false, true>")
        if n_u:
            line = json.load(open(os.path.join(root, "prof_trace.json")))
            apl = line["roofline"]["alg_bytes_per_launch"]
            print(f"== k_dstep_attn_down<8, false, true> in the kernel trace (whole run, every queue): {n_u} launches, sum of durations {sum_u:.3f} s, UNION of the dispatch intervals {uni_u:.3f} s "
                  f"(x {sum_u / uni_u:.2f} overlap); at the stamped launches' {apl / 1e6:.1f} MB per launch: per launch {apl * n_u / sum_u / 8e12:.3f} of 8 TB/s, on the union {apl * n_u / uni_u / 8e12:.3f} "
                  "(the ramp-up bars of every stage read less than that per launch: the bench's own figures count bytes exactly)")
        # every kernel of the run: the time during which ANY kernel was running against the span from the first dispatch to the last (idle = host gaps + dependency bubbles)
        n_a, sum_a, uni_a = union_of(os.path.join(root, "prof_trace"), "")
        lo, hi = None, None
        for f in glob.glob(os.path.join(root, "prof_trace", "**", "*kernel_trace.csv"), recursive=True):
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    a_, b_ = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
                    lo = a_ if lo is None or a_ < lo else lo; hi = b_ if hi is None or b_ > hi else hi
        # the decoder's kernels alone (steps + batched prefill of every engine): their union against the decode stages' wall time of the same run (the JSON line)
        dec_pref = ("k_dstep", "k_dmlp", "k_pqkv", "k_pattn", "k_resid_ln", "k_dembed", "k_dgemm", "k_dattn", "k_dargmax", "k_gather_rows", "k_ln_rows", "k_linear<1")
        ivd = []
        for f in glob.glob(os.path.join(root, "prof_trace", "**", "*kernel_trace.csv"), recursive=True):
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    if short(r.get("Kernel_Name", "")).startswith(dec_pref):
                        ivd.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
        if ivd:
            ivd.sort(); tot_d = 0; cs, ce = ivd[0]
            for a_, b_ in ivd[1:]:
                if a_ > ce:
                    tot_d += ce - cs; cs, ce = a_, b_
                elif b_ > ce:
                    ce = b_
            tot_d += ce - cs
            print(f"== decoder kernels (decode steps + batched prefill, every engine): {len(ivd)} launches, UNION {tot_d * 1e-9:.3f} s (some decoder kernel running); the run holds one warm-up, "
                  "one timed and one stamped decode stage + the 4-bar serial pass: compare with ~3.05 x roofline.decode_stage.stage_s_per_step of the line")
        if n_a and hi:
            print(f"== every kernel of the trace: {n_a} launches, sum of durations {sum_a:.3f} s, UNION {uni_a:.3f} s of the {(hi - lo) * 1e-9:.3f} s between the first dispatch and the last "
                  f"(some kernel running {100 * uni_a / ((hi - lo) * 1e-9):.1f} % of that span; the span includes the host-side set-up between the stages)")
    except Exception as e:      # noqa: BLE001
        print("(no union of the attention launches:", e, ")")
    fe = pmc(os.path.join(root, "prof_fetch"), "FETCH_SIZE")
    wr = pmc(os.path.join(root, "prof_write"), "WRITE_SIZE")
    traffic = {}
    print("== HBM traffic per launch (PMC, separate passes): kernel, launches, fetch_MB(x2 corrected), write_MB, total_MB")
    for k in sorted(set(fe) | set(wr), key=lambda k: -(fe.get(k, [0, 0])[1] + wr.get(k, [0, 0])[1])):
        nf, f = fe.get(k, [0, 0.0])
        nw, w = wr.get(k, [0, 0.0])
        fb = 2.0 * f * 1024 / max(nf, 1)
        wb = w * 1024 / max(nw, 1)
        traffic[k] = {"fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb, "bytes_per_launch": fb + wb, "launches": max(nf, nw)}
        print(f"{k:40s} {max(nf,nw):9d} {fb/1e6:12.3f} {wb/1e6:12.3f} {(fb+wb)/1e6:12.3f}")
    with open(os.path.join(root, "traffic.json"), "w") as fh:
        json.dump(traffic, fh, indent=1)
    # the same table under the names bench.py's launch profiler uses (template instances merged, launch-weighted):
    # profiles/traffic.json, read by bench.py for `roofline.traffic`
    groups = {"k_linear": ("k_linear<0>", "k_linear<1>"), "k_linear_ln": ("k_linear<2>",), "k_linear_dec": ("k_linear<10>", "k_linear<11>", "k_linear<12>", "k_linear<14>"),
              "k_dgemm_s": ("k_dgemm_s<true, 5>", "k_dgemm_s<true, 3>"), "k_dstep_attn_down": ("k_dstep_attn_down<4>", "k_dstep_attn_down<8>", "k_dstep_attn_down<16>"),
              "k_proj256": ("k_proj256<true>", "k_proj256<false>")}
    bench = {}
    for k, v in traffic.items():
        name = next((g for g, members in groups.items() if k in members), k)
        if k.startswith("k_dstep_attn_down<"):      # <waves per workgroup, row finish>: one kernel for the bench's profiler
            name = "k_dstep_attn_down"
        name = re.sub(r"<.*", "", name)              # any other template instance goes under its kernel's name
        b = bench.setdefault(name, [0.0, 0])
        b[0] += v["bytes_per_launch"] * v["launches"]; b[1] += v["launches"]
    out = {k: b[0] / max(b[1], 1) for k, b in bench.items() if k and not k.startswith("__amd")}
    out["_note"] = ("HBM bytes per launch (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, separate rocprofv3 --pmc passes) of `bench.py --steps 1 --warmup 0 --max-bars 8` "
                    "(the headline's batch and engine layout, first 8 bars of every job: same rows per launch and, from bar 4 on, the same contexts as the full run); see the round's profile_summary.txt")
    # the algorithmic bytes of the SAME launches, from the PMC run's own bench line (exact counters of the library)
    try:
        line = json.load(open(os.path.join(root, "prof_fetch.json")))
        n_attn = sum(v["launches"] for k, v in traffic.items() if k.startswith("k_dstep_attn_down<")) or 1      # both attention forms: the line's bytes cover every step
        ds = line["roofline"]["decode_stage"]["alg_bytes_per_step"]
        steps = n_attn / 8.0
        w = 50.3e6
        alg = (ds - steps * w) / n_attn + (2048 + 512) * 512 * 2
        out["_alg_bytes_per_launch_k_dstep_attn_down_in_the_pmc_run"] = alg
        print(f"== k_dstep_attn_down in the PMC run: algorithmic {alg/1e6:.1f} MB per launch (K+V of every row's context + down / dense weights) vs measured {out.get('k_dstep_attn_down', 0)/1e6:.1f} MB")
    except Exception as e:      # noqa: BLE001
        print("(no algorithmic byte count for the PMC run:", e, ")")
    # the headline's attention form at the steady-state contexts: what bench.py prints as roofline.traffic next to the stamped launches' algorithmic bytes
    try:
        form = "k_dstep_attn_down<8, false, true>"
        nf, f = pmc_steady(os.path.join(root, "prof_fetch"), "FETCH_SIZE", form)
        nw, w = pmc_steady(os.path.join(root, "prof_write"), "WRITE_SIZE", form)
        if nf and nw:
            cfgl = {}
            try:
                cfgl = json.load(open(os.path.join(root, "prof_fetch.json")))
            except Exception:      # noqa: BLE001
                pass
            out["k_dstep_attn_down_steady"] = {"form": form, "launches": min(nf, nw), "fetch_bytes_per_launch": 2.0 * f * 1024, "write_bytes_per_launch": w * 1024,
                                               "bytes_per_launch": 2.0 * f * 1024 + w * 1024,
                                               "rows_per_launch": cfgl.get("config", {}).get("decoder_streams_per_engine"), "engines": cfgl.get("config", {}).get("decoder_engines"),
                                               "build_id": cfgl.get("build_id"),
                                               "what": "second half of the form's launches in dispatch order = bars 4..7 of the 8-bar PMC passes: prompts at the 512-token truncation, "
                                                       "the contexts of the stamped launches"}
            print(f"== {form}, steady-state half of the PMC run: {nf} launches, {(2.0 * f * 1024 + w * 1024) / 1e6:.1f} MB per launch")
    except Exception as e:      # noqa: BLE001
        print("(no steady-state traffic:", e, ")")
    with open(os.path.join(root, "traffic_bench.json"), "w") as fh:
        json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()

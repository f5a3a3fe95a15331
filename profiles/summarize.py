#!/usr/bin/env python3
"""Condenses rocprofv3 CSV output (kernel_stats / counter_collection) into the
small per-kernel summaries committed under profiles/.

usage: summarize.py stats <kernel_stats.csv> | trace <kernel_trace.csv> | pmc <counter_collection.csv>
                    | sq <counter_collection.csv>   (any other counters: raw per-dispatch averages)
                    | pc <kernel_trace.csv>         (the pair-count scan's launches of the benchmark size only)
PMC note (guide: /opt/skills/guides/MI355X_MICROARCH.md, HBM): FETCH_SIZE and
WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports half the bytes of a wide
coalesced read, so the corrected read traffic is 2 x FETCH_SIZE x 1024.
"""
import collections
import csv
import re
import sys


def short(name):
    m = re.search(r"(k_\w+)(<\w+>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else None


def main():
    mode, path = sys.argv[1], sys.argv[2]
    if mode == "stats":
        print("kernel,calls,total_ms,avg_us,min_us,max_us,pct")
        for r in csv.DictReader(open(path)):
            k = short(r["Name"]) or r["Name"].split("(")[0][-60:]
            print("%s,%s,%.3f,%.2f,%.2f,%.2f,%s" % (k, r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                     float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3,
                                                     float(r["MaxNs"]) / 1e3, r["Percentage"]))
    elif mode == "trace":
        # per-dispatch durations: every sequence launches all of its kernels and the device decides
        # which of them work, so most dispatches return at once; "working" = at least 20 us
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(path)):
            k = short(r["Kernel_Name"])
            if k:
                agg[k].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
        print("kernel,dispatches,total_ms,working,working_total_ms,working_avg_us,working_max_us")
        for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            w = [x for x in v if x >= 20.0]
            print("%s,%d,%.3f,%d,%.3f,%.2f,%.2f" % (k, len(v), sum(v) / 1e3, len(w), sum(w) / 1e3,
                                                  sum(w) / max(len(w), 1), max(v)))
    elif mode == "pc":
        # The pair-count scan alone.  A bench.py process launches it on the benchmark corpus (inside every training and
        # in the timed groups at the end) AND on small inputs (published_workload: 1.1 MB, ~40 us): an average over
        # all calls -- what rocprofv3's kernel_stats gives -- is diluted by the small ones (round 3's "0.525").  Only
        # launches of at least half the longest one count here.
        v = []
        for r in csv.DictReader(open(path)):
            k = short(r["Kernel_Name"])
            if k and k.startswith("k_pair_count"):
                v.append((float(r["Start_Timestamp"]), (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3))
        v.sort()
        d = [x for _, x in v]
        full = [x for x in d if x >= 0.5 * max(d)]
        last30 = full[-30:]
        print("k_pair_count launches,%d" % len(d))
        print("launches of the benchmark size (>= half the longest),%d" % len(full))
        print("their mean_us,%.2f" % (sum(full) / len(full)))
        print("their min_us,%.2f" % min(full))
        print("their max_us,%.2f" % max(full))
        print("mean_us of the last 30 of them (bench.py's timed groups),%.2f" % (sum(last30) / len(last30)))
        print("of the 8 TB/s peak at 2^32 bytes (last 30),%.4f" % (4294967296.0 / (sum(last30) / len(last30) * 1e-6) / 8e12))
    elif mode == "trace20":
        # the driver's timed region: the k_fused_batch working dispatches number 6..25 (5 warm-up sequences,
        # 20 timed ones; the first passes of a run are fused from the second sequence on) and everything between
        rows = [r for r in csv.DictReader(open(path)) if short(r["Kernel_Name"])]
        rows.sort(key=lambda r: float(r["Start_Timestamp"]))
        fused = [i for i, r in enumerate(rows) if short(r["Kernel_Name"]).startswith("k_fused_batch")
                 and float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) >= 20e3]
        print("k_fused_batch working dispatch,duration_us")
        for n, i in enumerate(fused[:40]):
            print("%d,%.2f" % (n, (float(rows[i]["End_Timestamp"]) - float(rows[i]["Start_Timestamp"])) / 1e3))
    elif mode == "sq":
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(path)):
            k = short(r["Kernel_Name"])
            if k:
                agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
        print("kernel,counter,dispatches,working,avg,min,max")
        for (k, c), v in sorted(agg.items()):
            w = [x for x in v if x >= 0.01 * max(v)] if max(v) > 0 else v
            print("%s,%s,%d,%d,%.0f,%.0f,%.0f" % (k, c, len(v), len(w), sum(w) / len(w), min(w), max(w)))
    else:
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(path)):
            k = short(r["Kernel_Name"])
            if k:
                agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
        # Most launches of the batch-sequence kernels return at once (every sequence launches all of
        # them and the device decides which one works): averages are over the WORKING dispatches,
        # those that moved at least 1 % of the largest one.
        print("kernel,counter,dispatches,working,avg_KB,min_KB,max_KB,avg_bytes_raw,avg_bytes_corrected")
        for (k, c), v in sorted(agg.items()):
            w = [x for x in v if x >= 0.01 * max(v)] if max(v) > 0 else v
            avg = sum(w) / len(w)
            corr = avg * 1024 * (2 if c == "FETCH_SIZE" else 1)
            print("%s,%s,%d,%d,%.1f,%.1f,%.1f,%.0f,%.0f" % (k, c, len(v), len(w), avg, min(w), max(w), avg * 1024, corr))


if __name__ == "__main__":
    main()

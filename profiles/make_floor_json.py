"""profiles/r03_fused_floor.json, read by bench.py's floor_model: the two floors of a fused pass from measurements.

  python3 profiles/make_floor_json.py --diag-log gpurun_out/r03/fused_diag.log \
          --counters <counter_collection.csv> --trace <kernel_trace.csv> > profiles/r03_fused_floor.json

  --diag-log   output of tools/fused_diag.py (a -DMBPE_DIAG build): the line "diag 4 ..." is the copy-only build of
               k_fused_batch (load tile + summaries, store tile) on 4.29e9 slots = 17.18 GB
  --counters / --trace   rocprofv3 --pmc SQ_BUSY_CYCLES SQ_INSTS_VALU ... --kernel-trace of ONE bench.py run: the
               k_fused_batch dispatch with the most vector instructions (a full-size pass: 8,388,608 tiles) gives the
               instructions per tile and, with its own duration, the shader clock (SQ_BUSY_CYCLES is summed over the
               32 shader engines)
"""
import argparse, collections, csv, json, re

ap = argparse.ArgumentParser()
ap.add_argument("--diag-log", required=True)
ap.add_argument("--counters", required=True)
ap.add_argument("--trace", required=True)
ap.add_argument("--slots", type=float, default=4294967296.0)
a = ap.parse_args()
copy_ms = None
ladder = {}
matches = None
for line in open(a.diag_log):
    m = re.match(r"diag (\d) .*avg ms ([0-9.]+)", line)
    if m:
        ladder[int(m.group(1))] = float(m.group(2))
    m = re.match(r"diag 0 .*matches per launch ([0-9.]+)", line)
    if m:
        matches = float(m.group(1))
copy_ms = ladder.get(4)
by = collections.defaultdict(dict)
for r in csv.DictReader(open(a.counters)):
    if "k_fused_batch" in r["Kernel_Name"]:
        by[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
did, c = max(by.items(), key=lambda kv: kv[1].get("SQ_INSTS_VALU", 0.0))
dur_ms = None
for r in csv.DictReader(open(a.trace)):
    if r["Dispatch_Id"] == did:
        dur_ms = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e6
tiles = a.slots / 512.0
clock = c["SQ_BUSY_CYCLES"] / 32.0 / (dur_ms * 1e-3) / 1e9
print(json.dumps({
    "copy_only_GBps": 4.0 * a.slots / (copy_ms * 1e-3) / 1e9,
    "copy_only_ms": copy_ms,
    # the same four passes (sequences 7..10 of the benchmark workload) with timing-only builds of the kernel
    "ladder_ms": {"copy": ladder.get(4), "copy_lookups": ladder.get(3), "all_but_count_deltas": ladder.get(2), "shipped": ladder.get(0)},
    "ladder_matches_per_launch": matches,
    "delta_ps_per_match": (ladder[0] - ladder[2]) * 1e9 / matches if matches and 0 in ladder and 2 in ladder else None,
    "valu_per_tile": c["SQ_INSTS_VALU"] / tiles,
    "valu_per_launch": c["SQ_INSTS_VALU"],
    "salu_per_launch": c.get("SQ_INSTS_SALU"),
    "profiled_launch_ms": dur_ms,
    "clock_GHz": clock,
    "valu_busy_frac": c["SQ_INSTS_VALU"] * 4.0 / (c["SQ_BUSY_CYCLES"] / 32.0 * 1024.0),
    "simds": 1024,
    "sources": "tools/fused_diag.py (MBPE_FUSED_DIAG=4) + rocprofv3 --pmc SQ_BUSY_CYCLES SQ_INSTS_VALU --kernel-trace of one bench.py run, collected by profiles/collect_r0N.sh",
}, indent=1))

"""profiles/r03_fused_floor.json, read by bench.py's floor_model: the two floors of a fused pass from measurements.

  python3 profiles/make_floor_json.py --diag-log gpurun_out/r03/fused_diag.log --sq profiles/r03_pmc_sq.csv \
          --kernel-ms 8.41 > profiles/r03_fused_floor.json

  --diag-log   output of tools/fused_diag.py (a -DMBPE_DIAG build): the line "diag 4 ..." is the copy-only build of
               k_fused_batch (load tile + summaries, store tile) on 4.29e9 slots = 17.18 GB
  --sq         summarize.py sq output holding SQ_INSTS_VALU and SQ_BUSY_CYCLES of k_fused_batch (maxima = the
               full-size passes: 8,388,608 tiles)
  --kernel-ms  duration of such a full-size pass in the same profiled run (for the shader clock:
               SQ_BUSY_CYCLES is summed over the 32 shader engines)
"""
import argparse, csv, json, re

ap = argparse.ArgumentParser()
ap.add_argument("--diag-log", required=True)
ap.add_argument("--sq", required=True)
ap.add_argument("--kernel-ms", type=float, required=True)
ap.add_argument("--slots", type=float, default=4294967296.0)
a = ap.parse_args()
copy_ms = None
for line in open(a.diag_log):
    m = re.match(r"diag 4 .*avg ms ([0-9.]+)", line)
    if m:
        copy_ms = float(m.group(1))
valu = busy = None
for r in csv.DictReader(open(a.sq)):
    if r["kernel"] == "k_fused_batch" and r["counter"] == "SQ_INSTS_VALU":
        valu = float(r["max"])
    if r["kernel"] == "k_fused_batch" and r["counter"] == "SQ_BUSY_CYCLES":
        busy = float(r["max"])
tiles = a.slots / 512.0
print(json.dumps({
    "copy_only_GBps": 4.0 * a.slots / (copy_ms * 1e-3) / 1e9,
    "copy_only_ms": copy_ms,
    "valu_per_tile": valu / tiles,
    "valu_per_launch": valu,
    "clock_GHz": busy / 32.0 / (a.kernel_ms * 1e-3) / 1e9,
    "simds": 1024,
    "sources": "tools/fused_diag.py (MBPE_FUSED_DIAG=4) + rocprofv3 --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES, collected by profiles/collect_r03.sh",
}, indent=1))

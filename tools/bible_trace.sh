#!/bin/bash
# kernel by kernel through the config-3 stand-in: rocprofv3 trace of 3 trainings, per-kernel durations of the working dispatches
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/bt
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/bt -o bt -- python3 bench.py --config bible --steps 3 --warmup 1 --no-cpu-baseline --no-full-run > gpurun_out/r4_bible_trace_bench.json 2> gpurun_out/bt.err
f=$(find gpurun_out/bt -name "bt_kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections, re
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"(k_\w+)", r["Kernel_Name"])
    if m: agg[m.group(1)].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
print("kernel,dispatches,total_ms,median_us,p90_us,max_us")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print("%s,%d,%.2f,%.1f,%.1f,%.1f" % (k, len(v), sum(v) / 1e3, v[len(v) // 2], v[int(len(v) * 0.9)], v[-1]))
PY
rm -rf gpurun_out/bt

#!/usr/bin/env python3
"""profiles/rNN_pmc_traffic.json from the condensed FETCH_SIZE / WRITE_SIZE passes (collect_rNN.sh; usage: make_traffic_json.py <dir> [r03]):
HBM bytes per working launch of every kernel, read by bench.py for `roofline.traffic`."""
import csv
import json
import os
import sys

d = sys.argv[1]
tag = sys.argv[2] if len(sys.argv) > 2 else "r02"
out = {
    "workload": {"config": "synthetic", "corpus_bytes": 4 << 30, "vocab_size": 32000, "n_gpus": 1,
                 "command": os.environ.get("MBPE_TRAFFIC_COMMAND",
                                           "bench.py --steps 20 --warmup 5 --no-full-run --no-cpu-baseline (sequences 0..25: 4.29e9 "
                                           "slots, live tokens 4.29e9 -> 3.8e9)") + "; averages over the working dispatches of each kernel"},
    "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate runs; bytes = KB x 1024, FETCH_SIZE doubled "
              "(gfx950 wide-read correction, /opt/skills/guides/MI355X_MICROARCH.md, HBM)",
}
for counter, key in (("FETCH_SIZE", "fetch_bytes"), ("WRITE_SIZE", "write_bytes")):
    path = os.path.join(d, "%s_pmc_%s.csv" % (tag, counter))
    for r in csv.DictReader(open(path)):
        k = r["kernel"].split("<")[0]
        out.setdefault(k, {})[key] = float(r["avg_bytes_corrected"])
for k, v in out.items():
    if isinstance(v, dict) and "fetch_bytes" in v and "write_bytes" in v:
        v["hbm_bytes"] = v["fetch_bytes"] + v["write_bytes"]
print(json.dumps(out, indent=1))

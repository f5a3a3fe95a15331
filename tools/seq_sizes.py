"""Batch by batch through one training of the benchmark workload: merges committed, live tokens, the stream kernels' time
(HIP events, "time_kernels") and the wall time of every sequence when the host synchronises after each ("batch" 1).
    python3 tools/seq_sizes.py [bytes] [vocab] > gpurun_out/seq_sizes.json
MBPE_BENCH_OPTS=name=value,... sets library options first."""
import json, os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "minbpe-cc_amd", "python"))
sys.path.insert(0, ROOT)
import torch, mbpe
from bench import splitmix64_device
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4 << 30
vocab = int(sys.argv[2]) if len(sys.argv) > 2 else 32000
keep, corpus = splitmix64_device(42, n, dev)
torch.cuda.synchronize()
tr = mbpe.Trainer(0)
for kv in filter(None, os.environ.get("MBPE_BENCH_OPTS", "").split(",")):
    tr.set_option(kv.split("=")[0], int(kv.split("=")[1]))
tr.load_corpus_device(corpus.data_ptr(), n, keep=keep)
tr.set_option("batch", 1)
tr.set_option("time_kernels", 1)
tr.train_begin(vocab)
rows, k = [], 0
prev = tr.stats()
while k < vocab - 256:
    t0 = time.perf_counter()
    got = tr.train_sequences(1)
    dt = (time.perf_counter() - t0) * 1e3
    st = tr.stats()
    if got == 0:
        break
    rows.append([k, got, st["n_live"], st["n_slots"], round(st["ms_merge_kernel"] - prev["ms_merge_kernel"], 3),
                 round(st["ms_fused_kernel"] - prev["ms_fused_kernel"], 3), round(dt, 3),
                 st["cut_conflict"] - prev["cut_conflict"], st["cut_full"] - prev["cut_full"],
                 st["n_skipped"] - prev["n_skipped"], st["n_sel_retry"] - prev["n_sel_retry"]])
    prev = st
    k += got
print(json.dumps({"columns": ["first_merge", "merges", "live_after", "slots", "ms_stream_kernels", "ms_fused", "ms_wall_seq",
                              "cut_conflict", "cut_full", "skipped", "sel_retry"],
                  "sequences": len(rows), "rows": rows}))

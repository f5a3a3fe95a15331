"""Where a whole training of the benchmark workload spends its time, kernel by kernel and sequence by sequence.
Run under the profiler on the GPU box (program directly after "--"):
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/st -o st -- python3 tools/seq_trace.py run
    python3 tools/seq_trace.py sum gpurun_out/st/.../st_kernel_trace.csv gpurun_out/seq_trace_merges.json
`run` trains config 4 once (host round trip after every 16 sequences, as bench.py's full run does) and writes the
merges of every sequence group; `sum` adds up the kernel durations by name between the first selection kernel and
the last kernel of the training, the idle time of the stream between them, and the same for the passes alone."""
import csv, json, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
if sys.argv[1] == "run":
    sys.path.insert(0, os.path.join(ROOT, "minbpe-cc_amd", "python"))
    sys.path.insert(0, ROOT)
    import time, torch, mbpe
    from bench import splitmix64_device
    dev = torch.device("cuda", 0)
    n, vocab = 4 << 30, 32000
    keep, corpus = splitmix64_device(42, n, dev)
    torch.cuda.synchronize()
    tr = mbpe.Trainer(0)
    for kv in os.environ.get("MBPE_BENCH_OPTS", "").split(","):
        if kv:
            tr.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    tr.load_corpus_device(corpus.data_ptr(), n, keep=keep)
    t0 = time.perf_counter()
    tr.train_begin(vocab)
    got = tr.train_steps(vocab - 256)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    st = tr.stats()
    json.dump({"merges": got, "seconds": t1 - t0, "stats": {k: (v if isinstance(v, (int, float)) else str(v)) for k, v in st.items()}},
              open(os.path.join(ROOT, "gpurun_out", "seq_trace_run.json"), "w"))
    print(got, t1 - t0)
else:
    import gzip
    rows = list(csv.DictReader(gzip.open(sys.argv[2], "rt") if sys.argv[2].endswith(".gz") else open(sys.argv[2])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    import re
    def short(nm):
        m = re.search(r"\bk_[a-z0-9_]+", nm)
        return m.group(0) if m else ("runtime:" + nm if nm.startswith("__amd") else "torch:" + nm[:40])
    names = [short(r["Kernel_Name"]) for r in rows]
    first = next(i for i, nm in enumerate(names) if nm.startswith("k_"))
    last = max(i for i, nm in enumerate(names) if nm.startswith("k_"))
    rows, names = rows[first:last + 1], names[first:last + 1]
    t_begin, t_end = int(rows[0]["Start_Timestamp"]), int(rows[-1]["End_Timestamp"])
    tot = {}
    busy = 0
    for r, nm in zip(rows, names):
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        a = tot.setdefault(nm, [0, 0])
        a[0] += 1
        a[1] += d
        busy += d
    wall = t_end - t_begin
    print("wall_ms %.3f  kernels_ms %.3f  idle_ms %.3f" % (wall / 1e6, busy / 1e6, (wall - busy) / 1e6))
    for nm, (c, d) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
        print("%-28s %6d %10.3f ms %6.2f %%" % (nm, c, d / 1e6, 100.0 * d / wall))

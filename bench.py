#!/usr/bin/env python3
"""bench.py -- BPE training hot path on MI355X.

Contract (driver): python bench.py --gpus N --steps K --warmup W
  N > 1 is launched by torch.distributed.run, one rank per GPU.
A "step" is one iteration of the training loop (reference Tokenizer.h:557-589):
one merge = argmax over the pair table + merge of that pair in the token stream
+ count update.  (The library applies several independent merges per pass over
the stream when it can prove the result identical; steps still count merges.)
With the defaults the timed region is the whole training run to vocab 32,000
minus the warm-up merges.  The workload is BASELINE.json config 4: a SplitMix64(seed 42)
uniform-random byte corpus, `basic` encoder (one chunk), vocab 32,000.  The
corpus is generated on the device, so it is resident in HBM before the timed
region; the initial pair-count scan and stream setup run before the timed
region and are reported separately (pair_count_scan_*).

Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "minbpe-cc_amd", "python"))

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
GOLDEN = 0x9E3779B97F4A7C15


def _s64(v):
    v &= (1 << 64) - 1
    return v - (1 << 64) if v >= (1 << 63) else v


def splitmix64_device(seed, n, device, offset_bytes=0):
    """SplitMix64 byte corpus of SURVEY.md 8d generated on the GPU (bytes
    [offset_bytes, offset_bytes+n) of the stream; offset must be a multiple of 8)."""
    import torch
    assert offset_bytes % 8 == 0
    m = (n + 7) // 8
    out = torch.empty(m, dtype=torch.int64, device=device)
    step = 1 << 25
    c1, c2 = _s64(0xBF58476D1CE4E5B9), _s64(0x94D049BB133111EB)
    base = offset_bytes // 8
    for s in range(0, m, step):
        e = min(s + step, m)
        idx = torch.arange(base + s + 1, base + e + 1, dtype=torch.int64, device=device)
        z = idx * _s64(GOLDEN) + _s64(seed)
        z = (z ^ ((z >> 30) & ((1 << 34) - 1))) * c1
        z = (z ^ ((z >> 27) & ((1 << 37) - 1))) * c2
        z = z ^ ((z >> 31) & ((1 << 33) - 1))
        out[s:e] = z
    b = out.view(torch.uint8)[:n]
    if offset_bytes == 0 and n and int(b[0]) == 0:
        b[0] = 1
    return out, b


def pmc_traffic(kernel, corpus_bytes, vocab, world):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/), for the
    workload they were taken on; None for any other configuration."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
            d = json.load(f)
        w = d["workload"]
        if w["corpus_bytes"] == corpus_bytes and w["vocab_size"] == vocab and w["n_gpus"] == world:
            return d[kernel]["hbm_bytes"]
    except (OSError, KeyError, ValueError):
        pass
    return None


def cpu_baseline(seed, sample_bytes, merges, vocab):
    """The CPU oracle (a port of the reference algorithm: sequential pass per
    merge + incremental counts + ordered argmax) timed on one host core."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    data = O.splitmix64_bytes(seed, sample_bytes)
    t0 = time.perf_counter()
    st = O.State(data)
    t1 = time.perf_counter()
    for i in range(merges):
        top = st.top()
        st.merge(top[0], top[1], 256 + i)
    t2 = time.perf_counter()
    st.close()
    return {
        "value": merges / (t2 - t1),
        "unit": "merges/s",
        "cores": 1,
        "kind": "port",
        "sample": "first %d MiB of the workload corpus, first %d merges, single thread (the reference is "
                  "single-threaded); per-merge cost is linear in corpus bytes" % (sample_bytes >> 20, merges),
        "sample_bytes": sample_bytes,
        "pair_count_scan_MBps": sample_bytes / 1e6 / (t1 - t0),
        "host_cores_available": os.cpu_count(),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=31728, help="merges timed (default: the whole training run)")
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--bytes", type=int, default=4 << 30, help="corpus bytes (whole job)")
    ap.add_argument("--vocab", type=int, default=32000)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--cpu-sample-mib", type=int, default=256)
    ap.add_argument("--cpu-merges", type=int, default=48)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import mbpe

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        dist.init_process_group("nccl", device_id=device)

    steps = min(args.steps, args.vocab - 256 - args.warmup)
    # ---- corpus shard of this rank (contiguous byte range, 8-byte aligned cuts)
    per = (args.bytes // world) // 4096 * 4096
    lo = rank * per
    hi = args.bytes if rank == world - 1 else lo + per
    keep, corpus = splitmix64_device(args.seed, hi - lo, device, lo)
    torch.cuda.synchronize()

    tr = mbpe.Trainer(local_rank)
    for kv in filter(None, os.environ.get("MBPE_BENCH_OPTS", "").split(",")):     # e.g. fused_min=16,max_batch=64
        k, v = kv.split("=")
        tr.set_option(k, int(v))
    if world > 1:
        uid = [mbpe.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        tr.comm_init(uid[0], rank, world)
    tr.load_corpus_device(corpus.data_ptr(), hi - lo, keep=keep)

    # ---- pair-count scan (graded kernel): a few timed launches
    scan_ms = []
    for _ in range(5):
        tr.pair_count_u8(want_table=False)
        scan_ms.append(tr.stats()["ms_pair_count"])
    scan_ms_best = sorted(scan_ms)[len(scan_ms) // 2]

    tr.set_option("time_kernels", 1)
    tr.train_begin(args.vocab)
    begin_stats = tr.stats()
    tr.train_steps(args.warmup)
    s0 = tr.stats()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    done = tr.train_steps(steps)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    s1 = tr.stats()
    elapsed = t1 - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    merges, counts = tr.train_result()
    # ---- the dominant kernel: k_fused_batch, one pass = one read of the stream + the merged stream
    # written to the other buffer, for every pair of the batch at once.  Bytes such a pass has to move:
    # 2 B read + 2 B written per slot.
    n_fused = s1["fused_launches"] - s0["fused_launches"]
    ms_fused = s1["ms_fused_kernel"] - s0["ms_fused_kernel"]
    slots_fused = s1["fused_slots"] - s0["fused_slots"]
    avg_fused_ms = ms_fused / max(n_fused, 1)
    pass_bytes = 4.0 * slots_fused / max(n_fused, 1)
    achieved = pass_bytes / (avg_fused_ms * 1e-3) / 1e9 if avg_fused_ms > 0 else 0.0
    n_pass = s1["n_batches"] - s0["n_batches"]
    roof_kernel = ("k_fused_batch: reads the stream once, counts the deltas of every pair of the batch and "
                   "writes the merged stream to the other buffer")
    if n_fused == 0:
        # a short timed region may hold small batches only: those read the stream once (k_scan_batch /
        # k_merge) and write where a match is; report that pass instead
        n_fused = s1["merge_launches"] - s0["merge_launches"]
        avg_fused_ms = (s1["ms_merge_kernel"] - s0["ms_merge_kernel"]) / max(n_fused, 1)
        pass_bytes = 2.0 * s1["n_slots"]
        achieved = pass_bytes / (avg_fused_ms * 1e-3) / 1e9 if avg_fused_ms > 0 else 0.0
        roof_kernel = "k_scan_batch / k_merge: the read-only stream pass of small batches (no fused pass in the timed region)"
    # SURVEY.md 8(d) prices a merge step at 2 B x L read + 2 B x L' written; a pass performs
    # merges_per_pass of them on one read: the same sum divided by the measured time
    live_avg = 0.5 * (s0["n_live"] + s1["n_live"])
    ref_model_bytes = 4.0 * live_avg * (done / max(n_pass, 1))
    scan_gbs = (hi - lo) / (scan_ms_best * 1e-3) / 1e9

    if rank == 0:
        out = {
            "metric": "bpe_train_merges_per_sec",
            "value": done / elapsed,
            "unit": "merges/s",
            "n_gpus": world,
            "steps": done,
            "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / max(done, 1),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "u16",
            "data": "synthetic",
            "config": {
                "workload": "SplitMix64(seed %d) uniform-random bytes, %d bytes whole job, basic encoder "
                            "(one chunk), vocab %d, lexicographic tie-break; steps are merges %d..%d"
                            % (args.seed, args.bytes, args.vocab, args.warmup, args.warmup + done),
                "corpus_bytes": args.bytes,
                "vocab_size": args.vocab,
                "parallelism": "stream sharded over %d GPU(s), pair table replicated" % world,
            },
            "pair_count_scan_MBps": scan_gbs * 1e3 * world,
            "pair_count_scan_ms": scan_ms_best,
            "roofline": {
                "kernel": roof_kernel,
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": pmc_traffic("k_fused_batch", args.bytes, args.vocab, world),
                "traffic_source": "profiles/r01_pmc_traffic.json (rocprofv3 PMC passes of this workload)",
                "limiter": "vector instruction issue, not HBM: ~350 VALU wave-instructions per 512-slot tile keep the "
                           "SIMDs ~90 % busy (profiles/r01_final_pmc_sq.csv, DESIGN.md section 4)",
                "algorithmic_bytes_per_launch": pass_bytes,
                "avg_launch_ms": avg_fused_ms,
                "launches": n_fused,
                "stream_passes": n_pass,
                "merges_per_pass": done / max(n_pass, 1),
                "survey_8d_model": {
                    "note": "SURVEY 8(d) counts 2 B x L read + 2 B x L' written PER MERGE; one pass serves "
                            "merges_per_pass merges, so that sum over the pass's merges divided by the pass time "
                            "exceeds the HBM peak: the saving is algorithmic, the kernel itself runs at `achieved`",
                    "bytes_per_pass": ref_model_bytes,
                    "effective_GBps": ref_model_bytes / (elapsed / max(n_pass, 1)) / 1e9,
                },
            },
            "roofline_pair_count": {
                "kernel": "k_pair_count_u8",
                "bound": "hbm",
                "achieved": scan_gbs,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": scan_gbs / HBM_PEAK_GBS,
                "traffic": pmc_traffic("k_pair_count_u8", args.bytes, args.vocab, world),
                "algorithmic_bytes_per_launch": hi - lo,
                "avg_launch_ms": scan_ms_best,
            },
            "begin_ms": begin_stats["ms_begin"],
            "stream": {"slots": s1["n_slots"], "live": s1["n_live"], "compactions": s1["n_compactions"],
                       "pairs": s1["n_pairs"]},
            "batches": {k: s1[k] for k in ("n_batches", "n_fused", "n_fused_dropped", "cut_conflict", "cut_bucket",
                                           "cut_single", "cut_full", "n_validation_drops", "ms_grow_table", "ms_compact",
                                           "n_table_grows", "ms_steps", "n_sel_fallback", "n_sel_retry", "size_hist", "n_skipped", "n_skip_cut")},
            "first_counts": [int(c) for c in counts[:3]],
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.seed, min(args.cpu_sample_mib << 20, args.bytes),
                                               args.cpu_merges, args.vocab)
        print(json.dumps(out))
    tr.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

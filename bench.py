#!/usr/bin/env python3
"""bench.py -- BPE training hot path on MI355X.

Contract (driver): python bench.py --gpus N --steps K --warmup W
  N > 1 is launched by torch.distributed.run, one rank per GPU.

A "step" is ONE WHOLE TRAINING of the workload: mbpe_train_begin (pair-count scan, slot stream, pair
table, first argmax) + the loop Tokenizer.h:557-589 run to the target vocabulary (31,744 merges on the
default workload) -- one pass of the hot path over one batch of synthetic input.  W trainings run
untimed, then exactly K trainings are timed between barriers;
    value = K x merges of one training / wall time                 (merges/s, whole job)
so the headline is the end-to-end number of a training run, not a window of it.

Workload (--config synthetic, the default): BASELINE.json config 4 -- SplitMix64(seed 42)
uniform-random bytes, 4 GiB whole job, `basic` encoder (one chunk), vocab 32,000.  The corpus is
generated on the device, so it is resident in HBM before anything is timed; the pair-count scan
(the other half of BASELINE.json's metric) is part of every step and is also timed on its own
(roofline_pair_count: start/stop events of each dispatch).
--config bible: BASELINE.json config 3 with the documented stand-in corpus (data/bible.txt is absent
from the reference checkout): shakespeare.txt x 4 = 4,461,576 bytes, vocab 10,000.

`roofline` describes the kernel that dominates the timed steps (k_fused_batch) over EVERY launch of it
inside the timed region (HIP events on the library's stream): `achieved` prices SURVEY.md 8(d)'s
algorithmic bytes -- 2 B per live token read + 2 B per live token written -- and `achieved_slot_bytes`
what the kernel physically moves (2 B + 2 B per slot, holes included).

After the timed region, outside it, the result of the last training is checked on the device
(`checks`): decode(stream) == corpus, chosen counts never increase, every rank holds the same merges,
and a further training is stopped at checkpoints where the table is compared with a recount.

Rank 0 prints ONE JSON line.
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "minbpe-cc_amd", "python"))

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
GOLDEN = 0x9E3779B97F4A7C15
PMC_FILE = os.path.join("profiles", "r04_pmc_traffic.json")


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


def bible_standin_device(device, lo, hi):
    """shakespeare.txt x 4, bytes [lo, hi), in a 16-byte aligned device buffer."""
    import numpy as np
    import torch
    with open(os.path.join(ROOT, "tests", "golden", "data", "shakespeare.txt"), "rb") as f:
        one = f.read()
    data = np.frombuffer(one * 4, dtype=np.uint8)[lo:hi]
    keep = torch.zeros((len(data) + 15) // 16 * 2, dtype=torch.int64, device=device)
    b = keep.view(torch.uint8)[:len(data)]
    b.copy_(torch.from_numpy(data.copy()))
    return keep, b


def pmc_traffic(kernel, config, corpus_bytes, vocab, world):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/), for the
    workload they were taken on; None for any other configuration or kernel."""
    try:
        with open(os.path.join(ROOT, PMC_FILE)) as f:
            d = json.load(f)
        w = d["workload"]
        if (w.get("config", "synthetic") == config and w["corpus_bytes"] == corpus_bytes
                and w["vocab_size"] == vocab and w["n_gpus"] == world):
            return d[kernel]["hbm_bytes"]
    except (OSError, KeyError, ValueError):
        pass
    return None


def cpu_baseline(data, merges, what):
    """The CPU oracle (a port of the reference algorithm: sequential pass per merge + incremental
    counts + ordered argmax) timed on one host core.  Checker code, used here only as the baseline."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    t0 = time.perf_counter()
    st = O.State(data)
    t1 = time.perf_counter()
    done = 0
    for i in range(merges):
        top = st.top()
        if top is None:
            break
        st.merge(top[0], top[1], 256 + i)
        done += 1
        if time.perf_counter() - t1 > 30.0:
            break
    t2 = time.perf_counter()
    st.close()
    return {
        "value": done / (t2 - t1),
        "unit": "merges/s",
        "cores": 1,
        "kind": "port",
        "sample": "%s, first %d merges, single thread (the reference is single-threaded); per-merge cost is "
                  "linear in corpus bytes" % (what, done),
        "sample_bytes": len(data),
        "pair_count_scan_MBps": len(data) / 1e6 / (t1 - t0),
        "host_cores_available": os.cpu_count(),
    }


# What a fused pass cannot go below, from measurements kept under profiles/ (profiles/make_floor_json.py writes the
# file from a timing-only build of the kernel and from the SQ counters of the shipped one; nothing is hard-coded here):
#   copy_only_GBps   k_fused_batch with everything but the tile copy compiled out (MBPE_FUSED_DIAG=4)
#   valu_per_tile    vector instructions the shipped kernel executes per 512-slot tile (SQ_INSTS_VALU / tiles) in the
#                    pass that executes most (a 4096-pair batch)
#   clock_GHz        shader clock during the pass (SQ_BUSY_CYCLES per shader engine / duration)
#   ladder_ms        the same four passes with timing-only builds: copy, + lookups, all but the count deltas, shipped;
#                    delta_ps_per_match = (shipped - all but the count deltas) / matches of a pass
# floor_model.floor_ms is the streaming floor; model_ms = "all but the count deltas" scaled to the pass's tiles + the
# count deltas of its matches, next to measured_ms.
FLOOR_FILE = os.path.join("profiles", "r04_fused_floor.json")
FUSED_LIMITER = ("issue and the count-delta atomics, additively: timing-only builds of the kernel on the same passes "
                 "(floor_model.ladder_ms from profiles/r04_fused_floor.json; tools/fused_diag.py) copy the stream in ~3.5 ms, "
                 "hide the nine lookups per lane under that, take ~6.4 ms with everything but the count-delta atomics and 15 ms "
                 "as shipped on passes of 2,100-4,096 pairs (1.9e8 matches): 44 ps per match that no other work of the wave "
                 "overlaps (four waves per SIMD: the lookup table fills the LDS).  Round 4 took a quarter of those atomics away "
                 "(a match between two raw bytes: one atomic on its pair's byte x byte cell block instead of two on the L / R "
                 "rows, k_pair_cells_fold behind the pass: 70 -> 44 ps per match) and made the stream's loads and stores "
                 "non-temporal (+3 %); what remains is the memory side's rate for scattered atomics (~30 G/s) in the passes of "
                 "thousands of pairs and ~1.5 ms of vector-issue-bound work per pass everywhere else (DESIGN.md section 4, "
                 "profiles/HISTORY.md)")


def published_workload(device, with_cpu):
    """The one workload the reference publishes a number for (README.md:103-113, code/examples/train.cpp:87-100): read
    shakespeare.txt, train the basic tokenizer and then the GPT-4-split tokenizer to vocab 512, lexical tie-break,
    end to end -- file read, regex pre-split, upload, training.  Here through mbpe_tok_train (the Tokenizer::train
    mirror; its verbose printing is off, the reference's timing has it on), next to the CPU oracle port on the same
    input and the reference's own 1.50 s (MacBook Pro M1 Pro, one thread) as quoted context.  1.1 MB is far too small to
    measure a GPU -- the time is launches and host work -- it is here because it is the only published point."""
    import numpy as np
    import mbpe
    path = os.path.join(ROOT, "tests", "golden", "data", "shakespeare.txt")
    gpt4 = mbpe.split_pattern("gpt4")
    res = {"input": "shakespeare.txt (1,115,394 bytes), vocab 512, basic then gpt4, lexical",
           "reference_published_seconds": 1.50,
           "reference_published_source": "/root/reference README.md:105 (train.cpp, M1 Pro, single thread, verbose on)"}
    for attempt in ("first (cold: library and PCRE2 loaded, buffers allocated)", "second"):
        t0 = time.perf_counter()
        with open(path, "rb") as f:
            text = f.read()
        models = []
        for pat in ("", gpt4):
            tk = mbpe.Tokenizer(pat)
            tk.train(text, 512, mbpe.Tokenizer.LEXICAL, False, device)
            models.append(tk.merges().copy())
            tk.close()
        dt = time.perf_counter() - t0
        res["seconds_" + attempt.split(" ")[0]] = dt
    res["merges"] = 512
    res["merges_per_s"] = 512 / res["seconds_second"]
    res["seconds"] = res["seconds_second"]
    if with_cpu:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle as O
        data = np.frombuffer(text, dtype=np.uint8)
        t0 = time.perf_counter()
        m0, _ = O.train(data, 512)
        off = mbpe.presplit(gpt4, text)
        m1, _ = O.train(data, 512, off)
        res["cpu_port_seconds"] = time.perf_counter() - t0
        res["cpu_port_is"] = "oracle/bpe_oracle.c, one core, the same two trainings (pre-split by the library's PCRE2 binding)"
        res["identical_merges"] = bool(np.array_equal(np.asarray(m0), models[0]) and np.array_equal(np.asarray(m1), models[1]))
    return res


def fused_floor():
    try:
        with open(os.path.join(ROOT, FLOOR_FILE)) as f:
            return json.load(f)
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3, help="whole trainings timed")
    ap.add_argument("--warmup", type=int, default=1, help="whole trainings before the timed region")
    ap.add_argument("--config", choices=["synthetic", "bible"], default="synthetic")
    ap.add_argument("--bytes", type=int, default=4 << 30, help="corpus bytes, whole job (synthetic)")
    ap.add_argument("--vocab", type=int, default=0, help="default 32000 (synthetic) / 10000 (bible)")
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--cpu-sample-mib", type=int, default=256)
    ap.add_argument("--cpu-merges", type=int, default=48)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-full-run", action="store_true", help="skip the untimed checks after the timed region")
    args = ap.parse_args()

    import numpy as np
    import torch
    import mbpe
    from mbpe import check as C

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

    bible = args.config == "bible"
    vocab = args.vocab or (10000 if bible else 32000)
    total_bytes = 4 * 1115394 if bible else args.bytes
    # ---- corpus shard of this rank (contiguous byte range, 4096-byte aligned cuts)
    per = (total_bytes // world) // 4096 * 4096
    lo = rank * per
    hi = total_bytes if rank == world - 1 else lo + per
    if bible:
        keep, corpus = bible_standin_device(device, lo, hi)
    else:
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

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def one_training():
        """One step: the whole training.  Returns (merges, seconds, stats after it)."""
        t = time.perf_counter()
        tr.train_begin(vocab)
        n = tr.train_steps(vocab - 256)          # (ends with a host synchronisation on the library's stream)
        return n, time.perf_counter() - t, tr.stats()

    # ---- timed region: K whole trainings after W warm-up trainings
    # (HIP events around the stream kernels of every sequence, 4 records each: nothing next to a 4 ms pass; on the 4.5 MB
    #  of config 3, where a sequence takes 0.16 ms, they would be a fifth of the run -- there the kernels are timed in
    #  one extra, untimed training)
    events_in_timed_region = not bible
    tr.set_option("time_kernels", 1)
    if not events_in_timed_region:
        e0 = tr.stats()
        one_training()
        e1 = tr.stats()
        tr.set_option("time_kernels", 0)
    for _ in range(args.warmup):
        one_training()
    s0 = tr.stats()
    barrier()
    t0 = time.perf_counter()
    steps = [one_training() for _ in range(args.steps)]
    barrier()
    t1 = time.perf_counter()
    s1 = tr.stats()
    comp0, comp1 = s0["n_compactions"], s1["n_compactions"]
    elapsed = max_over_ranks(t1 - t0)
    done = sum(n for n, _, _ in steps)
    per_training = steps[-1][0] if steps else 0
    last = steps[-1][2] if steps else s1

    # ---- the dominant kernel inside the timed region.  A fused pass (k_fused_batch) reads the stream once,
    # counts the deltas of every pair of the batch and writes the merged stream to the other buffer.
    # Algorithmic bytes: 2 B x L read + 2 B x L' written, L / L' the live tokens before / after the pass;
    # physically it moves 2 B + 2 B per SLOT (holes are squeezed out only at a compaction).
    n_pass = sum(st["n_batches"] for _, _, st in steps)
    if not events_in_timed_region:
        s0, s1 = e0, e1                          # (kernel events: the extra training above)
    n_fused = s1["fused_launches"] - s0["fused_launches"]
    n_other = (s1["merge_launches"] - s0["merge_launches"]) - n_fused
    ms_fused = s1["ms_fused_kernel"] - s0["ms_fused_kernel"]
    ms_all = s1["ms_merge_kernel"] - s0["ms_merge_kernel"]
    slot_bytes = None
    if n_fused > 0 and ms_fused >= 0.5 * ms_all:
        roof_name = "k_fused_batch"
        roof_desc = ("reads the stream once, counts the deltas of every pair of the batch and writes the merged stream "
                     "to the other buffer; `achieved` = SURVEY 8(d)'s bytes (2 B per live token read + 2 B per live token "
                     "written) over the kernel's time, `achieved_slot_bytes` = what it moves (2 B + 2 B per slot, holes "
                     "included), both over every launch of the timed trainings")
        launches = n_fused
        avg_ms = ms_fused / n_fused
        pass_bytes = 2.0 * (s1["fused_live_tokens"] - s0["fused_live_tokens"]) / n_fused
        slot_bytes = 4.0 * (s1["fused_slots"] - s0["fused_slots"]) / n_fused
        limiter = FUSED_LIMITER
    else:
        roof_name = "k_scan_batch"
        roof_desc = "the read-only stream pass of small batches (k_scan_batch / k_merge; 2 B read per slot)"
        launches = max(n_other, 1)
        avg_ms = (ms_all - ms_fused) / launches
        pass_bytes = 2.0 * s1["n_slots"]
        limiter = None
    achieved = pass_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    achieved_slots = slot_bytes / (avg_ms * 1e-3) / 1e9 if slot_bytes and avg_ms > 0 else None
    # what a fused pass cannot go below: its bytes at the rate of the copy-only build of the same kernel; and the ladder
    # of timing-only builds that explains the rest (profiles/, written by committed scripts)
    floor_model = None
    if roof_name == "k_fused_batch":
        fl = fused_floor()
        if fl:
            tiles = slot_bytes / 4.0 / 512.0
            stream_ms = slot_bytes / (fl["copy_only_GBps"] * 1e9) * 1e3
            matches = (len(steps) * (hi - lo) - sum(st["n_live"] for _, _, st in steps)) / max(n_pass, 1)
            lad = fl.get("ladder_ms") or {}
            full_tiles = 4294967296.0 / 512.0           # the ladder was measured on passes over 2^32 slots
            no_delta_ms = lad.get("all_but_count_deltas") * tiles / full_tiles if lad.get("all_but_count_deltas") else None
            delta_ms = fl["delta_ps_per_match"] * matches * 1e-9 if fl.get("delta_ps_per_match") else None
            floor_model = {"stream_ms": stream_ms, "copy_only_GBps": fl["copy_only_GBps"],
                           "tiles_per_launch": tiles, "matches_per_launch": matches,
                           "ladder_ms": lad, "ladder_matches_per_launch": fl.get("ladder_matches_per_launch"),
                           "all_but_count_deltas_ms": no_delta_ms, "count_deltas_ms": delta_ms,
                           "model_ms": (no_delta_ms + delta_ms) if no_delta_ms and delta_ms else None,
                           "floor_ms": stream_ms, "measured_ms": avg_ms,
                           "sources": FLOOR_FILE + " (" + fl.get("sources", "") + ")"}
    timed = {"trainings": args.steps, "merges": done, "merges_per_training": per_training,
             "stream_passes": n_pass, "fused_passes": n_fused,
             "seconds": elapsed, "seconds_each": [round(sec, 5) for _, sec, _ in steps],
             "ms_stream_kernels": ms_all, "ms_fused_kernel": ms_fused,
             "live_end": last["n_live"], "slots_end": last["n_slots"],
             "includes": "pair-count scan, stream + table setup, every sequence, host housekeeping (compactions)"}

    # ---- untimed: checks of the last timed training's result on the device, then a further training stopped at checkpoints
    full_run, checks = None, None
    if steps:
        sec = [x for _, x, _ in steps]
        full_run = {"merges": per_training, "seconds": sum(sec) / len(sec), "seconds_best": min(sec),
                    "merges_per_s": done / elapsed if elapsed > 0 else 0.0,
                    "is": "one training of the timed region (the value above is exactly this: a step is a whole training)",
                    "passes": last["n_batches"], "fused_passes": last["n_fused"], "fused_abandoned": last["n_fused_dropped"],
                    "begin_ms": last["ms_begin"], "steps_ms": last["ms_steps"],
                    "compactions": (comp1 - comp0) / max(len(steps), 1),
                    "live_end": last["n_live"], "slots_end": last["n_slots"], "pairs": last["n_pairs"],
                    "selection": {k: last[k] for k in ("n_sel_retry", "n_sel_fallback", "cut_conflict", "cut_bucket",
                                                       "cut_single", "cut_full", "n_skipped", "n_skip_cut",
                                                       "n_validation_drops")}}
    if not args.no_full_run and steps:
        tr.set_option("time_kernels", 0)
        merges, counts = tr.train_result()
        # decode(final stream) == corpus
        if dist is None:
            ref = corpus
        else:
            # a merge that straddles two shards leaves its token with the left one: this rank's stream decodes to
            # the corpus bytes that follow those of the ranks before it, whatever range it was loaded with
            mine = C.decoded_length(tr, merges, torch, device)
            lens = [None] * world
            dist.all_gather_object(lens, mine)
            start = sum(lens[:rank])
            if bible:
                _, whole = bible_standin_device(device, 0, total_bytes)
                ref = whole[start:start + mine]
            else:
                a0 = start // 8 * 8
                _, piece = splitmix64_device(args.seed, start - a0 + mine, device, a0)
                ref = piece[start - a0:]
        rt = C.decode_roundtrip(tr, merges, ref, torch, device)
        if dist is not None:
            rt["decoded_total_all_ranks"] = int(sum(lens))
            rt["ok"] = bool(rt["ok"] and sum(lens) == total_bytes)
        # ---- the ORDER of merges at full size (a further training on one GPU): at checkpoints spread over the run the
        # stream is recounted from scratch on the device; the table must equal the recount there, and the merge the
        # trainer commits next must be the (count desc, key asc) argmax of the recounted table
        argmax_checks = None
        if dist is None and last["n_pairs"] and vocab <= 32768:
            tr.train_begin(vocab)
            argmax_checks = []
            total = vocab - 256
            for frac in (0.0, 0.12, 0.35, 0.6, 0.8, 0.9, 0.97):
                target = int(total * frac)
                have = len(tr.train_result()[0])
                if target > have:
                    tr.train_steps(target - have)
                if len(tr.train_result()[0]) >= total:
                    break
                argmax_checks.append(C.argmax_at_checkpoint(tr, torch, device))
            tr.train_steps(total)          # to the end: this run's merges must equal the timed run's
            m2, c2 = tr.train_result()
            argmax_checks.append({"second_run_identical": bool(np.array_equal(m2, merges) and np.array_equal(c2, counts))})
        digest = hashlib.sha256(merges.tobytes() + counts.tobytes()).hexdigest()
        same = True
        if dist is not None:
            got = [None] * world
            dist.all_gather_object(got, digest)
            same = all(g == got[0] for g in got)
            flag = torch.tensor([1.0 if rt["ok"] else 0.0], dtype=torch.float64, device=device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            rt["ok_all_ranks"] = bool(flag.item() == 1.0)
        checks = {
            "decode_roundtrip": rt,
            "counts_nonincreasing": C.counts_nonincreasing(counts),
            "merges_complete": int(len(merges)) == vocab - 256,
            "ranks_identical_merges": same,
            "merges_sha256": digest,
            "first_counts": [int(c) for c in counts[:3]],
            "last_counts": [int(c) for c in counts[-3:]],
        }
        if argmax_checks is not None:
            checks["argmax_at_checkpoints"] = [bool(a.get("ok", a.get("second_run_identical"))) for a in argmax_checks]
            checks["argmax_checkpoints"] = argmax_checks
        checks["ok"] = bool(rt["ok"] and rt.get("ok_all_ranks", True) and checks["counts_nonincreasing"]
                            and checks["merges_complete"] and same
                            and all(checks.get("argmax_at_checkpoints", [True])))

    # ---- pair-count scan (the north star's graded kernel), timed last: an idle chip needs tens of launches of this
    # kernel (some tens of milliseconds of work) to reach its sustained clock -- launch times fall by a fifth
    # meanwhile; the checks before this point keep it busy with other kernels, and on some boxes the times are still
    # falling after 25 launches (group means 1.07 / 1.05 / 1.02 / 1.00 / 0.985 and then 0.985 / 0.977 / 0.975 ms:
    # profiles/README.md).  60 launches of warm-up, whose group means are reported too, then 30 launches in
    # three groups of ten back to back; every dispatch carries its own start / stop events (hipExtLaunchKernelGGL),
    # so avg_launch_ms is the mean KERNEL duration -- no gap between launches, no marker -- which is what rocprofv3
    # reports per dispatch.  (ms_bracket: the same launches between one pair of stream events per group, gaps included.)
    scan_ms, scan_bracket = [], []
    tr.set_option("pc_repeat", 5)
    n_warm_groups = 12
    for _ in range(n_warm_groups):           # 60 launches of warm-up in groups of five (their means are reported too)
        tr.pair_count_u8(want_table=False)
        scan_ms.append(tr.stats()["ms_pair_count_kernel"])
    tr.set_option("pc_repeat", 10)
    for _ in range(3):
        tr.pair_count_u8(want_table=False)
        st = tr.stats()
        scan_ms.append(st["ms_pair_count_kernel"])
        scan_bracket.append(st["ms_pair_count"])
    tr.set_option("pc_repeat", 1)
    scan_ms_avg = sum(scan_ms[n_warm_groups:]) / 3.0
    scan_gbs = (hi - lo) / (scan_ms_avg * 1e-3) / 1e9 if scan_ms_avg > 0 else 0.0

    if rank == 0:
        if bible:
            workload = ("BASELINE config 3 stand-in (data/bible.txt is absent from the reference checkout): "
                        "shakespeare.txt x 4 = %d bytes, basic encoder (one chunk), vocab %d, lexicographic "
                        "tie-break" % (total_bytes, vocab))
        else:
            workload = ("BASELINE config 4: SplitMix64(seed %d) uniform-random bytes, %d bytes whole job, basic "
                        "encoder (one chunk), vocab %d, lexicographic tie-break" % (args.seed, total_bytes, vocab))
        out = {
            "metric": "bpe_train_merges_per_sec",
            "value": done / elapsed if elapsed > 0 else 0.0,
            "unit": "merges/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / max(args.steps, 1),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "u16",
            "data": "synthetic" if not bible else "shakespeare.txt x 4 (fixture)",
            "config": {
                "workload": workload + "; a step is ONE WHOLE TRAINING (pair-count scan, stream and table setup, every "
                            "merge up to the vocabulary: %d merges)" % per_training,
                "corpus_bytes": total_bytes,
                "vocab_size": vocab,
                "parallelism": "stream sharded over %d GPU(s), pair table replicated, one sum all-reduce of the "
                               "count deltas per sequence" % world,
            },
            "timed_region": timed,
            "merges_per_step": per_training,
            "pair_count_scan_MBps": scan_gbs * 1e3 * world,
            "pair_count_scan_ms": scan_ms_avg,
            "roofline": {
                "kernel": roof_name,
                "what": roof_desc,
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "achieved_slot_bytes": achieved_slots,
                "frac_slot_bytes": achieved_slots / HBM_PEAK_GBS if achieved_slots else None,
                "traffic": pmc_traffic(roof_name, args.config, total_bytes, vocab, world),
                "traffic_source": PMC_FILE + " (rocprofv3 PMC passes of this workload; null when none was taken "
                                  "for this kernel / configuration)",
                "limiter": limiter,
                "algorithmic_bytes_per_launch": pass_bytes,
                "slot_bytes_per_launch": slot_bytes,
                "avg_launch_ms": avg_ms,
                "launches": launches,
                "stream_passes": n_pass,
                "merges_per_pass": done / max(n_pass, 1),
                "floor_model": floor_model,
            },
            "roofline_pair_count": {
                "kernel": "k_pair_count_u8_fast",
                "bound": "hbm",
                "bound_is": "the LDS: one atomic per pair into a 128-KiB histogram per CU; 2^32 random LDS atomics alone take "
                            "0.81 ms on this chip (tools/lds_atomic_floor.hip: 0.66 of the HBM peak), the HBM roof is the "
                            "reporting convention",
                "achieved": scan_gbs,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": scan_gbs / HBM_PEAK_GBS,
                "traffic": pmc_traffic("k_pair_count_u8_fast", args.config, total_bytes, vocab, world),
                "algorithmic_bytes_per_launch": hi - lo,
                "avg_launch_ms": scan_ms_avg,
                "avg_launch_ms_is": "mean kernel duration of launches 61-90 of 90 (three groups of ten launches back to back), "
                                    "each dispatch timed by its own start/stop events (hipExtLaunchKernelGGL), after 60 "
                                    "warm-up launches (the kernel's sustained clock: an idle or differently loaded chip "
                                    "takes tens of launches of it to get there; launch_ms_all shows the approach), taken "
                                    "after the training runs of this process",
                "launch_ms_all": [round(x, 4) for x in scan_ms],
                "launch_ms_all_is": "mean kernel duration of the twelve warm-up groups of five launches, then of each timed group of ten",
                "ms_bracket": [round(x, 4) for x in scan_bracket],
                "ms_bracket_is": "the same groups of ten between ONE pair of stream events, per launch: gaps between "
                                 "launches included",
            },
            "begin_ms": last["ms_begin"],
            "full_run": full_run,
            "checks": checks,
            "batches": {k: last[k] for k in ("n_batches", "n_fused", "n_fused_dropped", "cut_conflict", "cut_bucket",
                                             "cut_single", "cut_full", "n_validation_drops", "n_sel_fallback",
                                             "n_sel_retry", "size_hist", "n_skipped", "n_skip_cut")},
        }
        if not args.no_cpu_baseline and world == 1:
            if bible:
                sample = corpus.cpu().numpy()
                what = "the whole stand-in corpus (%d bytes)" % len(sample)
            else:
                sys.path.insert(0, os.path.join(ROOT, "oracle"))
                import oracle as O
                sample = O.splitmix64_bytes(args.seed, min(args.cpu_sample_mib << 20, total_bytes))
                what = "first %d MiB of the workload corpus" % (len(sample) >> 20)
            out["cpu_baseline"] = cpu_baseline(sample, args.cpu_merges if not bible else vocab - 256, what)
        if world == 1 and not args.no_full_run:
            out["published_workload"] = published_workload(local_rank, not args.no_cpu_baseline)
        print(json.dumps(out))
        if checks is not None and not checks["ok"]:
            print("bench.py: the full run FAILED its checks: %s" % json.dumps(checks), file=sys.stderr)
    failed = checks is not None and not checks["ok"]
    tr.close()
    if dist is not None:
        dist.destroy_process_group()
    if failed:
        sys.exit(3)


if __name__ == "__main__":
    main()

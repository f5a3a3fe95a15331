"""CPU, world_size 2 and 3 over gloo: the sharded training protocol
(tests/dist_model.py, the executable spec of the multi-GPU path) against the
single-rank oracle.  Each rank holds one contiguous shard; the only
communication is sum all-reduces: one at the start, then one per merge
(per_merge) or two per batch sequence (batch_sequences: the count deltas of
all the batch's pairs, then the shard edges) -- the exchange the product runs."""
import os
import socket
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _cases():
    import oracle as O
    rng = np.random.default_rng(5)
    cases = []
    cases.append((O.splitmix64_bytes(3, 3000).tobytes(), None, 256 + 25, None))
    for _ in range(6):
        n = int(rng.integers(2, 700))
        data = rng.integers(97, 97 + int(rng.integers(1, 4)), size=n, dtype=np.uint8).tobytes()
        cases.append((data, None, 256 + 20, None))
    cases.append((b"xy" + b"ab" * 151 + b"z", None, 256 + 8, [2 + 151]))        # cut inside a match
    cases.append((b"xy" + b"ab" * 151 + b"z", None, 256 + 8, [2 + 150]))        # cut between matches
    cases.append((b"a" * 257, None, 256 + 10, [128]))
    cases.append((b"a" * 64, None, 256 + 8, [1]))
    text = open(os.path.join(HERE, "golden", "data", "taylorswift.txt"), "rb").read()[:6000]
    cases.append((text, None, 256 + 30, None))
    cases.append((text, "gpt4", 256 + 30, None))
    return cases


def _worker(rank, world, port, q, batched=False, first=False):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "minbpe-cc_amd", "python"))
    sys.path.insert(0, HERE)
    import torch
    import torch.distributed as dist
    import oracle as O
    import mbpe
    from dist_model import Shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def allreduce(arr):
        t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.int64))
        dist.all_reduce(t)
        return t.numpy()

    try:
        for ci, (data, enc, vocab, cuts) in enumerate(_cases()):
            off = None if enc is None else mbpe.presplit(O.PATTERNS[enc], data)
            n = len(data)
            if cuts is None or len(cuts) != world - 1:
                cuts = [n * (r + 1) // world for r in range(world - 1)]
            if off is not None:      # shards hold whole chunks
                cuts = [int(off[np.searchsorted(off, c)]) for c in cuts]
            bounds = [0] + list(cuts) + [n]
            lo, hi = bounds[rank], bounds[rank + 1]
            loff = None
            if off is not None:
                sel = off[(off >= lo) & (off <= hi)]
                loff = (sel - lo).astype(np.int64)
            sh = Shard(data[lo:hi], loff, rank, world, allreduce)
            if batched:
                merges, counts = sh.train_batched(vocab, limit=8)
                # what crossed the wire per sequence: header + batch header + exactly the LR rows of the batch's pairs
                for nb, ids, words in sh.exchanged:
                    assert words == sh.hdr + (sh.BATCH_MAX + sh.BATCH_MAX ** 2) + 2 * nb * ((ids + 63) & ~63)
                if ci in (0, 11, 12):       # (random bytes, text: the tiny-alphabet cases have nothing independent to batch)
                    assert sum(nb >= 2 for nb, _, _ in sh.exchanged) >= 3, "case %d ran no batches" % ci
            elif first:
                merges, counts = sh.train_first(vocab)
            else:
                merges, counts = sh.train(vocab)
            want_m, want_c = O.train(data, vocab, off, mode=O.FIRST if first else O.LEXICAL)
            assert [list(m) for m in merges] == want_m.tolist(), "case %d merges" % ci
            assert list(counts) == want_c.tolist(), "case %d counts" % ci
            # the shards, concatenated, are the oracle's final stream
            st = O.State(data, off, mode=O.FIRST if first else O.LEXICAL)
            for i, (a, b) in enumerate(want_m):
                st.merge(int(a), int(b), 256 + i)
            mine = np.array([t & sh.idmask for t in sh.toks], dtype=np.int64)
            sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
            dist.all_gather(sizes, torch.tensor([len(mine)], dtype=torch.int64))
            pad = int(max(int(s.item()) for s in sizes))
            buf = torch.zeros(pad, dtype=torch.int64)
            buf[:len(mine)] = torch.from_numpy(mine)
            outs = [torch.zeros(pad, dtype=torch.int64) for _ in range(world)]
            dist.all_gather(outs, buf)
            whole = np.concatenate([o.numpy()[:int(s.item())] for o, s in zip(outs, sizes)])
            assert np.array_equal(whole, st.stream()[0].astype(np.int64)), "case %d stream" % ci
            # replicated table == oracle table (zero-count members aside)
            want_tab = {((a << 16) | b): c for (a, b), c in st.table_dict().items() if c}
            assert {k: v for k, v in sh.table.items() if v} == want_tab, "case %d table" % ci
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("batched", [False, True], ids=["per_merge", "batch_sequences"])
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_protocol_over_gloo(world, batched):
    """per_merge: one all-reduce of (m, adj, L, R) per merge.  batch_sequences: what the product runs -- several
    independent pairs per stream pass, one all-reduce of [header][m_j, ADJ][L_j, R_j rows] per sequence, validation of
    the batch against the one-at-a-time order, a second small all-reduce for the shard edges."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, batched)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in results:
        assert msg == "ok", "rank %d: %s" % (rank, msg)


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_first_tie_break_over_gloo(world):
    """The `first` tie-break (the reference CLI's default) on a sharded stream: one merge per exchange of count deltas
    plus, when counts tie, one small exchange of every rank's earliest tied pair (dist_model.Shard.argmax_first: the
    protocol of k_first_pos / k_first_publish / k_first_pick_global) -- against the oracle's rebuilt-table mode."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, False, True)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in results:
        assert msg == "ok", "rank %d: %s" % (rank, msg)

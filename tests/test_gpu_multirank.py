"""GPU tests of the sharded (multi-GPU) algorithm on ONE GPU: several contexts,
one per "rank", run the real kernels on their shard of the corpus; the test
plays the collective (sum of the exchange buffers) through the external-transport
C-ABI.  Everything except the ncclAllReduce call itself is the production path.
Results must equal the single-rank oracle on the whole corpus."""
import ctypes
import os

import numpy as np
import pytest

import mbpe
import oracle as O
from conftest import read_data

pytestmark = pytest.mark.gpu

_hip = None


def hip():
    global _hip
    if _hip is None:
        _hip = ctypes.CDLL("libamdhip64.so")
        _hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    return _hip


FUSED_MIN = 24
EXTRA_OPTS = {}       # further options set on every rank's context


@pytest.fixture(params=[2, 1000], autouse=True, ids=["fused", "scan_rewrite"])
def _both_batch_paths(request):
    """Every test runs with the fused pass forced on (k_fused_batch; matches between raw bytes counted in the pairs' cell
    blocks, folded into the L / R rows before the exchange) and forced off (rows only)."""
    global FUSED_MIN, EXTRA_OPTS
    FUSED_MIN = request.param
    EXTRA_OPTS = dict(EXTRA_OPTS, pair_cells=1 if request.param == 2 else 0)
    yield
    FUSED_MIN = 24
    EXTRA_OPTS = {k: v for k, v in EXTRA_OPTS.items() if k != "pair_cells"}


EXCHANGES = []        # (u32 words, merges committed before) of every exchange of the run in progress
# batch header of the exchange buffer: m_j (kBatchMax = 4096 words) and the ADJ block, whose row pitch is the largest batch
# a training selects -- 1,024 pairs with several ranks, so that the block stays 4 MB
HDRB = (4096 + 1024 * 1024 + 3) // 4 * 4


def _allreduce(trainers):
    bufs = [t.exchange_buffer() for t in trainers]
    n = bufs[0][1]
    assert all(b[1] == n for b in bufs)
    EXCHANGES.append((n, trainers[0].stats()["n_merges"]))
    total = np.zeros(n, dtype=np.uint32)
    tmp = np.zeros(n, dtype=np.uint32)
    for ptr, _ in bufs:
        assert hip().hipMemcpy(tmp.ctypes.data, ptr, n * 4, 2) == 0      # D2H
        total += tmp
    for ptr, _ in bufs:
        assert hip().hipMemcpy(ptr, total.ctypes.data, n * 4, 1) == 0    # H2D


def _train_sharded(data, cuts, vocab, chunk_off=None, decode_check=False):
    data = np.frombuffer(bytes(data), dtype=np.uint8)
    bounds = [0] + list(cuts) + [len(data)]
    R = len(bounds) - 1
    trainers = [mbpe.Trainer(0) for _ in range(R)]
    try:
        for r, t in enumerate(trainers):
            lo, hi = bounds[r], bounds[r + 1]
            t.comm_init_external(r, R)
            t.set_option("fused_min", FUSED_MIN)
            for k, v in EXTRA_OPTS.items():
                t.set_option(k, v)
            off = None
            if chunk_off is not None:
                sel = chunk_off[(chunk_off >= lo) & (chunk_off <= hi)]
                assert sel[0] == lo and sel[-1] == hi, "shards must hold whole chunks"
                off = (sel - lo).astype(np.uint64)
            t.load_corpus(data[lo:hi], off)
        codes = [t.train_begin(vocab) for t in trainers]
        assert all(c == mbpe.NEED_EXCHANGE for c in codes)
        n_begin = 0
        while codes[0] == mbpe.NEED_EXCHANGE:       # (the byte-pair tables; with `first`: then the earliest tied pairs)
            assert all(c == mbpe.NEED_EXCHANGE for c in codes)
            _allreduce(trainers)
            codes = [t.exchange_done() for t in trainers]
            n_begin += 1
        assert all(c == mbpe.OK for c in codes) and n_begin == (2 if EXTRA_OPTS.get("conflict_resolution", 1) == 0 else 1)
        del EXCHANGES[:]
        codes = [t.train_steps(vocab - 256) for t in trainers]
        while codes[0] == mbpe.NEED_EXCHANGE:
            assert all(c == mbpe.NEED_EXCHANGE for c in codes)
            _allreduce(trainers)
            codes = [t.exchange_done() for t in trainers]
        assert all(c == mbpe.OK for c in codes)
        results = [t.train_result() for t in trainers]
        # What crossed the "wire": a sequence exchanges its count deltas -- both headers + exactly the rows L_j, R_j
        # (lr_pitch(ids) cells each) of the pairs of its batch, not the whole 2 x 1024 x ids block -- and then the
        # shard edges (the small header alone).
        hdr = (2 + 8 * R + 3) // 4 * 4
        n_final = len(results[0][0])
        deltas = [(n, k) for n, k in EXCHANGES if n != hdr]
        for i, (n, k) in enumerate(deltas):
            pitch = (256 + k + 63) & ~63
            rows, rest = divmod(n - hdr - HDRB, 2 * pitch)
            committed = (deltas[i + 1][1] if i + 1 < len(deltas) else n_final) - k
            assert rest == 0 and 1 <= rows <= 1024, (n, k)
            assert committed <= rows, "a sequence exchanged %d pairs' rows and committed %d merges" % (rows, committed)

        streams = [t.stream()[0] for t in trainers]
        tables = [t.pairs_dict() for t in trainers]
        if decode_check:
            # bench.py's check of a sharded run: rank r's stream decodes to the corpus bytes that follow
            # those of the ranks before it (a merge straddling two shards leaves its token with the left one)
            import torch
            from mbpe import check as C
            dev = torch.device("cuda", 0)
            merges = results[0][0]
            lens = [C.decoded_length(t, merges, torch, dev) for t in trainers]
            assert sum(lens) == len(data)
            whole = torch.from_numpy(data.copy()).to(dev)
            start = 0
            for t, n in zip(trainers, lens):
                rt = C.decode_roundtrip(t, merges, whole[start:start + n], torch, dev)
                assert rt["ok"], rt
                start += n
        return results, streams, tables
    finally:
        for t in trainers:
            t.close()


def _check(data, cuts, vocab, chunk_off=None):
    want_m, want_c = O.train(data, vocab, chunk_off)
    results, streams, tables = _train_sharded(data, cuts, vocab, chunk_off)
    for m, c in results:                      # every rank takes the same decisions
        assert m.tolist() == want_m.tolist()
        assert c.tolist() == want_c.tolist()
    st = O.State(data, chunk_off)
    for i, (a, b) in enumerate(want_m):
        st.merge(int(a), int(b), 256 + i)
    assert np.array_equal(np.concatenate(streams), st.stream()[0])
    want_tab = {k: v for k, v in st.table_dict().items() if v}
    for tab in tables:                        # replicated pair table
        assert {k: v for k, v in tab.items() if v} == want_tab


def test_two_ranks_random_bytes():
    data = O.splitmix64_bytes(3, 40000).tobytes()
    _check(data, [17777], 256 + 60)


@pytest.mark.parametrize("R", [2, 3, 4])
def test_small_alphabet_many_cuts(R):
    rng = np.random.default_rng(R)
    for _ in range(3):
        n = int(rng.integers(50, 5000))
        data = rng.integers(97, 97 + int(rng.integers(1, 4)), size=n, dtype=np.uint8).tobytes()
        cuts = sorted(set(int(x) for x in rng.integers(1, n, size=R - 1)))
        while len(cuts) < R - 1:
            cuts = sorted(set(cuts + [int(rng.integers(1, n))]))
        _check(data, cuts, 256 + 30)


def test_runs_and_touching_matches_across_the_cut():
    # "abab" cut inside a match and between matches; runs of one byte over the cut
    _check(b"xy" + b"ab" * 301 + b"z", [2 + 301], 256 + 12)        # cut between a and b
    _check(b"xy" + b"ab" * 301 + b"z", [2 + 300], 256 + 12)        # cut between two matches
    _check(b"q" + b"a" * 1001 + b"r" + b"a" * 700, [400, 1100], 256 + 14)
    _check(b"a" * 1024, [512], 256 + 12)
    _check(b"a" * 1025, [1, 1024], 256 + 12)
    # runs of several tokens, longer than a tile, cut inside: several (t,t) members per batch, and the run
    # that ends the left shard continues into the right one
    data = b"a" * 3000 + b"b" * 2501 + b"ab" * 100 + b"c" * 1600 + b"b" * 777 + b"a" * 1300
    _check(data, [1000, 4100], 256 + 24)
    _check(data, [2999, 3001, 5501], 256 + 24)


def test_sharded_decode_check():
    # cuts inside matches: shards whose streams begin / end a byte off the range they were loaded with
    data = np.frombuffer(b"xy" + b"ab" * 3001 + b"z" + b"abc" * 500, dtype=np.uint8)
    _train_sharded(data, [2 + 3001, 5000], 256 + 12, decode_check=True)
    data = O.splitmix64_bytes(9, 300000)
    _train_sharded(data, [100001, 200003], 256 + 200, decode_check=True)


def test_empty_and_tiny_shards():
    data = b"abcabcabcabcabcabc"
    _check(data, [1], 256 + 6)
    _check(data, [len(data) - 1], 256 + 6)
    _check(data, [5, 6, 7], 256 + 6)


def test_chunked_shards_text():
    data = read_data("taylorswift.txt")[:50000]
    off = mbpe.presplit(O.GPT4_SPLIT_PATTERN, data)
    cuts = [int(off[len(off) // 3]), int(off[2 * len(off) // 3])]
    _check(data, cuts, 256 + 40, off)


def test_chunked_shards_barrier_layout():
    # chunk ends as barrier slots (the layout of vocabularies beyond 32,766 ids): a shard then ends on a barrier,
    # which its right neighbour sees as the token before its first one
    global EXTRA_OPTS
    EXTRA_OPTS = {"chunk_barrier": 1}
    try:
        data = read_data("taylorswift.txt")[:50000]
        off = mbpe.presplit(O.GPT4_SPLIT_PATTERN, data)
        _check(data, [int(off[len(off) // 3]), int(off[2 * len(off) // 3])], 256 + 40, off)
        rng = np.random.default_rng(31)
        data = rng.integers(97, 100, size=6000, dtype=np.uint8).tobytes()
        off = np.array(sorted(set([0, 6000] + [int(x) for x in rng.integers(1, 6000, size=900)])), dtype=np.uint64)
        _check(data, [int(off[100]), int(off[101]), int(off[500])], 256 + 30, off)      # (one shard of a single chunk)
    finally:
        EXTRA_OPTS = {}


def test_text_three_ranks():
    data = read_data("shakespeare.txt")[:200000]
    _check(data, [65536, 131073], 256 + 64)


def test_tie_heavy_text_three_ranks():
    # repeated text trained into the region of many equal counts: key-aware validation and the
    # selection's overflow path must take the same decisions on every rank
    base = read_data("shakespeare.txt")[:40000]
    data = (base + b"\n") * 8
    _check(data, [len(data) // 3 + 5, 2 * len(data) // 3 - 7], 256 + 900)


@pytest.mark.parametrize("seed", range(int(os.environ.get("MBPE_FUZZ_SHARDED", "3"))))
def test_fuzz_sharded(seed):
    # random corpora (small alphabets, repeated blocks, runs), random cuts, 2-4 ranks; a third of them chunked (shards hold
    # whole chunks), a quarter with the `first` tie-break, some long enough for the pair-count scan's whole iterations
    rng = np.random.default_rng(4200 + seed)
    for _ in range(6):
        n = int(rng.integers(20, 30000)) if rng.integers(0, 4) else int(rng.integers(30000, 150000))
        kind = int(rng.integers(0, 3))
        if kind == 0:
            data = rng.integers(97, 97 + int(rng.choice([2, 4, 26])), size=n, dtype=np.uint8)
        elif kind == 1:
            blk = rng.integers(97, 110, size=int(rng.integers(3, 200)), dtype=np.uint8)
            data = np.tile(blk, n // len(blk) + 1)[:n]
        else:
            vals = rng.integers(97, 100, size=max(n // 20, 1), dtype=np.uint8)
            data = np.repeat(vals, rng.integers(1, 60, size=len(vals)))[:n]
        R = int(rng.integers(2, 5))
        chunk_off = None
        if rng.integers(0, 3) == 0 and len(data) > 8:
            inner = np.unique(rng.integers(1, len(data), size=max(len(data) // int(rng.integers(2, 300)), R)))
            chunk_off = np.concatenate([[0], inner, [len(data)]]).astype(np.uint64)
            cuts = sorted(set(int(x) for x in rng.choice(inner, size=min(R - 1, len(inner)), replace=False)))
        else:
            cuts = sorted(set(int(x) for x in rng.integers(1, len(data), size=R - 1)))
        vocab = 256 + int(rng.integers(5, 120))
        if rng.integers(0, 4) == 0:
            _check_first(data.tobytes(), cuts, min(vocab, 256 + 40), chunk_off)
        else:
            _check(data.tobytes(), cuts, vocab, chunk_off)


def _check_first(data, cuts, vocab, chunk_off=None):
    """The `first` tie-break on a sharded stream: every rank must take the reference's choice -- among the pairs of
    maximal count the one whose first occurrence comes first in the WHOLE corpus (PairCount.h:65-74, :141-166), which
    may lie in another rank's shard or straddle two."""
    global EXTRA_OPTS
    want_m, want_c = O.train(data, vocab, chunk_off, mode=O.FIRST)
    EXTRA_OPTS = dict(EXTRA_OPTS, conflict_resolution=0)
    try:
        results, streams, tables = _train_sharded(data, cuts, vocab, chunk_off)
    finally:
        EXTRA_OPTS = {k: v for k, v in EXTRA_OPTS.items() if k != "conflict_resolution"}
    for m, c in results:
        assert m.tolist() == want_m.tolist()
        assert c.tolist() == want_c.tolist()
    st = O.State(data, chunk_off, mode=O.FIRST)
    for i, (a, b) in enumerate(want_m):
        st.merge(int(a), int(b), 256 + i)
    assert np.array_equal(np.concatenate(streams), st.stream()[0])


@pytest.mark.parametrize("R", [2, 3])
def test_first_mode_sharded(R):
    # ties everywhere (repeated blocks, small alphabets): the earliest occurrence decides, in whichever shard it lies
    rng = np.random.default_rng(900 + R)
    for case in range(6):
        n = int(rng.integers(200, 20000))
        if case % 2:
            blk = rng.integers(97, 110, size=int(rng.integers(3, 120)), dtype=np.uint8)
            data = np.tile(blk, n // len(blk) + 1)[:n]
        else:
            data = rng.integers(97, 97 + int(rng.choice([2, 3, 5, 12])), size=n, dtype=np.uint8)
        cuts = sorted(set(int(x) for x in rng.integers(1, len(data), size=R - 1)))
        while len(cuts) < R - 1:
            cuts = sorted(set(cuts + [int(rng.integers(1, len(data)))]))
        _check_first(data.tobytes(), cuts, 256 + int(rng.integers(5, 60)))


def test_first_mode_sharded_tie_across_the_cut():
    # "abab...": (a,b) and (b,a) tie; the first occurrence of the winner straddles the cut, or lies right behind it
    _check_first(b"ab" * 300, [1], 256 + 6)
    _check_first(b"ab" * 300, [2], 256 + 6)
    _check_first(b"cd" * 5 + b"ab" * 5 + b"cd" * 5 + b"ab" * 5, [10], 256 + 6)       # every "ab" lies in the right shard
    _check_first(b"xyxy" + b"ab" * 200 + b"xy" * 198, [3, 404], 256 + 8)


def test_first_mode_sharded_text_chunked():
    data = read_data("taylorswift.txt")[:40000]
    off = mbpe.presplit(O.GPT4_SPLIT_PATTERN, data)
    _check_first(data, [int(off[len(off) // 3]), int(off[2 * len(off) // 3])], 256 + 60, off)


def test_rccl_single_rank_communicator():
    # A 1-rank RCCL communicator through the production transport: exercises the
    # run-time RCCL binding, ncclCommInitRank and the per-merge ncclAllReduce on
    # the context's stream (the 8-GPU run itself is the driver's).
    data = O.splitmix64_bytes(12, 100000).tobytes()
    want_m, want_c = O.train(data, 256 + 80)
    with mbpe.Trainer(0) as t:
        t.comm_init(mbpe.comm_unique_id(), 0, 1)
        t.set_option("force_exchange", 1)
        t.set_option("fused_min", FUSED_MIN)
        t.load_corpus(np.frombuffer(data, dtype=np.uint8))
        assert t.train_begin(256 + 80) == mbpe.OK
        assert t.train_steps(80) == 80
        m, c = t.train_result()
    assert m.tolist() == want_m.tolist() and c.tolist() == want_c.tolist()


@pytest.mark.parametrize("seed", range(int(os.environ.get("MBPE_FUZZ_SHARDED_MEDIUM", "2"))))
def test_fuzz_sharded_medium(seed):
    """Shards of thousands of tiles (1-5 MiB over 2-4 ranks): the pair-count scan's whole iterations per shard, batches of
    hundreds of pairs and the rows they exchange, compactions per rank, matches and runs across the cuts."""
    rng = np.random.default_rng(52000 + seed)
    n = int(rng.integers(1 << 20, 5 << 20))
    kind = int(rng.integers(0, 4))
    if kind == 0:
        data = rng.integers(0, int(rng.choice([3, 7, 20, 256])), size=n, dtype=np.uint8)
    elif kind == 1:
        text = read_data("shakespeare.txt")
        s0 = int(rng.integers(0, len(text) - 300000))
        ln = int(rng.integers(1000, 300000))
        data = np.frombuffer((text[s0:s0 + ln] * (n // ln + 1))[:n], dtype=np.uint8).copy()
    elif kind == 2:
        vals = rng.integers(97, 103, size=max(n // 20, 1), dtype=np.uint8)
        data = np.repeat(vals, rng.integers(1, 40, size=len(vals)))[:n]
    else:
        data = O.splitmix64_bytes(int(rng.integers(1, 1 << 30)), n)
    R = int(rng.integers(2, 5))
    chunk_off = None
    if rng.integers(0, 3) == 0:
        inner = np.unique(rng.integers(1, len(data), size=len(data) // int(rng.integers(4, 4000))))
        chunk_off = np.concatenate([[0], inner, [len(data)]]).astype(np.uint64)
        cuts = sorted(set(int(x) for x in rng.choice(inner, size=R - 1, replace=False)))
        # (a chunk that starts with NUL and goes on like a number is ONE inert token in the reference, Tokenizer.h:86-93,
        #  and inert bytes here: same merges, another stream listing -- tests/test_gpu_fuzz.py; not what this test is about)
        data = np.where(data == 0, 1, data).astype(np.uint8)
    else:
        # (cuts next to the scan's units now and then: k x 32 KiB + a few bytes per shard)
        cuts = sorted(set(int(x) // 32768 * 32768 + int(rng.integers(0, 20)) if rng.integers(0, 2) else int(x)
                          for x in rng.integers(32768, len(data), size=R - 1)))
    _check(np.ascontiguousarray(data).tobytes(), cuts, 256 + int(rng.integers(40, 400)), chunk_off)

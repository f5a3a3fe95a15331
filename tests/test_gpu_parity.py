"""GPU parity tests: the HIP path, called through the C-ABI (libmbpe.so),
against the CPU oracle and the golden fixtures.  Integer work: bit-exact."""
import json
import os

import numpy as np
import pytest

import torch

import mbpe
import oracle as O
from conftest import GOLDEN, read_data, read_golden
from mbpe import check

pytestmark = pytest.mark.gpu

INDEX = json.load(open(os.path.join(GOLDEN, "index.json")))
LEXICAL_GOLDENS = sorted(k for k, v in INDEX.items() if v["mode"] == "lexical")


@pytest.fixture(scope="module")
def tr():
    t = mbpe.Trainer(0)
    yield t
    t.close()


def _input(spec):
    if spec.startswith("splitmix:"):
        _, seed, n = spec.split(":")
        return O.splitmix64_bytes(int(seed), int(n)).tobytes()
    return read_data(spec)


def _random_chunks(rng, n, mean):
    if n == 0:
        return np.array([0], dtype=np.uint64)
    cuts = np.unique(rng.integers(1, max(n, 2), size=max(n // mean, 1)))
    cuts = cuts[cuts < n]
    return np.concatenate([[0], cuts, [n]]).astype(np.uint64)


# ---------------------------------------------------------------- pair count
@pytest.mark.parametrize("n", [0, 1, 2, 3, 15, 16, 17, 31, 32, 33, 1023, 16384, 16385, 100003])
def test_pair_count_sizes(tr, n):
    rng = np.random.default_rng(n)
    data = rng.integers(0, 256, size=n, dtype=np.uint8)
    if n:
        data[0] = 1
    tr.load_corpus(data)
    assert np.array_equal(tr.pair_count_u8(), O.pair_count_u8(data))


def test_pair_count_lengths_around_block_edges(tr):
    """Corpus lengths around the scan kernels' units -- 16-byte vectors, 2-KiB wave blocks, 32-KiB workgroup iterations -- with
    0..20 bytes before and behind them: which loop counts the pair that straddles into the ragged tail depends on where the
    whole iterations end (a length of k x 32 KiB + 4..15 bytes once had it counted twice)."""
    rng = np.random.default_rng(5)
    base = rng.integers(97, 101, size=1 << 21, dtype=np.uint8)
    sizes = set()
    for k in (1, 2, 3, 4, 8, 16, 17, 32, 33, 48, 64):
        for unit in (2048, 16384, 32768):
            for d in list(range(-20, 21)) + [2047, 2049, 4099]:
                if 2 <= k * unit + d <= len(base):
                    sizes.add(k * unit + d)
    bad = []
    for n in sorted(sizes):
        data = base[:n]
        tr.load_corpus(data)
        want = np.bincount((data[:-1].astype(np.uint32) << 8) | data[1:], minlength=65536).astype(np.uint32)
        if not np.array_equal(tr.pair_count_u8(), want):
            bad.append(n)
    assert not bad, bad[:20]
    # one byte repeated, and nearly so: every segment of the fast loop wraps a 16-bit counter, is restored and recounted by the
    # generic loop -- which leaves the pair that straddles into the tail to the tail code
    skew = np.full(len(base), 97, dtype=np.uint8)
    skew[::4099] = 98
    for n in sorted(sizes)[::5]:
        for data in (np.full(n, 97, dtype=np.uint8), skew[:n]):
            tr.load_corpus(data)
            want = np.bincount((data[:-1].astype(np.uint32) << 8) | data[1:], minlength=65536).astype(np.uint32)
            if not np.array_equal(tr.pair_count_u8(), want):
                bad.append(("skewed", n))
    assert not bad, bad[:20]
    # (long enough for a workgroup's counter to wrap inside one segment: 256 workgroups x 4 iterations of 32 KiB)
    for d in (0, 3, 4, 9, 15, 16, 32768 + 7):
        n = (32 << 20) + d
        data = np.full(n, 97, dtype=np.uint8)
        tr.load_corpus(data)
        got = tr.pair_count_u8()
        if got[(97 << 8) | 97] != n - 1 or int(got.sum()) != n - 1:
            bad.append(("one byte", n, int(got[(97 << 8) | 97])))
    assert not bad, bad[:20]
    # the same lengths cut into chunks (the masked scan): a pair whose first byte ends a chunk does not count
    for n in sorted(sizes)[::7]:
        data = base[:n]
        off = _random_chunks(rng, n, int(rng.integers(2, 3000)))
        tr.load_corpus(data, off)
        keep = np.ones(n - 1, dtype=bool)
        keep[off[1:-1].astype(np.int64) - 1] = False
        want = np.bincount(((data[:-1].astype(np.uint32) << 8) | data[1:])[keep], minlength=65536).astype(np.uint32)
        if not np.array_equal(tr.pair_count_u8(), want):
            bad.append(("chunked", n))
    assert not bad, bad[:20]


def test_pair_count_random_1mib(tr):
    data = O.splitmix64_bytes(42, 1 << 20)
    tr.load_corpus(data)
    assert np.array_equal(tr.pair_count_u8(), O.pair_count_u8(data))


def test_pair_count_text_and_chunks(tr):
    data = read_data("shakespeare.txt")
    tr.load_corpus(data)
    assert np.array_equal(tr.pair_count_u8(), O.pair_count_u8(data))
    off = mbpe.presplit(O.GPT4_SPLIT_PATTERN, data)
    tr.load_corpus(data, off)
    assert np.array_equal(tr.pair_count_u8(), O.pair_count_u8(data, off))


def test_pair_count_16bit_counter_drain(tr):
    # one pair far beyond 65,535 per workgroup: exercises the epoch drain of the
    # packed 16-bit LDS counters
    n = 24 << 20
    data = np.full(n, 97, dtype=np.uint8)
    data[5::1000] = 98
    tr.load_corpus(data)
    got = tr.pair_count_u8()
    assert np.array_equal(got, O.pair_count_u8(data))
    assert int(got[(97 << 8) | 97]) > (1 << 24)


def test_pair_count_random_chunks(tr):
    rng = np.random.default_rng(11)
    for n in (50, 4097, 70001):
        data = rng.integers(97, 123, size=n, dtype=np.uint8)
        off = _random_chunks(rng, n, 5)
        tr.load_corpus(data, off)
        assert np.array_equal(tr.pair_count_u8(), O.pair_count_u8(data, off))


# ---------------------------------------------------------------- training
@pytest.mark.parametrize("name", LEXICAL_GOLDENS)
def test_train_golden_model_bytes(tr, name):
    meta = INDEX[name]
    data = _input(meta["input"])
    enc = meta["encoder"]
    off = None if enc == "basic" else mbpe.presplit(O.PATTERNS[enc], data)
    for lockstep in (0, 1):        # whole groups of sequences with every kernel variant / one sequence at a time, its kernels only
        tr.set_option("lockstep", lockstep)
        try:
            merges, counts, stats = tr.train_lexical(data, meta["vocab"], off)
        finally:
            tr.set_option("lockstep", -1)
        assert O.model_bytes(O.PATTERNS[enc], merges) == read_golden(name + ".model"), lockstep
        assert int(counts[0]) == meta["first_count"] and int(counts[-1]) == meta["last_count"]
        # (a merge of (t,t) removes fewer tokens than its overlapping-window count)
        assert stats["n_live"] == meta["final_len"] >= len(data) - int(counts.sum())


FIRST_GOLDENS = sorted(k for k, v in INDEX.items() if v["mode"] == "first")


# first_batches 1: pairs whose count no other candidate shares are merged in batches (position tie-breaks only where
# counts are equal); 0, the default: one merge per pass
@pytest.mark.parametrize("first_batches", [0, 1])
@pytest.mark.parametrize("name", FIRST_GOLDENS)
def test_train_first_mode_golden_model_bytes(tr, name, first_batches):
    # the reference CLI's default tie-break (PairCountInsertOrder), on the device
    meta = INDEX[name]
    data = _input(meta["input"])
    enc = meta["encoder"]
    off = None if enc == "basic" else mbpe.presplit(O.PATTERNS[enc], data)
    tr.set_option("first_batches", first_batches)
    try:
        merges, counts, stats = tr.train(data, meta["vocab"], off, conflict_resolution=0)
    finally:
        _defaults(tr)
    assert O.model_bytes(O.PATTERNS[enc], merges) == read_golden(name + ".model")
    want_m, want_c = O.train(data, meta["vocab"], off, mode=O.FIRST)
    assert counts.tolist() == want_c.tolist()


def test_train_first_mode_kats(tr):
    # small.txt @275: 7 merges, then no pair is left and the loop breaks (Tokenizer.h:586-588)
    m, c, _ = tr.train(read_data("small.txt"), 275, conflict_resolution=0)
    assert m.tolist() == [[98, 99], [256, 100], [257, 101], [258, 258], [97, 259], [260, 258], [261, 10]]
    # test.cpp:136-186 "abcbcde": (98,99) -> 256, (97,256) -> 257, (257,256) -> 258
    m, c, _ = tr.train(b"abcbcde", 259, conflict_resolution=0)
    assert m.tolist() == [[98, 99], [97, 256], [257, 256]]
    for data, vocab in ((b"", 300), (b"a", 300), (b"ab", 258), (b"aaaa", 262), (b"abababab", 260)):
        want_m, want_c = O.train(data, vocab, mode=O.FIRST)
        m, c, _ = tr.train(data, vocab, conflict_resolution=0)
        assert m.tolist() == want_m.tolist() and c.tolist() == want_c.tolist()


@pytest.mark.parametrize("first_batches", [0, 1])
@pytest.mark.parametrize("seed", range(6))
def test_train_first_mode_fuzz(tr, seed, first_batches):
    """Tie-heavy inputs (small alphabets, low counts) against the oracle's rebuilt-table first mode: both
    table layouts, chunked and not, holes and compactions in the stream."""
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(200, 30000))
    data = rng.integers(97, 97 + int(rng.integers(2, 12)), size=n, dtype=np.uint8)
    off = _random_chunks(rng, n, int(rng.integers(3, 40))) if seed % 2 else None
    vocab = 256 + int(rng.integers(20, 400))
    want_m, want_c = O.train(data, vocab, off, mode=O.FIRST)
    tr.set_option("dense_table", seed % 3 != 0)
    tr.set_option("compact_den", 40 if seed % 2 else 16)
    tr.set_option("first_batches", first_batches)
    try:
        m, c, _ = tr.train(data, vocab, off, conflict_resolution=0)
    finally:
        _defaults(tr)
    assert m.tolist() == want_m.tolist() and c.tolist() == want_c.tolist()


def _zipf_words(seed, n_words, vocab=3000):
    """Words of random letters with Zipf frequencies: pair counts spread over orders of magnitude, ties only inside
    frequent words (the tokens of a word that occurs nowhere else share its count)."""
    rng = np.random.default_rng(seed)
    words = [bytes(rng.integers(97, 123, size=int(rng.integers(2, 9)), dtype=np.uint8)) + b" " for _ in range(vocab)]
    w = 1.0 / np.arange(1, vocab + 1) ** 1.07
    idx = rng.choice(vocab, size=n_words, p=w / w.sum())
    return np.frombuffer(b"".join(words[i] for i in idx), dtype=np.uint8)


@pytest.mark.parametrize("chunked", [False, True])
def test_first_mode_batches_on_word_text(tr, chunked):
    """`first` with batches on text-like data (1 MB, 1,000 merges): several merges per pass between the tied ones,
    same merges and counts as the oracle's rebuilt-table loop, and the same final stream and pair table."""
    data = _zipf_words(5, 160000)
    off = mbpe.presplit(O.GPT4_SPLIT_PATTERN, data) if chunked else None
    vocab = 256 + 1000
    want_m, want_c = O.train(data, vocab, off, mode=O.FIRST)
    tr.set_option("first_batches", 1)
    try:
        tr.load_corpus(data, off)
        tr.set_option("conflict_resolution", 0)
        tr.train_begin(vocab)
        done = tr.train_steps(40)                   # (before the run may hand over to the one-merge-per-pass loop)
        st = tr.stats()
        done += tr.train_steps(vocab - 256 - 40)
        m, c = tr.train_result()
        toks = tr.stream()[0]
        table = {k: v for k, v in tr.pairs_dict().items() if v}
    finally:
        tr.load_corpus(data[:2])                    # (the tie-break can only be changed between trainings)
        tr.set_option("conflict_resolution", 1)
        _defaults(tr)
    assert done == len(want_m)
    assert m.tolist() == want_m.tolist() and c.tolist() == want_c.tolist()
    assert st["n_batches"] < 40                     # batches were formed
    ost = O.State(data, off)
    for i, (a, b) in enumerate(want_m):
        ost.merge(int(a), int(b), 256 + i)
    assert np.array_equal(toks, ost.stream()[0])
    assert table == {k: v for k, v in ost.table_dict().items() if v}
    ost.close()


def test_pair_count_rollback_paths(tr):
    """The segment checksum of the pair-count scan: a pair repeated far beyond 65,535 times per workgroup
    wraps a 16-bit counter, the segment is voided, restored from its snapshot and recounted; mixed with
    stretches that pass, so snapshots with a non-zero residue are restored too."""
    rng = np.random.default_rng(5)
    n = 96 << 20
    data = rng.integers(0, 256, size=n, dtype=np.uint8)
    data[0] = 1
    for lo in range(8 << 20, n, 24 << 20):          # 4 MiB stretches of one pair inside random bytes
        data[lo:lo + (4 << 20)] = 97
    tr.load_corpus(data)
    assert np.array_equal(tr.pair_count_u8(), O.pair_count_u8(data))
    off = _random_chunks(rng, n, 4096)
    tr.load_corpus(data, off)
    assert np.array_equal(tr.pair_count_u8(), O.pair_count_u8(data, off))
    data = np.full(64 << 20, 7, dtype=np.uint8)      # every segment of every workgroup wraps
    tr.load_corpus(data)
    got = tr.pair_count_u8()
    assert int(got[(7 << 8) | 7]) == len(data) - 1 and int(got.sum()) == len(data) - 1


def test_pair_count_beyond_31_bits_is_reported(tr):
    # the table keeps "ever inserted" in bit 31 of a cell, and the reference counts in int (PairCount.h:188): a
    # pair that occurs 2^31 times or more is an error, not a silently corrupted table
    n = (1 << 31) + 4096
    data = np.full(n, 7, dtype=np.uint8)
    tr.load_corpus(data)
    with pytest.raises(mbpe.MbpeError) as e:
        tr.train_begin(300)
    assert e.value.code == mbpe.ERR_OVERFLOW
    tr.load_corpus(b"abab")                      # the context stays usable
    assert tr.train_lexical(b"abab", 258)[0].tolist() == [[97, 98], [256, 256]]


def test_train_kat_small_and_aaaa(tr):
    m, c, _ = tr.train_lexical(read_data("small.txt"), 275)
    want = [[98, 99], [100, 101], [256, 257], [258, 258], [97, 259], [258, 10], [260, 261]] + [[97, 98]] * 12
    assert m.tolist() == want and c.tolist()[7:] == [0] * 12
    m, c, _ = tr.train_lexical(b"aaaa", 262)
    assert m.tolist() == [[97, 97], [256, 256], [97, 97], [97, 97], [97, 97], [97, 97]]
    assert c.tolist() == [3, 1, 0, 0, 0, 0]


@pytest.mark.parametrize("data,vocab", [(b"", 300), (b"a", 300), (b"ab", 258), (b"\x00123abcabc", 260),
                                        (b"\x00abcabc", 258), (b"abcbcde", 259)])
def test_train_tiny_inputs(tr, data, vocab):
    want_m, want_c = O.train(data, vocab)
    m, c, _ = tr.train_lexical(data, vocab)
    assert m.tolist() == want_m.tolist() and c.tolist() == want_c.tolist()


DEFAULTS = {"compact_den": 16, "batch": 16, "multi_merge": 1, "max_batch": 4096, "fused_min": 24, "hier_argmax": -1,
            "dense_table": -1, "threshold_select": 1, "sel_cap": 8192, "chunk_barrier": -1, "first_batches": 0, "byte_table": 1,
            "wide_from": -1, "lockstep": -1, "pair_cells": -1}


def _defaults(tr):
    for k, v in DEFAULTS.items():
        tr.set_option(k, v)


def _step_parity(tr, data, off, vocab, stride=1, **opts):
    """After every `stride` merges (1 = every step; larger values let several independent
    merges share one stream pass): chosen pairs, counts, live stream, chunk ends, pair table."""
    # (small corpora run in lockstep by default -- the host waits for every selection and enqueues only that sequence's
    #  kernels; large ones enqueue whole groups with every kernel variant: the cases alternate between the two)
    opts.setdefault("lockstep", (len(data) + vocab) & 1)
    # (likewise the count of a match between two raw bytes: one atomic on the pair's cell block -- large streams -- or two
    #  on its L and R rows)
    opts.setdefault("pair_cells", ((len(data) + vocab) >> 1) & 1)
    for k, v in opts.items():
        tr.set_option(k, v)
    try:
        st = O.State(data, off)
        tr.load_corpus(data, off)
        tr.train_begin(vocab)
        i = 0
        rng = np.random.default_rng(len(data) + vocab)
        while i < vocab - 256:
            want = 1 if stride == 1 else int(rng.integers(1, stride + 1))
            want = min(want, vocab - 256 - i)
            tops = []
            for j in range(want):
                top = st.top()
                if top is None:
                    break
                tops.append(top)
                st.merge(top[0], top[1], 256 + i + j)
            done = tr.train_steps(want)
            if not tops:
                assert done == 0 or len(tr.train_result()[0]) == 0
                break
            assert done == len(tops)
            m, c = tr.train_result()
            for j, top in enumerate(tops):
                assert (int(m[i + j][0]), int(m[i + j][1]), int(c[i + j])) == top, "step %d" % (i + j)
            i += len(tops) - 1
            want_toks, want_clen = st.stream()
            toks, ends = tr.stream()
            assert np.array_equal(toks, want_toks), "stream differs at step %d" % i
            if off is not None:
                pos = np.cumsum(want_clen[want_clen > 0]).astype(np.int64) - 1
                want_ends = np.zeros(len(want_toks), dtype=np.uint8)
                want_ends[pos] = 1
                assert np.array_equal(ends, want_ends), "chunk ends differ at step %d" % i
            want_tab = {k_: v_ for k_, v_ in st.table_dict().items() if v_}
            got_tab = {k_: v_ for k_, v_ in tr.pairs_dict().items() if v_}
            assert got_tab == want_tab, "pair table differs at step %d" % i
            # (every kernel that rewrites a tile keeps its live tokens in the tile's first slots: the fused pass relies on it)
            assert check.tiles_in_prefix_form(tr, torch, torch.device("cuda", 0)), "a tile with a hole before a token at step %d" % i
            i += 1
    finally:
        _defaults(tr)


@pytest.mark.parametrize("seed", range(6))
def test_step_parity_small_alphabet(tr, seed):
    # 2-4 symbols: long runs (a==b merges), touching matches ("abab"), dense holes
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1, 6000))
    data = rng.integers(97, 97 + int(rng.integers(1, 5)), size=n, dtype=np.uint8)
    _step_parity(tr, data, None, 256 + 40, batch=1, compact_den=int(rng.choice([0, 2, 3, 8, 50, 100000])))


@pytest.mark.parametrize("seed", range(6))
def test_step_parity_chunked(tr, seed):
    rng = np.random.default_rng(100 + seed)
    n = int(rng.integers(2, 9000))
    data = rng.integers(97, 97 + int(rng.integers(1, 6)), size=n, dtype=np.uint8)
    off = _random_chunks(rng, n, int(rng.integers(2, 12)))
    _step_parity(tr, data, off, 256 + 40, batch=1, compact_den=int(rng.choice([0, 2, 3, 8, 50, 100000])))


def test_step_parity_runs_across_tiles(tr):
    # runs of one byte far longer than a 512-slot tile, odd and even lengths,
    # ending exactly on / just before / just after tile and lane boundaries
    parts = [b"a" * 5001, b"b", b"a" * 4096, b"cc", b"a" * 2047, b"d", b"a" * 2049, b"ab" * 3000,
             b"a" * 511, b"e", b"a" * 513, b"f", b"a" * 7, b"g", b"a" * 8, b"h", b"a" * 9, b"ba" * 700]
    _step_parity(tr, b"".join(parts), None, 256 + 24, batch=1, compact_den=3)


def test_step_parity_text_gpt4(tr):
    data = read_data("taylorswift.txt")[:60000]
    off = mbpe.presplit(O.GPT4_SPLIT_PATTERN, data)
    _step_parity(tr, data, off, 256 + 48, batch=1)


@pytest.mark.parametrize("batch,den", [(1, 1000), (7, 16), (64, 0), (256, 8)])
def test_train_options_do_not_change_results(tr, batch, den):
    data = read_data("taylorswift.txt")
    want_m, want_c = O.train(data, 400)
    tr.set_option("batch", batch)
    tr.set_option("compact_den", den)
    try:
        m, c, st = tr.train_lexical(data, 400)
    finally:
        tr.set_option("batch", 16)
        tr.set_option("compact_den", 16)
    assert m.tolist() == want_m.tolist() and c.tolist() == want_c.tolist()
    if den == 1:
        assert st["n_compactions"] > 0


def test_train_is_deterministic(tr):
    data = O.splitmix64_bytes(5, 300000)
    a = tr.train_lexical(data, 700)
    b = tr.train_lexical(data, 700)
    assert a[0].tolist() == b[0].tolist() and a[1].tolist() == b[1].tolist()


def test_vocab_limits(tr):
    data = b"hello world hello world"
    tr.load_corpus(data)
    with pytest.raises(mbpe.MbpeError) as e:
        tr.train_begin(255)
    assert e.value.code == mbpe.ERR_ARG
    with pytest.raises(mbpe.MbpeError) as e:
        tr.train_begin((1 << 24) + 1)           # beyond MBPE_MAX_VOCAB_WIDE
    assert e.value.code == mbpe.ERR_VOCAB
    tr.set_option("conflict_resolution", 0)     # the 32-bit continuation is lexical only
    try:
        with pytest.raises(mbpe.MbpeError) as e:
            tr.train_begin(65535)
        assert e.value.code == mbpe.ERR_VOCAB
    finally:
        tr.set_option("conflict_resolution", 1)
    tr.train_begin(65535)                       # lexical: trains (tests/test_gpu_wide.py)
    # (chunked corpora: tests/test_gpu_barrier.py)


@pytest.mark.parametrize("dense", [0, 1])
def test_large_vocab_many_merges(tr, dense):
    # vocab 4096 on 256 KiB of random bytes: table growth (hashed layout) + thousands of steps
    data = O.splitmix64_bytes(9, 1 << 18)
    want_m, want_c = O.train(data, 4096)
    tr.set_option("dense_table", dense)
    try:
        m, c, _ = tr.train_lexical(data, 4096)
    finally:
        _defaults(tr)
    assert m.tolist() == want_m.tolist() and c.tolist() == want_c.tolist()


def test_vocab_beyond_the_dense_table(tr):
    # vocab > 32,768: the hashed pair table is the only layout; more merges than the text has pairs
    data = read_data("taylorswift.txt")[:30000]
    want_m, want_c = O.train(data, 40000)
    m, c, _ = tr.train_lexical(data, 40000)
    assert m.tolist() == want_m.tolist() and c.tolist() == want_c.tolist()


@pytest.mark.parametrize("hier", [0, 1])
def test_argmax_variants_agree(tr, hier):
    # full-scan argmax vs the hierarchical one (block upper bounds), incl. table growth
    data = O.splitmix64_bytes(21, 1 << 17)
    want_m, want_c = O.train(data, 3000)
    tr.set_option("hier_argmax", hier)
    tr.set_option("batch", 16)
    try:
        m, c, _ = tr.train_lexical(data, 3000)
    finally:
        tr.set_option("hier_argmax", -1)
        tr.set_option("batch", 16)
    assert m.tolist() == want_m.tolist() and c.tolist() == want_c.tolist()


def test_hier_argmax_text_zero_counts(tr):
    tr.set_option("hier_argmax", 1)
    try:
        m, c, _ = tr.train_lexical(read_data("small.txt"), 275)
        want = [[98, 99], [100, 101], [256, 257], [258, 258], [97, 259], [258, 10], [260, 261]] + [[97, 98]] * 12
        assert m.tolist() == want
        data = read_data("taylorswift.txt")
        off = mbpe.presplit(O.GPT4_SPLIT_PATTERN, data)
        want_m, want_c = O.train(data, 512, off)
        m, c, _ = tr.train_lexical(data, 512, off)
        assert m.tolist() == want_m.tolist() and c.tolist() == want_c.tolist()
    finally:
        tr.set_option("hier_argmax", -1)


# fused_min 2: every multi-pair batch takes the fused pass (k_fused_batch, output in the other
# token buffer); 1000: never (k_scan_batch + k_rewrite_marked)
# dense_table 1: one cell per possible pair (vocab <= 32,768); 0: the hashed table of larger vocabularies
@pytest.mark.parametrize("dense", [0, 1])
@pytest.mark.parametrize("fused_min", [2, 1000])
@pytest.mark.parametrize("seed", range(8))
def test_batched_merges_parity_random_bytes(tr, seed, fused_min, dense):
    # several independent merges per stream pass: compare stream + table at random strides
    rng = np.random.default_rng(500 + seed)
    n = int(rng.integers(2000, 60000))
    data = rng.integers(0, int(rng.choice([8, 40, 256])), size=n, dtype=np.uint8)
    data[0] = max(int(data[0]), 1)
    _step_parity(tr, data, None, 256 + 120, stride=int(rng.choice([3, 7, 16, 40, 100])),
                 compact_den=int(rng.choice([0, 3, 8, 50])), fused_min=fused_min, dense_table=dense)


@pytest.mark.parametrize("dense", [0, 1])
@pytest.mark.parametrize("fused_min", [2, 1000])
@pytest.mark.parametrize("seed", range(4))
def test_batched_merges_parity_chunked_text(tr, seed, fused_min, dense):
    data = read_data("taylorswift.txt")[seed * 20000:seed * 20000 + 40000]
    off = mbpe.presplit(O.GPT4_SPLIT_PATTERN if seed % 2 else O.GPT2_SPLIT_PATTERN, data)
    _step_parity(tr, data, off, 256 + 150, stride=25, fused_min=fused_min, dense_table=dense)


@pytest.mark.parametrize("fused_min", [2, 1000])
def test_batched_parity_sparse_stream(tr, fused_min):
    # few distinct tokens and no compaction: long hole runs, empty lanes and tiles, matches that
    # straddle lanes and tiles, touching matches ("abab")
    rng = np.random.default_rng(9)
    data = np.frombuffer(b"abcdabab" * 3000 + bytes(rng.integers(97, 103, size=30000, dtype=np.uint8)),
                         dtype=np.uint8)
    _step_parity(tr, data, None, 256 + 200, stride=64, compact_den=0, fused_min=fused_min)


def test_abandoned_fused_pass_over_empty_tiles(tr):
    """A fused pass that validation abandons is redone by k_rewrite_marked over EVERY tile -- empty ones included.  A run
    of 64 tiles of one byte halves with every merge of its (t,t) pair; without compaction the tenth merge (64 tokens, one
    per tile: 63 overlapping pairs) leaves half of those tiles without a live token, and the text behind the run keeps
    producing batches that validation cuts."""
    text = read_data("shakespeare.txt")[3000:27000]
    data = np.frombuffer(b"a" * (512 * 64) + text, dtype=np.uint8)
    vocab = 256 + 700
    want_m, want_c = O.train(data, vocab)
    k_empty = next(k for k in range(len(want_c)) if want_c[k] == 63 and want_m[k][0] == want_m[k][1] and want_m[k][0] >= 256)
    for k, v in {"compact_den": 0, "fused_min": 2, "lockstep": 0}.items():
        tr.set_option(k, v)
    try:
        tr.load_corpus(data)
        tr.train_begin(vocab)
        assert tr.train_steps(k_empty + 1) == k_empty + 1
        before = tr.stats()
        assert before["n_compactions"] == 0 and before["n_live"] <= len(text) + 32      # (32 tokens of the run are left)
        assert tr.train_steps(vocab - 256 - (k_empty + 1)) == vocab - 256 - (k_empty + 1)
        after = tr.stats()
        m, c = tr.train_result()
    finally:
        _defaults(tr)
    assert after["n_fused_dropped"] > before["n_fused_dropped"], (before["n_fused_dropped"], after["n_fused_dropped"])
    assert after["n_compactions"] == 0
    assert m.tolist() == want_m.tolist() and c.tolist() == want_c.tolist()


def test_batched_vs_single_merge_mode(tr):
    data = O.splitmix64_bytes(77, 1 << 19)
    want_m, want_c = O.train(data, 256 + 400)
    for mode, mb, fm, ts in ((0, 16, 24, 1), (1, 16, 24, 1), (1, 64, 2, 1), (1, 64, 1000, 0), (1, 2, 2, 0),
                             (1, 5, 1000, 1), (1, 64, 24, 0)):
        tr.set_option("multi_merge", mode)
        tr.set_option("max_batch", mb)
        tr.set_option("fused_min", fm)
        tr.set_option("threshold_select", ts)
        try:
            m, c, st = tr.train_lexical(data, 256 + 400)
        finally:
            _defaults(tr)
        assert m.tolist() == want_m.tolist() and c.tolist() == want_c.tolist(), (mode, mb, fm, ts)


@pytest.mark.parametrize("name", LEXICAL_GOLDENS)
def test_train_golden_fused_pass(tr, name):
    # (and the hashed pair table; the default run above uses the dense one)
    meta = INDEX[name]
    data = _input(meta["input"])
    enc = meta["encoder"]
    off = None if enc == "basic" else mbpe.presplit(O.PATTERNS[enc], data)
    tr.set_option("fused_min", 2)
    tr.set_option("dense_table", 0)
    try:
        merges, counts, stats = tr.train_lexical(data, meta["vocab"], off)
    finally:
        _defaults(tr)
    assert O.model_bytes(O.PATTERNS[enc], merges) == read_golden(name + ".model")
    assert stats["n_live"] == meta["final_len"]


@pytest.mark.parametrize("chunked", [False, True])
def test_tie_heavy_text_large_vocab(tr, chunked):
    """A text repeated 16 times, trained far into the region where thousands of pairs share a count:
    candidate lists overflow (threshold found among the block bounds), new pairs tie with the
    candidates (validation has to compare keys), batch sizes adapt.  Merges, counts, final stream and
    pair table against the oracle."""
    base = read_data("shakespeare.txt")[:150000]
    data = (base + b"\n") * 16
    off = mbpe.presplit(O.GPT4_SPLIT_PATTERN, data) if chunked else None
    vocab = 256 + (2500 if chunked else 4000)
    want_m, want_c = O.train(data, vocab, off)
    tr.set_option("sel_cap", 256)                 # a short candidate list: ties overflow it
    m, c, st = tr.train_lexical(data, vocab, off)
    tr.set_option("sel_cap", DEFAULTS["sel_cap"])
    assert m.tolist() == want_m.tolist() and c.tolist() == want_c.tolist()
    ost = O.State(data, off)
    for i, (a, b) in enumerate(want_m):
        ost.merge(int(a), int(b), 256 + i)
    toks, _ = tr.stream()
    assert np.array_equal(toks, ost.stream()[0])
    assert {k: v for k, v in tr.pairs_dict().items() if v} == {k: v for k, v in ost.table_dict().items() if v}
    ost.close()
    assert st["n_batches"] < len(m) // 4          # several merges per pass even here
    if not chunked:
        assert st["n_sel_retry"] > 0              # the overflow -> block bounds -> gather path was taken


def test_one_giant_run(tr):
    """16 MiB of one byte: every merge is a (t,t) merge over a run that spans all 32,768 tiles (the
    run length before every tile comes from a segmented scan, not from walking the summaries back;
    1 GiB takes 0.06 s)."""
    data = np.full(16 << 20, 97, dtype=np.uint8)
    want_m, want_c = O.train(data, 256 + 24)
    m, c, st = tr.train_lexical(data, 256 + 24)
    assert m.tolist() == want_m.tolist() and c.tolist() == want_c.tolist()
    ost = O.State(data)
    for i, (a, b) in enumerate(want_m):
        ost.merge(int(a), int(b), 256 + i)
    assert np.array_equal(tr.stream()[0], ost.stream()[0])
    ost.close()
    # runs interrupted at odd places, across many tiles
    rng = np.random.default_rng(3)
    data = np.full(8 << 20, 97, dtype=np.uint8)
    data[rng.integers(0, len(data), size=40)] = 98
    want_m, want_c = O.train(data, 256 + 30)
    m, c, _ = tr.train_lexical(data, 256 + 30)
    assert m.tolist() == want_m.tolist() and c.tolist() == want_c.tolist()


@pytest.mark.parametrize("fused_min", [2, 1000])
def test_several_long_runs_in_one_batch(tr, fused_min):
    """Runs of five different bytes, many of them longer than a tile and of every parity, between
    stretches of small-alphabet noise: the pairs (a,a) .. (e,e) are candidates together, so batches
    hold several (t,t) members (each with its own stand-in id, one run scan for all of them), and
    their merged tokens form runs again.  Merges, counts, final stream and pair table."""
    _defaults(tr)
    rng = np.random.default_rng(11)
    parts = []
    for _ in range(160):
        parts.append(np.full(int(rng.integers(1, 2400)), 97 + int(rng.integers(0, 5)), dtype=np.uint8))
        parts.append(rng.integers(97, 103, size=int(rng.integers(0, 40)), dtype=np.uint8))
    data = np.concatenate(parts)
    vocab = 256 + 120
    want_m, want_c = O.train(data, vocab)
    tr.set_option("fused_min", fused_min)
    m, c, st = tr.train_lexical(data, vocab)
    tr.set_option("fused_min", DEFAULTS["fused_min"])
    assert m.tolist() == want_m.tolist() and c.tolist() == want_c.tolist()
    ost = O.State(data)
    for i, (a, b) in enumerate(want_m):
        ost.merge(int(a), int(b), 256 + i)
    assert np.array_equal(tr.stream()[0], ost.stream()[0])
    assert {k: v for k, v in tr.pairs_dict().items() if v} == {k: v for k, v in ost.table_dict().items() if v}
    ost.close()
    assert st["n_batches"] < len(m)               # (t,t) pairs did share passes


def test_large_corpus_properties(tr):
    """256 MiB of SplitMix64 bytes, 600 merges (tens of stream passes, compaction, table growth):
    size-independent properties instead of an oracle run --
      * chosen counts never increase (new pairs are bounded by the pair they came from),
      * expanding the final stream through the merges gives back the corpus' byte histogram
        and length (nothing lost, nothing duplicated),
      * the first merges equal the oracle's on the pair table of the whole corpus."""
    n = 256 << 20
    data = O.splitmix64_bytes(42, n)
    tr.set_option("compact_den", 400)      # compact when holes reach 1/400 of the slots
    try:
        m, c, st = tr.train_lexical(data, 256 + 600)
    finally:
        tr.set_option("compact_den", 16)
    assert len(m) == 600 and np.all(np.diff(c) <= 0)
    table = O.pair_count_u8(data)
    first = int(np.argmax(table))                      # ties resolve to the smallest key, like the trainer
    assert (int(m[0][0]), int(m[0][1])) == (first >> 8, first & 0xFF) and int(c[0]) == int(table[first])
    toks, _ = tr.stream()
    assert st["n_live"] == len(toks) and st["n_compactions"] >= 1
    occ = np.bincount(toks, minlength=256 + 600).astype(np.int64)
    hist = np.zeros((256 + 600, 256), dtype=np.int64)   # byte histogram of every token
    hist[np.arange(256), np.arange(256)] = 1
    for k, (a, b) in enumerate(m):
        hist[256 + k] = hist[int(a)] + hist[int(b)]
    assert np.array_equal(occ @ hist, np.bincount(data, minlength=256).astype(np.int64))


# For every hash multiplier of the batch lookup table (kernels.hip: kHashMul, 8192 buckets of two keys) three byte
# pairs that fall into ONE bucket under it -- six distinct bytes each, all 48 distinct: independent of each other.
_COLLIDING = [((1, 2), (43, 180), (60, 57)), ((3, 4), (22, 255), (71, 40)), ((5, 89), (54, 98), (103, 107)),
              ((6, 7), (29, 20), (52, 33)), ((9, 181), (16, 106), (23, 31)), ((8, 226), (10, 196), (12, 166)),
              ((11, 62), (34, 159), (90, 39)), ((13, 17), (41, 109), (69, 201))]
_HASH_MUL = [2531, 40503, 10007, 60493, 25013, 7919, 52361, 33391]


def test_colliding_triples_really_collide():
    for mul, triple in zip(_HASH_MUL, _COLLIDING):
        assert len({(b * mul + a) % 8192 for a, b in triple}) == 1


@pytest.mark.parametrize("n_triples,byte_table", [(1, 0), (3, 0), (7, 0), (8, 0), (8, 1)])
def test_batch_lookup_switches_hash_multiplier(tr, n_triples, byte_table):
    """The top pairs of the corpus are chosen so that, rank after rank, the third key of a bucket appears under the
    first n_triples hash multipliers: the selection has to drop those and the stream kernels run with another one
    (all eight used up: the batch ends there).  Pairs of raw bytes normally go through the byte x byte table, which
    has no buckets to fill ("byte_table" 1: one batch holds all 24); "byte_table" 0 keeps them in the hashed one.
    Merges, counts, stream and pair table against the oracle."""
    used = {x for t in _COLLIDING for p in t for x in p}
    seps = [x for x in range(1, 256) if x not in used]
    rng = np.random.default_rng(11)
    parts = []
    rank = 0
    for t in _COLLIDING[:n_triples]:
        for a, b in t:
            n = 4000 - 7 * rank                       # distinct, descending counts: the rank order is fixed
            rank += 1
            s = rng.choice(seps, size=n)
            blk = np.empty((n, 3), dtype=np.uint8)
            blk[:, 0], blk[:, 1], blk[:, 2] = a, b, s
            parts.append(blk.reshape(-1))
    data = np.concatenate(parts)
    vocab = 256 + 3 * n_triples + 20
    want_m, want_c = O.train(data, vocab)
    assert [tuple(x) for x in want_m[:3 * n_triples].tolist()] == [p for t in _COLLIDING[:n_triples] for p in t]
    tr.set_option("fused_min", 2)
    tr.set_option("byte_table", byte_table)
    try:
        tr.load_corpus(data)
        tr.train_begin(vocab)
        first = tr.train_sequences(1)
        st = tr.stats()
        tr.train_steps(vocab - 256 - first)
        m, c = tr.train_result()
        toks = tr.stream()[0]
        table = {k: v for k, v in tr.pairs_dict().items() if v}
    finally:
        _defaults(tr)
    assert m.tolist() == want_m.tolist() and c.tolist() == want_c.tolist()
    if n_triples < 8 or byte_table:
        assert first >= 3 * n_triples and st["cut_bucket"] == 0      # one batch held them all
    else:
        assert first == 23 and st["cut_bucket"] == 1                 # the 24th pair has no multiplier left
    ost = O.State(data, None)
    for i, (a, b) in enumerate(want_m):
        ost.merge(int(a), int(b), 256 + i)
    assert np.array_equal(toks, ost.stream()[0])
    assert table == {k: v for k, v in ost.table_dict().items() if v}
    ost.close()


# ---- the fused pass on data without a frequent pair ------------------------------------------------------
# Uniform bytes: the top count stays below one occurrence per 8192 live tokens, so the plain instantiations of the
# stream kernels run (no LDS delta cache), with a dozen matches per 512-slot tile: stream, chunk ends and the whole
# pair table against the oracle, one chunk and many.
@pytest.mark.parametrize("chunked", [False, True])
def test_fused_pass_uniform_bytes(tr, chunked):
    rng = np.random.default_rng(77 + chunked)
    n = 3 << 20
    data = rng.integers(0, 256, size=n, dtype=np.uint8)
    data[0] = max(int(data[0]), 1)
    data[5000:5400] = 97                   # a run: a (t,t) member among the batch pairs
    if chunked:
        data[data == 0] = 1                # (a NUL-led chunk that parses as a number is one token in the oracle's stream and
    off = _random_chunks(rng, n, 40) if chunked else None      #  inert bytes in the library's: Tokenizer.h:86-93)
    _step_parity(tr, data, off, 256 + 260, stride=90, fused_min=2)


def test_batch_option_changed_after_begin_on_flat_counts(tr):
    """The host does not launch the frequent-pair (HOT) instantiations of the stream kernels while it can prove that no
    pair becomes frequent (one occurrence per 8192 live tokens) within the merges of the next group of sequences.  The
    proof has to hold for the group that really runs.  "batch" 1 at mbpe_train_begin: the bound covers 1024 merges;
    then "batch" 64 and "max_batch" 32: the first group is 64 sequences of up to 32 merges.  On a flat-count corpus
    (uniform over 116 symbols: the top pair occurs once per 9434 tokens) the threshold is crossed after ~1250 merges:
    inside that group, beyond the bound.  (Without the per-group bound the pass is skipped; the plain instantiation
    now also flags that: kErrHotSkipped.)"""
    rng = np.random.default_rng(5)
    S, n = 116, 1_500_000
    data = rng.integers(1, S + 1, size=n, dtype=np.uint8)
    pairs = np.bincount(data[:-1].astype(np.int64) * 256 + data[1:], minlength=65536)
    top = int(pairs.max())
    assert top * 9216 < n, "not frequent within the 1024 merges the first bound covers"
    assert top * 8192 >= n - 1900 * top, "... but frequent within the first 64 x 32 merges"
    vocab = 256 + 2300
    want_m, want_c = O.train(data, vocab)
    try:
        tr.set_option("batch", 1)
        tr.load_corpus(data)
        tr.train_begin(vocab)
        tr.set_option("batch", 64)
        tr.set_option("max_batch", 32)
        assert tr.train_steps(vocab - 256) == vocab - 256
        m, c = tr.train_result()
    finally:
        _defaults(tr)
    assert m.tolist() == want_m.tolist() and c.tolist() == want_c.tolist()


@pytest.mark.parametrize("mode", ["first", "lexical"])
def test_one_merge_loop_ends_with_the_table(tr, mode):
    """The one-merge-per-pass loop (`first` by default, "multi_merge" 0) on a corpus whose pairs run out long before the
    vocabulary is full: `first` stops where the reference's loop breaks (Tokenizer.h:586-588) and reports only the
    real merges; `lexical` keeps choosing the last, zero-count pair (PairCountLexicalOrder never erases) without a pass
    per choice.  Either way no stream pass runs once the best count is 0."""
    data = np.frombuffer(b"abcabdabcabd" * 40 + b"xyz", dtype=np.uint8)
    vocab = 256 + 3000
    want_m, want_c = O.train(data, vocab, mode=O.FIRST if mode == "first" else O.LEXICAL)
    real = int(np.count_nonzero(want_c))
    assert 0 < real < 40
    tr.set_option("multi_merge", 0)
    tr.set_option("time_kernels", 1)          # (then mbpe_stats.merge_launches counts the stream passes)
    try:
        tr.load_corpus(data)
        tr.set_option("conflict_resolution", 0 if mode == "first" else 1)
        tr.train_begin(vocab)
        done = tr.train_steps(vocab - 256)
        m, c = tr.train_result()
        st = tr.stats()
    finally:
        _defaults(tr)
        tr.set_option("time_kernels", 0)
        tr.set_option("conflict_resolution", 1)
    assert m.tolist() == want_m.tolist() and c.tolist() == want_c.tolist()
    assert done == len(want_m)
    assert st["merge_launches"] <= real + 2 * DEFAULTS["batch"]   # (passes: the real merges and at most a group behind them)

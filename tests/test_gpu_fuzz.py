"""Differential fuzz: random corpora (random alphabets, repeated blocks, long runs, repeated text
slices, NUL-heavy bytes), random chunkings and random tuning options (batch size, fused pass on/off,
table layout, selection kernel, compaction, host round trips, chunk-end convention) -- merges, counts, final stream and
pair table against the CPU oracle."""
import numpy as np
import pytest

import torch

import mbpe
import oracle as O
from conftest import read_data
from mbpe import check

pytestmark = pytest.mark.gpu

DEFAULTS = {"compact_den": 16, "batch": 16, "multi_merge": 1, "max_batch": 4096, "fused_min": 24,
            "dense_table": -1, "threshold_select": 1, "sel_cap": 8192, "chunk_barrier": -1, "lockstep": -1, "pair_cells": -1}


def _case(rng, text):
    kind = int(rng.integers(0, 5))
    n = int(rng.integers(1, 120000))
    if kind == 0:
        data = rng.integers(0, int(rng.choice([2, 3, 5, 17, 64, 256])), size=n, dtype=np.uint8)
    elif kind == 1:      # repeated blocks: many equal counts
        blk = rng.integers(97, 97 + int(rng.integers(2, 26)), size=int(rng.integers(3, 400)), dtype=np.uint8)
        data = np.tile(blk, n // len(blk) + 1)[:n]
    elif kind == 2:      # long runs: (t,t) merges
        vals = rng.integers(97, 101, size=max(n // 50, 1), dtype=np.uint8)
        data = np.repeat(vals, rng.integers(1, 100, size=len(vals)))[:n]
    elif kind == 3:      # a text slice repeated
        s = int(rng.integers(0, len(text) - 5000))
        l = int(rng.integers(50, 5000))
        data = np.frombuffer((text[s:s + l] * (n // l + 1))[:n], dtype=np.uint8).copy()
    else:                # NUL-heavy
        data = rng.integers(0, 4, size=n, dtype=np.uint8)
    data = np.ascontiguousarray(data)
    off = None
    if rng.integers(0, 3) == 0 and len(data) > 4:
        cuts = np.unique(rng.integers(1, len(data), size=max(len(data) // int(rng.integers(2, 200)), 1)))
        off = np.concatenate([[0], cuts, [len(data)]]).astype(np.uint64)
    vocab = 256 + int(rng.integers(0, 400))
    opts = {"max_batch": int(rng.choice([1, 2, 3, 7, 16, 64, 128, 256, 512, 1024, 4096])), "fused_min": int(rng.choice([2, 24, 1000])),
            "dense_table": int(rng.choice([0, 1])), "threshold_select": int(rng.choice([0, 1])),
            "sel_cap": int(rng.choice([64, 256, 4096, 8192])),
            "compact_den": int(rng.choice([0, 2, 8])), "batch": int(rng.choice([1, 3, 64])),
            "multi_merge": int(rng.choice([0, 1, 1, 1])),
            "chunk_barrier": int(rng.choice([-1, 1])),        # (chunk ends as barrier slots: tests/test_gpu_barrier.py)
            # the host waits for every selection and enqueues only that sequence's kernels (the default at these sizes), or
            # enqueues whole groups of sequences with every kernel variant (what large corpora get)
            "lockstep": int(rng.choice([0, 1])),
            # matches between raw bytes counted with one atomic on the pair's byte x byte cell block (what streams from
            # 64 Mi slots on get) or with two on its L and R rows
            "pair_cells": int(rng.choice([0, 1]))}
    return data, off, vocab, opts


import os


@pytest.mark.parametrize("seed", range(int(os.environ.get("MBPE_FUZZ_SEEDS", "4"))))
def test_fuzz_against_oracle(seed):
    text = read_data("shakespeare.txt")
    with mbpe.Trainer(0) as tr:
        for case in range(60):
            rng = np.random.default_rng(7000 + seed * 1000 + case)
            data, off, vocab, opts = _case(rng, text)
            for k, v in {**DEFAULTS, **opts}.items():
                tr.set_option(k, v)
            want_m, want_c = O.train(data, vocab, off)
            m, c, _ = tr.train_lexical(data, vocab, off)
            tag = (seed, case, len(data), vocab, off is not None, opts)
            assert m.tolist() == want_m.tolist() and c.tolist() == want_c.tolist(), tag
            if not len(want_m):
                continue
            st = O.State(data, off)
            for i, (a, b) in enumerate(want_m):
                st.merge(int(a), int(b), 256 + i)
            starts = np.array([0], dtype=np.int64) if off is None else off[:-1].astype(np.int64)
            if not np.any(data[starts] == 0):
                # (a NUL-led chunk that parses as a number is one inert token in the reference,
                #  Tokenizer.h:86-93, and inert bytes here: same merges, different stream listing)
                assert np.array_equal(tr.stream()[0], st.stream()[0]), tag
            assert {k: v for k, v in tr.pairs_dict().items() if v} == \
                   {k: v for k, v in st.table_dict().items() if v}, tag
            assert check.tiles_in_prefix_form(tr, torch, torch.device("cuda", 0)), tag
            st.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("MBPE_FUZZ_SEEDS", "4"))))
def test_fuzz_first_mode_and_wide_handover(seed):
    """The same random cases through the two other loops of the library: the `first` tie-break (one merge per pass, ties by
    stream position; the oracle rebuilds its insertion-ordered table before every merge as Tokenizer.h:581-585 does) and
    the 32-bit continuation, handed over to after a random number of merges ("wide_from")."""
    text = read_data("shakespeare.txt")
    with mbpe.Trainer(0) as tr:
        for case in range(30):
            rng = np.random.default_rng(9000 + seed * 1000 + case)
            data, off, vocab, opts = _case(rng, text)
            data = data[:20000]                               # (the first-mode oracle recounts per merge)
            if off is not None:
                off = np.concatenate([off[off < len(data)], [len(data)]]).astype(np.uint64)
                off = np.unique(off)
                if off[0] != 0:
                    off = np.concatenate([[0], off]).astype(np.uint64)
            vocab = min(vocab, 256 + 120)
            for k, v in {**DEFAULTS, **opts}.items():
                tr.set_option(k, v)
            tag = (seed, case, len(data), vocab, off is not None, opts)
            # first
            want_m, want_c = O.train(data, vocab, off, mode=O.FIRST)
            m, c, _ = tr.train(data, vocab, off, conflict_resolution=0)
            assert m.tolist() == want_m.tolist() and c.tolist() == want_c.tolist(), ("first",) + tag
            # lexical, 32-bit tokens from merge `wide_from` on
            wf = int(rng.integers(0, max(vocab - 256, 1) + 1))
            tr.set_option("wide_from", wf)
            try:
                want_m, want_c = O.train(data, vocab, off)
                m, c, _ = tr.train_lexical(data, vocab, off)
                assert m.tolist() == want_m.tolist() and c.tolist() == want_c.tolist(), ("wide", wf) + tag
                if len(want_m):
                    st = O.State(data, off)
                    for i, (a, b) in enumerate(want_m):
                        st.merge(int(a), int(b), 256 + i)
                    starts = np.array([0], dtype=np.int64) if off is None else off[:-1].astype(np.int64)
                    if len(data) and not np.any(data[starts[starts < len(data)]] == 0):
                        assert np.array_equal(tr.stream()[0], st.stream()[0]), ("wide", wf) + tag
                    assert {k: v for k, v in tr.pairs_dict().items() if v} == \
                           {k: v for k, v in st.table_dict().items() if v}, ("wide", wf) + tag
                    st.close()
            finally:
                tr.set_option("wide_from", -1)


def _medium_case(rng, text):
    """1-6 MiB: thousands of tiles, every workgroup of the pair-count scan busy, compactions, selection retries
    (the oracle needs 5-15 s per case: MBPE_FUZZ_MEDIUM sets the number of cases)."""
    kind = int(rng.integers(0, 4))
    n = int(rng.integers(1 << 20, 6 << 20))
    if kind == 0:        # small alphabets: thousands of tied counts
        data = rng.integers(0, int(rng.choice([3, 7, 20, 256])), size=n, dtype=np.uint8)
    elif kind == 1:      # a text slice repeated, with a random prefix so that copies do not line up with tiles
        s = int(rng.integers(0, len(text) - 300000))
        l = int(rng.integers(1000, 300000))
        data = np.frombuffer((text[s:s + l] * (n // l + 1))[:n], dtype=np.uint8).copy()
    elif kind == 2:      # runs
        vals = rng.integers(97, 103, size=max(n // 20, 1), dtype=np.uint8)
        data = np.repeat(vals, rng.integers(1, 40, size=len(vals)))[:n]
    else:                # the benchmark's generator
        data = O.splitmix64_bytes(int(rng.integers(1, 1 << 30)), n)
    data = np.ascontiguousarray(data)
    off = None
    if rng.integers(0, 3) == 0:
        cuts = np.unique(rng.integers(1, len(data), size=len(data) // int(rng.integers(4, 4000))))
        off = np.concatenate([[0], cuts, [len(data)]]).astype(np.uint64)
    vocab = 256 + int(rng.integers(50, 700))
    opts = {"fused_min": int(rng.choice([2, 24, 24, 1000])), "dense_table": int(rng.choice([0, 1])),
            "compact_den": int(rng.choice([2, 16, 16])), "batch": int(rng.choice([1, 16, 16])),
            "max_batch": int(rng.choice([64, 4096, 4096])), "chunk_barrier": int(rng.choice([-1, 1])),
            "lockstep": int(rng.choice([0, 1, -1])), "pair_cells": int(rng.choice([0, 1]))}
    return data, off, vocab, opts


@pytest.mark.parametrize("seed", range(int(os.environ.get("MBPE_FUZZ_MEDIUM", "4"))))
def test_fuzz_medium_sizes_against_oracle(seed):
    text = read_data("shakespeare.txt")
    rng = np.random.default_rng(40000 + seed)
    data, off, vocab, opts = _medium_case(rng, text)
    # (a fifth of the cases with the `first` tie-break, a fifth handed over to the 32-bit continuation on the way)
    variant = int(rng.integers(0, 5))
    first = variant == 0
    if first:
        vocab = min(vocab, 256 + 100)
    if variant == 1:
        opts["wide_from"] = int(rng.integers(0, vocab - 256))
    want_m, want_c = O.train(data, vocab, off, mode=O.FIRST if first else O.LEXICAL)
    with mbpe.Trainer(0) as tr:
        for k, v in {**DEFAULTS, **opts}.items():
            tr.set_option(k, v)
        m, c, _ = tr.train(data, vocab, off, conflict_resolution=0 if first else 1)
        tag = (seed, len(data), vocab, off is not None, "first" if first else "lexical", opts)
        assert m.tolist() == want_m.tolist() and c.tolist() == want_c.tolist(), tag
        if len(want_m):
            st = O.State(data, off, mode=O.FIRST if first else O.LEXICAL)
            for i, (a, b) in enumerate(want_m):
                st.merge(int(a), int(b), 256 + i)
            assert {k: v for k, v in tr.pairs_dict().items() if v} == {k: v for k, v in st.table_dict().items() if v}, tag
            starts = np.array([0], dtype=np.int64) if off is None else off[:-1].astype(np.int64)
            if not np.any(data[starts] == 0):
                assert np.array_equal(tr.stream()[0], st.stream()[0]), tag
            st.close()

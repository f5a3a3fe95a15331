"""Chunked corpora whose token ids need all 16 bits (vocab_size above 32,766 with a split pattern): the stream
keeps a barrier slot after every chunk instead of a flag bit inside the slot (mbpe_dev.h: kBarrier).  The same
parity bar as the flag layout -- chosen pairs, counts, live stream, chunk ends and the whole pair table against
the oracle -- with the layout forced on at small vocabularies ("chunk_barrier" 1), and a GPT-4-split training
beyond 32,766 ids where the library picks it by itself.  Reference: Token = uint32_t (Tokenizer.h:37), pairs
counted and merged only inside a chunk (Tokenizer.h:135-144, :311-319)."""
import json
import os

import numpy as np
import pytest

import mbpe
import oracle as O
from conftest import GOLDEN, read_data, read_golden
from test_gpu_parity import DEFAULTS, _defaults, _input, _random_chunks, _step_parity

pytestmark = pytest.mark.gpu

INDEX = json.load(open(os.path.join(GOLDEN, "index.json")))
CHUNKED_GOLDENS = sorted(k for k, v in INDEX.items() if v["encoder"] != "basic")


@pytest.fixture(scope="module")
def tr():
    t = mbpe.Trainer(0)
    yield t
    t.close()


@pytest.mark.parametrize("seed", range(6))
def test_step_parity_chunked_barrier(tr, seed):
    rng = np.random.default_rng(100 + seed)
    n = int(rng.integers(2, 9000))
    data = rng.integers(97, 97 + int(rng.integers(1, 6)), size=n, dtype=np.uint8)
    off = _random_chunks(rng, n, int(rng.integers(2, 12)))
    _step_parity(tr, data, off, 256 + 40, batch=1, compact_den=int(rng.choice([0, 2, 3, 8, 50, 100000])),
                 chunk_barrier=1)


def test_step_parity_single_byte_chunks_and_runs(tr):
    # chunks of one byte (a barrier after every token), long runs cut by chunk ends, chunks ending on tile edges
    data = np.frombuffer(b"a" * 3000 + b"ab" * 2000 + b"abc" * 1000, dtype=np.uint8)
    cuts = sorted(set(list(range(0, 600)) + [1023, 1024, 1025, 2047, 2048, 2999, 3000, 3001, 5000, 7000, 7001]))
    off = np.array(cuts + [len(data)], dtype=np.uint64)
    _step_parity(tr, data, off, 256 + 30, batch=1, compact_den=3, chunk_barrier=1)
    _step_parity(tr, data, off, 256 + 30, stride=8, fused_min=2, chunk_barrier=1)


@pytest.mark.parametrize("dense", [0, 1])
@pytest.mark.parametrize("fused_min", [2, 1000])
@pytest.mark.parametrize("seed", range(4))
def test_batched_merges_parity_chunked_text_barrier(tr, seed, fused_min, dense):
    data = read_data("taylorswift.txt")[seed * 20000:seed * 20000 + 40000]
    off = mbpe.presplit(O.GPT4_SPLIT_PATTERN if seed % 2 else O.GPT2_SPLIT_PATTERN, data)
    _step_parity(tr, data, off, 256 + 150, stride=25, fused_min=fused_min, dense_table=dense, chunk_barrier=1)


@pytest.mark.parametrize("name", CHUNKED_GOLDENS)
def test_golden_models_barrier(tr, name):
    meta = INDEX[name]
    data = _input(meta["input"])
    off = mbpe.presplit(O.PATTERNS[meta["encoder"]], data)
    tr.set_option("chunk_barrier", 1)
    try:
        merges, counts, stats = tr.train(data, meta["vocab"], off,
                                         conflict_resolution=1 if meta["mode"] == "lexical" else 0)
    finally:
        _defaults(tr)
    assert O.model_bytes(O.PATTERNS[meta["encoder"]], merges) == read_golden(name + ".model")
    if meta["mode"] == "lexical":
        assert stats["n_live"] == meta["final_len"]          # (barriers are not tokens)


@pytest.mark.parametrize("seed", range(4))
def test_first_mode_fuzz_barrier(tr, seed):
    rng = np.random.default_rng(2000 + seed)
    n = int(rng.integers(200, 30000))
    data = rng.integers(97, 97 + int(rng.integers(2, 12)), size=n, dtype=np.uint8)
    off = _random_chunks(rng, n, int(rng.integers(3, 40)))
    vocab = 256 + int(rng.integers(20, 400))
    want_m, want_c = O.train(data, vocab, off, mode=O.FIRST)
    tr.set_option("chunk_barrier", 1)
    tr.set_option("dense_table", seed % 2)
    try:
        m, c, _ = tr.train(data, vocab, off, conflict_resolution=0)
    finally:
        _defaults(tr)
    assert m.tolist() == want_m.tolist() and c.tolist() == want_c.tolist()


def _final_state_equal(tr, data, off, merges):
    ost = O.State(data, off)
    for i, (a, b) in enumerate(merges):
        ost.merge(int(a), int(b), 256 + i)
    want_toks, want_clen = ost.stream()
    toks, ends = tr.stream()
    assert np.array_equal(toks, want_toks)
    pos = np.cumsum(want_clen[want_clen > 0]).astype(np.int64) - 1
    want_ends = np.zeros(len(want_toks), dtype=np.uint8)
    want_ends[pos] = 1
    assert np.array_equal(ends, want_ends)
    assert {k: v for k, v in tr.pairs_dict().items() if v} == {k: v for k, v in ost.table_dict().items() if v}
    ost.close()


def test_gpt4_split_vocab_50000(tr):
    """GPT-4 split pattern, vocab 50,000 (the run the 15-bit id limit used to refuse): merges and counts against
    the oracle, then the final stream, chunk ends and pair table."""
    data = read_data("shakespeare.txt")
    off = mbpe.presplit(O.GPT4_SPLIT_PATTERN, data)
    vocab = 50000
    want_m, want_c = O.train(data, vocab, off)
    m, c, st = tr.train_lexical(data, vocab, off)
    assert m.tolist() == want_m.tolist() and c.tolist() == want_c.tolist()
    # (this text runs out of repeated pairs near merge 21,000: the late merges have count 0 and are chosen in
    #  lexical order among thousands of zero-count pairs, new ids 32,768.. included; the next test has real counts
    #  up there)
    assert len(m) == vocab - 256 and int(c[-1]) == 0
    _final_state_equal(tr, data, off, want_m)
    n_zero = int((c == 0).sum())          # (a zero-count pair is merged alone)
    assert st["n_batches"] < n_zero + (len(m) - n_zero) // 8


def test_random_bytes_random_chunks_ids_beyond_15_bits(tr):
    """1 MiB of random bytes in random chunks: every merge up to id 36,000 still has a count of several, so tokens
    above 32,767 appear in matches, as neighbours (delta rows) and next to barriers."""
    rng = np.random.default_rng(77)
    n = 1 << 20
    # (no NUL bytes: a NUL-led chunk that parses as a number collapses to one token in the reference,
    #  Tokenizer.h:86-93; the library keeps such a chunk's bytes as inert tokens, so the streams would differ there)
    data = rng.integers(1, 256, size=n, dtype=np.uint8)
    off = _random_chunks(rng, n, 48)
    vocab = 36000
    want_m, want_c = O.train(data, vocab, off)
    m, c, st = tr.train_lexical(data, vocab, off)
    assert m.tolist() == want_m.tolist() and c.tolist() == want_c.tolist()
    assert int(c[-1]) >= 2
    _final_state_equal(tr, data, off, want_m)


def test_vocab_limits_chunked(tr):
    data = b"hello world hello world"
    tr.load_corpus(data, np.array([0, 5, len(data)], dtype=np.uint64))
    # beyond the 16-bit formats the lexical training continues on 32-bit tokens (tests/test_gpu_wide.py); `first` cannot
    tr.set_option("conflict_resolution", 0)
    try:
        with pytest.raises(mbpe.MbpeError) as e:
            tr.train_begin(65519)
        assert e.value.code == mbpe.ERR_VOCAB
        tr.set_option("chunk_barrier", 0)
        with pytest.raises(mbpe.MbpeError) as e:
            tr.train_begin(32767)
        assert e.value.code == mbpe.ERR_VOCAB
    finally:
        tr.set_option("conflict_resolution", 1)
        _defaults(tr)
    tr.train_begin(65518)
    assert tr.train_steps(5) == 5
    tr.train_begin(65519)
    assert tr.train_steps(5) == 5


@pytest.mark.parametrize("data,cuts", [(b"", []), (b"a", []), (b"ab", [1]), (b"abab", [2]), (b"aaaa", [1, 2, 3]),
                                       (b"\x00123abcabc", [4]), (b"abcabcabc" * 100, [3, 6, 9, 450, 451])])
def test_tiny_and_degenerate_chunkings_barrier(tr, data, cuts):
    off = np.array([0] + cuts + [len(data)], dtype=np.uint64) if len(data) else np.array([0], dtype=np.uint64)
    vocab = 256 + 8
    want_m, want_c = O.train(data, vocab, off)
    tr.set_option("chunk_barrier", 1)
    try:
        m, c, st = tr.train_lexical(np.frombuffer(data, dtype=np.uint8), vocab, off)
    finally:
        _defaults(tr)
    assert m.tolist() == want_m.tolist() and c.tolist() == want_c.tolist()

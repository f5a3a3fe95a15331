"""Vocabularies beyond the 16-bit slot format: the reference's Token is a uint32_t (Tokenizer.h:37-38), so a request such
as `--vocab-size 100000 --encoder gpt4` trains there.  The library runs the merges the slot stream can hold on it and
continues on 32-bit tokens with 64-bit pair keys (csrc/wide.h, one merge per pass).  Same parity bar as everywhere
else -- chosen pairs, counts, live stream, chunk ends, whole pair table against the oracle after every step -- with the
hand-over forced early ("wide_from": after that many merges) so that small corpora exercise the conversion and the
32-bit loop, plus trainings whose ids really pass 65,535."""
import numpy as np
import pytest

import mbpe
import oracle as O
from conftest import read_data
from test_gpu_parity import _defaults, _random_chunks, _step_parity

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tr():
    t = mbpe.Trainer(0)
    yield t
    t.close()


@pytest.mark.parametrize("seed", range(8))
def test_step_parity_small_alphabet_wide(tr, seed):
    # 1-4 symbols: runs (a == b merges: `a a a` -> `X a`), touching matches (`abab`), spans of candidates across the
    # 1,024-token spans of the 32-bit loop; hand-over after 0..12 merges
    rng = np.random.default_rng(500 + seed)
    n = int(rng.integers(1, 9000))
    data = rng.integers(97, 97 + int(rng.integers(1, 5)), size=n, dtype=np.uint8)
    _step_parity(tr, data, None, 256 + 40, batch=1, wide_from=int(rng.integers(0, 13)))


@pytest.mark.parametrize("seed", range(8))
def test_step_parity_chunked_wide(tr, seed):
    # chunk ends as flag bits (default) and as barrier slots before the conversion
    rng = np.random.default_rng(600 + seed)
    n = int(rng.integers(2, 9000))
    data = rng.integers(97, 97 + int(rng.integers(1, 6)), size=n, dtype=np.uint8)
    off = _random_chunks(rng, n, int(rng.integers(2, 12)))
    _step_parity(tr, data, off, 256 + 40, batch=1, wide_from=int(rng.integers(0, 13)), chunk_barrier=seed % 2)


def test_step_parity_runs_across_spans_wide(tr):
    # one byte repeated over several spans, cut by chunk ends at span edges: run parity travels through the scan
    data = np.frombuffer(b"a" * 5000 + b"ab" * 3000 + b"aab" * 1000, dtype=np.uint8)
    cuts = sorted(set([1023, 1024, 1025, 2047, 2048, 2049, 4096, 5000, 5001, 7000, 9000, 11001]))
    off = np.array([0] + cuts + [len(data)], dtype=np.uint64)
    _step_parity(tr, data, None, 256 + 30, batch=1, wide_from=0)
    _step_parity(tr, data, off, 256 + 30, batch=1, wide_from=0)
    _step_parity(tr, data, off, 256 + 30, stride=7, wide_from=3)


@pytest.mark.parametrize("seed", range(4))
def test_step_parity_text_wide(tr, seed):
    data = read_data("taylorswift.txt")[seed * 20000:seed * 20000 + 30000]
    off = mbpe.presplit(O.GPT4_SPLIT_PATTERN if seed % 2 else O.GPT2_SPLIT_PATTERN, data)
    _step_parity(tr, data, off, 256 + 120, stride=20, wide_from=40 + 10 * seed)


@pytest.mark.parametrize("data,cuts", [(b"", []), (b"a", []), (b"ab", [1]), (b"abab", []), (b"aaaa", []), (b"aaaa", [1, 2, 3]),
                                       (b"abcabcabc" * 100, [3, 6, 9, 450, 451]), (b"abcbcde", [])])
def test_tiny_inputs_wide(tr, data, cuts):
    off = np.array([0] + cuts + [len(data)], dtype=np.uint64) if cuts else None
    vocab = 256 + 8
    want_m, want_c = O.train(data, vocab, off)
    tr.set_option("wide_from", 0)
    try:
        m, c, st = tr.train_lexical(np.frombuffer(data, dtype=np.uint8), vocab, off)
    finally:
        _defaults(tr)
    assert m.tolist() == want_m.tolist() and c.tolist() == want_c.tolist()


def test_zero_count_tail_wide(tr):
    # the vocabulary outlasts the corpus: the reference keeps choosing the smallest zero-count pair (never-erased table)
    data = read_data("small.txt")
    want_m, want_c = O.train(data, 275)
    for wf in (0, 3, 7, 9):
        tr.set_option("wide_from", wf)
        try:
            m, c, _ = tr.train_lexical(data, 275)
        finally:
            _defaults(tr)
        assert m.tolist() == want_m.tolist() and c.tolist() == want_c.tolist(), wf


def _final_state_equal(tr, data, off, merges):
    ost = O.State(data, off)
    for i, (a, b) in enumerate(merges):
        ost.merge(int(a), int(b), 256 + i)
    want_toks, want_clen = ost.stream()
    toks, ends = tr.stream()
    assert np.array_equal(toks, want_toks)
    if off is not None:
        pos = np.cumsum(want_clen[want_clen > 0]).astype(np.int64) - 1
        want_ends = np.zeros(len(want_toks), dtype=np.uint8)
        want_ends[pos] = 1
        assert np.array_equal(ends, want_ends)
    assert {k: v for k, v in tr.pairs_dict().items() if v} == {k: v for k, v in ost.table_dict().items() if v}
    ost.close()


def _word_corpus(seed, n_words, word_len, reps):
    """`n_words` distinct random words of `word_len` bytes, each occurring `reps[0]..reps[1]` times, in random order:
    every pair inside a word keeps the word's count until the word is one token, so n_words x (word_len - 1) merges
    have real counts -- tens of thousands of them on less than a megabyte (the oracle walks the stream per merge)."""
    rng = np.random.default_rng(seed)
    words = rng.integers(1, 256, size=(n_words, word_len), dtype=np.uint8)      # (no NUL: see tests/test_gpu_barrier.py)
    occ = np.repeat(np.arange(n_words), rng.integers(reps[0], reps[1] + 1, size=n_words))
    rng.shuffle(occ)
    data = words[occ].reshape(-1)
    off = (np.arange(len(occ) + 1, dtype=np.uint64) * np.uint64(word_len))
    return data, off


def test_ids_beyond_16_bits_real_counts_chunked(tr):
    """4,200 random words of 24 bytes, 6-14 occurrences each, one chunk per occurrence, vocab 68,000: the merges past id
    65,518 (where the barrier layout of the slot stream ends) still have counts of 6 and more, so 17-bit ids appear in
    matches, as neighbours, in new pairs and next to chunk ends."""
    data, off = _word_corpus(78, 4200, 24, (6, 14))
    vocab = 68000
    want_m, want_c = O.train(data, vocab, off)
    m, c, st = tr.train_lexical(data, vocab, off)
    assert len(m) == vocab - 256
    assert m.tolist() == want_m.tolist() and c.tolist() == want_c.tolist()
    assert int(c[65518 - 256 + 500]) >= 2 and int(m.max()) > 65535
    _final_state_equal(tr, data, off, want_m)


def test_ids_beyond_16_bits_real_counts_one_chunk(tr):
    """The same kind of corpus as ONE chunk (the `basic` encoder): the slot stream carries ids up to 65,533, the 32-bit
    loop the rest; the decode of the final stream is the corpus."""
    import torch
    from mbpe import check
    data, _ = _word_corpus(79, 4200, 24, (6, 14))
    vocab = 67000
    want_m, want_c = O.train(data, vocab)
    m, c, st = tr.train_lexical(data, vocab)
    assert m.tolist() == want_m.tolist() and c.tolist() == want_c.tolist()
    assert int(c[65534 - 256 + 500]) >= 2 and int(m.max()) > 65535
    _final_state_equal(tr, data, None, want_m)
    dev = torch.device("cuda", 0)
    rt = check.decode_roundtrip(tr, m, torch.from_numpy(np.ascontiguousarray(data)).to(dev), torch, dev)
    assert rt["ok"], rt


def test_gpt4_split_vocab_100000_cli_shape(tr):
    """`--vocab-size 100000 --encoder gpt4` on shakespeare.txt: what the 16-bit format refused.  The text runs out of
    repeated pairs long before (zero-count merges in lexical order, new ids included): merges and counts against the
    oracle for the whole vocabulary."""
    data = read_data("shakespeare.txt")
    off = mbpe.presplit(O.GPT4_SPLIT_PATTERN, data)
    vocab = 100000
    want_m, want_c = O.train(data, vocab, off)
    m, c, st = tr.train_lexical(data, vocab, off)
    assert len(m) == vocab - 256
    assert m.tolist() == want_m.tolist() and c.tolist() == want_c.tolist()

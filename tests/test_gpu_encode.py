"""GPU tests of encode on the device (mbpe_encode_chunks / mbpe_tok_encode_device): the reference's
multi-pass greedy replacement (Tokenizer.h:325-377) as lookup + run-parity scan + compaction,
against the encode digests of SURVEY.md 8c, the oracle's sequential restatement and the host encode."""
import hashlib
import os
import subprocess

import numpy as np
import pytest

import mbpe
import oracle as O
from conftest import GOLDEN, ROOT, read_data, read_golden
from test_tokenizer_cpu import SPECIAL_SAMPLE_TOKENS, _golden_merges

pytestmark = pytest.mark.gpu
CLI = os.path.join(ROOT, "minbpe-cc_amd", "minbpe-cc")
DATA = os.path.join(GOLDEN, "data")


def test_encode_sample_with_shakespeare_basic_model_on_device():
    # endtoend-test.sh:7-10; SURVEY 8c: 15,677 tokens, sha256 624874b4...
    merges = _golden_merges("shakespeare_basic_lexical_512")
    data = read_data("sample.txt")
    enc, passes = mbpe.encode_chunks(data, None, merges)
    assert len(enc) == 15677 and passes >= 2
    assert hashlib.sha256(enc.astype("<u4").tobytes()).hexdigest() == \
        "624874b4a8bce9405f0a89ecb7b3e7eeaa94b2a3235e88c05acd6426c05cb409"
    tok = mbpe.Tokenizer("")
    tok.set_merges(merges)
    assert np.array_equal(tok.encode(data, device=0), enc)
    assert tok.decode(enc) == data


def test_encode_taylorswift_with_gpt4_model_on_device():
    # config 5; SURVEY 8c: 94,201 tokens, sha256 1b82232e...
    tok = mbpe.Tokenizer(O.GPT4_SPLIT_PATTERN)
    merges = _golden_merges("taylorswift_gpt4_lexical_512")
    tok.set_merges(merges)
    data = read_data("taylorswift.txt")
    enc = tok.encode(data, device=0)
    assert len(enc) == 94201
    assert hashlib.sha256(enc.astype("<u4").tobytes()).hexdigest() == \
        "1b82232e30d1972b1b9f8b54080fc8757bcce310b6b8f9de4d63fdd18f034d0d"
    assert np.array_equal(enc, tok.encode(data))           # the host encode
    assert tok.decode(enc) == data


def test_special_tokens_on_device():
    # endtoend-test.sh:13-16: special tokens travel as NUL-led chunks that become one token (Tokenizer.h:86-93)
    tok = mbpe.Tokenizer(O.GPT4_SPLIT_PATTERN)
    tok.set_special_tokens_from_file(read_data("special1.txt"))
    tok.set_merges(_golden_merges("taylorswift_gpt4_first_512"))
    data = read_data("specialtokensample.txt")
    assert tok.encode(data, device=0).tolist() == SPECIAL_SAMPLE_TOKENS
    # basic encoder: the parts between special tokens are the chunks
    tok = mbpe.Tokenizer("")
    tok.set_special_tokens_from_file(b"<|x|> 70000\n")
    tok.set_merges(np.array([[97, 98], [256, 99]], dtype=np.uint32))
    text = b"abc<|x|>abcab<|x|><|x|>c"
    assert tok.encode(text, device=0).tolist() == tok.encode(text).tolist() == [257, 70000, 257, 256, 70000, 70000, 99]


def test_encode_greedy_not_rank_ordered_and_tiny_inputs():
    m = np.array([[98, 99], [97, 98]], dtype=np.uint32)          # (b,c)->256, (a,b)->257
    assert mbpe.encode_chunks(b"abc", None, m)[0].tolist() == [257, 99]    # minbpe would give [97, 256]
    assert mbpe.encode_chunks(b"", None, m)[0].tolist() == []
    assert mbpe.encode_chunks(b"a", None, m)[0].tolist() == [97]
    assert mbpe.encode_chunks(b"abc", None, np.zeros((0, 2), dtype=np.uint32))[0].tolist() == [97, 98, 99]
    # a repeated pair keeps the last id (merges_lookup[pair] = idx)
    m = np.array([[97, 97], [97, 97]], dtype=np.uint32)
    assert mbpe.encode_chunks(b"aaaaa", None, m)[0].tolist() == [257, 257, 97]
    # candidate runs across the 64-lane groups and 1,024-token spans: one long run of a single byte
    for n in (63, 64, 65, 1023, 1024, 1025, 2049, 70001):
        m = np.array([[97, 97], [256, 256], [257, 257]], dtype=np.uint32)
        data = b"a" * n
        assert np.array_equal(mbpe.encode_chunks(data, None, m)[0], O.encode_chunks(data, None, m)), n


@pytest.mark.parametrize("seed", range(8))
def test_encode_fuzz_against_oracle(seed):
    """Random corpora over small alphabets (long candidate runs, deep merge trees), random chunking, merges
    trained by the oracle on other data of the same kind (so every depth of merge occurs)."""
    rng = np.random.default_rng(500 + seed)
    alpha = int(rng.integers(2, 20))
    train = rng.integers(97, 97 + alpha, size=int(rng.integers(500, 20000)), dtype=np.uint8)
    merges, _ = O.train(train, 256 + int(rng.integers(5, 300)))
    n = int(rng.integers(1, 200000))
    data = rng.integers(97, 97 + alpha, size=n, dtype=np.uint8)
    if seed % 3 == 0:                                   # runs
        data = np.repeat(data[:n // 8 + 1], rng.integers(1, 16, size=n // 8 + 1))[:n]
    off = None
    if seed % 2:
        cuts = np.unique(rng.integers(1, max(len(data), 2), size=max(len(data) // int(rng.integers(2, 50)), 1)))
        off = np.concatenate([[0], cuts[cuts < len(data)], [len(data)]]).astype(np.uint64)
    got, _ = mbpe.encode_chunks(data, off, merges)
    assert np.array_equal(got, O.encode_chunks(data, off, merges))


def test_encode_large_text_matches_host_encode():
    data = read_data("shakespeare.txt") * 8                       # 8.9 MB
    merges, _ = O.train(read_data("shakespeare.txt"), 2000)
    tok = mbpe.Tokenizer(O.GPT4_SPLIT_PATTERN)
    tok.set_merges(merges)
    dev = tok.encode(data, device=0)
    assert np.array_equal(dev, tok.encode(data))
    assert tok.decode(dev) == data


def test_cli_device_encode(tmp_path):
    model, enc, enc2, dec = tmp_path / "m", tmp_path / "enc", tmp_path / "enc2", tmp_path / "dec"
    model.write_bytes(read_golden("shakespeare_basic_lexical_512.model"))
    src = os.path.join(DATA, "sample.txt")
    for out, extra in ((enc, []), (enc2, ["--device-encode"])):
        r = subprocess.run([CLI, "--encode", "--input", src, "--model-path", str(model), "--output", str(out)] + extra,
                           capture_output=True, text=True)
        assert r.returncode == 0 and "Writing 15677 encoded tokens" in r.stdout, r.stdout + r.stderr
    assert enc.read_bytes() == enc2.read_bytes()
    r = subprocess.run([CLI, "--decode", "--input", str(enc2), "--model-path", str(model), "--output", str(dec)],
                       capture_output=True, text=True)
    assert r.returncode == 0 and dec.read_bytes() == read_data("sample.txt")


def test_training_with_a_pattern_that_leaves_gaps():
    """A custom split pattern that does not match every byte: the reference skips the bytes between matches
    (Tokenizer.h:506-540); here the chunks are packed and trained on (mbpe_load_corpus_ranges)."""
    data = read_data("taylorswift.txt")
    starts, ends = mbpe.presplit_ranges(r"\p{L}+", data)
    assert len(starts) > 1000 and int((ends - starts).sum()) < len(data)
    with pytest.raises(mbpe.MbpeError):
        mbpe.presplit(r"\p{L}+", data)                  # offsets exist only when the chunks tile the text
    packed = b"".join(data[int(s):int(e)] for s, e in zip(starts, ends))
    off = np.concatenate([[0], np.cumsum(ends - starts)]).astype(np.uint64)
    want_m, want_c = O.train(packed, 400, off)
    with mbpe.Trainer(0) as tr:
        tr.load_corpus_ranges(data, starts, ends)
        tr.train_begin(400)
        tr.train_steps(400 - 256)
        m, c = tr.train_result()
        assert tr.stats()["n_bytes"] == len(packed)
    assert m.tolist() == want_m.tolist() and c.tolist() == want_c.tolist()
    tok = mbpe.Tokenizer(r"\p{L}+")
    tok.train(data, 400)
    assert tok.merges().tolist() == want_m.tolist()


def test_encode_on_a_device_that_does_not_exist_reports_its_own_code():
    """mbpe_tok_encode_device hands on the code of the failing call (mbpe_encode_chunks), not a guess from the text."""
    tok = mbpe.Tokenizer("")
    tok.set_merges(np.array([[97, 98]], dtype=np.uint32))
    with pytest.raises(mbpe.MbpeError) as e:
        tok.encode(b"abab", device=99)
    assert e.value.code == mbpe.ERR_NO_DEVICE

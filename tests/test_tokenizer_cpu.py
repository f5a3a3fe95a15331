"""CPU tests of the host-side Tokenizer mirror (include/mbpe_tokenizer.h):
model files, special tokens, encode, decode -- against the golden fixtures,
the oracle's restatement of the reference encode, and the encode digests of
SURVEY.md 8c."""
import hashlib
import json
import os
import re
import subprocess

import numpy as np
import pytest

import mbpe
import oracle as O
from conftest import GOLDEN, ROOT, read_data, read_golden

INDEX = json.load(open(os.path.join(GOLDEN, "index.json")))
CLI = os.path.join(ROOT, "minbpe-cc_amd", "minbpe-cc")

# specialtokensample.txt with the taylorswift/gpt4/first + special1 model
SPECIAL_SAMPLE_TOKENS = [84, 104, 355, 32, 355, 306, 288, 101, 261, 101, 120, 116, 434, 287, 349, 262, 116, 97, 259,
                         115, 32, 100258, 261, 119, 111, 306, 112, 310, 478, 108, 32, 100257, 348, 107, 290, 115, 46]


def _golden_merges(name):
    return O.parse_model(read_golden(name + ".model"))[2]


def test_tokenizer_header_symbols_exported():
    header = open(os.path.join(ROOT, "include", "mbpe_tokenizer.h")).read()
    declared = set(re.findall(r"MBPE_API[^;]*?\b(mbpe_tok_\w+)\s*\(", header))
    assert declared == set(mbpe.TOK_EXPORTS)
    for s in declared:
        assert hasattr(mbpe.lib(), s), s


@pytest.mark.parametrize("name", sorted(INDEX))
def test_save_writes_the_golden_bytes_and_load_reads_them_back(name, tmp_path):
    meta = INDEX[name]
    pat = O.PATTERNS[meta["encoder"]]
    tok = mbpe.Tokenizer(pat)
    tok.set_merges(_golden_merges(name))
    path = tmp_path / "m.model"
    tok.save(path)
    assert path.read_bytes() == read_golden(name + ".model")
    tok2 = mbpe.Tokenizer("")            # load replaces the pattern with the file's
    tok2.load(path)
    assert tok2.merges().tolist() == _golden_merges(name).tolist()
    path2 = tmp_path / "m2.model"
    tok2.save(path2)
    assert path2.read_bytes() == read_golden(name + ".model")


def test_encode_sample_with_shakespeare_basic_model():
    # endtoend-test.sh:7-10; SURVEY 8c: 15,677 tokens, sha256 624874b4...
    tok = mbpe.Tokenizer("")
    tok.set_merges(_golden_merges("shakespeare_basic_lexical_512"))
    data = read_data("sample.txt")
    enc = tok.encode(data)
    assert len(enc) == 15677
    assert hashlib.sha256(enc.astype("<u4").tobytes()).hexdigest() == \
        "624874b4a8bce9405f0a89ecb7b3e7eeaa94b2a3235e88c05acd6426c05cb409"
    assert np.array_equal(enc, O.encode_chunks(data, None, _golden_merges("shakespeare_basic_lexical_512")))
    assert tok.decode(enc) == data


def test_encode_taylorswift_with_gpt4_model():
    # config 5; SURVEY 8c: 94,201 tokens, sha256 1b82232e...
    tok = mbpe.Tokenizer(O.GPT4_SPLIT_PATTERN)
    merges = _golden_merges("taylorswift_gpt4_lexical_512")
    tok.set_merges(merges)
    data = read_data("taylorswift.txt")
    enc = tok.encode(data)
    assert len(enc) == 94201
    assert hashlib.sha256(enc.astype("<u4").tobytes()).hexdigest() == \
        "1b82232e30d1972b1b9f8b54080fc8757bcce310b6b8f9de4d63fdd18f034d0d"
    off = mbpe.presplit(O.GPT4_SPLIT_PATTERN, data)
    assert np.array_equal(enc, O.encode_chunks(data, off, merges))
    assert tok.decode(enc) == data


def test_special_tokens_encode_decode(tmp_path):
    # endtoend-test.sh:13-16 with the golden first-mode merges
    tok = mbpe.Tokenizer(O.GPT4_SPLIT_PATTERN)
    tok.set_special_tokens_from_file(read_data("special1.txt"))
    tok.set_merges(_golden_merges("taylorswift_gpt4_first_512"))
    data = read_data("specialtokensample.txt")
    enc = tok.encode(data)
    assert enc.tolist() == SPECIAL_SAMPLE_TOKENS
    assert tok.decode(enc) == data
    # the model file carries the special tokens (Tokenizer.h:883-887) and loads back
    path = tmp_path / "sp.model"
    tok.save(path)
    lines = path.read_bytes().decode().split("\n")
    assert lines[0] == "minbpe v1" and lines[1] == O.GPT4_SPLIT_PATTERN and lines[2] == "5"
    assert lines[3] == "<|endoftext|> 100257" and lines[7] == "<|endofprompt|> 100276"
    tok2 = mbpe.Tokenizer(O.GPT4_SPLIT_PATTERN)
    tok2.load(path)
    assert tok2.encode(data).tolist() == SPECIAL_SAMPLE_TOKENS
    assert tok2.decode(enc) == data


def test_decode_skips_invalid_ids_and_vocab_file(tmp_path):
    tok = mbpe.Tokenizer("")
    tok.set_merges(np.array([[104, 105], [256, 33]], dtype=np.uint32))      # "hi", "hi!"
    assert tok.decode([257, 999999, 104]) == b"hi!h"                         # invalid id skipped, Tokenizer.h:739-742
    path = tmp_path / "v.model"
    tok.save(path, write_vocab=True)
    vocab = (tmp_path / "v.model.vocab").read_bytes().decode("utf-8").split("\n")
    assert vocab[65] == '65    : "A"'
    assert vocab[10] == '10    : "�"'
    assert vocab[256] == '256   : "hi"' and vocab[257] == '257   : "hi!"'


def test_encode_is_greedy_left_to_right_not_rank_ordered():
    # Tokenizer.h:325-367: any pair in merges_lookup is replaced in one left-to-right pass
    tok = mbpe.Tokenizer("")
    tok.set_merges(np.array([[98, 99], [97, 98]], dtype=np.uint32))          # (b,c)->256, (a,b)->257
    assert tok.encode(b"abc").tolist() == [257, 99]                          # minbpe would give [97, 256]
    assert np.array_equal(tok.encode(b"abc"), O.encode_chunks(b"abc", None, [[98, 99], [97, 98]]))
    assert tok.encode(b"").tolist() == [] and tok.encode(b"a").tolist() == [97]


def _reference_split_on_special(text, specials):
    """Tokenizer.h:605-650 as written: rescan from the cursor for the earliest occurrence of any special token."""
    if not specials:
        return [text]
    result, pos, last = [], 0, 0
    while pos < len(text):
        found_pos, found = -1, None
        for name, tid in specials:
            p = text.find(name, pos)
            if p != -1 and (found_pos == -1 or p < found_pos):
                found_pos, found = p, (name, tid)
        if found_pos == -1:
            break
        if found_pos > last:
            result.append(text[last:found_pos])
        result.append(b"\x00" + str(found[1]).encode())
        pos = last = found_pos + len(found[0])
    if last < len(text):
        result.append(text[last:])
    return result or [text]


def test_split_on_special_equals_the_reference_loop():
    specials = [(b"<|a|>", 1000), (b"<|ab|>", 1001), (b"|><|", 1002), (b"<<", 1003)]
    tok = mbpe.Tokenizer("")
    tok.set_special_tokens_from_file(b"".join(n + b" " + str(i).encode() + b"\n" for n, i in specials))
    tok.set_merges(np.zeros((0, 2), dtype=np.uint32))
    rng = np.random.default_rng(3)
    pieces = [b"<|a|>", b"<|ab|>", b"|><|", b"<<", b"<", b"|", b">", b"a", b"b", b"xyz", b" "]
    for _ in range(300):
        text = b"".join(pieces[int(i)] for i in rng.integers(0, len(pieces), size=int(rng.integers(0, 30))))
        want = []
        for part in _reference_split_on_special(text, specials):
            if part[:1] == b"\x00":
                want.append(int(part[1:]))
            else:
                want.extend(part)
        assert tok.encode(text).tolist() == want, text


def test_presplit_ranges_with_gaps():
    data = b"ab, cd! e"
    starts, ends = mbpe.presplit_ranges(r"\p{L}+", data)
    assert starts.tolist() == [0, 4, 8] and ends.tolist() == [2, 6, 9]
    with pytest.raises(mbpe.MbpeError):
        mbpe.presplit(r"\p{L}+", data)
    off = mbpe.presplit(O.GPT4_SPLIT_PATTERN, data)
    starts, ends = mbpe.presplit_ranges(O.GPT4_SPLIT_PATTERN, data)
    assert off[:-1].tolist() == starts.tolist() and off[1:].tolist() == ends.tolist()


def test_training_without_a_device_fails_loudly():
    # no CPU fallback for either conflict resolution: here (no GPU) train must fail with NO_DEVICE
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    for cr in (mbpe.Tokenizer.FIRST, mbpe.Tokenizer.LEXICAL):
        tok = mbpe.Tokenizer("")
        with pytest.raises(mbpe.MbpeError) as e:
            tok.train(b"abcabc", 300, conflict_resolution=cr)
        assert "no usable HIP device" in str(e.value)
    tok = mbpe.Tokenizer("")
    tok.set_merges(np.array([[97, 98]], dtype=np.uint32))
    with pytest.raises(mbpe.MbpeError) as e:
        tok.encode(b"abab", device=0)
    assert e.value.code == mbpe.ERR_NO_DEVICE


def test_cli_argument_errors():
    # minbpe-cc.cpp:135-144, :129-131, :174-176
    r = subprocess.run([CLI], capture_output=True, text=True)
    assert r.returncode == 255 and "Input file not specified" in r.stderr
    r = subprocess.run([CLI, "-i", "/nonexistent/file"], capture_output=True, text=True)
    assert r.returncode == 255 and "does not exist" in r.stderr
    r = subprocess.run([CLI, "-i", CLI, "-c", "middle"], capture_output=True, text=True)
    assert r.returncode == 105
    r = subprocess.run([CLI, "-i", CLI, "--encoder", "gpt5", "-t"], capture_output=True, text=True)
    assert r.returncode == 255 and "Encoder should be one of" in r.stdout


def test_cli_encode_decode_roundtrip(tmp_path):
    # encode + decode need no GPU: endtoend-test.sh:8-10 with the golden model
    model = tmp_path / "basic.model"
    model.write_bytes(read_golden("shakespeare_basic_lexical_512.model"))
    sample = os.path.join(GOLDEN, "data", "sample.txt")
    enc, dec = tmp_path / "s.enc", tmp_path / "s.dec"
    r = subprocess.run([CLI, "--encode", "--input", sample, "--model-path", str(model), "--output", str(enc)],
                       capture_output=True, text=True)
    assert r.returncode == 0 and "Writing 15677 encoded tokens" in r.stdout and "Execution time:" in r.stdout
    assert hashlib.sha256(enc.read_bytes()).hexdigest() == \
        "624874b4a8bce9405f0a89ecb7b3e7eeaa94b2a3235e88c05acd6426c05cb409"
    r = subprocess.run([CLI, "--decode", "--input", str(enc), "--model-path", str(model), "--output", str(dec)],
                       capture_output=True, text=True)
    assert r.returncode == 0 and "Loaded encoding with 15677 tokens" in r.stdout
    assert dec.read_bytes() == read_data("sample.txt")

"""GPU end-to-end tests through the drop-in command line (minbpe-cc_amd/minbpe-cc),
mirroring the reference's endtoend-test.sh: train -> encode -> decode -> diff."""
import os
import subprocess

import pytest

import mbpe
import oracle as O
from conftest import GOLDEN, ROOT, read_data, read_golden

pytestmark = pytest.mark.gpu

CLI = os.path.join(ROOT, "minbpe-cc_amd", "minbpe-cc")
DATA = os.path.join(GOLDEN, "data")


def _run(*args):
    r = subprocess.run([CLI] + [str(a) for a in args], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout


def test_e2e_basic_lexical(tmp_path):
    # endtoend-test.sh:7-10
    model, enc, dec = tmp_path / "basic-model", tmp_path / "enc", tmp_path / "dec"
    out = _run("--train", "--input", os.path.join(DATA, "shakespeare.txt"), "--model-path", model,
               "--vocab-size", 512, "--encoder", "basic", "--conflict-resolution", "lexical")
    assert "Training using file" in out and "Writing model..." in out and "Complete." in out and "Execution time:" in out
    assert model.read_bytes() == read_golden("shakespeare_basic_lexical_512.model")
    _run("--encode", "--input", os.path.join(DATA, "sample.txt"), "--model-path", model, "--output", enc)
    _run("--decode", "--input", enc, "--model-path", model, "--output", dec)
    assert dec.read_bytes() == read_data("sample.txt")


def test_e2e_gpt4_config5(tmp_path):
    # BASELINE config 5: taylorswift, --encoder gpt4 (CLI default), vocab 512, round trip exact
    model, enc, dec = tmp_path / "gpt4-model", tmp_path / "enc", tmp_path / "dec"
    src = os.path.join(DATA, "taylorswift.txt")
    out = _run("-t", "-i", src, "-m", model, "-c", "lexical", "-v", "-w")
    assert "Split input text into 46196 chunks" in out
    assert "merge 1/256: (101, 114) -> 256 (b'er') had 2359 occurrences" in out
    assert "merge 256/256: (306, 388) -> 511" in out and "had 82 occurrences" in out
    assert model.read_bytes() == read_golden("taylorswift_gpt4_lexical_512.model")
    assert os.path.exists(str(model) + ".vocab")
    out = _run("-e", "-i", src, "-m", model, "-o", enc)
    assert "Writing 94201 encoded tokens" in out
    _run("-d", "-i", enc, "-m", model, "-o", dec)
    assert dec.read_bytes() == read_data("taylorswift.txt")


def test_e2e_gpt4_first_special_tokens(tmp_path):
    # endtoend-test.sh:13-16: the reference's second scenario, with the CLI's own defaults
    # (--encoder gpt4, --conflict-resolution first, --vocab-size 512)
    model, enc, dec = tmp_path / "gpt4-special-model", tmp_path / "enc", tmp_path / "dec"
    special = os.path.join(DATA, "special1.txt")
    out = _run("--train", "--input", os.path.join(DATA, "taylorswift.txt"), "--model-path", model,
               "--special-tokens-path", special, "--encoder", "gpt4", "--conflict-resolution", "first")
    assert "Writing model..." in out
    got = mbpe_model_merges(model.read_bytes())
    assert got == mbpe_model_merges(read_golden("taylorswift_gpt4_first_512.model"))
    src = os.path.join(DATA, "specialtokensample.txt")
    _run("--encode", "--input", src, "--model-path", model, "--output", enc)
    import numpy as np
    toks = np.fromfile(str(enc), dtype=np.uint32).tolist()
    assert toks == [84, 104, 355, 32, 355, 306, 288, 101, 261, 101, 120, 116, 434, 287, 349, 262, 116, 97, 259, 115,
                    32, 100258, 261, 119, 111, 306, 112, 310, 478, 108, 32, 100257, 348, 107, 290, 115, 46]
    _run("--decode", "--input", enc, "--model-path", model, "--output", dec)
    assert dec.read_bytes() == read_data("specialtokensample.txt")


def mbpe_model_merges(blob):
    """The merge lines of a .model file (the special-token lines before them come in hash-map order)."""
    lines = blob.decode("utf-8").split("\n")
    n_special = int(lines[2])
    return lines[3 + n_special:]


def test_cli_defaults_train_first_mode(tmp_path):
    # `minbpe-cc --train -i x` with nothing else: gpt4 + first + vocab 512 (minbpe-cc.cpp:96-131)
    model = tmp_path / "m"
    _run("-t", "-i", os.path.join(DATA, "taylorswift.txt"), "-m", model)
    assert model.read_bytes() == read_golden("taylorswift_gpt4_first_512.model")
    # small.txt, basic, first: the loop ends after 7 merges (no pair left), SURVEY 8c
    _run("-t", "-i", os.path.join(DATA, "small.txt"), "-m", model, "--encoder", "basic", "--vocab-size", 275)
    assert mbpe_model_merges(model.read_bytes())[:-1] == ["98 99", "256 100", "257 101", "258 258", "97 259", "260 258", "261 10"]


def test_tokenizer_train_binding():
    tok = mbpe.Tokenizer(O.GPT2_SPLIT_PATTERN)
    tok.train(read_data("taylorswift.txt"), 512)
    assert O.model_bytes(O.GPT2_SPLIT_PATTERN, tok.merges()) == read_golden("taylorswift_gpt2_lexical_512.model")
    data = read_data("sample.txt")
    assert tok.decode(tok.encode(data)) == data


def test_e2e_gpt4_vocab_40000(tmp_path):
    # a GPT-style vocabulary with the split pattern: ids beyond 15 bits (the stream keeps chunk ends as barrier
    # slots, see tests/test_gpu_barrier.py); model bytes against the oracle, then the reference's round trip
    model, enc, dec = tmp_path / "big-model", tmp_path / "enc", tmp_path / "dec"
    src = os.path.join(DATA, "taylorswift.txt")
    out = _run("-t", "-i", src, "-m", model, "-c", "lexical", "--vocab-size", 40000)
    assert "Writing model..." in out
    data = read_data("taylorswift.txt")
    want_m, _ = O.train(data, 40000, mbpe.presplit(O.GPT4_SPLIT_PATTERN, data))
    assert model.read_bytes() == O.model_bytes(O.GPT4_SPLIT_PATTERN, want_m)
    _run("-e", "-i", src, "-m", model, "-o", enc)
    _run("-d", "-i", enc, "-m", model, "-o", dec)
    assert dec.read_bytes() == data


def test_e2e_gpt4_vocab_100000(tmp_path):
    # `--vocab-size 100000 --encoder gpt4`: a GPT-4-size vocabulary, beyond the 16-bit slot format (the reference's Token
    # is a uint32_t, Tokenizer.h:37-38): the training continues on 32-bit tokens (csrc/wide.hip).  Model bytes against
    # the oracle, then the reference's round trip.
    model, enc, dec = tmp_path / "huge-model", tmp_path / "enc", tmp_path / "dec"
    src = os.path.join(DATA, "taylorswift.txt")
    out = _run("-t", "-i", src, "-m", model, "-c", "lexical", "--vocab-size", 100000)
    assert "Writing model..." in out
    data = read_data("taylorswift.txt")
    want_m, _ = O.train(data, 100000, mbpe.presplit(O.GPT4_SPLIT_PATTERN, data))
    assert len(want_m) == 100000 - 256
    assert model.read_bytes() == O.model_bytes(O.GPT4_SPLIT_PATTERN, want_m)
    _run("-e", "-i", src, "-m", model, "-o", enc)
    _run("-d", "-i", enc, "-m", model, "-o", dec)
    assert dec.read_bytes() == data

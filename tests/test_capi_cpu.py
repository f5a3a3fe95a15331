"""CPU tests of the C-ABI library: it loads, exports every symbol of
include/mbpe.h, fails loudly without a GPU, and the host pre-split works."""
import ctypes
import os
import re

import numpy as np
import pytest

import mbpe
import oracle as O
from conftest import ROOT, read_data


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "mbpe.h")).read()
    declared = set(re.findall(r"MBPE_API[^;]*?\b(mbpe_\w+)\s*\(", header))
    assert declared == set(mbpe.EXPORTS)
    L = mbpe.lib()
    for s in declared:
        assert hasattr(L, s), s
    assert b"gfx950" in L.mbpe_version()


def test_library_is_built_from_this_tree():
    # build() compiles from source unless the binary carries the hash of exactly these sources (mbpe_version)
    import __graft_entry__ as ge
    want = ge.source_hash()
    assert ge.lib_hash() == want
    assert ("mbpe-src:" + want).encode() in mbpe.lib().mbpe_version()


def _has_gpu():
    import torch
    return torch.cuda.is_available()


@pytest.mark.skipif(_has_gpu(), reason="checks the no-device failure path")
def test_no_silent_cpu_fallback():
    with pytest.raises(mbpe.MbpeError) as e:
        mbpe.Trainer(0)
    assert e.value.code == mbpe.ERR_NO_DEVICE


def test_split_patterns_are_the_reference_strings():
    assert mbpe.split_pattern("gpt2") == O.GPT2_SPLIT_PATTERN
    assert mbpe.split_pattern("gpt4") == O.GPT4_SPLIT_PATTERN
    assert mbpe.split_pattern("basic") == ""
    with pytest.raises(ValueError):
        mbpe.split_pattern("gpt5")


def test_presplit_chunk_counts():
    # taylorswift gpt4 = 46,196 chunks; shakespeare gpt4 = 263,198
    ts = read_data("taylorswift.txt")
    off = mbpe.presplit(O.GPT4_SPLIT_PATTERN, ts)
    assert len(off) - 1 == 46196 and off[0] == 0 and off[-1] == len(ts)
    assert np.all(np.diff(off.astype(np.int64)) > 0)
    sh = read_data("shakespeare.txt")
    assert len(mbpe.presplit(O.GPT4_SPLIT_PATTERN, sh)) - 1 == 263198


def test_presplit_against_python_regex():
    regex = pytest.importorskip("regex")
    data = read_data("sample.txt")          # UTF-8 with emoji
    text = data.decode("utf-8")
    for pat in (O.GPT2_SPLIT_PATTERN, O.GPT4_SPLIT_PATTERN):
        off = mbpe.presplit(pat, data)
        pieces = [data[off[i]:off[i + 1]].decode("utf-8") for i in range(len(off) - 1)]
        assert pieces == regex.findall(pat, text)


def test_presplit_basic_and_empty():
    assert mbpe.presplit("", b"hello").tolist() == [0, 5]
    assert mbpe.presplit(O.GPT4_SPLIT_PATTERN, b"").tolist() == [0]


def test_presplit_bad_pattern():
    with pytest.raises(mbpe.MbpeError) as e:
        mbpe.presplit("(unclosed", b"abc")
    assert e.value.code == mbpe.ERR_REGEX


@pytest.mark.parametrize("enc", ["gpt2", "gpt4"])
def test_presplit_parallel_equals_sequential(enc, monkeypatch):
    """Texts of 8 MiB and more are split by several host threads (each from a guessed offset, stitched
    to the sequential walk): the chunk offsets must be those of the single-threaded loop exactly --
    ASCII text, text with multi-byte characters (sample.txt has emoji), long whitespace runs."""
    import mbpe
    from conftest import read_data
    pattern = mbpe.split_pattern(enc)
    texts = [
        (read_data("shakespeare.txt") + b"\n") * 9,
        (read_data("sample.txt") + b" \r\n\t  ") * 400,
        (b"word " * 1000 + b" " * 70000 + b"\n" * 3000 + "\u00e9\u4e2d\U0001F600 x1234567 ".encode()) * 120,
    ]
    for text in texts:
        assert len(text) >= (8 << 20)
        monkeypatch.setenv("MBPE_SPLIT_THREADS", "1")
        want = mbpe.presplit(pattern, text)
        for threads in ("2", "3", "7", "16"):
            monkeypatch.setenv("MBPE_SPLIT_THREADS", threads)
            got = mbpe.presplit(pattern, text)
            assert np.array_equal(got, want), (enc, threads, len(text))

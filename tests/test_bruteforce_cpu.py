"""An independent pin of the golden .model fixtures: a from-scratch brute-force BPE in numpy --
full recount of every adjacent pair before every merge, no incremental counts, no oracle code,
no product code on the training path -- must reproduce the committed fixtures byte for byte.

What it restates (reference, code/include/):
  * counts = overlapping-window adjacent pairs inside chunks (calculate_freqs, Tokenizer.h:127-146)
  * lexical: max count, then smallest (first, second) (CompareLexicalOrder, PairCount.h:194-207)
  * first:   max count, then the pair whose first occurrence in scan order comes first -- the
             insertion order of a table rebuilt before every merge (PairCount.h:65-74, :141-152;
             recount Tokenizer.h:581-585)
  * merge:   left to right, greedy, non-overlapping (Tokenizer.h:162-199 / :202-306)
Chunk boundaries of the gpt4 fixtures come from Python's `regex` module, not from PCRE2.

The fixtures' digests were recorded in SURVEY.md 8c by the survey session; this test is what ties
them to the algorithm independently of oracle/bpe_oracle.c."""
import numpy as np
import pytest

from conftest import read_data, read_golden

GPT4 = (r"""'(?i:[sdmt]|ll|ve|re)|[^\r\n\p{L}\p{N}]?+\p{L}+|\p{N}{1,3}| ?[^\s\p{L}\p{N}]++[\r\n]*|"""
        r"""\s*[\r\n]|\s+(?!\S)|\s+""")


def brute_force_bpe(data, n_merges, ends=None, mode="lexical"):
    """data: bytes; ends: bool array, True where a byte is the last of its chunk (None: one chunk)."""
    s = np.frombuffer(data, dtype=np.uint8).astype(np.int64)
    e = np.zeros(len(s), dtype=bool) if ends is None else ends.copy()
    V = 256 + n_merges
    merges = []
    for k in range(n_merges):
        ok = ~e[:-1]                                   # a pair starts here unless the token ends its chunk
        key = (s[:-1] * V + s[1:])[ok]
        if len(key) == 0:
            break
        cnt = np.bincount(key, minlength=V * V)
        top = cnt.max()
        if mode == "lexical":
            best = int(np.argmax(cnt))                 # first maximum = smallest (first, second)
        else:
            tied = cnt[key] == top                     # in scan order: the first position holding a top pair
            best = int(key[np.argmax(tied)])
        a, b = divmod(best, V)
        x = 256 + k
        merges.append((a, b))
        hit = np.flatnonzero((s[:-1] == a) & (s[1:] == b) & ok)
        if a == b and len(hit) > 1:
            # overlapping candidates form runs hit[i+1] == hit[i] + 1: keep every second one of a run
            start = np.r_[True, np.diff(hit) != 1]
            run_first = np.maximum.accumulate(np.where(start, np.arange(len(hit)), 0))
            hit = hit[(np.arange(len(hit)) - run_first) % 2 == 0]
        s[hit] = x
        e[hit] = e[hit + 1]                            # the merged token ends the chunk if its second half did
        keep = np.ones(len(s), dtype=bool)
        keep[hit + 1] = False
        s, e = s[keep], e[keep]
    return merges


def model_bytes(pattern, merges):
    # the .model writer, Tokenizer.h:875-893 (no special tokens)
    out = ["minbpe v1", pattern, "0"] + ["%d %d" % m for m in merges]
    return ("\n".join(out) + "\n").encode()


def gpt4_ends(data):
    regex = pytest.importorskip("regex")
    text = data.decode("utf-8")
    ends = np.zeros(len(data), dtype=bool)
    pos = 0
    for m in regex.finditer(GPT4, text):
        pos += len(m.group().encode("utf-8"))
        assert m.group()
        ends[pos - 1] = True
    assert pos == len(data)
    return ends


@pytest.mark.parametrize("name,inp", [("taylorswift_basic_lexical_512", "taylorswift.txt"),
                                      ("shakespeare_basic_lexical_512", "shakespeare.txt")])
def test_basic_lexical_fixtures(name, inp):
    merges = brute_force_bpe(read_data(inp), 256)
    assert model_bytes("", merges) == read_golden(name + ".model")


def test_basic_first_fixture():
    merges = brute_force_bpe(read_data("taylorswift.txt"), 256, mode="first")
    assert model_bytes("", merges) == read_golden("taylorswift_basic_first_512.model")


@pytest.mark.parametrize("name,mode", [("taylorswift_gpt4_lexical_512", "lexical"),
                                       ("taylorswift_gpt4_first_512", "first")])
def test_gpt4_fixtures(name, mode):
    data = read_data("taylorswift.txt")
    merges = brute_force_bpe(data, 256, ends=gpt4_ends(data), mode=mode)
    assert model_bytes(GPT4, merges) == read_golden(name + ".model")


def test_small_kats():
    # small.txt first mode stops after 7 merges' worth of pairs; "aaaa" run parity
    small = read_data("small.txt")
    assert brute_force_bpe(small, 7) == [(98, 99), (100, 101), (256, 257), (258, 258), (97, 259), (258, 10), (260, 261)]
    assert brute_force_bpe(small, 7, mode="first") == [(98, 99), (256, 100), (257, 101), (258, 258), (97, 259),
                                                        (260, 258), (261, 10)]
    assert brute_force_bpe(b"aaaa", 2) == [(97, 97), (256, 256)]

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

  * encode:  passes of "replace ANY pair found in merges_lookup, left to right" until a pass replaces
             nothing (internal_internal_encode, Tokenizer.h:325-367); special tokens are cut out first
             (split_on_special, :605-650) and a special token is one id
Chunk boundaries of the gpt4 / gpt2 fixtures come from Python's `regex` module, not from PCRE2; the
SplitMix64 corpora from the generator restated here, not from oracle/.

The fixtures' digests were recorded in SURVEY.md 8c by the survey session; this test is what ties
every one of them (ten .model files, three encode vectors) to the algorithm independently of
oracle/bpe_oracle.c."""
import hashlib

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
        if V * V <= 16 * len(key):
            cnt = np.bincount(key, minlength=V * V)
            top = cnt.max()
            if mode == "lexical":
                best = int(np.argmax(cnt))             # first maximum = smallest (first, second)
            else:
                tied = cnt[key] == top                 # in scan order: the first position holding a top pair
                best = int(key[np.argmax(tied)])
        else:                                          # many ids, short stream: count the pairs that occur
            uk, inv, uc = np.unique(key, return_inverse=True, return_counts=True)
            if mode == "lexical":
                best = int(uk[np.argmax(uc)])          # uk ascends: the first maximum is the smallest key
            else:
                best = int(key[np.argmax(uc[inv] == uc.max())])
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
    return regex_ends(GPT4, data)


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


# ---- the fixtures the first version of this file left out ------------------------------------------

GPT2 = r"""'(?:[sdmt]|ll|ve|re)| ?\p{L}+| ?\p{N}+| ?[^\s\p{L}\p{N}]+|\s+(?!\S)|\s+"""


def regex_ends(pattern, data):
    regex = pytest.importorskip("regex")
    text = data.decode("utf-8")
    ends = np.zeros(len(data), dtype=bool)
    pos = 0
    for m in regex.finditer(pattern, text):
        assert m.group()
        pos += len(m.group().encode("utf-8"))
        ends[pos - 1] = True
    assert pos == len(data)          # (these patterns leave no gaps on these files)
    return ends


def splitmix64_bytes(seed, n):
    """s += 0x9E3779B97F4A7C15; z = s; z = (z ^ z >> 30) * 0xBF58476D1CE4E5B9;
    z = (z ^ z >> 27) * 0x94D049BB133111EB; z ^= z >> 31; emitted little-endian; byte 0 never NUL."""
    m = (n + 7) // 8
    with np.errstate(over="ignore"):
        s = np.uint64(seed) + np.arange(1, m + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
        z = s
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    out = z.astype("<u8").view(np.uint8)[:n].copy()
    if out[0] == 0:
        out[0] = 1
    return out.tobytes()


def test_splitmix_generator_known_answers():
    # first 8 bytes and the sha256 prefixes of the two fixture corpora
    d = splitmix64_bytes(42, 1 << 20)
    assert d[:8].hex() == "956eeb2f2632d7bd"
    assert hashlib.sha256(d).hexdigest().startswith("5b2605c7")
    assert hashlib.sha256(d[:1 << 16]).hexdigest().startswith("c96c46e4")


def test_shakespeare_gpt4_fixture():
    # BASELINE config 2 names no encoder; the CLI's default is gpt4 (minbpe-cc.cpp:120)
    data = read_data("shakespeare.txt")
    merges = brute_force_bpe(data, 256, ends=regex_ends(GPT4, data))
    assert model_bytes(GPT4, merges) == read_golden("shakespeare_gpt4_lexical_512.model")


def test_taylorswift_gpt2_fixture():
    data = read_data("taylorswift.txt")
    assert read_golden("taylorswift_gpt2_lexical_512.model").split(b"\n")[1].decode() == GPT2
    merges = brute_force_bpe(data, 256, ends=regex_ends(GPT2, data))
    assert model_bytes(GPT2, merges) == read_golden("taylorswift_gpt2_lexical_512.model")


def test_splitmix_64KiB_fixture():
    merges = brute_force_bpe(splitmix64_bytes(42, 1 << 16), 2048 - 256)
    assert model_bytes("", merges) == read_golden("splitmix42_64KiB_basic_lexical_2048.model")


def test_splitmix_1MiB_fixtures():
    # (the order of merges does not depend on the target: the 512 model is the 1024 model's first 256 merges)
    merges = brute_force_bpe(splitmix64_bytes(42, 1 << 20), 1024 - 256)
    assert model_bytes("", merges[:256]) == read_golden("splitmix42_1MiB_basic_lexical_512.model")
    assert model_bytes("", merges) == read_golden("splitmix42_1MiB_basic_lexical_1024.model")


# ---- encode (Tokenizer.h:325-377, :605-722) ----------------------------------------------------------

def parse_model(raw):
    lines = raw.decode().split("\n")
    assert lines[0] == "minbpe v1"
    n_special = int(lines[2])
    special = {}
    for ln in lines[3:3 + n_special]:
        name, idx = ln.rsplit(" ", 1)
        special[name] = int(idx)
    merges = [tuple(int(v) for v in ln.split()) for ln in lines[3 + n_special:] if ln]
    return lines[1], special, merges


def brute_force_encode_chunk(tokens, lookup):
    """internal_internal_encode: one left-to-right pass replaces every pair found in the lookup
    (whatever its rank; after a replacement the scan continues behind it), repeated until a pass
    replaces nothing."""
    while True:
        out, i, merged = [], 0, False
        while i < len(tokens):
            if i + 1 < len(tokens) and (tokens[i], tokens[i + 1]) in lookup:
                out.append(lookup[(tokens[i], tokens[i + 1])])
                i += 2
                merged = True
            else:
                out.append(tokens[i])
                i += 1
        tokens = out
        if not merged:
            return tokens


def brute_force_encode(text_bytes, pattern, special, merges):
    regex = pytest.importorskip("regex")
    lookup = {m: 256 + k for k, m in enumerate(merges)}
    text = text_bytes.decode("utf-8")
    # split_on_special: earliest occurrence of any special token, cut, continue behind it
    parts, pos = [], 0
    while True:
        nxt = None
        for name in special:
            at = text.find(name, pos)
            if at >= 0 and (nxt is None or at < nxt[0]):
                nxt = (at, name)
        if nxt is None:
            parts.append((text[pos:], None))
            break
        parts.append((text[pos:nxt[0]], None))
        parts.append((nxt[1], special[nxt[1]]))
        pos = nxt[0] + len(nxt[1])
    out = []
    for piece, sid in parts:
        if sid is not None:
            out.append(sid)
            continue
        chunks = [m.group() for m in regex.finditer(pattern, piece)] if pattern else ([piece] if piece else [])
        for ch in chunks:
            out.extend(brute_force_encode_chunk(list(ch.encode("utf-8")), lookup))
    return out


def test_encode_vectors():
    # SURVEY.md 8c, the three encode known answers
    pat, special, merges = parse_model(read_golden("taylorswift_gpt4_lexical_512.model"))
    ids = brute_force_encode(read_data("taylorswift.txt"), pat, special, merges)
    assert len(ids) == 94201
    assert hashlib.sha256(np.asarray(ids, dtype="<u4").tobytes()).hexdigest() == \
        "1b82232e30d1972b1b9f8b54080fc8757bcce310b6b8f9de4d63fdd18f034d0d"

    pat, special, merges = parse_model(read_golden("shakespeare_basic_lexical_512.model"))
    ids = brute_force_encode(read_data("sample.txt"), pat, special, merges)
    assert len(ids) == 15677
    assert hashlib.sha256(np.asarray(ids, dtype="<u4").tobytes()).hexdigest() == \
        "624874b4a8bce9405f0a89ecb7b3e7eeaa94b2a3235e88c05acd6426c05cb409"

    # endtoend-test.sh:13-16: the gpt4 / first model with the special tokens of special1.txt
    pat, _, merges = parse_model(read_golden("taylorswift_gpt4_first_512.model"))
    special = {}
    for ln in read_data("special1.txt").decode().split("\n"):
        if ln.strip():
            name, idx = ln.rsplit(" ", 1)
            special[name] = int(idx)
    ids = brute_force_encode(read_data("specialtokensample.txt"), pat, special, merges)
    assert ids == [84, 104, 355, 32, 355, 306, 288, 101, 261, 101, 120, 116, 434, 287, 349, 262, 116, 97, 259, 115, 32,
                   100258, 261, 119, 111, 306, 112, 310, 478, 108, 32, 100257, 348, 107, 290, 115, 46]

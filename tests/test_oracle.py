"""CPU tests: the oracle (oracle/bpe_oracle.c) against the reference's own
known-answer tests and the golden .model fixtures (SURVEY.md 8c digests)."""
import hashlib
import json
import os

import numpy as np
import pytest

import oracle as O
import mbpe
from conftest import GOLDEN, read_data, read_golden

INDEX = json.load(open(os.path.join(GOLDEN, "index.json")))


def _input(spec):
    if spec.startswith("splitmix:"):
        _, seed, n = spec.split(":")
        return O.splitmix64_bytes(int(seed), int(n)).tobytes()
    return read_data(spec)


# ---- reference test.cpp:82-106, "PairCountLexicalOrder get most frequent" ----
def test_kat_lexical_order_top():
    pc = O.PairCountTable(lexical=True)
    assert pc.get_top_pair_count() is None
    pc.create_or_modify_pair(1, 2, 1)
    assert pc.get_top_pair_count() == (1, 2)
    pc.create_or_modify_pair(1, 2, 1)
    pc.create_or_modify_pair(2, 3, 1)
    assert pc.get_top_pair_count() == (1, 2)
    pc.create_or_modify_pair(2, 3, 1)       # tie at 2: (1,2) is lexically smaller
    assert pc.get_top_pair_count() == (1, 2)
    pc.create_or_modify_pair(0, 1, 3)       # (0,1)=3 ties nothing above it
    assert pc.get_top_pair_count() == (0, 1)


# ---- reference test.cpp:15-80, PairCountInsertOrder ----
def test_kat_insert_order():
    pc = O.PairCountTable(lexical=False)
    pc.create_or_modify_pair(10, 20, 1)
    assert pc.get_count() == 1
    pc.create_or_modify_pair(30, 40, 1)
    assert pc.get_count() == 2
    assert pc.get_pair(10, 20) == 1 and pc.get_pair(30, 40) == 1

    pc = O.PairCountTable(lexical=False)
    assert pc.get_count() == 0
    pc.create_or_modify_pair(1, 2, 1)
    pc.create_or_modify_pair(1, 2, 1)
    assert pc.get_count() == 1
    pc.create_or_modify_pair(2, 3, 1)
    assert pc.get_count() == 2

    pc = O.PairCountTable(lexical=False)
    assert pc.get_top_pair_count() is None
    pc.create_or_modify_pair(1, 2, 1)
    assert pc.get_top_pair_count() == (1, 2)
    pc.create_or_modify_pair(1, 2, 1)
    pc.create_or_modify_pair(2, 3, 1)
    assert pc.get_top_pair_count() == (1, 2)
    pc.create_or_modify_pair(2, 3, 1)
    pc.create_or_modify_pair(2, 3, 1)
    assert pc.get_top_pair_count() == (2, 3)
    pc.create_or_modify_pair(1, 2, 1)       # 3 vs 3: first inserted wins
    assert pc.get_top_pair_count() == (1, 2)
    pc.create_or_modify_pair(1, 2, 1)
    assert pc.get_top_pair_count() == (1, 2)


# ---- reference test.cpp:136-186, "Tokenizer training" on "abcbcde" (FIRST mode) ----
def test_kat_abcbcde():
    st = O.State(b"abcbcde", mode=O.FIRST)
    toks, _ = st.stream()
    assert len(toks) == 7
    assert st.top()[:2] == (98, 99)
    st.merge(98, 99, 256)
    assert st.top()[:2] == (97, 256)
    st.merge(97, 256, 257)
    assert st.top()[:2] == (257, 256)
    assert st.get_pair(256, 100) == 1
    assert st.get_pair(257, 256) == 1
    assert st.get_pair(100, 101) == 1
    assert len(st.table()[0]) == 3


# ---- SURVEY.md 8c KATs: zero-count retention and run parity ----
def test_kat_small_txt_zero_count_quirk():
    m, c = O.train(read_data("small.txt"), 275)
    want = [[98, 99], [100, 101], [256, 257], [258, 258], [97, 259], [258, 10], [260, 261]] + [[97, 98]] * 12
    assert m.tolist() == want
    assert c.tolist()[7:] == [0] * 12


def test_kat_small_txt_first_breaks():
    m, _ = O.train(read_data("small.txt"), 275, mode=O.FIRST)
    assert m.tolist() == [[98, 99], [256, 100], [257, 101], [258, 258], [97, 259], [260, 258], [261, 10]]


def test_kat_aaaa():
    m, c = O.train(b"aaaa", 262)
    assert m.tolist() == [[97, 97], [256, 256], [97, 97], [97, 97], [97, 97], [97, 97]]
    assert c.tolist() == [3, 1, 0, 0, 0, 0]


def test_nul_quirk_collapses_chunk():
    # Tokenizer.h:86-93: "\0" + digits -> one token -> no pairs -> no merges
    m, _ = O.train(b"\x00123abcabc", 260)
    assert len(m) == 0
    m, _ = O.train(b"\x00abcabc", 258)   # stoi fails -> normal text
    assert len(m) == 2


def test_empty_and_single_byte():
    assert len(O.train(b"", 300)[0]) == 0
    assert len(O.train(b"a", 300)[0]) == 0
    m, _ = O.train(b"ab", 258)
    assert m.tolist() == [[97, 98], [97, 98]]


# ---- golden .model fixtures: bytes and digests ----
@pytest.mark.parametrize("name", sorted(INDEX))
def test_golden_model(name):
    meta = INDEX[name]
    blob = read_golden(name + ".model")
    assert hashlib.sha256(blob).hexdigest() == meta["sha256"]
    data = _input(meta["input"])
    enc = meta["encoder"]
    off = None if enc == "basic" else mbpe.presplit(O.PATTERNS[enc], data)
    mode = O.LEXICAL if meta["mode"] == "lexical" else O.FIRST
    merges, counts = O.train(data, meta["vocab"], off, mode)
    assert O.model_bytes(O.PATTERNS[enc], merges) == blob
    assert int(counts[0]) == meta["first_count"] and int(counts[-1]) == meta["last_count"]
    pat, specials, m2 = O.parse_model(blob)
    assert pat == O.PATTERNS[enc] and specials == [] and m2.tolist() == merges.tolist()


def test_splitmix_corpus_digest():
    d = O.splitmix64_bytes(42, 1 << 20)
    assert d[:8].tobytes().hex() == "956eeb2f2632d7bd"
    assert hashlib.sha256(d.tobytes()).hexdigest().startswith("5b2605c7")
    assert hashlib.sha256(O.splitmix64_bytes(42, 1 << 16).tobytes()).hexdigest().startswith("c96c46e4")


def test_pair_count_matches_state_table():
    data = read_data("taylorswift.txt")
    off = mbpe.presplit(O.PATTERNS["gpt4"], data)
    for o in (None, off):
        table = O.pair_count_u8(data, o)
        st = O.State(data, o)
        d = st.table_dict()
        assert int(table.sum()) == sum(d.values())
        assert {(i >> 8, i & 255): int(table[i]) for i in np.nonzero(table)[0]} == d


def test_incremental_counts_stay_exact():
    """After every merge the incrementally maintained table equals a recount
    of the stream (SURVEY.md 8-S rule 3), zero-count members aside."""
    rng = np.random.default_rng(3)
    data = rng.integers(97, 101, size=400, dtype=np.uint8).tobytes()
    st = O.State(data)
    for i in range(40):
        a, b, _ = st.top()
        st.merge(a, b, 256 + i)
        toks, _ = st.stream()
        recount = {}
        for x, y in zip(toks[:-1], toks[1:]):
            recount[(int(x), int(y))] = recount.get((int(x), int(y)), 0) + 1
        assert {k: v for k, v in st.table_dict().items() if v} == recount


def test_oracle_known_answers_under_asan_ubsan():
    """SURVEY.md section 5 (sanitizers on the host code): the oracle built with -fsanitize=address,undefined
    (oracle/Makefile `asan`) runs the known-answer tests, the NUL quirk, the tiny inputs and the incremental-count
    check of this file in a child interpreter; any sanitizer report fails it."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    orc = os.path.join(os.path.dirname(here), "oracle")
    subprocess.check_call(["make", "-C", orc, "asan"], stdout=subprocess.DEVNULL)
    libasan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"]).decode().strip()
    libubsan = subprocess.check_output(["gcc", "-print-file-name=libubsan.so"]).decode().strip()
    env = dict(os.environ, BPE_ORACLE_LIB=os.path.join(orc, "libbpe_oracle_asan.so"),
               LD_PRELOAD=libasan + ":" + libubsan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider", os.path.abspath(__file__), "-k",
                        "kat or nul_quirk or empty_and_single or incremental_counts"],
                       env=env, cwd=os.path.dirname(here), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "passed" in r.stdout and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr

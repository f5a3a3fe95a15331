"""ctypes front end of the CPU oracle (oracle/bpe_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product (minbpe-cc_amd/, include/) never
imports this module.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libbpe_oracle.so")

LEXICAL = 0
FIRST = 1

# Tokenizer.h:59-60 (the text that goes into line 2 of a .model file)
GPT2_SPLIT_PATTERN = r"""'(?:[sdmt]|ll|ve|re)| ?\p{L}+| ?\p{N}+| ?[^\s\p{L}\p{N}]+|\s+(?!\S)|\s+"""
GPT4_SPLIT_PATTERN = r"""'(?i:[sdmt]|ll|ve|re)|[^\r\n\p{L}\p{N}]?+\p{L}+|\p{N}{1,3}| ?[^\s\p{L}\p{N}]++[\r\n]*|\s*[\r\n]|\s+(?!\S)|\s+"""
PATTERNS = {"basic": "", "gpt2": GPT2_SPLIT_PATTERN, "gpt4": GPT4_SPLIT_PATTERN}


def build(force=False):
    # (BPE_ORACLE_LIB: another build of the same source, e.g. `make asan`'s libbpe_oracle_asan.so)
    if os.environ.get("BPE_ORACLE_LIB"):
        return os.environ["BPE_ORACLE_LIB"]
    src = os.path.join(_HERE, "bpe_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libbpe_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = ctypes.CDLL(build())
    u8p = ctypes.c_void_p
    u32p = ctypes.c_void_p
    L.orc_create.restype = ctypes.c_void_p
    L.orc_create.argtypes = [u8p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int]
    L.orc_destroy.argtypes = [ctypes.c_void_p]
    L.orc_top.argtypes = [ctypes.c_void_p, u32p, u32p, ctypes.c_void_p]
    L.orc_merge.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]
    L.orc_train_loop.restype = ctypes.c_uint32
    L.orc_train_loop.argtypes = [ctypes.c_void_p, ctypes.c_uint32, u32p, ctypes.c_void_p]
    L.orc_train.restype = ctypes.c_uint32
    L.orc_train.argtypes = [u8p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64,
                            ctypes.c_uint32, ctypes.c_int, u32p, ctypes.c_void_p]
    L.orc_stream_len.restype = ctypes.c_uint64
    L.orc_stream_len.argtypes = [ctypes.c_void_p]
    L.orc_stream.argtypes = [ctypes.c_void_p, u32p, ctypes.c_void_p]
    L.orc_table_size.restype = ctypes.c_uint64
    L.orc_table_size.argtypes = [ctypes.c_void_p]
    L.orc_table_dump.restype = ctypes.c_uint64
    L.orc_table_dump.argtypes = [ctypes.c_void_p, u32p, u32p, ctypes.c_void_p]
    L.orc_get_pair.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p]
    L.orc_pair_count_u8.argtypes = [u8p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64, u32p]
    L.orc_pc_new.restype = ctypes.c_void_p
    L.orc_pc_new.argtypes = [ctypes.c_int]
    L.orc_pc_free.argtypes = [ctypes.c_void_p]
    L.orc_pc_add.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int32]
    L.orc_pc_get.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p]
    L.orc_pc_count.restype = ctypes.c_uint64
    L.orc_pc_count.argtypes = [ctypes.c_void_p]
    L.orc_pc_top.argtypes = [ctypes.c_void_p, u32p, u32p, ctypes.c_void_p]
    L.orc_encoder_new.restype = ctypes.c_void_p
    L.orc_encoder_new.argtypes = [u32p, ctypes.c_uint32]
    L.orc_encoder_free.argtypes = [ctypes.c_void_p]
    L.orc_encode_chunk.restype = ctypes.c_uint64
    L.orc_encode_chunk.argtypes = [ctypes.c_void_p, u32p, ctypes.c_uint64]
    L.orc_text_to_vector.restype = ctypes.c_uint64
    L.orc_text_to_vector.argtypes = [u8p, ctypes.c_uint64, u32p]
    _lib = L
    return L


def _u8(data):
    a = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data
    return np.ascontiguousarray(a, dtype=np.uint8)


def _off(chunk_off):
    if chunk_off is None:
        return None, 0, None
    a = np.ascontiguousarray(chunk_off, dtype=np.uint64)
    return a, len(a) - 1, a.ctypes.data


def train(data, vocab_size, chunk_off=None, mode=LEXICAL):
    """Tokenizer::train hot path. Returns (merges [k,2] uint32, counts [k] int32)."""
    text = _u8(data)
    off, n_chunks, off_p = _off(chunk_off)
    cap = max(vocab_size - 256, 0)
    merges = np.zeros((cap, 2), dtype=np.uint32)
    counts = np.zeros(cap, dtype=np.int32)
    k = lib().orc_train(text.ctypes.data, len(text), off_p, n_chunks, vocab_size, mode,
                        merges.ctypes.data, counts.ctypes.data)
    return merges[:k].copy(), counts[:k].copy()


def pair_count_u8(data, chunk_off=None):
    text = _u8(data)
    off, n_chunks, off_p = _off(chunk_off)
    table = np.zeros(65536, dtype=np.uint32)
    lib().orc_pair_count_u8(text.ctypes.data, len(text), off_p, n_chunks, table.ctypes.data)
    return table


class State:
    """Step-by-step trainer state (argmax / merge / dump) for step-level parity."""

    def __init__(self, data, chunk_off=None, mode=LEXICAL):
        self._text = _u8(data)
        self._off, self.n_chunks, off_p = _off(chunk_off)
        if chunk_off is None:
            self.n_chunks = 1
        self._h = lib().orc_create(self._text.ctypes.data, len(self._text), off_p,
                                   0 if chunk_off is None else self.n_chunks, mode)

    def close(self):
        if self._h:
            lib().orc_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def top(self):
        a = ctypes.c_uint32()
        b = ctypes.c_uint32()
        c = ctypes.c_int32()
        ok = lib().orc_top(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
        return (a.value, b.value, c.value) if ok else None

    def merge(self, a, b, new_id):
        lib().orc_merge(self._h, a, b, new_id)

    def stream(self):
        n = lib().orc_stream_len(self._h)
        out = np.zeros(n, dtype=np.uint32)
        clen = np.zeros(self.n_chunks, dtype=np.uint64)
        lib().orc_stream(self._h, out.ctypes.data, clen.ctypes.data)
        return out, clen

    def table(self):
        n = lib().orc_table_size(self._h)
        a = np.zeros(n, dtype=np.uint32)
        b = np.zeros(n, dtype=np.uint32)
        c = np.zeros(n, dtype=np.int32)
        lib().orc_table_dump(self._h, a.ctypes.data, b.ctypes.data, c.ctypes.data)
        return a, b, c

    def table_dict(self):
        a, b, c = self.table()
        return {(int(x), int(y)): int(z) for x, y, z in zip(a, b, c)}

    def get_pair(self, a, b):
        c = ctypes.c_int32()
        return c.value if lib().orc_get_pair(self._h, a, b, ctypes.byref(c)) else None


class PairCountTable:
    """PairCountLexicalOrder / PairCountInsertOrder (PairCount.h) for the KATs."""

    def __init__(self, lexical=True):
        self._h = lib().orc_pc_new(1 if lexical else 0)

    def __del__(self):
        if self._h:
            lib().orc_pc_free(self._h)
            self._h = None

    def create_or_modify_pair(self, a, b, freq):
        return bool(lib().orc_pc_add(self._h, a, b, freq))

    def get_pair(self, a, b):
        c = ctypes.c_int32()
        return c.value if lib().orc_pc_get(self._h, a, b, ctypes.byref(c)) else None

    def get_count(self):
        return lib().orc_pc_count(self._h)

    def get_top_pair_count(self):
        a = ctypes.c_uint32()
        b = ctypes.c_uint32()
        c = ctypes.c_int32()
        ok = lib().orc_pc_top(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
        return (a.value, b.value) if ok else None


def model_bytes(pattern, merges, special_tokens=()):
    """The 'minbpe v1' file Tokenizer::save writes, Tokenizer.h:875-893."""
    out = ["minbpe v1\n", pattern + "\n", "%d\n" % len(special_tokens)]
    for name, idx in special_tokens:
        out.append("%s %d\n" % (name, idx))
    for a, b in merges:
        out.append("%d %d\n" % (int(a), int(b)))
    return "".join(out).encode("utf-8")


def parse_model(blob):
    """Tokenizer::load, Tokenizer.h:754-872 -> (pattern, specials, merges)."""
    text = blob.decode("utf-8")
    lines = text.split("\n")
    assert lines[0] == "minbpe v1"
    pattern = lines[1]
    rest = "\n".join(lines[2:]).split()
    n_special = int(rest[0])
    specials = [(rest[1 + 2 * i], int(rest[2 + 2 * i])) for i in range(n_special)]
    nums = rest[1 + 2 * n_special:]
    merges = np.array([int(x) for x in nums], dtype=np.uint32).reshape(-1, 2)
    return pattern, specials, merges


def encode_chunks(data, chunk_off, merges):
    """internal_encode over chunks (Tokenizer.h:370-377), flattened (:713-717)."""
    text = _u8(data)
    merges = np.ascontiguousarray(merges, dtype=np.uint32)
    enc = lib().orc_encoder_new(merges.ctypes.data, len(merges))
    out = []
    try:
        if chunk_off is None:
            chunk_off = [0, len(text)]
        for c in range(len(chunk_off) - 1):
            s, e = int(chunk_off[c]), int(chunk_off[c + 1])
            buf = np.zeros(max(e - s, 1), dtype=np.uint32)
            seg = text[s:e]
            n = lib().orc_text_to_vector(seg.ctypes.data if e > s else None, e - s, buf.ctypes.data)
            n = lib().orc_encode_chunk(enc, buf.ctypes.data, n)
            out.append(buf[:n].copy())
    finally:
        lib().orc_encoder_free(enc)
    return np.concatenate(out) if out else np.zeros(0, dtype=np.uint32)


def vocab_from_merges(merges):
    """vocab rebuild of Tokenizer::load, Tokenizer.h:843-861."""
    vocab = [bytes([i]) for i in range(256)]
    for a, b in merges:
        vocab.append(vocab[int(a)] + vocab[int(b)])
    return vocab


def decode(tokens, merges):
    """Tokenizer::decode without special tokens, Tokenizer.h:725-751."""
    vocab = vocab_from_merges(merges)
    return b"".join(vocab[int(t)] for t in tokens if int(t) < len(vocab))


def splitmix64_bytes(seed, n):
    """SplitMix64 corpus of SURVEY.md 8d: z as 8 little-endian bytes; byte 0 forced non-zero."""
    m = (n + 7) // 8
    mask = (1 << 64) - 1
    idx = np.arange(1, m + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    del mask
    b = z.astype("<u8").view(np.uint8)[:n].copy()
    if n and b[0] == 0:
        b[0] = 1
    return b

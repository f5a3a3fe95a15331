"""ctypes binding of the C-ABI in include/mbpe.h (libmbpe.so).

Used by tests/, bench.py and __graft_entry__.py.  The library itself is the
product; this file only marshals numpy arrays / device pointers into it.
There is no fallback: if libmbpe.so is missing, import-time use raises.
"""
import ctypes
import os

import numpy as np

_PKG_DIR = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LIB_PATH = os.environ.get("MBPE_LIB") or os.path.join(_PKG_DIR, "libmbpe.so")

OK = 0
NEED_EXCHANGE = 1
ERR_ARG, ERR_NO_DEVICE, ERR_HIP, ERR_VOCAB, ERR_STATE = -1, -2, -3, -4, -5
ERR_OOM, ERR_REGEX, ERR_SPLIT_GAP, ERR_COMM, ERR_OVERFLOW, ERR_IO = -6, -7, -8, -9, -10, -11
COMM_ID_BYTES = 128

# every symbol include/mbpe.h declares
EXPORTS = [
    "mbpe_last_error", "mbpe_version", "mbpe_create", "mbpe_destroy", "mbpe_load_corpus",
    "mbpe_pair_count_u8", "mbpe_train_begin", "mbpe_train_steps", "mbpe_train_sequences", "mbpe_train_result",
    "mbpe_train_lexical", "mbpe_train", "mbpe_get_stats", "mbpe_get_stream", "mbpe_stream_device", "mbpe_table_device", "mbpe_get_pairs", "mbpe_compact",
    "mbpe_set_option", "mbpe_comm_unique_id", "mbpe_comm_init", "mbpe_comm_init_external",
    "mbpe_comm_exchange_buffer", "mbpe_comm_exchange_done", "mbpe_presplit",
    "mbpe_split_count", "mbpe_split_offsets", "mbpe_split_has_gaps", "mbpe_split_starts", "mbpe_split_ends",
    "mbpe_split_free", "mbpe_load_corpus_ranges", "mbpe_split_pattern", "mbpe_encode_chunks",
]
# include/mbpe_tokenizer.h
TOK_EXPORTS = [
    "mbpe_tok_create", "mbpe_tok_destroy", "mbpe_tok_set_special_tokens", "mbpe_tok_train", "mbpe_tok_set_merges",
    "mbpe_tok_get_merges", "mbpe_tok_save", "mbpe_tok_load", "mbpe_tok_encode", "mbpe_tok_encode_device",
    "mbpe_tok_decode",
]


class Stats(ctypes.Structure):
    _fields_ = [
        ("n_bytes", ctypes.c_uint64), ("n_chunks", ctypes.c_uint64), ("n_slots", ctypes.c_uint64),
        ("n_live", ctypes.c_uint64), ("n_merges", ctypes.c_uint32), ("n_compactions", ctypes.c_uint32),
        ("n_pairs", ctypes.c_uint64), ("ms_pair_count", ctypes.c_float), ("ms_begin", ctypes.c_float),
        ("ms_steps", ctypes.c_float), ("pair_count_launches", ctypes.c_uint32), ("merge_launches", ctypes.c_uint32),
        ("ms_merge_kernel", ctypes.c_float), ("n_batches", ctypes.c_uint32),
        ("n_fused", ctypes.c_uint32), ("n_fused_dropped", ctypes.c_uint32), ("cut_conflict", ctypes.c_uint32),
        ("cut_bucket", ctypes.c_uint32), ("cut_single", ctypes.c_uint32), ("cut_full", ctypes.c_uint32),
        ("n_validation_drops", ctypes.c_uint32), ("ms_grow_table", ctypes.c_float), ("ms_compact", ctypes.c_float),
        ("n_table_grows", ctypes.c_uint32), ("n_sel_fallback", ctypes.c_uint32),
        ("fused_launches", ctypes.c_uint32), ("ms_fused_kernel", ctypes.c_float), ("fused_slots", ctypes.c_uint64),
        ("n_sel_retry", ctypes.c_uint32), ("adapt_limit", ctypes.c_uint32), ("n_sel_blocks", ctypes.c_uint64), ("size_hist", ctypes.c_uint32 * 8),
        ("n_skipped", ctypes.c_uint32), ("n_skip_cut", ctypes.c_uint32),
        ("exchange_words", ctypes.c_uint64), ("exchanges", ctypes.c_uint32), ("pad_", ctypes.c_uint32),
        ("fused_live_tokens", ctypes.c_uint64), ("ms_pair_count_kernel", ctypes.c_float), ("pad2_", ctypes.c_uint32),
    ]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_}
        d["size_hist"] = list(self.size_hist)
        return d



class MbpeError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("mbpe error %d: %s" % (code, msg))
        self.code = code


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libmbpe.so is not built (%s); run __graft_entry__.build()" % LIB_PATH)
    L = ctypes.CDLL(LIB_PATH)
    vp, u64, u32, i32, i64 = ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int, ctypes.c_int64
    L.mbpe_last_error.restype = ctypes.c_char_p
    L.mbpe_version.restype = ctypes.c_char_p
    L.mbpe_create.argtypes = [i32, ctypes.POINTER(vp)]
    L.mbpe_destroy.argtypes = [vp]
    L.mbpe_destroy.restype = None
    L.mbpe_load_corpus.argtypes = [vp, vp, u64, vp, u64, i32]
    L.mbpe_pair_count_u8.argtypes = [vp, vp]
    L.mbpe_train_begin.argtypes = [vp, u32]
    L.mbpe_train_steps.argtypes = [vp, u32, vp]
    L.mbpe_train_sequences.argtypes = [vp, u32, vp]
    L.mbpe_train_result.argtypes = [vp, vp, vp, u32, vp]
    L.mbpe_train_lexical.argtypes = [vp, vp, u64, vp, u64, u32, vp, vp, vp, vp]
    L.mbpe_train.argtypes = [vp, vp, u64, vp, u64, u32, i32, vp, vp, vp, vp]
    L.mbpe_get_stats.argtypes = [vp, vp]
    L.mbpe_get_stream.argtypes = [vp, vp, vp, u64, vp]
    L.mbpe_stream_device.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(u64), ctypes.POINTER(u32), ctypes.POINTER(u32),
                                     ctypes.POINTER(u32)]
    L.mbpe_table_device.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(u32)]
    L.mbpe_get_pairs.argtypes = [vp, vp, vp, vp, u64, vp]
    L.mbpe_compact.argtypes = [vp]
    L.mbpe_set_option.argtypes = [vp, ctypes.c_char_p, i64]
    L.mbpe_comm_unique_id.argtypes = [vp]
    L.mbpe_comm_init.argtypes = [vp, vp, i32, i32]
    L.mbpe_comm_init_external.argtypes = [vp, i32, i32]
    L.mbpe_comm_exchange_buffer.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(u64)]
    L.mbpe_comm_exchange_done.argtypes = [vp]
    L.mbpe_presplit.argtypes = [ctypes.c_char_p, vp, u64, ctypes.POINTER(vp)]
    L.mbpe_split_count.argtypes = [vp]
    L.mbpe_split_count.restype = u64
    L.mbpe_split_offsets.argtypes = [vp]
    L.mbpe_split_offsets.restype = ctypes.POINTER(ctypes.c_uint64)
    L.mbpe_split_has_gaps.argtypes = [vp]
    L.mbpe_split_starts.argtypes = [vp]
    L.mbpe_split_starts.restype = ctypes.POINTER(ctypes.c_uint64)
    L.mbpe_split_ends.argtypes = [vp]
    L.mbpe_split_ends.restype = ctypes.POINTER(ctypes.c_uint64)
    L.mbpe_load_corpus_ranges.argtypes = [vp, vp, u64, vp, vp, u64, i32]
    L.mbpe_split_free.argtypes = [vp]
    L.mbpe_split_free.restype = None
    L.mbpe_split_pattern.argtypes = [ctypes.c_char_p]
    L.mbpe_split_pattern.restype = ctypes.c_char_p
    L.mbpe_encode_chunks.argtypes = [i32, vp, u64, vp, u64, vp, u32, vp, u64, vp, vp]
    L.mbpe_tok_create.argtypes = [ctypes.c_char_p, ctypes.POINTER(vp)]
    L.mbpe_tok_destroy.argtypes = [vp]
    L.mbpe_tok_destroy.restype = None
    L.mbpe_tok_set_special_tokens.argtypes = [vp, ctypes.c_char_p, u64]
    L.mbpe_tok_train.argtypes = [vp, vp, u64, u32, i32, i32, i32]
    L.mbpe_tok_set_merges.argtypes = [vp, vp, u32]
    L.mbpe_tok_get_merges.argtypes = [vp, vp, u32, vp]
    L.mbpe_tok_save.argtypes = [vp, ctypes.c_char_p, i32]
    L.mbpe_tok_load.argtypes = [vp, ctypes.c_char_p, i32]
    L.mbpe_tok_encode.argtypes = [vp, vp, u64, i32, vp, u64, vp]
    L.mbpe_tok_encode_device.argtypes = [vp, vp, u64, i32, i32, vp, u64, vp]
    L.mbpe_tok_decode.argtypes = [vp, vp, u64, i32, vp, u64, vp]
    _lib = L
    return L


def _check(rc):
    if rc < 0:
        raise MbpeError(rc, lib().mbpe_last_error().decode("utf-8", "replace"))
    return rc


def _u8(data):
    if isinstance(data, np.ndarray):
        return np.ascontiguousarray(data, dtype=np.uint8)
    return np.frombuffer(data, dtype=np.uint8)


def split_pattern(encoder):
    p = lib().mbpe_split_pattern(encoder.encode())
    if p is None:
        raise ValueError("Encoder should be one of: basic, gpt2 or gpt4")
    return p.decode("utf-8")


def presplit_ranges(pattern, data):
    """The same as (starts, ends) arrays: chunks need not tile the text (bytes between matches are skipped)."""
    text = _u8(data)
    h = ctypes.c_void_p()
    _check(lib().mbpe_presplit(pattern.encode("utf-8"), text.ctypes.data if len(text) else None,
                               len(text), ctypes.byref(h)))
    try:
        n = lib().mbpe_split_count(h)
        if n == 0:
            return np.zeros(0, dtype=np.uint64), np.zeros(0, dtype=np.uint64)
        return (np.ctypeslib.as_array(lib().mbpe_split_starts(h), shape=(n,)).copy(),
                np.ctypeslib.as_array(lib().mbpe_split_ends(h), shape=(n,)).copy())
    finally:
        lib().mbpe_split_free(h)


def presplit(pattern, data):
    """Tokenizer::train's regex pre-split (Tokenizer.h:500-540) -> uint64 offsets [n_chunks+1]."""
    text = _u8(data)
    h = ctypes.c_void_p()
    _check(lib().mbpe_presplit(pattern.encode("utf-8"), text.ctypes.data if len(text) else None,
                               len(text), ctypes.byref(h)))
    try:
        n = lib().mbpe_split_count(h)
        p = lib().mbpe_split_offsets(h)
        if not p:
            raise MbpeError(ERR_SPLIT_GAP, lib().mbpe_last_error().decode("utf-8", "replace"))
        return np.ctypeslib.as_array(p, shape=(n + 1,)).copy()
    finally:
        lib().mbpe_split_free(h)


def encode_chunks(data, chunk_off, merges, device=0):
    """internal_encode on the device (mbpe_encode_chunks) -> (uint32 tokens, passes)."""
    text = _u8(data)
    off = None if chunk_off is None else np.ascontiguousarray(chunk_off, dtype=np.uint64)
    m = np.ascontiguousarray(merges, dtype=np.uint32).reshape(-1, 2)
    n, passes = ctypes.c_uint64(), ctypes.c_uint32()
    out = np.zeros(max(len(text), 1), dtype=np.uint32)
    _check(lib().mbpe_encode_chunks(device, text.ctypes.data if len(text) else None, len(text),
                                    None if off is None else off.ctypes.data, 0 if off is None else len(off) - 1,
                                    m.ctypes.data if len(m) else None, len(m), out.ctypes.data, len(out),
                                    ctypes.byref(n), ctypes.byref(passes)))
    return out[:n.value].copy(), passes.value


class Trainer:
    """One mbpe_ctx.  Mirrors the order of Tokenizer::train (Tokenizer.h:489-598)."""

    def __init__(self, device=0):
        self._h = ctypes.c_void_p()
        _check(lib().mbpe_create(device, ctypes.byref(self._h)))
        self._keep = None
        self.vocab_size = 0

    def close(self):
        if self._h:
            lib().mbpe_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_option(self, name, value):
        _check(lib().mbpe_set_option(self._h, name.encode(), int(value)))

    def load_corpus(self, data, chunk_off=None):
        text = _u8(data)
        off = None if chunk_off is None else np.ascontiguousarray(chunk_off, dtype=np.uint64)
        self._keep = (text, off)
        _check(lib().mbpe_load_corpus(self._h, text.ctypes.data if len(text) else None, len(text),
                                      None if off is None else off.ctypes.data,
                                      0 if off is None else len(off) - 1, 0))

    def load_corpus_ranges(self, data, starts, ends):
        text = _u8(data)
        st = np.ascontiguousarray(starts, dtype=np.uint64)
        en = np.ascontiguousarray(ends, dtype=np.uint64)
        self._keep = (text, st, en)
        _check(lib().mbpe_load_corpus_ranges(self._h, text.ctypes.data if len(text) else None, len(text),
                                             st.ctypes.data if len(st) else None, en.ctypes.data if len(en) else None,
                                             len(st), 0))

    def load_corpus_device(self, dev_ptr, n_bytes, chunk_off=None, keep=None):
        """dev_ptr: device address of n_bytes corpus bytes (e.g. torch tensor .data_ptr())."""
        off = None if chunk_off is None else np.ascontiguousarray(chunk_off, dtype=np.uint64)
        self._keep = (keep, off)
        _check(lib().mbpe_load_corpus(self._h, ctypes.c_void_p(dev_ptr), n_bytes,
                                      None if off is None else off.ctypes.data,
                                      0 if off is None else len(off) - 1, 1))

    def pair_count_u8(self, want_table=True):
        table = np.zeros(65536, dtype=np.uint32) if want_table else None
        _check(lib().mbpe_pair_count_u8(self._h, None if table is None else table.ctypes.data))
        return table

    def train_begin(self, vocab_size):
        """Returns NEED_EXCHANGE in external-transport mode, else OK."""
        self.vocab_size = vocab_size
        return _check(lib().mbpe_train_begin(self._h, vocab_size))

    def train_steps(self, n_steps):
        """Returns the number of merges made (external-transport mode: NEED_EXCHANGE / OK code)."""
        done = ctypes.c_uint32()
        rc = _check(lib().mbpe_train_steps(self._h, n_steps, ctypes.byref(done)))
        return rc if self._external else done.value

    def train_sequences(self, n_sequences):
        """Runs up to n_sequences batch sequences (see mbpe_train_sequences); returns the merges they committed."""
        done = ctypes.c_uint32()
        _check(lib().mbpe_train_sequences(self._h, n_sequences, ctypes.byref(done)))
        return done.value

    _external = False

    def comm_init_external(self, rank, n_ranks):
        _check(lib().mbpe_comm_init_external(self._h, rank, n_ranks))
        self._external = n_ranks > 1

    def exchange_buffer(self):
        """(device pointer, number of u32) of the buffer to sum-all-reduce across ranks."""
        p = ctypes.c_void_p()
        n = ctypes.c_uint64()
        _check(lib().mbpe_comm_exchange_buffer(self._h, ctypes.byref(p), ctypes.byref(n)))
        return p.value, n.value

    def exchange_done(self):
        return _check(lib().mbpe_comm_exchange_done(self._h))

    def train_result(self):
        cap = max(self.vocab_size - 256, 1)
        merges = np.zeros((cap, 2), dtype=np.uint32)
        counts = np.zeros(cap, dtype=np.int32)
        n = ctypes.c_uint32()
        _check(lib().mbpe_train_result(self._h, merges.ctypes.data, counts.ctypes.data, cap, ctypes.byref(n)))
        return merges[:n.value].copy(), counts[:n.value].copy()

    def train_lexical(self, data, vocab_size, chunk_off=None):
        return self.train(data, vocab_size, chunk_off, conflict_resolution=1)

    def train(self, data, vocab_size, chunk_off=None, conflict_resolution=1):
        """conflict_resolution: 1 = lexical, 0 = first (mbpe_train)."""
        text = _u8(data)
        off = None if chunk_off is None else np.ascontiguousarray(chunk_off, dtype=np.uint64)
        cap = max(vocab_size - 256, 1)
        merges = np.zeros((cap, 2), dtype=np.uint32)
        counts = np.zeros(cap, dtype=np.int32)
        n = ctypes.c_uint32()
        st = Stats()
        _check(lib().mbpe_train(self._h, text.ctypes.data if len(text) else None, len(text),
                                None if off is None else off.ctypes.data,
                                0 if off is None else len(off) - 1, vocab_size, conflict_resolution,
                                merges.ctypes.data, counts.ctypes.data, ctypes.byref(n),
                                ctypes.byref(st)))
        self.vocab_size = vocab_size
        return merges[:n.value].copy(), counts[:n.value].copy(), st.as_dict()

    def stats(self):
        st = Stats()
        _check(lib().mbpe_get_stats(self._h, ctypes.byref(st)))
        return st.as_dict()

    def stream(self):
        n = ctypes.c_uint64()
        _check(lib().mbpe_get_stream(self._h, None, None, 0, ctypes.byref(n)))
        toks = np.zeros(max(n.value, 1), dtype=np.uint32)
        ends = np.zeros(max(n.value, 1), dtype=np.uint8)
        _check(lib().mbpe_get_stream(self._h, toks.ctypes.data, ends.ctypes.data, n.value, ctypes.byref(n)))
        return toks[:n.value], ends[:n.value]

    def stream_device(self):
        """(device pointer, n_slots, slot_bits, end_bit, barrier or None) of the live slot stream (see
        mbpe_stream_device)."""
        p, n = ctypes.c_void_p(), ctypes.c_uint64()
        bits, end, bar = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32()
        _check(lib().mbpe_stream_device(self._h, ctypes.byref(p), ctypes.byref(n), ctypes.byref(bits), ctypes.byref(end),
                                        ctypes.byref(bar)))
        return p.value, n.value, bits.value, end.value, (None if bar.value == 0xFFFFFFFF else bar.value)

    def table_device(self):
        """(device pointer, vshift) of the dense pair table (see mbpe_table_device)."""
        p, v = ctypes.c_void_p(), ctypes.c_uint32()
        _check(lib().mbpe_table_device(self._h, ctypes.byref(p), ctypes.byref(v)))
        return p.value, v.value

    def pairs(self):
        n = ctypes.c_uint64()
        _check(lib().mbpe_get_pairs(self._h, None, None, None, 0, ctypes.byref(n)))
        a = np.zeros(max(n.value, 1), dtype=np.uint32)
        b = np.zeros(max(n.value, 1), dtype=np.uint32)
        c = np.zeros(max(n.value, 1), dtype=np.int32)
        _check(lib().mbpe_get_pairs(self._h, a.ctypes.data, b.ctypes.data, c.ctypes.data, n.value, ctypes.byref(n)))
        return a[:n.value], b[:n.value], c[:n.value]

    def pairs_dict(self):
        a, b, c = self.pairs()
        return {(int(x), int(y)): int(z) for x, y, z in zip(a, b, c)}

    def compact(self):
        _check(lib().mbpe_compact(self._h))

    def comm_init(self, uid, rank, n_ranks):
        buf = (ctypes.c_uint8 * COMM_ID_BYTES).from_buffer_copy(bytes(uid))
        _check(lib().mbpe_comm_init(self._h, buf, rank, n_ranks))


def comm_unique_id():
    buf = (ctypes.c_uint8 * COMM_ID_BYTES)()
    _check(lib().mbpe_comm_unique_id(buf))
    return bytes(buf)


class Tokenizer:
    """Binding of include/mbpe_tokenizer.h: the host-side mirror of the reference Tokenizer."""

    FIRST, LEXICAL = 0, 1

    def __init__(self, pattern=""):
        self._h = ctypes.c_void_p()
        _check(lib().mbpe_tok_create(pattern.encode("utf-8"), ctypes.byref(self._h)))

    def close(self):
        if self._h:
            lib().mbpe_tok_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_special_tokens_from_file(self, text):
        b = text if isinstance(text, bytes) else text.encode("utf-8")
        _check(lib().mbpe_tok_set_special_tokens(self._h, b, len(b)))

    def train(self, data, vocab_size, conflict_resolution=1, verbose=False, device=0):
        text = _u8(data)
        _check(lib().mbpe_tok_train(self._h, text.ctypes.data if len(text) else None, len(text), vocab_size,
                                    conflict_resolution, int(verbose), device))

    def set_merges(self, merges):
        m = np.ascontiguousarray(merges, dtype=np.uint32).reshape(-1, 2)
        _check(lib().mbpe_tok_set_merges(self._h, m.ctypes.data if len(m) else None, len(m)))

    def merges(self):
        n = ctypes.c_uint32()
        _check(lib().mbpe_tok_get_merges(self._h, None, 0, ctypes.byref(n)))
        m = np.zeros((max(n.value, 1), 2), dtype=np.uint32)
        _check(lib().mbpe_tok_get_merges(self._h, m.ctypes.data, n.value, ctypes.byref(n)))
        return m[:n.value]

    def save(self, path, write_vocab=False):
        _check(lib().mbpe_tok_save(self._h, os.fsencode(path), int(write_vocab)))

    def load(self, path, verbose=False):
        _check(lib().mbpe_tok_load(self._h, os.fsencode(path), int(verbose)))

    def encode(self, data, device=None):
        """device None: internal_encode on the host; an int: on that HIP device (mbpe_tok_encode_device)."""
        text = _u8(data)
        n = ctypes.c_uint64()
        out = np.zeros(max(len(text), 1), dtype=np.uint32)
        if device is None:
            _check(lib().mbpe_tok_encode(self._h, text.ctypes.data if len(text) else None, len(text), 0,
                                         out.ctypes.data, len(out), ctypes.byref(n)))
        else:
            _check(lib().mbpe_tok_encode_device(self._h, text.ctypes.data if len(text) else None, len(text), 0, device,
                                                out.ctypes.data, len(out), ctypes.byref(n)))
        return out[:n.value].copy()

    def decode(self, tokens):
        t = np.ascontiguousarray(tokens, dtype=np.uint32)
        n = ctypes.c_uint64()
        _check(lib().mbpe_tok_decode(self._h, t.ctypes.data if len(t) else None, len(t), 0, None, 0, ctypes.byref(n)))
        out = np.zeros(max(n.value, 1), dtype=np.uint8)
        _check(lib().mbpe_tok_decode(self._h, t.ctypes.data if len(t) else None, len(t), 0, out.ctypes.data,
                                     len(out), ctypes.byref(n)))
        return out[:n.value].tobytes()

"""Size-independent checks of a training run, done on the device with torch (plumbing only: the
product never imports this; bench.py and tests/ do).

  * decode round trip: expanding the live token stream through the merges gives back the corpus,
    byte for byte (nothing lost, duplicated or reordered by thousands of in-place stream passes),
  * the chosen counts never increase: a pair created by a merge occurs at most as often as the pair
    it came from, every other pair can only lose occurrences (reference loop Tokenizer.h:557-589),
  * recount: the adjacent pairs of the final stream, counted from scratch, equal the pair table
    that was maintained incrementally (merge_incremental, Tokenizer.h:239-280).
"""
import numpy as np


def token_tables(merges, max_len=None):
    """Byte expansion of every token: lens[v] and bytes_[v, :lens[v]] (vocab[256+k] = vocab[a] ++ vocab[b],
    reference Tokenizer.h:562-564)."""
    n = 256 + len(merges)
    exp = [bytes([i]) for i in range(256)]
    for a, b in merges:
        exp.append(exp[int(a)] + exp[int(b)])
    lens = np.fromiter((len(e) for e in exp), dtype=np.int64, count=n)
    width = int(lens.max()) if max_len is None else max_len
    tab = np.zeros((n, width), dtype=np.uint8)
    for i, e in enumerate(exp):
        tab[i, :len(e)] = np.frombuffer(e, dtype=np.uint8)
    return lens, tab


def live_tokens(tr, torch, device, lo=0, hi=None):
    """Token ids (int32) and chunk-end flags (bool) of the live slots in [lo, hi) of the device stream.
    (Barrier layout: a token is the last of its chunk when a barrier slot follows it; a barrier that is the
    first live slot of the range belongs to a token before the range.)"""
    ptr, n_slots, bits, end_bit, barrier = tr.stream_device()
    hi = n_slots if hi is None else min(hi, n_slots)
    if bits == 16:
        raw = _as_tensor(torch, ptr + 2 * lo, hi - lo, torch.int16, device).to(torch.int32) & 0xFFFF
        hole = 0xFFFF
    else:
        raw = _as_tensor(torch, ptr + 4 * lo, hi - lo, torch.int32, device)
        hole = -1
    raw = raw[raw != hole]
    if barrier is not None:
        is_bar = raw == barrier
        ends = torch.zeros_like(is_bar)
        ends[:-1] = is_bar[1:]
        keep = ~is_bar
        return raw[keep], ends[keep]
    if end_bit and bits == 32:          # (the 32-bit continuation: bit 31 = last token of its chunk)
        return raw & 0x7FFFFFFF, raw < 0
    if end_bit:
        return raw & ~end_bit, (raw & end_bit) != 0
    return raw, None


class _DevArray:
    """__cuda_array_interface__ view of raw device memory (so torch can wrap it without a copy)."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


def _as_tensor(torch, ptr, n, dtype, device):
    typestr = {torch.int16: "<i2", torch.int32: "<i4", torch.uint8: "|u1"}[dtype]
    if n == 0:
        return torch.empty(0, dtype=dtype, device=device)
    return torch.as_tensor(_DevArray(ptr, n, typestr), device=device)


def decode_roundtrip(tr, merges, corpus, torch, device, slots_per_piece=1 << 28):
    """True iff decode(live stream) == corpus.  corpus: uint8 tensor on `device` (this rank's shard)."""
    lens_h, tab_h = token_tables(merges)
    lens_t = torch.from_numpy(lens_h).to(device)
    tab_t = torch.from_numpy(tab_h).to(device)
    width = tab_h.shape[1]
    n_slots = tr.stream_device()[1]
    base = 0
    n_live = 0
    ok = True
    for lo in range(0, n_slots, slots_per_piece):
        toks, _ = live_tokens(tr, torch, device, lo, lo + slots_per_piece)
        if toks.numel() == 0:
            continue
        n_live += int(toks.numel())
        if int(toks.max()) >= len(lens_h):
            return {"ok": False, "why": "token id beyond the vocabulary", "n_live": n_live}
        toks = toks.long()
        ln = lens_t[toks]
        ends = torch.cumsum(ln, 0)
        starts = ends - ln + base
        total = base + int(ends[-1])
        if total > corpus.numel():
            return {"ok": False, "why": "decoded stream longer than the corpus", "n_live": n_live}
        for j in range(width):
            if j == 0:
                sel_t, sel_s = toks, starts
            else:
                m = ln > j
                if not bool(m.any()):
                    break
                sel_t, sel_s = toks[m], starts[m]
            if not bool((corpus[sel_s + j] == tab_t[sel_t, j]).all()):
                ok = False
                break
        base = total
        del toks, ln, ends, starts
        if not ok:
            break
    if ok and base != corpus.numel():
        return {"ok": False, "why": "decoded %d bytes, corpus has %d" % (base, corpus.numel()), "n_live": n_live}
    return {"ok": ok, "why": "" if ok else "decoded bytes differ from the corpus", "n_live": n_live,
            "decoded_bytes": base, "max_token_bytes": width}


def decoded_length(tr, merges, torch, device, slots_per_piece=1 << 28):
    """Bytes the live stream of this context decodes to (a shard of a multi-GPU run owns the bytes of its tokens:
    a merge that straddles two shards leaves the new token with the left one, so a shard's stream need not
    cover exactly the byte range it was loaded with)."""
    lens_h, _ = token_tables(merges)
    lens_t = torch.from_numpy(lens_h).to(device)
    n_slots = tr.stream_device()[1]
    total = 0
    for lo in range(0, n_slots, slots_per_piece):
        toks, _ = live_tokens(tr, torch, device, lo, lo + slots_per_piece)
        if toks.numel():
            total += int(lens_t[toks.long()].sum())
    return total


def tiles_in_prefix_form(tr, torch, device, slots_per_piece=1 << 28):
    """Every 512-slot tile of the stream holds its live tokens in its first slots and its holes behind them: what the
    stream kernels keep whenever they rewrite a tile (tile_compact in csrc/kernels.hip) and the fused pass relies on."""
    ptr, n_slots, bits, end_bit, barrier = tr.stream_device()
    if bits == 32:
        return True          # (the 32-bit continuation compacts its stream at every merge: no holes, no tiles)
    assert bits == 16 and n_slots % 512 == 0
    for lo in range(0, n_slots, slots_per_piece):
        hi = min(lo + slots_per_piece, n_slots)
        hole = (_as_tensor(torch, ptr + 2 * lo, hi - lo, torch.int16, device) == -1).view(-1, 512)
        if bool((hole[:, :-1] & ~hole[:, 1:]).any()):       # a hole directly before a live slot
            return False
    return True


def counts_nonincreasing(counts):
    c = np.asarray(counts, dtype=np.int64)
    return bool(np.all(np.diff(c) <= 0))


def recount_pairs(tr, torch, device):
    """{(a, b): count} of the adjacent live pairs of the device stream, counted from scratch (pairs
    do not start at the last token of a chunk: Tokenizer.h:135-144)."""
    toks, ends = live_tokens(tr, torch, device)
    if toks.numel() < 2:
        return {}
    key = toks[:-1].long() * (1 << 32) + toks[1:].long()
    if ends is not None:
        key = key[~ends[:-1]]
    u, c = torch.unique(key, return_counts=True)
    u, c = u.cpu().numpy(), c.cpu().numpy()
    return {(int(k >> 32), int(k & 0xFFFFFFFF)): int(v) for k, v in zip(u, c)}


def dense_table_matches_recount(tr, torch, device):
    """The incrementally maintained dense pair table against a recount of the live stream, on the device:
    every pair of the stream has exactly its count in its cell, and no other cell holds a non-zero count."""
    toks, ends = live_tokens(tr, torch, device)
    ptr, vshift = tr.table_device()
    cells = _as_tensor(torch, ptr, 1 << (2 * vshift), torch.int32, device)
    a, b = toks[:-1].long(), toks[1:].long()
    if ends is not None:
        keep = ~ends[:-1]
        a, b = a[keep], b[keep]
    idx = ((((a >> 5) << (vshift - 5)) | (b >> 5)) << 10) | ((a & 31) << 5) | (b & 31)
    del a, b
    u, c = torch.unique(idx, return_counts=True)
    del idx
    got = cells[u].long() & 0x7FFFFFFF
    present = bool(((cells[u].long() >> 31) & 1).all())
    same = bool((got == c).all())
    nonzero = 0
    step = 1 << 28
    for lo in range(0, cells.numel(), step):
        nonzero += int(((cells[lo:lo + step] & 0x7FFFFFFF) != 0).sum())
    return {"ok": present and same and nonzero == int(u.numel()), "pairs_in_stream": int(u.numel()),
            "nonzero_cells": nonzero, "counts_equal": same, "all_present": present}


def recount_cells(tr, torch, device, vshift, slots_per_piece=1 << 27):
    """The adjacent pairs of the live stream counted from scratch into a dense array laid out like the
    library's dense pair table (tiles of 32 x 32 cells), piece by piece so that a 4 GiB stream needs no
    more than a few GB of scratch.  int32[1 << 2 * vshift]."""
    _, n_slots, _, _, _ = tr.stream_device()
    hist = torch.zeros(1 << (2 * vshift), dtype=torch.int32, device=device)
    prev_tok, prev_end = None, False
    for lo in range(0, n_slots, slots_per_piece):
        toks, ends = live_tokens(tr, torch, device, lo, lo + slots_per_piece)
        if toks.numel() == 0:
            continue
        toks = toks.long()
        if prev_tok is not None:          # the pair that straddles the pieces
            toks = torch.cat([prev_tok.view(1), toks])
            if ends is not None:
                ends = torch.cat([torch.tensor([prev_end], dtype=torch.bool, device=device), ends])
        a, b = toks[:-1], toks[1:]
        if ends is not None:
            keep = ~ends[:-1]
            a, b = a[keep], b[keep]
        idx = ((((a >> 5) << (vshift - 5)) | (b >> 5)) << 10) | ((a & 31) << 5) | (b & 31)
        del a, b
        hist += torch.bincount(idx, minlength=hist.numel()).to(torch.int32)
        del idx
        prev_tok = toks[-1].clone()
        prev_end = bool(ends[-1]) if ends is not None else False
        del toks
    return hist


def argmax_at_checkpoint(tr, torch, device):
    """At a host synchronisation point of a training run with the dense pair table: (i) the incrementally
    maintained table equals a recount of the live stream, cell by cell; (ii) the merge the trainer commits
    next is the argmax of the RECOUNTED table under (count descending, first ascending, second ascending) --
    CompareLexicalOrder, PairCount.h:194-207, get_top_pair_count :262-269.  Runs one merge (train_steps(1))."""
    ptr, vshift = tr.table_device()
    cells = _as_tensor(torch, ptr, 1 << (2 * vshift), torch.int32, device)
    hist = recount_cells(tr, torch, device, vshift)
    table_ok = bool(((cells & 0x7FFFFFFF) == hist).all())
    cmax = int(hist.max())
    e = torch.nonzero(hist == cmax).view(-1)                 # cells holding the maximum
    tile, w = e >> 10, e & 1023
    first = ((tile >> (vshift - 5)) << 5) | (w >> 5)
    second = ((tile & ((1 << (vshift - 5)) - 1)) << 5) | (w & 31)
    key = (first << 16) | second
    k = int(key.min())
    del hist, e, tile, w, first, second, key
    n_before = len(tr.train_result()[0])
    tr.train_steps(1)
    merges, counts = tr.train_result()
    got = (int(merges[n_before][0]), int(merges[n_before][1]), int(counts[n_before])) if len(merges) > n_before else None
    want = (k >> 16, k & 0xFFFF, cmax)
    return {"merge": n_before, "table_equals_recount": table_ok, "argmax_of_recount": list(want),
            "committed": list(got) if got else None, "ok": bool(table_ok and got == want and cmax > 0)}

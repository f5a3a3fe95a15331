"""Executable specification of the sharded (multi-GPU) training protocol.

A numpy/python model of what every rank does in the HIP path -- local pair
count, rank edges, boundary pairs, per-merge decisions from neighbours two
deep, the (m, adj, L, R) count deltas, the exchange buffer layout and the
replicated pair table -- with the collective supplied by the caller
(torch.distributed all_reduce over gloo in tests/test_dist_cpu.py).  It runs
on CPU with world_size > 1 and must reproduce the single-rank oracle.

It mirrors minbpe-cc_amd/csrc/kernels.hip (k_merge / merge_tile_full,
k_rank_edge, k_compose_edges, k_boundary_pairs, k_apply); it is test
infrastructure, not a fallback: nothing in the product imports it.
"""
import numpy as np

HOLE = 0xFFFF
ENDBIT = 0x8000


def header_words(n_ranks):
    return (2 + 8 * n_ranks + 3) // 4 * 4


class Shard:
    def __init__(self, data, chunk_off, rank, world, allreduce):
        """data: this rank's bytes; chunk_off: offsets relative to the shard or None."""
        self.rank, self.world, self.allreduce = rank, world, allreduce
        self.chunked = chunk_off is not None
        self.endbit = ENDBIT if self.chunked else 0
        self.idmask = 0x7FFF if self.chunked else 0xFFFF
        toks = [int(b) for b in bytes(data)]
        if self.chunked:
            for e in chunk_off[1:]:
                if e > 0:
                    toks[int(e) - 1] |= ENDBIT
        self.toks = toks
        self.table = {}
        self.k = 0
        self.merges, self.counts = [], []
        self.hdr = header_words(world)

    # -- k_rank_edge -------------------------------------------------------
    def edge(self):
        t = self.toks
        e = [HOLE, HOLE, HOLE, HOLE, len(t) & 0xFFFFFFFF, len(t) >> 32, 0, 0]
        if t:
            e[0] = t[0]
            e[3] = t[-1]
            if len(t) >= 2:
                e[1] = t[1]
                e[2] = t[-2]
            run = 0
            for v in reversed(t):
                if v != t[-1]:
                    break
                run += 1
            e[6], e[7] = run & 0xFFFFFFFF, run >> 32
        return e   # head0, head1, tail1, tail0, n_live lo/hi, tail_run lo/hi

    # -- k_compose_edges ---------------------------------------------------
    def compose(self, edges):
        l_t0 = l_t1 = HOLE
        need = 2
        for j in range(self.rank - 1, -1, -1):
            e = edges[j]
            nl = 0 if e[3] == HOLE else (1 if e[2] == HOLE else 2)
            if nl == 0 or need == 0:
                continue
            if need == 2:
                l_t0 = e[3]
                need = 1
                if nl >= 2:
                    l_t1 = e[2]
                    need = 0
            else:
                l_t1 = e[3]
                need = 0
        run = 0
        for j in range(self.rank - 1, -1, -1):
            e = edges[j]
            live = e[4] | (e[5] << 32)
            if not live:
                continue
            if e[3] != l_t0:
                break
            tr = e[6] | (e[7] << 32)
            run += tr
            if tr != live:
                break
        r_h0 = r_h1 = HOLE
        need = 2
        for j in range(self.rank + 1, self.world):
            e = edges[j]
            nl = 0 if e[0] == HOLE else (1 if e[1] == HOLE else 2)
            if nl == 0 or need == 0:
                continue
            if need == 2:
                r_h0 = e[0]
                need = 1
                if nl >= 2:
                    r_h1 = e[1]
                    need = 0
            else:
                r_h1 = e[0]
                need = 0
        self.left = (l_t0, l_t1, run)
        self.right = (r_h0, r_h1)

    # -- begin: pair count, exchange, boundary pairs, table ------------------
    def begin(self):
        bp = np.zeros(65536 + self.hdr, dtype=np.int64)
        t = self.toks
        for i in range(len(t) - 1):
            if not (t[i] & self.endbit):
                bp[((t[i] & 0xFF) << 8) | (t[i + 1] & 0xFF)] += 1
        base = 65536 + 2 + 8 * self.rank
        bp[base:base + 8] = self.edge()
        bp = self.allreduce(bp)
        edges = [[int(v) for v in bp[65536 + 2 + 8 * r:65536 + 10 + 8 * r]] for r in range(self.world)]
        prev_tail = HOLE                                   # k_boundary_pairs
        for e in edges:
            if e[0] == HOLE:
                continue
            if prev_tail != HOLE and not (prev_tail & self.endbit):
                bp[((prev_tail & 0xFF) << 8) | (e[0] & 0xFF)] += 1
            prev_tail = e[3]
        for idx in np.nonzero(bp[:65536])[0]:
            self.table[((int(idx) >> 8) << 16) | (int(idx) & 0xFF)] = int(bp[idx])
        self.compose(edges)

    def argmax(self):
        best = None
        for key, c in self.table.items():
            cand = (c, -key)
            if best is None or cand > best:
                best = cand
        return None if best is None else (-best[1] >> 16, -best[1] & 0xFFFF, best[0])

    # -- one merge: merge_tile_full + exchange + k_apply ---------------------
    def merge(self, a, b, X):
        t = self.toks
        n = len(t)
        idm, endbit = self.idmask, self.endbit
        same = a == b
        l_t0, l_t1, l_run = self.left
        r_h0, r_h1 = self.right

        def get(i):      # old token at shard position i, or the neighbour ranks' edge tokens
            if 0 <= i < n:
                return t[i]
            return {-1: l_t0, -2: l_t1, n: r_h0, n + 1: r_h1}[i]
        xb = np.zeros(self.hdr + 2 * X, dtype=np.int64)
        m = adj = 0
        new = []
        run = l_run if (same and l_t0 == a) else 0       # raw-a tokens right before token 0
        for i in range(n):
            self_ = t[i]
            p1, p2, n1, n2 = get(i - 1), get(i - 2), get(i + 1), get(i + 2)
            if not same:
                amatch = self_ == a and (n1 & idm) == b and n1 != HOLE
                bmatch = (self_ & idm) == b and p1 == a
                prev_adj = p1 == b and p2 == a
            else:
                odd = run & 1
                amatch = self_ == a and not odd and (n1 & idm) == a and n1 != HOLE
                bmatch = (self_ & idm) == a and odd
                prev_adj = (not odd) and run >= 2
                run = run + 1 if self_ == a else 0
            if amatch:
                new.append(X | (n1 & endbit))
                m += 1
                if p1 != HOLE and not (p1 & endbit):
                    if prev_adj:
                        adj += 1
                    else:
                        xb[self.hdr + 2 * p1] += 1
            elif bmatch:
                if not (self_ & endbit) and n1 != HOLE:
                    next_adj = n1 == a and (n2 & idm) == b and n2 != HOLE
                    if not next_adj:
                        xb[self.hdr + 2 * (n1 & idm) + 1] += 1
            else:
                new.append(self_)
        self.toks = new
        xb[0], xb[1] = m, adj
        base = 2 + 8 * self.rank
        xb[base:base + 8] = self.edge()
        xb = self.allreduce(xb)
        gm, gadj = int(xb[0]), int(xb[1])
        tab = self.table

        def add(key, d):
            tab[key] = tab.get(key, 0) + d
            assert tab[key] >= 0, "negative count"
        for x in range(X):                                 # k_apply
            l, r = int(xb[self.hdr + 2 * x]), int(xb[self.hdr + 2 * x + 1])
            if l:
                add((x << 16) | a, -l)
                add((x << 16) | X, l)
            if r:
                add((b << 16) | x, -r)
                add((X << 16) | x, r)
        if gm:
            add((a << 16) | b, -gm)
        if gadj:
            add((b << 16) | a, -gadj)
            add((X << 16) | X, gadj)
        edges = [[int(v) for v in xb[2 + 8 * r:10 + 8 * r]] for r in range(self.world)]
        self.compose(edges)

    def train(self, vocab_size):
        self.begin()
        for i in range(256, vocab_size):
            top = self.argmax()
            if top is None:
                break
            a, b, c = top
            self.merges.append((a, b))
            self.counts.append(c)
            if c > 0:
                self.merge(a, b, i)
        return self.merges, self.counts

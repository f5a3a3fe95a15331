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

    # ---- a BATCH SEQUENCE: what the product runs per stream pass (train.cpp seq_stage_a/b/c) -------------
    # select (identical on every rank: replicated table) -> one pass over the shard that counts the deltas of
    # every pair of the batch -> ONE sum all-reduce of [header][m_j, ADJ][LR rows L_j, R_j of lr_pitch(ids) cells]
    # = header + batch header + 2 * n * pitch words, exactly -> validate (longest prefix the one-at-a-time loop
    # would have chosen in this order) -> apply the prefix's deltas -> rewrite the shard -> edges all-reduce.
    BATCH_MAX = 16                  # the model's kBatchMax (row pitch of ADJ, size of the m_j block)

    @staticmethod
    def lr_pitch(ids):
        return (ids + 63) & ~63

    def select_batch(self, limit):
        """k_sel_pick without the pass-over of dependent candidates: maxima in argmax order while each is
        independent of the earlier members ((c,d) after (a,b): d != a and c != b), has a count, and is no (t,t)."""
        order = sorted(self.table.items(), key=lambda kv: (-kv[1], kv[0]))
        batch, firsts, seconds = [], set(), set()
        for key, c in order:
            a, b = key >> 16, key & 0xFFFF
            if c == 0 or a == b or b in firsts or a in seconds:
                break
            batch.append((a, b, c))
            firsts.add(a)
            seconds.add(b)
            if len(batch) == limit:
                break
        return batch

    def scan_batch(self, batch, n_merge):
        """The stream pass for the first n_merge pairs of the batch (k_fused_batch / k_scan_batch): returns the new
        shard and (m_j, ADJ, L, R) of THIS shard.  A match of j that directly follows a match of i counts in
        ADJ[i][j] instead of L_j / R_i."""
        t, n = self.toks, len(self.toks)
        idm, endbit = self.idmask, self.endbit
        l_t0, l_t1, _ = self.left
        r_h0, r_h1 = self.right
        index = {(a, b): j for j, (a, b, _) in enumerate(batch[:n_merge])}

        def get(i):
            if 0 <= i < n:
                return t[i]
            return {-1: l_t0, -2: l_t1, n: r_h0, n + 1: r_h1}[i]

        def match_at(i):            # index of the batch pair that starts at position i (halo positions too), else None
            s0, s1 = get(i), get(i + 1) if i + 1 <= n + 1 else HOLE
            if s0 == HOLE or s1 == HOLE or (s0 & endbit):
                return None
            return index.get((s0, s1 & idm))
        nb = len(batch)
        m = [0] * nb
        adj = {}
        L, R = {}, {}
        new = []
        i = 0
        # (a match whose first token is the left neighbour's last token is that rank's: skip its second token here)
        if n and match_at(-1) is not None:
            j = match_at(-1)
            y = get(1)
            if not (t[0] & endbit) and y != HOLE and match_at(1) is None:
                R[(j, y & idm)] = R.get((j, y & idm), 0) + 1
            i = 1
        while i < n:
            j = match_at(i)
            if j is None:
                new.append(t[i])
                i += 1
                continue
            X = 256 + self.k + j
            second = get(i + 1)
            new.append(X | (second & endbit))
            m[j] += 1
            x = get(i - 1)
            if x != HOLE and not (x & endbit):
                p = match_at(i - 2)
                if p is not None:                      # ... (a', b') (a, b): (b', a) -> (X', X)
                    adj[(p, j)] = adj.get((p, j), 0) + 1
                else:
                    L[(j, x)] = L.get((j, x), 0) + 1
            if i + 1 < n:                              # the second token is mine: its right neighbour
                y = get(i + 2)
                if not (second & endbit) and y != HOLE and match_at(i + 2) is None:
                    R[(j, y & idm)] = R.get((j, y & idm), 0) + 1
            i += 2
        return new, m, adj, L, R

    def sequence(self, limit):
        if not self.table:
            return 0
        batch = self.select_batch(limit)
        if len(batch) < 2:
            return -1                                   # a (t,t), zero-count or lone pair: the caller takes the one-pair path
        nb, ids, BM = len(batch), 256 + self.k, self.BATCH_MAX
        pitch = self.lr_pitch(ids)
        hdrb = BM + BM * BM
        words = self.hdr + hdrb + 2 * nb * pitch        # exchange_words(c, ids, n_pairs) in train.cpp
        _, m, adj, L, R = self.scan_batch(batch, nb)
        xb = np.zeros(words, dtype=np.int64)
        for j in range(nb):
            xb[self.hdr + j] = m[j]
        for (p, j), w in adj.items():
            xb[self.hdr + BM + p * BM + j] = w
        lr0 = self.hdr + hdrb
        for (j, x), w in L.items():
            xb[lr0 + (2 * j) * pitch + x] = w
        for (j, y), w in R.items():
            xb[lr0 + (2 * j + 1) * pitch + y] = w
        xb = self.allreduce(xb)
        assert len(xb) == words
        self.exchanged.append((nb, ids, words))
        gm = [int(xb[self.hdr + j]) for j in range(nb)]
        ADJ = xb[self.hdr + BM:self.hdr + BM + BM * BM].reshape(BM, BM)
        Lr = [xb[lr0 + (2 * j) * pitch:lr0 + (2 * j) * pitch + ids] for j in range(nb)]
        Rr = [xb[lr0 + (2 * j + 1) * pitch:lr0 + (2 * j + 1) * pitch + ids] for j in range(nb)]
        # ---- k_adj_sums / k_delta_max / k_adj_max / k_validate
        adj_in = [int(ADJ[:nb, j].sum()) for j in range(nb)]
        adj_out = [int(ADJ[j, :nb].sum()) for j in range(nb)]

        def packed(c, key):
            return (c, -key)
        maxp = [None] * nb
        for j in range(nb):
            Xj = ids + j
            best = None
            for x in np.nonzero(Lr[j])[0]:
                c = packed(int(Lr[j][x]) + adj_in[j], (int(x) << 16) | Xj)
                best = c if best is None or c > best else best
            for y in np.nonzero(Rr[j])[0]:
                c = packed(int(Rr[j][y]) + adj_out[j], (Xj << 16) | int(y))
                best = c if best is None or c > best else best
            maxp[j] = best
        for p in range(nb):
            for q in range(nb):
                w = int(ADJ[p, q])
                if not w:
                    continue
                def raise_(j, v):
                    if maxp[j] is None or v > maxp[j]:
                        maxp[j] = v
                raise_(max(p, q), packed(w, ((ids + p) << 16) | (ids + q)))
                raise_(q, packed(adj_in[q], (batch[p][1] << 16) | (ids + q)))
                raise_(p, packed(adj_out[p], ((ids + p) << 16) | batch[q][0]))
        commit = nb
        run = None
        for j in range(nb):
            key_j = (batch[j][0] << 16) | batch[j][1]
            if j >= 1 and run is not None and packed(batch[j][2], key_j) <= run:
                commit = j
                break
            if maxp[j] is not None and (run is None or maxp[j] > run):
                run = maxp[j]
        # a kept match that touches a dropped one keeps its plain neighbour
        if commit < nb:
            for r in range(nb):
                for q in range(nb):
                    w = int(ADJ[r, q])
                    if not w:
                        continue
                    if r >= commit and q < commit:
                        Lr[q][batch[r][1]] += w
                    elif r < commit and q >= commit:
                        Rr[r][batch[q][0]] += w
        # ---- k_apply_batch
        tab = self.table

        def add(key, d):
            tab[key] = tab.get(key, 0) + d
            assert tab[key] >= 0, "negative count"
        for j in range(commit):
            a, b, c = batch[j]
            X = ids + j
            for x in np.nonzero(Lr[j])[0]:
                w = int(Lr[j][x])
                add((int(x) << 16) | a, -w)
                add((int(x) << 16) | X, w)
            for y in np.nonzero(Rr[j])[0]:
                w = int(Rr[j][y])
                add((b << 16) | int(y), -w)
                add((X << 16) | int(y), w)
            assert gm[j] == c, "an (a,b), a != b, pair is merged wherever it occurs"
            add((a << 16) | b, -gm[j])
        for p in range(commit):
            for q in range(commit):
                w = int(ADJ[p, q])
                if w:
                    add((batch[p][1] << 16) | batch[q][0], -w)
                    add(((ids + p) << 16) | (ids + q), w)
        # ---- the rewrite (the fused pass's output when the whole batch is kept, else k_rewrite_marked for the prefix)
        self.toks = self.scan_batch(batch, commit)[0]
        for j in range(commit):
            self.merges.append((batch[j][0], batch[j][1]))
            self.counts.append(batch[j][2])
        self.k += commit
        # ---- the edges of the rewritten shards (second, small all-reduce)
        xe = np.zeros(self.hdr, dtype=np.int64)
        base = 2 + 8 * self.rank
        xe[base:base + 8] = self.edge()
        xe = self.allreduce(xe)
        self.compose([[int(v) for v in xe[2 + 8 * r:10 + 8 * r]] for r in range(self.world)])
        return commit

    def train_batched(self, vocab_size, limit=8):
        """The product's loop: batch sequences, with the one-pair path (self.merge) for (t,t) and zero-count pairs."""
        self.begin()
        self.exchanged = []
        target = vocab_size - 256
        while self.k < target:
            got = self.sequence(min(limit, target - self.k))
            if got == 0:
                break
            if got > 0:
                continue
            a, b, c = self.argmax()
            self.merges.append((a, b))
            self.counts.append(c)
            if c > 0:
                self.merge(a, b, 256 + self.k)
            self.k += 1
        return self.merges, self.counts

    # -- `first` tie-break on a sharded stream (k_first_gather / k_first_pos / k_first_publish / k_first_pick_global) --
    def argmax_first(self):
        """Among the pairs of maximal count the one whose first occurrence comes first in the whole corpus
        (PairCountInsertOrder on a table rebuilt per merge: PairCount.h:65-74, :141-166; Tokenizer.h:581-585).  Every rank
        finds the earliest tied pair of ITS shard -- the pair of its last token takes its second token from the right
        neighbour's edge -- and publishes (found, position, key) in its slot of a rank-indexed array; the sum over the
        ranks is an all-gather, and the hit of the lowest rank that has one wins: shards follow each other in rank order.
        A count of 0 means no pair is left: the reference's loop breaks (Tokenizer.h:586-588)."""
        m = max(self.table.values(), default=0)
        if m <= 0:
            return None
        ties = {k for k, c in self.table.items() if c == m}
        key = min(ties)
        if len(ties) > 1:                     # (the table is replicated: every rank sees the same ties)
            t, n, idm = self.toks, len(self.toks), self.idmask
            slot = [0, 0, 0]
            for i in range(n):
                if t[i] & self.endbit:
                    continue
                nx = t[i + 1] if i + 1 < n else self.right[0]
                if nx == HOLE:
                    continue
                kk = ((t[i] & idm) << 16) | (nx & idm)
                if kk in ties:
                    slot = [1, i, kk]
                    break
            buf = np.zeros(self.hdr, dtype=np.int64)
            buf[2 + 8 * self.rank:2 + 8 * self.rank + 3] = slot
            buf = self.allreduce(buf)
            for r in range(self.world):
                if buf[2 + 8 * r]:
                    key = int(buf[2 + 8 * r + 2])
                    break
        return key >> 16, key & 0xFFFF, m

    def train_first(self, vocab_size):
        self.begin()
        for i in range(256, vocab_size):
            top = self.argmax_first()
            if top is None:
                break
            a, b, c = top
            self.merges.append((a, b))
            self.counts.append(c)
            self.merge(a, b, i)
        return self.merges, self.counts

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

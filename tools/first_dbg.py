import sys, os
sys.path.insert(0, "minbpe-cc_amd/python"); sys.path.insert(0, "oracle"); sys.path.insert(0, "tests")
import numpy as np, mbpe, oracle as O
one = np.frombuffer(open("tests/golden/data/shakespeare.txt","rb").read(), dtype=np.uint8)
off1 = np.asarray(mbpe.presplit(O.GPT4_SPLIT_PATTERN, one), dtype=np.uint64)
vocab = 256 + 48
want_m, want_c = O.train(one, vocab, off1, mode=O.FIRST)
for copies in (1, 2, 3, 40, 1000):
    data = np.tile(one, copies)
    base = (np.arange(copies, dtype=np.uint64) * np.uint64(len(one)))[:, None]
    off = np.concatenate([(off1[None, :-1] + base).reshape(-1), np.array([len(data)], dtype=np.uint64)])
    with mbpe.Trainer(0) as tr:
        m, c, st = tr.train(data, vocab, off, conflict_resolution=0)
    ok = m.tolist() == want_m.tolist()
    bad = next((i for i in range(min(len(m), len(want_m))) if m[i].tolist() != want_m[i].tolist()), None)
    print(copies, ok, len(m), bad, None if bad is None else (m[bad].tolist(), int(c[bad]), want_m[bad].tolist(), int(want_c[bad])), flush=True)

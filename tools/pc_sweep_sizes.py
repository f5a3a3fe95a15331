"""Pair-count scan against numpy over corpus lengths around the block / iteration / vector boundaries of the scan kernels."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "minbpe-cc_amd", "python"))
import numpy as np
import mbpe
rng = np.random.default_rng(5)
base = rng.integers(97, 101, size=1 << 21, dtype=np.uint8)
bad = []
sizes = set()
for k in (1, 2, 3, 4, 8, 16, 17, 32, 33, 48, 64):
    for unit in (2048, 32768, 16384):
        for d in list(range(-20, 21)) + [2048 - 1, 2048 + 1, 4096 + 3]:
            n = k * unit + d
            if 2 <= n <= len(base):
                sizes.add(n)
with mbpe.Trainer(0) as tr:
    for n in sorted(sizes):
        data = base[:n]
        tr.load_corpus(data)
        got = tr.pair_count_u8()
        want = np.bincount((data[:-1].astype(np.uint32) << 8) | data[1:], minlength=65536).astype(np.uint32)
        if not np.array_equal(got, want):
            idx = np.nonzero(got != want)[0]
            bad.append((n, [(int(i) >> 8, int(i) & 255, int(got[i]), int(want[i])) for i in idx[:4]]))
# a few large ones: every workgroup busy, the last workgroups' ranges short or empty
big = rng.integers(97, 101, size=(96 << 20) + 64, dtype=np.uint8)
with mbpe.Trainer(0) as tr:
    for n in [(8 << 20) + d for d in (-17, -1, 0, 3, 4, 9, 15, 16, 31)] + [(8 << 20) + 32768 * 7 + 5, (24 << 20) + 11, (33 << 20) + 32768 + 4,
                                                                       (96 << 20) + 13, 256 * 32768 * 3 - 32768 + 6]:
        data = big[:n]
        tr.load_corpus(data)
        got = tr.pair_count_u8()
        want = np.bincount((data[:-1].astype(np.uint32) << 8) | data[1:], minlength=65536).astype(np.uint32)
        sizes.add(n)
        if not np.array_equal(got, want):
            idx = np.nonzero(got != want)[0]
            bad.append((n, [(int(i) >> 8, int(i) & 255, int(got[i]), int(want[i])) for i in idx[:4]]))
print("sizes", len(sizes), "bad", len(bad))
for b in bad[:40]:
    print(b)

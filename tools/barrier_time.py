"""Chunk ends as a slot bit against chunk ends as barrier slots: 71 MB of text under the
GPT-4 pattern, vocab 10,000 / 30,000 / 50,000."""
import sys, time
sys.path[:0] = ['tests', 'oracle', 'minbpe-cc_amd/python']      # run from the repo root
import numpy as np, mbpe, oracle as O
from conftest import read_data
base = read_data("shakespeare.txt")
data = np.frombuffer(base * 64, dtype=np.uint8)          # 71 MB of text
t = time.time(); off = mbpe.presplit(O.GPT4_SPLIT_PATTERN, data); print("split", time.time() - t, len(off))
tr = mbpe.Trainer(0)
for vocab, bar in ((10000, 0), (10000, 1), (10000, 0), (10000, 1), (30000, 0), (30000, 1), (50000, -1)):
    tr.set_option("chunk_barrier", bar)
    tr.load_corpus(data, off)
    t = time.time()
    tr.train_begin(vocab)
    done = tr.train_steps(vocab - 256)
    dt = time.time() - t
    st = tr.stats()
    m, c = tr.train_result()
    print("vocab", vocab, "barrier", bar, "s %.3f" % dt, "merges", done, "last count", int(c[-1]), "batches", st["n_batches"],
          "fused", st["n_fused"], "n_live", st["n_live"], "compactions", st["n_compactions"])

"""Launches of the pair-count scan through the C-ABI: 30 on the 4 GiB SplitMix64 corpus (clock ramp, box-to-box
spread) and 30 on 446 MB of text (shakespeare x 400: few distinct first bytes, what the bank hash is for).
MBPE_LIB selects a variant library built by tools/mkvar.sh.  Results: profiles/r0N_pair_count_ab.md."""
import os, sys, json
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "minbpe-cc_amd", "python"))
sys.path.insert(0, ROOT)
import numpy as np
import torch, mbpe
from bench import splitmix64_device
dev = torch.device("cuda", 0)
n = int(os.environ.get("PC_TIME_BYTES", str(4 << 30)))
keep, corpus = splitmix64_device(42, n, dev)
torch.cuda.synchronize()
tr = mbpe.Trainer(0)
tr.load_corpus_device(corpus.data_ptr(), n, keep=keep)
ms = []
reps = int(os.environ.get("PC_TIME_LAUNCHES", "30"))
for i in range(reps):
    tr.pair_count_u8(want_table=False)
    ms.append(round(tr.stats()["ms_pair_count_kernel"], 4))      # (the dispatch's own start / stop events)
tag = os.environ.get("MBPE_LIB", "default")
print(tag, "random 4 GiB", ms, "mean of launches 11-%d: %.4f" % (reps, sum(ms[10:]) / (reps - 10.0)), flush=True)
if os.environ.get("PC_TIME_TEXT", "1") == "0":
    sys.exit(0)
del keep, corpus
text = open(os.path.join(ROOT, "tests", "golden", "data", "shakespeare.txt"), "rb").read() * 400
t = torch.frombuffer(bytearray(text), dtype=torch.uint8).to(dev)
torch.cuda.synchronize()
tr.load_corpus_device(t.data_ptr(), len(text), keep=t)
ms = []
for i in range(30):
    tr.pair_count_u8(want_table=False)
    ms.append(round(tr.stats()["ms_pair_count_kernel"], 4))
print(tag, "text %d bytes" % len(text), ms, "mean of launches 11-30: %.4f" % (sum(ms[10:]) / 20.0), flush=True)

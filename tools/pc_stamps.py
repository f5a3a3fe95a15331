import os, sys
ROOT = os.getcwd()
sys.path.insert(0, os.path.join(ROOT, "minbpe-cc_amd", "python")); sys.path.insert(0, ROOT)
import torch, mbpe
from bench import splitmix64_device
dev = torch.device("cuda", 0)
n = 4 << 30
keep, corpus = splitmix64_device(42, n, dev)
torch.cuda.synchronize()
tr = mbpe.Trainer(0)
tr.load_corpus_device(corpus.data_ptr(), n, keep=keep)
for i in range(40):
    if i == 39: print("---- launch 40", file=sys.stderr, flush=True)
    tr.pair_count_u8(want_table=False)
print(tr.stats()["ms_pair_count_kernel"])

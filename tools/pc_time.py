"""30 launches of the pair-count scan through the C-ABI on the 4 GiB corpus (clock ramp, box-to-box spread);
MBPE_LIB selects a variant library built by tools/mkvar.sh.  Results: profiles/r02_pair_count_ab.md."""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "minbpe-cc_amd", "python"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, mbpe
from bench import splitmix64_device
dev = torch.device("cuda", 0)
n = 4 << 30
keep, corpus = splitmix64_device(42, n, dev)
torch.cuda.synchronize()
tr = mbpe.Trainer(0)
tr.load_corpus_device(corpus.data_ptr(), n, keep=keep)
ms = []
for i in range(30):
    tr.pair_count_u8(want_table=False)
    ms.append(round(tr.stats()["ms_pair_count"], 3))
print(os.environ.get("MBPE_LIB", "default"), ms)

"""Where a fused pass spends its time: the same batch run with timing-only instantiations of k_fused_batch
(library built with -DMBPE_DIAG: tools/mkvar.sh diag -DMBPE_DIAG; MBPE_LIB=build/libmbpe_diag.so).  Results: profiles/r0N_fused_diag.log.
diag 0 = the shipped kernel, 2 = everything but the count deltas, 3 / 5 = load + lookups + store, 4 = load + store."""
import os, sys, json
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "minbpe-cc_amd", "python"))
sys.path.insert(0, ROOT)
import torch, mbpe
from bench import splitmix64_device
dev = torch.device("cuda", 0)
n = 4 << 30
keep, corpus = splitmix64_device(42, n, dev)
torch.cuda.synchronize()
tr = mbpe.Trainer(0)
tr.load_corpus_device(corpus.data_ptr(), n, keep=keep)
tr.set_option("time_kernels", 1)
if os.environ.get("FUSED_DIAG_MAXBATCH"):
    tr.set_option("max_batch", int(os.environ["FUSED_DIAG_MAXBATCH"]))
for diag in [int(x) for x in os.environ.get("FUSED_DIAG_LIST", "0,2,3,5,4").split(",")]:
    os.environ.pop("MBPE_FUSED_DIAG", None)
    tr.train_begin(32000)
    tr.train_sequences(6)          # warm-up with the real kernel: a real batch is selected next
    s0 = tr.stats()
    os.environ["MBPE_FUSED_DIAG"] = str(diag)
    try:
        tr.train_sequences(4)
    except Exception as e:
        print("diag", diag, "error", e)
    s1 = tr.stats()
    nf = s1["fused_launches"] - s0["fused_launches"]
    print("diag", diag, "fused launches", nf, "avg ms", (s1["ms_fused_kernel"] - s0["ms_fused_kernel"]) / max(nf, 1),
          "matches per launch", (s0["n_live"] - s1["n_live"]) / max(nf, 1) if diag == 0 else "-",
          "merges per launch", (s1["n_merges"] - s0["n_merges"]) / max(nf, 1) if diag == 0 else "-", flush=True)

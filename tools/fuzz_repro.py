"""Re-runs one case of tests/test_gpu_fuzz.py::test_fuzz_against_oracle and, if it fails, looks for the options that matter.
    python3 tools/fuzz_repro.py SEED CASE"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest  # noqa: F401  (sets the import paths)
import numpy as np
import mbpe
import oracle as O
from conftest import read_data
import test_gpu_fuzz as F

seed, case = int(sys.argv[1]), int(sys.argv[2])
text = read_data("shakespeare.txt")
rng = np.random.default_rng(7000 + seed * 1000 + case)
data, off, vocab, opts = F._case(rng, text)
print("len", len(data), "vocab", vocab, "chunks", None if off is None else len(off) - 1, "opts", opts, flush=True)
want_m, want_c = O.train(data, vocab, off)


def run(o):
    with mbpe.Trainer(0) as tr:
        for k, v in {**F.DEFAULTS, **o}.items():
            tr.set_option(k, v)
        m, c, st = tr.train_lexical(data, vocab, off)
    bad = next((i for i in range(min(len(m), len(want_m))) if m[i].tolist() != want_m[i].tolist() or c[i] != want_c[i]), None)
    if bad is None and len(m) != len(want_m):
        bad = min(len(m), len(want_m))
    return bad, m, c, st


bad, m, c, st = run(opts)
print("first difference at merge", bad, flush=True)
if bad is not None:
    lo = max(bad - 3, 0)
    print("got ", m[lo:bad + 3].tolist(), c[lo:bad + 3].tolist())
    print("want", want_m[lo:bad + 3].tolist(), want_c[lo:bad + 3].tolist())
    print({k: st[k] for k in ("n_batches", "n_fused", "n_fused_dropped", "n_validation_drops", "n_skipped", "n_skip_cut", "n_compactions")})
    for k in opts:
        o = dict(opts)
        o[k] = F.DEFAULTS[k]
        b2 = run(o)[0]
        print("  with default %-18s (%s -> %s): first difference %s" % (k, opts[k], F.DEFAULTS[k], b2), flush=True)
    for rep in range(3):
        print("  again:", run(opts)[0])

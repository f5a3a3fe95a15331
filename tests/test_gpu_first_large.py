"""The `first` tie-break (the reference CLI's default: PairCountInsertOrder, PairCount.h:65-74, :141-166; recount per
merge, Tokenizer.h:581-585) at sizes the CPU oracle cannot reach, through properties that are exact:

  * 1.1 GB: shakespeare.txt x 1000 under the GPT-4 split, every copy its own chunks.  Pairs never cross chunks, so
    every count is exactly 1000 x the count in one copy and the first occurrences lie in the first copy in the same
    order: the merges must be the oracle's merges on ONE copy, pair for pair, with 1000 x its counts -- a text full of
    equal counts, so most merges go through the position tie-break.
  * 5 GiB: two pairs with the same, maximal count whose occurrences all lie BEYOND slot 2^32 (round 3 refused such a
    corpus: the tie-break packed a 32-bit slot position).  `first` must take the one that occurs first, lexical the
    smaller key."""
import numpy as np
import pytest

import mbpe
import oracle as O
from conftest import read_data

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def test_first_mode_text_x1000_equals_one_copy():
    one = np.frombuffer(read_data("shakespeare.txt"), dtype=np.uint8)
    copies = 1000
    off1 = np.asarray(mbpe.presplit(O.GPT4_SPLIT_PATTERN, one), dtype=np.uint64)
    vocab = 256 + 200            # (lexical and first order part ways at merge 147 on this text)
    want_m, want_c = O.train(one, vocab, off1, mode=O.FIRST)
    data = np.tile(one, copies)
    base = (np.arange(copies, dtype=np.uint64) * np.uint64(len(one)))[:, None]
    off = np.concatenate([(off1[None, :-1] + base).reshape(-1), np.array([len(data)], dtype=np.uint64)])
    with mbpe.Trainer(0) as tr:
        m, c, st = tr.train(data, vocab, off, conflict_resolution=0)
    assert m.tolist() == want_m.tolist()
    assert c.tolist() == [copies * int(x) for x in want_c]
    # (and the lexical order differs on this text: the position tie-break was really exercised)
    lex_m, _ = O.train(one, vocab, off1)
    assert lex_m.tolist() != want_m.tolist()


def test_first_mode_tie_beyond_slot_2_pow_32():
    dev = torch.device("cuda", 0)
    n = 5 << 30
    A, B, C, D = 65, 66, 67, 68
    buf = torch.empty(n + 16, dtype=torch.uint8, device=dev)
    step = 1 << 28
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    for lo in range(0, n, step):
        v = torch.randint(0, 252, (min(step, n - lo),), dtype=torch.int16, device=dev, generator=g)
        v = v + (v >= A).to(torch.int16) * 4                   # the filler never holds A, B, C or D
        buf[lo:lo + len(v)] = v.to(torch.uint8)
        del v
    # 2^19 occurrences each of "CD" and "AB", all in the last GiB; every "CD" 1 KiB before an "AB"
    tail = buf[4 << 30:n].view(-1, 2048)
    tail[:, 0], tail[:, 1] = C, D
    tail[:, 1024], tail[:, 1025] = A, B
    torch.cuda.synchronize()
    k = tail.shape[0]
    assert k == 1 << 19
    for mode, first_pair in ((0, [C, D]), (1, [A, B])):
        with mbpe.Trainer(0) as tr:
            tr.set_option("conflict_resolution", mode)
            tr.load_corpus_device(buf.data_ptr(), n, keep=buf)
            tr.train_begin(256 + 2)
            assert tr.train_steps(2) == 2
            m, c = tr.train_result()
        assert c.tolist() == [k, k], (mode, c.tolist())        # (a filler pair occurs ~85,000 times)
        assert m[0].tolist() == first_pair and sorted(m.tolist()) == [[A, B], [C, D]], (mode, m.tolist())

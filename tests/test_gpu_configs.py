"""GPU tests at the sizes BASELINE.json's configs name, through the C-ABI:

  config 3  bible stand-in (shakespeare.txt x 4), vocab 10,000: every merge and count
            against the oracle, decode round trip, recount of the final stream against the pair table
  config 4  SplitMix64 bytes to vocab 32,000 at 1 GiB: size-independent properties of the production
            layout (dense pair table with 4 GiB of cells, batches of 512, fused passes, compactions);
            and the first passes at 64 MiB with the same options against the oracle: merges, counts,
            stream and the whole pair table
  RCCL      the library's own communicator next to torch.distributed's (what bench.py does at N > 1)
"""
import os
import sys

import numpy as np
import pytest

import mbpe
import oracle as O
from mbpe import check as C
from conftest import ROOT, read_data

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def dev():
    torch.cuda.set_device(0)
    return torch.device("cuda", 0)


def test_config3_bible_standin_vocab_10000(dev):
    data = np.frombuffer(read_data("shakespeare.txt") * 4, dtype=np.uint8)
    assert len(data) == 4461576
    vocab = 10000
    want_m, want_c = O.train(data, vocab)
    with mbpe.Trainer(0) as tr:
        m, c, st = tr.train_lexical(data, vocab)
        assert m.tolist() == want_m.tolist() and c.tolist() == want_c.tolist()
        corpus = torch.from_numpy(data.copy()).to(dev)
        rt = C.decode_roundtrip(tr, m, corpus, torch, dev)
        assert rt["ok"], rt
        assert rt["n_live"] == st["n_live"]
        recount = C.recount_pairs(tr, torch, dev)
        table = {k: v for k, v in tr.pairs_dict().items() if v}
        assert recount == table
    assert C.counts_nonincreasing(c) and len(m) == vocab - 256
    assert st["n_batches"] < len(m) // 4          # merges did share stream passes


def test_config4_first_passes_against_oracle_64mib(dev):
    """Production options (dense table sized for vocab 32,000, batches of up to 512, fused pass): the first
    sequences on 64 MiB of the benchmark corpus, state compared with the oracle after them."""
    n = 64 << 20
    data = O.splitmix64_bytes(42, n)
    with mbpe.Trainer(0) as tr:
        tr.load_corpus(data)
        tr.train_begin(32000)
        done = 0
        while done < 256:
            got = tr.train_sequences(1)
            assert got > 0
            done += got
        m, c = tr.train_result()
        st = tr.stats()
        assert len(m) == done and st["n_fused"] >= 1 and st["n_batches"] < done // 8
        ost = O.State(data)
        want = []
        for i in range(done):
            top = ost.top()
            want.append(top)
            ost.merge(top[0], top[1], 256 + i)
        assert m.tolist() == [[a, b] for a, b, _ in want]
        assert c.tolist() == [cc for _, _, cc in want]
        assert np.array_equal(tr.stream()[0], ost.stream()[0])
        assert {k: v for k, v in tr.pairs_dict().items() if v} == {k: v for k, v in ost.table_dict().items() if v}
        ost.close()


def test_config4_full_vocab_properties_1gib(dev):
    """1 GiB of the benchmark corpus to vocab 32,000 (what bench.py's full_run does at 4 GiB): chosen counts
    never increase, decode(stream) == corpus, the incrementally maintained pair table equals a recount of the
    final stream."""
    from bench import splitmix64_device
    n = 1 << 30
    keep, corpus = splitmix64_device(42, n, dev)
    torch.cuda.synchronize()
    with mbpe.Trainer(0) as tr:
        tr.load_corpus_device(corpus.data_ptr(), n, keep=keep)
        tr.train_begin(32000)
        assert tr.train_steps(32000 - 256) == 32000 - 256
        m, c = tr.train_result()
        st = tr.stats()
        assert len(m) == 32000 - 256 and C.counts_nonincreasing(c)
        assert st["n_fused"] >= 10 and st["n_compactions"] >= 1
        rt = C.decode_roundtrip(tr, m, corpus, torch, dev)
        assert rt["ok"], rt
        assert rt["n_live"] == st["n_live"]
        tm = C.dense_table_matches_recount(tr, torch, dev)
        assert tm["ok"], tm
    # the first merges are the oracle's on the whole-corpus byte-pair table
    table = torch.zeros(65536, dtype=torch.int64, device=dev)
    b64 = corpus.long()
    table.scatter_add_(0, b64[:-1] * 256 + b64[1:], torch.ones(n - 1, dtype=torch.int64, device=dev))
    first = int(torch.argmax(table))                 # ties: smallest key, like the trainer
    assert (int(m[0][0]), int(m[0][1])) == (first >> 8, first & 0xFF) and int(c[0]) == int(table[first])


def test_config4_merge_order_at_checkpoints_1gib(dev):
    """The ORDER of merges deep into a full-size run, not only its end state: at checkpoints spread over a training
    of 1 GiB of the benchmark corpus to vocab 32,000 -- the last ones beyond merge 28,000, where batches are small,
    selections retry and compactions have happened -- the stream is recounted from scratch on the device; the
    incrementally maintained table must equal the recount cell by cell, and the merge the trainer commits next
    must be the argmax of the RECOUNTED table under (count desc, first asc, second asc): CompareLexicalOrder,
    PairCount.h:194-207; get_top_pair_count, :262-269."""
    from bench import splitmix64_device
    n = 1 << 30
    keep, corpus = splitmix64_device(42, n, dev)
    torch.cuda.synchronize()
    total = 32000 - 256
    with mbpe.Trainer(0) as tr:
        tr.load_corpus_device(corpus.data_ptr(), n, keep=keep)
        tr.train_begin(32000)
        seen = []
        for at in (0, 700, 5000, 16384, 24000, 28500, 30900, 31700):
            have = len(tr.train_result()[0])
            if at > have:
                assert tr.train_steps(at - have) == at - have
            r = C.argmax_at_checkpoint(tr, torch, dev)
            assert r["ok"], r
            seen.append(r["merge"])
        assert seen == [0, 700, 5000, 16384, 24000, 28500, 30900, 31700]
        tr.train_steps(total)
        m, c = tr.train_result()
        assert len(m) == total and C.counts_nonincreasing(c)


def _pair_table_torch(corpus):
    """count[(first << 8) | second] of a uint8 device tensor, piecewise."""
    n = corpus.numel()
    table = torch.zeros(65536, dtype=torch.int64, device=corpus.device)
    step = 1 << 28
    for lo in range(0, n - 1, step):
        hi = min(lo + step, n - 1)
        a = corpus[lo:hi].long()
        b = corpus[lo + 1:hi + 1].long()
        table += torch.bincount(a * 256 + b, minlength=65536)
        del a, b
    return table


def test_pair_count_void_segments_1gib(dev):
    """The pair-count scan at 1 GiB (128 iterations per workgroup: a first segment of 64, then the rest): uniform bytes
    (both segments pass their checksums), then the same corpus with the end of every other workgroup's range one repeated
    byte (> 65,535 equal pairs in the second segment: a 16-bit counter wraps, the segment is void, restored from its
    snapshot and recounted with sweeps), twice in a row; against a count done with torch on the device."""
    from bench import splitmix64_device
    n = 1 << 30
    keep, corpus = splitmix64_device(7, n, dev)
    torch.cuda.synchronize()
    with mbpe.Trainer(0) as tr:
        tr.load_corpus_device(corpus.data_ptr(), n, keep=keep)
        want = _pair_table_torch(corpus).cpu().numpy()
        for _ in range(2):
            assert np.array_equal(tr.pair_count_u8().astype(np.int64), want)
        grid = torch.cuda.get_device_properties(0).multi_processor_count
        per = n // grid
        for w in range(0, grid, 2):
            corpus[(w + 1) * per - per // 6:(w + 1) * per - 40000] = 7      # (inside the last eighth and a bit more)
        torch.cuda.synchronize()
        want = _pair_table_torch(corpus).cpu().numpy()
        assert int(want[(7 << 8) | 7]) > 1 << 26
        for _ in range(2):
            assert np.array_equal(tr.pair_count_u8().astype(np.int64), want)


def test_config4_order_inside_large_batches_4gib(dev):
    """BASELINE config 4 at its full size with default options: 48 checkpoints at RANDOM merge counts, three quarters of
    them inside merges 300..17,000 -- the phase whose batches hold 1,000-4,096 pairs (the 128 x 128 pairs of two halves
    of the byte alphabet) -- so that train_steps(n) cuts those batches at arbitrary members.  At every checkpoint the
    table equals a recount of the stream, cell by cell, and the merge committed next is the recount's argmax under
    (count desc, first asc, second asc), PairCount.h:194-207, :262-269; a second, uninterrupted training gives the same
    merges, so the order inside the batches that were NOT cut is the one the cut ones proved."""
    from bench import splitmix64_device
    n = 4 << 30
    keep, corpus = splitmix64_device(42, n, dev)
    torch.cuda.synchronize()
    total = 32000 - 256
    rng = np.random.default_rng(4)
    ats = sorted(set([int(x) for x in rng.integers(300, 17000, size=36)] + [int(x) for x in rng.integers(17000, total - 1, size=12)]))
    with mbpe.Trainer(0) as tr:
        tr.load_corpus_device(corpus.data_ptr(), n, keep=keep)
        tr.train_begin(32000)
        big = 0
        for at in ats:
            have = len(tr.train_result()[0])
            if at > have:
                before = tr.stats()["n_batches"]
                assert tr.train_steps(at - have) == at - have
                # (ONE pass of hundreds of merges up to the checkpoint: a batch that would have gone on -- the natural
                #  batches of this phase hold 1,000-4,096 pairs -- cut at an arbitrary member by the merge limit)
                if tr.stats()["n_batches"] - before == 1 and at - have >= 100:
                    big += 1
            r = C.argmax_at_checkpoint(tr, torch, dev)
            assert r["ok"] and r["merge"] == at, r
        assert big >= 12, big
        tr.train_steps(total)
        m1, c1 = tr.train_result()
        assert len(m1) == total and C.counts_nonincreasing(c1)
        tr.train_begin(32000)
        assert tr.train_steps(total) == total
        m2, c2 = tr.train_result()
        assert np.array_equal(m1, m2) and np.array_equal(c1, c2)


def test_config4_first_1500_merges_against_oracle_16mib(dev):
    """16 MiB of the benchmark bytes, default options, the first 1,500 merges against the oracle (every pair and count),
    then stream and pair table: batches of hundreds of pairs, pass-overs, the byte-table lookup."""
    n = 16 << 20
    data = O.splitmix64_bytes(42, n)
    k = 1500
    with mbpe.Trainer(0) as tr:
        tr.load_corpus(data)
        tr.train_begin(32000)
        assert tr.train_steps(k) == k
        m, c = tr.train_result()
        st = tr.stats()
        assert st["n_batches"] < k // 10
        ost = O.State(data)
        for i in range(k):
            top = ost.top()
            assert (int(m[i][0]), int(m[i][1]), int(c[i])) == top, i
            ost.merge(top[0], top[1], 256 + i)
        assert np.array_equal(tr.stream()[0], ost.stream()[0])
        assert {kk: v for kk, v in tr.pairs_dict().items() if v} == {kk: v for kk, v in ost.table_dict().items() if v}
        ost.close()


@pytest.mark.parametrize("chunked", [False, True])
def test_text_merge_order_at_checkpoints_71mb(dev, chunked):
    """The same proof on text (shakespeare.txt x 64, 71 MB; whole and under the GPT-4 split pattern, chunk ends as
    slot bits): frequent pairs (the delta-cache instantiations), batches that go through the hash table, (t,t)
    members, tiles that lose most of their tokens and stay in prefix form.  At every checkpoint the table equals a
    recount of the stream, the next merge is the recount's argmax, and every tile holds its tokens in front."""
    data = np.frombuffer(read_data("shakespeare.txt") * 64, dtype=np.uint8)
    off = mbpe.presplit(O.GPT4_SPLIT_PATTERN, data) if chunked else None
    vocab = 256 + 6000
    with mbpe.Trainer(0) as tr:
        tr.load_corpus(data, off)
        tr.train_begin(vocab)
        for at in (0, 3, 40, 400, 1500, 3500, 5900):
            have = len(tr.train_result()[0])
            if at > have:
                assert tr.train_steps(at - have) == at - have
            r = C.argmax_at_checkpoint(tr, torch, dev)
            assert r["ok"] and r["merge"] == at, r
            assert C.tiles_in_prefix_form(tr, torch, dev), at
        tr.train_steps(vocab - 256)
        m, c = tr.train_result()
        assert len(m) == vocab - 256 and C.counts_nonincreasing(c)


def test_library_rccl_next_to_torch_distributed(dev):
    """bench.py at N > 1 initialises torch.distributed's NCCL (= RCCL) backend first and the library's own
    communicator (dlopen of librccl.so.1, ncclCommInitRank) second, in the same process.  Both must work
    side by side; this is the 1-rank version of that sequence."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        created = True
    try:
        x = torch.ones(1024, device=dev)
        dist.all_reduce(x)                               # forces torch's communicator into existence
        torch.cuda.synchronize()
        uid = [mbpe.comm_unique_id()]
        dist.broadcast_object_list(uid, src=0)
        data = O.splitmix64_bytes(12, 200000)
        want_m, want_c = O.train(data, 256 + 120)
        with mbpe.Trainer(0) as t:
            t.comm_init(uid[0], 0, 1)
            t.set_option("force_exchange", 1)
            t.load_corpus(data)
            assert t.train_begin(256 + 120) == mbpe.OK
            assert t.train_steps(120) == 120
            m, c = t.train_result()
        assert m.tolist() == want_m.tolist() and c.tolist() == want_c.tolist()
        dist.all_reduce(x)                               # torch's communicator still works afterwards
        torch.cuda.synchronize()
        assert float(x[0]) == 1.0
    finally:
        if created:
            dist.destroy_process_group()

"""What a multi-GPU run of the benchmark workload would put on the wire, sequence by sequence: runs the whole
training on one GPU one sequence at a time and prices each sequence's sum all-reduce from the merges it committed and
the ids that existed when it started -- train.cpp exchange_words(): header + batch header (m_j, ADJ: 1024 + 1024^2
words) + the rows L_j, R_j (lr_pitch(ids) cells) of the batch's pairs; the round-2 layout exchanged 2 x 1024 x ids
words whatever the batch held.  (Pairs that validation dropped or the one-pair path are not visible from here: the
figure is a lower bound by a few per cent.)  Output: profiles/r0N_exchange_bytes.json."""
import json, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "minbpe-cc_amd", "python"))
sys.path.insert(0, ROOT)
import torch, mbpe
from bench import splitmix64_device
dev = torch.device("cuda", 0)
n, vocab = 4 << 30, 32000
keep, corpus = splitmix64_device(42, n, dev)
torch.cuda.synchronize()
tr = mbpe.Trainer(0)
tr.load_corpus_device(corpus.data_ptr(), n, keep=keep)
tr.set_option("batch", 1)
tr.set_option("max_batch", 1024)      # (what a sharded run caps its batches at: the exchange grows with the cap)
tr.train_begin(vocab)
hdr = (2 + 8 * 8 + 3) // 4 * 4
hdrb = 1024 + 1024 * 1024
k, seqs = 0, []
while k < vocab - 256:
    got = tr.train_sequences(1)
    if got == 0:
        break
    pitch = (256 + k + 63) & ~63
    seqs.append({"first_merge": k, "merges": got, "words_r03": hdr + hdrb + 2 * got * pitch,
                 "words_r02_layout": hdr + hdrb + 2 * 1024 * (256 + k + 1024)})
    k += got
tot3 = sum(s["words_r03"] for s in seqs) * 4
tot2 = sum(s["words_r02_layout"] for s in seqs) * 4
print(json.dumps({"workload": "config 4: 4 GiB SplitMix64, vocab 32000, 8 ranks (header of 8 rank edges)", "sequences": len(seqs),
                  "bytes_per_sequence_avg_r03": tot3 / len(seqs), "bytes_per_sequence_max_r03": max(s["words_r03"] for s in seqs) * 4,
                  "bytes_per_sequence_avg_r02_layout": tot2 / len(seqs), "bytes_total_r03": tot3, "bytes_total_r02_layout": tot2,
                  "of_which_batch_header_bytes": (hdr + hdrb) * 4,
                  "per_sequence": [[s["first_merge"], s["merges"], s["words_r03"] * 4] for s in seqs]}))

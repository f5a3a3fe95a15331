#!/usr/bin/env python3
"""Regenerates tests/golden/*.model with the CPU oracle and checks each file
against the sha256 recorded in SURVEY.md section 8c.  A fixture is only written
when its digest matches.

Where the digests come from: the survey session compiled the reference's two
headers against a stand-in for Boost.MultiIndex (the image has no Boost) and
hashed the .model files that build wrote.  That is NOT a build of the reference
as shipped, so on its own it pins nothing.  What ties the committed fixtures to
the reference's algorithm is tests/test_bruteforce_cpu.py: an independent
recount-per-merge BPE in numpy (no oracle code, chunk boundaries from Python's
`regex`) reproduces the lexical AND first-mode fixtures byte for byte, and the
reference-held known-answer tests (test.cpp:15-106, :136-186) are checked in
tests/test_oracle.py.

Inputs: tests/golden/data/*.txt are the reference's own data files
(data/taylorswift.txt, shakespeare.txt, small.txt, sample.txt, special1.txt,
specialtokensample.txt), copied verbatim; SplitMix64 corpora are generated.

Chunk boundaries for the gpt2/gpt4 encoders come from the system PCRE2
(libpcre2-8.so.0) through the host library's mbpe_presplit.

Usage: python tests/golden/make_golden.py   (needs the built host library)
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "minbpe-cc_amd", "python"))

import oracle as O  # noqa: E402
import mbpe  # noqa: E402

# (fixture name, input, encoder, mode, vocab, sha256 from SURVEY.md 8c)
CASES = [
    ("taylorswift_basic_lexical_512", "taylorswift.txt", "basic", "lexical", 512,
     "b43f3d5dfeec4cd322c706cc963189ad95d4b7b891f3ac2660c13f5010e44b07"),
    ("taylorswift_basic_first_512", "taylorswift.txt", "basic", "first", 512,
     "ecff325f8c514737fa4d3ce72c719ebb211dd645b59da77697bf0853f33e0954"),
    ("taylorswift_gpt4_lexical_512", "taylorswift.txt", "gpt4", "lexical", 512,
     "1b5c2bbcf04d0794d7f8486336a70e12c291e8f41796ed4ee8a85f6c2c8a124b"),
    ("taylorswift_gpt4_first_512", "taylorswift.txt", "gpt4", "first", 512,
     "b58d827c48c4403a3a141a9f9a8a7cedaec18a0b094dc8462612ecaa8c0f57df"),
    ("taylorswift_gpt2_lexical_512", "taylorswift.txt", "gpt2", "lexical", 512,
     "d0eb6352e1a44fb4dd1ddd5c9ef4b6ed7248f8bb9186fe2eeabe2908f3b05494"),
    ("shakespeare_basic_lexical_512", "shakespeare.txt", "basic", "lexical", 512,
     "48e73bab6052ef168b2ccc8f2ab6ff1a20280e8f0f2e4dd1c84a8d33e8f9993a"),
    ("shakespeare_gpt4_lexical_512", "shakespeare.txt", "gpt4", "lexical", 512,
     "6df25711b0f138bc4b79a90d373f6ee0142b5607949afac8f5f836b50aad7703"),
    ("splitmix42_1MiB_basic_lexical_512", "splitmix:42:1048576", "basic", "lexical", 512,
     "ec1ff6f2b1140a5ac514812b435ceeb2f2cd6df717407f4e5fa5e01867ad324d"),
    ("splitmix42_1MiB_basic_lexical_1024", "splitmix:42:1048576", "basic", "lexical", 1024,
     "1085f599f1e619fb1527e3f8d1363bd98a7b32df617358708b31c769ed98a8fa"),
    ("splitmix42_64KiB_basic_lexical_2048", "splitmix:42:65536", "basic", "lexical", 2048,
     "bcec3c0ef1096a54787d218bc0762757a5df559f02fbb7ef6f4237d3ba473288"),
]


def load_input(spec):
    if spec.startswith("splitmix:"):
        _, seed, n = spec.split(":")
        return O.splitmix64_bytes(int(seed), int(n)).tobytes()
    with open(os.path.join(HERE, "data", spec), "rb") as f:
        return f.read()


def main():
    index = {}
    for name, spec, enc, mode, vocab, sha in CASES:
        data = load_input(spec)
        off = None if enc == "basic" else mbpe.presplit(O.PATTERNS[enc], data)
        omode = O.LEXICAL if mode == "lexical" else O.FIRST
        st = O.State(data, off, omode)
        merges_l, counts_l = [], []
        for i in range(vocab - 256):
            top = st.top()
            if top is None:
                break
            merges_l.append(top[:2])
            counts_l.append(top[2])
            st.merge(top[0], top[1], 256 + i)
        final_len = int(len(st.stream()[0]))
        st.close()
        import numpy as np
        merges = np.array(merges_l, dtype=np.uint32).reshape(-1, 2)
        counts = np.array(counts_l, dtype=np.int32)
        m2, c2 = O.train(data, vocab, off, omode)
        assert m2.tolist() == merges.tolist() and c2.tolist() == counts.tolist()
        blob = O.model_bytes(O.PATTERNS[enc], merges)
        got = hashlib.sha256(blob).hexdigest()
        if got != sha:
            raise SystemExit("%s: oracle digest %s != SURVEY digest %s" % (name, got, sha))
        with open(os.path.join(HERE, name + ".model"), "wb") as f:
            f.write(blob)
        index[name] = {"input": spec, "encoder": enc, "mode": mode, "vocab": vocab,
                       "sha256": sha, "bytes": len(blob),
                       "first_count": int(counts[0]), "last_count": int(counts[-1]),
                       "final_len": final_len}
        print("ok", name, len(blob))
    with open(os.path.join(HERE, "index.json"), "w") as f:
        json.dump(index, f, indent=1, sort_keys=True)
        f.write("\n")


if __name__ == "__main__":
    main()

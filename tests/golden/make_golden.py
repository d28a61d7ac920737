"""Generates the committed fixtures in tests/golden/.

There is no reference implementation to import or run (rust/sparrowhawk-asm is an empty submodule,
no Rust toolchain) and the reference ships no fixture for this path, so:
  * hand_cases.json   — inputs built by tests/cases.py with expected contig sets that follow from
                        SPEC.md by reasoning (tip / bubble / cycle / hairpin); no implementation
                        was used to produce the expectations;
  * synth_*.json      — seeded synthetic reads with the outputs of the CPU oracle (oracle/), pinning
                        the oracle against silent change and giving the GPU tests an expectation
                        that does not need the oracle library at run time.
Run from the repo root:  python tests/golden/make_golden.py
"""
import base64
import gzip
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import cases  # noqa: E402
from util import make_dataset, run_oracle  # noqa: E402


def b64gz(data: bytes) -> str:
    return base64.b64encode(gzip.compress(data, 9, mtime=0)).decode()


def main():
    hand = {}
    c = cases.tip_case()
    hand["tip"] = dict(k=c["k"], min_count=0, min_qual=0, fastq=c["fastq"].decode(),
                       expect=sorted(c["with_removal"]), expect_no_dead_end_removal=sorted(c["without_removal"]))
    c = cases.bubble_case()
    hand["bubble"] = dict(k=c["k"], min_count=0, min_qual=0, fastq=c["fastq"].decode(),
                          expect=sorted(c["with_collapse"]), expect_no_bubble_collapse=sorted(c["without_collapse"]))
    c = cases.cycle_case()
    hand["cycle"] = dict(k=c["k"], min_count=0, min_qual=0, fastq=c["fastq"].decode(), expect=sorted(c["expect"]))
    c = cases.palindrome_case()
    hand["hairpin"] = dict(k=c["k"], min_count=0, min_qual=0, fastq=c["fastq"].decode(), expect=sorted(c["expect"]))
    json.dump(hand, open(os.path.join(HERE, "hand_cases.json"), "w"), indent=1)

    for name, (glen, cov, err, k, mc, fit, seed) in {
        "synth_k31_clean": (6000, 30, 0.0, 31, 3, False, 11),
        "synth_k31_err": (6000, 40, 0.01, 31, 2, True, 12),
        "synth_k51_err": (6000, 40, 0.01, 51, 2, False, 13),
    }.items():
        g, fq = make_dataset(glen, cov, err=err, seed=seed)
        o = run_oracle([fq], k=k, min_count=mc, min_qual=20, do_fit=fit)
        pre = o.preprocessing_json()
        o.assemble()
        asm = o.assembly_json()
        json.dump(dict(k=k, min_count=mc, min_qual=20, do_fit=fit, fastq_gz_b64=b64gz(fq),
                       preprocessing_info=json.loads(pre),
                       assembly_sha256=hashlib.sha256(asm.encode()).hexdigest(),
                       outfasta=json.loads(asm)["outfasta"], ncontigs=json.loads(asm)["ncontigs"],
                       total_instances=o.total_instances),
                  open(os.path.join(HERE, name + ".json"), "w"))
    print("written:", sorted(os.listdir(HERE)))


if __name__ == "__main__":
    main()

"""Generates tests/golden/ref_vectors.npz from the REFERENCE's own kernel headers compiled for the CPU
(oracle/_ref/libpsdf_ref.so, `make -C oracle ref`; needs /root/reference) and tests/golden/encoding_vectors.npz from
the encoding oracle (self-golden: the upstream encoding source is absent, so these only guard the oracle against
drift -- parity of the encoding stays UNPINNED, see oracle/permuto_oracle.py).

    python tests/golden/make_golden.py

Inputs are regenerated from the seeds in tests/golden/cases.py by the test; only outputs are stored."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from tests.golden import cases  # noqa: E402

if __name__ == "__main__":
    O.build(ref=True)
    ref = O.Oracle("ref")
    out = cases.run_all(ref)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref_vectors.npz"), **out)
    print("ref_vectors.npz:", len(out), "arrays,", sum(v.nbytes for v in out.values()) // 1024, "KiB raw")
    enc = cases.run_encoding()
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "encoding_vectors.npz"), **enc)
    print("encoding_vectors.npz:", len(enc), "arrays")

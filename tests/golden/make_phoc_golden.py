"""Generates tests/golden/phoc_words.npz from the REFERENCE's own PHOC C extension (compiled from
/root/reference/pythia/utils/phoc/src/cphoc.c into oracle/_ref/ by oracle/Makefile).  Runs only in the authoring
container; the fixture is data: raw tokens, their normalised form (build_phoc.py:9-12 applied by the reference module
itself where importable, else by the same three Python operations) and the 604 output bits, bit-packed."""
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import phoc_oracle as po  # noqa: E402

ref = po.reference_build_phoc_raw()
assert ref is not None, "run `make -C oracle` first (needs /root/reference)"
random.seed(20241022)
alpha = "abcdefghijklmnopqrstuvwxyz0123456789"
raw = ["", " ", "a", "z", "0", "9", "th", "he", "The", "THERE", "Coca-Cola", "  stop ", "EXIT->", "24/7", "O'Neil", "naïve", "İstanbul", "K",
       "x" * 63, "internationalization", "aaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaa", "llanfairpwllgwyngyll", "e" * 5, "ab" * 20, "0123456789" * 3]
common = ("the he in er an re es on st nt en at ed nd to or ea ti ar te ng al it as is ha et se ou of le sa ve ro ra ri hi ne me de co "
          "ta ec si ll so na li la el").split()
for n in range(1, 41):
    raw.append("".join(random.choice(alpha) for _ in range(n)))
    raw.append("".join(random.choice(common) for _ in range(n))[:n])
for _ in range(150):
    n = random.randint(1, 24)
    raw.append("".join(random.choice(alpha + "ABCXYZ -.'") for _ in range(n)))
norm = [po.normalize(t) for t in raw]
bits = np.stack([np.packbits(np.array(ref(w), dtype=np.float32) > 0) for w in norm])
for w, b in zip(norm, bits):       # the values are exactly 0.0 / 1.0
    v = np.array(ref(w), dtype=np.float32)
    assert set(np.unique(v)) <= {0.0, 1.0}
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "phoc_words.npz"), raw=np.array(raw), norm=np.array(norm),
                    bits=bits)
print(len(raw), "tokens,", bits.shape, "packed bits")

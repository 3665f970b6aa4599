"""TEST INFRASTRUCTURE: Python face of oracle/phoc_oracle.c (the CPU restatement of the reference's PHOC descriptor).
Only tests/, ``__graft_entry__.smoke()`` and bench.py's cpu_baseline leg may import this module.

``normalize`` restates ``pythia/utils/phoc/build_phoc.py:9-12`` (lower, strip, keep [a-z0-9]); ``build_phoc`` is the
composition the reference's ``PhocProcessor`` applies per OCR token (``pythia/datasets/processors.py:904-928``).
``reference_build_phoc`` calls the reference's own C extension compiled from its source into ``oracle/_ref/``
(``oracle/Makefile``) when that build is present - it is what pins this restatement.
"""
import ctypes
import importlib.util
import os
import glob
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
PHOC_DIM = 604
_ALPHABET = set("abcdefghijklmnopqrstuvwxyz0123456789")
_lib = None


def _load():
    global _lib
    if _lib is None:
        so = os.path.join(_HERE, "_build", "libphoc_oracle.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-C", _HERE, "_build/libphoc_oracle.so"], stdout=subprocess.DEVNULL)
        _lib = ctypes.CDLL(so)
        _lib.phoc_build.argtypes = [ctypes.c_char_p, ctypes.c_void_p]
        _lib.phoc_build.restype = ctypes.c_int
        _lib.phoc_build_batch.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_void_p]
        _lib.phoc_build_batch.restype = ctypes.c_int
    return _lib


def normalize(token):
    token = token.lower().strip()
    return "".join(c for c in token if c in _ALPHABET)


def build_phoc_raw(word):
    """604 floats for an already normalised word (raises like the reference on a foreign symbol)."""
    out = np.zeros(PHOC_DIM, dtype=np.float32)
    if _load().phoc_build(word.encode("ascii"), out.ctypes.data) != 0:
        raise RuntimeError("unigram outside [a-z0-9] in %r" % word)
    return out


def build_phoc(token):
    return build_phoc_raw(normalize(token))


def build_phoc_batch(slots):
    """slots: uint8 [n, width] NUL-padded normalised tokens -> float32 [n, 604]."""
    slots = np.ascontiguousarray(slots, dtype=np.uint8)
    out = np.zeros((slots.shape[0], PHOC_DIM), dtype=np.float32)
    rc = _load().phoc_build_batch(slots.ctypes.data, slots.shape[0], slots.shape[1], out.ctypes.data)
    if rc != 0:
        raise RuntimeError("phoc_build_batch failed (%d)" % rc)
    return out


def reference_build_phoc_raw():
    """The reference extension's ``build_phoc`` (list of 604 floats for a normalised str), or None if oracle/_ref has
    not been built (it is built in the authoring container only and travels with the snapshot)."""
    hits = glob.glob(os.path.join(_HERE, "_ref", "cphoc*.so"))
    if not hits:
        return None
    spec = importlib.util.spec_from_file_location("cphoc", hits[0])
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.build_phoc

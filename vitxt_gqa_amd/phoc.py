"""OCR-token PHOC features on the GPU: host-side token normalisation + byte packing around ``ops.phoc``.

Reference: ``PhocProcessor`` (``pythia/datasets/processors.py:904-928``) calls ``build_phoc(token)`` per OCR token, which
lower-cases / strips / filters the token in Python (``pythia/utils/phoc/build_phoc.py:9-12``) and hands the result to a C
extension that returns 604 floats (``pythia/utils/phoc/src/cphoc.c``); padded slots of the ``[max_length, 604]`` output
keep the fill value 0.  Here the Python normalisation stays on the host (it is Unicode-aware string work), the tokens
travel as ``width``-byte slots, and the 604-d rows are produced in HBM by ``t2s_phoc`` - 2416 bytes per token never
cross PCIe.
"""
import numpy as np
import torch

from . import ops

PHOC_DIM = 604
_ALPHABET = frozenset("abcdefghijklmnopqrstuvwxyz0123456789")


def normalize_token(token):
    """``build_phoc.py:10-11``: lower, strip, keep only [a-z0-9]."""
    token = token.lower().strip()
    return "".join(c for c in token if c in _ALPHABET)


def pack_tokens(tokens, max_length, width=64):
    """List of raw OCR token strings (one sample) -> uint8 [max_length, width] NUL-padded slots.  Like the reference
    processor, tokens beyond ``max_length`` are dropped and missing ones stay empty (all-zero features)."""
    out = np.zeros((max_length, width), dtype=np.uint8)
    for i, tok in enumerate(tokens[:max_length]):
        b = normalize_token(tok).encode("ascii")
        if len(b) > width:
            raise ValueError("normalised OCR token of %d bytes does not fit the %d-byte slot: %r" % (len(b), width, tok))
        out[i, :len(b)] = np.frombuffer(b, dtype=np.uint8)
    return out


def check_slots(slots):
    """Reject what the reference extension would raise on (a byte outside [a-z0-9]) and malformed padding."""
    a = np.asarray(slots)
    ok = ((a >= ord("a")) & (a <= ord("z"))) | ((a >= ord("0")) & (a <= ord("9"))) | (a == 0)
    if not ok.all():
        raise RuntimeError("PHOC token bytes outside [a-z0-9] (normalise tokens with normalize_token first)")
    nz = a != 0
    if (nz[..., 1:] & ~nz[..., :-1]).any():
        raise RuntimeError("PHOC token slots must be NUL padded at the end only")


def phoc_features(slots, device="cuda:0", out=None):
    """uint8 [..., width] host slots (numpy or CPU tensor) or an already uploaded CUDA tensor -> fp32 [..., 604] in HBM."""
    if torch.is_tensor(slots) and slots.is_cuda:
        return ops.phoc(slots, out=out)
    arr = slots.numpy() if torch.is_tensor(slots) else np.ascontiguousarray(slots, dtype=np.uint8)
    check_slots(arr)
    return ops.phoc(torch.from_numpy(arr).to(device, non_blocking=True), out=out)

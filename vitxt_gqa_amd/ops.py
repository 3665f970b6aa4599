"""Thin tensor-level wrappers over the C ABI (no autograd here; see functional.py).

Every function validates shapes on the host before launching (an out-of-bounds kernel can take the
whole GPU node down) and raises RuntimeError with the library's message on failure.
"""
import math
import os

import torch

from . import hipext as X

HID = 768
HEADS = 12
HEAD_DIM = 64
# matrix products per (query, key) pair the attention backward EXECUTES: the algorithm needs 5 (S, dP, dV, dK, dQ); the fused form
# (t2s_attn_bwd_fused) runs exactly those, the two-kernel form (key-stationary dK/dV + query-stationary dQ, no atomics) recomputes
# S and dP: 7.  attn_bwd() records which form a call took in LAST_ATTN_BWD_PRODUCTS (bench.py reports executed and algorithmic).
LAST_ATTN_BWD_PRODUCTS = 7


def _rows768(t):
    assert t.is_contiguous() and t.shape[-1] == HID, "expected contiguous [..., 768], got %s" % (tuple(t.shape),)
    return t.numel() // HID


class KeyList:
    """Compacted visible-key list of a batch (replaces the [B,1,L,L] additive mask, t2s.py:413-419,609-618)."""

    def __init__(self, idx, cnt, n_dec, dec_q0, cap_hint=None, valid8=None):
        self.idx, self.cnt, self.n_dec, self.dec_q0 = idx, cnt, n_dec, dec_q0
        self.valid8 = valid8        # [B, L] uint8 mask the list was compacted from (lets the backward skip its zero fill)
        # static upper bound on (#valid prefix keys + n_dec): lets the dK/dV grid skip empty key blocks
        self.cap_hint = cap_hint if cap_hint is not None else idx.shape[1]
        self.bound_is_structural = False   # the caller vouches that cap_hint can never be exceeded (top-k masks): allows compact()
        self._compact = {}

    def compact(self, L):
        """For a list whose static bound is structural and small (the pos / neg MMT passes: <= 537 / 62 of 10 132 rows are
        ever keys): the rows of a [B * L, .] row tensor that are keys (-> ``flat_rows`` int64 [B * capK], list order, the
        positions behind a sample's list point at its row 0) and the key list of a [B, capK, .] buffer gathered that way
        (``keys_c``: position p is row p; same counts, same decoder rule, same dropout positions).  K and V - and their
        gradients - then only ever exist for those rows."""
        if L not in self._compact:
            B = self.idx.shape[0]
            capK = min(self.idx.shape[1], (self.cap_hint + 63) // 64 * 64)
            pos = torch.arange(capK, device=self.idx.device, dtype=torch.int32)
            # The caller vouches for the bound (bound_is_structural); should it be violated all the same, the list is CLAMPED to the
            # buffer (no host sync): the prefix keys beyond capK - n_dec are dropped - a wrong result for that sample, but no key-list
            # position ever points past the [B, capK] buffers the kernels index (an out-of-bounds K / V read can take the node down).
            cnt_c = torch.clamp(self.cnt, max=capK - self.n_dec)
            live = pos.unsqueeze(0) < (cnt_c + self.n_dec).unsqueeze(1)
            src = self.idx[:, :capK]
            if capK < self.idx.shape[1]:
                # the decoder keys close the list at positions cnt .. cnt + n_dec - 1 of the ORIGINAL list: re-read them there, so
                # that a clamped list still ends in its decoder rows
                dec_pos = (self.cnt.unsqueeze(1) + (pos.unsqueeze(0) - cnt_c.unsqueeze(1))).clamp(0, self.idx.shape[1] - 1).long()
                is_dec = pos.unsqueeze(0) >= cnt_c.unsqueeze(1)
                src = torch.where(is_dec, torch.gather(self.idx, 1, dec_pos), src)
            rows = torch.where(live, src, torch.zeros((), dtype=torch.int32, device=self.idx.device)).long()
            flat = (rows + torch.arange(B, device=rows.device).unsqueeze(1) * L).reshape(-1)
            keys_c = KeyList(pos.unsqueeze(0).expand(B, capK).contiguous(), cnt_c, self.n_dec, self.dec_q0, min(self.cap_hint, capK), None)
            self._compact[L] = (keys_c, flat, capK)
        return self._compact[L]


def compact_keys(valid, n_dec=0, dec_row0=0, cap_hint=None):
    """valid: [B, L] bool/uint8 over the prefix rows -> KeyList."""
    assert valid.dim() == 2
    B, L = valid.shape
    v8 = valid.to(torch.uint8).contiguous()
    cap = L + n_dec
    idx = torch.empty(B, cap, dtype=torch.int32, device=valid.device)
    cnt = torch.empty(B, dtype=torch.int32, device=valid.device)
    X.check(X.lib().t2s_compact_keys(X.ptr(v8), X.ptr(idx), X.ptr(cnt), B, L, cap, n_dec, dec_row0, X.stream()),
            "t2s_compact_keys")
    if cap_hint is not None:
        cap_hint = min(cap, cap_hint)
    return KeyList(idx, cnt, n_dec, dec_row0, cap_hint, v8)


def _attn_views(qkv):
    """qkv: [B, L, 2304] fused projection -> (q, k, v) views, row stride 2304."""
    assert qkv.dim() == 3 and qkv.shape[-1] == 3 * HID and qkv.is_contiguous()
    return qkv[..., :HID], qkv[..., HID:2 * HID], qkv[..., 2 * HID:]


def attn_dropout_mask(B, Lq, Lk, drop_p, drop_seed, device):
    """[B, 12, Lq, Lk] 0/1 keep mask of the attention-probability dropout (Lk = key-list positions); tests."""
    out = torch.empty(B, HEADS, Lq, Lk, dtype=torch.uint8, device=device)
    X.check(X.lib().t2s_attn_dropout_mask(X.ptr(out), B, HEADS, Lq, Lk, float(drop_p), int(drop_seed), X.stream()),
            "t2s_attn_dropout_mask")
    return out


def _split_views(q, kv, keys):
    """q [B, L, 768] + kv [B, capK, 1536] (K | V of the gathered key rows, ``keys`` = the compact list) -> (q, k, v) views."""
    B, L, _ = q.shape
    assert q.shape == (B, L, HID) and q.is_contiguous() and kv.dim() == 3 and kv.shape[0] == B and kv.shape[2] == 2 * HID and kv.is_contiguous()
    assert kv.dtype == q.dtype and kv.shape[1] == keys.idx.shape[1] and keys.valid8 is None
    return q, kv[..., :HID], kv[..., HID:]


def attn_fwd(qkv, keys, scale=1.0 / 8.0, drop_p=0.0, drop_seed=0, kv=None):
    """Self-attention over a fused QKV buffer [B, L, 2304]; or, with ``kv`` [B, capK, 1536], over queries ``qkv`` = Q [B, L, 768]
    and the K | V rows of a compact key buffer (KeyList.compact).  Returns (ctx [B, L, 768], lse [B, 12, L])."""
    B, L, _ = qkv.shape
    q, k, v = _attn_views(qkv) if kv is None else _split_views(qkv, kv, keys)
    out = torch.empty(B, L, HID, dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty(B, HEADS, L, dtype=torch.float32, device=qkv.device)
    _check_keys(keys, B, L)
    X.check(X.lib().t2s_attn_fwd(
        X.ptr(q), X.ptr(k), X.ptr(v), X.ptr(out), X.ptr(lse), X.ptr(keys.idx), X.ptr(keys.cnt),
        B, HEADS, L, keys.idx.shape[1], keys.n_dec, keys.dec_q0,
        q.stride(1), q.stride(0), k.stride(1), k.stride(0), out.stride(1), out.stride(0),
        scale, X.dtype_code(qkv), float(drop_p), int(drop_seed), X.stream()),
        "t2s_attn_fwd")
    return out, lse


# True: bf16 launches (with or without attention dropout) that may see >= ATTN_BWD_FUSED_MIN_KEYS keys use the fused 5-product kernel
# (t2s_attn_bwd_fused: S and dP computed once, dQ summed across the key blocks of a (sample, head)); False (T2S_ATTN_BWD_FUSED=0):
# always the two-kernel 7-product form.  Both are bit-reproducible when the dQ sum runs as the ordered hand-off (ATTN_BWD_DQ_MODE 1,
# the default: key blocks add their tiles in block order, plain stores); T2S_ATTN_BWD_DQ=atomic restores the fp32-atomic sum of
# rounds 2-3 (dQ then differs in its last fp32 bits from run to run).
ATTN_BWD_FUSED = os.environ.get("T2S_ATTN_BWD_FUSED", "1") != "0"
ATTN_BWD_FUSED_MIN_KEYS = 2048
ATTN_BWD_FUSED_MIN_ROWS = 1024
ATTN_BWD_DQ_MODE = 0 if os.environ.get("T2S_ATTN_BWD_DQ", "handoff") == "atomic" else 1
# The hand-off's running sums stay in the L2 of the XCD that runs a (sample, head)'s key blocks (plain stores); the kernel checks that
# every XCD group of workgroups really sits on one XCD and reports a violation through the status word (HandoffPlacement below).
# T2S_FB_HANDOFF_SCOPE=agent (dq_mode bit 9) selects write-through stores instead - correct under any placement, 3 - 5 % slower.
if ATTN_BWD_DQ_MODE == 1 and os.environ.get("T2S_FB_HANDOFF_SCOPE", "xcd") == "agent":
    ATTN_BWD_DQ_MODE |= 0x200
_KEEP_DQ32 = os.environ.get("T2S_KEEP_DQ32", "0") == "1"     # tools/fused_stamps.py: keep the workspace, whose tail holds the diagnostic
_LAST_DQ32 = None                                             # build's cycle stamps (otherwise it is freed with the call: 2 GB at B=64)
FUSED_CTRL_STATUS_WORD = 24                                   # include/t2s_hip.h: uint32 word of the workspace, bit 0 = a hand-off spin timed out, bit 1 = an XCD group ran on two XCDs
_STICKY = {}                                                  # device index -> int32 [4]: (OR of every fused launch's status word, launches seen, bit 0 as a 0/1 flag, bit 1 as a 0/1 flag)


class HandoffError(RuntimeError):
    """The fused attention backward's dQ hand-off reported a failure in its status word: the gradients of that step are not usable and
    the optimizer step that followed was gated off on the device."""


class HandoffTimeout(HandoffError):
    """A bounded wait of the fused attention backward's dQ hand-off timed out (a workgroup died or the card is oversubscribed): the dQ
    rows behind it are NaN and the optimizer step that follows is gated off on the device - the gradients of this step are not usable."""


class HandoffPlacement(HandoffError):
    """Workgroups of one XCD group (equal blockIdx % 8) ran on different XCDs, so the running dQ sums - kept in ONE XCD's L2 - may have
    been read stale (status bit 1).  Does not happen on an MI355X in its default (SPX) mode; on a device that places workgroups
    differently set T2S_FB_HANDOFF_SCOPE=agent (write-through sums)."""


def fused_status_tensor(device):
    """The sticky status words of ``device`` (created on first use).  Every fused-backward call ORs its workspace's status word into
    word 0 (one 1-thread kernel behind the launch - no copy, no allocation per call); ``FusedClipAdam`` gates its step on it and raises."""
    dev = torch.device(device)
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    if key not in _STICKY:
        _STICKY[key] = torch.zeros(4, dtype=torch.int32, device=dev)
    return _STICKY[key]


def fused_handoff_status(device=None):
    """OR of the status words of every fused-backward call since the last ``reset_fused_status`` (synchronises): 0 = clean, bit 0 = a
    bounded spin of the dQ hand-off timed out (the dQ rows behind it are NaN; cannot happen unless a workgroup died), bit 1 = an XCD
    group of workgroups ran on more than one XCD (the XCD-local running sums may have been read stale)."""
    if not _STICKY:
        return 0
    if device is None:
        st = 0
        for t in _STICKY.values():          # OR over the devices of this process (a sum would turn two timeouts into "placement")
            st |= decode_status_words(t.tolist())
        return st
    return decode_status_words(fused_status_tensor(device).tolist())


def decode_status_words(w):
    """The status bits from the four sticky words: word 0 is the OR of the launches' status words on THIS device; words 2 and 3 carry
    bit 0 and bit 1 once more as 0 / 1 flags, which survive the MAX reduction over ranks of ``GradBuckets.finish()`` (the MAX of the
    OR-ed words of two ranks - a timeout here, a placement violation there - would keep only the larger)."""
    return (int(w[0]) | int(w[2]) | (int(w[3]) << 1)) & 3


def fused_launches_seen(device):
    return int(fused_status_tensor(device)[1].item())


def reset_fused_status():
    for t in _STICKY.values():
        t.zero_()


def _fused_policy(fused, qkv, keys, L, mode):
    """Which backward form a call takes when the caller does not say.  With the ordered hand-off (dq_mode 1, the default) the fused
    five-product kernel wins at EVERY key count of a long query sequence - B = 64, L = 10 132: 3.2 vs 4.6 ms at 75 keys, 6.0 vs 8.5 at 550,
    15.1 vs 19.4 at 1 939 (profiles/r04_light_launches.txt) - so the pos / neg passes' short lists take it too.  With fp32 atomics it only
    pays once the list is long (the round 2-3 rule: a static key bound of >= 2 048), and sequences of a few rows (text_bert: 20) stay on
    the two-kernel form (a 147 KB-LDS workgroup per (sample, head) for one query tile)."""
    if fused is None:
        fused = ATTN_BWD_FUSED and (keys.cap_hint >= ATTN_BWD_FUSED_MIN_KEYS or ((mode & 0xff) == 1 and L >= ATTN_BWD_FUSED_MIN_ROWS))
    return bool(fused) and qkv.dtype == torch.bfloat16


def _call_fused(head, klist3, tail, B, L, mode, device):
    # workspace: control words, hand-off flags and the fp32 dQ sums (either form); cleared as needed inside the call
    global _LAST_DQ32
    nbytes = int(X.lib().t2s_attn_bwd_fused_workspace_bytes(B, HEADS, L))
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=device)
    if _KEEP_DQ32:
        _LAST_DQ32 = ws
    X.check(X.lib().t2s_attn_bwd_fused(*head, X.ptr(ws), nbytes, mode, *klist3, *tail), "t2s_attn_bwd_fused")
    # the launch's status word goes into the device's sticky word (the 2 GB workspace dies with this call)
    X.check(X.lib().t2s_status_accumulate(X.ptr(ws), FUSED_CTRL_STATUS_WORD, X.ptr(fused_status_tensor(device)), X.stream()), "t2s_status_accumulate")


def attn_bwd(qkv, out, dout, lse, keys, scale=1.0 / 8.0, drop_p=0.0, drop_seed=0, fused=None, kv=None, dq_mode=None):
    """Returns dqkv [B, L, 2304] (rows of keys outside the key list get exact zeros in the K/V thirds); with ``kv`` (see
    attn_fwd) returns (dq [B, L, 768], dkv [B, capK, 1536]) - the two-kernel form, positions behind a sample's list zero."""
    global LAST_ATTN_BWD_PRODUCTS
    B, L, _ = qkv.shape
    assert out.is_contiguous() and dout.is_contiguous() and out.shape == (B, L, HID) and dout.shape == (B, L, HID)
    assert lse.shape == (B, HEADS, L) and lse.is_contiguous()
    _check_keys(keys, B, L)
    mode = ATTN_BWD_DQ_MODE if dq_mode is None else int(dq_mode)
    if kv is not None:
        q, k, v = _split_views(qkv, kv, keys)
        dq_, dkv = torch.empty_like(qkv), torch.zeros_like(kv)
        delta = torch.empty_like(lse)
        args = (B, HEADS, L, keys.idx.shape[1], keys.n_dec, keys.dec_q0, keys.cap_hint,
                q.stride(1), q.stride(0), k.stride(1), k.stride(0), out.stride(1), out.stride(0),
                scale, X.dtype_code(qkv), float(drop_p), int(drop_seed), X.stream())
        head = (X.ptr(q), X.ptr(k), X.ptr(v), X.ptr(out), X.ptr(dout), X.ptr(lse), X.ptr(delta),
                X.ptr(dq_), X.ptr(dkv[..., :HID]), X.ptr(dkv[..., HID:]))
        if _fused_policy(fused, qkv, keys, L, mode):
            # the compact key buffer through the fused form (dk / dv rows behind a sample's list stay the zeros written above)
            LAST_ATTN_BWD_PRODUCTS = 5
            _call_fused(head, (X.ptr(keys.idx), X.ptr(keys.cnt), None), args, B, L, mode, qkv.device)
        else:
            LAST_ATTN_BWD_PRODUCTS = 7
            X.check(X.lib().t2s_attn_bwd(*head, X.ptr(keys.idx), X.ptr(keys.cnt), *args), "t2s_attn_bwd")
        return dq_, dkv
    q, k, v = _attn_views(qkv)
    cap = keys.idx.shape[1]
    # self-attention layout: the rows the mask covers, then decoder rows - this call's n_dec of them at dec_q0 (cap == L), or those
    # of all three MMT passes with only this call's listed (shared-prefix layout, cap < L)
    fill_in_kernel = (qkv.dtype == torch.bfloat16 and keys.valid8 is not None and cap <= L and keys.valid8.shape == (B, cap - keys.n_dec)
                      and (keys.n_dec == 0 or keys.dec_q0 >= cap - keys.n_dec))
    # rows outside the key list have exactly zero dK / dV: written by the dQ kernel when it is handed the mask (bf16 path),
    # otherwise by a zero fill of the whole buffer (3 GB at B=64) before the launch
    dqkv = torch.empty_like(qkv) if fill_in_kernel else torch.zeros_like(qkv)
    dq, dk, dv = _attn_views(dqkv)
    delta = torch.empty_like(lse)
    head = (X.ptr(q), X.ptr(k), X.ptr(v), X.ptr(out), X.ptr(dout), X.ptr(lse), X.ptr(delta),
            X.ptr(dq), X.ptr(dk), X.ptr(dv))
    klist = (X.ptr(keys.idx), X.ptr(keys.cnt))
    dims = (B, HEADS, L, cap, keys.n_dec, keys.dec_q0, keys.cap_hint,
            qkv.stride(1), qkv.stride(0), qkv.stride(1), qkv.stride(0), out.stride(1), out.stride(0),
            scale, X.dtype_code(qkv))
    use_fused = _fused_policy(fused, qkv, keys, L, mode)
    LAST_ATTN_BWD_PRODUCTS = 5 if use_fused else 7
    if use_fused:
        _call_fused(head, (*klist, X.ptr(keys.valid8) if fill_in_kernel else None), (*dims, float(drop_p), int(drop_seed), X.stream()), B, L, mode,
                    qkv.device)
    elif fill_in_kernel:
        X.check(X.lib().t2s_attn_bwd_fill(*head, *klist, X.ptr(keys.valid8), *dims, float(drop_p), int(drop_seed), X.stream()), "t2s_attn_bwd_fill")
    else:
        X.check(X.lib().t2s_attn_bwd(*head, *klist, *dims, float(drop_p), int(drop_seed), X.stream()), "t2s_attn_bwd")
    return dqkv


def attn_fwd_rows(q_rows, kv_buf, keys, scale=1.0 / 8.0):
    """Attention of a SUBSET of query rows against a fused QKV buffer: q_rows [B, Lq, 2304-strided view or
    contiguous [B, Lq, 768]], kv_buf [B, L, 2304] (keys/values taken from its K/V thirds).  The decoder rule of
    ``keys`` is applied with the query rows numbered 0..Lq-1 (keys.dec_q0 is ignored: the rows ARE the decoder
    rows).  Used by the greedy decoder, which recomputes only the 12 decoding rows per step.  Returns ctx [B, Lq, 768]."""
    B, L, _ = kv_buf.shape
    Lq = q_rows.shape[1]
    assert q_rows.shape[0] == B and q_rows.shape[2] == HID and q_rows.stride(2) == 1 and q_rows.dtype == kv_buf.dtype
    _, k, v = _attn_views(kv_buf)
    out = torch.empty(B, Lq, HID, dtype=kv_buf.dtype, device=kv_buf.device)
    lse = torch.empty(B, HEADS, Lq, dtype=torch.float32, device=kv_buf.device)
    assert keys.idx.shape[0] == B and keys.idx.is_contiguous() and keys.n_dec == Lq
    X.check(X.lib().t2s_attn_fwd(
        X.ptr(q_rows), X.ptr(k), X.ptr(v), X.ptr(out), X.ptr(lse), X.ptr(keys.idx), X.ptr(keys.cnt),
        B, HEADS, Lq, keys.idx.shape[1], keys.n_dec, 0,
        q_rows.stride(1), q_rows.stride(0), kv_buf.stride(1), kv_buf.stride(0), out.stride(1), out.stride(0),
        scale, X.dtype_code(kv_buf), 0.0, 0, X.stream()), "t2s_attn_fwd")
    return out


def _check_keys(keys, B, L):
    assert keys.idx.dtype == torch.int32 and keys.cnt.dtype == torch.int32
    assert keys.idx.shape[0] == B and keys.cnt.shape == (B,) and keys.idx.is_contiguous()
    assert 0 < keys.cap_hint <= keys.idx.shape[1]
    assert 0 <= keys.n_dec <= keys.idx.shape[1]
    assert keys.dec_q0 + keys.n_dec <= L or keys.n_dec == 0, "decoder rows must lie inside the sequence"


class NormRes:
    """A residual-stream value kept in NORMALISED form: ``LN(z)`` with the statistics and affine of the block that produced
    ``z``.  The next residual+LayerNorm kernel normalises it on the fly (t2s_add_layernorm_fwd_nres), so the fp32 stream
    value is never written to HBM between two blocks of a layer stack."""
    __slots__ = ("z", "stats", "gamma", "beta")

    def __init__(self, z, stats, gamma, beta):
        self.z, self.stats, self.gamma, self.beta = z, stats, gamma, beta

    @property
    def shape(self):
        return self.z.shape


def add_layernorm_fwd(x, res, gamma, beta, eps=1e-12, save=True, inplace_z=True, stream_dtype=None, want_lo=False,
                      want_y=True, drop_p=0.0, drop_seed=0):
    """y = LN(x + res).  x: GEMM-output dtype; res / y / z: residual-stream dtype (``stream_dtype``, default
    x.dtype).  Returns (y, y_lo, z, stats): y_lo = bf16 copy of y (if want_lo), z = x + res kept for backward
    (written over x when the dtypes match and inplace_z), stats [rows, 2] = (mean, rstd).  res may be None, a tensor, or
    a ``NormRes`` (the previous block's output in normalised form)."""
    rows = _rows768(x)
    sdt = stream_dtype or x.dtype
    nres = isinstance(res, NormRes)
    if nres:
        assert res.z.shape == x.shape and res.z.is_contiguous() and res.z.dtype == sdt and res.stats.shape == (rows, 2)
        assert res.gamma.dtype == torch.float32 and res.beta.dtype == torch.float32 and res.gamma.numel() == HID
    elif res is not None:
        assert res.shape == x.shape and res.is_contiguous() and res.dtype == sdt
    assert gamma.dtype == torch.float32 and beta.dtype == torch.float32 and gamma.numel() == HID
    assert want_y or want_lo
    y = torch.empty(x.shape, dtype=sdt, device=x.device) if want_y else None
    y_lo = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device) if want_lo else None
    z = None
    if save:
        z = x if (inplace_z and sdt == x.dtype) else torch.empty(x.shape, dtype=sdt, device=x.device)
    stats = torch.empty(rows, 2, dtype=torch.float32, device=x.device) if save else None
    tail = (X.ptr(gamma), X.ptr(beta), X.ptr(y), X.ptr(y_lo), X.ptr(z), X.ptr(stats), rows, eps, X.dtype_code(x),
            X.T2S_F32 if sdt == torch.float32 else X.T2S_BF16, float(drop_p), int(drop_seed), X.stream())
    if nres:
        X.check(X.lib().t2s_add_layernorm_fwd_nres(X.ptr(x), X.ptr(res.z), X.ptr(res.stats), X.ptr(res.gamma), X.ptr(res.beta), *tail),
                "t2s_add_layernorm_fwd_nres")
    else:
        X.check(X.lib().t2s_add_layernorm_fwd(X.ptr(x), X.ptr(res), *tail), "t2s_add_layernorm_fwd")
    return y, y_lo, z, stats


def add_layernorm_bwd(dy, z, stats, gamma, out_dtype=None, drop_p=0.0, drop_seed=0, want_bias=False):
    """Returns (dz [out_dtype, default dy.dtype], dzx, dgamma, dbeta); dzx = gradient of the dropped branch input
    (dz itself when drop_p == 0).  ``want_bias`` appends dbias [768] fp32 = column sums of dzx: the bias gradient of the
    dense layer that fed this block, accumulated in the same pass."""
    rows = _rows768(dy)
    assert z.shape == dy.shape and z.is_contiguous() and stats.shape == (rows, 2)
    parts = X.lib().t2s_layernorm_bwd_parts(rows)
    dz = torch.empty(dy.shape, dtype=out_dtype or dy.dtype, device=dy.device)
    dzx = torch.empty_like(dz) if drop_p > 0 else None
    part = torch.empty(3 if want_bias else 2, parts, HID, dtype=torch.float32, device=dy.device)
    head = (X.ptr(dy), X.ptr(z), X.ptr(stats), X.ptr(gamma), X.ptr(dz), X.ptr(dzx), X.ptr(part[0]), X.ptr(part[1]))
    tail = (rows, X.dtype_code(dy), X.dtype_code(z), X.dtype_code(dz), float(drop_p), int(drop_seed), X.stream())
    if want_bias:
        X.check(X.lib().t2s_add_layernorm_bwd_bias(*head, X.ptr(part[2]), *tail), "t2s_add_layernorm_bwd_bias")
    else:
        X.check(X.lib().t2s_add_layernorm_bwd(*head, *tail), "t2s_add_layernorm_bwd")
    sums = part.sum(1)                          # one reduction launch for all partial-sum tables
    out = (dz, (dzx if dzx is not None else dz), sums[0], sums[1])
    return out + (sums[2],) if want_bias else out


def dropout_mask(n, drop_p, drop_seed, device):
    """0/1 keep mask of the fused hidden dropout for element indices 0..n-1 (tests)."""
    out = torch.empty(n, dtype=torch.uint8, device=device)
    X.check(X.lib().t2s_dropout_mask(X.ptr(out), n, float(drop_p), int(drop_seed), X.stream()), "t2s_dropout_mask")
    return out


def gelu_fwd(u):
    assert u.is_contiguous() and u.numel() % 4 == 0
    y = torch.empty_like(u)
    X.check(X.lib().t2s_gelu_fwd(X.ptr(u), X.ptr(y), u.numel(), X.dtype_code(u), X.stream()), "t2s_gelu_fwd")
    return y


def gelu_bwd(dy, u):
    """Returns (du, dbias [cols] fp32)."""
    assert dy.is_contiguous() and u.is_contiguous() and dy.shape == u.shape and dy.dtype == u.dtype
    cols = u.shape[-1]
    rows = u.numel() // cols
    assert cols % 4 == 0
    parts = X.lib().t2s_gelu_bwd_parts(rows)
    du = torch.empty_like(u)
    dbp = torch.empty(parts, cols, dtype=torch.float32, device=u.device)
    X.check(X.lib().t2s_gelu_bwd(X.ptr(dy), X.ptr(u), X.ptr(du), X.ptr(dbp), rows, cols, X.dtype_code(u), X.stream()),
            "t2s_gelu_bwd")
    return du, dbp.sum(0)


def ptr_scores(q, k, mask01, out, col0, exact_fp32=False):
    """out[b, j, col0 + n] = q[b,j].k[b,n]/sqrt(768) + mask01[b,n] written in place into the logits buffer
    out [B, D, V+N] fp32.  q: [B, D, 768] fp32; k: [B, N, 768] bf16/fp32; mask01: [B, N] fp32."""
    B, D, _ = q.shape
    N = k.shape[1]
    assert q.dtype == torch.float32 and q.is_contiguous() and q.shape[2] == HID
    assert k.is_contiguous() and k.shape == (B, N, HID) and mask01.shape == (B, N) and mask01.dtype == torch.float32
    assert out.dtype == torch.float32 and out.is_contiguous() and out.shape[:2] == (B, D) and out.shape[2] >= col0 + N
    X.check(X.lib().t2s_ptr_scores(X.ptr(q), X.ptr(k), X.ptr(mask01.contiguous()), X.ptr(out), B, D, N, out.shape[2], col0,
                                   1.0 / math.sqrt(HID), X.dtype_code(k), 1 if exact_fp32 else 0, X.stream()), "t2s_ptr_scores")
    return out


def question_pool(qp, w, bias, qmask):
    """qp: [B, T, 768] fp32 projected question -> [B, 768] (t2s.py:453-459)."""
    B, T, _ = qp.shape
    assert qp.dtype == torch.float32 and qp.is_contiguous() and w.numel() == HID and qmask.shape == (B, T)
    out = torch.empty(B, HID, dtype=torch.float32, device=qp.device)
    X.check(X.lib().t2s_question_pool(X.ptr(qp), X.ptr(w.contiguous()), X.ptr(bias), X.ptr(qmask.float().contiguous()),
                                      X.ptr(out), B, T, X.stream()), "t2s_question_pool")
    return out


def _rows_view(k):
    """[B, M, 768] with contiguous rows; the samples may be slices of a longer sequence (batch stride >= M * 768)."""
    B, M, _ = k.shape
    assert k.shape[2] == HID and k.stride(2) == 1 and k.stride(1) == HID and (B == 1 or k.stride(0) >= M * HID), \
        "expected [B, M, 768] with dense rows, got strides %s" % (k.stride(),)
    return k.stride(0) if B > 1 else M * HID


def attention_score(q, k, mask):
    """q: [B, 768] fp32, k: [B, M, 768] (a row slice of a longer [B, L, 768] buffer is fine), mask: [B, M] fp32 -> [B, M] fp32
    (spatio_temporal_grounding.py:15-23)."""
    B, M, _ = k.shape
    assert q.shape == (B, HID) and q.dtype == torch.float32 and q.is_contiguous()
    assert mask.shape == (B, M) and mask.dtype == torch.float32 and mask.is_contiguous()
    score = torch.empty(B, M, dtype=torch.float32, device=k.device)
    X.check(X.lib().t2s_attention_score(X.ptr(q), X.ptr(k), _rows_view(k), X.ptr(mask), X.ptr(score), B, M, X.dtype_code(k), X.stream()),
            "t2s_attention_score")
    return score


def ocr_tail_fwd(a, bbox, w_box, b_box, ga, ba, gb, bb, drop_p=0.0, drop_seed=0, eps=1e-12):
    """dropout(LN(a; ga, ba) + LN(bbox @ w_box^T + b_box; gb, bb)): a [rows, 768] (fp32 / bf16), bbox [rows, 4] fp32 ->
    (out [rows, 768] fp32, stats [rows, 4])  (T2S._forward_ocr_encoding t2s.py:221-258)."""
    rows = _rows768(a)
    assert bbox.shape == (rows, 4) and bbox.dtype == torch.float32 and bbox.is_contiguous()
    assert w_box.shape == (HID, 4) and w_box.is_contiguous() and b_box.shape == (HID,)
    for t in (w_box, b_box, ga, ba, gb, bb):
        assert t.dtype == torch.float32 and t.is_contiguous()
    out = torch.empty(rows, HID, dtype=torch.float32, device=a.device)
    stats = torch.empty(rows, 4, dtype=torch.float32, device=a.device)
    X.check(X.lib().t2s_ocr_tail_fwd(X.ptr(a), X.dtype_code(a), X.ptr(bbox), X.ptr(w_box), X.ptr(b_box), X.ptr(ga), X.ptr(ba), X.ptr(gb), X.ptr(bb),
                                     X.ptr(out), X.ptr(stats), rows, float(eps), float(drop_p), int(drop_seed), X.stream()), "t2s_ocr_tail_fwd")
    return out, stats


def ocr_tail_bwd(g_out, a, bbox, w_box, b_box, ga, gb, stats, drop_p=0.0, drop_seed=0):
    """-> (d_a in a's dtype, dga, dba, dgb, dbb, db_box [768], dw_box [768, 4]) (all parameter gradients fp32)."""
    rows = _rows768(a)
    assert g_out.shape == (rows, HID) and g_out.dtype == torch.float32 and g_out.is_contiguous() and stats.shape == (rows, 4)
    d_a = torch.empty_like(a)
    n_part = X.lib().t2s_ocr_tail_parts(rows)
    part = torch.empty(n_part, 9, HID, dtype=torch.float32, device=a.device)
    X.check(X.lib().t2s_ocr_tail_bwd(X.ptr(g_out), X.ptr(a), X.dtype_code(a), X.ptr(bbox), X.ptr(w_box), X.ptr(b_box), X.ptr(ga), X.ptr(gb),
                                     X.ptr(stats), X.ptr(d_a), X.ptr(part), rows, float(drop_p), int(drop_seed), X.stream()), "t2s_ocr_tail_bwd")
    p = part.sum(0)
    return d_a, p[0], p[1], p[2], p[3], p[4], p[5:9].t().contiguous()


def tanh_residual_fwd(x, enc_out):
    """x + tanh(enc_out): [B, L, 768] fp32 contiguous (QTV's residual, t2s.py:428-432)."""
    B, L, _ = x.shape
    assert x.shape == enc_out.shape and x.shape[2] == HID and x.dtype == enc_out.dtype == torch.float32 and x.is_contiguous() and enc_out.is_contiguous()
    y = torch.empty_like(x)
    X.check(X.lib().t2s_tanh_residual_fwd(X.ptr(x), X.ptr(enc_out), X.ptr(y), B, L, L * HID, X.stream()), "t2s_tanh_residual_fwd")
    return y


def tanh_residual_bwd(gy, enc_out, out_dtype):
    """gy * (1 - tanh(enc_out)^2) -> [B, L, 768] in out_dtype; gy fp32, possibly a row slice of a longer buffer."""
    B, L, _ = enc_out.shape
    assert gy.shape == enc_out.shape and gy.dtype == torch.float32 and enc_out.dtype == torch.float32 and enc_out.is_contiguous()
    g = torch.empty(B, L, HID, dtype=out_dtype, device=gy.device)
    X.check(X.lib().t2s_tanh_residual_bwd(X.ptr(gy), _rows_view(gy), X.ptr(enc_out), X.ptr(g), X.dtype_code(g), B, L, X.stream()),
            "t2s_tanh_residual_bwd")
    return g


def add_cast(a, b):
    """a (fp32 [B, L, 768], possibly a row slice of a longer buffer) + b (contiguous, fp32 or bf16) -> fp32 contiguous."""
    B, L, _ = a.shape
    assert a.dtype == torch.float32 and b.numel() == a.numel() and b.is_contiguous()
    out = torch.empty(B, L, HID, dtype=torch.float32, device=a.device)
    X.check(X.lib().t2s_add_cast(X.ptr(a), _rows_view(a), X.ptr(b), X.dtype_code(b), X.ptr(out), B, L, X.stream()), "t2s_add_cast")
    return out


def ground_select(frame_score, frame_mask, expo_frame, frame_id, q_global, ocr_feat, expo_ocr, temporal_id, bbox,
                  F, P, frame_topk, ocr_topk):
    """Temporal + spatial grounding selection (see include/t2s_hip.h).  Returns a dict of masks / outputs.
    With fewer than ``ocr_topk`` OCR slots per frame the reference's slice ``sorted[:, :, :o_topk]``
    (spatio_temporal_grounding.py:104,112) simply takes all P of them, so ``ground_box`` is [B, F * min(P, ocr_topk), 4];
    ``frame_topk > F`` fails there too (``torch.topk``) and is refused by the kernel."""
    B = frame_score.shape[0]
    N = F * P
    ocr_topk = min(int(ocr_topk), int(P))
    dev = frame_score.device
    f32 = dict(dtype=torch.float32, device=dev)
    assert frame_score.shape == (B, F) and frame_mask.shape == (B, F) and expo_frame.shape == (B, 2, F)
    assert expo_ocr.shape == (B, 2, N) and ocr_feat.shape == (B, N, HID) and bbox.shape == (B, N, 4)
    assert frame_id.dtype == torch.int64 and temporal_id.dtype == torch.int64 and temporal_id.shape == (B, N)
    for t in (frame_score, frame_mask, expo_frame, expo_ocr, q_global, bbox):
        assert t.dtype == torch.float32 and t.is_contiguous()
    assert frame_id.is_contiguous() and temporal_id.is_contiguous()
    o = dict(pos_obj_mask=torch.empty(B, F, **f32), neg_obj_mask=torch.empty(B, F, **f32),
             ground_frame=torch.empty(B, frame_topk, dtype=torch.int64, device=dev),
             new_ocr_mask=torch.empty(B, N, **f32), ocr_score=torch.empty(B, N, **f32),
             pos_ocr_mask=torch.empty(B, N, **f32), neg_ocr_mask=torch.empty(B, N, **f32),
             ground_box=torch.empty(B, F * ocr_topk, 4, **f32))
    X.check(X.lib().t2s_ground_select(
        X.ptr(frame_score), X.ptr(frame_mask), X.ptr(expo_frame), X.ptr(frame_id), X.ptr(q_global), X.ptr(ocr_feat),
        _rows_view(ocr_feat), X.dtype_code(ocr_feat), X.ptr(expo_ocr), X.ptr(temporal_id), X.ptr(bbox), X.ptr(o["pos_obj_mask"]),
        X.ptr(o["neg_obj_mask"]), X.ptr(o["ground_frame"]), X.ptr(o["new_ocr_mask"]), X.ptr(o["ocr_score"]),
        X.ptr(o["pos_ocr_mask"]), X.ptr(o["neg_ocr_mask"]), X.ptr(o["ground_box"]), B, F, P, frame_topk, ocr_topk,
        X.stream()), "t2s_ground_select")
    return o


def embed_rows(f0, f1, id0, emb0, id1, emb1, out_dtype):
    """[ L2norm(f0) | L2norm(f1) | emb0[id0] | emb1[id1] ] -> [..., d0 + d1 + 50k] in out_dtype (t2s.py:192-258)."""
    lead = f0.shape[:-1]
    rows = f0.numel() // f0.shape[-1]
    d0 = f0.shape[-1]
    d1 = f1.shape[-1] if f1 is not None else 0
    edim = emb0.shape[1] if emb0 is not None else 0
    width = d0 + d1 + (edim if id0 is not None else 0) + (edim if id1 is not None else 0)
    assert f0.dtype == torch.float32 and f0.is_contiguous() and (f1 is None or (f1.dtype == torch.float32 and f1.is_contiguous()))
    for ids, emb in ((id0, emb0), (id1, emb1)):
        if ids is not None:
            assert ids.dtype == torch.int64 and ids.is_contiguous() and ids.numel() == rows
            assert emb.dtype == torch.float32 and emb.is_contiguous() and emb.shape[1] == edim
    out = torch.empty(*lead, width, dtype=out_dtype, device=f0.device)
    X.check(X.lib().t2s_embed_rows(X.ptr(f0), d0, X.ptr(f1), d1, X.ptr(id0), X.ptr(emb0), X.ptr(id1), X.ptr(emb1), edim,
                                   emb0.shape[0] if emb0 is not None else 1, X.ptr(out), width, rows, X.dtype_code(out),
                                   X.stream()), "t2s_embed_rows")
    return out


def phoc(tokens, out=None):
    """tokens: uint8 [..., width] NUL-padded normalised OCR tokens on the GPU -> fp32 [..., 604] PHOC features
    (include/t2s_hip.h: t2s_phoc).  ``out`` may be a preallocated [..., >=604]-strided fp32 view (e.g. the arena field
    the model reads as context_feature_1)."""
    assert tokens.dtype == torch.uint8 and tokens.is_contiguous() and tokens.is_cuda
    lead, width = tokens.shape[:-1], tokens.shape[-1]
    n = tokens.numel() // width
    if out is None:
        out = torch.empty(*lead, 604, dtype=torch.float32, device=tokens.device)
    assert out.dtype == torch.float32 and out.shape[-1] == 604 and out.numel() == n * 604 and out.stride(-1) == 1
    rs = out.stride(-2) if out.dim() > 1 else 604
    assert out.dim() <= 1 or out.view(-1, 604).stride(0) == rs, "rows of `out` must be evenly strided"
    X.check(X.lib().t2s_phoc(X.ptr(tokens), n, width, X.ptr(out), rs, X.stream()), "t2s_phoc")
    return out


def bce_masked(scores, targets, row_mask):
    """Returns (row_loss [rows], grad [rows, cols]) -- see include/t2s_hip.h."""
    cols = scores.shape[-1]
    rows = scores.numel() // cols
    assert scores.dtype == torch.float32 and targets.dtype == torch.float32 and scores.is_contiguous() and targets.is_contiguous()
    assert targets.shape == scores.shape and row_mask.numel() == rows and row_mask.dtype == torch.float32
    row_loss = torch.empty(rows, dtype=torch.float32, device=scores.device)
    grad = torch.empty_like(scores)
    X.check(X.lib().t2s_bce_masked(X.ptr(scores), X.ptr(targets), X.ptr(row_mask.contiguous()), X.ptr(row_loss), X.ptr(grad),
                                   rows, cols, X.stream()), "t2s_bce_masked")
    return row_loss, grad


def infonce_stats(q, p, n):
    cols = q.shape[-1]
    rows = q.numel() // cols
    for t in (q, p, n):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.shape == q.shape
    stats = torch.empty(rows, 5, dtype=torch.float32, device=q.device)
    X.check(X.lib().t2s_infonce_stats(X.ptr(q), X.ptr(p), X.ptr(n), X.ptr(stats), rows, cols, X.stream()), "t2s_infonce_stats")
    return stats


def infonce_bwd(q, p, n, gstats):
    cols = q.shape[-1]
    rows = q.numel() // cols
    assert gstats.shape == (rows, 5) and gstats.dtype == torch.float32 and gstats.is_contiguous()
    dq, dp, dn = torch.empty_like(q), torch.empty_like(p), torch.empty_like(n)
    X.check(X.lib().t2s_infonce_bwd(X.ptr(q), X.ptr(p), X.ptr(n), X.ptr(gstats), X.ptr(dq), X.ptr(dp), X.ptr(dn), rows, cols,
                                    X.stream()), "t2s_infonce_bwd")
    return dq, dp, dn

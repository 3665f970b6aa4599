"""Per-frame ViT-L CLS features (``video_feat`` rows) on the GPU.

Counterpart of the reference's offline script ``tools/video_feat/obtain_vit_feat.py:37-53``: Hugging Face
``ViTImageProcessor`` + ``ViTModel('vit-large-patch16-224-in21k')``, ``last_hidden_state[:, 0, :]`` saved as one ``[1, 1024]``
``.npy`` per frame (the format ``dataset.py:267-282`` reads).  Here the encoder is run with this build's kernels: the
self-attention of every layer goes through the flash-style MFMA kernel of the T2S path (head_dim 64: ViT-L's 16 x 64 layout
fits it as is; dense key list), the MLP activation through the exact-erf GELU kernel, the GEMMs through the library; frames
are batched (the script feeds one image per call).  Weights come from a Hugging Face ViT state_dict (same key names).

Architecture restated from the model definition the script loads (``transformers`` ``modeling_vit.py``): patch embedding
(16 x 16 convolution = a GEMM over unfolded patches) + [CLS] + learned position embeddings; ``num_hidden_layers`` PRE-LayerNorm
blocks ``x += attn(LN(x)); x += mlp(LN(x))`` with GELU(erf); final LayerNorm; layer_norm_eps from the config (1e-12)."""
import os

import numpy as np
import torch

from . import hipext as X
from . import ops


def preprocess(images, size=224):
    """``ViTImageProcessor`` defaults (preprocessor_config of vit-large-patch16-224-in21k): resize to size x size (PIL bilinear),
    rescale by 1/255, normalise with mean = std = 0.5.  images: PIL images or HxWx3 uint8 arrays -> [B, 3, size, size] fp32."""
    from PIL import Image
    out = []
    for im in images:
        if not isinstance(im, Image.Image):
            im = Image.fromarray(np.asarray(im, dtype=np.uint8))
        im = im.convert("RGB").resize((size, size), resample=Image.BILINEAR)
        a = np.asarray(im, dtype=np.float32) * (1.0 / 255.0)
        out.append(((a - 0.5) / 0.5).transpose(2, 0, 1))
    return torch.from_numpy(np.stack(out))


def add_layernorm(h, branch, bias, gamma, beta, eps, out_dtype):
    """One pass over the rows of the fp32 stream h [..., W] (updated IN PLACE): h += branch + bias (both optional), returns
    LN(h) * gamma + beta as out_dtype (t2s_wide_add_layernorm_fwd; W a multiple of 4, <= 1280).  h may be a strided row view (the CLS rows)."""
    W = h.shape[-1]
    rows = h.numel() // W
    assert h.dtype == torch.float32 and h.is_cuda and h.stride(-1) == 1
    stride = W if h.dim() == 1 or h.is_contiguous() else h.stride(-2)
    if not h.is_contiguous():
        assert h.dim() == 2, "a non-contiguous stream must be a [rows, W] view with one row stride"
    if branch is not None:
        assert branch.dtype in (torch.bfloat16, torch.float32) and branch.is_contiguous() and branch.numel() == rows * W
    y = torch.empty(h.shape, dtype=out_dtype, device=h.device)
    X.check(X.lib().t2s_wide_add_layernorm_fwd(X.ptr(h), stride, X.ptr(branch) if branch is not None else None,
                                               X.dtype_code(branch) if branch is not None else 0, X.ptr(bias) if bias is not None else None, X.ptr(gamma), X.ptr(beta), X.ptr(y),
                                               X.dtype_code(y), rows, W, float(eps), X.stream()), "t2s_wide_add_layernorm_fwd")
    return y


def attention_dense(qkv, n_heads):
    """softmax(Q K^T / 8) V over a fused [B, L, 3 * n_heads * 64] projection, every key visible (t2s_attn_fwd with a dense list)."""
    B, L, W = qkv.shape
    hid = n_heads * 64
    assert W == 3 * hid and qkv.is_contiguous() and qkv.is_cuda
    out = torch.empty(B, L, hid, dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty(B, n_heads, L, dtype=torch.float32, device=qkv.device)
    q, k, v = qkv[..., :hid], qkv[..., hid:2 * hid], qkv[..., 2 * hid:]
    X.check(X.lib().t2s_attn_fwd(X.ptr(q), X.ptr(k), X.ptr(v), X.ptr(out), X.ptr(lse), None, None, B, n_heads, L, L, 0, 0,
                                 qkv.stride(1), qkv.stride(0), qkv.stride(1), qkv.stride(0), out.stride(1), out.stride(0),
                                 0.125, X.dtype_code(qkv), 0.0, 0, X.stream()), "t2s_attn_fwd")
    return out


_RENAMES = (("layers.", "encoder.layer."), (".attention.q_proj.", ".attention.attention.query."), (".attention.k_proj.", ".attention.attention.key."),
            (".attention.v_proj.", ".attention.attention.value."), (".attention.o_proj.", ".attention.output.dense."),
            (".mlp.fc1.", ".intermediate.dense."), (".mlp.fc2.", ".output.dense."))


def _classic_key(k):
    """Checkpoint key spellings: the published vit-large-patch16-224-in21k files use ``encoder.layer.N.attention.attention.query`` ...;
    recent ``transformers`` releases save ``layers.N.attention.q_proj`` / ``mlp.fc1`` ...; an optional ``vit.`` prefix.  -> the former."""
    if k.startswith("vit."):
        k = k[4:]
    if k.startswith("layers."):
        for a, b in _RENAMES:
            k = k.replace(a, b, 1) if a != "layers." else (b + k[len(a):] if k.startswith(a) else k)
    return k


class ViTFeatureExtractor:
    def __init__(self, state_dict, hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096,
                 image_size=224, patch_size=16, layer_norm_eps=1e-12, device="cuda:0", dtype=torch.bfloat16):
        if hidden_size != num_attention_heads * 64:
            raise ValueError("the attention kernels are built for head_dim 64 (ViT-L: 16 x 64)")
        self.h, self.nl, self.nh, self.ffn = hidden_size, num_hidden_layers, num_attention_heads, intermediate_size
        self.img, self.patch, self.eps, self.dt, self.dev = image_size, patch_size, layer_norm_eps, dtype, torch.device(device)
        sd = {_classic_key(k): v for k, v in state_dict.items()}
        g = lambda k: sd[k].detach().to(self.dev, torch.float32)
        dt = dtype
        self.cls = g("embeddings.cls_token")
        self.pos = g("embeddings.position_embeddings")
        self.w_patch = g("embeddings.patch_embeddings.projection.weight").reshape(hidden_size, -1).to(dt)       # [hid, 3 * p * p]
        self.b_patch = g("embeddings.patch_embeddings.projection.bias")
        self.layers = []
        for i in range(num_hidden_layers):
            p = "encoder.layer.%d." % i
            a = p + "attention.attention."
            self.layers.append(dict(
                ln1=(g(p + "layernorm_before.weight"), g(p + "layernorm_before.bias")),
                w_qkv=torch.cat([g(a + "query.weight"), g(a + "key.weight"), g(a + "value.weight")], 0).to(dt),
                b_qkv=torch.cat([g(a + "query.bias"), g(a + "key.bias"), g(a + "value.bias")], 0).to(dt),
                w_ao=g(p + "attention.output.dense.weight").to(dt), b_ao=g(p + "attention.output.dense.bias"),
                ln2=(g(p + "layernorm_after.weight"), g(p + "layernorm_after.bias")),
                w_i=g(p + "intermediate.dense.weight").to(dt), b_i=g(p + "intermediate.dense.bias").to(dt),
                w_o=g(p + "output.dense.weight").to(dt), b_o=g(p + "output.dense.bias")))
        self.ln_f = (g("layernorm.weight"), g("layernorm.bias"))

    @classmethod
    def from_pretrained(cls, path, **kw):
        """A Hugging Face ViT checkpoint directory (config.json + pytorch_model.bin / model.safetensors)."""
        import json
        cfg = json.load(open(os.path.join(path, "config.json")))
        f = os.path.join(path, "model.safetensors")
        if os.path.isfile(f):
            from safetensors.torch import load_file
            sd = load_file(f)
        else:
            sd = torch.load(os.path.join(path, "pytorch_model.bin"), map_location="cpu", weights_only=True)
        keys = ("hidden_size", "num_hidden_layers", "num_attention_heads", "intermediate_size", "image_size", "patch_size", "layer_norm_eps")
        return cls(sd, **{k: cfg[k] for k in keys if k in cfg}, **kw)

    @torch.no_grad()
    def __call__(self, pixel_values):
        """pixel_values [B, 3, S, S] fp32 (``preprocess``) -> CLS features [B, hidden] fp32 (``last_hidden_state[:, 0, :]``)."""
        x = pixel_values.to(self.dev, torch.float32)
        B, P, dt = x.shape[0], self.patch, self.dt
        # 16 x 16 / stride-16 convolution as a GEMM: patches in row-major (h, w) order, channel-major inside a patch
        patches = x.unfold(2, P, P).unfold(3, P, P).permute(0, 2, 3, 1, 4, 5).reshape(B, -1, 3 * P * P)
        h = (patches.to(dt) @ self.w_patch.t()).float() + self.b_patch
        h = torch.cat([self.cls.expand(B, -1, -1), h], 1) + self.pos
        L = h.shape[1]
        h = h.contiguous()
        # pre-LN blocks: the residual update of one block and the LayerNorm in front of the next are ONE pass over the stream
        y = add_layernorm(h, None, None, self.layers[0]["ln1"][0], self.layers[0]["ln1"][1], self.eps, dt)
        for i, ly in enumerate(self.layers):
            qkv = torch.addmm(ly["b_qkv"], y.view(B * L, self.h), ly["w_qkv"].t()).view(B, L, 3 * self.h)
            att = attention_dense(qkv.contiguous(), self.nh)
            y = add_layernorm(h, att.view(B * L, self.h) @ ly["w_ao"].t(), ly["b_ao"], ly["ln2"][0], ly["ln2"][1], self.eps, dt)
            u = ops.gelu_fwd(torch.addmm(ly["b_i"], y.view(B * L, self.h), ly["w_i"].t()))
            if i + 1 < self.nl:
                nxt = self.layers[i + 1]["ln1"]
                y = add_layernorm(h, u @ ly["w_o"].t(), ly["b_o"], nxt[0], nxt[1], self.eps, dt)
            else:
                # last block: only the CLS rows are read (last_hidden_state[:, 0, :]); their residual update + the final LayerNorm
                cls_branch = (u.view(B, L, self.ffn)[:, 0].contiguous() @ ly["w_o"].t())
                return add_layernorm(h[:, 0], cls_branch, ly["b_o"], self.ln_f[0], self.ln_f[1], self.eps, torch.float32)


def extract_video_features(model, frames_dir, out_dir, batch=64, size=224):
    """The loop of obtain_vit_feat.py:24-53 for one video: every frame image of ``frames_dir`` -> ``out_dir/<frame>.npy`` holding
    the ``[1, hidden]`` fp32 CLS row (existing files are kept, as the script does)."""
    from PIL import Image
    os.makedirs(out_dir, exist_ok=True)
    names = [n for n in sorted(os.listdir(frames_dir)) if not os.path.exists(os.path.join(out_dir, n.split(".")[0] + ".npy"))]
    for i in range(0, len(names), batch):
        chunk = names[i:i + batch]
        feats = model(preprocess([Image.open(os.path.join(frames_dir, n)) for n in chunk], size)).cpu().numpy()
        for n, f in zip(chunk, feats):
            np.save(os.path.join(out_dir, n.split(".")[0] + ".npy"), f[None, :].astype(np.float32))
    return len(names)

"""The ViT frame-feature producer (vitxt_gqa_amd/vit.py; reference: tools/video_feat/obtain_vit_feat.py:37-53) against the model
the reference script runs - Hugging Face ``ViTModel`` - with the same (random) weights: a tiny configuration and a ViT-L-shaped
one (1024 = 16 x 64, 224 x 224 input, 197 tokens; 2 layers), fp32 and bf16 operands; the image preprocessing against
``ViTImageProcessor``; the per-frame ``[1, hidden]`` .npy files the dataset reader expects (dataset.py:267-282)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


@pytest.mark.parametrize("width", [128, 768, 1024, 1280])
def test_wide_add_layernorm_kernel(width):
    """t2s_wide_add_layernorm_fwd (the pre-LN block's residual update + next LayerNorm in one pass) against torch: the updated stream
    bit for bit (same fp32 adds in the same order), the normalised rows within fp32 round-off / one bf16 ulp; plain LayerNorm form
    (no branch) leaves the stream untouched; strided rows (the CLS rows of a [B, L, W] stream)."""
    _need_gpu()
    import torch.nn.functional as F
    from vitxt_gqa_amd.vit import add_layernorm
    torch.manual_seed(width)
    rows = 1001
    h = torch.randn(rows, width, device="cuda") * 3 + 0.5
    br = torch.randn(rows, width, device="cuda").to(torch.bfloat16)
    bias, g, b = (torch.randn(width, device="cuda") for _ in range(3))
    want_h = h + br.float() + bias
    want = F.layer_norm(want_h, (width,), g, b, 1e-12)
    h1 = h.clone()
    y32 = add_layernorm(h1, br, bias, g, b, 1e-12, torch.float32)
    assert torch.equal(h1, want_h)
    assert (y32 - want).abs().max().item() < 2e-5
    h2 = h.clone()
    y16 = add_layernorm(h2, br, bias, g, b, 1e-12, torch.bfloat16)
    assert torch.equal(h2, want_h) and y16.dtype == torch.bfloat16
    assert (y16.float() - want).abs().max().item() <= 2 ** -7 * want.abs().max().item()
    h3 = h.clone()
    y = add_layernorm(h3, None, None, g, b, 1e-12, torch.float32)
    assert torch.equal(h3, h) and (y - F.layer_norm(h, (width,), g, b, 1e-12)).abs().max().item() < 2e-5
    # strided rows: row 0 of every sample of a [B, L, W] stream; the other rows stay as they were
    s3 = torch.randn(5, 7, width, device="cuda")
    keep = s3.clone()
    brc = torch.randn(5, width, device="cuda").to(torch.bfloat16)
    yc = add_layernorm(s3[:, 0], brc, bias, g, b, 1e-12, torch.float32)
    assert torch.equal(s3[:, 1:], keep[:, 1:]) and torch.equal(s3[:, 0], keep[:, 0] + brc.float() + bias)
    assert (yc - F.layer_norm(s3[:, 0], (width,), g, b, 1e-12)).abs().max().item() < 2e-5
    # an fp32 branch (the fp32 operand mode of the producer)
    h4 = h.clone()
    y4 = add_layernorm(h4, br.float(), bias, g, b, 1e-12, torch.float32)
    assert torch.equal(h4, want_h) and torch.equal(y4, y32)
    with pytest.raises(RuntimeError, match="width"):
        add_layernorm(torch.zeros(4, 2048, device="cuda"), None, None, torch.ones(2048, device="cuda"), torch.zeros(2048, device="cuda"), 1e-12, torch.float32)


@pytest.mark.parametrize("hidden,heads,layers,ffn,img", [(128, 2, 2, 256, 32), (1024, 16, 2, 4096, 224)])
def test_cls_features_match_hf_vit(hidden, heads, layers, ffn, img):
    _need_gpu()
    from transformers import ViTConfig, ViTModel
    from vitxt_gqa_amd.vit import ViTFeatureExtractor
    torch.manual_seed(0)
    cfg = ViTConfig(hidden_size=hidden, num_hidden_layers=layers, num_attention_heads=heads, intermediate_size=ffn, image_size=img,
                    patch_size=16, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    ref = ViTModel(cfg).eval()
    with torch.no_grad():                      # spread the weights a little: the default init gives near-uniform attention
        for n, p in ref.named_parameters():
            if p.dim() > 1:
                p.mul_(3.0)
    x = torch.randn(3, 3, img, img)
    with torch.no_grad():
        want = ref(pixel_values=x).last_hidden_state[:, 0, :]
    kw = dict(hidden_size=hidden, num_hidden_layers=layers, num_attention_heads=heads, intermediate_size=ffn, image_size=img, patch_size=16,
              layer_norm_eps=cfg.layer_norm_eps)
    got32 = ViTFeatureExtractor(ref.state_dict(), dtype=torch.float32, **kw)(x).cpu()
    assert got32.shape == want.shape
    assert (got32 - want).abs().max().item() < 2e-3, (got32 - want).abs().max().item()
    got16 = ViTFeatureExtractor(ref.state_dict(), dtype=torch.bfloat16, **kw)(x).cpu()
    e16 = (got16 - want).abs().max().item()
    r16 = (got16 - want).pow(2).mean().sqrt().item()
    print("ViT %d x %d heads, bf16 operands vs transformers.ViTModel fp32: max abs err %.4f, RMS %.5f, |want| max %.2f (fp32 mode %.2e)" % (
        hidden, heads, e16, r16, want.abs().max().item(), (got32 - want).abs().max().item()))
    # bf16 operands through two 1024-wide pre-LN blocks: the maximum over the 3 x 1024 CLS features measured 0.058 with the round-5
    # forward (fp32 row sums on the vector pipe) and 0.061 with round 6's (row sums of the ROUNDED probabilities on the matrix pipe: the
    # weights of a row then sum to exactly one) - an extreme value that moves by a few % with any change of rounding; RMS bounded too
    assert e16 < 7e-2 and r16 < 2e-2, (e16, r16)


def test_preprocess_and_npy_files(tmp_path):
    _need_gpu()
    from PIL import Image
    from transformers import ViTConfig, ViTImageProcessor, ViTModel
    from vitxt_gqa_amd.vit import ViTFeatureExtractor, extract_video_features, preprocess
    rng = np.random.default_rng(0)
    frames = tmp_path / "frames" / "7"
    os.makedirs(frames)
    imgs = []
    for i in range(5):
        a = rng.integers(0, 256, size=(90 + 7 * i, 160, 3), dtype=np.uint8)
        Image.fromarray(a).save(str(frames / ("%06d.png" % i)))
        imgs.append(Image.open(str(frames / ("%06d.png" % i))))
    proc = ViTImageProcessor(size={"height": 32, "width": 32}, image_mean=[0.5, 0.5, 0.5], image_std=[0.5, 0.5, 0.5], resample=2)   # vit-large-patch16-224-in21k defaults at 32 px
    want = proc(images=imgs, return_tensors="pt")["pixel_values"]
    got = preprocess(imgs, size=32)
    assert (got - want).abs().max().item() < 1e-6
    cfg = ViTConfig(hidden_size=128, num_hidden_layers=1, num_attention_heads=2, intermediate_size=256, image_size=32, patch_size=16)
    torch.manual_seed(1)
    ref = ViTModel(cfg).eval()
    model = ViTFeatureExtractor(ref.state_dict(), hidden_size=128, num_hidden_layers=1, num_attention_heads=2, intermediate_size=256,
                                image_size=32, patch_size=16, layer_norm_eps=cfg.layer_norm_eps, dtype=torch.float32)
    out_dir = tmp_path / "feat" / "7"
    assert extract_video_features(model, str(frames), str(out_dir), batch=2, size=32) == 5
    with torch.no_grad():
        cls = ref(pixel_values=want).last_hidden_state[:, 0, :].numpy()
    for i in range(5):
        f = np.load(str(out_dir / ("%06d.npy" % i)))
        assert f.shape == (1, 128) and f.dtype == np.float32 and np.abs(f[0] - cls[i]).max() < 2e-3
    assert extract_video_features(model, str(frames), str(out_dir), batch=2, size=32) == 0         # existing features are kept

#!/usr/bin/env python3
"""The REFERENCE's own bf16 deviation: how far pythia.models.t2s.T2S under ``torch.autocast("cpu", dtype=torch.bfloat16)`` lands from its
own fp32 outputs on the full-length fixtures - the yardstick for "bf16 within tolerance" where the north star's 1e-2 is not reachable by
ANY bf16-operand implementation (peaky attention: tests/golden/full_peaky_b2_f100_p100).

Runs ONLY in the build container (imports /root/reference through the shim of make_golden.py).  For each fixture: the reference with the
fixture's weights, inputs, noise and (injected) selection masks, train mode, dropout 0; once in fp32 (must reproduce the fixture: checked)
and once under bf16 autocast (every nn.Linear / matmul / SDPA in bf16, LayerNorm / softmax / losses in fp32 - torch's autocast policy),
forward + both losses + backward.  Writes tests/golden/bf16_floor.json: per pass the max and RMS logit deviation (vocabulary / pointer
logits), the worst and total gradient-norm deviation and the per-parameter relative deviations.  Data only; no reference source is stored.

    python tests/golden/make_bf16_floor.py [fixture ...]
"""
import json
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)

import make_golden as MG  # noqa: E402


def run(case):
    from golden_util import Fixture
    from pythia.modules.losses import InfoNCE, POSBCEWithMaskLoss
    import pythia.models.t2s as t2s_mod
    fx = Fixture(case)
    m = MG.build_reference(fx.F, fx.P, fx.V, fx.meta["text_vocab"])
    m.load_state_dict(fx.state_dict())
    m.train()
    sl = MG.AD(fx.batch())
    sl.dataset_name, sl.dataset_type = "vtextgqa", "train"
    masks = fx.masks()
    orig = t2s_mod.Grounding_Module.forward

    def wrap(self, sample_list, fwd_results):
        r = orig(self, sample_list, fwd_results)
        for k, v in masks.items():          # the fixture's own selections (under autocast a near-tie selection may flip)
            fwd_results[k] = v.clone()
        return r

    t2s_mod.Grounding_Module.forward = wrap
    out_rec = {}
    try:
        for mode in ("fp32", "bf16"):
            m.zero_grad(set_to_none=True)
            torch.manual_seed(fx.meta["noise_seed"])
            t0 = time.time()
            if mode == "bf16":
                with torch.autocast("cpu", dtype=torch.bfloat16):
                    out = m.forward(sl)
            else:
                out = m.forward(sl)
            scores = {k: out[k].float() for k in ("ref_scores", "pos_scores", "neg_scores")}
            loss = 1.0 * POSBCEWithMaskLoss()(sl, scores) + 1000.0 * InfoNCE()(sl, scores)
            loss.backward()
            rec = {"seconds": time.time() - t0, "loss": float(loss)}
            for k, v in scores.items():
                e = (v.detach() - fx[k]).abs()
                rec[k] = {"vocab_max": float(e[..., :fx.V].max()), "ptr_max": float(e[..., fx.V:].max()),
                          "vocab_rms": float(e[..., :fx.V].pow(2).mean().sqrt()), "ptr_rms": float(e[..., fx.V:].pow(2).mean().sqrt()),
                          "argmax_flips": int((v.argmax(-1) != fx[k].argmax(-1)).sum())}
            names, ref = fx.meta["grad_names"], fx["grad_norms"].tolist()
            params = dict(m.named_parameters())
            total = fx["grad_total_norm"].item()
            rel, sq = {}, 0.0
            for n, r in zip(names, ref):
                g = float(params[n].grad.double().norm())
                sq += g * g
                rel[n] = abs(g - r) / (r + 1e-6 * total)
            live = {n: v for n, v in rel.items() if not n.endswith("attention.self.key.bias")}      # (mathematically zero gradients: noise both sides)
            # the gradient SLICES the fixture stores element by element: relative L2 distance under autocast
            slices = {}
            for k, v in fx.arr.items():
                if k.startswith("grad:"):
                    n = k[5:]
                    g = params[n[:-1].split("[:")[0]].grad[:int(n[:-1].split("[:")[1])] if n.endswith("]") else params[n].grad
                    slices[n] = float((g.float() - v).norm() / (v.norm() + 1e-12))
            rec["grad_slices_rel_l2"] = slices
            rec["grad"] = {"worst_rel": max(live.values()), "worst_name": max(live, key=live.get), "total_rel": abs(sq ** 0.5 - total) / total,
                           "key_bias_max_over_total": max(float(params[n].grad.double().norm()) for n in names if n.endswith("attention.self.key.bias")) / total,
                           "rel": rel}
            print(case, mode, "%.0f s" % rec["seconds"], {k: (rec[k]["vocab_max"], rec[k]["ptr_max"]) for k in scores}, "grad worst",
                  rec["grad"]["worst_rel"], rec["grad"]["worst_name"], "total", rec["grad"]["total_rel"], flush=True)
            out_rec[mode] = rec
    finally:
        t2s_mod.Grounding_Module.forward = orig
    assert max(out_rec["fp32"][k]["vocab_max"] for k in ("ref_scores", "pos_scores", "neg_scores")) < 1e-5, "the fp32 run must reproduce the fixture"
    return {"autocast_bf16": out_rec["bf16"], "fp32_rerun_max_dev": max(max(out_rec["fp32"][k]["vocab_max"], out_rec["fp32"][k]["ptr_max"])
                                                                        for k in ("ref_scores", "pos_scores", "neg_scores")),
            "torch": torch.__version__}


def main():
    MG.install_shims()
    torch.set_num_threads(8)
    cases = sys.argv[1:] or ["full_b1_f100_p100", "full_peaky_b2_f100_p100", "full_peaky_s29_b2_f100_p100"]
    path = os.path.join(HERE, "bf16_floor.json")
    res = json.load(open(path)) if os.path.exists(path) else {}
    for case in cases:
        res[case] = run(case)
        json.dump(res, open(path, "w"), indent=1, sort_keys=True)
    print("->", path)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Shared-prefix MMT passes vs three separate encoder calls at the FULL sequence length (100 x 100, B=2, bf16 operands, dropout 0):
same loss and scores (the forward is the same arithmetic), parameter gradients equal up to the bf16 rounding of the summed dQKV
(worst relative differences are on the key biases, whose exact gradient is zero: softmax is shift-invariant).
usage (GPU box): python tools/shared_prefix_check.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vitxt_gqa_amd.synth import make_batch, make_noise
from vitxt_gqa_amd.testing import make_model, to_device
dev='cuda:0'
F,P,V,B=100,100,5000,2
model=make_model(F,P,V,dtype=torch.bfloat16,attn_gain=4.0,dropout=0.0).to(dev).train()
batch=make_batch(B,F,P,V=V,seed=5)
batch["train_prev_inds"][:,3]=V+7
s=to_device(batch,dev); s.grounding_noise=tuple(t.to(dev) for t in make_noise(B,F,P,seed=5))
res={}
for mode in (False,True):
    model.share_mmt_prefix=mode
    model.zero_grad(set_to_none=True)
    out=model(s); loss=sum(l.mean() for l in out["losses"].values()); loss.backward()
    res[mode]=(loss.item(), {n:p.grad.detach().double().clone() for n,p in model.named_parameters() if p.grad is not None}, {k: out[k].detach().float().clone() for k in ("ref_scores","pos_scores","neg_scores")})
print("loss", res[False][0], res[True][0])
for k in res[False][2]: print(k, (res[False][2][k]-res[True][2][k]).abs().max().item())
tot=sum(g.norm().item()**2 for g in res[False][1].values())**0.5
worst=[]
for n,g in res[False][1].items():
    d=(g-res[True][1][n]).norm().item(); worst.append((d/(g.norm().item()+1e-30), d/tot, n))
worst.sort(reverse=True)
print("total grad norm", tot)
for w in worst[:8]: print("rel %.3e  rel_total %.3e  %s"%w)

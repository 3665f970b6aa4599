"""The data-parallel train step on the REAL T2S model (tiny golden fixture): the rank-averaged gradients that come out of
``ddp.GradBuckets`` (bucketed all-reduce overlapped with backward, dead parameters frozen statically) equal the
single-process gradient of the whole batch - the guarantee the reference gets from DistributedDataParallel
(base_trainer.py:51-71,128-137) with DistributedSampler shards (samplers.py:42-60).

  * two ranks sharing the one card of a 1-GPU box over gloo (always runs under -m gpu);
  * the same over RCCL (backend "nccl"), one rank per GPU, where the box has two GPUs.
Ranks are fresh child processes (never forked from this process, which holds the GPU)."""
import os
import socket
import subprocess
import sys

import pytest
import torch

from golden_util import Fixture

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
DEV = "cuda:0"


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_ranks(case, backend, one_gpu, tmp_path, world=2):
    out = str(tmp_path / "ddp_grads.pt")
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(0 if one_gpu else r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "ddp_rank_worker.py"), case, backend, "1" if one_gpu else "0", out],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = []
    try:
        for p in procs:
            o, _ = p.communicate(timeout=420)
            logs.append(o.decode(errors="replace")[-3000:])
    finally:
        for p in procs:                       # exact PIDs we started, nothing else
            if p.poll() is None:
                p.kill()
    assert all(p.returncode == 0 for p in procs), "\n----\n".join(logs)
    return torch.load(out)


def _single_process(case):
    from vitxt_gqa_amd.schema import is_dead_param
    from vitxt_gqa_amd.testing import build_model_for_fixture, to_device
    fx = Fixture(case)
    model = build_model_for_fixture(fx, torch.float32).to(DEV).train()
    s = to_device(fx.batch(), DEV)
    s.grounding_noise = (fx["E1"], fx["E2"])
    s.grounding_masks = fx.masks()
    out = model(s)
    loss = sum(l.mean() for l in out["losses"].values())
    loss.backward()
    grads = {n: p.grad.detach().cpu() for n, p in model.named_parameters() if p.grad is not None}
    dead = {n for n, _ in model.named_parameters() if is_dead_param(n)}
    return loss.item(), grads, dead


def _compare(res, case, world=2):
    loss, ref, dead = _single_process(case)
    assert res["world"] == world and res["n_buckets"] > 1
    assert res["collectives"] == 2 * res["n_buckets"]                        # two iterations, every bucket reduced in each
    assert abs(res["mean_loss"] - loss) < 1e-4 * abs(loss), (res["mean_loss"], loss)
    got = res["grads"]
    assert set(got) == set(ref) and not (set(got) & dead)                    # dead parameters are skipped, nothing else is
    total = sum(g.double().norm().item() ** 2 for g in ref.values()) ** 0.5
    for n, g in ref.items():
        d = (got[n].double() - g.double()).norm().item()
        assert d <= 2e-4 * g.double().norm().item() + 1e-6 * total, (n, d, g.norm().item())


def test_two_ranks_one_card_gloo_equal_single_process_gradient(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    res = _run_ranks("tiny_b2_f6_p8", "gloo", True, tmp_path)
    assert res["backend"] == "gloo"
    _compare(res, "tiny_b2_f6_p8")


def test_two_ranks_rccl_equal_single_process_gradient(tmp_path):
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device)")
    res = _run_ranks("tiny_b2_f6_p8", "nccl", False, tmp_path)
    assert res["backend"] == "nccl"
    _compare(res, "tiny_b2_f6_p8")


def test_single_rank_rccl_collectives_on_the_card(tmp_path):
    """What a 1-GPU box can show of the RCCL path: a ONE-rank ``nccl`` group on the card, the bucket all-reduces launched from
    the backward hooks as in a multi-rank run (communicator set-up, RCCL kernels on the bucket memory, the hand-over between
    RCCL's stream and the compute stream, barrier) - they must leave the gradients exactly as the plain backward made them."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    res = _run_ranks("tiny_b2_f6_p8", "nccl", True, tmp_path, world=1)
    assert res["backend"] == "nccl"
    _compare(res, "tiny_b2_f6_p8", world=1)

"""registering this build's T2S with the REFERENCE's own registry / BaseModel.
Runs only where the reference checkout exists (the build container); skipped on the GPU box."""
import os
import subprocess
import sys
import textwrap

import pytest

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "pythia")), reason="reference checkout not present")
def test_bind_reference_registry():
    code = textwrap.dedent("""
        import sys, types
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        ed = types.ModuleType("editdistance"); ed.eval = lambda a, b: 0; sys.modules["editdistance"] = ed
        from pythia.common.registry import registry as R
        class W:
            def write(self, *a, **k): pass
        class AD(dict):
            __getattr__ = dict.get
        R.register("writer", W())
        R.register("config", AD(datasets="vtextgqa", training_parameters=AD(evalai_inference=False)))
        R.register("vtextgqa_num_final_outputs", 64 + 48)
        class AP: BOS_IDX = 1
        R.register("vtextgqa_answer_processor", AP())
        from pythia.models.base_model import BaseModel
        import vitxt_gqa_amd
        T2S = vitxt_gqa_amd.bind_reference(R, BaseModel)
        assert R.get_model_class("t2s") is T2S and issubclass(T2S, BaseModel)
        cfg = vitxt_gqa_amd.t2s_model_config(6, 8); cfg.text_bert["vocab_size"] = 100
        m = T2S(cfg); m.build(); m.init_losses_and_metrics()            # reference's Losses wrapper, our criteria
        crit = [type(l.loss_criterion).__module__ for l in m.losses.losses]
        assert crit == ["vitxt_gqa_amd.losses"] * 2, crit
        assert len(m.state_dict()) == 220
        print("OK")
    """) % (REF, ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "OK" in out.stdout, out.stderr[-2000:]

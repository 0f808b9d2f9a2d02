"""CPU-side checks: the C-ABI library loads and exports every symbol of include/vlhip.h; host logic (config,
state_dict surface, loader) behaves like the reference's."""
import ctypes
import os

import pytest
import torch

from helpers import TASK_CFG, golden_config, load_golden, uc2_cfg_dict
from clg_vqa_amd import _lib
from clg_vqa_amd.config import BertConfig, M3PConfig, uc2_topology_check
from clg_vqa_amd.encoders import BertForVLTasks


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), "build it first: python -c 'import __graft_entry__ as g; g.build()'"
    names = _lib.header_symbols()
    assert len(names) >= 20
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(handle, n), n
    assert set(names) == set(_lib._SIGS), "python binding table out of sync with include/vlhip.h"
    assert _lib.lib().vl_version() == 100


def test_state_dict_surface_matches_reference_fixture():
    for name in ("uc2_tiny.npz", "uc2_wide.npz"):
        g = load_golden(name)
        cfg = golden_config(g)  # uc2_tiny = BASELINE config c1 (hidden 128, head dim 32): runs on the HIP path too
        m = BertForVLTasks(cfg, TASK_CFG, ["TASK15"])
        assert list(m.state_dict().keys()) == bytes(g["state_keys"]).decode().split("\n")


def test_unsupported_head_dim_is_rejected_loudly():
    bad = uc2_cfg_dict(hidden=768, heads=8, n_layers=1, vocab=100)  # head dim 96
    with pytest.raises(ValueError, match="head dim 64 or 32"):
        BertForVLTasks(BertConfig.from_dict(bad), TASK_CFG, ["TASK15"])


def test_full_config_census_on_meta_device():
    with torch.device("meta"):
        m = BertForVLTasks(BertConfig.from_dict(uc2_cfg_dict()), TASK_CFG, ["TASK15"])
    assert len(m.state_dict()) == 408
    assert sum(p.numel() for p in m.parameters()) == 281637426
    assert len(m.engine.param_list()) == 15 + 12 * 16


def test_config_defaults_and_topology_guard():
    c = BertConfig.from_dict({"hidden_size": 768})
    assert c.fusion_act == "relu" and c.layer_norm_eps == 1e-12 and c.fusion_method == "mul"
    assert M3PConfig.from_dict({}).n_layers == 12
    bad = uc2_cfg_dict(n_layers=2)
    bad["single_ln_sublayers"] = [0, 1]
    with pytest.raises(ValueError, match="UC2 single-stream topology"):
        uc2_topology_check(BertConfig.from_dict(bad))


def test_from_pretrained_local_file(tmp_path):
    cfg = BertConfig.from_dict(uc2_cfg_dict(n_layers=1, vocab=64))
    m = BertForVLTasks(cfg, TASK_CFG, ["TASK15"])
    sd = {k.replace("LayerNorm.weight", "LayerNorm.gamma").replace("LayerNorm.bias", "LayerNorm.beta")
          .replace("bert.", "roberta.", 1): v for k, v in m.state_dict().items()}
    path = tmp_path / "pytorch_model.bin"
    torch.save(sd, str(path))
    m2 = BertForVLTasks.from_pretrained(str(tmp_path), config=cfg, task_cfg=TASK_CFG, task_ids=["TASK15"])
    assert not m2.training
    for k, v in m.state_dict().items():
        assert torch.equal(v, m2.state_dict()[k]), k
    assert BertForVLTasks.from_pretrained(str(tmp_path / "nope.bin"), config=cfg, task_cfg=TASK_CFG,
                                          task_ids=["TASK15"]) is None


def test_m3p_state_dict_surface_matches_reference_fixture():
    import json
    from clg_vqa_amd.m3p import M3PForVLTasks
    g = load_golden("m3p_small.npz")
    cfg = golden_config(g, m3p=True)
    m = M3PForVLTasks(cfg, TASK_CFG, ["TASK15"])
    sd = m.state_dict()
    assert list(sd.keys()) == bytes(g["state_keys"]).decode().split("\n")
    assert [list(v.shape) for v in sd.values()] == json.loads(bytes(g["state_shapes"]).decode())
    assert len(m.engine.param_list()) == 10 + cfg.n_layers * 16


def test_m3p_full_config_census_on_meta_device():
    from clg_vqa_amd.m3p import M3PForVLTasks
    import json as _json
    import os
    cfg = dict(attention_probs_dropout_prob=0.1, hidden_act="gelu", hidden_dropout_prob=0.1, hidden_size=768,
               initializer_range=0.02, intermediate_size=3072, max_position_embeddings=514, n_heads=12, pooler_size=768,
               type_vocab_size=1, vocab_size=250002, pad_token_id=1, num_locs=5, image_embeddings="m3p",
               model="roberta", v_feature_size=2048, v_hidden_size=768, norm_embeddings=True, fusion_method="text",
               itm_dim=1, clf_hidden_size=1536)  # volta/config/m3p_base.json
    with torch.device("meta"):
        m = M3PForVLTasks(M3PConfig.from_dict(cfg), TASK_CFG, ["TASK15"])
    assert sum(p.numel() for p in m.parameters()) == 376903735
    with_grad = sum(p.numel() for p in m.engine.param_list()) + m.bert.pooler.dense.weight.numel() + \
        m.bert.pooler.dense.bias.numel() + sum(p.numel() for p in m.clfs_dict.parameters())
    assert with_grad == 283638066  # parameters that receive gradients


def test_warmup_linear_matches_transformers_schedule():
    """WarmupLinearSchedule of pytorch_transformers (train_task.py:271-274) lives on in transformers as
    get_linear_schedule_with_warmup (importable here): pins the schedule of FusedAdamW and of the oracle's restated
    optimizer step, including the reference's quirk that the FIRST optimizer step runs at lr = 0 when warm-up > 0."""
    from transformers.optimization import get_linear_schedule_with_warmup
    from clg_vqa_amd.optim import warmup_linear
    from oracle.adamw_oracle import warmup_linear as oracle_warmup
    for warm, total in ((0, 10), (3, 20), (10, 100), (7, 7)):
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.SGD([p], lr=1.0)
        sched = get_linear_schedule_with_warmup(opt, num_warmup_steps=warm, num_training_steps=total)
        for step in range(total + 3):
            lr = opt.param_groups[0]["lr"]  # the lr optimizer.step() number `step` runs at
            assert abs(lr - warmup_linear(step, warm, total)) < 1e-12, (warm, total, step)
            assert abs(lr - oracle_warmup(step, warm, total)) < 1e-12
            opt.step()
            sched.step()


def test_resume_uses_the_weights_only_loader(tmp_path):
    """resume() must not unpickle arbitrary objects.  A checkpoint of this package's format loads; a file
    holding a pickled object (what the reference's .tar does with its tbLogger) is refused with a clear message."""
    from clg_vqa_amd import train_utils

    class Opt(object):
        sched_step = 3

        def state_dict(self):
            return {"exp_avg": torch.zeros(2), "opt_step": 3, "sched_step": 3, "names": ["a"], "row_flags": None}

        def load_state_dict(self, sd):
            self.loaded = sd

    lin = torch.nn.Linear(2, 2)
    train_utils._ckpt(str(tmp_path), lin.state_dict(), Opt(), 7, 1, 0.5, dropout_rng=[11, 22])
    o = Opt()
    _, gs, ep, _, best = train_utils.resume(str(tmp_path / "pytorch_ckpt_latest.tar"), lin, o, None, None)
    assert (gs, ep, best) == (7, 2, 0.5) and o.loaded["names"] == ["a"]

    import argparse
    torch.save({"model_state_dict": {}, "tb_logger": argparse.Namespace(x=1)}, str(tmp_path / "ref.tar"))
    with pytest.raises(RuntimeError, match="weights-only"):
        train_utils.resume(str(tmp_path / "ref.tar"), lin, o, None, None)


def test_task_head_selection_and_prepared_weight_wiring():
    """host logic of the native task head (head.py): which feature sizes it takes, what it binds, and that the engine's
    single weight-preparation table covers the head Linears; small-M workspace sizes of the C ABI (no compute)."""
    from clg_vqa_amd.head import TaskHead
    cfg = BertConfig.from_dict(uc2_cfg_dict(n_layers=2, vocab=300))
    model = BertForVLTasks(cfg, TASK_CFG, ["TASK15"])
    head = model._task_head("TASK15")
    assert isinstance(head, TaskHead) and head.supported and head is model._task_head("TASK15")
    assert (head.H, head.P, head.C, head.NL) == (cfg.hidden_size, cfg.pooler_size, cfg.clf_hidden_size, 1842)
    assert head.act == cfg.fusion_act and [p.shape for p in head.params()] == [
        model.bert.t_pooler.dense.weight.shape, model.bert.t_pooler.dense.bias.shape,
        model.clfs_dict["TASK15"].logit_fc[0].weight.shape, model.clfs_dict["TASK15"].logit_fc[0].bias.shape,
        model.clfs_dict["TASK15"].logit_fc[2].weight.shape, model.clfs_dict["TASK15"].logit_fc[2].bias.shape,
        model.clfs_dict["TASK15"].logit_fc[3].weight.shape, model.clfs_dict["TASK15"].logit_fc[3].bias.shape]
    lins = model.engine.head_linears()
    assert lins == [model.bert.t_pooler.dense, model.clfs_dict["TASK15"].logit_fc[0], model.clfs_dict["TASK15"].logit_fc[3]]
    odd = BertConfig.from_dict(dict(uc2_cfg_dict(n_layers=2, vocab=300), clf_hidden_size=200))
    assert not BertForVLTasks(odd, TASK_CFG, ["TASK15"])._task_head("TASK15").supported  # module-by-module head instead
    L = _lib.lib()
    for (M, N, K) in ((256, 768, 3072), (256, 3072, 768), (256, 1842, 1536), (8, 1842, 128), (2048, 768, 768)):
        w = L.vl_gemm_small_ws_floats(M, N, K)
        assert w >= M * ((N + 3) // 4 * 4) and w % (M * ((N + 3) // 4 * 4)) == 0  # whole K-range slabs
        assert w // (M * ((N + 3) // 4 * 4)) <= max(1, (K + 63) // 64)
    assert L.vl_gemm_nt_path(14336, 768, 768, 3, 1) == 2 and L.vl_gemm_nt_path(256, 768, 3072, 3, 1) == 3
    assert L.vl_gemm_nt_path(256, 768, 3072, 3, 0) == 2 and L.vl_gemm_nt_path(100, 96, 40, 1, 0) == 0

"""IMP mask generation on the device (bit-exact index sets vs torch.nn.utils.prune) and the train / prune / sft /
eval drivers end to end on a small UC2 config.  Needs a real MI355X."""
import json
import os

import numpy as np
import pytest
import torch
import yaml

pytestmark = pytest.mark.gpu

from helpers import TASK_CFG, load_golden, uc2_cfg_dict  # noqa: E402
from clg_vqa_amd import ops, sft, train_task  # noqa: E402
from clg_vqa_amd.config import BertConfig  # noqa: E402
from clg_vqa_amd.encoders import BertForVLTasks  # noqa: E402


def test_imp_select_matches_reference_fixture_three_rounds():
    g = load_golden("imp_sft.npz")  # produced by the real prune.global_unstructured(L1Unstructured, 0.1)
    ws = [torch.from_numpy(g["w%d" % i]) for i in range(int(g["n"]))]
    w = torch.cat([x.reshape(-1) for x in ws]).cuda()
    mask = torch.ones_like(w)
    for r in range(3):
        k = round(0.1 * int(mask.sum().item()))
        new = torch.empty_like(mask)
        ops.imp_select(w, mask, new, k)
        idx = np.sort(np.nonzero(new.cpu().numpy() == 0)[0])
        np.testing.assert_array_equal(idx, g["pruned_idx_round%d" % r])
        mask = new


def test_imp_select_large_random_bit_exact_vs_torch_topk():
    gen = torch.Generator().manual_seed(3)
    n = 3_000_001
    w = (torch.randn(n, generator=gen) * 0.02)
    mask = (torch.rand(n, generator=gen) < 0.8).float()
    k = round(0.1 * int(mask.sum().item()))
    slc = mask == 1
    sub = (w * mask)[slc].abs()
    ref = mask.clone()
    part = torch.ones_like(sub)
    part[torch.topk(sub, k, largest=False).indices] = 0
    ref[slc] = part
    new = torch.empty(n, device="cuda")
    ops.imp_select(w.cuda(), mask.cuda(), new, k)
    assert torch.equal(new.cpu(), ref)
    assert int((new.cpu() == 0).sum()) == int((mask == 0).sum()) + k


def test_imp_select_threshold_ties_lowest_index_first():
    w = torch.tensor([0.5, 0.1, 0.3, 0.1, 0.1, 0.9, 0.1, 0.05] * 700, device="cuda")
    mask = torch.ones_like(w)
    k = 700 + 1000  # all 0.05s + 1000 of the 2800 values equal to 0.1
    new = torch.empty_like(w)
    ops.imp_select(w, mask, new, k)
    pruned = torch.nonzero(new == 0).flatten().cpu()
    assert pruned.numel() == k
    vals = w.cpu()[pruned]
    assert int((vals == 0.05).sum()) == 700 and int((vals == 0.1).sum()) == 1000
    tie_idx = torch.nonzero(w.cpu() == 0.1).flatten()
    assert torch.equal(torch.sort(pruned[vals == 0.1]).values, tie_idx[:1000])  # lowest flat indices first


def _write_cfgs(tmp_path, n_layers=1, vocab=300):
    cfg = uc2_cfg_dict(n_layers=n_layers, vocab=vocab)
    cpath = tmp_path / "uc2_small.json"
    cpath.write_text(json.dumps(cfg))
    task = {"TASK15": dict(TASK_CFG["TASK15"], name="GQA", max_seq_length=40, max_region_num=36, batch_size=8,
                           eval_batch_size=8, lr=4e-5, num_epoch=2, task_id=15)}
    tpath = tmp_path / "tasks.yml"
    tpath.write_text(yaml.safe_dump(task))
    return str(cpath), str(tpath)


def test_prune_then_sft_then_eval_drivers(tmp_path):
    cfg, tasks = _write_cfgs(tmp_path)
    out_p = str(tmp_path / "prune")
    common = ["--config_file", cfg, "--tasks_config_file", tasks, "--task", "15", "--steps_per_epoch", "3",
              "--val_batches", "1", "--adam_correct_bias", "--clip_grad_norm", "1.0", "--weight_decay", "0.0001"]
    train_task.main(["--mode", "prune", "--output_dir", out_p, "--num_epoch", "2"] + common)
    m0 = torch.load(os.path.join(out_p, "mask_lt0.pt"))
    m1 = torch.load(os.path.join(out_p, "mask_best.pt"))
    names = sft.uc2_prunable_names(2)
    assert set(k.replace(".weight_mask", "") for k in m1 if "v_" not in k and "mask" in k) >= set(names)
    tot = sum(m1[n + ".weight_mask"].numel() for n in names)
    z0 = sum(int((m0[n + ".weight_mask"] == 0).sum()) for n in names)
    z1 = sum(int((m1[n + ".weight_mask"] == 0).sum()) for n in names)
    assert z0 == round(0.1 * tot) and z1 == z0 + round(0.1 * (tot - z0))  # 10 %, then 19 %
    # the saved weights are masked, plain keys (save_prunned format)
    sd = torch.load(os.path.join(out_p, "pytorch_model_1.bin"))
    assert not any("_orig" in k or "_mask" in k for k in sd)
    w = sd[names[0] + ".weight"]
    assert torch.all(w[m1[names[0] + ".weight_mask"] == 0] == 0)
    # SFT from the pruned checkpoint under mask_best.pt
    out_s = str(tmp_path / "sft")
    train_task.main(["--mode", "sft", "--output_dir", out_s, "--num_epoch", "1", "--from_pretrained",
                     os.path.join(out_p, "pytorch_model_1.bin"), "--mask_dict_target",
                     os.path.join(out_p, "mask_best.pt")] + common)
    masked = torch.load(os.path.join(out_s, "pytorch_model_0.bin"))
    unmasked = torch.load(os.path.join(out_s, "pytorch_model_unmasked0.bin"))
    k0 = names[0] + ".weight"
    zero = m1[names[0] + ".weight_mask"] == 0
    assert torch.all(masked[k0][zero] == 0)
    assert torch.equal(masked[k0][~zero], unmasked[k0][~zero])
    # masked entries never moved: their gradient is exactly zero, and they carry no weight decay drift from 0
    assert torch.all(unmasked[k0][zero] == 0)
    # eval driver writes the reference's result format
    out_e = str(tmp_path / "eval")
    train_task.main(["--mode", "eval", "--output_dir", out_e, "--from_pretrained",
                     os.path.join(out_s, "pytorch_model_0.bin")] + common)
    res = json.load(open(os.path.join(out_e, "val_result.json")))
    assert len(res) == 8 and set(res[0]) == {"questionId", "prediction"}


def test_dense_driver_resume(tmp_path):
    cfg, tasks = _write_cfgs(tmp_path)
    out = str(tmp_path / "dense")
    common = ["--mode", "dense", "--config_file", cfg, "--tasks_config_file", tasks, "--task", "15", "--output_dir", out,
              "--steps_per_epoch", "2", "--val_batches", "1", "--adam_correct_bias", "--clip_grad_norm", "1.0"]
    train_task.main(common + ["--num_epoch", "1"])
    assert os.path.exists(os.path.join(out, "pytorch_ckpt_latest.tar"))
    sd = torch.load(os.path.join(out, "pytorch_model_0.bin"))
    model = BertForVLTasks(BertConfig.from_json_file(cfg), TASK_CFG, ["TASK15"])
    assert list(sd.keys()) == list(model.state_dict().keys())
    train_task.main(common + ["--num_epoch", "2", "--resume_file", os.path.join(out, "pytorch_ckpt_latest.tar")])
    assert os.path.exists(os.path.join(out, "pytorch_model_1.bin"))

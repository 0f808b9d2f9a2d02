"""The oracle (oracle/uc2_oracle.py) against fixtures produced by the real reference
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from helpers import TASK_CFG, golden_batch, golden_config, grad_digest, load_golden
from oracle import uc2_oracle as O
from clg_vqa_amd.synthetic import seeded_state_dict


@pytest.mark.parametrize("name", ["uc2_tiny.npz", "uc2_wide.npz", "uc2_deep.npz"])
def test_oracle_matches_reference_fixture(name):
    g = load_golden(name)
    config = golden_config(g)
    model = O.OracleUC2ForVLTasks(config, TASK_CFG, ["TASK15"])
    sd = seeded_state_dict(model.state_dict(), seed=int(g["seed"]))
    model.load_state_dict(sd, strict=True)
    model.eval()
    batch = golden_batch(g)
    loss, score, logits = O.forward_train(model, batch)
    # fp32 eager vs fp32 eager with a different (single-stream) op order: 1e-5 relative
    np.testing.assert_allclose(logits.detach().numpy(), g["logits"], rtol=1e-4, atol=2e-5)
    assert abs(float(loss.detach()) - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    assert float(score) == float(g["score"]) == 0.5  # half of the rows carry the reference's own argmax as label
    loss.backward()
    names = bytes(g["grad_names"]).decode().split("\n")
    params = dict(model.named_parameters())
    assert set(names) == {n for n, p in params.items() if p.grad is not None}
    for n in names:
        ref = g["grad::" + n]
        got = grad_digest(params[n].grad)
        # attention_self.key.bias has a mathematically zero gradient (softmax is invariant to a
        # per-query shift of all scores): both sides hold only rounding noise -> check it is ~0.
        if n.endswith("attention_self.key.bias"):
            assert got[-1] <= 1e-4 * g["grad::" + n.replace("key.bias", "query.bias")][-1], n
            continue
        scale = max(ref[-1], 1e-3)  # L2 norm of the reference gradient
        np.testing.assert_allclose(got[:256], ref[:256], rtol=2e-4, atol=2e-5 * scale, err_msg=n)
        assert abs(got[-1] - ref[-1]) <= 1e-4 * scale, n
        assert abs(got[-2] - ref[-2]) <= 1e-4 * max(ref[-2], 1e-2), n


def test_state_dict_keys_full_config():
    """408 state_dict keys / 215 unique tensors / 281 637 426 params for full UC2
    (checked on the 'meta' device so nothing is allocated)."""
    from helpers import uc2_cfg_dict
    from clg_vqa_amd.config import BertConfig
    with torch.device("meta"):
        m = O.OracleUC2ForVLTasks(BertConfig.from_dict(uc2_cfg_dict()), TASK_CFG, ["TASK15"])
    assert len(m.state_dict()) == 408
    ps = list(m.parameters())
    assert len(ps) == 215
    assert sum(p.numel() for p in ps) == 281637426
    names = O.uc2_prunable_names()
    mods = dict(m.named_modules())
    assert len(names) == 73 and sum(mods[n].weight.numel() for n in names) == 85524480
    # named_modules order == the order of the list (IMP concatenation order)
    order = [n for n, _ in m.named_modules() if n in set(names)]
    assert order == names


def test_imp_rounds_bit_exact():
    g = load_golden("imp_sft.npz")
    ws = [torch.from_numpy(g["w%d" % i]) for i in range(int(g["n"]))]
    masks = [torch.ones_like(w) for w in ws]
    for r in range(3):
        masks = O.imp_round(ws, masks, 0.1)
        flat = torch.cat([m.reshape(-1) for m in masks]).numpy()
        idx = np.sort(np.nonzero(flat == 0)[0])
        np.testing.assert_array_equal(idx, g["pruned_idx_round%d" % r])


def test_imp_fixture_ties_straddle_the_threshold():
    """Every round of the fixture has a planted group of equal magnitudes AT the k-th order statistic; the real
    torch.topk (CPU) pruned some members and left others -- its choice is recorded, and is not lowest-index-first."""
    from helpers import check_imp_contract
    g = load_golden("imp_sft.npz")
    w = np.concatenate([g["w%d" % i].reshape(-1) for i in range(int(g["n"]))])
    mask = np.ones_like(w)
    lowest_first = []
    for r in range(3):
        new = mask.copy()
        new[g["pruned_idx_round%d" % r]] = 0
        T, ties, pruned_ties = check_imp_contract(w, mask, new, int(g["k_round%d" % r]), g["tie_idx_round%d" % r])
        assert T == g["tie_value_round%d" % r] and len(ties) == 7 and 0 < len(pruned_ties) < 7
        np.testing.assert_array_equal(pruned_ties, g["tie_pruned_round%d" % r])
        lowest_first.append(bool((pruned_ties == ties[:len(pruned_ties)]).all()))
        mask = new
    assert not all(lowest_first)  # documents that torch's CPU pick among threshold ties is not an index order


def test_sft_masked_grad():
    g = load_golden("imp_sft.npz")
    w = torch.from_numpy(g["w0"]).clone().requires_grad_(True)
    mask = torch.from_numpy(g["sft_mask"])
    x = torch.from_numpy(g["sft_x"])
    y = x @ O.sft_apply(w, mask).t()
    np.testing.assert_allclose(y.detach().numpy(), g["sft_y"], rtol=1e-6, atol=1e-7)
    (y * y).sum().backward()
    np.testing.assert_allclose(w.grad.numpy(), g["sft_grad_orig"], rtol=1e-5, atol=1e-7)
    assert np.all(w.grad.numpy()[g["sft_mask"] == 0] == 0)


def test_m3p_oracle_matches_reference_fixture():
    from oracle import m3p_oracle as M
    g = load_golden("m3p_small.npz")
    config = golden_config(g, m3p=True)
    model = M.OracleM3PForVLTasks(config, TASK_CFG, ["TASK15"])
    sd = seeded_state_dict(model.state_dict(), seed=int(g["seed"]))
    model.load_state_dict(sd, strict=True)
    model.eval()
    batch = golden_batch(g)
    loss, score, logits = O.forward_train(model, batch)
    np.testing.assert_allclose(logits.detach().numpy(), g["logits"], rtol=1e-4, atol=2e-5)
    assert abs(float(loss.detach()) - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    loss.backward()
    names = bytes(g["grad_names"]).decode().split("\n")
    params = dict(model.named_parameters())
    got_names = {n for n, p in params.items() if p.grad is not None}
    # the reference registers the pooler under both bert.encoder.pooled_layer and bert.pooler
    assert {n.replace("bert.pooler.", "bert.encoder.pooled_layer.") for n in names} == \
        {n.replace("bert.pooler.", "bert.encoder.pooled_layer.") for n in got_names}
    for n in names:
        ref = g["grad::" + n]
        pn = n if n in params else n.replace("bert.pooler.", "bert.encoder.pooled_layer.")
        got = grad_digest(params[pn].grad)
        if n.endswith("k_lin.bias"):  # mathematically zero gradient (softmax shift invariance)
            assert got[-1] <= 1e-4 * g["grad::" + n.replace("k_lin.bias", "q_lin.bias")][-1], n
            continue
        scale = max(ref[-1], 1e-3)
        np.testing.assert_allclose(got[:256], ref[:256], rtol=2e-4, atol=2e-5 * scale, err_msg=n)
        assert abs(got[-1] - ref[-1]) <= 1e-4 * scale, n

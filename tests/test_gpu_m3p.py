"""M3PForVLTasks on the native engine vs the reference fixture and the M3P oracle.  Needs a real MI355X."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import TASK_CFG, golden_batch, golden_config, load_golden  # noqa: E402
from oracle import m3p_oracle as M  # noqa: E402
from oracle import uc2_oracle as O  # noqa: E402
from clg_vqa_amd import task_utils  # noqa: E402
from clg_vqa_amd.config import M3PConfig  # noqa: E402
from clg_vqa_amd.m3p import M3PForVLTasks  # noqa: E402
from clg_vqa_amd.optim import FusedAdamW  # noqa: E402
from clg_vqa_amd.synthetic import make_batch, seeded_state_dict  # noqa: E402

LOGIT_TOL, GRAD_REL_L2 = 1e-3, 4e-2


def _build(config, seed):
    model = M3PForVLTasks(config, TASK_CFG, ["TASK15"])
    sd = seeded_state_dict(model.state_dict(), seed=seed)
    model.load_state_dict(sd, strict=True)
    oracle = M.OracleM3PForVLTasks(config, TASK_CFG, ["TASK15"])
    oracle.load_state_dict({k: v for k, v in sd.items() if k in oracle.state_dict()}, strict=True)
    return model.cuda(), oracle


def _check(model, oracle, batch, ref_logits=None):
    model.eval()
    model.zero_grad()
    loss, score = task_utils.ForwardModelsTrain(model.config, TASK_CFG, "cuda", "TASK15", batch, model,
                                                torch.nn.CrossEntropyLoss())
    loss.backward()
    oracle.eval()
    oracle.zero_grad()
    oloss, oscore, ologits = O.forward_train(oracle, batch)
    oloss.backward()
    with torch.no_grad():
        b = tuple(t.cuda() for t in batch)
        logits = model(b[3], b[0], b[1], "TASK15", b[6], b[5], b[2])[0].cpu()
    err = (logits - ologits.detach()).abs().max().item()
    if ref_logits is not None:
        err = max(err, float(np.abs(logits.numpy() - ref_logits).max()))
    assert err <= LOGIT_TOL, err
    assert abs(float(loss.detach()) - float(oloss.detach())) <= 2e-4 * abs(float(oloss.detach()))
    og = {n: p.grad for n, p in oracle.named_parameters()}
    worst = (0.0, None)
    n_nograd = 0
    for n, p in model.named_parameters():
        if n not in og or og[n] is None:
            assert p.grad is None, "%s: a never-used M3P parameter received a gradient" % n
            n_nograd += 1
            continue
        g, r = p.grad.double().cpu(), og[n].double()
        if n.endswith("k_lin.bias"):
            assert g.norm().item() <= 1e-2 * og[n.replace("k_lin.bias", "q_lin.bias")].norm().item(), n
            continue
        rel = (g - r).norm().item() / max(r.norm().item(), 1e-12)
        worst = max(worst, (rel, n))
        assert rel <= GRAD_REL_L2, "%s: %.3e" % (n, rel)
    print("M3P: max |dlogit| %.2e, worst grad rel-L2 %.2e at %s, %d params without grad" % (err, worst[0], worst[1], n_nograd))
    return n_nograd


def test_m3p_matches_reference_fixture():
    g = load_golden("m3p_small.npz")
    config = golden_config(g, m3p=True)
    model, oracle = _build(config, int(g["seed"]))
    assert _check(model, oracle, golden_batch(g), ref_logits=g["logits"]) > 40


@pytest.mark.parametrize("T,V", [(20, 100), (40, 100)])
def test_m3p_long_visual_stream(T, V):
    """c4 shapes: 100 mmf boxes, S = 120 / 140, padded questions (length masks active)."""
    g = load_golden("m3p_small.npz")
    config = golden_config(g, m3p=True)
    model, oracle = _build(config, seed=9)
    batch = make_batch(3, seq_len=T, num_boxes=V, vocab_size=config.n_words, num_locs=5, l2_normalize=True, seed=77)
    _check(model, oracle, batch)


def test_m3p_training_step_skips_unused_parameters():
    g = load_golden("m3p_small.npz")
    config = golden_config(g, m3p=True)
    model, _ = _build(config, seed=4)
    model.train()
    opt = FusedAdamW(model, base_lr=1e-3, weight_decay=0.1, correct_bias=True, max_grad_norm=1.0)
    unused = model.bert.encoder.mrfr_dense.weight
    used = model.bert.encoder.ffns[0].lin1.weight
    u0, w0 = unused.detach().clone(), used.detach().clone()
    batch = make_batch(4, num_boxes=36, vocab_size=config.n_words, num_locs=5, l2_normalize=True, seed=5)
    for _ in range(2):
        loss, _ = task_utils.ForwardModelsTrain(config, TASK_CFG, "cuda", "TASK15", batch, model, torch.nn.CrossEntropyLoss())
        loss.backward()
        opt.step()
    assert torch.equal(unused.detach(), u0)       # no gradient -> untouched (no decay), like `if p.grad is None: continue`
    assert not torch.equal(used.detach(), w0)
    assert torch.isfinite(loss)


@pytest.mark.parametrize("m3p", [False, True])
def test_gradient_exchange_during_backward_is_bit_identical_to_the_plain_path(m3p):
    """FusedAdamW(overlap_reduce=True) takes each layer's gradients during backward (copy into the arena + the
    asynchronous all-reduce at world size > 1); with one rank it must give the gradient of the plain path, also when
    two backward passes are accumulated before step()."""
    from helpers import uc2_cfg_dict
    from clg_vqa_amd.config import BertConfig
    from clg_vqa_amd.encoders import BertForVLTasks
    if m3p:
        g = load_golden("m3p_small.npz")
        config = golden_config(g, m3p=True)
        build = lambda: _build(config, seed=4)[0]  # noqa: E731
        batch = make_batch(4, num_boxes=36, vocab_size=config.n_words, num_locs=5, l2_normalize=True, seed=5)
    else:
        config = BertConfig.from_dict(uc2_cfg_dict(vocab=999, n_layers=2))

        def build():
            torch.manual_seed(11)
            return BertForVLTasks(config, TASK_CFG, ["TASK15"]).cuda()
        batch = make_batch(8, vocab_size=999, seed=6)
    crit = torch.nn.CrossEntropyLoss()
    grads, losses, acc = [], [], []
    for overlap in (False, True):
        model = build()
        model.eval()  # no dropout: both runs see the same function
        opt = FusedAdamW(model, base_lr=1e-4, weight_decay=0.1, correct_bias=True, max_grad_norm=1.0, overlap_reduce=overlap)
        assert (model.engine.stack.layer_done_hook is not None) == overlap
        loss, _ = task_utils.ForwardModelsTrain(config, TASK_CFG, "cuda", "TASK15", batch, model, crit)
        loss.backward()
        assert bool(opt._pre) == overlap
        opt.arena.gather_grads(opt._pre)   # what step() sees: the complete flat gradient
        torch.cuda.synchronize()
        grads.append(opt.arena.grad.clone())
        opt.zero_grad()
        # gradient accumulation over two micro-batches (one rank): twice the single gradient, in both modes
        for _ in range(2):
            loss, _ = task_utils.ForwardModelsTrain(config, TASK_CFG, "cuda", "TASK15", batch, model, crit)
            loss.backward()
        opt.arena.gather_grads(opt._pre)
        torch.cuda.synchronize()
        acc.append(opt.arena.grad.clone())
        opt.zero_grad()
        # (round 3: the embedding scatter-adds, the box-location Linear and the clip norm reduce in a fixed order, so the two
        # gradient routes are bit-identical -- also through three optimizer steps)
        for _ in range(3):
            loss, _ = task_utils.ForwardModelsTrain(config, TASK_CFG, "cuda", "TASK15", batch, model, crit)
            loss.backward()
            opt.step()
        losses.append(float(loss))
    # same kernels, same data, fixed summation orders: bit-identical
    assert torch.equal(grads[0], grads[1]), int((grads[0] != grads[1]).sum())
    assert losses[0] == losses[1]
    assert torch.equal(acc[0], acc[1])
    assert torch.equal(acc[0], 2 * grads[0])

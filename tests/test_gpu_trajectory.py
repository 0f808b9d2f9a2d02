"""the training-step wrapper as a TRAJECTORY.  N optimizer steps on fixed batches with dropout 0:
native engine + FusedAdamW (3-pass bf16 forward, single-pass bf16 backward GEMMs, fused clip + AdamW + schedule) against
the oracle (CPU fp32, pinned to the reference) + the restated reference optimizer step (oracle/adamw_oracle.py; AdamW
arithmetic "parity unpinned", see its header).  This is the evidence for running the backward GEMMs in bf16: what
matters is not the per-tensor gradient error of one step but whether the loss sequence and the parameters follow the
fp32 run.  Needs a real MI355X."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import TASK_CFG, uc2_cfg_dict  # noqa: E402
from oracle import adamw_oracle as A  # noqa: E402
from oracle import uc2_oracle as O  # noqa: E402
from clg_vqa_amd import task_utils  # noqa: E402
from clg_vqa_amd.config import BertConfig  # noqa: E402
from clg_vqa_amd.encoders import BertForVLTasks  # noqa: E402
from clg_vqa_amd.optim import FusedAdamW  # noqa: E402
from clg_vqa_amd.synthetic import make_batch, seeded_state_dict  # noqa: E402

# Stated tolerances.  The loss is CE * 1842 (~1.3e4): 1e-3 relative is ~0.1 % of it.  Measured drift is printed.
LOSS_REL_TOL = 1e-3
PARAM_DEV_TOL = 0.05      # ||theta_native - theta_oracle|| / ||theta_oracle - theta_0|| per tensor after N steps (Adam's
                          # sign-like first updates turn gradient noise on near-zero entries into +-lr steps: reported)
# At the full depth the single-pass bf16 backward does NOT hold 1e-3 over 12 steps, and the tests say so: the CPU rehearsal
# (tests/precision_study.py --traj-only --trajectory 12 --traj-layers 12 [--traj-lr-scale 10]; outputs committed as
# profiles/r03_trajectory_rehearsal_12layers*.txt) gives, as the largest relative loss difference against the fp32 run over 12
# steps at lr 4e-5 / 4e-4:  fp32 backward behind the 3-pass forward 5.8e-7 / 3.8e-7 (so the run is not chaotic: what is
# measured below is gradient rounding, not amplification) | bf16 operands 1.8e-3 / 9.8e-3 | fp16 operands 4.6e-4 / 4.0e-3 |
# bf16 with two-term dY 2.4e-3 / 5.2e-3.  Every single-term scheme leaves ~0.1 - 1 % of noise per gradient tensor (far below
# the sampling noise of a mini-batch, but visible against a noise-free fp32 twin); only a 3-term backward (3 x the MFMA
# work of backward) would follow the fp32 run to 1e-3.  The bounds below are what the shipped precision delivers, with
# margin 2 - 3 x over the measured drift; the first steps (before the rounding noise has moved the weights) hold 1e-3.
# (measured on MI355X at lr 4e-5: loss 2.4e-3 max, parameters median 5.9e-2 / worst 1.3e-1 of the distance travelled)
DEEP_LOSS_REL_TOL = {1.0: 5e-3, 10.0: 3e-2}
DEEP_PARAM_DEV_TOL = {1.0: 0.25, 10.0: 0.5}


# (12 layers = the depth of BASELINE configs[1] (fixture uc2_deep's config), bs 8, 12 optimizer steps, warm-up over after 2
# steps; the x10 case runs at lr 4e-4 so that the weights move ~10x further from the common start: if the single-pass bf16
# backward bent the trajectory, this is where the loss sequences would separate)
@pytest.mark.parametrize("n_layers,n_steps,grad_acc,lr_scale", [(2, 8, 1, 1.0), (2, 4, 2, 1.0), (12, 12, 1, 1.0), (12, 12, 1, 10.0)])
def test_training_trajectory_follows_the_fp32_reference(n_layers, n_steps, grad_acc, lr_scale):
    cfg = uc2_cfg_dict(n_layers=n_layers, vocab=2000)
    cfg.update(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, v_hidden_dropout_prob=0.0,
               v_attention_probs_dropout_prob=0.0)
    config = BertConfig.from_dict(cfg)
    model = BertForVLTasks(config, TASK_CFG, ["TASK15"], dropout_prob=0.0)
    sd = seeded_state_dict(model.state_dict(), seed=77)
    model.load_state_dict(sd, strict=True)
    oracle = O.OracleUC2ForVLTasks(config, TASK_CFG, ["TASK15"], dropout_prob=0.0)
    oracle.load_state_dict(sd, strict=True)
    theta0 = {n: p.detach().clone() for n, p in oracle.named_parameters()}
    model = model.cuda().train()
    oracle.train()
    hp = dict(base_lr=4e-5 * lr_scale, weight_decay=1e-4, betas=(0.9, 0.999), eps=1e-6, correct_bias=True, warmup_steps=2,
              t_total=20, max_grad_norm=1.0)  # experiments/zero_shot/uc2/xgqa/train.dtu.sh:20-28; short schedule
    opt = FusedAdamW(model, overlap_reduce=None if grad_acc == 1 else False, **hp)
    opt.return_norm = True
    ref = A.ReferenceAdamW(oracle.named_parameters(), **hp)
    batches = [make_batch(8, vocab_size=2000, seed=900 + i) for i in range(4)]
    crit = torch.nn.CrossEntropyLoss()
    drift = []
    it = 0
    for s in range(n_steps):
        l_nat, l_ref, norms = 0.0, 0.0, None
        for _ in range(grad_acc):  # train_task.py:316-343: loss / grad_acc, backward, step every grad_acc batches
            b = batches[it % len(batches)]
            it += 1
            loss, score = task_utils.ForwardModelsTrain(config, TASK_CFG, "cuda", "TASK15", b, model, crit)
            (loss / grad_acc).backward()
            oloss, oscore, _ = O.forward_train(oracle, b)
            (oloss / grad_acc).backward()
            l_nat += float(loss) / grad_acc
            l_ref += float(oloss) / grad_acc
            assert float(score) == float(oscore)
        n_nat = float(opt.step())
        n_ref = float(ref.step())
        rel = abs(l_nat - l_ref) / abs(l_ref)
        drift.append(rel)
        print("step %d: loss native %.4f  reference %.4f  rel diff %.2e | grad norm %.3f vs %.3f" % (
            s, l_nat, l_ref, rel, n_nat, n_ref))
        tol = LOSS_REL_TOL if (n_layers < 12 or s < 3) else DEEP_LOSS_REL_TOL[lr_scale]
        assert rel <= tol, (s, l_nat, l_ref)
        assert abs(n_nat - n_ref) <= (2e-2 if n_layers < 12 else 5e-2) * n_ref
    assert opt.sched_step == ref.sched_step == n_steps
    # the loss moved (the comparison is not vacuous) and the parameters followed the same path
    nat = dict(model.named_parameters())
    worst = (0.0, None)
    devs, table = [], []
    for n, p in oracle.named_parameters():
        if n.endswith("attention_self.key.bias"):
            continue  # mathematically zero gradient (softmax shift invariance): both runs move it by rounding noise only
        moved = (p.detach() - theta0[n]).double().norm().item()
        if moved == 0.0:
            continue
        d = (nat[n].detach().cpu().double() - p.detach().double()).norm().item() / moved
        devs.append(d)
        table.append((d, moved, n))
        worst = max(worst, (d, n))
    print("max loss drift %.2e over %d steps; parameter deviation relative to the distance travelled: median %.3e, worst "
          "%.3e at %s" % (max(drift), n_steps, float(np.median(devs)), worst[0], worst[1]))
    table.sort(reverse=True)
    print("per-tensor deviation (||native - reference|| / ||reference - start||), the ten largest of %d tensors:" % len(table))
    for d, moved, n in table[:10]:
        print("    %.3e   (travelled %.3e)   %s" % (d, moved, n))
    first, last = float(drift[0]), float(drift[-1])
    print("loss drift first / last step: %.2e / %.2e" % (first, last))
    assert worst[0] <= (PARAM_DEV_TOL if n_layers < 12 else DEEP_PARAM_DEV_TOL[lr_scale])


@pytest.mark.parametrize("overlap", [None, False])
def test_fixed_layers_stay_fixed_and_the_rest_follows_the_reference(overlap):
    """config.fixed_layers -> train_utils.freeze_layers (volta/volta/train_utils.py:305-311, called from train_task.py before
    the optimizer is built): frozen tensors (all embeddings + one attention sub-layer + one feed-forward sub-layer of another
    layer) must not move by a single bit through three optimizer steps; everything else follows oracle + restated optimizer.
    Both gradient routes: written into the optimizer's arena during backward (a layer with a frozen tensor falls back to the
    autograd route on its own) and gathered in step()."""
    from clg_vqa_amd.train_utils import freeze_layers
    cfg = uc2_cfg_dict(n_layers=3, vocab=2000)
    cfg.update(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, v_hidden_dropout_prob=0.0,
               v_attention_probs_dropout_prob=0.0)
    cfg["fixed_layers"] = ["embeddings", "encoder.layer.2.", "encoder.layer.5.output"]
    config = BertConfig.from_dict(cfg)
    model = BertForVLTasks(config, TASK_CFG, ["TASK15"], dropout_prob=0.0)
    sd = seeded_state_dict(model.state_dict(), seed=78)
    model.load_state_dict(sd, strict=True)
    oracle = O.OracleUC2ForVLTasks(config, TASK_CFG, ["TASK15"], dropout_prob=0.0)
    oracle.load_state_dict(sd, strict=True)
    freeze_layers(model)
    frozen = sorted(n for n, p in model.named_parameters() if not p.requires_grad)
    assert any("word_embeddings" in n for n in frozen) and any("layer.2.attention_self.query" in n for n in frozen)
    assert any("layer.5.output.dense" in n for n in frozen) and not any("layer.5.intermediate" in n for n in frozen)
    op = dict(oracle.named_parameters())
    for n in frozen:
        op[n].requires_grad = False
    theta0 = {n: p.detach().clone() for n, p in oracle.named_parameters()}
    model = model.cuda().train()
    oracle.train()
    hp = dict(base_lr=4e-4, weight_decay=1e-4, betas=(0.9, 0.999), eps=1e-6, correct_bias=True, warmup_steps=1, t_total=20,
              max_grad_norm=1.0)
    opt = FusedAdamW(model, overlap_reduce=overlap, **hp)
    ref = A.ReferenceAdamW(oracle.named_parameters(), **hp)
    crit = torch.nn.CrossEntropyLoss()
    for s_ in range(3):
        b = make_batch(8, vocab_size=2000, seed=950 + s_)
        loss, _ = task_utils.ForwardModelsTrain(config, TASK_CFG, "cuda", "TASK15", b, model, crit)
        loss.backward()
        oloss, _, _ = O.forward_train(oracle, b)
        oloss.backward()
        assert abs(float(loss) - float(oloss)) <= LOSS_REL_TOL * abs(float(oloss)), (s_, float(loss), float(oloss))
        opt.step()
        ref.step()
    nat = dict(model.named_parameters())
    worst = (0.0, None)
    for n, p in oracle.named_parameters():
        got = nat[n].detach().cpu()
        if n in frozen:
            assert torch.equal(got, theta0[n]), "frozen tensor moved: " + n
            assert torch.equal(p.detach(), theta0[n])
            continue
        if n.endswith("attention_self.key.bias"):
            continue
        moved = (p.detach() - theta0[n]).double().norm().item()
        assert moved > 0.0, n
        worst = max(worst, ((got.double() - p.detach().double()).norm().item() / moved, n))
    print("fixed_layers (%s): %d frozen tensors bit-identical; worst deviation of the others relative to the distance "
          "travelled %.3e at %s" % ("arena sinks" if overlap is None else "gathered in step()", len(frozen), worst[0], worst[1]))
    assert worst[0] <= PARAM_DEV_TOL

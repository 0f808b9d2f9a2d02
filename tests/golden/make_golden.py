#!/usr/bin/env python3
"""Generate golden vectors from the REAL reference (nooralahzadeh/CLG-VQA at /root/reference).

Run in the build container only (the reference does not travel):

    python tests/golden/make_golden.py

The reference is imported read-only with harness-side stub modules for packages the image lacks
(SURVEY.md §8c): boto3/botocore (S3 helpers in volta/volta/utils.py:20-22), and -- for
``volta.task_utils`` -- lmdb / tensorpack / msgpack_numpy / the broken ``volta.datasets`` package
(``volta/volta/datasets/__init__.py:78`` raises NameError as published).  ``Tensor.cuda`` is made a
no-op because ``ForwardModelsTrain`` calls ``.cuda()`` unconditionally (task_utils.py:309, :708).

What is written (data only -- inputs and expected outputs, never reference source):

* uc2_tiny.npz   c1 of BASELINE.json: 2-layer / hidden-128 / 4 heads, bs=4, T=20, V=36.
* uc2_wide.npz   1 full-width layer (H=768, 12 heads, I=3072), bs=4 -- pins the head-dim-64 path.
* m3p_small.npz  M3P (emb_dim 256 / 4 heads / 2 layers, 36 boxes, L2-normalised 5-d locations), bs=4: logits, loss,
                 gradients of the parameters jointfwd touches, and the full state_dict key/shape list (incl. the
                 never-used modules).
* imp_sft.npz    3 rounds of prune.global_unstructured(L1Unstructured, 0.1) on a toy weight list
                 (incl. a forced tie group) + CustomFromMask gradients.

Weights are not stored: they are a pure function of (parameter name, shape, seed) --
``clg_vqa_amd.synthetic.seeded_state_dict`` -- loaded into the reference model with
``load_state_dict(strict=True)``.
"""
import importlib.machinery
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference/volta"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def import_reference():
    _stub("boto3")
    b = _stub("botocore")
    b.exceptions = _stub("botocore.exceptions", ClientError=type("ClientError", (Exception,), {}))
    _stub("lmdb")
    tp = _stub("tensorpack")
    tp.dataflow = _stub("tensorpack.dataflow")
    _stub("msgpack_numpy", patch=lambda: None)
    sys.path.insert(0, REF)
    import volta  # noqa: F401  (the reference package)
    ds = _stub("volta.datasets", DatasetMapTrain={}, DatasetMapEval={})
    ds.__path__ = []
    _stub("volta.datasets._image_features_reader", ImageFeaturesH5Reader=object)
    torch.Tensor.cuda = lambda self, *a, **k: self
    from volta.config import BertConfig, M3PConfig
    from volta.encoders import BertForVLTasks, M3PForVLTasks
    try:
        from volta import task_utils
    except Exception as e:  # pragma: no cover
        print("volta.task_utils not importable (%r); loss golden falls back to the survey's formula" % (e,))
        task_utils = None
    import_reference.m3p = (M3PConfig, M3PForVLTasks)
    return BertConfig, BertForVLTasks, task_utils


def uc2_cfg_dict(hidden, heads, inter, n_layers, vocab):
    cfg = json.load(open(os.path.join(REF, "config/uc2_base.json")))
    n_sub = 2 * n_layers
    cfg.update(hidden_size=hidden, num_attention_heads=heads, intermediate_size=inter,
               v_hidden_size=hidden, v_num_attention_heads=heads, v_intermediate_size=inter,
               pooler_size=hidden, clf_hidden_size=hidden, vocab_size=vocab)
    for k in ("tt_attn_sublayers", "tv_attn_sublayers", "vt_attn_sublayers", "vv_attn_sublayers"):
        cfg[k] = list(range(0, n_sub, 2))
    for k in ("t_ff_sublayers", "v_ff_sublayers"):
        cfg[k] = list(range(1, n_sub, 2))
    for k in ("shared_sublayers", "single_ln_sublayers"):
        cfg[k] = list(range(n_sub))
    cfg["bert_layer2attn_sublayer"] = {str(i): 2 * i for i in range(n_layers)}
    cfg["bert_layer2ff_sublayer"] = {str(i): 2 * i + 1 for i in range(n_layers)}
    return cfg


TASK_CFG = {"TASK15": {"type": "VL-classifier-GQA", "num_labels": 1842, "process": "normal",
                       "semantic_lambda": 10, "loss": "CrossEntropyLoss"}}


def grad_digest(g):
    """Compact, order-sensitive fingerprint of a gradient tensor: first 256 elements + moments."""
    f = g.detach().reshape(-1).double()
    return np.concatenate([f[:256].numpy(), [f.sum().item(), f.abs().sum().item(), (f * f).sum().sqrt().item()]])


def m3p_cfg_dict(dim, heads, n_layers, vocab):
    cfg = json.load(open(os.path.join(REF, "config/m3p_base.json")))
    cfg.update(emb_dim=dim, n_heads=heads, n_layers=n_layers, refine_layers=1, n_words=vocab, vocab_size=vocab,
               hidden_size=dim, pooler_size=dim, clf_hidden_size=2 * dim)
    return cfg


def run_model_case(BertConfig, BertForVLTasks, task_utils, cfg, seed, out_path, vocab, m3p=False):
    from clg_vqa_amd.synthetic import make_batch, seeded_state_dict
    config = BertConfig.from_dict(cfg)
    torch.manual_seed(0)
    model = BertForVLTasks(config, TASK_CFG, ["TASK15"])
    sd = seeded_state_dict(model.state_dict(), seed=seed)
    model.load_state_dict(sd, strict=True)
    batch = make_batch(4, seq_len=20, num_boxes=36, vocab_size=vocab, seed=100 + seed, fp16_exact=True,
                       num_locs=5 if m3p else 7, l2_normalize=m3p)
    model.eval()  # dropout off; parity under dropout is not defined across RNG streams (SURVEY §7 hard part 4)
    crit = torch.nn.CrossEntropyLoss()
    if task_utils is not None:
        loss, score = task_utils.ForwardModelsTrain(config, TASK_CFG, "cpu", "TASK15", batch, model, crit)
    else:
        raise SystemExit("task_utils import failed; refusing to write goldens without the reference loss")
    model.zero_grad()
    loss.backward()
    feats, spat, imask, q, target, tmask, seg, _, _, dist = batch
    with torch.no_grad():
        logits = model(q, feats, spat, "TASK15", seg, tmask, imask)[0]
    out = dict(
        cfg_json=np.frombuffer(json.dumps(cfg).encode(), dtype=np.uint8),
        seed=np.int64(seed), vocab=np.int64(vocab),
        features=feats.numpy().astype(np.float16), spatials=spat.numpy().astype(np.float16),
        image_mask=imask.numpy().astype(np.int8), question=q.numpy().astype(np.int32),
        label=target.argmax(1).numpy().astype(np.int32), input_mask=tmask.numpy().astype(np.int8),
        distances=dist.numpy().astype(np.float16),
        logits=logits.numpy(), loss=np.float64(float(loss)), score=np.float64(float(score)),
    )
    seen = set()
    names = []
    for n, p in model.named_parameters():  # named_parameters de-duplicates the aliased modules
        if p.grad is None or id(p) in seen:
            continue
        seen.add(id(p))
        names.append(n)
        out["grad::" + n] = grad_digest(p.grad)
    out["grad_names"] = np.frombuffer("\n".join(names).encode(), dtype=np.uint8)
    out["state_keys"] = np.frombuffer("\n".join(model.state_dict().keys()).encode(), dtype=np.uint8)
    out["state_shapes"] = np.frombuffer(json.dumps([list(v.shape) for v in model.state_dict().values()]).encode(),
                                        dtype=np.uint8)
    np.savez_compressed(out_path, **out)
    print("wrote", out_path, "loss=%.6f score=%.4f n_grads=%d size=%.1f KB" % (
        float(loss), float(score), len(names), os.path.getsize(out_path) / 1024))


def run_imp_sft_case(out_path):
    """torch.nn.utils.prune is the arithmetic the reference calls (train_task_prunning.py:80-84,
    train_task_sft.py:128-132); run the real thing on a toy module list."""
    from torch import nn
    from torch.nn.utils import prune
    rs = np.random.RandomState(7)
    shapes = [(24, 16), (16, 16), (40, 16), (16, 40), (16, 16)]
    ws = [torch.from_numpy(rs.randn(*s).astype(np.float32) * 0.02) for s in shapes]
    # forced ties: a group of equal magnitudes straddling the first-round threshold region
    ws[1].view(-1)[:12] = 0.0023
    ws[3].view(-1)[:9] = -0.0023
    mods = [nn.Linear(s[1], s[0], bias=False) for s in shapes]
    for m, w in zip(mods, ws):
        m.weight.data.copy_(w)
    params = tuple((m, "weight") for m in mods)
    out = {"n": np.int64(len(shapes))}
    for i, w in enumerate(ws):
        out["w%d" % i] = w.numpy().copy()
    for r in range(3):
        prune.global_unstructured(params, pruning_method=prune.L1Unstructured, amount=0.1)
        flat = torch.cat([m.weight_mask.reshape(-1) for m in mods])
        out["pruned_idx_round%d" % r] = np.sort(np.nonzero(flat.numpy() == 0)[0]).astype(np.int64)
        # rewind weight_orig to theta_0 like train_task_prunning.py:803-806 (values unchanged here)
    # SFT: grads under the mask (CustomFromMask on a fresh module)
    lin = nn.Linear(16, 24, bias=True)
    lin.weight.data.copy_(ws[0])
    lin.bias.data.zero_()
    mask = mods[0].weight_mask.clone()
    prune.CustomFromMask.apply(lin, "weight", mask=mask)
    x = torch.from_numpy(rs.randn(5, 16).astype(np.float32))
    y = lin(x)
    (y * y).sum().backward()
    out["sft_x"] = x.numpy()
    out["sft_mask"] = mask.numpy()
    out["sft_y"] = y.detach().numpy()
    out["sft_grad_orig"] = lin.weight_orig.grad.numpy()
    np.savez_compressed(out_path, **out)
    print("wrote", out_path, [len(out["pruned_idx_round%d" % r]) for r in range(3)])


def main():
    BertConfig, BertForVLTasks, task_utils = import_reference()
    run_model_case(BertConfig, BertForVLTasks, task_utils, uc2_cfg_dict(128, 4, 512, 2, 1000), 1,
                   os.path.join(HERE, "uc2_tiny.npz"), 1000)
    run_model_case(BertConfig, BertForVLTasks, task_utils, uc2_cfg_dict(768, 12, 3072, 1, 1000), 2,
                   os.path.join(HERE, "uc2_wide.npz"), 1000)
    M3PConfig, M3PForVLTasks = import_reference.m3p
    run_model_case(M3PConfig, M3PForVLTasks, task_utils, m3p_cfg_dict(256, 4, 2, 300), 3,
                   os.path.join(HERE, "m3p_small.npz"), 300, m3p=True)
    run_imp_sft_case(os.path.join(HERE, "imp_sft.npz"))


if __name__ == "__main__":
    main()

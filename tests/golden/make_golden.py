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
* uc2_deep.npz   the full-depth trunk (12 layers, H=768, 12 heads, I=3072; vocab 2000), bs=8 -- pins 48 GEMMs deep.
* m3p_small.npz  M3P (emb_dim 256 / 4 heads / 2 layers, 36 boxes, L2-normalised 5-d locations), bs=4: logits, loss,
                 gradients of the parameters jointfwd touches, and the full state_dict key/shape list (incl. the
                 never-used modules).
* imp_sft.npz    3 rounds of prune.global_unstructured(L1Unstructured, 0.1) on a toy weight list with, in EVERY
                 round, a group of 7 equal magnitudes planted AT the k-th order statistic (4 of the 7 fall inside the k
                 smallest, 3 outside): the fixture records which members torch's CPU top-k pruned (its choice among
                 threshold ties is implementation-defined), the tie value and the group's indices + CustomFromMask
                 gradients.

In every model fixture half of the rows carry the reference's own argmax as gold label, so ``score`` is non-zero
and a wrong ``compute_score_with_logits`` cannot pass.

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


def run_model_case(BertConfig, BertForVLTasks, task_utils, cfg, seed, out_path, vocab, m3p=False, batch_size=4,
                   n_grad_digests=None):
    from clg_vqa_amd.synthetic import make_batch, seeded_state_dict
    config = BertConfig.from_dict(cfg)
    torch.manual_seed(0)
    model = BertForVLTasks(config, TASK_CFG, ["TASK15"])
    sd = seeded_state_dict(model.state_dict(), seed=seed)
    model.load_state_dict(sd, strict=True)
    batch = make_batch(batch_size, seq_len=20, num_boxes=36, vocab_size=vocab, seed=100 + seed, fp16_exact=True,
                       num_locs=5 if m3p else 7, l2_normalize=m3p)
    model.eval()  # dropout off; parity under dropout is not defined across RNG streams (SURVEY §7 hard part 4)
    crit = torch.nn.CrossEntropyLoss()
    # gold label := the reference's own prediction for the first half of the rows (score = 0.5 instead of the 0.0 a
    # random label gives), the prior-distance row keeps 0 at the gold label like the dataset builds it
    feats, spat, imask, q, target, tmask, seg, qid, ix, dist = batch
    with torch.no_grad():
        pred = model(q, feats, spat, "TASK15", seg, tmask, imask)[0].argmax(1)
    half = batch_size // 2
    old = target.argmax(1)
    for b in range(half):
        dist[b, old[b]] = dist[b, pred[b]] if dist[b, pred[b]] > 0 else 0.5
        dist[b, pred[b]] = 0.0
        target[b].zero_()
        target[b, pred[b]] = 1.0
    batch = (feats, spat, imask, q, target, tmask, seg, qid, ix, dist)
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


def plant_straddling_ties(ws, rounds=3, amount=0.1, group=7, below=4):
    """For every round r: let k_r = round(amount * n_remaining) and v_r the (k_r - below + 1)-th smallest remaining
    |w|; overwrite `group - 1` of the largest entries with +-v_r.  The tie group then occupies ranks
    k_r-below+1 .. k_r-below+group: exactly `below` of its `group` members are among the k_r smallest, whichever
    members the selection picks.  Planning only needs the multiset of remaining values (all ties are equal), so it
    does not depend on the pick.  Returns [(v_r, k_r)]."""
    flat = torch.cat([w.view(-1) for w in ws])  # copies; edits go through `views`
    sizes = [w.numel() for w in ws]

    def write(i, val):
        t = 0
        while i >= sizes[t]:
            i -= sizes[t]
            t += 1
        ws[t].view(-1)[i] = val

    donors = torch.argsort(flat.abs(), descending=True).tolist()  # largest first: never pruned in 3 rounds of 10 %
    removed = torch.zeros(flat.numel(), dtype=torch.bool)
    plan = []
    for r in range(rounds):
        rem = torch.nonzero(~removed).flatten()
        k = round(amount * rem.numel())
        order = rem[torch.argsort(flat[rem].abs(), stable=True)]
        v = float(flat[order[k - below]].abs())
        for j in range(group - 1):
            d = donors.pop(0)
            val = v if j % 2 == 0 else -v
            write(d, val)
            flat[d] = val
        # this round removes the k smallest values (which tie members go is the selection's choice; equal values)
        order = rem[torch.argsort(flat[rem].abs(), stable=True)]
        removed[order[:k]] = True
        plan.append((v, k))
    return plan


def run_imp_sft_case(out_path):
    """torch.nn.utils.prune is the arithmetic the reference calls (train_task_prunning.py:80-84,
    train_task_sft.py:128-132); run the real thing on a toy module list."""
    from torch import nn
    from torch.nn.utils import prune
    rs = np.random.RandomState(7)
    shapes = [(24, 16), (16, 16), (40, 16), (16, 40), (16, 16)]
    ws = [torch.from_numpy(rs.randn(*s).astype(np.float32) * 0.02) for s in shapes]
    # a group of equal magnitudes well below the first threshold (pruned as a whole in round 0)
    ws[1].view(-1)[:12] = 0.0023
    ws[3].view(-1)[:9] = -0.0023
    plan = plant_straddling_ties(ws)
    mods = [nn.Linear(s[1], s[0], bias=False) for s in shapes]
    for m, w in zip(mods, ws):
        m.weight.data.copy_(w)
    params = tuple((m, "weight") for m in mods)
    out = {"n": np.int64(len(shapes))}
    for i, w in enumerate(ws):
        out["w%d" % i] = w.numpy().copy()
    absw = np.abs(np.concatenate([w.numpy().reshape(-1) for w in ws]))
    prev = np.zeros(absw.size, dtype=bool)
    for r in range(3):
        prune.global_unstructured(params, pruning_method=prune.L1Unstructured, amount=0.1)
        flat = torch.cat([m.weight_mask.reshape(-1) for m in mods]).numpy()
        pruned = flat == 0
        out["pruned_idx_round%d" % r] = np.sort(np.nonzero(pruned)[0]).astype(np.int64)
        v, k = plan[r]
        ties = np.nonzero((absw == np.float32(v)) & ~prev)[0]  # tie members still unpruned when the round starts
        new = pruned & ~prev
        assert new.sum() == k, (new.sum(), k)
        assert (absw[new] <= np.float32(v)).all() and (absw[~pruned] >= np.float32(v)).all()
        inside = np.nonzero(new[ties])[0]
        assert 0 < inside.size < ties.size, "the tie group must straddle the k-th boundary"
        out["tie_value_round%d" % r] = np.float32(v)
        out["tie_idx_round%d" % r] = ties.astype(np.int64)
        out["tie_pruned_round%d" % r] = ties[inside].astype(np.int64)  # torch CPU top-k's choice
        out["k_round%d" % r] = np.int64(k)
        print("round %d: k=%d tie value %.6g, group %s, torch (CPU) pruned %s -> lowest-index-first: %s" % (
            r, k, v, ties.tolist(), ties[inside].tolist(), bool((inside == np.arange(inside.size)).all())))
        prev = pruned.copy()
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


def run_records_case(out_path):
    """The reference's per-sample preprocessing (BertPreprocessBatch.__call__, gqa_dataset_semantic_code_mix.py:564-651)
    and its per-sample prior-distance loop (get_embeddingdist, :371-381) on synthetic records: UC2 (7-d locations) and
    M3P (5-d, norm_embeddings) variants, one record with fewer boxes than region_len in each (the M3P one pins the
    NaN rows of Appendix B quirk 4), one question longer than seq_len."""
    import types
    import volta.datasets as ds_pkg
    ds_pkg.__path__ = [os.path.join(REF, "volta", "datasets")]  # the stub package: its broken __init__ never runs
    from volta.datasets import gqa_dataset_semantic_code_mix as G

    class Tok(object):  # question string -> ids; the real tokenizer (XLM-R sentencepiece) is not in the image
        def encode(self, q):
            return [0] + [5 + (sum(ord(c) for c in wd) % 900) for wd in q.split()] + [2]

    rs = np.random.RandomState(11)
    region_len, seq_len, C = 6, 8, 40
    out = {"region_len": np.int64(region_len), "seq_len": np.int64(seq_len), "num_labels": np.int64(C)}
    recs = []
    for i, (n, nwords) in enumerate([(6, 4), (4, 11), (6, 6)]):
        w, h = float(rs.randint(300, 800)), float(rs.randint(200, 600))
        x1 = rs.uniform(0, 0.6 * w, n); y1 = rs.uniform(0, 0.6 * h, n)
        boxes = np.stack([x1, y1, x1 + rs.uniform(5, 0.4 * w, n), y1 + rs.uniform(5, 0.4 * h, n)], 1).astype(np.float32)
        feats = (np.maximum(rs.randn(n, 2048), 0) * 1.5).astype(np.float16).astype(np.float32)
        q = " ".join("w%d" % rs.randint(0, 50) for _ in range(nwords)) + " ?"
        lab = int(rs.randint(0, C))
        recs.append(dict(features=feats, boxes=boxes, img_w=w, img_h=h, img_id=100 + i,
                         entry=dict(question_id=7000 + i, image_id=100 + i, question=q, labels=[lab], scores=[1.0])))
        out["rec%d_features" % i] = feats.astype(np.float16)
        out["rec%d_boxes" % i] = boxes
        out["rec%d_wh" % i] = np.array([w, h])
        out["rec%d_tokens" % i] = np.array(Tok().encode(q), dtype=np.int64)
        out["rec%d_label" % i] = np.int64(lab)
    sem = {(j, t): float(np.float32(rs.uniform(0.05, 1.0))) for j in range(C) for t in range(C) if j != t}
    out["semantic_pairs"] = np.array([[j, t] for (j, t) in sem], dtype=np.int64)
    out["semantic_vals"] = np.array(list(sem.values()), dtype=np.float64)
    for tag, nl, norm in (("uc2", 7, False), ("m3p", 5, True)):
        pre = G.BertPreprocessBatch(Tok(), "xlm-roberta-base", seq_len, region_len, len(recs), num_locs=nl,
                                    padding_index=1, norm_embeddings=norm)
        with np.errstate(invalid="ignore", divide="ignore"):
            cols = [pre(r) for r in recs]
        names = ("image_feat", "image_loc", "image_mask", "input_ids", "input_mask", "segment_ids", "labels", "scores",
                 "image_id", "question_id")
        for ci, nm in enumerate(names):
            out["%s_%s" % (tag, nm)] = np.stack([np.asarray(c[ci]) for c in cols])
        labels = np.stack([np.asarray(c[6]) for c in cols])
        fake = types.SimpleNamespace(num_labels=C, semantic_dict=sem)
        out["%s_distances" % tag] = G.GQAClassificationLoader.get_embeddingdist(fake, labels)
    np.savez_compressed(out_path, **out)
    print("wrote", out_path, "%.1f KB" % (os.path.getsize(out_path) / 1024))


def main():
    BertConfig, BertForVLTasks, task_utils = import_reference()
    run_records_case(os.path.join(HERE, "records.npz"))
    run_model_case(BertConfig, BertForVLTasks, task_utils, uc2_cfg_dict(128, 4, 512, 2, 1000), 1,
                   os.path.join(HERE, "uc2_tiny.npz"), 1000)
    run_model_case(BertConfig, BertForVLTasks, task_utils, uc2_cfg_dict(768, 12, 3072, 1, 1000), 2,
                   os.path.join(HERE, "uc2_wide.npz"), 1000)
    run_model_case(BertConfig, BertForVLTasks, task_utils, uc2_cfg_dict(768, 12, 3072, 12, 2000), 4,
                   os.path.join(HERE, "uc2_deep.npz"), 2000, batch_size=8)
    M3PConfig, M3PForVLTasks = import_reference.m3p
    run_model_case(M3PConfig, M3PForVLTasks, task_utils, m3p_cfg_dict(256, 4, 2, 300), 3,
                   os.path.join(HERE, "m3p_small.npz"), 300, m3p=True)
    run_imp_sft_case(os.path.join(HERE, "imp_sft.npz"))


if __name__ == "__main__":
    main()

"""Shared test helpers: load golden fixtures, rebuild their inputs / configs / weights."""
import json
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

TASK_CFG = {"TASK15": {"type": "VL-classifier-GQA", "num_labels": 1842, "process": "normal",
                       "semantic_lambda": 10, "loss": "CrossEntropyLoss"}}


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name))
    return {k: z[k] for k in z.files}


def golden_config(g, m3p=False):
    from clg_vqa_amd.config import BertConfig, M3PConfig
    return (M3PConfig if m3p else BertConfig).from_dict(json.loads(bytes(g["cfg_json"]).decode()))


def golden_batch(g):
    """Rebuild the 10-tuple batch (reference layout) from a model fixture."""
    B = g["question"].shape[0]
    target = torch.zeros(B, 1842)
    target[torch.arange(B), torch.from_numpy(g["label"].astype(np.int64))] = 1.0
    t = torch.from_numpy
    return (t(g["features"].astype(np.float32)), t(g["spatials"].astype(np.float32)),
            t(g["image_mask"].astype(np.int64)), t(g["question"].astype(np.int64)), target,
            t(g["input_mask"].astype(np.int64)), torch.zeros(B, g["question"].shape[1], dtype=torch.int64),
            torch.arange(B), torch.arange(B), t(g["distances"].astype(np.float32)))


def grad_digest(g):
    f = g.detach().reshape(-1).double().cpu()
    return np.concatenate([f[:256].numpy(), [f.sum().item(), f.abs().sum().item(), (f * f).sum().sqrt().item()]])


def uc2_cfg_dict(hidden=768, heads=12, inter=3072, n_layers=12, vocab=250002):
    """UC2 config (volta/config/uc2_base.json values) with a configurable width / depth / vocab."""
    n_sub = 2 * n_layers
    cfg = dict(attention_probs_dropout_prob=0.1, hidden_act="gelu", hidden_dropout_prob=0.1,
               hidden_size=hidden, initializer_range=0.02, intermediate_size=inter,
               max_position_embeddings=514, num_attention_heads=heads, pooler_size=hidden,
               type_vocab_size=2, vocab_size=vocab, pad_token_id=1, num_locs=7, add_global_imgfeat=None,
               image_embeddings="uc2", model="roberta", v_attention_probs_dropout_prob=0.1,
               v_hidden_act="gelu", v_hidden_dropout_prob=0.1, v_feature_size=2048,
               v_hidden_size=hidden, v_initializer_range=0.02, v_pooler_size=1024,
               v_num_attention_heads=heads, v_intermediate_size=inter, layer_norm_eps=1e-5,
               fusion_method="text", clf_hidden_size=hidden)
    for k in ("tt_attn_sublayers", "tv_attn_sublayers", "vt_attn_sublayers", "vv_attn_sublayers"):
        cfg[k] = list(range(0, n_sub, 2))
    for k in ("t_ff_sublayers", "v_ff_sublayers"):
        cfg[k] = list(range(1, n_sub, 2))
    for k in ("shared_sublayers", "single_ln_sublayers"):
        cfg[k] = list(range(n_sub))
    return cfg


def check_imp_contract(w_flat, mask_before, mask_after, k, expect_tie_group=None):
    """What global magnitude pruning defines without reference to an implementation (see clg_vqa_amd/sft.py):
    exactly k newly pruned entries, every previously unmasked entry with |w| strictly below the threshold value T
    pruned, none above T pruned; entries equal to T form the tie group, of which any k - (#below T) may be pruned.
    Returns (T, tie group indices, pruned members of the tie group).  numpy arrays, fp32 values."""
    w = np.abs(np.asarray(w_flat, dtype=np.float32) * np.asarray(mask_before, dtype=np.float32))
    before = np.asarray(mask_before) == 1
    after = np.asarray(mask_after) == 1
    assert not (after & ~before).any(), "a pruned entry came back"
    new = before & ~after
    assert int(new.sum()) == int(k), (int(new.sum()), int(k))
    T = np.partition(w[before], k - 1)[k - 1]  # k-th smallest remaining magnitude
    assert new[before & (w < T)].all(), "an entry strictly below the threshold survived"
    assert not new[before & (w > T)].any(), "an entry strictly above the threshold was pruned"
    ties = np.nonzero(before & (w == T))[0]
    if expect_tie_group is not None:
        np.testing.assert_array_equal(ties, expect_tie_group)
    return T, ties, ties[new[ties]]

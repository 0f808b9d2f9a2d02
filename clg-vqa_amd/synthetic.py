"""Synthetic GQA batches and seeded weights (SURVEY.md §8d "Synthetic inputs").

There is no dataset, tokenizer or checkpoint in the build/bench environment, so every test and
benchmark runs on batches of the reference's *shape*: the 10-tuple produced at
``volta/volta/datasets/gqa_dataset_semantic_code_mix.py:440-452`` and consumed at
``volta/volta/task_utils.py:315``:

    (features, spatials, image_mask, question, target, input_mask, segment_ids, question_id, ix, distances)

The generator is numpy ``RandomState`` based (bit-stable across numpy/torch versions) so the golden
fixtures under ``tests/golden`` can be regenerated anywhere.
"""
import zlib

import numpy as np
import torch

PAD_ID = 1  # XLM-R <pad>; <s> = 0, </s> = 2


def make_batch(batch_size, seq_len=20, num_boxes=36, num_labels=1842, vocab_size=250002,
               num_locs=7, feat_dim=2048, seed=1234, l2_normalize=False, full_length=False,
               fp16_exact=False):
    """One synthetic batch, CPU tensors.

    question: random ids in [5, vocab), <s> at 0, </s> at len-1, pad(1) after; len ~ U{6..T}.
    features: relu(randn)*1.5 (RoI features are post-ReLU, sparse-ish, non-negative).
    spatials: [x1,y1,x2,y2,w,h,w*h] (UC2, num_locs=7) or [x1,y1,x2,y2,w*h] (M3P, num_locs=5).
    target: one-hot(label)*1.0 ; distances: U(0,1) with 0 at the gold label.
    ``fp16_exact`` rounds features/spatials/distances to fp16-representable values (fixtures store
    them as fp16 without loss).
    """
    rs = np.random.RandomState(seed)
    B, T, V = batch_size, seq_len, num_boxes
    q = rs.randint(5, vocab_size, size=(B, T)).astype(np.int64)
    lens = np.full((B,), T) if full_length else rs.randint(min(6, T), T + 1, size=(B,))
    for b in range(B):
        q[b, 0] = 0
        q[b, lens[b] - 1] = 2
        q[b, lens[b]:] = PAD_ID
    input_mask = (q != PAD_ID).astype(np.int64)
    segment_ids = np.zeros((B, T), dtype=np.int64)
    feats = (np.maximum(rs.randn(B, V, feat_dim), 0.0) * 1.5).astype(np.float32)
    x1 = rs.uniform(0, 0.7, size=(B, V)); y1 = rs.uniform(0, 0.7, size=(B, V))
    w = rs.uniform(0.05, 0.3, size=(B, V)); h = rs.uniform(0.05, 0.3, size=(B, V))
    if num_locs == 7:
        loc = np.stack([x1, y1, x1 + w, y1 + h, w, h, w * h], axis=-1)
    elif num_locs == 5:
        loc = np.stack([x1, y1, x1 + w, y1 + h, w * h], axis=-1)
    else:
        raise ValueError("num_locs must be 5 (M3P) or 7 (UC2)")
    loc = loc.astype(np.float32)
    if l2_normalize:  # M3P norm_embeddings (gqa_dataset_semantic_code_mix.py:608-611)
        feats = feats / np.linalg.norm(feats, axis=-1, keepdims=True)
        loc = loc / np.linalg.norm(loc, axis=-1, keepdims=True)
    image_mask = np.ones((B, V), dtype=np.int64)
    labels = rs.randint(0, num_labels, size=(B,))
    target = np.zeros((B, num_labels), dtype=np.float32)
    target[np.arange(B), labels] = 1.0
    dist = rs.uniform(0, 1, size=(B, num_labels)).astype(np.float32)
    dist[np.arange(B), labels] = 0.0
    if fp16_exact:
        feats = feats.astype(np.float16).astype(np.float32)
        loc = loc.astype(np.float16).astype(np.float32)
        dist = dist.astype(np.float16).astype(np.float32)
    t = torch.from_numpy
    return (t(feats), t(loc), t(image_mask), t(q), t(target), t(input_mask), t(segment_ids),
            torch.arange(B, dtype=torch.int64), torch.arange(B, dtype=torch.int64), t(dist))


def seeded_tensor(name, shape, seed=0, std=0.02):
    """Deterministic N(0, std) tensor that depends only on (name, shape, seed)."""
    rs = np.random.RandomState((zlib.crc32(name.encode()) + 7919 * seed) & 0x7FFFFFFF)
    return torch.from_numpy((rs.randn(*shape) * std).astype(np.float32))


def seeded_state_dict(template_state_dict, seed=0, std=0.02):
    """Seeded weights for any module tree using the reference's parameter names.

    Linear / embedding weights ~ N(0, std) (reference ``init_weights``, ``encoders.py:908-919``);
    LayerNorm weights get 1 + N(0, std) and biases N(0, std) so that every affine term and bias path
    is exercised by parity tests (a fresh reference model has gamma=1, beta=0, bias=0 which would
    hide errors there).  Aliased keys (``v_query`` ...) receive the tensor of their canonical name.
    """
    out = {}
    for key, ref in template_state_dict.items():
        canon = canonical_key(key)
        val = seeded_tensor(canon, tuple(ref.shape), seed=seed, std=std)
        if ref.dim() == 1 and canon.endswith("weight"):  # every 1-D "weight" on this path is a LayerNorm gamma
            val = val + 1.0
        out[key] = val.to(ref.dtype) if ref.dtype.is_floating_point else ref.clone()
    return out


_ALIASES = (
    (".attention_self.v_query.", ".attention_self.query."),
    (".attention_self.v_key.", ".attention_self.key."),
    (".attention_self.v_value.", ".attention_self.value."),
    (".attention_output.v_dense.", ".attention_output.dense."),
    (".attention_output.v_LayerNorm.", ".attention_output.LayerNorm."),
    (".intermediate.v_dense.", ".intermediate.dense."),
    (".output.v_dense.", ".output.dense."),
    (".output.v_LayerNorm.", ".output.LayerNorm."),
    ("embeddings.image_token_type_embeddings.", "embeddings.new_token_type_embeddings."),
)


def canonical_key(key):
    """Map the reference's duplicated alias keys onto the shared module's key (SURVEY.md §5:
    408 state_dict keys, 215 unique tensors for UC2)."""
    for a, c in _ALIASES:
        if a in key:
            return key.replace(a, c)
    return key

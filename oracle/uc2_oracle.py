"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product path (clg_vqa_amd/*).

CPU restatement (plain PyTorch, fp32, eager) of the reference's UC2 VQA fine-tuning hot path, in
the single-stream form of SURVEY.md Appendix A.  It exists to (1) be checked against the imported
reference itself (tests/golden/make_golden.py generates the fixtures from the real
``volta.encoders.BertForVLTasks`` in the build container; tests/test_oracle_golden.py re-checks the
restatement against them anywhere), and (2) be the checker for the HIP path in ``tests/ -m gpu``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``.

Parity status: PINNED by fixtures generated from the reference (tests/golden/*.npz), for logits,
loss, score and gradients.  The reference itself ships no tests for this path.

Every function cites the reference lines it restates (paths relative to /root/reference/volta).
"""
import math

import torch
import torch.nn.functional as F
from torch import nn


# --------------------------------------------------------------------------------------------- #
# primitives
# --------------------------------------------------------------------------------------------- #
def gelu_erf(x):
    """volta/encoders.py:131-137 -- exact erf GELU (not the tanh approximation)."""
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


class TFLayerNorm(nn.Module):
    """volta/encoders.py:49-62 -- biased variance, epsilon inside the sqrt (the reference's
    fallback when apex is absent, which is also what apex FusedLayerNorm computes)."""

    def __init__(self, hidden_size, eps):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.bias = nn.Parameter(torch.zeros(hidden_size))
        self.variance_epsilon = eps

    def forward(self, x):
        u = x.mean(-1, keepdim=True)
        s = (x - u).pow(2).mean(-1, keepdim=True)
        return self.weight * ((x - u) / torch.sqrt(s + self.variance_epsilon)) + self.bias


def roberta_position_ids(input_ids, padding_idx):
    """volta/embeddings.py:157-170 -- pos = cumsum(ids != pad) * (ids != pad) + pad."""
    mask = input_ids.ne(padding_idx).int()
    return (torch.cumsum(mask, dim=1).type_as(mask) * mask).long() + padding_idx


# --------------------------------------------------------------------------------------------- #
# module tree with the reference's parameter names (incl. the aliased v_* registrations)
# --------------------------------------------------------------------------------------------- #
class _Embeddings(nn.Module):
    """volta/embeddings.py:605-669 (UC2Embeddings)."""

    def __init__(self, c):
        super().__init__()
        self.padding_idx = c.pad_token_id
        self.word_embeddings = nn.Embedding(c.vocab_size, c.hidden_size, padding_idx=c.pad_token_id)
        self.position_embeddings = nn.Embedding(c.max_position_embeddings, c.hidden_size)
        self.new_token_type_embeddings = nn.Embedding(c.type_vocab_size, c.hidden_size)
        self.LayerNorm = TFLayerNorm(c.hidden_size, c.layer_norm_eps)
        self.image_embeddings = nn.Linear(c.v_feature_size, c.v_hidden_size)
        self.image_location_embeddings = nn.Linear(c.num_locs, c.v_hidden_size)
        self.image_token_type_embeddings = self.new_token_type_embeddings  # alias (:628)
        self.image_layer_norm = TFLayerNorm(c.hidden_size, c.layer_norm_eps)
        self.image_location_layer_norm = TFLayerNorm(c.hidden_size, c.layer_norm_eps)
        self.v_LayerNorm = TFLayerNorm(c.hidden_size, c.layer_norm_eps)
        self.p = c.hidden_dropout_prob

    def forward(self, token_ids, image_feat, image_loc, token_type_ids):
        pos = roberta_position_ids(token_ids, self.padding_idx)
        e = self.word_embeddings(token_ids) + self.position_embeddings(pos) \
            + self.new_token_type_embeddings(token_type_ids)
        e = F.dropout(self.LayerNorm(e), self.p, self.training)
        img = self.image_layer_norm(self.image_embeddings(image_feat))
        loc = self.image_location_layer_norm(self.image_location_embeddings(image_loc))
        ones = torch.ones_like(image_feat[:, :, 0].long())
        v = self.v_LayerNorm(img + loc + self.image_token_type_embeddings(ones))
        v = F.dropout(v, self.p, self.training)
        return e, v


class _SelfAttention(nn.Module):
    def __init__(self, c):
        super().__init__()
        H = c.hidden_size
        self.query = nn.Linear(H, H)
        self.key = nn.Linear(H, H)
        self.value = nn.Linear(H, H)
        self.v_query, self.v_key, self.v_value = self.query, self.key, self.value  # encoders.py:209-214


class _SelfOutput(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.dense = nn.Linear(c.hidden_size, c.hidden_size)
        self.LayerNorm = TFLayerNorm(c.hidden_size, c.layer_norm_eps)
        self.v_dense, self.v_LayerNorm = self.dense, self.LayerNorm  # encoders.py:385-389


class _AttnSublayer(nn.Module):
    """volta/encoders.py:164-359 (+362-425).  With has_tt=tv=vt=vv and shared weights the four
    gated score blocks + two concatenated softmaxes are exactly one multi-head attention over
    X=[text;vision] with the additive key mask [t_mask;v_mask]."""

    def __init__(self, c):
        super().__init__()
        self.attention_self = _SelfAttention(c)
        self.attention_output = _SelfOutput(c)
        self.nh = c.num_attention_heads
        self.p_attn = c.attention_probs_dropout_prob
        self.p_hid = c.hidden_dropout_prob

    def forward(self, x, add_mask):
        B, S, H = x.shape
        dh = H // self.nh
        a = self.attention_self

        def heads(t):
            return t.view(B, S, self.nh, dh).permute(0, 2, 1, 3)

        q, k, v = heads(a.query(x)), heads(a.key(x)), heads(a.value(x))
        scores = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(dh) + add_mask  # [B,nh,S,S]
        probs = F.dropout(torch.softmax(scores, dim=-1), self.p_attn, self.training)
        ctx = torch.matmul(probs, v).permute(0, 2, 1, 3).reshape(B, S, H)
        o = self.attention_output
        return o.LayerNorm(F.dropout(o.dense(ctx), self.p_hid, self.training) + x)


class _Intermediate(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.dense = nn.Linear(c.hidden_size, c.intermediate_size)
        self.v_dense = self.dense


class _Output(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.dense = nn.Linear(c.intermediate_size, c.hidden_size)
        self.LayerNorm = TFLayerNorm(c.hidden_size, c.layer_norm_eps)
        self.v_dense, self.v_LayerNorm = self.dense, self.LayerNorm


class _FFSublayer(nn.Module):
    """volta/encoders.py:453-502 (intermediate) + 505-567 (output)."""

    def __init__(self, c):
        super().__init__()
        self.intermediate = _Intermediate(c)
        self.output = _Output(c)
        self.p_hid = c.hidden_dropout_prob

    def forward(self, x):
        h = gelu_erf(self.intermediate.dense(x))
        o = self.output
        return o.LayerNorm(F.dropout(o.dense(h), self.p_hid, self.training) + x)


class _Encoder(nn.Module):
    """volta/encoders.py:821-892: sub-layers alternate attention (even) / feed-forward (odd)."""

    def __init__(self, c, n_layers):
        super().__init__()
        subs = []
        for _ in range(n_layers):
            subs += [_AttnSublayer(c), _FFSublayer(c)]
        self.layer = nn.ModuleList(subs)


class _Pooler(nn.Module):
    """volta/encoders.py:597-608; act = ReLU for UC2 (config.py:254 default fusion_act)."""

    def __init__(self, c):
        super().__init__()
        self.dense = nn.Linear(c.hidden_size, c.pooler_size)
        self.relu = (c.fusion_act == "relu")

    def forward(self, x):
        y = self.dense(x[:, 0])
        return torch.relu(y) if self.relu else torch.tanh(y)


class _Bert(nn.Module):
    def __init__(self, c, n_layers):
        super().__init__()
        self.embeddings = _Embeddings(c)
        self.encoder = _Encoder(c, n_layers)
        self.t_pooler = _Pooler(c)


class _GeLU(nn.Module):
    def forward(self, x):
        return gelu_erf(x)


class _Classifier(nn.Module):
    """volta/encoders.py:788-815: Linear -> GeLU -> LN(eps=layer_norm_eps) -> Linear."""

    def __init__(self, in_dim, hid, out, eps):
        super().__init__()
        self.logit_fc = nn.Sequential(nn.Linear(in_dim, hid), _GeLU(), TFLayerNorm(hid, eps), nn.Linear(hid, out))

    def forward(self, x):
        return self.logit_fc(x)


class OracleUC2ForVLTasks(nn.Module):
    """Restates ``BertForVLTasks`` (volta/encoders.py:1154-1259) for config.image_embeddings=="uc2",
    fusion_method=="text", task type VL-classifier-GQA.  state_dict keys are identical to the
    reference's (408 keys for the full config, incl. the duplicated alias keys)."""

    def __init__(self, config, task_cfg, task_ids, dropout_prob=0.1):
        super().__init__()
        n_layers = len(config.tt_attn_sublayers)
        assert (list(config.tt_attn_sublayers) == list(range(0, 2 * n_layers, 2))
                and list(config.t_ff_sublayers) == list(range(1, 2 * n_layers, 2))), "UC2 topology only"
        self.config = config
        self.task_cfg = task_cfg
        self.bert = _Bert(config, n_layers)
        self.p_pool = dropout_prob
        self.clfs_dict = nn.ModuleDict({
            tid: _Classifier(config.pooler_size, config.clf_hidden_size,
                             task_cfg[tid]["num_labels"], config.layer_norm_eps) for tid in task_ids})

    def forward(self, input_txt, input_imgs, image_loc, task_id, token_type_ids=None, attention_mask=None,
                image_attention_mask=None, output_all_encoded_layers=False, output_all_attention_masks=False):
        # BertModel.forward, encoders.py:958-1021
        if attention_mask is None:
            attention_mask = torch.ones_like(input_txt)
        if token_type_ids is None:
            token_type_ids = torch.zeros_like(input_txt)
        if image_attention_mask is None:
            image_attention_mask = torch.ones(input_imgs.size(0), input_imgs.size(1)).type_as(input_txt)
        e, v = self.bert.embeddings(input_txt, input_imgs, image_loc, token_type_ids)
        x = torch.cat([e, v], dim=1)                                   # single stream, text first
        m = torch.cat([attention_mask, image_attention_mask], dim=1).to(x.dtype)
        add_mask = ((1.0 - m) * -10000.0)[:, None, None, :]            # encoders.py:978-995
        for i, layer in enumerate(self.bert.encoder.layer):
            x = layer(x, add_mask) if i % 2 == 0 else layer(x)
        pooled = self.bert.t_pooler(x)                                 # token 0 is text (<s>)
        pooled = F.dropout(pooled, self.p_pool, self.training)          # encoders.py:1238-1239
        logits = self.clfs_dict[task_id](pooled)                       # encoders.py:1254
        return logits, None, None, None


# --------------------------------------------------------------------------------------------- #
# loss / score glue
# --------------------------------------------------------------------------------------------- #
def compute_score_with_logits(logits, labels):
    """volta/task_utils.py:706-711."""
    idx = torch.max(logits, 1)[1]
    one_hots = torch.zeros_like(labels)
    one_hots.scatter_(1, idx.view(-1, 1), 1)
    return one_hots * labels


def gqa_train_loss(logits, target, distances, semantic_lambda=10.0, topk=10):
    """volta/task_utils.py:413-428 (type VL-classifier-GQA, train):
    CE(logits, argmax target) * C + lambda * mean_b(sum_k p_topk * dist[b, idx_topk]) * C."""
    C = target.size(1)
    p_top_k, idx_top_k = torch.topk(F.softmax(logits, dim=-1), k=topk)
    sem = p_top_k * distances[torch.arange(distances.size(0)).unsqueeze(1), idx_top_k]
    sem = torch.mean(torch.sum(sem, dim=-1), dim=0)
    loss = F.cross_entropy(logits, torch.argmax(target.long(), dim=1)).mean() * C
    loss = loss + (semantic_lambda * sem.mean()) * C
    score = compute_score_with_logits(logits, target).sum() / float(logits.size(0))
    return loss, score


def gqa_val_loss(logits, target):
    """volta/task_utils.py:265-269 (val: CE * C only; score is a sum, not a mean)."""
    loss = F.cross_entropy(logits, torch.argmax(target.long(), dim=1)).mean() * target.size(1)
    return loss, compute_score_with_logits(logits, target).sum()


def forward_train(model, batch, task_id="TASK15", semantic_lambda=10.0):
    """volta/task_utils.py:308-428 restricted to the VL-classifier-GQA branch; positional call
    model(question, features, spatials, task_id, segment_ids, input_mask, image_mask) as at :403."""
    features, spatials, image_mask, question, target, input_mask, segment_ids, _qid, _ix, distances = batch
    logits = model(question, features, spatials, task_id, segment_ids, input_mask, image_mask)[0]
    loss, score = gqa_train_loss(logits, target, distances, semantic_lambda)
    return loss, score, logits


# --------------------------------------------------------------------------------------------- #
# sparse fine-tuning: IMP mask generation and mask application
# --------------------------------------------------------------------------------------------- #
def uc2_prunable_names(n_sublayers=24, pooler=True):
    """Module-name list of volta/train_task_prunning.py:45-63 == train_task_sft.py:44-85, in
    ``named_modules()`` order (query,key,value,attention_output.dense per even sub-layer;
    intermediate.dense, output.dense per odd one; t_pooler.dense last)."""
    names = []
    for ii in range(n_sublayers):
        if ii % 2 == 0:
            names += ["bert.encoder.layer.%d.attention_self.%s" % (ii, s) for s in ("query", "key", "value")]
            names += ["bert.encoder.layer.%d.attention_output.dense" % ii]
        else:
            names += ["bert.encoder.layer.%d.intermediate.dense" % ii, "bert.encoder.layer.%d.output.dense" % ii]
    if pooler:
        names.append("bert.t_pooler.dense")
    return names


def imp_round(weights, masks, amount=0.1):
    """One round of ``prune.global_unstructured(..., L1Unstructured, amount)`` as issued at
    volta/train_task_prunning.py:80-84, restating torch ``nn/utils/prune.py``:
    ``global_unstructured`` (:1095-1151) concatenates ``module.weight`` (= orig*mask after round 1)
    and the existing masks; ``PruningContainer.compute_mask`` (:315-409) restricts to entries with
    mask==1; ``L1Unstructured.compute_mask`` (:514-534) zeroes the k = round(amount * n_remaining)
    smallest |w| found by ``torch.topk(largest=False)``.

    weights / masks: lists of same-shaped fp32 tensors (masks hold {0,1}).  Returns the new masks.
    """
    t = torch.cat([(w * m).reshape(-1) for w, m in zip(weights, masks)])
    mask = torch.cat([m.reshape(-1) for m in masks]).clone()
    slc = mask == 1
    sub = t[slc]
    k = round(amount * sub.nelement())
    if k != 0:
        topk = torch.topk(torch.abs(sub).view(-1), k=k, largest=False)
        part = torch.ones_like(sub)
        part[topk.indices] = 0
        mask[slc] = part
    out, ptr = [], 0
    for w in weights:
        out.append(mask[ptr:ptr + w.numel()].view_as(w).clone())
        ptr += w.numel()
    return out


def sft_apply(weight_orig, mask):
    """torch prune forward pre-hook (prune.py:20-31): weight = weight_orig * weight_mask; the
    autograd of that product is the "mask (*) grad" of north_star: dL/dweight_orig = dL/dweight * mask
    (train_task_sft.py:128-132 installs it via CustomFromMask.apply)."""
    return weight_orig * mask

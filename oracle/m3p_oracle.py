"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

CPU restatement (plain PyTorch fp32) of the reference's M3P VQA path: ``M3PForVLTasks.forward``
(volta/volta/encoders.py:1311-1353) -> ``M3PModel.forward`` (:1033-1042) -> ``M3PTransformerModel.jointfwd``
(volta/volta/m3p_transformer.py:877-964) with ``MultiHeadAttention`` (:127-210), ``TransformerFFN`` (:213-227),
``BertImageEmbeddings`` (:231-269), ``get_masks`` (:59-78), ``BertPooler`` (:548-560).  Only the modules that
``jointfwd`` touches are built (the reference also constructs ~93 M never-used parameters; the
product module mirrors those names, the oracle does not need them).

Parity status: PINNED by tests/golden/m3p_small.npz, produced from the real reference by tests/golden/make_golden.py.
"""
import math

import torch
import torch.nn.functional as F
from torch import nn

from .uc2_oracle import _Classifier, gelu_erf

N_MAX_POSITIONS = 514  # m3p_transformer.py


class _ImageEmbeddings(nn.Module):
    """m3p_transformer.py:231-269 (image_distbution_embeddings exists but is unused: input_dist is None)."""

    def __init__(self, dim, v_feat, num_locs, p):
        super().__init__()
        self.image_embeddings = nn.Linear(v_feat, dim)
        self.image_location_embeddings = nn.Linear(num_locs, dim)
        self.LayerNorm = nn.LayerNorm(dim, eps=1e-12)
        self.p = p

    def forward(self, feat, loc):
        return F.dropout(self.LayerNorm(self.image_embeddings(feat) + self.image_location_embeddings(loc)), self.p,
                         self.training)


class _MHA(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.q_lin, self.k_lin, self.v_lin, self.out_lin = (nn.Linear(dim, dim) for _ in range(4))


class _FFN(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.lin1, self.lin2 = nn.Linear(dim, 4 * dim), nn.Linear(4 * dim, dim)


class _Pooler(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.dense = nn.Linear(dim, dim)


class _Encoder(nn.Module):
    def __init__(self, c):
        super().__init__()
        D = c.emb_dim
        self.position_embeddings = nn.Embedding(N_MAX_POSITIONS, D)
        self.embeddings = nn.Embedding(c.n_words, D, padding_idx=c.pad_index)
        self.layer_norm_emb = nn.LayerNorm(D, eps=1e-12)
        self.image_embeddings = _ImageEmbeddings(D, c.v_feature_size, c.num_locs, c.dropout)
        self.attentions = nn.ModuleList([_MHA(D) for _ in range(c.n_layers)])
        self.layer_norm1 = nn.ModuleList([nn.LayerNorm(D, eps=1e-12) for _ in range(c.n_layers)])
        self.ffns = nn.ModuleList([_FFN(D) for _ in range(c.n_layers)])
        self.layer_norm2 = nn.ModuleList([nn.LayerNorm(D, eps=1e-12) for _ in range(c.n_layers)])
        self.pooled_layer = _Pooler(D)


class _M3PModel(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.encoder = _Encoder(c)
        self.pooler = self.encoder.pooled_layer  # encoders.py:1029 alias


class OracleM3PForVLTasks(nn.Module):
    def __init__(self, config, task_cfg, task_ids, dropout_prob=0.1):
        super().__init__()
        self.config = config
        self.bert = _M3PModel(config)
        self.p_pool = dropout_prob
        self.clfs_dict = nn.ModuleDict({
            tid: _Classifier(config.pooler_size, config.clf_hidden_size, task_cfg[tid]["num_labels"],
                             config.layer_norm_eps) for tid in task_ids})

    def forward(self, input_txt, input_imgs, image_loc, task_id, token_type_ids=None, attention_mask=None,
                image_attention_mask=None, output_all_encoded_layers=False, output_all_attention_masks=False):
        c, e = self.config, self.bert.encoder
        B, T = input_txt.shape
        V = input_imgs.shape[1]
        S, nh = T + V, c.n_heads
        dh = c.emb_dim // nh
        lens = attention_mask.sum(1) + image_attention_mask.sum(1)                      # encoders.py:1036-1037
        img = e.image_embeddings(input_imgs, image_loc)
        t = torch.cat([img, e.embeddings(input_txt)], dim=1)                            # [image ; text] (:926)
        t = t + e.position_embeddings(torch.arange(S))[None]                            # :929-933
        mask = (torch.arange(S)[None, :] < lens[:, None])                               # get_masks :59-78
        mf = mask.unsqueeze(-1).to(t.dtype)
        t = t * mf                                                                      # :937
        t = F.dropout(e.layer_norm_emb(t), c.dropout, self.training)
        for i in range(c.n_layers):
            a = e.attentions[i]

            def shape(x):
                return x.view(B, S, nh, dh).transpose(1, 2)

            q = shape(a.q_lin(t)) / math.sqrt(dh)                                       # :197
            scores = torch.matmul(q, shape(a.k_lin(t)).transpose(2, 3))
            scores = scores.masked_fill((mask == 0)[:, None, None, :], -float("inf"))   # :199-200
            w = F.dropout(F.softmax(scores.float(), dim=-1), c.attention_dropout, self.training)
            ctx = torch.matmul(w, shape(a.v_lin(t))).transpose(1, 2).contiguous().view(B, S, c.emb_dim)
            t = e.layer_norm1[i](t + F.dropout(a.out_lin(ctx), c.dropout, self.training))
            f = e.ffns[i]
            t = e.layer_norm2[i](t + F.dropout(f.lin2(gelu_erf(f.lin1(t))), c.dropout, self.training))
            t = t * mf                                                                  # :955
        pooled = torch.tanh(self.bert.pooler.dense(t[:, 0]))                            # token 0 = first image region
        pooled = F.dropout(pooled, self.p_pool, self.training)                          # encoders.py:1337
        return self.clfs_dict[task_id](pooled), None, None, None

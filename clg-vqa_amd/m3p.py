"""``M3PForVLTasks`` on the native MI355X engine.

Reference: ``M3PForVLTasks`` (volta/volta/encoders.py:1262-1353) -> ``M3PModel`` (:1024-1042) ->
``M3PTransformerModel.jointfwd`` (volta/volta/m3p_transformer.py:877-964).  Differences from UC2 that the engine honours:
stream order **[image ; text]** (:926), positions 0..S-1 over the concatenation (:929-933), one LayerNorm for the
region embedding ``LN(feat W + b + loc W + b)`` (:231-269, 5-d locations), multiplicative length masks ``tensor *=
mask`` after the embeddings and after every layer (:937, :955) with ``mask = arange(S) < len_img + len_txt``
(:59-78), key mask -inf (:199-200), LayerNorm eps 1e-12, tanh pooler on token 0 = the first image region
(:548-560), classifier hidden 1536.

The module tree reproduces the reference's state_dict (names, shapes, order) **including the modules jointfwd never
touches** (``refine_embeddings``, ``encoder_attn``, ``layer_norm15``, ``latent_transforms``, ``original_transforms``,
``cross_alignment``, ``pooled_layer2``, ``seq_relationship{,2}``, ``mrfr_dense``, ``transformer_obj``,
``cross_lang_embeddings``, ``image_distbution_embeddings``): 93.3 M parameters that receive no gradient but live in
checkpoints and in the M3P prune list (volta/train_task_sft.py:139-205).  They are plain parameter holders here.
"""
import torch
from torch import nn

from . import ops
from .config import M3PConfig
from .encoders import GeLU, PreTrainedModel, SimpleClassifier, VLLinear  # noqa: F401
from .head import TaskHead
from .engine import BF16, EPI_F32, EngineBase, LayerSpec, LayerStack, TrunkFunction, dw_gemm, linear_params

N_MAX_POSITIONS = 514


def _ln(dim):
    return nn.LayerNorm(dim, eps=1e-12)


class _Holder(nn.Module):
    """A module that only registers children (never called)."""

    def __init__(self, **children):
        super().__init__()
        for k, v in children.items():
            setattr(self, k, v)


class BertImageEmbeddings(nn.Module):
    """m3p_transformer.py:231-247."""

    def __init__(self, dim, v_feat=2048, num_locs=5):
        super().__init__()
        self.image_embeddings = VLLinear(v_feat, dim)
        self.image_distbution_embeddings = nn.Linear(1600, dim)  # unused: input_dist is None
        self.image_location_embeddings = VLLinear(num_locs, dim)
        self.LayerNorm = _ln(dim)


class MultiHeadAttention(nn.Module):
    """m3p_transformer.py:127-150 (parameters only)."""

    def __init__(self, dim):
        super().__init__()
        self.q_lin, self.k_lin, self.v_lin, self.out_lin = VLLinear(dim, dim), VLLinear(dim, dim), VLLinear(dim, dim), VLLinear(dim, dim)


class TransformerFFN(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.lin1, self.lin2 = VLLinear(dim, 4 * dim), VLLinear(4 * dim, dim)


class BertPooler(nn.Module):
    """m3p_transformer.py:548-560: tanh(W h[:,0] + b)."""

    def __init__(self, dim):
        super().__init__()
        self.dense = VLLinear(dim, dim)

    def forward(self, hidden_states):
        return torch.tanh(self.dense(hidden_states[:, 0].contiguous()))


def _refiner(dim, n_layers):
    """AoA_Refiner_Core parameter tree (never used by jointfwd with refine_image=False)."""
    layers = []
    for _ in range(n_layers):
        layers.append(_Holder(
            self_attn=_Holder(linears=nn.ModuleList([nn.Linear(dim, dim) for _ in range(3)]),
                              aoa_layer=nn.ModuleList([nn.Linear(2 * dim, 2 * dim)])),
            feed_forward=_Holder(lin1=nn.Linear(dim, 4 * dim), lin2=nn.Linear(4 * dim, dim)),
            sublayer=nn.ModuleList([_Holder(norm=_ln(dim)), _Holder(norm=_ln(dim))])))
    return _Holder(layers=nn.ModuleList(layers), norm=_ln(dim))


class M3PTransformerModel(nn.Module):
    """Parameter tree of m3p_transformer.py:609-728 in the reference's registration order."""

    def __init__(self, c):
        super().__init__()
        D, nl = c.emb_dim, c.n_layers
        self.position_embeddings = nn.Embedding(N_MAX_POSITIONS, D)
        if c.n_langs > 1:
            self.cross_lang_embeddings = nn.Embedding(c.n_langs, D)
        self.embeddings = nn.Embedding(c.n_words, D, padding_idx=c.pad_index)
        self.layer_norm_emb = _ln(D)
        self.image_embeddings = BertImageEmbeddings(D, c.v_feature_size, c.num_locs)
        self.refine_embeddings = _refiner(D, getattr(c, "refine_layers", 6))
        self.cross_alignment = _Holder(att_weight_c=nn.Linear(D, 1), att_weight_q=nn.Linear(D, 1),
                                       att_weight_cq=nn.Linear(D, 1), align_output=nn.Linear(D, D), layer_norm=_ln(D))
        self.attentions = nn.ModuleList([MultiHeadAttention(D) for _ in range(nl)])
        self.layer_norm1 = nn.ModuleList([_ln(D) for _ in range(nl)])
        self.ffns = nn.ModuleList([TransformerFFN(D) for _ in range(nl)])
        self.layer_norm2 = nn.ModuleList([_ln(D) for _ in range(nl)])
        self.layer_norm15 = nn.ModuleList([_ln(D) for _ in range(nl)])
        self.encoder_attn = nn.ModuleList([
            _Holder(q_lin=nn.Linear(D, D), k_lin=nn.Linear(D, D), v_lin=nn.Linear(D, D), out_lin=nn.Linear(D, D))
            for _ in range(nl)])
        self.latent_transforms = nn.ModuleList([
            _Holder(x_to_mu=nn.Linear(D, D), x_to_logvar=nn.Linear(D, D), out_dense=nn.Linear(2 * D, D)) for _ in range(2)])
        self.original_transforms = nn.ModuleList([
            _Holder(dense=nn.Linear(D, D), dense_mu=nn.Linear(D, D), LayerNorm=_ln(D)) for _ in range(2)])
        self.pooled_layer = BertPooler(D)
        self.seq_relationship = nn.Linear(D, 1)
        self.pooled_layer2 = _Holder(dense=nn.Linear(D, D))
        self.seq_relationship2 = nn.Linear(D, 1)
        self.mrfr_dense = nn.Linear(D, 2048)
        self.transformer_obj = _Holder(dense=nn.Linear(D, D), LayerNorm=_ln(D))


class M3PModel(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.encoder = M3PTransformerModel(config)
        self.pooler = self.encoder.pooled_layer  # encoders.py:1029


class M3PEngine(EngineBase):
    def __init__(self, model):
        self.model = model
        c = model.config
        self.H, self.nh, self.I = c.emb_dim, c.n_heads, 4 * c.emb_dim
        self.eps = 1e-12
        self._init_common(self.H, self.nh)
        e = model.bert.encoder
        specs = [LayerSpec(a.q_lin, a.k_lin, a.v_lin, a.out_lin, e.layer_norm1[i], e.ffns[i].lin1, e.ffns[i].lin2,
                           e.layer_norm2[i]) for i, a in enumerate(e.attentions)]
        self.stack = LayerStack(specs, self.H, self.nh, self.I, self.eps)

    def image_linear(self):
        return self.model.bert.encoder.image_embeddings.image_embeddings

    def head_linears(self):
        lins = [self.model.bert.pooler.dense]
        for clf in self.model.clfs_dict.values():
            lins += [clf.logit_fc[0], clf.logit_fc[3]]
        return lins

    def param_list(self):
        e = self.model.bert.encoder
        ie = e.image_embeddings
        ps = [e.position_embeddings.weight, e.embeddings.weight, e.layer_norm_emb.weight, e.layer_norm_emb.bias,
              linear_params(ie.image_embeddings)[0], ie.image_embeddings.bias, ie.image_location_embeddings.weight,
              ie.image_location_embeddings.bias, ie.LayerNorm.weight, ie.LayerNorm.bias]
        for sp in self.stack.specs:
            ps += sp.params()
        return ps

    def forward(self, ids, feats, locs, seg, tmask, imask, training, need_grad=True):
        c = self.model.config
        e = self.model.bert.encoder
        ie = e.image_embeddings
        dev = feats.device
        if not feats.is_cuda:
            raise RuntimeError("clg_vqa_amd: M3PForVLTasks runs on the MI355X only (no CPU path)")
        B, T = ids.shape
        V, F, L = feats.shape[1], feats.shape[2], locs.shape[2]
        S, H = T + V, self.H
        M, BT, BV = B * S, B * T, B * V
        p_hid = float(c.dropout) if training else 0.0
        p_att = float(c.attention_dropout) if training else 0.0
        seed0, seed = self.next_seed()
        pw = self.prepared(dev)
        ar = self.stack.arena(B, S, dev, need_grad)
        self._last_arena = ar
        f32 = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)  # noqa: E731
        b16 = lambda *s: torch.empty(*s, dtype=BF16, device=dev)  # noqa: E731
        ids = ids.contiguous()
        feats2, locs2 = feats.contiguous().view(BV, F), locs.contiguous().view(BV, L)
        # length mask of the [image ; text] stream: arange(S) < len_img + len_txt  (get_masks, :59-78)
        lens = tmask.sum(1) + imask.sum(1)
        mask = torch.arange(S, device=dev)[None, :] < lens[:, None]
        rowmask = mask.to(torch.float32)
        rm_img, rm_txt = rowmask[:, :V].contiguous().view(BV), rowmask[:, V:].contiguous().view(BT)
        ar.row_post.copy_(rowmask.contiguous().view(M))
        rowmask = ar.row_post
        ar.addmask.copy_(torch.where(mask, 0.0, float("-inf")).to(torch.float32).contiguous().view(M))
        pos = e.position_embeddings.weight.detach()
        ge, be = e.layer_norm_emb.weight.detach(), e.layer_norm_emb.bias.detach()

        x32, x_hi, x_lo = self.stack.input_buffers(ar)
        # image rows: dropout(LN(feat W + b + loc W + b)) + pos[v] -> * mask -> layer_norm_emb -> dropout
        f_hi, f_lo = b16(BV, F), b16(BV, F)
        ops.split_f32(feats2, f_hi, f_lo)
        z_i, z_l = f32(BV, H), f32(BV, H)
        ops.gemm_nt(f_hi, f_lo, pw["img"].hi, pw["img"].lo, BV, H, F, 3, EPI_F32, bias=pw["img"].bias, out32=z_i)
        ops.loc_linear_fwd(locs2, ie.image_location_embeddings.weight.detach(),
                           ie.image_location_embeddings.bias.detach(), z_l, BV, L, H)
        a32, mean_i, rstd_i = f32(BV, H), f32(BV), f32(BV)
        ops.ln_fwd(z_i, z_l, None, ie.LayerNorm.weight.detach(), ie.LayerNorm.bias.detach(), self.eps, a32, None, None,
                   mean_i, rstd_i, BV, H, p_post=p_hid, seed=seed(1))
        mean_2, rstd_2 = f32(BV), f32(BV)
        ops.ln_fwd(a32, None, pos[0:V], ge, be, self.eps, x32, x_hi, x_lo, mean_2, rstd_2, BV, H, group=V,
                   out_stride=S, out_off=0, p_post=p_hid, seed=seed(2), row_pre=rm_img)
        # text rows: embeddings(x) + pos[V + t] -> * mask -> layer_norm_emb -> dropout
        z_t, mean_t, rstd_t = f32(BT, H), f32(BT), f32(BT)
        ops.embed_gather_fwd(ids, e.embeddings.weight.detach(), z_t, BT, H)
        ops.ln_fwd(z_t, None, pos[V:V + T], ge, be, self.eps, x32, x_hi, x_lo, mean_t, rstd_t, BT, H, group=T,
                   out_stride=S, out_off=V, p_post=p_hid, seed=seed(3), row_pre=rm_txt)
        sv = dict(B=B, T=T, V=V, F=F, L=L, S=S, p_hid=p_hid, p_att=p_att, seed=seed, seed0=seed0, ids=ids, locs=locs2, pw=pw,
                  arena=ar, rowmask=rowmask, rm_img=rm_img, rm_txt=rm_txt, f_hi=f_hi, z1=z_i, mean_i=mean_i, rstd_i=rstd_i,
                  z2=a32, mean_2=mean_2, rstd_2=rstd_2, z_t=z_t, mean_t=mean_t, rstd_t=rstd_t)
        return self.stack.forward(ar, pw["layers"], p_hid, p_att, seed0, row_post=rowmask), sv

    def backward(self, sv, dx):
        c = self.model.config
        e = self.model.bert.encoder
        ie = e.image_embeddings
        B, T, V, F, L, S = sv["B"], sv["T"], sv["V"], sv["F"], sv["L"], sv["S"]
        H = self.H
        M, BT, BV = B * S, B * T, B * V
        dev = dx.device
        p_hid, p_att, seed, pw = sv["p_hid"], sv["p_att"], sv["seed"], sv["pw"]
        f32 = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)  # noqa: E731
        ws = ops.ln_bwd_ws(M, H, dev)
        dy, layer_grads = self.stack.backward(sv["arena"], pw["layers"], dx.contiguous().view(-1, H), p_hid, p_att,
                                              sv["seed0"], row_post=sv["rowmask"])
        ge = e.layer_norm_emb.weight.detach()
        dpos = torch.zeros_like(e.position_embeddings.weight)
        sink = self.word_grad_sink
        use_sink = sink is not None and sink.shape == e.embeddings.weight.shape and sink.device == dev
        defer = use_sink and self.defer_word_grad
        dword = None if defer else (sink if use_sink else torch.zeros_like(e.embeddings.weight))
        # text rows
        dz_t, dg_t, db_t = f32(BT, H), f32(H), f32(H)
        ops.ln_bwd(dy, sv["z_t"], sv["mean_t"], sv["rstd_t"], ge, dz_t, None, None, dg_t, db_t, None, ws, BT, H, group=T,
                   out_stride=S, out_off=V, p_post=p_hid, seed=seed(3), row_pre=sv["rm_txt"])
        if defer:
            self._push_word_grad(sv["ids"].view(-1), dz_t, int(c.pad_index))
        else:
            if ops.DETERMINISTIC_EMBED_BWD and BT <= 16384:  # fixed summation order (csrc/scatter.hip)
                ops.scatter_add_det([(sv["ids"].view(-1), 0, dword, int(c.pad_index),
                                      self.word_row_flags if use_sink else None, T)], dz_t, BT, H)
            else:
                ops.embed_scatter_add(sv["ids"], dz_t, dword, BT, H, int(c.pad_index),
                                      row_flags=self.word_row_flags if use_sink else None)
        dpos[V:V + T] += dz_t.view(B, T, H).sum(0)
        # image rows
        dz2, dg_2, db_2 = f32(BV, H), f32(H), f32(H)
        ops.ln_bwd(dy, sv["z2"], sv["mean_2"], sv["rstd_2"], ge, dz2, None, None, dg_2, db_2, None, ws, BV, H, group=V,
                   out_stride=S, out_off=0, p_post=p_hid, seed=seed(2), row_pre=sv["rm_img"])
        dpos[0:V] += dz2.view(B, V, H).sum(0)
        dz1, dimg16 = f32(BV, H), torch.empty(BV, H, dtype=BF16, device=dev)
        dg_i, db_i, dbias_img = f32(H), f32(H), f32(H)
        ops.ln_bwd(dz2, sv["z1"], sv["mean_i"], sv["rstd_i"], ie.LayerNorm.weight.detach(), dz1, dimg16, None, dg_i,
                   db_i, dbias_img, ws, BV, H, p_post=p_hid, seed=seed(1))
        dWimg = dw_gemm(dimg16, sv["f_hi"], BV, H, F, mask=linear_params(ie.image_embeddings)[1])
        dWl = torch.zeros_like(ie.image_location_embeddings.weight)
        dbl = torch.zeros_like(ie.image_location_embeddings.bias)
        ops.loc_linear_bwd(sv["locs"], dz1, dWl, dbl, BV, L, H)
        grads = [dpos, None if use_sink else dword, dg_t + dg_2, db_t + db_2, dWimg, dbias_img, dWl, dbl, dg_i, db_i]
        for lg in layer_grads:
            grads += lg
        return grads


class M3PForVLTasks(PreTrainedModel):
    """Drop-in for volta.encoders.M3PForVLTasks (encoders.py:1262-1353) on the VL-classifier(-GQA) path."""

    config_class = M3PConfig

    def __init__(self, config, task_cfg, task_ids, dropout_prob=0.1):
        super().__init__(config)
        self.bert = M3PModel(config)
        self.dropout = nn.Dropout(dropout_prob)
        self.config = config
        self.task_cfg = task_cfg
        task2clf = {}
        for task_id in task_ids:
            task_type = task_cfg[task_id]["type"]
            if task_type in {"VL-classifier", "VL-classifier-GQA"}:
                task2clf[task_id] = SimpleClassifier(config.pooler_size, config.clf_hidden_size,
                                                     task_cfg[task_id]["num_labels"], config.layer_norm_eps)
            else:
                raise ValueError("clg_vqa_amd supports VL-classifier / VL-classifier-GQA heads only (got %s)" % task_type)
        self.clfs_dict = nn.ModuleDict(task2clf)
        self.fusion_method = config.fusion_method
        self.apply(self.init_weights)
        object.__setattr__(self, "_engine", M3PEngine(self))

    @property
    def engine(self):
        return self._engine

    def mark_weights_dirty(self):
        self._engine.mark_dirty()
        lins = self.__dict__.get("_vl_linears")
        if lins is None:
            lins = [m for m in self.modules() if isinstance(m, VLLinear)]
            object.__setattr__(self, "_vl_linears", lins)
        for m in lins:
            object.__setattr__(m, "_vl_dirty", True)

    def forward(self, input_txt, input_imgs, image_loc, task_id, token_type_ids=None, attention_mask=None,
                image_attention_mask=None, output_all_encoded_layers=False, output_all_attention_masks=False):
        if output_all_encoded_layers:
            raise NotImplementedError("output_all_encoded_layers for M3P")  # as the reference (encoders.py:1349)
        params = self._engine.param_list()
        self._engine.grad_mode = torch.is_grad_enabled()
        x = TrunkFunction.apply(self._engine, self.training, input_txt, input_imgs, image_loc, token_type_ids,
                                attention_mask, image_attention_mask, *params)
        head = self._task_head(task_id)
        if head.supported:
            return head(x, self.training), None, None, None
        pooled_output = self.dropout(self.bert.pooler(x))
        return self.clfs_dict[task_id](pooled_output), None, None, None

    def _task_head(self, task_id):
        heads = self.__dict__.setdefault("_vl_heads", {})
        if task_id not in heads:
            heads[task_id] = TaskHead(self._engine, self.bert.pooler.dense, "tanh", self.dropout, self.clfs_dict[task_id])
        return heads[task_id]

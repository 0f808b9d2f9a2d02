"""``BertForVLTasks`` for UC2 on the native MI355X engine -- the reference's call surface, parameter names and
checkpoint keys (volta/volta/encoders.py:1154-1259), with the compute in libvlhip.so.

The module tree reproduces the reference's ``state_dict`` exactly (408 keys / 215 tensors for the full UC2 config,
including the aliased ``v_query`` / ``v_dense`` / ``v_LayerNorm`` / ``image_token_type_embeddings`` keys), so
checkpoints, ``mask_best.pt`` files, the name-keyed optimizer grouping (volta/train_task.py:249-260) and
``torch.nn.utils.prune`` re-parametrisation by module name (volta/train_task_sft.py:122-132) all keep working.
Leaf modules are holders of parameters: the fused trunk reads ``weight`` (or ``weight_orig`` / ``weight_mask``)
directly instead of calling their ``forward``.

There is no CPU / eager fallback: calling the model with CPU tensors raises.
"""
import logging
import math
import os

import torch
from torch import nn

from . import ops
from .config import BertConfig, uc2_topology_check
from .engine import UC2Engine, UC2TrunkFunction, PreparedWeight, dw_gemm, linear_params, _ceil8
from .ops import BF16, EPI_F32

logger = logging.getLogger(__name__)
WEIGHTS_NAME = "pytorch_model.bin"


# ------------------------------------------------------------------------------------------------------------------
# native autograd ops used by the (tiny, M = batch) head
# ------------------------------------------------------------------------------------------------------------------
class NativeLinearFunction(torch.autograd.Function):
    """y = x W^T + b through vl_gemm_nt: forward 3-pass split bf16 (fp32-grade), backward bf16."""

    @staticmethod
    def forward(ctx, x, weight, bias, module):
        if not x.is_cuda:
            raise RuntimeError("clg_vqa_amd: native Linear needs device tensors (no CPU fallback)")
        pw = module._vl_prepared(x.device)
        M, K = x.shape
        N = weight.shape[0]
        x = x.contiguous()
        x_hi, x_lo = torch.empty(M, K, dtype=BF16, device=x.device), torch.empty(M, K, dtype=BF16, device=x.device)
        ops.split_f32(x, x_hi, x_lo)
        y = torch.empty(M, N, dtype=torch.float32, device=x.device)
        ops.gemm_nt(x_hi, x_lo, pw.hi, pw.lo, M, N, K, 3, EPI_F32, bias=bias.detach(), out32=y)
        ctx.save_for_backward(x_hi)
        ctx.pw, ctx.module = pw, module
        return y

    @staticmethod
    def backward(ctx, dy):
        (x_hi,) = ctx.saved_tensors
        pw = ctx.pw
        M, K = x_hi.shape
        N, Np = pw.N, pw.Np
        dy = dy.contiguous()
        dy16 = torch.zeros(M, Np, dtype=BF16, device=dy.device)
        dy16[:, :N] = dy
        dx = torch.empty(M, K, dtype=torch.float32, device=dy.device)
        ops.gemm_nt(dy16, None, pw.t_hi, None, M, K, Np, 1, EPI_F32, out32=dx)
        dw = dw_gemm(dy16[:, :N], x_hi, M, N, K, mask=linear_params(ctx.module)[1])
        return dx, dw, dy.sum(0), None


class NativeLayerNormFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        if not x.is_cuda:
            raise RuntimeError("clg_vqa_amd: native LayerNorm needs device tensors (no CPU fallback)")
        M, H = x.shape
        z = x.contiguous().clone()
        out = torch.empty_like(z)
        mean, rstd = torch.empty(M, device=x.device), torch.empty(M, device=x.device)
        ops.ln_fwd(z, None, None, gamma.detach(), beta.detach(), eps, out, None, None, mean, rstd, M, H)
        ctx.save_for_backward(z, mean, rstd, gamma)
        return out

    @staticmethod
    def backward(ctx, dy):
        z, mean, rstd, gamma = ctx.saved_tensors
        M, H = z.shape
        dz = torch.empty_like(z)
        dg, db = torch.empty(H, device=z.device), torch.empty(H, device=z.device)
        ops.ln_bwd(dy.contiguous(), z, mean, rstd, gamma.detach(), dz, None, None, dg, db, None,
                   ops.ln_bwd_ws(M, H, z.device), M, H)
        return dz, dg, db, None


# ------------------------------------------------------------------------------------------------------------------
# leaf modules (parameter holders with the reference's names)
# ------------------------------------------------------------------------------------------------------------------
class BertLayerNorm(nn.Module):
    """Reference: encoders.py:44-62 (apex FusedLayerNorm or the TF-style fallback)."""

    def __init__(self, hidden_size, eps=1e-12):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.bias = nn.Parameter(torch.zeros(hidden_size))
        self.variance_epsilon = eps

    def forward(self, x):
        shp = x.shape
        return NativeLayerNormFunction.apply(x.reshape(-1, shp[-1]), self.weight, self.bias,
                                             self.variance_epsilon).view(shp)


class VLLinear(nn.Linear):
    """nn.Linear holder; ``forward`` (used by the head only) runs the native GEMM."""

    def _vl_prepared(self, device):
        pw = self.__dict__.get("_vl_engine_pw")  # head Linears: prepared by the engine's per-step launch
        if pw is not None and pw.hi.device == device:
            return pw
        pw = getattr(self, "_vl_pw", None)
        if pw is None or pw.hi.device != device:
            pw = PreparedWeight([self], device)
            object.__setattr__(self, "_vl_pw", pw)
        pw.refresh(getattr(self, "_vl_dirty", False))
        object.__setattr__(self, "_vl_dirty", False)
        return pw

    def forward(self, x):
        w = linear_params(self)[0]
        shp = x.shape
        y = NativeLinearFunction.apply(x.reshape(-1, shp[-1]), w, self.bias, self)
        return y.view(*shp[:-1], self.out_features)


class GeLU(nn.Module):
    """erf-GELU (encoders.py:131-137) on the [B, clf_hidden] head activation."""

    def forward(self, x):
        return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


class UC2Embeddings(nn.Module):
    """Reference: volta/volta/embeddings.py:605-669 (parameters only; compute is in the fused trunk)."""

    def __init__(self, config):
        super().__init__()
        self.padding_idx = config.pad_token_id
        self.word_embeddings = nn.Embedding(config.vocab_size, config.hidden_size, padding_idx=self.padding_idx)
        self.position_embeddings = nn.Embedding(config.max_position_embeddings, config.hidden_size)
        self.new_token_type_embeddings = nn.Embedding(config.type_vocab_size, config.hidden_size)
        self.LayerNorm = BertLayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)
        self.image_embeddings = VLLinear(config.v_feature_size, config.v_hidden_size)
        self.image_location_embeddings = VLLinear(config.num_locs, config.v_hidden_size)
        self.image_token_type_embeddings = self.new_token_type_embeddings
        self.image_layer_norm = BertLayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.image_location_layer_norm = BertLayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.v_LayerNorm = BertLayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.v_dropout = nn.Dropout(config.hidden_dropout_prob)


class BertGatedSelfAttention(nn.Module):
    """Reference: encoders.py:164-219 (share_layer: v_* are the text modules)."""

    def __init__(self, config, layer_num):
        super().__init__()
        H = config.hidden_size
        self.num_attention_heads = config.num_attention_heads
        self.attention_head_size = H // config.num_attention_heads
        self.query, self.key, self.value = VLLinear(H, H), VLLinear(H, H), VLLinear(H, H)
        self.dropout = nn.Dropout(config.attention_probs_dropout_prob)
        self.v_query, self.v_key, self.v_value, self.v_dropout = self.query, self.key, self.value, self.dropout


class BertGatedSelfOutput(nn.Module):
    """Reference: encoders.py:362-397."""

    def __init__(self, config, layer_num):
        super().__init__()
        self.dense = VLLinear(config.hidden_size, config.hidden_size)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)
        self.LayerNorm = BertLayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.v_dense, self.v_dropout, self.v_LayerNorm = self.dense, self.dropout, self.LayerNorm


class BertGatedAttention(nn.Module):
    def __init__(self, config, layer_num):
        super().__init__()
        self.attention_self = BertGatedSelfAttention(config, layer_num)
        self.attention_output = BertGatedSelfOutput(config, layer_num)


class BertGatedIntermediate(nn.Module):
    """Reference: encoders.py:453-485."""

    def __init__(self, config, layer_num):
        super().__init__()
        self.dense = VLLinear(config.hidden_size, config.intermediate_size)
        self.v_dense = self.dense


class BertGatedOutput(nn.Module):
    """Reference: encoders.py:505-540."""

    def __init__(self, config, layer_num):
        super().__init__()
        self.dense = VLLinear(config.intermediate_size, config.hidden_size)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)
        self.LayerNorm = BertLayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.v_dense, self.v_dropout, self.v_LayerNorm = self.dense, self.dropout, self.LayerNorm


class BertGatedFeedForward(nn.Module):
    def __init__(self, config, layer_num):
        super().__init__()
        self.intermediate = BertGatedIntermediate(config, layer_num)
        self.output = BertGatedOutput(config, layer_num)


class BertEncoder(nn.Module):
    """Reference: encoders.py:821-846: attention on even sub-layer numbers, feed-forward on odd ones."""

    def __init__(self, config):
        super().__init__()
        n_layers = uc2_topology_check(config)
        subs = []
        for l in range(n_layers):
            subs += [BertGatedAttention(config, 2 * l), BertGatedFeedForward(config, 2 * l + 1)]
        self.layer = nn.ModuleList(subs)
        self.num2type = {i: ("attn" if i % 2 == 0 else "ff") for i in range(2 * n_layers)}


class BertTextPooler(nn.Module):
    """Reference: encoders.py:597-608; activation = ReLU unless config.fusion_act says otherwise."""

    def __init__(self, config):
        super().__init__()
        self.dense = VLLinear(config.hidden_size, config.pooler_size)
        self.activation = nn.ReLU() if config.fusion_act == "relu" else nn.Tanh()

    def forward(self, hidden_states):
        return self.activation(self.dense(hidden_states[:, 0].contiguous()))


class SimpleClassifier(nn.Module):
    """Reference: encoders.py:788-815."""

    def __init__(self, in_dim, hid_dim, out_dim, layer_norm_eps=1e-12, dropout_prob=0.0):
        super().__init__()
        self.logit_fc = nn.Sequential(VLLinear(in_dim, hid_dim), GeLU(), BertLayerNorm(hid_dim, eps=layer_norm_eps),
                                      VLLinear(hid_dim, out_dim))

    def forward(self, hidden_states):
        return self.logit_fc(hidden_states)


class BertModel(nn.Module):
    """Reference: encoders.py:922-1021 for image_embeddings == "uc2", fusion_method == "text"."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.shared_embeddings = True
        self.embeddings = UC2Embeddings(config)
        self.encoder = BertEncoder(config)
        self.fusion_method = config.fusion_method
        self.t_pooler = BertTextPooler(config)


class PreTrainedModel(nn.Module):
    """Loader surface of volta/volta/utils.py:250-580 restricted to local files (no network in scope)."""

    base_model_prefix = "bert"

    def __init__(self, config, *inputs, **kwargs):
        super().__init__()
        self.config = config

    def init_weights(self, module):
        """Reference: encoders.py:908-919."""
        if isinstance(module, (nn.Linear, nn.Embedding)):
            module.weight.data.normal_(mean=0.0, std=self.config.initializer_range)
        elif isinstance(module, BertLayerNorm):
            module.bias.data.zero_()
            module.weight.data.fill_(1.0)
        if isinstance(module, nn.Linear) and module.bias is not None:
            module.bias.data.zero_()

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, *model_args, **kwargs):
        """Local file or directory (+ ``pytorch_model.bin``); renames gamma/beta -> weight/bias and roberta. ->
        bert. like the reference (utils.py:462-518); returns the model in eval() mode (utils.py:570); returns None
        when the file is missing (utils.py:445).  ``state_dict=`` short-circuits the file read."""
        config = kwargs.pop("config", None)
        state_dict = kwargs.pop("state_dict", None)
        for k in ("cache_dir", "from_tf", "from_hf", "output_loading_info", "default_gpu"):
            kwargs.pop(k, None)
        assert config is not None
        path = pretrained_model_name_or_path
        if os.path.isdir(path):
            path = os.path.join(path, WEIGHTS_NAME)
        if state_dict is None and not os.path.isfile(path):
            logger.error("Model name '%s' was not found; we assumed it was a path but couldn't find any file", path)
            return None
        model = cls(config, *model_args, **kwargs)
        if state_dict is None:
            state_dict = torch.load(path, map_location="cpu", weights_only=True)
        renamed = {}
        for key, val in state_dict.items():
            nk = key.replace("gamma", "weight") if "gamma" in key else key
            nk = nk.replace("beta", "bias") if "beta" in nk else nk
            if getattr(config, "model", "bert") == "roberta":
                nk = nk.replace("roberta", "bert")
            renamed[nk] = val
        has_prefix = any(k.startswith(cls.base_model_prefix + ".") for k in renamed)
        target = model if has_prefix else getattr(model, cls.base_model_prefix)
        missing, unexpected = target.load_state_dict(renamed, strict=False)
        if missing:
            logger.info("Weights of %s not initialized from pretrained model: %s", cls.__name__, missing)
        if unexpected:
            logger.info("Weights from pretrained model not used in %s: %s", cls.__name__, unexpected)
        model.eval()
        return model


class BertForVLTasks(PreTrainedModel):
    """Drop-in for volta.encoders.BertForVLTasks (encoders.py:1154-1259) on the VL-classifier(-GQA) path."""

    config_class = BertConfig

    def __init__(self, config, task_cfg, task_ids, dropout_prob=0.1):
        super().__init__(config)
        self.bert = BertModel(config)
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
        object.__setattr__(self, "_engine", UC2Engine(self))

    @property
    def engine(self):
        return self._engine

    def mark_weights_dirty(self):
        """Tell the engine that parameters changed behind torch's back (raw-pointer optimizer update)."""
        self._engine.mark_dirty()
        lins = self.__dict__.get("_vl_linears")
        if lins is None:  # the module tree is static after construction: walk it once, not every optimizer step
            lins = [m for m in self.modules() if isinstance(m, VLLinear)]
            object.__setattr__(self, "_vl_linears", lins)
        for m in lins:
            object.__setattr__(m, "_vl_dirty", True)

    def forward(self, input_txt, input_imgs, image_loc, task_id, token_type_ids=None, attention_mask=None,
                image_attention_mask=None, output_all_encoded_layers=False, output_all_attention_masks=False):
        if output_all_encoded_layers or output_all_attention_masks:
            raise NotImplementedError("clg_vqa_amd: per-layer outputs / attention maps are not materialised by the "
                                      "fused trunk")
        if attention_mask is None:
            attention_mask = torch.ones_like(input_txt)
        if token_type_ids is None:
            token_type_ids = torch.zeros_like(input_txt)
        if image_attention_mask is None:
            image_attention_mask = torch.ones(input_imgs.size(0), input_imgs.size(1)).type_as(input_txt)
        params = self._engine.param_list()
        self._engine.grad_mode = torch.is_grad_enabled()
        x = UC2TrunkFunction.apply(self._engine, self.training, input_txt, input_imgs, image_loc, token_type_ids,
                                   attention_mask, image_attention_mask, *params)
        if self._engine.chunks_pending:  # a pipelined optimizer update: the heads' parameters are its last chunk
            self._engine.wait_chunk(self._engine.n_chunks() - 1)
            self._engine.chunks_pending = False
            self._engine.stack.layer_ready_events = None
        head = self._task_head(task_id)
        if head.supported:  # pooler -> dropout -> classifier as one native autograd node (head.py)
            vil_prediction = head(x, self.training)
        else:
            pooled_output_t = self.bert.t_pooler(x)
            pooled_output = self.dropout(pooled_output_t)  # fusion_method == "text" (encoders.py:1238-1239)
            vil_prediction = self.clfs_dict[task_id](pooled_output)
        return vil_prediction, None, None, ([None], [None])

    def _task_head(self, task_id):
        heads = self.__dict__.setdefault("_vl_heads", {})
        if task_id not in heads:
            from .head import TaskHead
            heads[task_id] = TaskHead(self._engine, self.bert.t_pooler.dense, self.config.fusion_act, self.dropout,
                                      self.clfs_dict[task_id])
        return heads[task_id]

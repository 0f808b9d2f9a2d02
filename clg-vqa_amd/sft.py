"""Sparse fine-tuning host logic: iterative magnitude pruning (mask generation) and mask application.

Mirrors of the reference's helpers, same names and argument meaning:
* ``pruning_model_uc2(model, px, ..., global_pruning=True)``  -- volta/train_task_prunning.py:45-91 (one IMP round)
* ``see_weight_rate_uc2(model)``                              -- :92-177 (% of pruned entries)
* ``rewind_uc2(pre_weight, model_prefix)``                    -- :179-256 (theta_0 snapshot with ``_orig`` key renames)
* ``pruning_model_custom(model, mask_dict, module)``          -- volta/train_task_sft.py:44-132 (apply mask_best.pt)

The module list (73 Linear weights, 85 524 480 elements for full UC2) and its ``named_modules()`` order are the
reference's.  The selection itself -- torch's ``prune.global_unstructured(L1Unstructured, amount)`` -- runs on the
device as an exact radix select (``vl_imp_select``, csrc/imp.hip); re-parametrisation (``weight_orig`` /
``weight_mask`` + forward pre-hook) is installed with ``torch.nn.utils.prune.CustomFromMask`` exactly like the
reference so that state_dict keys, ``mask_lt*.pt`` / ``mask_best.pt`` files and the reference's own scripts keep working.
"""
import torch
from torch.nn.utils import prune

from . import ops


def uc2_prunable_names(n_sublayers=24, embeddings=False, cls=False, task="TASK15"):
    names = []
    if embeddings:
        names += ["bert.embeddings.word_embeddings", "bert.embeddings.image_embeddings"]
    for ii in range(n_sublayers):
        if ii % 2 == 0:
            names += ["bert.encoder.layer.%d.attention_self.%s" % (ii, s) for s in ("query", "key", "value")]
            names.append("bert.encoder.layer.%d.attention_output.dense" % ii)
        if ii > 0 and ii % 2 == 1:
            names += ["bert.encoder.layer.%d.intermediate.dense" % ii, "bert.encoder.layer.%d.output.dense" % ii]
    names.append("bert.t_pooler.dense")
    if cls:
        names += ["clfs_dict.%s.logit_fc.%d" % (task, ii) for ii in (0, 2, 3)]
    return names


def _selected_modules(model, names):
    want = set(names)
    return [(n, m) for n, m in model.named_modules() if n in want]  # named_modules() order, like the reference


def _n_sublayers(model):
    return len(model.bert.encoder.layer)


def pruning_model_uc2(model, px, embeddings=False, global_pruning=True, cls=False, bias=False):
    """One round of global L1 magnitude pruning: the ``round(px * n_remaining)`` smallest |weight| among the still
    unmasked entries of the prunable modules get mask 0.  Returns the number of newly pruned entries."""
    if not global_pruning or bias:
        raise NotImplementedError("clg_vqa_amd: the reference runs global weight pruning only "
                                  "(train_task_prunning.py:723-726: bias=False, global_pruning=True)")
    mods = _selected_modules(model, uc2_prunable_names(_n_sublayers(model), embeddings, cls))
    ws, ms = [], []
    for _, m in mods:
        if "weight_orig" in m._parameters:
            ws.append(m._parameters["weight_orig"].detach().reshape(-1))
            ms.append(m._buffers["weight_mask"].reshape(-1))
        else:
            ws.append(m.weight.detach().reshape(-1))
            ms.append(torch.ones_like(ws[-1]))
    w_flat, m_flat = torch.cat(ws), torch.cat(ms)
    n_remaining = int(m_flat.sum().item())  # once per epoch; every mask entry is exactly 0 or 1
    k = round(px * n_remaining)             # torch prune.py:_compute_nparams_toprune
    new_flat = torch.empty_like(m_flat)
    ops.imp_select(w_flat, m_flat, new_flat, k)
    ptr = 0
    for _, m in mods:
        n = m._parameters["weight_orig"].numel() if "weight_orig" in m._parameters else m.weight.numel()
        new_mask = new_flat[ptr:ptr + n]
        ptr += n
        if "weight_orig" in m._parameters:
            m._buffers["weight_mask"].copy_(new_mask.view_as(m._buffers["weight_mask"]))
        else:
            prune.CustomFromMask.apply(m, "weight", mask=new_mask.view_as(m.weight).clone())
    if hasattr(model, "mark_weights_dirty"):
        model.mark_weights_dirty()
    return k


def see_weight_rate_uc2(model, embedding=False, cls=False, bias=False):
    total, zeros = 0.0, 0.0
    for _, m in _selected_modules(model, uc2_prunable_names(_n_sublayers(model), embedding, cls)):
        mask = m._buffers.get("weight_mask")
        if mask is None:
            total += m.weight.numel()
            continue
        total += float(mask.nelement())
        zeros += float(torch.sum(mask == 0))
    return 100 * zeros / total


def rewind_uc2(pre_weight, model="", embeddings=False, cls=False, bias=False, n_sublayers=24):
    """Snapshot theta_0 -> state_dict update that rewinds ``*.weight_orig`` (train_task_prunning.py:179-256,
    applied at :803-806 as ``model_dict.update(orig); load_state_dict``)."""
    name_list = set()
    for n in uc2_prunable_names(n_sublayers, embeddings, cls):
        name_list.add(model + n + ".weight")
        # aliased registrations of the shared modules
        for a, b in ((".attention_self.query", ".attention_self.v_query"), (".attention_self.key", ".attention_self.v_key"),
                     (".attention_self.value", ".attention_self.v_value"), (".attention_output.dense", ".attention_output.v_dense"),
                     (".intermediate.dense", ".intermediate.v_dense"), (".output.dense", ".output.v_dense")):
            if n.endswith(a) and not (a == ".output.dense" and n.endswith(".attention_output.dense")):
                name_list.add(model + n[:-len(a)] + b + ".weight")
    recover = {}
    for key, val in pre_weight.items():
        if "bert" in key or "clfs_dict" in key:
            recover[key + "_orig" if key in name_list else key] = val
    return recover


def pruning_model_custom(model, mask_dict, module="", embeddings=False, cls=False, bias=False):
    """Install the masks of a ``mask_best.pt`` dict (keys ``<module>....weight_mask``) by module name."""
    if bias:
        raise NotImplementedError("clg_vqa_amd: bias masks are not used by the reference scripts (bias=False)")
    names = uc2_prunable_names(_n_sublayers(model), embeddings, cls)
    for n, m in _selected_modules(model, names):
        mask = mask_dict["%s%s.weight_mask" % (module, n)].to(m.weight.device, torch.float32)
        prune.CustomFromMask.apply(m, "weight", mask=mask)
    if hasattr(model, "mark_weights_dirty"):
        model.mark_weights_dirty()


def premultiply_by_mask(model, mask_dict):
    """train_task_sft.py:432-453: weights are multiplied by their mask once before the re-parametrisation."""
    by_weight_key = {k.replace("_mask", ""): v for k, v in mask_dict.items()}
    sd = model.state_dict()
    new = {k: (v * by_weight_key[k].to(v.device) if k in by_weight_key else v) for k, v in sd.items()}
    model.load_state_dict(new)

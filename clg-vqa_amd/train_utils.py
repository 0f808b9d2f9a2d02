"""Checkpoint / mask file formats of the reference (volta/volta/train_utils.py:351-510) -- the compatibility
surface SURVEY.md §5 names: ``pytorch_model_{epoch}.bin``, ``pytorch_model_best.bin``, ``pytorch_ckpt_latest.tar``,
``save_sft``'s masked / unmasked pair, ``save_prunned``'s ``mask_lt{e}.pt`` / ``mask_best.pt``, ``resume``.
(The pickled tbLogger of the reference's ``.tar`` is replaced by ``None``: tensorboardX is out of scope.)"""
import os

import torch


def freeze_layers(model):
    """train_utils.py:305-311: every parameter whose name CONTAINS one of config.fixed_layers (e.g. "embeddings",
    "v_embeddings.LayerNorm", "encoder.layer.3.") stops receiving gradients."""
    fixed = [str(n) for n in (getattr(model.config, "fixed_layers", []) or [])]
    for key, value in dict(model.named_parameters()).items():
        for name in fixed:
            if name in key:
                value.requires_grad = False


def _rng_of(model):
    """(base_seed, calls) of the engine's counter-based dropout RNG: stored so that a resumed run continues the mask
    sequence instead of replaying it."""
    eng = getattr(model, "engine", None)
    return None if eng is None else [int(eng.base_seed), int(eng.calls)]


def _fold(state_dict, masked=True):
    out = {}
    for k, v in state_dict.items():
        if "_orig" in k:
            out[k.replace("_orig", "")] = v * state_dict[k.replace("_orig", "_mask")] if masked else v
        elif "_mask" in k:
            continue
        else:
            out[k] = v
    return out


def _ckpt(path, model_state, optimizer, global_step, epoch_id, score, extra=None, dropout_rng=None):
    d = {"model_state_dict": model_state, "optimizer_state_dict": optimizer.state_dict(),
         "scheduler_state_dict": {"last_epoch": getattr(optimizer, "sched_step", 0)}, "global_step": global_step,
         "epoch_id": epoch_id, "tb_logger": None, "score": score, "dropout_rng": dropout_rng}
    d.update(extra or {})
    torch.save(d, os.path.join(path, "pytorch_ckpt_latest.tar"))


def save(path, logger, epoch_id, model, optimizer, scheduler, global_step, tb_logger, default_gpu, score, is_best=False):
    if not default_gpu:
        return
    m = model.module if hasattr(model, "module") else model
    sd = m.state_dict()
    torch.save(sd, os.path.join(path, "pytorch_model_%s.bin" % epoch_id))
    if is_best:
        torch.save(sd, os.path.join(path, "pytorch_model_best.bin"))
    _ckpt(path, sd, optimizer, global_step, epoch_id, score, dropout_rng=_rng_of(m))


def save_sft(path, logger, epoch_id, model, optimizer, scheduler, global_step, tb_logger, default_gpu, score, is_best=False):
    if not default_gpu:
        return
    m = model.module if hasattr(model, "module") else model
    sd = m.state_dict()
    masked, unmasked = _fold(sd, True), _fold(sd, False)
    torch.save(masked, os.path.join(path, "pytorch_model_%s.bin" % epoch_id))
    torch.save(unmasked, os.path.join(path, "pytorch_model_unmasked%s.bin" % epoch_id))
    if is_best:
        torch.save(masked, os.path.join(path, "pytorch_model_best.bin"))
        torch.save(unmasked, os.path.join(path, "pytorch_model_unmasked_best.bin"))
    _ckpt(path, masked, optimizer, global_step, epoch_id, score, dropout_rng=_rng_of(m))


def save_prunned(path, logger, epoch_id, model, optimizer, scheduler, global_step, tb_logger, default_gpu, score,
                 is_best=False):
    if not default_gpu:
        return
    m = model.module if hasattr(model, "module") else model
    sd = m.state_dict()
    mask_dict = {k: v.cpu() for k, v in sd.items() if "mask" in k}
    torch.save(mask_dict, os.path.join(path, "mask_lt%s.pt" % epoch_id))
    masked = _fold(sd, True)
    torch.save(masked, os.path.join(path, "pytorch_model_%s.bin" % epoch_id))
    if is_best:
        torch.save(masked, os.path.join(path, "pytorch_model_best.bin"))
        torch.save(mask_dict, os.path.join(path, "mask_best.pt"))
    _ckpt(path, masked, optimizer, global_step, epoch_id, score, {"mask_dict": mask_dict}, dropout_rng=_rng_of(m))


def resume(path, model, optimizer, scheduler, tb_logger):
    start_iter_id, global_step, start_epoch, best_score = 0, 0, 0, float("-inf")
    if path != "" and os.path.exists(path):
        # everything _ckpt() writes (tensors, ints, floats, lists of str, None) loads with the weights-only loader; the
        # reference's own pytorch_ckpt_latest.tar pickles its tbLogger / optimizer objects and is refused here
        try:
            ck = torch.load(path, map_location="cpu", weights_only=True)
        except Exception as e:
            raise RuntimeError("clg_vqa_amd.resume: %s cannot be read with the weights-only loader (checkpoints written "
                               "by the reference pickle a tbLogger / optimizer objects and are not accepted; resume from "
                               "a pytorch_ckpt_latest.tar written by this package): %s" % (path, e)) from e
        sd = {(k.replace("module.", "", 1) if k.startswith("module.") else k): v for k, v in ck["model_state_dict"].items()}
        model.load_state_dict(sd)
        optimizer.load_state_dict(ck["optimizer_state_dict"])
        eng = getattr(model, "engine", None)
        if eng is not None and ck.get("dropout_rng") is not None:
            eng.base_seed, eng.calls = int(ck["dropout_rng"][0]), int(ck["dropout_rng"][1])
        global_step = ck["global_step"]
        start_epoch = int(ck["epoch_id"]) + 1
        best_score = ck.get("score", float("-inf"))
        if hasattr(model, "mark_weights_dirty"):
            model.mark_weights_dirty()
    return start_iter_id, global_step, start_epoch, tb_logger, best_score

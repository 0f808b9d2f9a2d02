"""Loss / score glue of the VQA (GQA) fine-tuning step -- the `VL-classifier-GQA` branches of
volta/volta/task_utils.py (``ForwardModelsTrain`` :308-428, ``ForwardModelsVal`` :195-269,
``compute_score_with_logits`` :706-711, ``LoadLoss`` / ``LossMap`` :179-189).

The loss operates on the [B, 1842] logits only (negligible work) and is written with torch
tensor ops on the device; everything upstream of the logits is the native engine.
"""
import torch
import torch.nn.functional as F
from torch import nn

LossMap = {
    "BCEWithLogitLoss": nn.BCEWithLogitsLoss(reduction="mean"),
    "CrossEntropyLoss": nn.CrossEntropyLoss(),
}


def LoadLoss(args, task_cfg, task_id):
    task = "TASK" + task_id
    loss_name = getattr(args, "loss", None) or task_cfg[task]["loss"]
    return LossMap[loss_name]


def compute_score_with_logits(logits, labels):
    idx = torch.max(logits, 1)[1]
    one_hots = torch.zeros_like(labels)
    one_hots.scatter_(1, idx.view(-1, 1), 1)
    return one_hots * labels


FUSED_GQA_LOSS = True  # A/B knob (bench.py BENCH_TORCH_LOSS=1): False = the eager torch arithmetic below


class GQALossFunction(torch.autograd.Function):
    """loss, score = GQA loss with semantic prior (task_utils.py:413-428) through vl_gqa_loss: one launch computes the
    loss, the batch score and d(loss)/d(logits); backward only scales the saved gradient."""

    @staticmethod
    def forward(ctx, logits, target, distances, semantic_lambda):
        from . import _lib, ops
        B = logits.shape[0]
        logits = logits.contiguous()
        out = torch.empty(2, dtype=torch.float32, device=logits.device)
        dlogits = torch.empty_like(logits)
        ws = torch.empty(_lib.lib().vl_gqa_loss_ws_bytes(B), dtype=torch.uint8, device=logits.device)
        ops.gqa_loss(logits, target.contiguous().float(), distances.contiguous().float(), semantic_lambda, out, dlogits, ws)
        ctx.save_for_backward(dlogits)
        loss, score = out[0], out[1]  # bind the views once: the mark has to be on the object that is returned
        ctx.mark_non_differentiable(score)
        return loss, score

    @staticmethod
    def backward(ctx, g_loss, g_score):
        (dlogits,) = ctx.saved_tensors
        return dlogits * g_loss, None, None, None


def _to_device(batch, device):
    return tuple(t.to(device=device, non_blocking=True) for t in batch)


def ForwardModelsTrain(config, task_cfg, device, task_id, batch, model, criterion):
    """Returns (loss, batch_score) like the reference (task_utils.py:308-428, GQA branch :413-428):
    loss = CE(logits, argmax target) * C + semantic_lambda * mean_b(sum_top10 p * dist) * C."""
    batch = _to_device(batch, device)
    features, spatials, image_mask, question, target, input_mask, segment_ids, question_id, ixs, distances = batch
    batch_size = features.size(0)
    ttype = task_cfg[task_id]["type"]
    vil_prediction = model(question, features, spatials, task_id, segment_ids, input_mask, image_mask)[0]
    plain_ce = type(criterion) is nn.CrossEntropyLoss and criterion.reduction == "mean" and criterion.weight is None \
        and criterion.ignore_index == -100 and getattr(criterion, "label_smoothing", 0.0) == 0.0
    if ttype == "VL-classifier-GQA" and plain_ce and vil_prediction.is_cuda and vil_prediction.dtype == torch.float32 \
            and vil_prediction.shape[1] <= 4096 and FUSED_GQA_LOSS:
        # the reference's arithmetic (the branch below) as one native launch: loss, score and d(loss)/d(logits)
        loss, batch_score = GQALossFunction.apply(vil_prediction, target, distances, float(task_cfg[task_id]["semantic_lambda"]))
    elif ttype == "VL-classifier-GQA":
        semantic_lambda = task_cfg[task_id]["semantic_lambda"]
        p_top_k, idx_top_k = torch.topk(F.softmax(vil_prediction, dim=-1), k=10)
        semantic_loss = p_top_k * distances[torch.arange(distances.size(0), device=distances.device).unsqueeze(1),
                                            idx_top_k]
        semantic_loss = torch.mean(torch.sum(semantic_loss, dim=-1), dim=0)
        loss = criterion(vil_prediction, torch.argmax(target.long(), dim=1))
        loss = loss.mean() * target.size(1)
        loss = loss + (semantic_lambda * semantic_loss.mean()) * target.size(1)
        batch_score = compute_score_with_logits(vil_prediction, target).sum() / float(batch_size)
    elif ttype == "VL-classifier":
        loss = criterion(vil_prediction, target)
        loss = loss.mean() * target.size(1)
        batch_score = compute_score_with_logits(vil_prediction, target).sum() / float(batch_size)
    else:
        raise ValueError("clg_vqa_amd.task_utils: unsupported task type %s" % ttype)
    return loss, batch_score


def ForwardModelsVal(config, task_cfg, device, task_id, batch, model, criterion):
    """Returns (float loss, float batch_score, batch_size) (task_utils.py:195-269): CE * C, summed score."""
    batch = _to_device(batch, device)
    features, spatials, image_mask, question, target, input_mask, segment_ids = batch[:7]
    batch_size = features.size(0)
    with torch.no_grad():
        vil_prediction = model(question, features, spatials, task_id, segment_ids, input_mask, image_mask)[0]
    ttype = task_cfg[task_id]["type"]
    if ttype == "VL-classifier-GQA":
        loss = criterion(vil_prediction, torch.argmax(target.long(), dim=1))
    elif ttype == "VL-classifier":
        loss = criterion(vil_prediction, target)
    else:
        raise ValueError("clg_vqa_amd.task_utils: unsupported task type %s" % ttype)
    loss = loss.mean() * target.size(1)
    batch_score = compute_score_with_logits(vil_prediction, target).sum()
    return float(loss), float(batch_score), batch_size


class _LabelSpace(object):
    """Stand-in for ``dataloader.dataset`` where only ``label2ans`` is read (task_utils.py:839)."""

    def __init__(self, label2ans):
        self.label2ans = label2ans


class _Loader(object):
    def __init__(self, label2ans):
        self.dataset = _LabelSpace(label2ans)


def label_space(label2ans):
    """A ``dataloader``-shaped holder of the answer vocabulary for ``EvaluatingModel`` (the reference reads
    ``dataloader.dataset.label2ans[...]``; its GQA dataset loads the list from ``trainval_label2ans.pkl``)."""
    return _Loader(list(label2ans))


def EvaluatingModel(config, task_cfg, device, task_id, batch, model, dataloader, criterion, results, others):
    """volta/volta/task_utils.py:716-841, the VL-classifier(-GQA) branches: forward without gradients, argmax ->
    ``{"questionId": str(id), "prediction": label2ans[argmax]}`` appended to ``results`` (the format
    scripts/GQA_score.py reads); returns ``(loss, score, batch_size, results, others)`` with loss = score = 0 for GQA
    like the reference (:832-833).  ``batch`` is the eval tuple (features, spatials, image_mask, question, target,
    input_mask, segment_ids, question_id, ixs[, distances])."""
    batch = _to_device(batch, device)
    features, spatials, image_mask, question, target, input_mask, segment_ids, question_id = batch[:8]
    batch_size = features.size(0)
    with torch.no_grad():
        vil_prediction = model(question, features, spatials, task_id, segment_ids, input_mask, image_mask)[0]
    ttype = task_cfg[task_id]["type"]
    if ttype == "VL-classifier-GQA":
        logits = torch.max(vil_prediction, 1)[1]
        loss, batch_score = 0, 0
        label2ans = dataloader.dataset.label2ans
        for qid, lab in zip(question_id.tolist(), logits.tolist()):
            results.append({"questionId": str(qid), "prediction": label2ans[lab]})
    elif ttype == "VL-classifier":
        logits = torch.max(vil_prediction, 1)[1]
        loss, batch_score = 0, 0
        label2ans = dataloader.dataset.label2ans
        for qid, lab in zip(question_id.tolist(), logits.tolist()):
            results.append({"question_id": qid, "answer": label2ans[lab]})
    else:
        raise ValueError("clg_vqa_amd.task_utils: unsupported task type %s" % ttype)
    return float(loss), float(batch_score), batch_size, results, others

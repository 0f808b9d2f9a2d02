"""Training-step wrappers of the reference on the native engine:

    python -m clg_vqa_amd.train_task --mode dense   ...   == volta/train_task.py          (:141-370)
    python -m clg_vqa_amd.train_task --mode prune   ...   == volta/train_task_prunning.py (:548-880, IMP)
    python -m clg_vqa_amd.train_task --mode sft --mask_dict_target mask_best.pt ...  == volta/train_task_sft.py
    python -m clg_vqa_amd.train_task --mode eval    ...   == volta/eval_task.py (+ scripts/GQA_score.py format)

``--is_m3p`` (or a config with ``image_embeddings: "m3p"``) selects ``M3PForVLTasks`` and the M3P prune / SFT lists in
every mode (train_task_prunning.py:727-729, :783-787; train_task_sft.py:455-461; eval_task.py:153-156).

Same flags (subset that reaches the hot path), config JSON / task YAML, checkpoint and mask file formats, and loop:
``loss/grad_acc -> backward -> every grad_acc: clip -> AdamW -> scheduler -> zero_grad`` (train_task.py:316-343).
The dataset readers (tensorpack LMDB) are out of scope: batches come from
``clg_vqa_amd.synthetic.make_batch`` in the reference's 10-tuple layout, ``--steps_per_epoch`` of them per epoch.
Multi-GPU: launch with torchrun (one process per GPU); gradients are all-reduced by FusedAdamW over RCCL.
"""
import argparse
import json
import logging
import os
import sys
import time

import torch
import torch.distributed as dist

from . import gqa_score, sft, task_utils, train_utils
from .config import BertConfig, M3PConfig, load_task_cfg
from .data import DevicePrefetcher
from .encoders import BertForVLTasks
from .optim import FusedAdamW
from .synthetic import make_batch

logger = logging.getLogger("clg_vqa_amd.train_task")


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--mode", default="dense", choices=["dense", "prune", "sft", "eval"])
    p.add_argument("--from_pretrained", default="", type=str)
    p.add_argument("--bert_model", default="xlm-roberta-base", type=str)
    p.add_argument("--config_file", required=True, type=str)
    p.add_argument("--tasks_config_file", required=True, type=str)
    p.add_argument("--task", default="15", type=str)
    p.add_argument("--output_dir", default="save", type=str)
    p.add_argument("--resume_file", default="", type=str)
    p.add_argument("--mask_dict_target", default="", type=str)
    p.add_argument("--num_epoch", default=None, type=int)
    p.add_argument("--optim_train_epochs", default=20, type=int)
    p.add_argument("--gradient_accumulation_steps", dest="grad_acc_steps", type=int, default=1)
    p.add_argument("--batch_size", default=None, type=int)
    p.add_argument("--lr", default=None, type=float)
    p.add_argument("--warmup_proportion", default=0.1, type=float)
    p.add_argument("--warmup_steps", default=None, type=float)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--adam_epsilon", default=1e-6, type=float)
    p.add_argument("--adam_betas", default=(0.9, 0.999), nargs="+", type=float)
    p.add_argument("--adam_correct_bias", default=False, action="store_true")
    p.add_argument("--weight_decay", default=0.01, type=float)
    p.add_argument("--clip_grad_norm", default=0.0, type=float)
    p.add_argument("--prune_amount", default=0.1, type=float)
    p.add_argument("--steps_per_epoch", default=8, type=int, help="synthetic batches per epoch (per rank)")
    p.add_argument("--val_batches", default=2, type=int)
    p.add_argument("--seq_len", default=20, type=int)
    p.add_argument("--local_rank", type=int, default=-1)
    p.add_argument("--vocab_size", default=None, type=int, help="override config.vocab_size (tests)")
    p.add_argument("--is_m3p", action="store_true", default=False, help="Use M3P (train_task_prunning.py:447)")
    p.add_argument("--label2ans", default="", type=str, help="eval: JSON list mapping label index -> answer string "
                   "(the reference reads dataset.label2ans from trainval_label2ans.pkl); default: synthetic 'ans<i>'")
    p.add_argument("--truth_file", default="", type=str, help="eval: GQA questions JSON to score against (GQA_score.py)")
    p.add_argument("--split", default="", type=str, help="eval: name of the result file (<split>_result.json)")
    return p.parse_args(argv)


def _batch(args, task_cfg, task, config, step, rank, bs):
    return make_batch(bs, seq_len=args.seq_len, num_boxes=int(task_cfg[task]["max_region_num"]),
                      num_labels=int(task_cfg[task]["num_labels"]), vocab_size=config.vocab_size,
                      num_locs=config.num_locs, feat_dim=config.v_feature_size, seed=(1234 + 7919 * step + rank) % (2 ** 32),
                      l2_normalize=bool(getattr(config, "norm_embeddings", False)))


def evaluate(config, task_cfg, device, task, model, criterion, args, epoch):
    """train_task.py:372-388: validation loss / score over the val batches."""
    model.eval()
    tot_loss, tot_score, n = 0.0, 0.0, 0
    for i in range(args.val_batches):
        batch = _batch(args, task_cfg, task, config, 10 ** 6 + i, 0, int(task_cfg[task].get("eval_batch_size", 64)))
        loss, score, bs = task_utils.ForwardModelsVal(config, task_cfg, device, task, batch, model, criterion)
        tot_loss += loss * bs
        tot_score += score
        n += bs
    model.train()
    return tot_loss / n, 100.0 * tot_score / n


def run_eval(config, task_cfg, device, task, model, criterion, args):
    """eval_task.py:183-210: EvaluatingModel over the split, results in the reference's JSON format
    (``<split>_result.json`` = [{"questionId": str, "prediction": answer string}], ``<split>_others.json``)."""
    num_labels = int(task_cfg[task]["num_labels"])
    if args.label2ans:
        label2ans = json.load(open(args.label2ans))
        if len(label2ans) != num_labels:
            raise ValueError("--label2ans holds %d answers, the task has %d labels" % (len(label2ans), num_labels))
    else:
        label2ans = ["ans%d" % i for i in range(num_labels)]
    loader = task_utils.label_space(label2ans)
    model.eval()
    results, others = [], []
    for i in range(args.val_batches):
        batch = _batch(args, task_cfg, task, config, 10 ** 6 + i, 0, int(task_cfg[task].get("eval_batch_size", 64)))
        _, _, _, results, others = task_utils.EvaluatingModel(config, task_cfg, device, task, batch, model, loader,
                                                              criterion, results, others)
    name = args.split or task_cfg[task].get("val_split", "val")
    json_path = os.path.join(args.output_dir, name)
    json.dump(results, open(json_path + "_result.json", "w"))
    json.dump(others, open(json_path + "_others.json", "w"))
    score = None
    if args.truth_file:
        score = 100 * gqa_score.evaluate(results, json.load(open(args.truth_file)))
        logger.info("GQA score %.3f", score)
    return results, score


def main(argv=None):
    args = parse_args(argv)
    logging.basicConfig(level=logging.INFO, format="%(asctime)s - %(message)s")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", max(args.local_rank, 0)))  # torchrun env, else --local_rank
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend="nccl", device_id=device)
    default_gpu = rank == 0

    raw = json.load(open(args.config_file))
    is_m3p = args.is_m3p or raw.get("image_embeddings") == "m3p"
    config = (M3PConfig if is_m3p else BertConfig).from_json_file(args.config_file)
    if is_m3p:
        from .m3p import M3PForVLTasks as Model
    else:
        Model = BertForVLTasks
    if args.vocab_size:
        config.vocab_size = args.vocab_size
        if is_m3p:
            config.n_words = args.vocab_size
    task_cfg = load_task_cfg(args.tasks_config_file)
    task = "TASK" + args.task.strip()
    base_lr = args.lr or task_cfg[task]["lr"]
    num_epoch = args.num_epoch or task_cfg[task]["num_epoch"]
    bs = (args.batch_size or task_cfg[task]["batch_size"]) // args.grad_acc_steps // world  # task_utils.py:473-479
    torch.manual_seed(args.seed)
    os.makedirs(args.output_dir, exist_ok=True)
    if default_gpu:
        with open(os.path.join(args.output_dir, "command.txt"), "w") as f:
            print(args, file=f)
            print(config, file=f)

    if args.from_pretrained:
        model = Model.from_pretrained(args.from_pretrained, config=config, task_cfg=task_cfg, task_ids=[task])
        if model is None:
            raise FileNotFoundError(args.from_pretrained)
    else:
        model = Model(config, task_cfg, [task])
    model.to(device)
    criterion = task_utils.LoadLoss(args, task_cfg, args.task.strip())

    if args.mode == "eval":
        return run_eval(config, task_cfg, device, task, model, criterion, args)[1]

    if args.mode == "sft":  # train_task_sft.py:410-461
        # a mask_best.pt holds tensors only: the weights-only loader reads it without executing anything from the file
        mask_dict = {k: v.cpu() for k, v in torch.load(args.mask_dict_target, map_location="cpu", weights_only=True).items()}
        sft.premultiply_by_mask(model, mask_dict)
        (sft.pruning_model_custom_m3p if is_m3p else sft.pruning_model_custom)(model, mask_dict, "")
    train_utils.freeze_layers(model)
    theta0, prune_names = None, None
    if args.mode == "prune":  # train_task_prunning.py:728-729
        snap = {k: v.clone() for k, v in model.state_dict().items()}
        if is_m3p:
            prune_names = sft.m3p_prunable_names(len(model.bert.encoder.attentions))
            theta0 = sft.rewind_m3p(snap, "", n_layers=len(model.bert.encoder.attentions))
        else:
            prune_names = sft.uc2_prunable_names(len(model.bert.encoder.layer))
            theta0 = sft.rewind_uc2(snap, "", n_sublayers=len(model.bert.encoder.layer))

    def new_optimizer():
        t_total = args.steps_per_epoch * args.optim_train_epochs // args.grad_acc_steps  # train_task.py:271
        warm = args.warmup_steps or args.warmup_proportion * t_total
        return FusedAdamW(model, base_lr=base_lr, weight_decay=args.weight_decay, betas=tuple(args.adam_betas),
                          eps=args.adam_epsilon, correct_bias=args.adam_correct_bias,
                          max_grad_norm=args.clip_grad_norm if args.clip_grad_norm > 0 else float("inf"),
                          warmup_steps=warm, t_total=t_total,
                          # gradients are exchanged during backward unless micro-batches are accumulated
                          overlap_reduce=None if args.grad_acc_steps == 1 else False,
                          # (pipeline_update=True would run the update of step i under the forward of step i + 1 -- safe in
                          # this loop, which reads parameters itself only behind the synchronize that closes an epoch -- but
                          # it measures neutral at c2: profiles/r03_ab_log.txt section 10)
                          pipeline_update=False)

    opt = new_optimizer()
    _, global_step, start_epoch, _, max_score = train_utils.resume(args.resume_file, model, opt, None, None)
    saver = {"dense": train_utils.save, "sft": train_utils.save_sft, "prune": train_utils.save_prunned}[args.mode]
    model.train()
    step_id = 0
    scored = None
    for epoch in range(start_epoch, num_epoch):
        t0, seen = time.time(), 0
        # batches are staged to HBM two steps ahead on a copy stream (the reference copies inside the step)
        first = step_id
        host_batches = (_batch(args, task_cfg, task, config, first + i, rank, bs) for i in range(args.steps_per_epoch))
        for it, batch in enumerate(DevicePrefetcher(host_batches, device, depth=2)):
            step_id += 1
            loss, score = task_utils.ForwardModelsTrain(config, task_cfg, device, task, batch, model, criterion)
            if args.grad_acc_steps > 1:
                loss = loss / args.grad_acc_steps
            loss.backward()
            seen += bs * world
            if (it + 1) % args.grad_acc_steps == 0:
                if args.mode == "prune" and it + 1 + args.grad_acc_steps > args.steps_per_epoch:
                    # last optimizer step of the epoch: the reference's global_unstructured will score module.weight as
                    # this forward's pre-hook left it (weight_orig * mask BEFORE the step), see clg_vqa_amd/sft.py
                    scored = sft.snapshot_scored_weights(model, prune_names)
                opt.step()
                global_step += 1
        torch.cuda.synchronize()
        dt = time.time() - t0
        if args.mode == "prune":  # train_task_prunning.py:797-866: prune, report, rewind, fresh optimizer
            if is_m3p:
                sft.pruning_model_m3p(model, args.prune_amount, global_pruning=True, scored=scored)
                rate = sft.see_weight_rate_m3p(model)
            else:
                sft.pruning_model_uc2(model, args.prune_amount, global_pruning=True, scored=scored)
                rate = sft.see_weight_rate_uc2(model)
            sd = model.state_dict()
            sd.update({k: v.to(device) for k, v in theta0.items() if k in sd})
            model.load_state_dict(sd)
            model.mark_weights_dirty()
            opt = new_optimizer()
            logger.info("epoch %d: zero rate %.2f %%", epoch, rate)
        vloss, vscore = evaluate(config, task_cfg, device, task, model, criterion, args, epoch)
        is_best = vscore > max_score
        max_score = max(max_score, vscore)
        if default_gpu:
            logger.info("epoch %d: train loss %.4f  %.1f samples/s  val loss %.4f score %.3f", epoch, float(loss),
                        seen / dt, vloss, vscore)
        saver(args.output_dir, logger, epoch, model, opt, None, global_step, None, default_gpu, max_score,
              is_best or args.mode == "prune")
    if world > 1:
        dist.destroy_process_group()
    return max_score


if __name__ == "__main__":
    main(sys.argv[1:])

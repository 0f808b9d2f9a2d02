"""clg_vqa_amd -- MI355X-native engine for the UC2 / M3P VQA fine-tuning hot path of CLG-VQA.

Import as ``clg_vqa_amd`` (the shim package next to this directory).  Public surface mirrors the
reference's for this path: ``BertConfig`` / ``M3PConfig`` (volta/volta/config.py), ``BertForVLTasks``
(volta/volta/encoders.py:1154), the GQA loss glue (volta/volta/task_utils.py) and the sparse
fine-tuning helpers (volta/train_task_prunning.py, volta/train_task_sft.py).
"""
from .config import BertConfig, M3PConfig, TaskCfg, load_task_cfg  # noqa: F401
# BertForVLTasks: clg_vqa_amd.encoders ; M3PForVLTasks: clg_vqa_amd.m3p (imported lazily: they load libvlhip.so)

__version__ = "0.1.0"

"""Configuration and checkpoint compatibility with the reference.

* ``load_config`` -- what transoar/utils/io.py:20-38 does: ``yaml.safe_load`` of an experiment file
  (config/attn_fpn_foc_dec_<dataset>.yaml) and, when it names a ``dataset``, a merge of that dataset's
  ``data_info.json`` (written by the reference's preprocessing, data/preprocessor_*.py:87-93: the
  ``bbox_properties`` the Focused Decoder builds its masks, anchors and restrictions from, plus
  statistics).  The reference finds both relative to the working directory; here the directories are
  arguments.  The dataset-level YAML (config/<dataset>.yaml: ``num_classes``, ``labels`` ...) is only read by
  the reference's preprocessing, whose output already carries those keys; it can be merged explicitly.
* ``save_checkpoint`` / ``load_checkpoint`` -- the dict of trainer.py:230-241 (``epoch``,
  ``metric_max_val``, ``model_state_dict``, ``optimizer_state_dict``, ``scheduler_state_dict``) and the
  resume logic of scripts/train.py:68-80 (the scheduler's ``step_size`` is overridden by the config's
  ``lr_drop``).  Parameter names and shapes of this package's modules equal the reference's (golden fixtures
  g4-g7), so a reference-trained ``model_state_dict`` loads with ``strict=True``.
* ``build_scheduler`` -- scripts/train.py:65: ``StepLR(optimizer, lr_drop)``, stepped once per epoch
  (trainer.py:220).
"""
import json
import os

import torch
import yaml

CHECKPOINT_KEYS = ("epoch", "metric_max_val", "model_state_dict", "optimizer_state_dict", "scheduler_state_dict")


def _read_yaml(path):
    with open(path, "r") as f:
        return yaml.safe_load(f)


def load_config(name_or_path, config_dir="config", dataset_root="dataset", data_info=None, dataset_yaml=None):
    """-> dict with the reference's keys.

    name_or_path  experiment name (``attn_fpn_foc_dec_visceral``: read from config_dir/<name>.yaml) or a path
    data_info     dict, or path of a data_info.json; default dataset_root/<config['dataset']>/data_info.json
                  (io.py:33-36).  Raises FileNotFoundError when the config names a dataset and none is found:
                  the model cannot be built without ``bbox_properties``.
    dataset_yaml  optional config/<dataset>.yaml (num_classes, labels ...) merged first, for data_info files
                  that do not repeat those keys.
    """
    path = name_or_path if os.path.isfile(str(name_or_path)) else os.path.join(config_dir, str(name_or_path) + ".yaml")
    config = _read_yaml(path)
    if dataset_yaml is not None:
        config.update(_read_yaml(dataset_yaml))
    if "dataset" in config or data_info is not None:
        if data_info is None:
            data_info = os.path.join(dataset_root, config["dataset"], "data_info.json")
        if not isinstance(data_info, dict):
            with open(data_info, "r") as f:
                data_info = json.load(f)
        config.update(data_info)
    if "bbox_properties" in config:      # JSON object keys are strings; the model indexes classes by str(c)
        config["bbox_properties"] = {str(k): v for k, v in config["bbox_properties"].items()}
    return config


def build_scheduler(optimizer, config):
    return torch.optim.lr_scheduler.StepLR(optimizer, int(config["lr_drop"]))


def _portable_optimizer_state(optimizer):
    """optimizer.state_dict() in the reference Trainer's format (trainer.py:230-241): Python-float learning rates.  The
    capturable AdamW of TrainStep(graph=True) keeps `lr` / `initial_lr` as device tensors; written as they are, the file
    would not load into a non-capturable optimizer (eager mode, CPU, the reference's own AdamW)."""
    state = optimizer.state_dict()
    groups = []
    for g in state["param_groups"]:
        g = dict(g)
        for k in ("lr", "initial_lr"):
            if torch.is_tensor(g.get(k)):
                g[k] = float(g[k])
        groups.append(g)
    return {"state": state["state"], "param_groups": groups}


def save_checkpoint(path, model, optimizer, scheduler, epoch, metric_max_val=0.0):
    torch.save({
        "epoch": epoch,
        "metric_max_val": metric_max_val,
        "model_state_dict": model.state_dict(),
        "optimizer_state_dict": _portable_optimizer_state(optimizer),
        "scheduler_state_dict": scheduler.state_dict(),
    }, path)


def load_optimizer_state(optimizer, state_dict):
    """optimizer.load_state_dict that keeps what a captured step depends on (round-2 ADVICE).

    ``Optimizer.load_state_dict`` replaces the param groups' hyper-parameters by the checkpoint's: for the capturable
    AdamW of TrainStep(graph=True) that would turn the device-tensor learning rates into Python floats and switch
    ``capturable`` off -- the captured update would raise, or (after capture) keep running on the old lr / moment /
    step tensors.  Here the loaded values are copied INTO the existing tensors: lr tensors are filled in place, the
    per-group flags capturable / fused / foreach / differentiable stay as constructed, and where per-parameter state
    already exists (a warmed-up or captured optimizer) it is overwritten in place instead of being replaced."""
    keep = ("capturable", "fused", "foreach", "differentiable")
    before = [{k: g.get(k) for k in keep + ("lr",)} for g in optimizer.param_groups]
    old_state = {p: dict(st) for p, st in optimizer.state.items()}
    optimizer.load_state_dict(state_dict)
    for g, b in zip(optimizer.param_groups, before):
        for k in keep:
            if b[k] is not None or k in g:
                g[k] = b[k]
        if torch.is_tensor(b["lr"]):
            loaded = g["lr"]
            b["lr"].fill_(float(loaded))
            g["lr"] = b["lr"]
        elif torch.is_tensor(g["lr"]):
            # the other direction (round-3 ADVICE): a checkpoint written by the capturable optimizer, loaded into one
            # with Python-float rates -- foreach AdamW refuses tensor rates, fused AdamW wants them on its own device
            g["lr"] = float(g["lr"])
        if "initial_lr" in g and torch.is_tensor(g["initial_lr"]):
            g["initial_lr"] = float(g["initial_lr"])
    for p, st in optimizer.state.items():
        prev = old_state.get(p)
        for k, v in list(st.items()):
            if not torch.is_tensor(v):
                continue
            if prev is not None and torch.is_tensor(prev.get(k)):      # keep the tensor a captured graph knows
                prev[k].copy_(v.to(prev[k].device))
                st[k] = prev[k]
            elif k == "step" and any(g.get("capturable") or g.get("fused") for g in optimizer.param_groups):
                st[k] = v.to(p.device, torch.float32)                  # capturable / fused AdamW keep step on the device
            else:
                st[k] = v.to(p.device)


def load_checkpoint(path, model, optimizer=None, scheduler=None, config=None, strict=True, map_location="cpu"):
    """Load a checkpoint written by the reference's Trainer (or by save_checkpoint).  -> (epoch, metric_max_val).
    Optimizer / scheduler states are restored when the objects are given; with ``config`` the scheduler's
    step size follows ``lr_drop`` as in scripts/train.py:70."""
    ckpt = torch.load(path, map_location=map_location, weights_only=False)
    missing = [k for k in ("model_state_dict",) if k not in ckpt]
    if missing:
        raise KeyError("%s is not a transoar checkpoint: no %s" % (path, missing))
    model.load_state_dict(ckpt["model_state_dict"], strict=strict)
    if optimizer is not None and "optimizer_state_dict" in ckpt:
        load_optimizer_state(optimizer, ckpt["optimizer_state_dict"])
    if scheduler is not None and "scheduler_state_dict" in ckpt:
        state = dict(ckpt["scheduler_state_dict"])
        if config is not None:
            state["step_size"] = int(config["lr_drop"])
        scheduler.load_state_dict(state)
    return ckpt.get("epoch", 0), ckpt.get("metric_max_val", 0)

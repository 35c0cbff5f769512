"""the reference's experiment YAML + data_info.json and its checkpoint dict
(utils/io.py:20-38, trainer.py:230-241, scripts/train.py:52-80) load into this package."""
import json
import os

import pytest
import torch
import yaml

from tests._inputs import small_model_config

REF = "/root/reference"


def _small_model(refine=True):
    from oracle.torch_ref import msda3d_core_torch
    from transoar_amd import ms_deform_attn
    from transoar_amd.transoarnet import TransoarNet
    ms_deform_attn.register_debug_core(msda3d_core_torch)
    cfg = small_model_config(refine, use_cuda=False)
    torch.manual_seed(3)
    return cfg, TransoarNet(cfg)


def test_yaml_plus_data_info_round_trip(tmp_path):
    """An experiment YAML with the reference's layout (top-level keys + backbone/neck sub-dicts + `dataset`)
    and a data_info.json next to the dataset build the same model as the dict it was dumped from."""
    from transoar_amd.checkpoint import load_config
    from transoar_amd.config import synthetic_bbox_properties, visceral_config
    from transoar_amd.transoarnet import TransoarNet
    cfg = visceral_config(refine=False, use_cuda=False)
    cfg["volume_shape"] = list(cfg["volume_shape"])
    (tmp_path / "config").mkdir()
    (tmp_path / "dataset" / cfg["dataset"]).mkdir(parents=True)
    with open(tmp_path / "config" / "exp.yaml", "w") as f:
        yaml.safe_dump(cfg, f)
    info = {"num_classes": 20, "labels": {str(i): "organ%d" % i for i in range(1, 21)},
            "bbox_properties": {int(k): v for k, v in synthetic_bbox_properties(20, seed=0).items()}}
    with open(tmp_path / "dataset" / cfg["dataset"] / "data_info.json", "w") as f:
        json.dump(info, f)
    with pytest.raises(FileNotFoundError):
        load_config("exp", config_dir=tmp_path / "config", dataset_root=tmp_path / "nowhere")
    got = load_config("exp", config_dir=tmp_path / "config", dataset_root=tmp_path / "dataset")
    assert got["backbone"] == cfg["backbone"] and got["neck"] == cfg["neck"] and got["lr_drop"] == cfg["lr_drop"]
    assert sorted(got["bbox_properties"]) == sorted(str(c) for c in range(1, 21))
    got["backbone"].update(start_channels=4, fpn_channels=48, hidden_dim=48)      # keep the CPU test small
    got["neck"].update(hidden_dim=48, dim_feedforward=64)
    net = TransoarNet(got)
    assert net._anchors.shape == (540, 6)


def test_reference_checkpoint_dict_round_trip(tmp_path):
    """save -> load of the trainer's checkpoint dict: weights, AdamW moments, scheduler position and the
    lr_drop override of scripts/train.py:70; the loaded model reproduces the saved one's outputs."""
    from transoar_amd.checkpoint import CHECKPOINT_KEYS, build_scheduler, load_checkpoint, save_checkpoint
    from transoar_amd.train_step import build_optimizer
    cfg, net = _small_model()
    opt = build_optimizer(net, cfg, fused=False)
    sched = build_scheduler(opt, cfg)
    assert sched.step_size == cfg["lr_drop"] and [g["lr"] for g in opt.param_groups] == [2e-5, 2e-4]
    # a couple of optimizer steps so that the moments are non-trivial, 3 epochs of schedule
    for _ in range(2):
        for p in net.parameters():
            p.grad = torch.full_like(p, 1e-3)
        opt.step()
    for _ in range(3):
        sched.step()
    path = tmp_path / "model_last.pt"
    save_checkpoint(path, net, opt, sched, epoch=3, metric_max_val=0.25)
    raw = torch.load(path, weights_only=False)
    assert tuple(raw.keys()) == CHECKPOINT_KEYS          # trainer.py:235-241
    cfg2, net2 = _small_model()
    with torch.no_grad():
        for p in net2.parameters():
            p.add_(1.0)
    opt2 = build_optimizer(net2, cfg2, fused=False)
    sched2 = build_scheduler(opt2, dict(cfg2, lr_drop=7))
    epoch, best = load_checkpoint(path, net2, opt2, sched2, config=dict(cfg2, lr_drop=7))
    assert (epoch, best) == (3, 0.25)
    assert sched2.last_epoch == 3 and sched2.step_size == 7
    for (n, a), (_, b) in zip(net.state_dict().items(), net2.state_dict().items()):
        assert torch.equal(a, b), n
    s1, s2 = opt.state_dict()["state"], opt2.state_dict()["state"]
    assert s1.keys() == s2.keys() and all(torch.equal(s1[k]["exp_avg"], s2[k]["exp_avg"]) for k in s1)


def test_tensor_learning_rates_are_saved_and_loaded_as_floats(tmp_path):
    """Round-3 ADVICE: the capturable AdamW of TrainStep(graph=True) keeps lr / initial_lr as tensors.  A checkpoint
    written from it holds Python floats (the reference Trainer's format), and a state dict that does carry tensor
    rates (an older file) still loads into an optimizer with float rates and steps."""
    from transoar_amd.checkpoint import build_scheduler, load_checkpoint, load_optimizer_state, save_checkpoint
    lin = torch.nn.Linear(4, 3)
    opt = torch.optim.AdamW([{"params": [lin.weight], "lr": torch.tensor(2e-5)}, {"params": [lin.bias], "lr": torch.tensor(2e-4)}],
                            weight_decay=1e-4, foreach=False, capturable=False)
    for g in opt.param_groups:
        g["initial_lr"] = torch.tensor(float(g["lr"]))
    sched = build_scheduler(opt, {"lr_drop": 5})
    lin.weight.grad, lin.bias.grad = torch.ones_like(lin.weight), torch.ones_like(lin.bias)
    path = tmp_path / "ckpt.pt"
    save_checkpoint(path, lin, opt, sched, epoch=1)
    raw = torch.load(path, weights_only=False)["optimizer_state_dict"]["param_groups"]
    assert all(isinstance(g["lr"], float) and isinstance(g["initial_lr"], float) for g in raw)
    assert [round(g["lr"], 9) for g in raw] == [2e-5, 2e-4]
    # an eager optimizer with float rates takes both the portable file and a raw tensor-rate state dict
    lin2 = torch.nn.Linear(4, 3)
    opt2 = torch.optim.AdamW([{"params": [lin2.weight], "lr": 1.0}, {"params": [lin2.bias], "lr": 1.0}], weight_decay=1e-4, foreach=True)
    load_checkpoint(path, lin2, opt2)
    assert [g["lr"] for g in opt2.param_groups] == pytest.approx([2e-5, 2e-4])
    load_optimizer_state(opt2, opt.state_dict())
    assert all(isinstance(g["lr"], float) for g in opt2.param_groups)
    lin2.weight.grad, lin2.bias.grad = torch.ones_like(lin2.weight), torch.ones_like(lin2.bias)
    opt2.step()          # foreach AdamW raises on tensor rates with capturable=False


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout only exists in the build container")
def test_reference_yaml_files_build_models_with_the_reference_keys(golden_dir):
    """Container only: the reference's own experiment files load through load_config (with a synthetic
    data_info) and the resulting models carry exactly the parameter names the golden whole-model fixture
    recorded from the reference model."""
    import numpy as np
    from transoar_amd.checkpoint import load_config
    from transoar_amd.config import synthetic_bbox_properties
    from transoar_amd.transoarnet import TransoarNet
    for name, n_cls, n_q in (("attn_fpn_foc_dec_visceral", 20, 540), ("attn_fpn_foc_dec_amos", 15, 405)):
        info = {"num_classes": n_cls, "bbox_properties": synthetic_bbox_properties(n_cls, seed=0)}
        cfg = load_config(name, config_dir=os.path.join(REF, "config"), data_info=info)
        assert cfg["neck"]["num_queries"] == n_q and cfg["backbone"]["use_cuda"] is False
        cfg["backbone"].update(start_channels=4, fpn_channels=48, hidden_dim=48, dim_feedforward=64)
        cfg["neck"].update(hidden_dim=48, dim_feedforward=64)
        net = TransoarNet(cfg)
        if n_cls == 20:
            z = np.load(os.path.join(golden_dir, "g7_whole_model.npz"))
            assert [n for n, _ in net.named_parameters()] == list(z["plain.grad_names"])

"""The checkpoint directory layout of the reference (train.py:279-288, src/utils.py:29-53 -> accelerate.save_state /
load_state), pinned against `accelerate` itself: a directory written by accelerate loads here, and a directory written
here loads through accelerate.  CPU only."""
import os

import pytest
import torch


def _model():
    import mm_unet_amd.unet as pu
    torch.manual_seed(3)
    return pu.Unet(3, 1)


def _train_a_little(model, opt, sch):
    model.train()
    for _ in range(2):
        opt.zero_grad()
        model(torch.randn(2, 3, 32, 32)).mean().backward()
        opt.step()
        sch.step()


def test_accelerate_directory_loads_here(tmp_path):
    accelerate = pytest.importorskip("accelerate")
    from mm_unet_amd.checkpoint import load_state
    acc = accelerate.Accelerator(cpu=True)
    m = _model()
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3)
    sch = torch.optim.lr_scheduler.StepLR(opt, 1, gamma=0.5)
    m, opt, sch = acc.prepare(m, opt, sch)
    _train_a_little(m, opt, sch)
    d = str(tmp_path / "checkpoint")
    acc.save_state(output_dir=d)                                         # train.py:286
    torch.save({"epoch": 4, "best_acc": torch.tensor(0.81), "best_class": [0.8]}, os.path.join(d, "epoch.pth.tar"))
    assert "model.safetensors" in os.listdir(d) or "pytorch_model.bin" in os.listdir(d)

    m2 = _model()
    with torch.no_grad():
        for p in m2.parameters():
            p.add_(1.0)
    opt2 = torch.optim.AdamW(m2.parameters(), lr=1e-3)
    sch2 = torch.optim.lr_scheduler.StepLR(opt2, 1, gamma=0.5)
    info = load_state(d, m2, opt2, sch2)
    assert info["epoch"] == 4 and abs(float(info["best_acc"]) - 0.81) < 1e-6
    ref = acc.unwrap_model(m).state_dict()
    for k, v in m2.state_dict().items():
        assert torch.equal(v, ref[k]), k
    assert sch2.state_dict()["last_epoch"] == 2
    s_ref, s_new = opt.state_dict()["state"], opt2.state_dict()["state"]
    assert s_ref.keys() == s_new.keys() and len(s_new) > 0
    for i in s_ref:
        assert torch.equal(s_ref[i]["exp_avg"], s_new[i]["exp_avg"])


@pytest.mark.parametrize("safe", [True, False])
def test_directory_written_here_loads_through_accelerate(tmp_path, safe):
    accelerate = pytest.importorskip("accelerate")
    from mm_unet_amd.checkpoint import load_state, save_state
    m = _model()
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3)
    sch = torch.optim.lr_scheduler.StepLR(opt, 1, gamma=0.5)
    _train_a_little(m, opt, sch)
    d = str(tmp_path / "best")
    save_state(d, m, opt, sch, epoch=7, best_acc=0.5, best_class=[0.5], safe_serialization=safe)
    # our own loader round-trips
    m1 = _model()
    info = load_state(d, m1)
    assert info["epoch"] == 7
    for k, v in m1.state_dict().items():
        assert torch.equal(v, m.state_dict()[k]), k
    if not safe:
        return          # this accelerate version only looks for model.safetensors in load_state
    # accelerate needs its RNG file to resume: write one the way it does (rank 0), then let it load everything else
    import pickle, random
    import numpy as np
    with open(os.path.join(d, "random_states_0.pkl"), "wb") as f:
        pickle.dump({"step": 0, "random_state": random.getstate(), "numpy_random_seed": np.random.get_state(),
                     "torch_manual_seed": torch.get_rng_state()}, f)
    acc = accelerate.Accelerator(cpu=True)
    m2 = _model()
    with torch.no_grad():
        for p in m2.parameters():
            p.mul_(0.0)
    opt2 = torch.optim.AdamW(m2.parameters(), lr=1e-3)
    sch2 = torch.optim.lr_scheduler.StepLR(opt2, 1, gamma=0.5)
    m2, opt2, sch2 = acc.prepare(m2, opt2, sch2)
    acc.load_state(d)                                                     # src/utils.py:44
    for k, v in acc.unwrap_model(m2).state_dict().items():
        assert torch.equal(v, m.state_dict()[k]), k
    assert sch2.state_dict()["last_epoch"] == 2


def test_optimizer_bin_in_timm_group_order_round_trips_through_make_optimizer(tmp_path):
    """The reference's optimizer is timm's ``create_optimizer_v2`` (train.py:197-201), whose parameter groups are
    ``[no_decay (weight_decay 0), decay]`` (optim_factory.py: param_groups_weight_decay / add_weight_decay).
    ``optimizer.state_dict()`` numbers the parameters in group order, so an ``optimizer.bin`` written by the
    reference's training run loads into ``make_optimizer``'s optimizer only if the groups are laid out the same way:
    a state dict built in timm's order by an INDEPENDENT restatement must load, attach every moment to the parameter
    it belongs to, and keep the weight-decay assignment."""
    from mm_unet_amd.checkpoint import load_state, save_state
    from mm_unet_amd.train_step import make_optimizer
    m = _model()
    # "reference side": groups exactly as timm builds them
    named = list(m.named_parameters())
    no_decay = [p for n, p in named if p.ndim <= 1 or n.endswith(".bias")]
    decay = [p for n, p in named if not (p.ndim <= 1 or n.endswith(".bias"))]
    assert len(no_decay) != len(decay)          # a swapped order could not even load
    ref_opt = torch.optim.AdamW([{"params": no_decay, "weight_decay": 0.0}, {"params": decay, "weight_decay": 0.05}],
                                lr=1e-3, betas=(0.9, 0.95))
    sch = torch.optim.lr_scheduler.StepLR(ref_opt, 1, gamma=0.5)
    _train_a_little(m, ref_opt, sch)
    d = str(tmp_path / "checkpoint")
    save_state(d, m, ref_opt, sch, epoch=1, best_acc=0.1, best_class=[0.1])
    moments = {n: ref_opt.state[p]["exp_avg"].clone() for n, p in named}

    m2 = _model()
    opt2 = make_optimizer(m2, lr=1e-3, weight_decay=0.05, betas=(0.9, 0.95), fused=False)
    load_state(d, m2, opt2)                      # raises ValueError on a group-size mismatch
    assert [g["weight_decay"] for g in opt2.param_groups] == [0.0, 0.05]
    for n, p in m2.named_parameters():
        assert torch.equal(opt2.state[p]["exp_avg"], moments[n]), f"moment of {n} attached to another parameter"
        wd = next(g["weight_decay"] for g in opt2.param_groups if any(q is p for q in g["params"]))
        assert wd == (0.0 if (p.ndim <= 1 or n.endswith(".bias")) else 0.05), n
    # and the other direction: a directory written from make_optimizer's optimizer loads into the timm-ordered one
    d2 = str(tmp_path / "best")
    save_state(d2, m2, opt2)
    m3 = _model()
    named3 = list(m3.named_parameters())
    opt3 = torch.optim.AdamW([{"params": [p for n, p in named3 if p.ndim <= 1 or n.endswith(".bias")], "weight_decay": 0.0},
                              {"params": [p for n, p in named3 if not (p.ndim <= 1 or n.endswith(".bias"))],
                               "weight_decay": 0.05}], lr=1e-3, betas=(0.9, 0.95))
    load_state(d2, m3, opt3)
    for n, p in named3:
        assert torch.equal(opt3.state[p]["exp_avg"], moments[n]), n


def test_load_state_keeps_a_tensor_learning_rate(tmp_path):
    """``make_optimizer(capturable=True)`` keeps the learning rate in a tensor that a captured optimizer
    step reads at replay time.  ``Optimizer.load_state_dict`` swaps each group's ``lr`` for the saved value (a float from
    a reference-written ``optimizer.bin``, a CPU tensor from our own): load_state must write the value INTO the
    existing tensor instead, so that ``set_lr`` / schedulers still drive what the step reads."""
    from mm_unet_amd.checkpoint import load_state, save_state
    from mm_unet_amd.train_step import set_lr
    m = _model()
    # (a) float-lr file, as the reference's accelerate writes it; (b) tensor-lr file, as save_state writes ours
    for tensor_lr_in_file in (False, True):
        lr_saved = torch.tensor(3e-4) if tensor_lr_in_file else 3e-4
        opt = torch.optim.AdamW(m.parameters(), lr=lr_saved, foreach=False)
        sch = torch.optim.lr_scheduler.StepLR(opt, 10)
        _train_a_little(m, opt, sch)
        d = str(tmp_path / f"ck{int(tensor_lr_in_file)}")
        save_state(d, m, opt)
        m2 = _model()
        lr_live = torch.tensor(1e-3)
        opt2 = torch.optim.AdamW(m2.parameters(), lr=lr_live, foreach=False)
        load_state(d, m2, opt2)
        for g in opt2.param_groups:
            assert g["lr"] is lr_live, "load_state replaced the live learning-rate tensor"
        assert abs(float(lr_live) - 3e-4) < 1e-9, "the loaded value did not reach the live tensor"
        set_lr(opt2, 0.0)
        assert float(lr_live) == 0.0
        before = [p.detach().clone() for p in m2.parameters()]
        m2(torch.randn(2, 3, 32, 32)).mean().backward()
        opt2.step()
        assert all(torch.equal(a, b.detach()) for a, b in zip(before, m2.parameters())), "lr = 0 did not reach the step"

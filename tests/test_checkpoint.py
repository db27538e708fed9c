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

"""Checkpoints in the layout the reference's training script writes and resumes from.

train.py:279-288 saves with ``accelerator.save_state(output_dir=.../best | .../checkpoint)`` plus an
``epoch.pth.tar`` ({'epoch', 'best_acc', 'best_class'}); src/utils.py:29-53 resumes with ``accelerator.load_state``.
``accelerate``'s directory holds ``model.safetensors`` (``pytorch_model.bin`` with ``safe_serialization=False`` / older
versions), ``optimizer.bin``, ``scheduler.bin`` (both ``torch.save`` of the objects' ``state_dict()``) and
``random_states_<rank>.pkl``.  Parameter names are the module's ``state_dict`` keys, which this package reproduces
exactly (tests/test_host_logic.py::test_seeded_construction_reproduces_reference_weights), so a reference-trained
directory loads into ``MM_Net`` / ``Unet`` here and vice versa.  Neither function needs ``accelerate``.
"""
import os

import torch

MODEL_FILES = ("model.safetensors", "pytorch_model.bin")


def _load_model_file(path):
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path, device="cpu")
    return torch.load(path, map_location="cpu", weights_only=True)


def _load_obj_state(obj, state):
    """``obj.load_state_dict(state)``; for an optimizer whose learning rate is a DEVICE TENSOR
    (``train_step.make_optimizer(capturable=True)``: a captured optimizer step reads it at replay time) the tensor
    object is kept and the loaded value is written into it.  ``Optimizer.load_state_dict`` replaces each group's
    ``lr`` with what the file holds -- a Python float from a reference-written ``optimizer.bin``, a CPU tensor from
    this package's own -- and either would be baked into (or break) the captured step."""
    groups = getattr(obj, "param_groups", None)
    kept = [g["lr"] if isinstance(g.get("lr"), torch.Tensor) else None for g in groups] if groups is not None else []
    obj.load_state_dict(state)
    for g, lr in zip(getattr(obj, "param_groups", []), kept):
        if lr is not None:
            lr.fill_(float(g["lr"]))
            g["lr"] = lr


def load_state(directory, model, optimizer=None, scheduler=None, strict=True):
    """Loads an ``accelerator.save_state`` directory: model weights (required), optimizer / scheduler state when the
    objects are passed and their files exist.  Returns the contents of ``epoch.pth.tar`` ({} when absent)."""
    for name in MODEL_FILES:
        path = os.path.join(directory, name)
        if os.path.exists(path):
            model.load_state_dict(_load_model_file(path), strict=strict)
            break
    else:
        raise FileNotFoundError(f"no {' / '.join(MODEL_FILES)} in {directory}")
    for obj, name in ((optimizer, "optimizer.bin"), (scheduler, "scheduler.bin")):
        path = os.path.join(directory, name)
        if obj is not None and os.path.exists(path):
            _load_obj_state(obj, torch.load(path, map_location="cpu", weights_only=False))
    epoch_file = os.path.join(directory, "epoch.pth.tar")
    return torch.load(epoch_file, map_location="cpu", weights_only=False) if os.path.exists(epoch_file) else {}


def save_state(directory, model, optimizer=None, scheduler=None, epoch=None, best_acc=None, best_class=None,
               safe_serialization=True):
    """Writes the same directory (single process: weights are saved from the calling rank; under data parallelism
    every replica holds the same weights, so rank 0 alone should call this)."""
    os.makedirs(directory, exist_ok=True)
    sd = {k: v.detach().to("cpu").contiguous() for k, v in model.state_dict().items()}
    if safe_serialization:
        from safetensors.torch import save_file
        save_file(sd, os.path.join(directory, "model.safetensors"), metadata={"format": "pt"})
    else:
        torch.save(sd, os.path.join(directory, "pytorch_model.bin"))
    if optimizer is not None:
        torch.save(optimizer.state_dict(), os.path.join(directory, "optimizer.bin"))
    if scheduler is not None:
        torch.save(scheduler.state_dict(), os.path.join(directory, "scheduler.bin"))
    if epoch is not None:   # train.py:287-288
        torch.save({"epoch": epoch, "best_acc": best_acc, "best_class": best_class},
                   os.path.join(directory, "epoch.pth.tar"))

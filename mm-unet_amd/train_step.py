"""One MM-UNet training step with the semantics of the reference's ``train_one_epoch`` body
(train.py:35-54): forward -> loss -> backward (+ DP gradient all-reduce, train.py:52/252) -> AdamW
step -> zero_grad.  The per-step host syncs of the reference (``float(loss)`` and the MONAI metrics on
the training batch, train.py:42-48,61) are not part of the step; the loss is returned as a device tensor.

``make_optimizer`` mirrors the reference's ``create_optimizer_v2('adamw', lr 1e-3, weight_decay .05,
betas (.9,.95))`` (train.py:197-201): decoupled weight decay, none on 1-D parameters / biases.
"""
import torch

from .dp import GradAllReducer


def make_optimizer(module, lr=1e-3, weight_decay=0.05, betas=(0.9, 0.95), fused=None):
    decay, no_decay = [], []
    for name, p in module.named_parameters():
        if not p.requires_grad:
            continue
        (no_decay if p.ndim <= 1 or name.endswith(".bias") else decay).append(p)
    if fused is None:
        fused = all(p.is_cuda for p in decay + no_decay)
    return torch.optim.AdamW([{"params": decay, "weight_decay": weight_decay},
                              {"params": no_decay, "weight_decay": 0.0}], lr=lr, betas=betas, fused=fused)


class TrainStep:
    def __init__(self, model, loss_fn, optimizer, group=None, amp_dtype=None, bucket_bytes=16 << 20, overlap=True):
        self.model, self.loss_fn, self.optimizer = model, loss_fn, optimizer
        self.amp_dtype = amp_dtype
        self.reducer = GradAllReducer(model, group=group, bucket_bytes=bucket_bytes, overlap=overlap)

    def forward_backward(self, images, targets):
        if self.amp_dtype is not None:
            with torch.autocast(images.device.type, dtype=self.amp_dtype):
                logits = self.model(images)
            loss = self.loss_fn(logits.float(), targets)
        else:
            loss = self.loss_fn(self.model(images), targets)
        loss.backward()
        return loss.detach()

    def __call__(self, images, targets):
        loss = self.forward_backward(images, targets)
        self.reducer.finish()
        self.optimizer.step()
        self.reducer.zero_grad()
        return loss

"""mm-unet_amd -- MI355X-native (gfx950) implementation of the MM-UNet hot path.

Directory name follows the project layout (``mm-unet_amd/``); it is importable as
``mm_unet_amd`` through the alias package next to it.

Layers (bottom-up), each mirroring the reference interface it replaces:
  csrc/                      hand-written HIP kernels + the C-ABI (include/mmunet_amd.h)
  selective_scan_hip         <-> selective_scan_cuda  (pybind module of the reference)
  causal_conv1d_hip          <-> causal_conv1d_cuda
  selective_scan_interface   <-> mamba_ssm/ops/selective_scan_interface.py
  causal_conv1d_interface    <-> causal_conv1d/causal_conv1d_interface.py
  mamba_simple               <-> requirements/mamba_simple.py  (Mamba module)
  mmunet                     <-> src/UM_Net/MMUNet.py          (MMConv, RCG, MM_Net ...)
  unet, loss                 <-> model.py, loss.py
  dp, train_step             <-> train.py:28-79,252 (DDP step) -- RCCL gradient all-reduce

There is no CPU fallback anywhere in this package: ops raise RuntimeError when the HIP
library is missing or when handed CPU tensors.
"""
__version__ = "0.1.0"

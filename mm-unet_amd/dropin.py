"""Make this package answer to the module names the reference imports, so that the reference's
``mamba_simple.py`` / ``MMUNet.py`` / ``train.py`` run unchanged on MI355X:

    import mm_unet_amd.dropin; mm_unet_amd.dropin.install()
    from mamba_ssm import Mamba                                   # MMUNet.py:7
    from mamba_ssm.ops.selective_scan_interface import ...        # mamba_simple.py:19
    from causal_conv1d import causal_conv1d_fn                    # mamba_simple.py:14
    import selective_scan_cuda, causal_conv1d_cuda                # selective_scan_interface.py:10-11

    import monai                                                  # train.py:7 -- ONLY if MONAI itself is absent

Only ``sys.modules`` entries are added; nothing is installed or patched on disk.  The MONAI names train.py uses
(sliding-window inferer, post-transforms, the seven metrics, DiceFocalLoss: monai_shim.py) are registered only when no
real ``monai`` can be imported.
"""
import sys
import types


def install(force=False):
    from . import causal_conv1d_hip, causal_conv1d_interface, mamba_simple, selective_scan_hip, \
        selective_scan_interface

    def put(name, mod):
        if force or name not in sys.modules:
            sys.modules[name] = mod
        return sys.modules[name]

    put("selective_scan_cuda", selective_scan_hip)
    put("causal_conv1d_cuda", causal_conv1d_hip)
    cc = types.ModuleType("causal_conv1d")
    cc.causal_conv1d_fn = causal_conv1d_interface.causal_conv1d_fn
    cc.causal_conv1d_update = causal_conv1d_interface.causal_conv1d_update
    cc.causal_conv1d_interface = causal_conv1d_interface
    cc.__path__ = []
    put("causal_conv1d", cc)
    put("causal_conv1d.causal_conv1d_interface", causal_conv1d_interface)
    ms = types.ModuleType("mamba_ssm")
    ms.Mamba = mamba_simple.Mamba
    ms.__path__ = []
    ops = types.ModuleType("mamba_ssm.ops")
    ops.__path__ = []
    ops.selective_scan_interface = selective_scan_interface
    mods = types.ModuleType("mamba_ssm.modules")
    mods.__path__ = []
    mods.mamba_simple = mamba_simple
    ms.ops, ms.modules = ops, mods
    put("mamba_ssm", ms)
    put("mamba_ssm.ops", ops)
    put("mamba_ssm.ops.selective_scan_interface", selective_scan_interface)
    put("mamba_ssm.modules", mods)
    put("mamba_ssm.modules.mamba_simple", mamba_simple)
    if "monai" not in sys.modules:
        import importlib.util
        if importlib.util.find_spec("monai") is None:
            from . import monai_shim
            monai_shim.install(put)

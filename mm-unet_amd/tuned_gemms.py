"""Library GEMM selection for the shapes MM-UNet sends to rocBLAS / hipBLASLt (the Mamba projections of the three large
RCG blocks, the DSC convolutions of the small-map blocks: 2.7 ms of a 34 ms training step).

PyTorch's TunableOp times every solution the two libraries offer for a GEMM shape and keeps the fastest; the defaults are
up to 1.6x slower on these skinny shapes (36 x 128 against 524,288 tokens: 94 us default).  The selections for the
BASELINE configurations were recorded once on an MI355X (``tools/tune_gemms.sh``) and are shipped in ``tuned/``;
``enable()`` -- called by ``TrainStep`` / ``InferStep`` -- switches TunableOp on in LOOK-UP mode: shapes in the file get
their recorded solution, everything else the library default, NOTHING is tuned at run time.  A file recorded with other
library versions or on another GPU fails TunableOp's validators and is ignored (defaults everywhere).  Only float32 shapes
are recorded: tuning the bfloat16 shapes of config 3 ended in a GPU memory fault inside one of the candidate library
kernels (round 3), so the tool does not try them and this module never switches tuning on.

    MMUNET_TUNED_GEMMS=0        leave TunableOp alone
A process that sets PYTORCH_TUNABLEOP_ENABLED itself keeps full control: ``enable()`` does nothing then.
"""
import os

import torch

FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned", "gemm_gfx950.csv")
_done = False


def enable():
    global _done
    if _done:
        return
    _done = True
    mode = os.environ.get("MMUNET_TUNED_GEMMS", "1")
    if mode == "0" or "PYTORCH_TUNABLEOP_ENABLED" in os.environ or not torch.cuda.is_available():
        return
    if not os.path.exists(FILE):
        return
    # a private copy, created exclusively (no predictable name, no symlink to follow) and removed at exit; TunableOp is
    # told not to write anything back
    import atexit
    import shutil
    import tempfile
    fd, mine = tempfile.mkstemp(prefix="mmunet_tuned_gemm_", suffix=".csv")
    with os.fdopen(fd, "wb") as dst, open(FILE, "rb") as src:
        shutil.copyfileobj(src, dst)
    atexit.register(lambda: os.path.exists(mine) and os.remove(mine))
    tun = torch.cuda.tunable
    tun.enable(True)
    tun.tuning_enable(False)        # look-up only
    if hasattr(tun, "write_file_on_exit"):
        tun.write_file_on_exit(False)
    tun.set_filename(mine, insert_device_ordinal=False)
    global _status
    try:
        ok = bool(tun.read_file(mine))
    except Exception:   # noqa: BLE001 -- a file the validators reject must not stop training: defaults everywhere
        ok = False
    _status = {"file": os.path.basename(FILE), "validators_ok": ok,
               "entries": len(tun.get_results()) if ok and hasattr(tun, "get_results") else 0}


_status = {"file": None, "validators_ok": False, "entries": 0}


def status():
    """What ``enable()`` found: the file, whether TunableOp's validators (library versions, GPU) accepted it, the number
    of recorded selections in effect -- ``bench.py`` prints it in its line (``config.tuned_gemm_lookups``)."""
    return dict(_status)

"""m3ae_amd: the MI355X-native M3AE hot path (modules mirroring m3ae/modules of the reference, ops over libm3ae_hip.so)."""
import os

# The model's text half runs on a second HIP stream beside the image half (modules/m3ae_module.py::_fusion_two_streams).  The HIP
# runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4): once RCCL has created its own streams the side
# stream lands on the caller's queue and the two halves serialise again (measured, one GPU with a one-rank RCCL group: the step
# is 2.6 % slower, the whole gain of the second stream; with 8 queues it is back, profiles/r03_two_stream_hw_queues.log).  Read
# by the runtime when HIP initialises, so it is set at import (a value the user exported wins).
if "GPU_MAX_HW_QUEUES" not in os.environ:
    os.environ["GPU_MAX_HW_QUEUES"] = "8"
    import sys as _sys
    _t = _sys.modules.get("torch")
    if _t is not None and _t.cuda.is_initialized():
        import warnings
        warnings.warn("m3ae_amd was imported after HIP initialised: GPU_MAX_HW_QUEUES=8 cannot take effect any more; beside RCCL "
                      "the model's second stream then shares a hardware queue with the first (about -2.6 % on the step). "
                      "Import m3ae_amd (or export GPU_MAX_HW_QUEUES=8) before the first CUDA call.")

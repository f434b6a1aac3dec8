"""cmpc-refseg_amd -- MI355X-native CMPC (Cross-Modal Progressive Comprehension) hot path.

The directory name carries a hyphen (fixed by the project layout), so import it with
    importlib.import_module("cmpc-refseg_amd")
Public surface mirrors the reference: LSTM_model (CMPC_model.py:13) and
get_segmentation_model (get_model.py:15-17).
"""
from . import _lib                      # noqa: F401  (ctypes binding of libcmpc_hip.so)
from .params import HeadCfg, ParamStore, head_param_specs, init_head_params   # noqa: F401


def __getattr__(name):
    # model / backbone import torch.cuda-facing code lazily so that CPU-only tooling can still
    # inspect the manifest and the C ABI.
    if name in ("LSTM_model", "get_segmentation_model"):
        from . import model
        return getattr(model, name)
    if name in ("model", "backbone", "dist", "engine", "checkpoint", "hostutil"):
        import importlib
        return importlib.import_module(f"{__name__}.{name}")
    raise AttributeError(name)

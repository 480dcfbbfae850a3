"""ctrl-vio_amd: MI355X-native sliding-window continuous-time VIO solve (the Ctrl-VIO hot path).

The directory name contains a hyphen: import it with
    importlib.import_module("ctrl-vio_amd")
(with the repo root on sys.path).

  window   Window: flat description of one sliding window (state + factors), the C ABI's wire format
  capi     ctypes binding of libctvio.so (include/ctvio.h); build_library() drives hipcc for gfx950
  solver   Solver: batch of windows on one GPU (device-resident LM)
  packer   factor packing rules of TrajectoryManager::UpdateTrajectory (bias index, bias chain weights)
  splines  host-side NumPy spline evaluation (generator / small queries)
  synth    deterministic synthetic windows for the BASELINE.json configs
"""
from .window import Window, rel_state_error  # noqa: F401
from . import splines, packer, synth, capi, sharding  # noqa: F401
from .solver import Solver  # noqa: F401

"""ctrl-vio_amd: MI355X-native sliding-window continuous-time VIO solve (Ctrl-VIO hot path).

The directory name contains a hyphen: import it with
    importlib.import_module("ctrl-vio_amd")
(the repo root on sys.path).  Sub-modules: window, splines, packer, synth, capi, solver.
"""
from .window import Window, rel_state_error  # noqa: F401
from . import splines, packer, synth  # noqa: F401

import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def cv():
    """The product package (directory name has a hyphen -> import_module)."""
    return importlib.import_module("ctrl-vio_amd")


@pytest.fixture(scope="session")
def oracle():
    """ctypes binding of the CPU fp64 oracle (test infrastructure)."""
    import pyctvo
    pyctvo.build()
    return pyctvo


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def oracle_solved(cv, oracle):
    """(config, seed) -> (the oracle's solved copy of the synthetic window, its summary), computed once per test session: several GPU tests
    compare different batch shapes of the same seeds with the same 15-iteration oracle solve (a config-5 solve takes the oracle 5 s)."""
    cache = {}

    def get(cfg, seed, iters=15):
        key = (cfg, seed, iters)
        if key not in cache:
            ref = cv.synth.make_window(cfg, seed=seed)
            cache[key] = (ref, oracle.OracleWindow(ref).solve(iters))
        return cache[key]
    return get

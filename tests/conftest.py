import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    """The parity oracle (oracle/libofdg_oracle.so), built on demand with g++."""
    import oracle_lib
    oracle_lib.lib()
    return oracle_lib


@pytest.fixture(scope="session")
def ofdg():
    """The product package; libofdg.so must exist (built by __graft_entry__.build())."""
    mod = importlib.import_module("optical-flow-2d-data-generation_amd")
    if not os.path.exists(mod.LIB_PATH):
        mod.build()
    mod.lib()
    return mod

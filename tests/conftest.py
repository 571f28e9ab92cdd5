import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def cfg():
    from openpano_amd.config import PanoConfig
    return PanoConfig()


@pytest.fixture(scope="session")
def oracle(cfg):
    from checkers import Oracle
    return Oracle(cfg)


@pytest.fixture(scope="session")
def ref(cfg):
    from checkers import Ref, ref_available
    if not ref_available():
        pytest.skip("oracle/_ref not built (reference sources absent)")
    return Ref(cfg)

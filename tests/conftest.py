import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dark-archon_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_binding
    return oracle_binding.Oracle()


@pytest.fixture(scope="session")
def archon():
    """The HIP path through the C ABI.  Fails loudly when the extension is missing."""
    import torch
    if torch.cuda.is_available():
        torch.cuda.init()            # torch first: one HIP runtime initialisation order for the whole session
    import pyarchon
    pyarchon.lib()
    if pyarchon.device_count() < 1:
        pytest.fail("no HIP device visible: GPU tests must run on the MI355X box")
    return pyarchon

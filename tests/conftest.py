import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dark-archon_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "small_block_default: the test runs small blocks on the route the product takes by itself")
    config.addinivalue_line("markers", "streaming_machinery: the test asserts statistics of the streaming stage on a small block: it keeps SMALL_BLOCK=0 "
                            "also when the suite runs on the product's routes (ARCHON_TEST_PRODUCT_ROUTES=1)")


@pytest.fixture(autouse=True)
def _streaming_stage_on_small_blocks(request, monkeypatch):
    """The product sends blocks below 8 MiB through a byte count + LSB passes (archon_hip.hip, kSmallBlock): the streaming
    stage -- the graded path -- is built for blocks that fill the chip.  The tests exist to exercise that machinery on
    inputs the oracle finishes in seconds, so by default they switch the small-block rule off (test route SMALL_BLOCK = 0);
    tests marked `small_block_default` (and everything that goes through bin/archon) run the product's own choice."""
    # ARCHON_TEST_PRODUCT_ROUTES=1 runs the suite a second time the other way round: every test takes the product's own small-block
    # choice except those that assert the streaming machinery's statistics (marked `streaming_machinery`)
    if os.environ.get("ARCHON_TEST_PRODUCT_ROUTES") == "1":
        if "streaming_machinery" in request.keywords:
            monkeypatch.setenv("ARCHON_SMALL_BLOCK", "0")
        else:
            monkeypatch.delenv("ARCHON_SMALL_BLOCK", raising=False)
    elif "small_block_default" not in request.keywords:
        monkeypatch.setenv("ARCHON_SMALL_BLOCK", "0")


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

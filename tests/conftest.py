import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "sstem-restoration_amd")
for p in (PKG, REPO):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def repo_root():
    return REPO


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(REPO, "tests", "golden")


@pytest.fixture(params=["auto", "bf16x6"])
def conv_algo_matrix(request):
    """Runs a test twice: under ALGO_AUTO, and with every 3x3 forward / data gradient forced through the split-bf16 X6 kernels
    (fp32 operands as three bf16 pieces; csrc/conv_split_kernels.hip) -- same goldens, same tolerances.  Modules opt in with
    ``pytest.mark.usefixtures("conv_algo_matrix")``."""
    import hipnn.functional as HF
    prev = HF.get_algorithm()
    if request.param == "bf16x6":
        HF.set_algorithm(HF.ALGO_MFMA_BF16X6)
    yield request.param
    HF.set_algorithm(prev)

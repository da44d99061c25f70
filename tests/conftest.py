import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# the product package directory is not a valid python identifier: put it on the path and
# import its host-side module `gfship`
sys.path.insert(0, os.path.join(ROOT, "gerris-fft-particles_amd"))

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    # a fresh checkout has no built artefacts (they are git-ignored): build them once, as
    # __graft_entry__.build() does (hipcc cross-compiles without a GPU).  Nothing is rebuilt when the
    # library, the front end and the oracle are there.
    pkg = os.path.join(ROOT, "gerris-fft-particles_amd")
    need = [os.path.join(pkg, "lib", "libgfship.so"), os.path.join(pkg, "bin", "gfship2D"),
            os.path.join(ROOT, "oracle", "libgfsoracle.so"),
            os.path.join(ROOT, "tests", "mock_rccl", "librccl_mock.so")]
    if not all(os.path.exists(f) for f in need):
        import subprocess
        subprocess.check_call(["bash", os.path.join(pkg, "csrc", "build.sh")])
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
        subprocess.check_call(["bash", os.path.join(ROOT, "tests", "mock_rccl", "build.sh")])


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False

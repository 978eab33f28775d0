import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    return pyoracle.Oracle()


@pytest.fixture(scope="session")
def ref():
    """The compiled reference (oracle/_ref/libjref.so).  Built in the dev
    container from /root/reference; travels prebuilt to the GPU box."""
    from oracle import pyoracle
    if not pyoracle.REF_SO.exists():
        if Path("/root/reference/libsent/src/phmm/outprob.c").exists():
            pyoracle.build(ref=True)
        else:
            pytest.skip("oracle/_ref/libjref.so not built and /root/reference absent")
    return pyoracle.Ref()


@pytest.fixture(scope="session")
def engine():
    """HIP engine on cuda:0 -- fails loudly if the extension or GPU is missing."""
    from julius_amd import lib
    eng = lib.Engine(0)
    yield eng
    eng.close()


GOLDEN = ROOT / "tests" / "golden"

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def root():
    return ROOT


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def arena():
    import ctpn_amd
    return ctpn_amd.make_synthetic_arena(0)


@pytest.fixture(autouse=True)
def _cfg_is_per_test():
    """cfg is a process-wide object like the reference's: a test that edits TEST.PRECISION / DETECT_MODE / MAX_BATCH (or runs demo.main,
    which merges a text.yml into it) does not leak that into the next test."""
    from ctpn_amd.lib.fast_rcnn.config import cfg
    saved = (cfg.TEST.PRECISION, cfg.TEST.DETECT_MODE, cfg.TEST.MAX_BATCH)
    yield
    cfg.TEST.PRECISION, cfg.TEST.DETECT_MODE, cfg.TEST.MAX_BATCH = saved

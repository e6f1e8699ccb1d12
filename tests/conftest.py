import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A GPU session on a fresh box starts with minutes of silence: the first `import torch` pages the image in, and the
    first test (tests/test_dist_gloo.py) starts two more processes that do the same. A harness that takes three silent
    minutes for a hang would kill the run there, so a session that selects GPU tests says that it is alive once a minute
    (a daemon thread writing to the real stderr; every test still has its own --timeout)."""
    if "gpu" not in (session.config.getoption("-m") or "") or "not gpu" in (session.config.getoption("-m") or ""):
        return
    import threading
    import time

    def beat():
        t0 = time.time()
        while True:
            time.sleep(60)
            try:
                sys.__stderr__.write("[gpu tests] running, %d s\n" % (time.time() - t0))
                sys.__stderr__.flush()
            except Exception:
                return

    threading.Thread(target=beat, daemon=True).start()


@pytest.fixture(scope="session")
def goldens():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "goldens.npz"))


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import oracle as O
    O.lib()
    return O


def pytest_sessionfinish(session, exitstatus):
    """16-bit parity numbers collected by tests/parity16.py during a GPU session (worst case per scene or scene family)."""
    import json
    import parity16
    if not parity16.REPORT:
        return
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity16_report.json"), "w") as f:
        json.dump(parity16.REPORT, f, indent=1, sort_keys=True)

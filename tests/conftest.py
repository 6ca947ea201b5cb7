import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # development aid: ENVGS_TEST_DEBUG_TRACE=<bits> runs the whole suite with that ENVGS_DBG_TRACE default (e.g. a collection kernel under test);
    # tests that pin the switch themselves still do
    v = os.environ.get("ENVGS_TEST_DEBUG_TRACE")
    if v:
        from envgs_amd import _lib
        for kind in ("product", "diag"):
            old = _lib.select(kind)
            _lib.load().envgs_debug_set(0, int(v))
            _lib.select(old)


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "boundary_golden.npz"))


def pytest_sessionfinish(session, exitstatus):
    """ENVGS_PARITY_COLLECT turns every parity assertion into a recording (diagnosis runs).  A stray variable must not make the suite pass
    without checking anything: such a session always ends with a failing exit status."""
    if os.environ.get("ENVGS_PARITY_COLLECT"):
        session.exitstatus = 1


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """Measured parity errors of this run (tests/util.py:check_close), so that they are part of the test log the driver records."""
    try:
        from tests.util import ERROR_TABLE
    except Exception:
        return
    if not ERROR_TABLE:
        return
    tr = terminalreporter
    if os.environ.get("ENVGS_PARITY_COLLECT"):
        tr.write_sep("!", "ENVGS_PARITY_COLLECT is set: parity assertions were NOT evaluated in this run -- exit status forced to 1")
    tr.write_sep("=", "measured parity errors (elementwise |a-b| / (|b| + mean|b|); fragile pixels / rays excluded and counted)")
    worst = {}
    for r in ERROR_TABLE:
        key = (r["test"].split("[")[0], r["tensor"])
        w = worst.get(key)
        if w is None or r["max_err"] > w["max_err"]:
            worst[key] = dict(r, cases=(w["cases"] + 1 if w else 1))
        else:
            w["cases"] += 1
    for (test, tensor), r in sorted(worst.items()):
        tol = "" if r["tol"] is None else " (tol %.0e)" % r["tol"]
        tr.write_line("%-58s %-14s max %.2e%s  cases %d  n %d  excl %d %s" % (test[:58], tensor, r["max_err"], tol, r["cases"], r["n"], r["excluded"], r.get("note", "")))
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        import json
        with open(os.path.join(out, "parity_errors.json"), "w") as f:
            json.dump(ERROR_TABLE, f, indent=0)

"""Round 6: the two training tests (tests/test_train_convergence.py: activated parameters, FusedAdam, densification-free) with the deferred surfel
gradients switched ON for the whole run -- envgs_forward routes the activated env tensors through tracing.defer_barrier.  python scratch/defer_convergence.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytest
from envgs_amd import envgs_step, tracing
envgs_step.DEFER["on"] = True
n = [0, 0]
orig = tracing.trace_backward
def counting(saved, *a, **kw):
    r = orig(saved, *a, **kw)
    n[0] += 1; n[1] += int(saved["lists"].defer_reduce & 1) if saved["lists"] is not None else 0
    return r
tracing.trace_backward = counting
rc = pytest.main(["tests/test_train_convergence.py", "-x", "-q", "-m", "gpu"])
print("traced backward calls %d, deferred %d, pytest rc %s" % (n[0], n[1], rc))

"""pytest configuration: registers the `gpu` marker, puts the repo root / package dir on sys.path, orders the GPU suite by importance
(hot-path parity first, screens last: a failure in a widening row cannot hide the hot path behind `-x`) and arms the stall watchdog
(tests/stall_diag.py) around every GPU test."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "self-corr-pose_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

# SURVEY 8 rows in the order their evidence matters: a1-a4 rasteriser (vs oracle, vs the reference's own kernels), a7-a10 correspondence,
# a11-a12 ViT, a5-a6 render, a13 losses, a14 whole step, then the widening rows (f1 encoder, f2-f4), the loop-level tests, and the
# co-residency screen last.  SCP_TEST_ORDER=alpha keeps pytest's alphabetical order.
ORDER = ["test_softras_gpu", "test_softras_ref_gpu", "test_corr", "test_pretrained_golden", "test_vit_gpu", "test_split_accuracy_gpu",
         "test_render_golden", "test_project", "test_fused_losses", "test_losses_golden", "test_step_gpu", "test_step_golden",
         "test_conv_gpu", "test_fused_conv", "test_fused_bn", "test_imgops", "test_posefit", "test_data", "test_parallel",
         "test_graphed_gpu", "test_coresidency_gpu"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "child_process: loop-level GPU test (DataLoader workers, whole train / test loops): runs in a process "
                                       "of its own with a hard timeout, so that whatever it does to the device cannot take the suite with it")


def pytest_collection_modifyitems(config, items):
    """gpu-marked tests SKIP (not fail) where there is no GPU, so `pytest tests/` is meaningful on a CPU box too"""
    import torch
    if os.environ.get("SCP_TEST_ORDER", "importance") != "alpha":
        rank = {name: i for i, name in enumerate(ORDER)}
        items.sort(key=lambda it: rank.get(os.path.splitext(os.path.basename(str(it.fspath)))[0], len(ORDER) - 1.5))   # stable
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs a GPU (marked gpu)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.hookimpl(hookwrapper=True)
def pytest_runtest_call(item):
    if "gpu" not in item.keywords:
        yield
        return
    import torch
    if not torch.cuda.is_available():
        yield
        return
    import stall_diag
    with stall_diag.Watch(item.nodeid):
        yield


CHILD_TIMEOUT = float(os.environ.get("SCP_CHILD_TIMEOUT", "170"))


@pytest.hookimpl(tryfirst=True)
def pytest_pyfunc_call(pyfuncitem):
    """`child_process` tests: the parent runs `pytest <nodeid>` in a fresh interpreter (its own HIP context, its own DataLoader workers)
    under a hard limit and reports the child's verdict; a stalled child is killed with its process group, its stall report
    (tests/stall_diag.py, printed by the child after 75 s) is part of the failure message"""
    if pyfuncitem.get_closest_marker("child_process") is None or os.environ.get("SCP_TEST_CHILD") == "1":
        return None
    import signal
    import subprocess
    import torch
    if not torch.cuda.is_available():
        return None
    env = dict(os.environ, SCP_TEST_CHILD="1")
    cmd = [sys.executable, "-m", "pytest", pyfuncitem.nodeid, "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider", "--timeout=%d" % int(CHILD_TIMEOUT - 10)]
    proc = subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, start_new_session=True)
    try:
        out, _ = proc.communicate(timeout=CHILD_TIMEOUT)
    except subprocess.TimeoutExpired:
        os.killpg(proc.pid, signal.SIGKILL)
        out, _ = proc.communicate()
        pytest.fail("child process of %s exceeded %.0f s and was killed\n%s" % (pyfuncitem.nodeid, CHILD_TIMEOUT, (out or "")[-8000:]), pytrace=False)
    if proc.returncode != 0:
        pytest.fail("child process of %s failed (rc %d)\n%s" % (pyfuncitem.nodeid, proc.returncode, (out or "")[-8000:]), pytrace=False)
    return True

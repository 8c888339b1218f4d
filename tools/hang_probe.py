"""tools/hang_probe.py -- pytest plugin (PYTHONPATH=tools, -p hang_probe): when tests/test_data.py::test_trainer_train_loop_on_disk_dataset
has run for SCP_PROBE_AFTER seconds (default 150), report which HIP streams / events of the process are still busy, then leave.
Diagnostic for the round-5 full-suite stall; not part of any shipped path."""
import gc
import os
import subprocess
import sys
import threading
import time

import pytest
import torch

AFTER = float(os.environ.get("SCP_PROBE_AFTER", "150"))


def _report():
    out = sys.__stderr__
    print("\n==== hang_probe: test still running after %.0f s" % AFTER, file=out)
    try:
        print(subprocess.run(["rocm-smi", "-u"], capture_output=True, text=True, timeout=30).stdout[-400:], file=out)
    except Exception as e:
        print("rocm-smi:", e, file=out)
    names = {}
    for o in gc.get_objects():
        try:
            d = getattr(o, "__dict__", None)
            if isinstance(d, dict):
                for k, v in d.items():
                    if isinstance(v, (torch.cuda.Stream, torch.cuda.Event)):
                        names.setdefault(id(v), "%s.%s" % (type(o).__name__, k))
        except Exception:
            pass
    seen = set()
    for o in gc.get_objects():
        if isinstance(o, torch.cuda.Stream) and o.cuda_stream not in seen:
            seen.add(o.cuda_stream)
            print("stream %#x %-40s idle=%s" % (o.cuda_stream, names.get(id(o), "?"), o.query()), file=out)
    print("default stream idle=%s" % torch.cuda.default_stream().query(), file=out)
    n_ev = busy = 0
    for o in gc.get_objects():
        if isinstance(o, torch.cuda.Event):
            n_ev += 1
            try:
                if not o.query():
                    busy += 1
                    print("event busy: %s" % names.get(id(o), "?"), file=out)
            except Exception as e:
                print("event query failed:", e, file=out)
    print("events: %d, busy %d" % (n_ev, busy), file=out)
    out.flush()
    os._exit(7)


@pytest.hookimpl(hookwrapper=True)
def pytest_runtest_call(item):
    if item.name != "test_trainer_train_loop_on_disk_dataset":
        yield
        return
    done = threading.Event()

    def watch():
        if not done.wait(AFTER):
            _report()
    threading.Thread(target=watch, daemon=True).start()
    yield
    done.set()

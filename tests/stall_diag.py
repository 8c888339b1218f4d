"""tests/stall_diag.py -- what is the device doing when a GPU test stops making progress?

A watchdog thread per test (armed by tests/conftest.py on GPU boxes): a test still running after SCP_STALL_AFTER seconds (default 75, below
pytest.ini's timeout) gets a report on stderr and in gpurun_out/stall_<test>.txt BEFORE pytest-timeout ends the run:
  * the Python stack of every thread (faulthandler);
  * every torch stream / event of the process: busy or idle, and which attribute holds it;
  * KFD's view of the process (/sys/class/kfd/kfd/proc/<pid>): queues, per-GPU `cu_occupancy` (waves resident = a kernel that does not end;
    zero = queues evicted / nothing dispatched) and `evicted_ms`, sampled twice one second apart;
  * rocm-smi use / pids and the kernel log tail where readable;
  * rocgdb attached to the process: agents, queues, the dispatches in flight with their kernel names, and the first wavefronts with their PCs.
Round 5's GPU suite stalled inside Trainer.train with none of this on record; this file is what makes the next stall a
diagnosis instead of a timeout.  Test infrastructure only."""
import faulthandler
import gc
import glob
import os
import subprocess
import sys
import threading
import time

AFTER = float(os.environ.get("SCP_STALL_AFTER", "75"))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sh(cmd, timeout=60):
    try:
        r = subprocess.run(cmd, shell=True, capture_output=True, text=True, timeout=timeout)
        return (r.stdout + r.stderr)[-6000:]
    except Exception as e:                                      # noqa: BLE001
        return "%s: %r\n" % (cmd, e)


def kfd_state(pid):
    base = "/sys/class/kfd/kfd/proc/%d" % pid
    out = []
    if not os.path.isdir(base):
        return "no %s\n" % base
    for f in sorted(glob.glob(base + "/stats_*/*")) + sorted(glob.glob(base + "/vram_*")) + sorted(glob.glob(base + "/sdma_*")):
        try:
            out.append("%s = %s" % (f[len(base) + 1:], open(f).read().strip()))
        except Exception as e:                                  # noqa: BLE001
            out.append("%s: %r" % (f, e))
    q = sorted(glob.glob(base + "/queues/*"))
    out.append("queues: %d" % len(q))
    for d in q[:48]:
        vals = []
        for k in ("type", "size", "gpuid"):
            try:
                vals.append("%s=%s" % (k, open(os.path.join(d, k)).read().strip()))
            except Exception:                                   # noqa: BLE001
                pass
        out.append("  queue %s %s" % (os.path.basename(d), " ".join(vals)))
    return "\n".join(out) + "\n"


def torch_state():
    import torch
    lines, names = [], {}
    for o in gc.get_objects():
        try:
            d = getattr(o, "__dict__", None)
            if isinstance(d, dict):
                for k, v in d.items():
                    if isinstance(v, (torch.cuda.Stream, torch.cuda.Event)):
                        names.setdefault(id(v), "%s.%s" % (type(o).__name__, k))
        except Exception:                                       # noqa: BLE001
            pass
    seen = set()
    for o in gc.get_objects():
        try:
            if isinstance(o, torch.cuda.Stream) and o.cuda_stream not in seen:
                seen.add(o.cuda_stream)
                lines.append("stream %#x %-44s idle=%s" % (o.cuda_stream, names.get(id(o), "?"), o.query()))
        except Exception as e:                                  # noqa: BLE001
            lines.append("stream query failed: %r" % (e,))
    try:
        lines.append("default stream idle=%s" % torch.cuda.default_stream().query())
    except Exception as e:                                      # noqa: BLE001
        lines.append("default stream query failed: %r" % (e,))
    n_ev = busy = 0
    for o in gc.get_objects():
        if isinstance(o, torch.cuda.Event):
            n_ev += 1
            try:
                if not o.query():
                    busy += 1
                    lines.append("event busy: %s" % names.get(id(o), "?"))
            except Exception as e:                              # noqa: BLE001
                lines.append("event query failed: %r" % (e,))
    lines.append("events: %d, busy %d" % (n_ev, busy))
    return "\n".join(lines) + "\n"


def report(what, path=None, gdb=True):
    """write the report section by section (a section that hangs -- rocgdb on a wedged device -- must not cost the ones before it)"""
    pid = os.getpid()
    out = sys.__stderr__
    fh = None
    if path:
        try:
            os.makedirs(os.path.dirname(path), exist_ok=True)
            fh = open(path, "w")
        except Exception as e:                                  # noqa: BLE001
            print("stall_diag: could not open %s: %r" % (path, e), file=out)

    def emit(text):
        for f in (out, fh):
            if f is not None:
                try:
                    f.write(text)
                    f.flush()
                    os.fsync(f.fileno())
                except Exception:                               # noqa: BLE001
                    pass

    emit("\n==== stall_diag: %s still running after %.0f s (pid %d) ====\n" % (what, AFTER, pid))
    emit("---- python stacks\n")
    for f in (out, fh):
        if f is not None:
            faulthandler.dump_traceback(file=f, all_threads=True)
            f.flush()
    emit("---- KFD (first sample)\n" + kfd_state(pid))
    try:
        emit("---- torch streams / events\n" + torch_state())
    except Exception as e:                                      # noqa: BLE001
        emit("torch_state failed: %r\n" % (e,))
    time.sleep(1.0)
    emit("---- KFD (1 s later)\n" + kfd_state(pid))
    emit("---- children\n" + _sh("ps -o pid,ppid,stat,etime,wchan:24,cmd --ppid %d; ps -L -o tid,stat,wchan:28,comm -p %d | head -60" % (pid, pid)))
    emit("---- rocm-smi\n" + _sh("rocm-smi --showuse --showpids --showmemuse 2>&1 | tail -30", 60))
    emit("---- dmesg\n" + _sh("dmesg 2>&1 | tail -40", 20))
    emit("---- debugfs (best effort)\n" + _sh("mountpoint -q /sys/kernel/debug || mount -t debugfs none /sys/kernel/debug 2>&1; "
                                              "for f in /sys/kernel/debug/kfd/hqds /sys/kernel/debug/kfd/rls /sys/kernel/debug/dri/*/amdgpu_fence_info; do "
                                              "echo \"== $f\"; head -c 6000 $f 2>&1; done", 30))
    if gdb and os.path.exists("/opt/rocm/bin/rocgdb"):
        cmd = ("timeout -s KILL 80 /opt/rocm/bin/rocgdb -p %d -batch -ex 'set pagination off' -ex 'info agents' -ex 'info queues' -ex 'info dispatches' "
               "-ex 'info threads' 2>&1 | grep -v '^\\[New\\|^warning: \\|Thread debugging\\|^Reading symbols\\|No such file' | head -260" % pid)
        emit("---- rocgdb\n")
        emit(_sh(cmd, 90))
    emit("==== stall_diag: end of report\n")
    if fh is not None:
        fh.close()


class Watch:
    """with Watch(name): ...  -- report once if the block is still running after AFTER seconds"""

    def __init__(self, name, after=None, exit_code=None):
        self.name, self.after, self.exit_code = name, AFTER if after is None else after, exit_code
        self.done = threading.Event()

    def _run(self):
        if self.done.wait(self.after):
            return
        safe = "".join(c if c.isalnum() or c in "-_." else "_" for c in self.name)[-120:]
        report(self.name, os.path.join(ROOT, "gpurun_out", "stall_%s.txt" % safe))
        if self.exit_code is not None:
            os._exit(self.exit_code)

    def __enter__(self):
        threading.Thread(target=self._run, daemon=True, name="stall_diag").start()
        return self

    def __exit__(self, *exc):
        self.done.set()
        return False

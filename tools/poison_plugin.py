"""tools/poison_plugin.py -- pytest plugin (python -m pytest -p poison_plugin with PYTHONPATH=tools): before every test, fill the caching
allocator's free device blocks (and some page-locked host blocks) with a byte pattern, so that a kernel reading memory nobody wrote sees
garbage instead of the zeros a fresh process happens to hand out.  SCP_POISON=7f (default; floats 3.4e38, ints 2.1e9) or ff (NaN, -1).
Diagnostic only: finds reads of uninitialised memory whose symptom (a stall, a NaN) depends on what ran earlier in the process."""
import os

import torch

PATTERN = int(os.environ.get("SCP_POISON", "7f"), 16)
SMALL = [512, 1024, 2048, 4096, 8192, 16384, 65536, 262144, 1 << 20]
LARGE = [2 << 20, 4 << 20, 8 << 20, 20 << 20, 64 << 20, 200 << 20, 1 << 30]


def pytest_runtest_setup(item):
    if not torch.cuda.is_available():
        return
    torch.cuda.synchronize()
    keep = []
    for n in SMALL:
        keep += [torch.full((n,), PATTERN, dtype=torch.uint8, device="cuda") for _ in range(64)]
    for n in LARGE:
        keep += [torch.full((n,), PATTERN, dtype=torch.uint8, device="cuda") for _ in range(3)]
    for n in (4096, 1 << 16, 1 << 20, 16 << 20):
        keep += [torch.full((n,), PATTERN, dtype=torch.uint8).pin_memory() for _ in range(4)]
    torch.cuda.synchronize()
    del keep

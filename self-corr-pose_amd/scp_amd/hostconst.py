"""Small host-built tensors without a host<->device synchronisation.

`torch.tensor(list, device="cuda")` stages through pageable memory, which blocks the host until the
stream has drained up to the copy -- one such call in the middle of the step serialises the host with the
device (measured: 5 ms per call inside the rotation-cycle branch).  These helpers stage through pinned
memory with an asynchronous copy, and memoise true constants per device."""
import functools

import torch


def small_tensor(values, dtype, device):
    """per-call values (e.g. this iteration's rotation angle)"""
    device = torch.device(device)
    t = torch.tensor(values, dtype=dtype)
    if device.type != "cuda":
        return t.to(device)
    return t.pin_memory().to(device, non_blocking=True)


@functools.lru_cache(maxsize=512)
def _memo(values, dtype, device_str):
    return small_tensor(values, dtype, device_str)


def _freeze(v):
    return tuple(_freeze(x) for x in v) if isinstance(v, (list, tuple)) else float(v)


def const_tensor(values, dtype, device):
    """memoised constant (light colours, default camera vectors ...); callers must not write into it"""
    if torch.is_tensor(values):
        return values.to(device=device, dtype=dtype)
    if hasattr(values, "tolist"):
        values = values.tolist()
    if not isinstance(values, (list, tuple)):
        values = (values,)
    return _memo(_freeze(values), dtype, str(torch.device(device)))

"""Optional per-launch timing of the tensor-product kernel with HIP events on the launch stream
(``torch.cuda.Event`` on the current stream == the stream handed to the C ABI).  Off by default."""
from __future__ import annotations

import torch

_records = None  # list of (tag, rows, algorithmic_bytes, start_event, end_event) when enabled


def enable():
    global _records
    _records = []


def disable():
    global _records
    _records = None


def enabled() -> bool:
    return _records is not None


def begin():
    ev = torch.cuda.Event(enable_timing=True)
    ev.record()
    return ev


def end(tag, rows, nbytes, start, flops=0, kernel="", executed_flops=None):
    ev = torch.cuda.Event(enable_timing=True)
    ev.record()
    _records.append((tag, rows, nbytes, start, ev, flops, kernel, executed_flops))


def summary():
    """-> {tag: dict(launches, rows, bytes_per_launch, avg_ms, total_ms)} (synchronises)."""
    torch.cuda.synchronize()
    out = {}
    for tag, rows, nbytes, s, e, flops, kernel, xfl in _records or []:
        d = out.setdefault(tag, {"launches": 0, "rows": rows, "bytes_per_launch": nbytes, "total_ms": 0.0,
                                 "flops_per_launch": flops, "kernel": kernel,
                                 "executed_flops_per_launch": xfl if xfl is not None else flops})
        d["launches"] += 1
        d["total_ms"] += s.elapsed_time(e)
    for d in out.values():
        d["avg_ms"] = d["total_ms"] / d["launches"]
    return out

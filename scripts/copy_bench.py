#!/usr/bin/env python3
"""Staged (xfer.hip) against plain hipMemcpyAsync copies of pageable memory (tuning aid)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open3d_amd import _lib
ctx = _lib.Context(0)
f = ctx._lib.o3d_debug_copy_bench
for mb in (8, 16, 24, 48, 72, 128, 240):
    row = []
    for to_dev in (1, 0):
        for staged in (1, 0):
            ms = C.c_double()
            ctx.check(f(ctx.handle, mb << 20, to_dev, staged, 5, C.byref(ms)))
            row.append("%s %s %.2f ms (%.1f GB/s)" % ("H2D" if to_dev else "D2H", "staged" if staged else "plain ", ms.value, (mb << 20) / ms.value / 1e6))
    print("%4d MiB: " % mb + " | ".join(row), flush=True)

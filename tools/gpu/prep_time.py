#!/usr/bin/env python3
"""GPU time of rt_prepare_kernel (through rt_test_tile_order) for frame-sized tile counts."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import raytracing_c_amd as rt
assert rt.lib.rt_init(0) == 0
for n in (4096, 16384, 32400, 130560):
    rng = np.random.default_rng(n)
    cost = rng.lognormal(6, 2.5, n).astype(np.uint32); cost[: n // 3] = 4096
    order = np.zeros(n, np.uint32)
    rt.lib.rt_test_tile_order(n, cost.ctypes.data, order.ctypes.data)
    t0 = time.perf_counter()
    for _ in range(20):
        rt.lib.rt_test_tile_order(n, cost.ctypes.data, order.ctypes.data)
    print(n, "tiles:", (time.perf_counter() - t0) / 20 * 1e3, "ms per call incl. allocations and copies")

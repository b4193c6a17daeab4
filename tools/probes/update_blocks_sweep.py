"""Sweep of the update kernel's grid (option update_blocks) on BASELINE configs[1] with the row-dictionary product."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from fenicssolver_amd import backend as B
B.init(0)
prob = bench.Problem(99, 99, 99, (1.0, 1.0, 1.0), (0, 100), 2, 0, 1)
prob.pipelined = False
for blocks in (256, 384, 512, 640, 768, 1024, 2048):
    B.set_option("update_blocks", blocks)
    for _ in range(3):
        prob.step(1e-8)
    B.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        st, _ = prob.step(1e-8)
    B.synchronize()
    ms = (time.perf_counter() - t0) * 100
    print("update_blocks %4d: %.3f ms / step, product %.1f us, update %.1f us, %d iterations" % (blocks, ms, st["spmv_ms"] * 1e3, st["update_ms"] * 1e3, st["iterations"]), flush=True)

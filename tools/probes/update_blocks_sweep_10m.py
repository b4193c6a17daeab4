import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from fenicssolver_amd import backend as B
B.init(0)
prob = bench.Problem(215, 215, 215, (1.0, 1.0, 1.0), (0, 216), 2, 0, 1)
prob.pipelined = False
for blocks in (512, 1024, 2048, 4096):
    B.set_option("update_blocks", blocks)
    prob.step(1e-8)
    B.synchronize()
    t0 = time.perf_counter()
    for _ in range(2):
        st, _ = prob.step(1e-8)
    B.synchronize()
    ms = (time.perf_counter() - t0) * 500
    print("update_blocks %4d: %.3f ms / step, product %.1f us, update %.1f us, %d iterations" % (blocks, ms, st["spmv_ms"] * 1e3, st["update_ms"] * 1e3, st["iterations"]), flush=True)

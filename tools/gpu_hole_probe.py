"""Does the GPU stall for tens of ms when load resumes after an idle period?  (clock / power-state transitions)
Runs trains of ~1 ms kernels after idle periods of different length and prints every hole > 3 ms between the events."""
import sys
import time

import torch

x = torch.zeros(96 * 1024 * 1024, dtype=torch.float64, device="cuda")   # 768 MB: x.add_ is ~0.3 ms
s = torch.cuda.current_stream()


def train(n_iter, idle_s, per_iter=3):
    torch.cuda.synchronize()
    time.sleep(idle_s)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n_iter + 1)]
    ev[0].record()
    for k in range(n_iter):
        for _ in range(per_iter):
            x.add_(1.0)
        ev[k + 1].record()
        if k >= 1:
            ev[k - 1].synchronize()     # the host stays one iteration ahead, as the solver does
    torch.cuda.synchronize()
    dt = [ev[k].elapsed_time(ev[k + 1]) for k in range(n_iter)]
    holes = [(k, round(sum(dt[:k]), 1), round(d, 1)) for k, d in enumerate(dt) if d > 3.0]
    med = sorted(dt)[len(dt) // 2]
    print("idle %.3f s: %d iterations, median %.3f ms, first %.3f ms, total %.1f ms, holes (k, at ms, ms): %s" % (
        idle_s, n_iter, med, dt[0], sum(dt), holes))


for idle in (0.0, 0.02, 0.05, 0.1, 0.1, 0.3, 1.0, 0.1, 0.05):
    train(150, idle)

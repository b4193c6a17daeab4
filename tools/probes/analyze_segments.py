"""Offline prototype (CPU, numpy) of the row-segment form of the dictionary product on a dumped CSR (dump_p2_csr.py):
segments = maximal runs of rows whose offset sets are nested with their neighbour's; items = chunks of <= 128 rows; item list =
offsets of the widest row; plan = runs of <= 3 consecutive offsets, 8 per round; class = coefficients in plan-slot layout."""
import sys
import numpy as np

d = np.load(sys.argv[1])
rp, ci, va = d['rp'].astype(np.int64), d['ci'].astype(np.int64), d['va']
n = len(rp) - 1
rows = np.repeat(np.arange(n), np.diff(rp))
diag = np.zeros(n)
diag[rows[ci == rows]] = va[ci == rows]
sv = va / np.sqrt(diag[rows] * diag[ci])
off = ci - rows
offs = [off[rp[r]:rp[r + 1]] for r in range(n)]          # STRUCTURAL offsets (zeros of eliminated columns stay)
sets = [set(o.tolist()) for o in offs]
cont = np.zeros(n, dtype=bool)
for r in range(1, n):
    a, b = sets[r - 1], sets[r]
    cont[r] = a <= b or b <= a
starts = np.nonzero(~cont)[0]
ends = np.append(starts[1:], n)
print("rows", n, "segments", len(starts), "mean len %.1f" % (n / len(starts)), "max", (ends - starts).max())
items = []
for s, e in zip(starts, ends):
    for a in range(s, e, 128):
        items.append((a, min(a + 128, e)))
print("items", len(items), "mean rows %.1f" % (n / len(items)))
bad = 0
rounds_hist = {}
classes = {}
plans = {}
nslots_max = 0
for a, e in items:
    w = max(range(a, e), key=lambda r: len(offs[r]))
    L = offs[w]
    Ls = sets[w]
    if any(not (sets[r] <= Ls) for r in range(a, e)):
        bad += 1
        continue
    runs = []
    k = 0
    while k < len(L):
        ln = 1
        while k + ln < len(L) and ln < 3 and L[k + ln] == L[k + ln - 1] + 1:
            ln += 1
        runs.append((L[k], ln))
        k += ln
    nr = len(runs)
    rounds = 1 + max(0, (nr - 7 + 7) // 8)
    rounds_hist[rounds] = rounds_hist.get(rounds, 0) + 1
    slot_of = {}
    for j, (st, ln) in enumerate(runs):
        jj = j if j < 7 else j + 1          # slot 7 of round 0 is z
        for t in range(ln):
            slot_of[st + t] = 3 * jj + t
    nslots_max = max(nslots_max, 3 * (len(runs) + 1))
    plans[tuple(runs)] = plans.get(tuple(runs), 0) + 1
    for r in range(a, e):
        key = tuple(sorted((slot_of[o], v) for o, v in zip(offs[r].tolist(), sv[rp[r]:rp[r + 1]].tolist()) if v != 0.0))
        classes[key] = classes.get(key, 0) + 1
print("items whose widest row does not cover all rows:", bad)
print("rounds per item:", rounds_hist, "distinct plans", len(plans), "slots max", nslots_max)
print("classes (plan-slot layout):", len(classes))
# distinct classes per item
cnt = []

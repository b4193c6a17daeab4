import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np
from fenicssolver_amd import backend as B
B.init(0)
n = 99
mesh = B.DeviceMesh.box(n, n, n)
V = B.DeviceSpace(mesh, 1)
A = B.DeviceMatrix(V)
A.assemble(stiffness=20.0)
rp, ci, va, shape = A.to_csr()
N = shape[0]
# interior rows: 15 entries; group by offsets
cnt = np.diff(rp)
rows = np.nonzero(cnt == 15)[0]
vals = va[(rp[rows][:, None] + np.arange(15))]
offs = ci[(rp[rows][:, None] + np.arange(15))] - rows[:, None]
print("interior rows", len(rows), "distinct offset patterns", len(np.unique(offs, axis=0)))
u, inv, c = np.unique(vals, axis=0, return_inverse=True, return_counts=True)
print("distinct value rows (bitwise)", len(u), "largest classes", np.sort(c)[::-1][:5])
ref = vals[len(vals)//2]
print("max rel deviation from the middle row", np.abs(vals - ref).max() / np.abs(ref).max())
# per 64-row slice: fraction of slices whose rows are all bitwise equal
full = np.zeros(N, dtype=bool); full[rows] = True
key = np.zeros(N, dtype=np.int64); key[rows] = inv + 1
ns = N // 64
k2 = key[:ns*64].reshape(ns, 64)
uni = (k2.min(axis=1) == k2.max(axis=1)) & (k2.min(axis=1) > 0)
print("slices", ns, "bitwise-uniform slices", int(uni.sum()))

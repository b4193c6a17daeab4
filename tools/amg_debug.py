"""Debug aid: per-level eigenvalue estimate vs scipy, and definiteness of the V-cycle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spl
from fenicssolver_amd import backend as gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_gpu_amg as T
gpu.init(0)
for clamp in (None, (0,)):
    V, A, b, Ab, bb, rbm = T._elasticity(gpu, clamp_components=clamp)
    amg = gpu.AMG(A, nullspace=rbm, coarse_size=100)
    info = amg.info()
    print("clamp", clamp, info)
    for l in range(info["levels"]):
        li = amg.level_info(l)
        Al = amg.level_matrix(l, "A")
        d = Al.diagonal()
        S = sp.diags(1 / np.sqrt(np.abs(d))) @ Al @ sp.diags(1 / np.sqrt(np.abs(d)))
        w = np.linalg.eigvalsh(S.toarray()) if S.shape[0] < 3000 else spl.eigsh(S, k=1, which="LA")[0]
        print(" level", l, li, "true lmax(D^-1A)=%.4f min=%.3e  neg diag: %d" % (w.max(), w.min(), (d <= 0).sum()))
    n = V.n_owned
    r = gpu.DeviceVector(n); z = gpu.DeviceVector(V.n_local)
    M = np.empty((n, n)); e = np.zeros(n)
    for i in range(n):
        e[:] = 0; e[i] = 1; r.set(e); amg.apply(r, z); M[:, i] = z.get()[:n]
    print(" sym err", np.abs(M - M.T).max() / np.abs(M).max(), "eig min/max of sym(M):", np.linalg.eigvalsh(0.5 * (M + M.T))[[0, -1]])
    ev = np.linalg.eigvals(M @ Ab.toarray()).real
    print(" eig(MA) min %.3e max %.3f" % (ev.min(), ev.max()))

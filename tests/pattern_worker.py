"""Run by tests/test_gpu_kernels.py in two processes (FS_PATTERN_BY_ROWS unset / 0): the sparsity patterns and assembled operators of
a few CG1 spaces, written to an .npz file (argv[1]) for comparison."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fenicssolver_amd import backend as B  # noqa: E402
from oracle import fem_oracle as fo  # noqa: E402

B.init(0)
out = {}


def record(name, mesh, ncomp, degree=1, **form):
    V = B.DeviceSpace(mesh, ncomp, degree)
    A = B.DeviceMatrix(V)
    A.assemble(**form)
    rp, ci, va, shape = A.to_csr()
    out[name + "_rp"], out[name + "_ci"], out[name + "_va"] = rp, ci, va
    x = B.DeviceVector(V.n_local)           # (dofs: nodes x components)
    y = B.DeviceVector(V.n_owned)
    x.set(np.linspace(-1.0, 2.0, x.n))
    A.spmv(x, y)
    out[name + "_y"] = y.get()


record("box", B.DeviceMesh.box(13, 9, 11), 1, stiffness=2.0, mass=0.5)
record("box_vector", B.DeviceMesh.box(7, 6, 5), 3, lame=(1.0, 1.5))
# the same kind of cube as a mesh file delivers it: vertices and cells in random order (high-valence rows, no structure)
co, ce = fo.box_mesh((0, 0, 0), (1.0, 0.7, 1.3), 9, 8, 7)
rng = np.random.default_rng(4)
perm = rng.permutation(len(co))
inv = np.empty_like(perm)
inv[perm] = np.arange(len(co))
ce2 = inv[ce][rng.permutation(len(ce))]
record("shuffled", B.DeviceMesh(co[perm], ce2.astype(np.int32)), 1, stiffness=1.0)
# triangles
co2, ce2d = fo.rectangle_mesh((0.0, 0.0), (2.0, 1.0), 17, 11)
record("rectangle", B.DeviceMesh(co2, ce2d.astype(np.int32)), 1, stiffness=3.0, mass=1.0)
# CG2 (10 / 6 nodes per cell: the larger per-row set)
record("box_p2", B.DeviceMesh.box(6, 5, 4), 1, 2, stiffness=1.5, mass=0.25)
record("rectangle_p2", B.DeviceMesh(co2, ce2d.astype(np.int32)), 1, 2, stiffness=1.0)
# a triangle fan: the centre has 40 neighbours, more than the per-row set of a CG1 space holds - the whole space goes back to the
# sorted-keys path (and says so)
m = 40
ang = 2.0 * np.pi * np.arange(m) / m
cof = np.vstack([[0.0, 0.0], np.stack([np.cos(ang), np.sin(ang)], axis=1)])
cef = np.stack([np.zeros(m, dtype=np.int32), 1 + np.arange(m, dtype=np.int32), 1 + (np.arange(m, dtype=np.int32) + 1) % m], axis=1)
record("fan", B.DeviceMesh(cof, np.sort(cef, axis=1).astype(np.int32)), 1, stiffness=1.0)
np.savez(sys.argv[1], **out)
print("ok")

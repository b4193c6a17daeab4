"""Assembly time of the block / CG2 gather kernels at the BASELINE sizes: configs[2] elasticity (P1 vector), configs[3] P2 scalar."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fenicssolver_amd import backend as B
B.init(0)


def timed(fn, reps=5):
    for _ in range(2):
        fn()
    B.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    B.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("all", "elasticity"):
    mesh = B.DeviceMesh.box(472, 59, 59, (0, 0, 0), (10., 1., 1.)); V = B.DeviceSpace(mesh, 3); A = B.DeviceMatrix(V)
    b = B.DeviceVector(V.n_owned)
    print("configs[2] elasticity matrix  %.3f ms" % timed(lambda: A.assemble(lame=(7.9e10, 9.2e10))), flush=True)
    print("configs[2] body-force vector  %.3f ms" % timed(lambda: B.assemble_vector(V, b, vector_value=(0, 0, -7.8e4))), flush=True)
    for h in (b, A, V, mesh):
        h.close()
if which in ("all", "p2"):
    mesh = B.DeviceMesh.box(107, 107, 107); V = B.DeviceSpace(mesh, 1, 2); A = B.DeviceMatrix(V); b = B.DeviceVector(V.n_owned)
    print("configs[3] P2 stiffness matrix %.3f ms" % timed(lambda: A.assemble(stiffness=20.0)), flush=True)
    print("configs[3] P2 source vector    %.3f ms" % timed(lambda: B.assemble_vector(V, b, source=1.0)), flush=True)
if which in ("all", "p1src"):
    import numpy as np
    mesh = B.DeviceMesh.box(99, 99, 99); V = B.DeviceSpace(mesh, 1); b = B.DeviceVector(V.n_owned)
    fn = np.linspace(0, 1, V.n_owned)
    print("configs[1] P1 source vector (const) %.3f ms" % timed(lambda: B.assemble_vector(V, b, source=1.0), 20), flush=True)
    print("configs[1] P1 source vector (nodal) %.3f ms" % timed(lambda: B.assemble_vector(V, b, source=("nodal", fn)), 20), flush=True)

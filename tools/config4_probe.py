import sys, time
sys.path.insert(0, '.')
import numpy as np
from fenicssolver_amd import backend as B
B.init(0)
for n in ([int(a) for a in sys.argv[1:]] or (26, 53, 107)):
    t0=time.perf_counter(); mesh = B.DeviceMesh.box(n,n,n); V = B.DeviceSpace(mesh,1,degree=2); B.synchronize(); t1=time.perf_counter()
    A = B.DeviceMatrix(V); b = B.DeviceVector(V.n_owned); x = B.DeviceVector(V.n_owned)
    xyz,_,_ = mesh.get(); edges = V.edges().astype(np.int64)
    X = np.concatenate([xyz, 0.5*(xyz[edges[:,0]]+xyz[edges[:,1]])])
    lo = np.nonzero(X[:,2]==0.0)[0]; hi = np.nonzero(X[:,2]==1.0)[0]
    dofs = np.concatenate([lo,hi]); vals = np.concatenate([np.full(len(lo),350.),np.full(len(hi),300.)])
    for rep in range(2):
        t2=time.perf_counter(); A.assemble(stiffness=20.0); b.fill(0.0); A.apply_dirichlet(b,dofs,vals,True); B.synchronize(); t3=time.perf_counter()
        st = B.krylov_solve(A,b,x,rtol=1e-8,max_iter=50000); t4=time.perf_counter()
        print('n=%d dofs %d nnz %d (%.1f/row, max %d) stored %d dia %d/%d symbolic %.0f ms asm+bc %.2f ms solve %.1f ms it %d true %.2e spmv %.1f us upd %.1f us DOF/s %.3g'%(n,V.n_owned,V.nnz,V.nnz/V.n_owned,0,V.sell_entries,V.n_dia_slices,V.n_slices,(t1-t0)*1e3,(t3-t2)*1e3,(t4-t3)*1e3,st['iterations'],st['true_rel_residual'],st['spmv_ms']*1e3,st['update_ms']*1e3,V.n_owned/(t4-t2)))
    print('   err vs exact', np.abs(x.get()-(350-50*X[:,2])).max(), 'alg spmv GB/s', st['spmv_bytes']/st['spmv_ms']/1e6, 'streamed GB/s', (V.spmv_matrix_bytes+24*V.n_owned)/st['spmv_ms']/1e6)

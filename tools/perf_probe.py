import sys, time, json
sys.path.insert(0, '.')
import numpy as np
from fenicssolver_amd import backend as B
B.init(0)
print(B.device_info())
for n in (99, 215):
    t0=time.perf_counter(); mesh = B.DeviceMesh.box(n,n,n); B.synchronize(); t1=time.perf_counter()
    V = B.DeviceSpace(mesh, 1); B.synchronize(); t2=time.perf_counter()
    A = B.DeviceMatrix(V); b = B.DeviceVector(V.n_owned); x = B.DeviceVector(V.n_owned)
    P=(n+1)**2
    lo=np.arange(P); hi=np.arange(n*P,(n+1)*P)
    dofs=np.concatenate([lo,hi]); vals=np.concatenate([np.full(P,350.),np.full(P,300.)])
    for rep in range(3):
        B.synchronize(); t3=time.perf_counter()
        A.assemble(stiffness=20.0); t4=time.perf_counter()
        b.fill(0.0); A.apply_dirichlet(b,dofs,vals,True); B.synchronize(); t5=time.perf_counter()
        st = B.krylov_solve(A,b,x,rtol=1e-8,max_iter=5000); t6=time.perf_counter()
        print(n, 'dofs',V.n_owned,'nnz',V.nnz,'sell',V.sell_entries,'mesh %.1fms symbolic %.1fms asm %.2fms bc %.2fms solve %.2fms'%((t1-t0)*1e3,(t2-t1)*1e3,(t4-t3)*1e3,(t5-t4)*1e3,(t6-t5)*1e3), st)
        bytes_ = st['spmv_bytes']
        if st['spmv_ms']>0: print('   spmv GB/s', bytes_/st['spmv_ms']/1e6, 'update GB/s', V.n_owned*96/st['update_ms']/1e6, 'DOF/s', V.n_owned/((t6-t3)))
    xl = B.DeviceVector(V.n_local, np.random.default_rng(0).standard_normal(V.n_local)); y=B.DeviceVector(V.n_owned)
    ms = A.spmv_benchmark(xl,y,50); print('   bare spmv ms',ms,'GB/s',bytes_/ms/1e6)
    sol=x.get(); xyz,_,_=mesh.get(); print('   err', np.abs(sol-(350-50*xyz[:,2])).max())

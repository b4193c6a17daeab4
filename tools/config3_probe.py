import sys, time
sys.path.insert(0, '.')
import numpy as np
from fenicssolver_amd import backend as B
B.init(0)
E, nu = 2e11, 0.27
mu = E/(2*(1+nu)); lm = E*nu/((1+nu)*(1-2*nu))
for dims in ((118,15,15),(236,30,30),(472,59,59)):
    nx,ny,nz = dims
    t0=time.perf_counter(); mesh = B.DeviceMesh.box(nx,ny,nz,(0,0,0),(10.,1.,1.)); V = B.DeviceSpace(mesh,3); B.synchronize(); t1=time.perf_counter()
    A = B.DeviceMatrix(V); b = B.DeviceVector(V.n_owned); x = B.DeviceVector(V.n_owned)
    P=(nx+1)*(ny+1)
    # clamp x=0 face: nodes with ix==0
    nodes = np.arange((nx+1)*(ny+1)*(nz+1)); left = nodes[nodes % (nx+1) == 0]
    dofs = (left[:,None]*3+np.arange(3)).ravel()
    for rep in range(2):
        t2=time.perf_counter(); A.assemble(lame=(mu,lm)); B.assemble_vector(V,b,vector_value=(0,0,-7800*10.)); A.apply_dirichlet(b,dofs,0.0,True); B.synchronize(); t3=time.perf_counter()
        st = B.krylov_solve(A,b,x,rtol=1e-8,max_iter=200000); t4=time.perf_counter()
        print(dims,'dofs',V.n_owned,'nnz',V.nnz,'dia',V.n_dia_slices,'/',V.n_slices,'symbolic %.1f ms asm %.2f ms solve %.1f ms it %d conv %d true %.2e spmv %.1f us upd %.1f us DOF/s %.3g'%((t1-t0)*1e3,(t3-t2)*1e3,(t4-t3)*1e3,st['iterations'],st['converged'],st['true_rel_residual'],st['spmv_ms']*1e3,st['update_ms']*1e3,V.n_owned/(t4-t2)))
    u = x.get().reshape(-1,3); print('   tip deflection', u[:,2].min(), 'beam theory ~', -7800*10*1*10**4/(8*E*(1/12)))

import sys, time
sys.path.insert(0, '.')
import numpy as np
from fenicssolver_amd import backend as B
B.init(0)
for n in (99, 215):
    mesh = B.DeviceMesh.box(n,n,n); V = B.DeviceSpace(mesh, 1)
    A = B.DeviceMatrix(V); b = B.DeviceVector(V.n_owned); x = B.DeviceVector(V.n_owned)
    P=(n+1)**2
    dofs=np.concatenate([np.arange(P),np.arange(n*P,(n+1)*P)]); vals=np.concatenate([np.full(P,350.),np.full(P,300.)])
    def run(tag, **opts):
        for k,v in opts.items(): B.set_option(k,v)
        best=1e9
        for rep in range(4):
            t0=time.perf_counter()
            A.assemble(stiffness=20.0); b.fill(0.0); A.apply_dirichlet(b,dofs,vals,True)
            t1=time.perf_counter()
            st=B.krylov_solve(A,b,x,rtol=1e-8,max_iter=5000)
            t2=time.perf_counter()
            best=min(best,t2-t0)
        print('n=%d %-40s asm+bc %.2f ms solve %.2f ms it %d spmv %.1f us upd %.1f us  step %.2f ms DOF/s %.3g'%(n,tag,(t1-t0)*1e3,(t2-t1)*1e3,st['iterations'],st['spmv_ms']*1e3,st['update_ms']*1e3,best*1e3,V.n_owned/best))
    run('unfused sums', cg_fuse_sums=0, spmv_blocks=2048, update_blocks=2048)
    for sb in (1024, 2048):
        for ub in (512, 1024, 2048):
            run('fused spmv_blocks=%d update_blocks=%d'%(sb,ub), cg_fuse_sums=1, spmv_blocks=sb, update_blocks=ub)
    run('fused batch 64', cg_fuse_sums=1, spmv_blocks=2048, update_blocks=1024, cg_batch=64)
    B.set_option('cg_batch',32)

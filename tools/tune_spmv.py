import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fenicssolver_amd import backend as B
B.init(0)
for n in (215, 99):
    mesh = B.DeviceMesh.box(n,n,n); V = B.DeviceSpace(mesh, 1)
    A = B.DeviceMatrix(V); A.assemble(stiffness=20.0)
    rng=np.random.default_rng(0)
    xl = B.DeviceVector(V.n_local, rng.standard_normal(V.n_local)); y=B.DeviceVector(V.n_owned, rng.standard_normal(V.n_owned))
    bytes_ = V.nnz*12 + V.n_owned*20
    print('n',n,'bytes',bytes_)
    for blocks in (512,1024,2048,4096):
        for unroll in (2,4,8,16):
            B.set_option('spmv_blocks', blocks); B.set_option('spmv_unroll', unroll)
            bare = min(A.spmv_benchmark(xl,y,30) for _ in range(3))
            fused = min(A.spmv_benchmark(xl,y,-30) for _ in range(3))
            print('  blocks %4d unroll %2d bare %.4f ms %.0f GB/s | fused %.4f ms %.0f GB/s'%(blocks,unroll,bare,bytes_/bare/1e6,fused,bytes_/fused/1e6))

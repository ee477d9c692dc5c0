import sys, time, numpy as np, ctypes as C
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from test_gpu_parity import _ldlt
for n in [1,2,3,7,31,32,33,64,65,257,600,1000,1472,2048,3000]:
    rng=np.random.default_rng(n)
    nh=max(1,(2*n)//3)
    H=rng.standard_normal((nh,nh)); H=H+H.T
    # some zero-curvature variables -> 2x2 pivots
    if nh>4: H[:nh//3,:]=0; H[:,:nh//3]=0
    J=rng.standard_normal((n-nh,nh))
    A=np.block([[H,J.T],[J,np.zeros((n-nh,n-nh))]]) if n>nh else H+np.eye(nh)*0.1
    b=rng.standard_normal(n)
    t=time.time(); sol,nneg,nzero,sec=_ldlt(A,True,b); 
    ev=np.linalg.eigvalsh(A)
    res=np.linalg.norm(A@sol-b)/(np.linalg.norm(A,2)*max(np.linalg.norm(sol),1))
    print(n, 'nneg',nneg,int(np.sum(ev<0)),'nzero',nzero,'res %.2e'%res,'factor %.4f s'%sec, flush=True)
    assert nneg==int(np.sum(ev<0)) and res<1e-9
print("ok")

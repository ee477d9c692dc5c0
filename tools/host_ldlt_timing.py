"""CPU baseline diagnostics (test infrastructure: the host build of the solver core): seconds per factorisation of the two
host KKT factorisations of bench.py cpu_baseline at order n with `threads` BLAS threads; DNLP_HOST_LDLT_TIMING=1 adds the
phase split of the blocked LDL^T.   python tools/host_ldlt_timing.py n threads"""
import sys, time, numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle.oracle_capi import OracleProblem, use_lapack, use_blocked_ldlt, set_blas_threads, dgemm_gflops
import dnlp_amd as cp
from dnlp_amd.dnlp2smooth import Dnlp2Smooth
from dnlp_amd.nlp_solver import HIPNLP, build_nlp_data
from dnlp_amd.tape import serialize
n=int(sys.argv[1]); thr=int(sys.argv[2])
print("blas threads", use_lapack(thr), "blocked", use_blocked_ldlt(True), "dgemm gflops", dgemm_gflops(2000, thr))
rng=np.random.default_rng(0)
G=rng.standard_normal((n,n)); A=(G+G.T)/2; A+= 4*np.sqrt(n)*np.outer(np.ones(n),np.ones(n))/n
x=cp.Variable(n); x.value=np.ones(n)/np.sqrt(n)+0.1*rng.standard_normal(n)/np.sqrt(n)
prob=cp.Problem(cp.Minimize(-cp.quad_form(x,A)),[cp.sum_squares(x)==1])
smooth,_=Dnlp2Smooth().apply(prob); data,_=build_nlp_data(smooth); blob=serialize(data["tape_arrays"])
for col in ("dsytrf","blocked"):
    o=OracleProblem(blob)
    for k,v in HIPNLP.DEFAULT_OPTIONS.items(): o.set_option(k,v)
    o.set_option("kkt_pivot_max_n", 10**9 if col=="dsytrf" else 0)
    o.ipm_begin(data["x0"]); t0=time.time(); rc,k=o.ipm_step(2); dt=time.time()-t0; st=o.stats()
    nf=max(int(st[1]),1)
    print(col, "iters", k, "sec", dt, "s/fact", st[4]/nf, "GF/s", (n+1)**3/3/(st[4]/nf)/1e9, "obj", o.ipm_finish()["obj_val"])

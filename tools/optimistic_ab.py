"""A/B of the option kkt_optimistic_min_n (dense KKT orders above it start on the unpivoted blocked LDL^T and fall back
to Bunch-Kaufman) on the two paper examples that are bound by the Bunch-Kaufman factorisation."""
import sys, time, json
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import dnlp_amd as cp
from paper_examples import PAPER, PAPER_LARGE, PUBLISHED
allp=dict(PAPER); allp.update(PAPER_LARGE)
for name in ["nb_phase_retrieval","nb_sparse_recovery"]:
    for optn in [None, 256]:
        for rep in range(2):
            prob=allp[name](cp)
            chain=prob._build_chain(None)
            data,inv=chain.apply(prob)
            opts=dict(PUBLISHED.get(name,{}).get("options",{}))
            if optn: opts["kkt_optimistic_min_n"]=optn
            t=time.time(); info=chain.solver.solve_via_data(data,True,False,opts); dt=time.time()-t
        print(name, "optimistic_min_n", optn, "status", info["status"], "iters", info["iterations"], "obj", info["obj_val"], "sec", round(dt,3), "nfact", int(info["stats"][1]), flush=True)

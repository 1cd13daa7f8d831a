import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as e
nle = e.load_package()
rng=np.random.default_rng(3)
out={}
for n,k in [(200,50),(196,50),(64,20),(400,60),(900,100),(30,9)]:
    U,_=np.linalg.qr(rng.standard_normal((n,n)))
    lam=np.concatenate([[1.0],0.97*0.95**np.arange(n-1)])
    A=(U*lam)@U.T; A=(A+A.T)/2
    ts=[]
    for rep in range(5):
        t=time.perf_counter(); U1,D1,r1=nle.eigen_decomposition_topk(A,k); ts.append(time.perf_counter()-t)
    w=np.linalg.eigvalsh(A)[::-1]
    print(n,k,"res",np.abs(A@U1-U1*D1).max(),"orth",np.abs(U1.T@U1-np.eye(k)).max(),"dD",np.abs(D1-w[:k]).max(),"ms %.3f"%(min(ts)*1e3))
    out[f"{n}_{k}"]=U1
np.savez(sys.argv[1],**out)

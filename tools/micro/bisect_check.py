import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as e
nle = e.load_package()
rng=np.random.default_rng(1)
for n,k in [(8,2),(16,8),(50,10),(196,50),(200,50),(200,100),(287,60),(400,50)]:
    # spectrum decaying geometrically through the cut, like Q
    U,_=np.linalg.qr(rng.standard_normal((n,n)))
    lam=np.concatenate([np.geomspace(1.5,1e-9,n-n//10), np.geomspace(9e-11,1e-14,n//10)])
    A=(U*lam)@U.T; A=(A+A.T)/2
    w=np.linalg.eigvalsh(A)[::-1]
    t=time.perf_counter(); Uk,Dk,r=nle.eigen_decomposition_topk(A,k); t1=time.perf_counter()-t
    t=time.perf_counter(); Uo,Do,ro=nle.eigen_decomposition_top(A,k); t2=time.perf_counter()-t
    res=np.abs(A@Uk-Uk*Dk).max(); orth=np.abs(Uk.T@Uk-np.eye(k)).max()
    print(n,k,"r",r,ro,int((w>=1e-10).sum()),"dD",np.abs(Dk-w[:k]).max(),np.abs(Do[:k]-w[:k]).max(),"res",res,"orth",orth,"ms %.3f %.3f"%(t1*1e3,t2*1e3))
# clustered / degenerate
n=120
A=np.diag(np.r_[np.ones(5),np.full(5,0.5),np.linspace(0.4,0,n-10)]); Qm,_=np.linalg.qr(rng.standard_normal((n,n))); A=Qm@A@Qm.T; A=(A+A.T)/2
Uk,Dk,r=nle.eigen_decomposition_topk(A,12); w=np.linalg.eigvalsh(A)[::-1]
print("clustered",np.abs(Dk-w[:12]).max(),np.abs(A@Uk-Uk*Dk).max(),np.abs(Uk.T@Uk-np.eye(12)).max(),r)
A=np.eye(40); print(nle.eigen_decomposition_topk(A,5)[1:], )
A=np.zeros((40,40)); print(nle.eigen_decomposition_topk(A,5)[1:], )

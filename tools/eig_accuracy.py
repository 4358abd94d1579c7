import sys; sys.path.insert(0,'.')
import numpy as np, time
from pyimcom_amd import linalg
rng=np.random.default_rng(1)
for n,b in ((300,3),(1000,2),(1500,2)):
    pts=rng.uniform(0,np.sqrt(n)/1.5,(b,n,2))
    A=np.exp(-((pts[:,:,None]-pts[:,None])**2).sum(-1)/(2*1.2**2))
    t=time.time(); w,Q=linalg.eigh(A); dt=time.time()-t
    for s in range(b):
        wr=np.linalg.eigvalsh(A[s])
        print(n, "dt %.3f"%dt, "eig err", np.abs(w[s]-wr).max(), "resid", np.abs(A[s]@Q[s]-Q[s]*w[s]).max(), "orth", np.abs(Q[s].T@Q[s]-np.eye(n)).max(), "anorm", np.abs(wr).max())

import sys, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, np_gp, oracle
from gp_amd import synth
LD=np.longdouble
def chol_ld(A):
    A=A.copy(); n=A.shape[0]; nb=64
    for j in range(0,n,nb):
        e=min(n,j+nb)
        # factor diagonal block unblocked
        for k in range(j,e):
            A[k,k]=np.sqrt(A[k,k]); A[k+1:e,k]/=A[k,k]
            if k+1<e: A[k+1:e,k+1:e]-=np.outer(A[k+1:e,k],A[k+1:e,k])
        if e<n:
            # A[e:,j:e] = A[e:,j:e] L^-T
            L=np.tril(A[j:e,j:e])
            B=A[e:,j:e].copy()
            for k in range(e-j):
                B[:,k]=(B[:,k]-B[:,:k]@L[k,:k])/L[k,k]
            A[e:,j:e]=B
            A[e:,e:]-=B@B.T
    return np.tril(A)
def trsm_ld(L,B):  # solve L X = B
    X=B.copy(); n=L.shape[0]; nb=64
    for j in range(0,n,nb):
        e=min(n,j+nb)
        for k in range(j,e):
            X[k]=(X[k]-L[k,j:k]@X[j:k])/L[k,k]
        if e<n: X[e:]-=L[e:,j:e]@X[j:e]
    return X
n=int(sys.argv[1]); m=1024; d=20; seed=20240601+3
hyp=synth.eval_hyps(d); X=synth.make_inputs(seed,n,d); y=synth.make_targets(seed,X,hyp); U=X[:m].copy(); jit=1e-8
nll64,g64=np_gp.fitc_nll(0,X,y,U,jit,hyp)
nllo=oracle.fitc_nll_all_cores(0,X,y,U,jit,hyp)[0] if n<=4096 else np.nan
ell=np.exp(hyp[:d]); sf2=np.exp(2*hyp[d]); sn2=np.exp(2*hyp[d+1]); mean=hyp[d+2]
Kuu64=sf2*np.exp(-0.5*np_gp.sqd_per_dim(U,U,ell).sum(0))+jit*np.eye(m)
ev=np.linalg.eigvalsh(Kuu64); print('cond Kuu+jI', ev[-1]/ev[0], ev[0], ev[-1])
t=time.time()
Xs=(X.astype(LD)/ell.astype(LD)); Us=Xs[:m]
def sq(A,B):
    return ((A*A).sum(1)[:,None]+(B*B).sum(1)[None,:]-2*(A@B.T))
# direct differences in long double (per dim accumulation)
def sqd(A,B):
    s=np.zeros((A.shape[0],B.shape[0]),LD)
    for i in range(A.shape[1]):
        t_=A[:,i][:,None]-B[:,i][None,:]; s+=t_*t_
    return s
sf2L=np.exp(2*LD(hyp[d])); sn2L=np.exp(2*LD(hyp[d+1]))
Kuu=sf2L*np.exp(-sqd(Us,Us)/2)+LD(jit)*np.eye(m,dtype=LD)
K=sf2L*np.exp(-sqd(Xs,Us)/2)
print('built',time.time()-t)
Lu=chol_ld(Kuu); print('chol',time.time()-t)
V=trsm_ld(Lu,K.T.copy()).T; print('trsm',time.time()-t)
Gam=(sn2L+sf2L-(V*V).sum(1))/sn2L; g_=1/Gam
r=(y.astype(LD)-LD(mean))
A=sn2L*Kuu+K.T@(g_[:,None]*K); print('A',time.time()-t)
LA=chol_ld(A)
p=g_*r; F=K.T@p
z=trsm_ld(LA,F[:,None])[:,0]
nll=0.5*((r@p-z@z)/sn2L+2*np.log(np.diag(LA)).sum()-2*np.log(np.diag(Lu)).sum()+np.log(Gam).sum()+(n-m)*np.log(sn2L)+n*np.log(2*LD(np.pi)))
print('n',n,'nll longdouble',repr(nll),'numpy64',repr(nll64),'oracle',repr(nllo))
print('rel numpy64-ld',float(abs(nll64-nll)/abs(nll)),'rel oracle-ld',float(abs(nllo-nll)/abs(nll)))
np.save('/tmp/ld_fitc_nll_%d.npy'%n,np.array([float(nll),float(nll-LD(float(nll)))]))

import sys, numpy as np, scipy.linalg as sla
sys.path.insert(0,'tests/golden'); sys.path.insert(0,'.')
import scenarios
from scipy.linalg import lapack
from adaptive_matrix_solver_amd import Context
c=Context(0)
for n in (5,17):
    A=scenarios.hermitian(n,1000+n)
    c.set_matrix(A)
    d,e=c.herm_tridiag()
    cc,dl,el,tau,info=lapack.zhetrd(A,lower=1)
    print(n,"d diff",np.abs(d-dl).max(),"e diff",np.abs(e-el).max(), "abs e diff", np.abs(np.abs(e)-np.abs(el)).max())
    print(" e ", e[:8]); print(" el", el[:8])
    w,Z=sla.eigh_tridiagonal(d,e)
    c.herm_backtransform(Z)
    V=c.get_eigvecs()
    wl,Vl=sla.eigh(A)
    print(" flips", sum(np.linalg.norm(V[:,k]+Vl[:,k])<np.linalg.norm(V[:,k]-Vl[:,k]) for k in range(n)))
    wz,Zl=sla.eigh_tridiagonal(dl,el)
    print(" Z flips (our T vs lapack T)", sum(np.linalg.norm(Z[:,k]+Zl[:,k])<np.linalg.norm(Z[:,k]-Zl[:,k]) for k in range(n)))

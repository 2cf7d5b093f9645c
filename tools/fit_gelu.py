import numpy as np
from scipy.special import erf
def gelu(x): return 0.5*x*(1+erf(x/np.sqrt(2)))
def fit(c,deg,iters=300):
    s=np.linspace(1e-9,c*c*(1-1e-9),40001); u=np.sqrt(s)
    f=0.5*erf(u/np.sqrt(2))/u
    wt=s/np.maximum(1,u)
    # q(s) = 0.5/c + (s-c^2) r(s), r degree deg-1
    tgt=(f-0.5/c)/(s-c*c)
    wt2=wt*np.abs(s-c*c)
    V=np.polynomial.chebyshev.chebvander(2*s/(c*c)-1,deg-1)
    w=np.ones_like(s)
    for it in range(iters):
        coef=np.linalg.lstsq(V*(w*wt2)[:,None],tgt*w*wt2,rcond=None)[0]
        e=np.abs((V@coef-tgt)*wt2); w=w*(0.3+e/e.max()); w/=w.max()
    r=np.polynomial.chebyshev.Chebyshev(coef,domain=[0,c*c]).convert(kind=np.polynomial.Polynomial)
    q=np.polynomial.Polynomial([0.5/c])+np.polynomial.Polynomial([-c*c,1.0])*r
    return q.coef
def evalf32(x,pc,c):
    x=x.astype(np.float32); u=np.clip(x,-np.float32(c),np.float32(c)); s=u*u
    q=np.float32(pc[-1])*np.ones_like(s)
    for a in pc[-2::-1]: q=q*s+np.float32(a)
    p=q*u+np.float32(0.5)
    return x*p
for c,deg in ((4.0,6),(3.875,6),(3.75,6),(4.0,7),(3.5,5),(3.75,5)):
    pc=fit(c,deg)
    x=np.linspace(-12,12,480001)
    g=evalf32(x,pc,c).astype(np.float64)
    err=np.abs(g-gelu(x)); rel=err/np.maximum(1,np.abs(x))
    print(c,deg,"abs",err.max(),"at",x[err.argmax()],"rel",rel.max(), "x<-c:",np.abs(g[x<-c]).max(), "x>c err:",np.abs(g-x)[x>c].max())
    print("  coef",", ".join("%.10ef"%np.float32(a) for a in pc))

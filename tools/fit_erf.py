import numpy as np
from scipy.special import erf
from numpy.polynomial import chebyshev as C, polynomial as P
SPLIT=0.921875
# region 1: erf(x) = x + x*s*p(s) ... fit g(s) = (erf(x)/x - 2/sqrt(pi))... keep simple: erf(x)/x = q(s), deg 6
def fit_cheb(f, lo, hi, deg, n=4000):
    k=np.arange(n); u=np.cos(np.pi*(k+0.5)/n); x=0.5*(hi-lo)*u+0.5*(hi+lo)
    c=C.chebfit(u, f(x), deg)
    # convert to monomial in x
    pu=C.cheb2poly(c)  # poly in u
    # u = (2x-(hi+lo))/(hi-lo)
    a=2/(hi-lo); b=-(hi+lo)/(hi-lo)
    px=np.zeros(1)
    for i,ci in enumerate(pu):
        px=P.polyadd(px, ci*P.polypow([b,a], i))
    return px
# region 1 in s = x^2 over [0, SPLIT^2]
f1=lambda s: np.where(s>0, erf(np.sqrt(s))/np.sqrt(np.maximum(s,1e-300)), 2/np.sqrt(np.pi))
p1=fit_cheb(f1, 0.0, SPLIT**2, 6)
# region 2: Q(t) = -log(1-erf(t)) = -log(erfc(t)), t in [SPLIT, 4.2]
from scipy.special import erfc
f2=lambda t: -np.log(erfc(t))
TMAX=4.0
p2=fit_cheb(f2, SPLIT, TMAX, 9)
def horner32(p, x):
    x=x.astype(np.float32); r=np.full_like(x, np.float32(p[-1]))
    for c in p[-2::-1]:
        r=(r.astype(np.float64)*x.astype(np.float64)+np.float64(np.float32(c))).astype(np.float32)  # fma emulation
    return r
def erf32(x):
    x=x.astype(np.float32); t=np.minimum(np.abs(x), np.float32(TMAX)); s=(x*x).astype(np.float32)
    r1=(horner32(p1, s)*x).astype(np.float32)
    q=horner32(p2, t)
    r2=(np.float32(1)-np.exp(-q.astype(np.float32)).astype(np.float32)).astype(np.float32)
    r2=np.copysign(r2, x)
    return np.where(np.abs(x)<=np.float32(SPLIT), r1, r2)
xs=np.concatenate([np.linspace(-6,6,2000001), np.linspace(-1e-3,1e-3,10001)])
err=np.abs(erf32(xs).astype(np.float64)-erf(xs.astype(np.float32).astype(np.float64)))
print('max abs err', err.max(), 'at', xs[err.argmax()])
print('p1', [float(np.float32(c)) for c in p1])
print('p2', [float(np.float32(c)) for c in p2])
# gelu error
g=lambda x: 0.5*x*(1+erf(x/np.sqrt(2)))
x32=xs.astype(np.float32)
g32=(np.float32(0.5)*x32*(np.float32(1)+erf32((x32*np.float32(0.7071067811865476)).astype(np.float32)))).astype(np.float32)
gerr=np.abs(g32.astype(np.float64)-g(x32.astype(np.float64)))
print('gelu max abs err', gerr.max(), 'at', xs[gerr.argmax()], ' rel(max over |g|>1e-3):', (gerr/np.maximum(np.abs(g(x32.astype(np.float64))),1e-3)).max())

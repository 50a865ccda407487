"""numpy emulation of the guarded contraction form of the pairwise distances: error of d and of softmax(-scale d) against float64
for the guarded form, the direct form and the unguarded contraction, over data scales / offsets with injected near-duplicates and ties."""
import numpy as np
rng=np.random.default_rng(0)
def run(B1,B2,C,scale,offset=0.0,sigma=1.0,dups=True):
    a=(rng.standard_normal((B1,C))*sigma+offset).astype(np.float32)
    b=(rng.standard_normal((B2,C))*sigma+offset).astype(np.float32)
    if dups:
        for j,eps in enumerate([0,1e-3,1e-2,0.1,0.3,1.0,2.0,5.0]):
            b[j]=a[17*j+3]+(rng.standard_normal(C)*eps*sigma/np.sqrt(C)*np.sqrt(C)).astype(np.float32)*0+ (rng.standard_normal(C).astype(np.float32)*np.float32(eps*sigma/np.sqrt(C)))
            # a tie: two rows equally near
            a[17*j+4]=b[j]+ (a[17*j+3]-b[j])[::-1]
    # contraction in fp32 with sequential accumulation (emulates chain; mult+add two roundings)
    dot=np.zeros((B1,B2),np.float32)
    for k in range(C):
        dot+= a[:,k:k+1]*b[None,:,k]
    na=np.zeros(B1,np.float32); nb=np.zeros(B2,np.float32)
    # 16 partial sums
    pa=(a*a).reshape(B1,16,C//16).astype(np.float32)
    na=pa.sum(2,dtype=np.float32).sum(1,dtype=np.float32)
    nb=(b*b).reshape(B2,16,C//16).sum(2,dtype=np.float32).sum(1,dtype=np.float32)
    nsum=na[:,None]+nb[None,:]
    d2=(nsum-np.float32(2)*dot).astype(np.float32)
    flag=~(d2>=np.float32(0.25)*nsum)
    # exact
    d2x=((a.astype(np.float64)[:,None,:]-b.astype(np.float64)[None,:,:])**2).sum(2) if B1*B2*C<3e8 else None
    d2d=np.zeros((B1,B2),np.float32)
    for k in range(C):
        t=a[:,k:k+1]-b[None,:,k]
        d2d+=t*t
    d2g=np.where(flag,d2d,d2)
    res={}
    for name,dd in (("guarded",d2g),("direct",d2d),("contraction",np.maximum(d2,0))):
        d=np.sqrt(dd.astype(np.float64)); dx=np.sqrt(d2x)
        x=-scale*d; xx=-scale*dx
        s=np.exp(x-x.max(0)); s/=s.sum(0)
        sx=np.exp(xx-xx.max(0)); sx/=sx.sum(0)
        res[name]=(np.abs(d-dx).max()/max(dx.max(),1), np.abs(s-sx).max())
    return flag.mean(), res
for kw in [dict(offset=0,sigma=1),dict(offset=0,sigma=0.05),dict(offset=0,sigma=30),dict(offset=3.0,sigma=1),dict(offset=0.5,sigma=1)]:
    for scale in (0.9,5.0):
        f,r=run(2048,64,384,scale,**kw)
        print(kw,scale,"flag frac %.4f"%f, {k:("%.2e"%v[0],"%.2e"%v[1]) for k,v in r.items()})

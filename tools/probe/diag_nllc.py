"""NumPy model of the fuzz sweep's failing N << C case (C = 96, N = 4, scale 10): cyclic Jacobi in fp32 with V in fp32 or rounded to 22 bits, first-order
completion of the spectral functions on (a) the TRACKED rotated matrix, (b) the rotated matrix recomputed from V and the untouched covariance.  Round 5: the
experiment behind refresh_needed (csrc/wct.hip).  Result: profiles/r05_parity_holes.txt."""
import numpy as np, sys
import os
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import oracle
from conftest import rel_err
def features(rng, n, c, scale, mix=True):
    g = rng.standard_normal((n, c))
    if mix: g = g @ (rng.standard_normal((c, c)) / np.sqrt(c))
    g = np.maximum(g, 0) * 10.0 ** rng.uniform(-0.7, 0.7, c)
    return np.float32(g * scale)
def case(c,hc,wc,hs,ws,log_scale,seed):
    rng=np.random.default_rng(seed); scale=10.0**log_scale
    return features(rng,hc*wc,c,scale), features(rng,hs*ws,c,scale*10.0**rng.uniform(-1,1))
f32=np.float32
def round_bits(x, bits):
    # round fp32 array to `bits` significand bits (emulates the split-fp16 product precision), abs floor 2^-24
    m, e = np.frexp(x.astype(np.float64))
    q = np.ldexp(np.round(m * 2.0**bits) / 2.0**bits, e)
    q = np.round(q * 2.0**24) / 2.0**24
    return q.astype(np.float32)
def jacobi(A0, vbits=None, sweeps_max=16, tol=1e-2):
    C=A0.shape[0]; A=A0.astype(f32).copy(); V=np.eye(C,dtype=f32)
    for sw in range(sweeps_max):
        offmax=0.0
        for p in range(C-1):
            for q in range(p+1,C):
                app,aqq,apq=A[p,p],A[q,q],A[p,q]
                den=abs(app*aqq); 
                if abs(apq) < 1e-36 or apq*apq <= 1e-12*den: continue
                big=max(abs(app),abs(aqq))
                rel=min(abs(apq)/np.sqrt(den),1.0) if den>0 else 1.0
                floor_m=1e-4*np.abs(np.diag(A)).max()
                if min(abs(app),abs(aqq))>floor_m: offmax=max(offmax,rel)
                elif big>floor_m: offmax=max(offmax,min(rel,abs(apq)/big))
                tau=f32(0.5)*(aqq-app); h=np.sqrt(tau*tau+apq*apq)
                t=abs(apq)/(abs(tau)+h); t = t if (tau>=0)==(apq>=0) else -t
                c=f32(1/np.sqrt(1+t*t)); s=f32(c*t)
                # rows/cols p,q
                Ap=A[:,p].copy(); Aq=A[:,q].copy()
                A[:,p]=c*Ap-s*Aq; A[:,q]=s*Ap+c*Aq
                Ap=A[p,:].copy(); Aq=A[q,:].copy()
                A[p,:]=c*Ap-s*Aq; A[q,:]=s*Ap+c*Aq
                Vp=V[:,p].copy(); Vq=V[:,q].copy()
                V[:,p]=c*Vp-s*Vq; V[:,q]=s*Vp+c*Vq
                if vbits: V[:,p]=round_bits(V[:,p],vbits); V[:,q]=round_bits(V[:,q],vbits)
        if offmax<tol and sw>=1: break
    return A,V,sw+1
def spectral(A,kind,shift,first_order=True):
    d=np.diag(A).astype(np.float64); C=len(d); E=A.astype(np.float64)-np.diag(d)
    kept=d>1e-5
    f=np.where(kept,np.where(kept,d+shift,1.0)**(-0.5 if kind==0 else 0.5),0.0)
    G=np.diag(f)
    if first_order:
        sa=np.sqrt(np.where(kept,d+shift,1.0))
        for p in range(C):
            for q in range(C):
                if p==q: continue
                if kept[p] and kept[q]:
                    G[p,q]= -E[p,q]/(sa[p]*sa[q]*(sa[p]+sa[q])) if kind==0 else E[p,q]/(sa[p]+sa[q])
                elif kept[p]!=kept[q]:
                    k=p if kept[p] else q; dd=q if kept[p] else p
                    G[p,q]=E[p,q]*f[k]/max(d[k]-d[dd],2*abs(E[p,q]))
    return G
def transform(fc,fs,alpha,Ac,Vc,As,Vs,shift=1e-5):
    Gc=spectral(Ac,0,shift); Gs=spectral(As,1,shift)
    Tw=Vc.astype(np.float64)@Gc@Vc.astype(np.float64).T; Tcs=Vs.astype(np.float64)@Gs@Vs.astype(np.float64).T
    x=(fc-fc.mean(0)).astype(np.float64)
    out=alpha*((Tcs@Tw@x.T).T+fs.mean(0))+(1-alpha)*x
    return out
def run(verbose=False):
    """-> {'reference': r, (name, what): distance}: distances to the nearest exact (float64) outcome of the kept-count band"""
    c,alpha=96,1.0
    fc,fs=case(96,2,2,2,2,1.0,0)
    shaped=(fc.reshape(1,2,2,c),fs.reshape(1,2,2,c))
    exact=[np.asarray(oracle.wct_np(np.float64(shaped[0]),np.float64(shaped[1]),alpha,keep=(kc,3))).reshape(4,c) for kc in range(1,40)]
    def best(o): return min(rel_err(o,e) for e in exact)
    out={'reference': best(np.asarray(oracle.wct_np(*shaped,alpha)).reshape(4,c))}
    covc=np.cov(fc.T.astype(np.float64)).astype(f32); covs=np.cov(fs.T.astype(np.float64)).astype(f32)
    for vb,name,tl in ((None,"V fp32",1e-4),(22,"V 22-bit",1e-4),(22,"V 22-bit tol 1e-6",1e-6)):
        Ac,Vc,sc=jacobi(covc,vb,16,tl); As,Vs,ss=jacobi(covs,vb,16,tl)
        out[(name,'tracked')]=best(transform(fc,fs,alpha,Ac,Vc,As,Vs))
        Vc64,Vs64=Vc.astype(np.float64),Vs.astype(np.float64)
        out[(name,'refreshed')]=best(transform(fc,fs,alpha,Vc64.T@covc.astype(np.float64)@Vc64,Vc,Vs64.T@covs.astype(np.float64)@Vs64,Vs))
        out[(name,'no completion')]=best(transform(fc,fs,alpha,np.diag(np.diag(Ac)),Vc,np.diag(np.diag(As)),Vs))
        out[(name,'sweeps')]=(sc,ss); out[(name,'V orth err')]=np.abs(Vc64.T@Vc64-np.eye(c)).max()
    if verbose:
        for k,v in out.items(): print(k,v)
    return out
if __name__=='__main__':
    run(True)

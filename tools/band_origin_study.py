"""Which origin and which length scale rho should the exact mode's rounding band use?  (round 4; CPU simulation, numpy)

The band of a (pixel, hypothesis) test is |m| <= kband |d| |u|, and the scoring kernel needs it as a PRODUCT of a per-hypothesis and
a per-pixel factor: |d| <= (R + rho)(1 + r / rho) with R = |h - o|, r = |c - o| for an origin o and any rho > 0.  The looser that
bound, the more tests (and cells of 16) are re-evaluated.  This script scores benchmark images in float64 and counts the tests
inside the band for the image's median pixel (rounds 1-3) and for an estimate of the key-point (component-wise median of eight
fixed-pair intersections -- what the hypothesis kernel computes since round 4) as o, over a range of rho.
    python tools/band_origin_study.py        (the simulation of the flagged fractions the MI355X then measured: profiles/r04_band_origin_study.txt)"""
import numpy as np, sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pvnet_amd import synth
rng=np.random.default_rng(0)
def kband(t):
    u=2.0**-24; tau=np.sqrt(1-t*t)/t; t0=np.arccos(t); d=10*u
    lo=np.arccos(min(1,t+d)); hi=np.arccos(t-d)
    return max(np.sin(t0-lo),np.sin(hi-t0))/t*1.001 + u*(1+tau)*(14.3+8)
for thresh in (0.99,0.999):
    kb=kband(thresh); t0=np.arccos(thresh)
    acc={}
    for img in range(4):
        mask,planar,kpts=synth.make_image(img,noise=True,background="normal")
        ys,xs=np.nonzero(mask); tn=len(xs)
        c=np.stack([xs,ys],1).astype(np.float64)
        omed=c[tn//2]
        for k in range(9):
            u=np.stack([planar[2*k][ys,xs],planar[2*k+1][ys,xs]],1).astype(np.float64)
            idx=rng.integers(0,tn,(512,2))
            n0=np.stack([u[idx[:,0],1],-u[idx[:,0],0]],1); n1=np.stack([u[idx[:,1],1],-u[idx[:,1],0]],1)
            det=n0[:,0]*n1[:,1]-n0[:,1]*n1[:,0]
            b0,b1=(n0*c[idx[:,0]]).sum(1),(n1*c[idx[:,1]]).sum(1)
            ok=np.abs(det)>1e-6; det=np.where(ok,det,1.0)
            H=np.stack([(b0*n1[:,1]-b1*n0[:,1])/det,(n0[:,0]*b1-n1[:,0]*b0)/det],1)*ok[:,None]
            d=H[:,None,:]-c[None,:,:]; nd=np.linalg.norm(d,axis=2)
            cos=(d*u[None]).sum(2)/(nd*np.linalg.norm(u,axis=1)[None]+1e-30)
            th=np.arccos(np.clip(cos,-1,1))
            marg=np.abs(np.sin(t0-th))/thresh   # |m|/(|d||u|)
            # origin estimate: component-wise median of 8 candidate hypotheses from fixed pairs
            ca=(np.arange(8)*tn//8+tn//16); cb=(ca+tn//2)%tn
            n0=np.stack([u[ca,1],-u[ca,0]],1); n1=np.stack([u[cb,1],-u[cb,0]],1)
            det2=n0[:,0]*n1[:,1]-n0[:,1]*n1[:,0]; okc=np.abs(det2)>1e-6; det2=np.where(okc,det2,1)
            bb0,bb1=(n0*c[ca]).sum(1),(n1*c[cb]).sum(1)
            C=np.stack([(bb0*n1[:,1]-bb1*n0[:,1])/det2,(n0[:,0]*bb1-n1[:,0]*bb0)/det2],1)[okc]
            okp=np.round(np.median(C,axis=0))
            for name,o,rho in [("median pixel, rho=disk",omed,max(8,np.sqrt(tn/np.pi)))]+[("median pixel, rho=%d"%r,omed,r) for r in (16,24,32,56,80,120)]+[("kp estimate, rho=%d"%r,okp,r) for r in (8,12,16,24,32,40)]+[("true kp, rho=16",np.round(kpts[k]),16)]:
                R=np.linalg.norm(H-o,axis=1); r=np.linalg.norm(c-o,axis=1)
                bound=(R[:,None]+rho)*(1+r[None,:]/rho)
                flagged=marg < kb*bound/(0.9*np.maximum(nd,1e-9))
                acc.setdefault(name,[]).append(flagged.mean())
    print("thresh",thresh)
    for k,v in acc.items(): 
        p=np.mean(v); print("  %-28s test-level band fraction %.3e   cell-level ~%.3e"%(k,p,1-(1-p)**16))

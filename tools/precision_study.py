import numpy as np, sys
sys.path.insert(0,'/root/repo')
from oracle import ransac_voting_oracle as O
from pvnet_amd import synth
f32=np.float32; f64=np.float64
def fma(a,b,c): return (a.astype(f64)*b.astype(f64)+c.astype(f64)).astype(f32)
mask, planar, kpts = synth.make_batch(2, first_index=0, radius=40, noise=True, background="normal")
v = synth.planar_to_vertex_view(planar)
thresh=f32(0.99); tau=f32(np.sqrt(1-float(thresh)**2)/float(thresh)); S=f32(2.0**90)
tot=0; res={}
for bi in range(2):
    coords,direct=O.compact(O.foreground(mask[bi]), v[bi]); tn=len(coords)
    idxs=O.draw_idxs(1,bi,1024,9,tn)
    hyp=O.generate_hypothesis(direct,coords,idxs,np.float32)
    ox,oy = coords[:,0].mean().round(), coords[:,1].mean().round()
    for k in range(9):
        u=direct[:,k].astype(f32); h=hyp[:,k].astype(f32)   # [tn,2],[hn,2]
        cx=coords[:,0][None,:]; cy=coords[:,1][None,:]; hx=h[:,0][:,None]; hy=h[:,1][:,None]
        # float64 truth of the reference predicate
        dx=hx.astype(f64)-cx; dy=hy.astype(f64)-cy
        n1=np.sqrt(u[:,0].astype(f64)**2+u[:,1].astype(f64)**2)[None,:]; n2=np.sqrt(dx*dx+dy*dy)
        ang=(dx*u[:,0][None,:]+dy*u[:,1][None,:])/(n1*n2)
        truth=(ang>f64(thresh))&(n1>=1e-6)&(n2>=1e-6)
        # literal fp32
        lit=O._inlier_block(direct[:,k],coords,h,thresh,np.float32)
        # current 7-op
        Mx=(u[:,0]*S)[None,:]; My=(u[:,1]*S)[None,:]; Tx=(tau*Mx).astype(f32); Ty=(tau*My).astype(f32)
        dxf=(hx-cx).astype(f32); dyf=(hy-cy).astype(f32)
        cr=fma(dyf,-Mx,(dxf*My).astype(f32)); e=fma(dxf,Tx,-np.abs(cr)); s=fma(dyf,Ty,e); v7=s>0
        # 6-op e-form, global coords and local origin
        def eform(ox,oy):
            hxl=(hx-f32(ox)).astype(f32); hyl=(hy-f32(oy)).astype(f32); cxl=(cx-f32(ox)).astype(f32); cyl=(cy-f32(oy)).astype(f32)
            Ec=fma(cyl,-Mx,(cxl*My).astype(f32))      # per pixel
            Ed=fma(cyl,Ty,(cxl*Tx).astype(f32))
            cr=fma(hxl,My,fma(hyl,-Mx,-Ec)); t=(-Ed-np.abs(cr)).astype(f32); s=fma(hxl,Tx,fma(hyl,Ty,t)); return s>0
        g=eform(0,0); l=eform(ox,oy)
        def mfmaform(ox,oy):   # D = fma(a1,b1, fma(a0,b0,C)) per v_mfma_f32_32x32x2_f32, then s = D_dt - |D_cr|
            hxl=(hx-f32(ox)).astype(f32); hyl=(hy-f32(oy)).astype(f32); cxl=(cx-f32(ox)).astype(f32); cyl=(cy-f32(oy)).astype(f32)
            Ec=fma(cyl,-Mx,(cxl*My).astype(f32)); Ed=fma(cyl,Ty,(cxl*Tx).astype(f32))
            Dcr=fma(hyl,-Mx,fma(hxl,My,-Ec)); Ddt=fma(hyl,Ty,fma(hxl,Tx,-Ed))
            return (Ddt-np.abs(Dcr)).astype(f32)>0
        mm=mfmaform(ox,oy)
        for name,x in (("literal32",lit),("op7",v7),("e6_global",g),("e6_local",l),("mfma_local",mm)):
            d=(x!=truth); r=res.setdefault(name,[0,0,0]); r[0]+=d.sum(); r[1]=max(r[1],np.abs(x.sum(1)-truth.sum(1)).max()); r[2]+= (x.sum(1)!=truth.sum(1)).sum()
        tot+=truth.size
for n,(a,b,c) in res.items(): print(f"{n:10s} flipped pairs {a:8d} of {tot} ({a/tot:.2e}), max |count diff| {b}, hyps with count diff {c} of {2*9*1024}")

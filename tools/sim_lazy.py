#!/usr/bin/env python3
"""Development simulation (NOT product, NOT oracle): full slack passes, sweeps and iterations of the stage-wise dual active set on
config-5 problems with the global most-violated rule against "cached rows first" (lazy slacks, mpcqp_stagew.hip). usage: sim_lazy.py [count]"""
import sys; sys.path.insert(0,'.')
import numpy as np
from oracle import stagewise_np as S
from qpmpc_amd import workloads as W

def solve(sp, R=8, prefer_cached=False, tol=1e-9):
    N,nx,nu,mk=sp.N,sp.nx,sp.nu,sp.mk; m=N*mk; n=N*nu
    ric=S.Riccati(sp)
    qlin=np.zeros((N,nx)); pN=np.zeros(nx)
    U0,X0=ric.solve(qlin,np.zeros((N,nu)),x0=sp.x0,pN=pN)
    s=(sp.e-np.einsum("kri,ki->kr",sp.C,X0[:N])-np.einsum("kri,ki->kr",sp.D,U0)).reshape(-1)
    e=sp.e.reshape(-1); tolh=tol*(1+abs(e)); invn=S._row_inv_norms(sp).reshape(-1)
    act=[]; H=[]; lam=[]; Wm=np.zeros((0,0)); cache={}
    iters=sweeps=passes=drops=0
    def hvec(row):
        k,r=divmod(row,mk); ql=np.zeros((N,nx)); rl=np.zeros((N,nu)); ql[k]=-sp.C[k,r]; rl[k]=-sp.D[k,r]
        Vp,Xp=ric.solve(ql,rl)
        return (np.einsum("kri,ki->kr",sp.C,Xp[:N])+np.einsum("kri,ki->kr",sp.D,Vp)).reshape(-1)
    while True:
        viol=(s<-tolh); viol[act]=False
        if not viol.any(): break
        score=np.where(viol,s*invn,np.inf)
        p=int(np.argmin(score))
        if prefer_cached:
            c=[r for r in cache if viol[r]]
            if c: p=min(c,key=lambda r:score[r])
            else: passes+=1
        else: passes+=1
        if p not in cache:
            sweeps+=1
            order=np.argsort(score)[:R]; cache={}
            for r in [p]+[int(o) for o in order if np.isfinite(score[o]) and o!=p][:R-1]: cache[r]=hvec(r)
        hp=cache.pop(p)
        up=0.0
        while True:
            iters+=1
            c=np.array([hp[a] for a in act]); dpp=hp[p]
            r_=Wm@c if act else np.zeros(0); d2=dpp-c@r_ if act else dpp
            t2=-s[p]/d2 if (d2>1e-13*dpp and len(act)<n) else np.inf
            t1=np.inf; l=-1
            for a in range(len(act)):
                if r_[a]>0 and lam[a]/r_[a]<t1: t1=lam[a]/r_[a]; l=a
            t=min(t1,t2)
            if not np.isfinite(t): return None,iters,sweeps,passes,drops
            gz=hp.copy()
            for a in range(len(act)): gz-=r_[a]*H[a]
            s=s+t*gz
            for a in range(len(act)): lam[a]=max(lam[a]-t*r_[a],0.0)
            s[act]=0; up+=t
            if t2<=t1:
                q=len(act); Wn=np.zeros((q+1,q+1)); Wn[:q,:q]=Wm+np.outer(r_,r_)/d2; Wn[:q,q]=-r_/d2; Wn[q,:q]=-r_/d2; Wn[q,q]=1/d2; Wm=Wn
                act.append(p); H.append(hp); lam.append(up); s[p]=0; break
            drops+=1
            wl=Wm[:,l].copy(); Wm=Wm-np.outer(wl,wl)/Wm[l,l]; keep=[a for a in range(len(act)) if a!=l]; Wm=Wm[np.ix_(keep,keep)]
            for lst in (act,H,lam): del lst[l]
    return act,iters,sweeps,passes,drops

w=W.synthetic_ltv_batch(int(sys.argv[1]))
B=w["x0"].shape[0]; N=64
rows=[]
for b in range(B):
    C=np.broadcast_to(w["C"],(N,16,12)); D=np.broadcast_to(w["D"],(N,16,4)); e=np.broadcast_to(w["e"],(N,16))
    sp=S.StageProblem(w["A"][b],w["B"][b],C,D,e,w["x0"][b],np.zeros(12),np.zeros((N,12)),w["wt"],w["wx"],w["wu"])
    a0=solve(sp,prefer_cached=False); a1=solve(sp,prefer_cached=True)
    rows.append((a0[1],a0[2],a0[3],a0[4],a1[1],a1[2],a1[3],a1[4], set(a0[0])==set(a1[0])))
r=np.array(rows,float)
print("global rule : iters %.2f (max %d) sweeps %.2f full passes %.2f drops %.2f"%(r[:,0].mean(),r[:,0].max(),r[:,1].mean(),r[:,2].mean(),r[:,3].mean()))
print("cached first: iters %.2f (max %d) sweeps %.2f full passes %.2f drops %.2f  same active set %.2f"%(r[:,4].mean(),r[:,4].max(),r[:,5].mean(),r[:,6].mean(),r[:,7].mean(),r[:,8].mean()))
k=int(np.argmax(r[:,0])); print("longest problem: global iters %d sweeps %d | cached iters %d sweeps %d passes %d"%(r[k,0],r[k,1],r[k,4],r[k,5],r[k,6]))

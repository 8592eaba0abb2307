#!/usr/bin/env python3
"""Developer probe (CPU, numpy): can AFF3CT's Decoder_LDPC_BP_flooding_Gallager_E be restated from the literature?
Richardson-Urbanke algorithm E (messages in {-1, 0, +1}, check = product of the others, variable = sign(w y + sum of the others)) on
the config-2 code, six variants of weight / tie / erasure handling, against the reference's published rows
(BS/data_dvb/data3 (DVB S2)/DVB_S2_N_64800_K_51840_CR_0.8.txt:5-8: 0/30, 0/30, 14/30, 30/30 frame errors at QBER 1.0, 1.5, 2.0, 2.5 %).
Result: every variant fails 10-12 of 12 frames already at 1.5 %, so AFF3CT's variant is something else and cannot be
pinned without its source: not built."""
import sys, numpy as np, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _qldpc_loader
q=_qldpc_loader.load()
code=q.Code.ira(65536,52429,0.125,11,3,7)
var,chk=code.edges()
var=np.asarray(var); chk=np.asarray(chk)
E=len(var); N=code.N; M=code.M; K=52429
# CN-major order
oc=np.lexsort((var,chk)); cv=var[oc]; cc=chk[oc]
cn_ptr=np.searchsorted(cc,np.arange(M+1))
ov=np.lexsort((chk,var))  # VN-major permutation of CN-major positions: edges sorted by var
# position mapping: VN-major index -> CN-major index
vn_of=np.lexsort((cc,cv))   # indices into CN-major arrays sorted by var
vv=cv[vn_of]; vn_ptr=np.searchsorted(vv,np.arange(N+1))
def run(qber,F,variant,n_ite=50,seed=0,w=1):
    rng=np.random.default_rng(seed)
    # all-zero codeword is fine for a symmetric decoder: y=+1, flips on key VNs only (parity VNs are 'confirmed': no flips)
    y=np.ones((F,N),np.int16); flips=rng.random((F,K))<qber; y[:,:K][flips]=-1
    ye=y[:,cv]            # [F,E] in CN-major
    v2c=ye.copy()
    fails=None
    for ite in range(n_ite):
        # CN: product of others
        z=(v2c==0)
        nz=np.add.reduceat(z.astype(np.int16),cn_ptr[:-1],axis=1)
        neg=np.add.reduceat((v2c<0).astype(np.int16),cn_ptr[:-1],axis=1)
        deg=np.diff(cn_ptr)
        nz_e=np.repeat(nz,deg,axis=1)-z
        par_e=(np.repeat(neg,deg,axis=1)-(v2c<0))&1
        c2v=np.where(nz_e>0,0,np.where(par_e==1,-1,1)).astype(np.int16)
        # VN
        cvn=c2v[:,vn_of]
        s=np.add.reduceat(cvn,vn_ptr[:-1],axis=1)      # [F,N]
        tot=s+w*y
        if variant['tie']=='zero': hard=tot<0
        else: hard=(tot<0)|((tot==0)&(y<0))
        # syndrome
        hb=hard[:,cv].astype(np.int16)
        synd=np.add.reduceat(hb,cn_ptr[:-1],axis=1)&1
        ok=~synd.any(axis=1)
        fails=~( ok & ~hard.any(axis=1) )
        if ok.all(): break
        dvn=np.diff(vn_ptr)
        t_e=np.repeat(tot,dvn,axis=1)-cvn
        if variant['zero']=='erase': nv=np.sign(t_e)
        else: nv=np.where(t_e==0,np.repeat(y,dvn,axis=1),np.sign(t_e))
        v2c_new=np.empty_like(v2c); v2c_new[:,vn_of]=nv.astype(np.int16)
        v2c=v2c_new
    return int(fails.sum()), ite+1
for variant in ({'tie':'zero','zero':'erase'},{'tie':'y','zero':'erase'},{'tie':'y','zero':'y'}):
    for w in (1,2):
        res=[]
        for qb in (0.015,0.02,0.025):
            t=time.time(); f,it=run(qb,12,variant,w=w); res.append((qb,f,it,round(time.time()-t,1)))
        print(variant,'w',w,res,flush=True)

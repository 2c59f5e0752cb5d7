#!/usr/bin/env python
"""Which weights did the wave use? Forensics of one corrupted conv tile (round 6, profiles/r06_barrier_war.txt).

Input: the npz tools/r6_pipeline_race.py --diagnose --dump writes for the first layer whose output differs from a synchronous forward --
the differing patch (got), the synchronous values (ref) and the layer's input around it. For every 8-channel group (= one LDS-DMA piece of a
weight strip) and 4-row block (= one wave) that differs, the script fits
    got - ref  =  sum over ONE K step (chunk c, tap t) and ONE range of its 64 input channels of  x * (W[other step] - W[this step])
for other step = this step +- 3, +- 6 (the strip buffer's previous / next occupants) and reports the best three candidates by max residual
(ReLU-clipped outputs excluded). Test infrastructure: imports oracle-side weights only (ctpn_amd.make_synthetic_arena).

    python tools/r6_war_forensic.py profiles/r06_barrier_war_patch.npz
"""
import sys, numpy as np
sys.path.insert(0,'/root/repo')
import ctpn_amd, torch
z=np.load(sys.argv[1])
layer=str(z['layer'])
w=ctpn_amd.arena_views(ctpn_amd.make_synthetic_arena(0))
def bf(x): return torch.from_numpy(np.ascontiguousarray(x)).to(torch.bfloat16).to(torch.float32).numpy()
W=bf(np.asarray(w[layer+'/weights'],np.float32))
got,ref,inp=z['got'],z['ref'],z['inp']
oy,ox,iy,ix=int(z['oy']),int(z['ox']),int(z['iy']),int(z['ix'])
d=np.argwhere(got!=ref)
ys=np.arange(d[:,0].min(),d[:,0].max()+1); xs=np.arange(d[:,1].min(),d[:,1].max()+1)
Ci=W.shape[2]; nch=Ci//64; Co=W.shape[3]
delta=(got-ref)[ys[0]:ys[-1]+1,xs[0]:xs[-1]+1]
def xin(c,ky,kx):
    Y=oy+ys[:,None]+ky-1-iy; X=ox+xs[None,:]+kx-1-ix
    return inp[Y,X][:,:,c*64:(c+1)*64]
steps=[(c,t) for c in range(nch) for t in range(9)]
X=[xin(c,t%3,t//3) for c,t in steps]
def wts(si,cs): c,t=steps[si]; return W[t%3,t//3,c*64:(c+1)*64,cs]
chs=sorted(set(d[:,2])); groups=sorted(set(c//8 for c in chs))
for g in groups:
    cs=slice(g*8,g*8+8)
    for yb in range(0,len(ys),4):
        dd=delta[yb:yb+4,:,cs]
        if not np.any(dd): continue
        best=[]
        for si in range(len(steps)):
            for off in (3,-3,6,-6):
                sj=si+off
                if not (0<=sj<len(steps)): continue
                for ksl in ((48,64),(32,64),(0,16),(0,64),(16,32),(32,48)):
                    k0,k1=ksl
                    xi=X[si][yb:yb+4,:,k0:k1]
                    dw=wts(sj,cs)[k0:k1]-wts(si,cs)[k0:k1]
                    pred=xi@dw
                    # ReLU/rounding: compare only where ref>0 and got>0
                    m=(ref[ys[0]+yb:ys[0]+yb+4,xs[0]:xs[-1]+1,cs]>0)&(got[ys[0]+yb:ys[0]+yb+4,xs[0]:xs[-1]+1,cs]>0)
                    if m.sum()<8: continue
                    res=float(np.abs((dd-pred)[m]).max())
                    best.append((res,steps[si],off,ksl))
        best.sort()
        print('group',g*8,'rows',int(oy+ys[0]+yb),'max|delta| %.3f'%np.abs(dd).max(),[(round(a,3),b,c,e) for a,b,c,e in best[:3]])

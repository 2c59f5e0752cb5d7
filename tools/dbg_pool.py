import numpy as np, sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import ctpn_amd
from ctpn_amd import _binding as B
from oracle import network as N
rng=np.random.default_rng(1)
for prec in ("fp32","bf16"):
  for (n,h,w,ci,co) in [(1,16,64,128,128),(2,35,50,128,128)]:
    x=np.maximum(rng.standard_normal((n,h,w,ci)).astype(np.float32),0)
    wt=(rng.standard_normal((3,3,ci,co))*(2.0/(9*ci))**0.5).astype(np.float32)
    b=(rng.standard_normal(co)*0.1).astype(np.float32)
    full,pool=B.debug_conv3x3(x,wt,b,prec,1,True,True)
    ref=N.maxpool2x2(full)
    bad=np.argwhere(pool!=ref)
    print(prec,(n,h,w,ci,co),"mismatches",len(bad),"of",pool.size)
    if len(bad):
        print(" first:",bad[:6].tolist())
        print(" channels hist:",np.bincount(bad[:,3],minlength=co)[:130].reshape(-1,8).sum(1))
        print(" X hist:",np.bincount(bad[:,2],minlength=w//2), " Y hist:",np.bincount(bad[:,1],minlength=h//2))
        i=tuple(bad[0]); print(" got",pool[i]," want",ref[i])
    full2,pool2=B.debug_conv3x3(x,wt,b,prec,1,True,False)
    print("   pool-only equals pool+full:",np.array_equal(pool2,pool), " pool-only vs ref mism:",int((pool2!=ref).sum()))

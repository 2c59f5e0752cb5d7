#!/usr/bin/env python
"""Long differential run of the host C++ connector (ctpn_text_lines, device_id = -1) against oracle/postproc.py::text_detect -- itself pinned
to the reference's TextDetector on the fixtures -- on random CTPN-shaped proposals at three image sizes, both modes: the hypothesis test of
tests/test_properties.py with a clock instead of an example count. Structural differences (number of lines, scores, coordinates beyond
1e-3) are printed; last-bit differences of the fitted coordinates (numpy's polyfit is LAPACK's SVD least squares in double, the C++ a double
closed form) are only tracked as `worst`.
    python tools/diff_connector.py SEED SECONDS
Round 4: 131 k cases, no structural difference, worst coordinate difference 2.4e-4 (two fp32 ulps at x ~ 1900)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ctpn_amd
from ctpn_amd import _binding as B
from oracle import postproc as P
rng=np.random.default_rng(int(sys.argv[1])); T=float(sys.argv[2]); t0=time.time(); n=0; bad=0; worst=0
while time.time()-t0<T:
    cnt=int(rng.integers(0,400)); W=int(rng.choice([200,900,1920])); H=int(rng.choice([140,600,1280]))
    col=rng.integers(0,W//16,cnt); y1=rng.uniform(0,H-8,cnt).astype(np.float32); h=rng.uniform(4,80,cnt).astype(np.float32)
    boxes=np.stack([16.0*col,y1,16.0*col+15.0,np.minimum(y1+h,H-1)],1).astype(np.float32)
    scores=rng.choice(np.concatenate([np.linspace(0.5,1.0,40),rng.uniform(0.69,0.71,20)]).astype(np.float32),cnt)
    mode=str(rng.choice(["H","O"]))
    want=P.text_detect(boxes.copy(),scores[:,None].copy(),(H,W),mode)
    got=B.text_lines(boxes,scores,(H,W),mode,device_id=-1)
    n+=1
    if got.shape!=want.shape or not np.array_equal(got[:,8],want[:,8]) or not np.allclose(got[:,:8],want[:,:8],rtol=0,atol=1e-3):
        bad+=1; print("DIFF", n, cnt, mode, got.shape, want.shape, (np.abs(got-want).max() if got.shape==want.shape else None))
    elif got.size: worst=max(worst,float(np.abs(got[:,:8]-want[:,:8]).max()))
print("cases",n,"divergent",bad,"worst coord diff",worst)

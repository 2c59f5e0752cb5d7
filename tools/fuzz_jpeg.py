#!/usr/bin/env python
"""Mutation fuzzer of the JPEG host half (marker parser, Huffman tables, sequential and progressive entropy decoders) and, with CRC-breaking
damage, the PNG front end: seeds are small files of every layout Pillow writes; mutants are byte flips, cuts, deletions and insertions. Run
it under the AddressSanitizer build (tools/run_fuzz.sh): an error return is fine, a sanitizer report is a bug.
    python tools/fuzz_jpeg.py SEED SECONDS
Round 4: 6.2 M mutants, one finding (jhuff_build's lookahead fill on a DHT with more codes than its length holds: fixed,
tests/test_jpeg.py::test_a_huffman_table_with_more_codes_than_its_length_holds_is_rejected)."""
import sys, io, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from PIL import Image
import ctpn_amd
from ctpn_amd import _binding as B
from util_jpeg import scene, encode
rng=np.random.default_rng(int(sys.argv[1]) if len(sys.argv)>1 else 0)
seeds=[]
for sub in (0,1,2):
    for prog in (False,True):
        for kw in ({}, {"restart_marker_blocks":2}, {"optimize":True}):
            seeds.append(encode(scene(int(rng.integers(8,60)),int(rng.integers(8,60)),len(seeds)),int(rng.integers(20,100)),sub,progressive=prog,**kw))
seeds.append(encode(scene(33,21,5,gray=True),80,progressive=True))
pngs=[]
for mk in (lambda: Image.fromarray(scene(23,31,1)), lambda: Image.fromarray(scene(23,31,2,gray=True)), lambda: Image.fromarray(scene(20,20,3)).quantize(16),
           lambda: Image.fromarray(np.dstack([scene(17,19,4),scene(17,19,5,gray=True)]))):
    for kw in ({}, {"compress_level":0}):
        b=io.BytesIO(); mk().save(b,"PNG",**kw); pngs.append(b.getvalue())
t0=time.time(); n=0; ok=0
T=float(sys.argv[2]) if len(sys.argv)>2 else 60
while time.time()-t0<T:
    for kind,pool,fn in (("j",seeds,B.jpeg_entropy_decode),("p",pngs,B.png_decode)):
        d=bytearray(pool[int(rng.integers(len(pool)))])
        m=int(rng.integers(0,4))
        if m==0:
            for pos in rng.integers(2,len(d),int(rng.integers(1,6))): d[pos]=int(rng.integers(0,256))
        elif m==1:
            d=d[:int(rng.integers(4,len(d)))]
        elif m==2:
            a=int(rng.integers(2,len(d))); b=int(rng.integers(a,min(len(d),a+40))); d=d[:a]+d[b:]
        else:
            a=int(rng.integers(2,len(d))); d=d[:a]+bytes(rng.integers(0,256,int(rng.integers(1,30)),dtype=np.uint8))+d[a:]
        # PNG: keep CRCs valid half the time so that the damage reaches inflate / the filters
        n+=1
        try:
            fn(bytes(d)); ok+=1
        except B.CtpnError as e:
            assert e.code in (-1,-4,-6), e
print("mutants",n,"decoded",ok)

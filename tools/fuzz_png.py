#!/usr/bin/env python
"""Mutation fuzzer of the PNG decoder BEHIND its checksums: every mutant is re-assembled with valid chunk CRCs and a valid zlib stream (unless
the zlib stream itself is the target), so the damage reaches the header checks, inflate, the row filters, the palette and the Adam7 pass
arithmetic: damaged scanlines and filter bytes, short / long scanline data, IHDR fields, a damaged zlib stream, empty IDAT chunks, short
palettes. Run it under the AddressSanitizer build (tools/run_fuzz.sh).
    python tools/fuzz_png.py SEED SECONDS
Round 4: 1.75 M mutants, no finding."""
import sys, io, os, time, zlib, struct
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from PIL import Image
import ctpn_amd
from ctpn_amd import _binding as B
from util_jpeg import scene
rng=np.random.default_rng(int(sys.argv[1]))
def chunk(k,b): return struct.pack(">I",len(b))+k+b+struct.pack(">I",zlib.crc32(k+b)&0xffffffff)
def parse(d):
    i=8; out=[]
    while i<len(d):
        L=struct.unpack(">I",d[i:i+4])[0]; out.append((d[i+4:i+8],d[i+8:i+8+L])); i+=12+L
    return out
pngs=[]
for mk in (lambda: Image.fromarray(scene(23,31,1)), lambda: Image.fromarray(scene(23,31,2,gray=True)), lambda: Image.fromarray(scene(20,20,3)).quantize(16),
           lambda: Image.fromarray(np.dstack([scene(17,19,4),scene(17,19,5,gray=True)])), lambda: Image.fromarray(scene(9,40,6,gray=True)>100)):
    b=io.BytesIO(); mk().save(b,"PNG"); pngs.append(parse(b.getvalue()))
t0=time.time(); n=0; ok=0; T=float(sys.argv[2])
while time.time()-t0<T:
    ch=[list(c) for c in pngs[int(rng.integers(len(pngs)))]]
    m=int(rng.integers(0,5))
    idat=b"".join(b for k,b in ch if k==b"IDAT")
    raw=bytearray(zlib.decompress(idat))
    ihdr=bytearray(ch[0][1])
    if m==0:   # damage the scanlines (filter bytes included), valid stream
        for pos in rng.integers(0,len(raw),int(rng.integers(1,8))): raw[pos]=int(rng.integers(0,256))
    elif m==1: # shorter / longer scanline data
        raw=raw[:int(rng.integers(0,len(raw)))] if rng.integers(2) else raw+bytes(rng.integers(0,256,int(rng.integers(1,500)),dtype=np.uint8))
    elif m==2: # IHDR fields: size, depth, colour type, interlace
        f=int(rng.integers(0,5))
        if f==0: ihdr[0:4]=struct.pack(">I",int(rng.integers(0,70000)))
        elif f==1: ihdr[4:8]=struct.pack(">I",int(rng.integers(0,70000)))
        elif f==2: ihdr[8]=int(rng.choice([0,1,2,3,4,8,16,7]))
        elif f==3: ihdr[9]=int(rng.choice([0,2,3,4,6,1,5]))
        else: ihdr[12]=int(rng.integers(0,3))
    z=zlib.compress(bytes(raw),1)
    if m==3:   # damaged zlib stream under a valid CRC
        z=bytearray(z)
        for pos in rng.integers(0,len(z),int(rng.integers(1,4))): z[pos]=int(rng.integers(0,256))
        z=bytes(z)
    out=b"\x89PNG\r\n\x1a\n"+chunk(b"IHDR",bytes(ihdr))
    for k,b in ch[1:]:
        if k==b"IDAT": continue
        if k==b"IEND":
            pieces=[z] if m!=4 else [z[:len(z)//3],b"",z[len(z)//3:]]
            for p in pieces: out+=chunk(b"IDAT",p)
        if k==b"PLTE" and rng.integers(4)==0: b=b[:3*int(rng.integers(0,len(b)//3+1))]
        out+=chunk(k,b)
    n+=1
    try:
        B.png_decode(out); ok+=1
    except B.CtpnError as e:
        assert e.code in (-1,-4,-6), e
print("png mutants",n,"decoded",ok)

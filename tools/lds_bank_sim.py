#!/usr/bin/env python
"""LDS bank-conflict simulator for the fragment reads of this library's kernels (CPU only).

Model: MI355X_MICROARCH.md, section "LDS": 64 banks x 4 B; a wave64 access is serviced in fixed lane groups
(ds_read_b128: 4 groups of 16 lanes in the documented interleaved order; ds_read_b64 / ds_read_b64_tr_b16: 2 groups
of 32); within a group every distinct dword address on a busy bank costs one extra LDS cycle.  `cycles()` returns
the LDS-array cycles of one wave instruction (ideal: 4 for b128, 2 for the 8-byte reads).  Validated against
rocprofv3: the swizzled weight-gradient ring (simulated 2 = ideal) measures 0.9 % SQ_LDS_BANK_CONFLICT /
SQ_LDS_IDX_ACTIVE, the 144-byte-pitch flash tiles (simulated 8 and 4 = 2x ideal) measured 31-36 %
(profiles/r02c_pmc_utilisation_b128.txt).  It is what chose KP = 160 (flash), PIX_PITCH = 96 (conv),
PITCH_KC = 160 and the swizzled RC rows (register-staged GEMM)."""
import itertools
G128=[list(range(0,4))+list(range(12,16))+list(range(20,28)), list(range(4,12))+list(range(16,20))+list(range(28,32)),
      list(range(32,36))+list(range(44,48))+list(range(52,60)), list(range(36,44))+list(range(48,52))+list(range(60,64))]
G64=[list(range(0,32)), list(range(32,64))]
def cycles(addrs, nbytes, groups):
    tot=0
    for g in groups:
        banks={}
        for l in g:
            a=addrs[l]
            for w in range(nbytes//4):
                b=((a+4*w)//4)%64
                banks.setdefault(b,set()).add((a+4*w)//4)
        tot+=max(len(v) for v in banks.values())
    return tot
def kc(KP, swz=None):
    res=[]
    for s in range(2):
        ad=[]
        for l in range(64):
            r,g=l&15,l>>4
            ch=s*4+g
            if swz: ch^=swz(r)
            ad.append(r*KP+ch*16)
        res.append(cycles(ad,16,G128))
    return res
def tr(KP, swz=None):
    res=[]
    for s in range(2):
      for cb in range(0,64,16):
        for half in (0,16):
            ad=[]
            for l in range(64):
                r,g=l&15,l>>4
                a,b=r>>2,r&3
                row=32*s+half+4*g+a
                col=cb+4*b
                if swz:
                    ch=(col*2)//16; off=(col*2)%16
                    ch^=swz(row)
                    ad.append(row*KP+ch*16+off)
                else:
                    ad.append(row*KP+col*2)
            res.append(cycles(ad,8,G64))
    return res
print('pitch144 kc', kc(144), 'ideal 4 each'); print('pitch144 tr', tr(144), 'ideal 2 each')
rr=lambda k: ((k&3)|(((k>>3)&1)<<2))<<1
print('rr_swz kc', kc(128,rr)); print('rr_swz tr', tr(128,rr))
s7=lambda k: k&7
print('row&7 kc', kc(128,s7)); print('row&7 tr', tr(128,s7))
for name,f in [('(k&3)<<1|(k>>2&1)', lambda k: ((k&3)<<1)|((k>>2)&1)), ('k&7 ^ (k>>3&1)', lambda k:(k&7)^((k>>3)&1)), ('((k>>1)&7)', lambda k:(k>>1)&7), ('(k&3)*2 + ((k>>3)&1)*... alt', lambda k: ((k&3)<<1) ^ (((k>>2)&3)))]:
    print(name,'kc',kc(128,f),'tr',tr(128,f))
for KP in (136,152,160,272):
    print('pitch',KP,'kc',kc(KP),'tr',tr(KP))
print('--- conv: addr=(p0+r)*PP+g*16, b128')
for PP in (64,80,96,112,144,160,208):
    ad=[( (l&15))*PP+(l>>4)*16 for l in range(64)]
    print('PP',PP,cycles(ad,16,G128))
print('--- reg-staged GEMM RC: addr=(s*32+g*8+a (+4))*P + (rbase+b*4)*2, tr64; KC: (rbase+r)*P+(s*4+g)*16 b128')
for P in (272,288,304,320,256+64):
    res=[]
    for s in range(2):
        for rb in (0,16,32,48,64,80,96,112):
            for hi in (0,4):
                ad=[]
                for l in range(64):
                    r,g=l&15,l>>4; a,b=r>>2,r&3
                    ad.append((s*32+g*8+a+hi)*P+(rb+b*4)*2)
                res.append(cycles(ad,8,G64))
    print('RC pitch',P,set(res))
for P in (144,160):
    res=[]
    for s in range(2):
        ad=[((l&15))*P+(s*4+(l>>4))*16 for l in range(64)]
        res.append(cycles(ad,16,G128))
    print('KC pitch',P,res)
print('--- RC with rr_swz, pitch 256 (the rr_ring kernel layout)')
res=[]
for s in range(2):
    for rb in range(0,128,16):
        for hi in (0,4):
            ad=[]
            for l in range(64):
                r,g=l&15,l>>4; a,b=r>>2,r&3
                k0=s*32+g*8+a+hi
                col=rb+b*4
                off=(((col>>3)^rr(k0))<<4)+((col&7)<<1)
                ad.append(k0*256+off)
            res.append(cycles(ad,8,G64))
print(set(res))
print('--- conv wgrad tr reads: pixel k=8g+a4 (+4): row=(k>>4), col=k&15; addr = (row*W + col)*PP + (cf*16+b4*4)*2 ; W=TF(16) or HF(18)')
for PP in (80,96,112,128,144,160):
    out={}
    for W in (16,18):
        res=[]
        for hi in (0,4):
            for dfo in (0,1,2):
                ad=[]
                for l in range(64):
                    r,g=l&15,l>>4; a4,b4=r>>2,r&3
                    k=8*g+a4
                    ad.append(((k>>4)*W+(k&15)+dfo)*PP + hi*PP + (b4*4)*2)
                res.append(cycles(ad,8,G64))
        out[W]=set(res)
    print('PP',PP,out)

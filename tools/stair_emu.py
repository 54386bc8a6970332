#!/usr/bin/env python3
"""CPU emulation of the staircase bridges (td_spec_kernel.inc, bwd_stair) in float32 with the reference's own logsum table:
one barcode HMM of the config-3 architecture, bridged from unknown rows `W` positions above the hand-over, for windows of 1, 2,
3, 4 and all columns in interval arithmetic.  Every enclosure is asserted against the dense backward sweep's floats at every
step (a column may only close on the reference's value), and the number of positions until all columns are exact is printed:
the measurement behind TDS_STAIR_WN = 2.  Slow (pure Python); usage: tools/stair_emu.py [reads]"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench          # noqa: E402

f32 = np.float32
NEG = f32(-np.inf)
# init_logsum(), src/misc.c:57-63
T = np.log(1.0 + np.exp(-np.arange(16000, dtype=np.float64) / np.float64(np.float32(1000.0)))).astype(np.float32)
def lsum(a,b):
    a=f32(a); b=f32(b)
    mx=max(a,b); mn=min(a,b)
    if mn==NEG or f32(mx-mn)>=f32(15.7): return mx
    return f32(mx+T[int(f32(f32(mx-mn)*f32(1000.0)))])
def lterm(d):
    d=f32(d)
    if not (d<f32(15.7)): return f32(0.0)   # inf / >= 15.7
    if d!=d: return T[0]
    return T[int(f32(d*f32(1000.0)))]
class SI:
    __slots__=('lo','hi')
    def __init__(s,lo,hi=None): s.lo=f32(lo); s.hi=f32(lo if hi is None else hi)
    def add(s,c): return SI(f32(s.lo+f32(c)), f32(s.hi+f32(c)))
    def closed(s): return s.lo==s.hi or (s.lo==NEG and s.hi==NEG)
def silsum(a,b):
    if a.lo==NEG and a.hi==NEG: return b
    if b.lo==NEG and b.hi==NEG: return a
    with np.errstate(invalid='ignore'):
        dmin=max(f32(a.lo-b.hi), f32(b.lo-a.hi), f32(0.0))
        dmax=max(f32(a.hi-b.lo), f32(b.hi-a.lo))
    return SI(f32(max(a.lo,b.lo)+lterm(dmax)), f32(max(a.hi,b.hi)+lterm(dmin)))

def load(wl="c3"):
    bench.select_workload(wl); m=bench.load_model()
    S=int(m['S']); nh=m['n_hmm'].astype(int); nc=m['n_col'].astype(int)
    co=np.concatenate([[0],np.cumsum(nh*nc)])
    return dict(S=S,nh=nh,nc=nc,co=co,t=m['trans'].reshape(-1,9).astype(f32),eM=m['eM'].reshape(-1,5).astype(f32),eI=m['eI'].reshape(-1,5).astype(f32),
                sM=m['sM'].astype(f32),sI=m['sI'].astype(f32),skip=m['skip'].astype(f32))

def backward(md,x):
    """full backward, returns SB rows [S+1][L+2] and per (j,f) matrices MB,IB [NC][L+2]"""
    L=len(x); S=md['S']
    xx=np.concatenate([[0],x,[0]]).astype(int)   # x_1..x_L at 1..L, x_{L+1}=0
    SB=np.full((S+1,L+2),NEG,f32)
    SB[S,L+1]=0.0
    run=f32(0)
    for j in range(S-1,-1,-1):
        run=f32(run+md['skip'][j]); SB[j,L+1]=run
    mats={}
    for j in range(S-1,-1,-1):
        P=SB[j+1]; Cs=SB[j]; NC=md['nc'][j]; K=NC-1
        for f in range(md['nh'][j]):
            q0=md['co'][j]+f*NC
            MB=np.full((NC,L+2),NEG,f32); IB=np.full((NC,L+2),NEG,f32); DB=np.full((NC,L+2),NEG,f32)
            t=md['t']; eM=md['eM']; eI=md['eI']
            for i in range(L,0,-1):
                c=xx[i+1]; xi=xx[i]
                q=q0+K
                MB[K,i]=f32(P[i+1]+t[q,7])
                I=f32(P[i+1]+t[q,8])
                I=lsum(I,f32(f32(MB[K,i+1]+t[q,4])+eM[q,c]))
                I=lsum(I,f32(f32(IB[K,i+1]+t[q,3])+eI[q,c]))
                IB[K,i]=I
                Cs[i]=lsum(Cs[i],f32(f32(MB[K,i]+md['sM'][q])+eM[q,xi]))
                Cs[i]=lsum(Cs[i],f32(f32(IB[K,i]+md['sI'][q])+eI[q,xi]))
                for g in range(K-1,-1,-1):
                    q=q0+g; p=g+1
                    M=f32(f32(MB[p,i+1]+eM[q+1,c])+t[q,0])
                    M=lsum(M,f32(P[i+1]+t[q,7]))
                    M=lsum(M,f32(f32(IB[g,i+1]+eI[q,c])+t[q,1]))
                    M=lsum(M,f32(DB[p,i]+t[q,2]))
                    I=f32(f32(IB[g,i+1]+t[q,3])+eI[q,c])
                    I=lsum(I,f32(P[i+1]+t[q,8]))
                    I=lsum(I,f32(f32(MB[p,i+1]+t[q,4])+eM[q+1,c]))
                    D=f32(DB[p,i]+t[q,5])
                    D=lsum(D,f32(f32(MB[p,i]+eM[q+1,xi])+t[q,6]))
                    MB[g,i]=M; IB[g,i]=I; DB[g,i]=D
                    Cs[i]=lsum(Cs[i],f32(f32(M+md['sM'][q])+eM[q,xi]))
                    Cs[i]=lsum(Cs[i],f32(f32(I+md['sI'][q])+eI[q,xi]))
                Cs[i]=lsum(Cs[i],f32(P[i]+md['skip'][j]))
            mats[(j,f)]=(MB,IB,DB)
    return SB,mats
LN=[f32(0), f32(0), f32(0.69320), f32(1.09870), f32(1.38640)]
def ub(terms):
    live=[t for t in terms if t is not None]
    n=len(live)
    if n==0: return NEG
    mx=max(live)
    if n==1: return mx
    return f32(mx+f32(f32(LN[n]+f32(1e-3*(n-1)))+f32(min(abs(mx),f32(1e30))*f32(4e-7))))
def live(c): return c!=NEG
def stairw(x,j,f,p_hi,p_lo,WN=3,loose=8.0):
    md = load("c3") if not hasattr(stairw, "md") else stairw.md
    stairw.md = md
    SB,mats=backward(md,x); MBt,IBt,DBt=mats[(j,f)]
    xx=np.concatenate([[0],x,[0]]).astype(int)
    P=SB[j+1]; NC=md['nc'][j]; K=NC-1; q0=md['co'][j]+f*NC
    t=md['t']; eM=md['eM']; eI=md['eI']
    h=f32(max(MBt[:,p_hi+1].max(), IBt[:,p_hi+1].max())+loose)
    M=[SI(NEG,h) for _ in range(NC)]; I=[SI(NEG,h) for _ in range(NC)]
    gc=K; log=[]
    for i in range(p_hi,p_lo,-1):
        for _ in range(2):
            if gc>=0 and M[gc].closed() and I[gc].closed():
                assert M[gc].lo==MBt[gc,i+1] and I[gc].lo==IBt[gc,i+1]
                log.append((p_hi-i,gc)); gc-=1
        c=xx[i+1]; xi=xx[i]; Pn=P[i+1]
        Mc=[None]*NC; Ic=[None]*NC
        q=q0+K
        Mk=SI(f32(Pn+t[q,7]))
        Ik=SI(f32(Pn+t[q,8]))
        if live(t[q,4]): Ik=silsum(Ik,M[K].add(t[q,4]).add(eM[q,c]))
        if live(t[q,3]): Ik=silsum(Ik,I[K].add(t[q,3]).add(eI[q,c]))
        Mc[K]=Mk; Ic[K]=Ik
        D=SI(NEG)
        for g in range(K-1,-1,-1):
            q=q0+g; epc=eM[q+1,c]; eic=eI[q,c]
            if g>gc-WN:   # exact or window: SI arithmetic (degenerate for exact)
                Mg=M[g+1].add(epc).add(t[q,0]); Ig=I[g].add(t[q,3]).add(eic)
                Mg=silsum(Mg,SI(f32(Pn+t[q,7]))); Ig=silsum(Ig,SI(f32(Pn+t[q,8])))
                Mg=silsum(Mg,I[g].add(eic).add(t[q,1])); Ig=silsum(Ig,M[g+1].add(t[q,4]).add(epc))
                Mg=silsum(Mg,D.add(t[q,2]))
                Dn=silsum(D.add(t[q,5]), Mc[g+1].add(eM[q+1,xi]).add(t[q,6]))
                Mc[g]=Mg; Ic[g]=Ig; D=Dn
            else:
                tm=[f32(f32(M[g+1].hi+epc)+t[q,0]) if live(t[q,0]) else None, f32(Pn+t[q,7]) if live(t[q,7]) else None,
                    f32(f32(I[g].hi+eic)+t[q,1]) if live(t[q,1]) else None, f32(D.hi+t[q,2]) if live(t[q,2]) else None]
                ti=[f32(f32(I[g].hi+t[q,3])+eic) if live(t[q,3]) else None, f32(Pn+t[q,8]) if live(t[q,8]) else None, f32(f32(M[g+1].hi+t[q,4])+epc) if live(t[q,4]) else None]
                td=[f32(D.hi+t[q,5]) if live(t[q,5]) else None, f32(f32(Mc[g+1].hi+eM[q+1,xi])+t[q,6]) if live(t[q,6]) else None]
                Mc[g]=SI(NEG,ub(tm)); Ic[g]=SI(NEG,ub(ti)); D=SI(NEG,ub(td))
        M=Mc; I=Ic
        for g in range(NC):
            assert M[g].lo<=MBt[g,i]<=M[g].hi and I[g].lo<=IBt[g,i]<=I[g].hi, (g,i)
    for _ in range(2):
        if gc>=0 and M[gc].closed() and I[gc].closed(): log.append((p_hi-p_lo,gc)); gc-=1
    return log,gc

if __name__ == "__main__":
    np.seterr(all="ignore")
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    md = load("c3")
    X = bench.synth_batch(64, 3)
    for WN in (1, 2, 3, 4, 6):
        res = []
        for f in (0, 2, 8):
            steps = []
            for r in range(n_reads):
                log, gc = stairw(X[r], 0, f, 75, 41, WN)
                steps.append(log[-1][0] if gc < 0 else 99)
            res.append("%.1f / %d" % (np.mean(steps), max(steps)))
        print("window %d: positions until every column is exact, mean / max over %d reads, HMMs 0, 2, 8: %s" % (WN, n_reads, "   ".join(res)))

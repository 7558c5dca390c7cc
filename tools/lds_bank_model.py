#!/usr/bin/env python3
"""Bank-conflict model of the pass buffer of p1_run_t (hevc_core.h): LDS cycles of the three matrix-stage reads of one pass for a
tile padding / row swizzle, per TU size, by the serving rules of the MI355X guide (ds_read_b64: 2 x 32 lanes, ds_read_b128: 4 x 16
lanes, 64 banks).  Prints the layout of round 2 ("baseline") and the best padding found.  Runs without a GPU."""
import itertools
def diag_order(n):
    out=[]
    for d in range(2*n-1):
        for y in range(min(d,n-1),-1,-1):
            x=d-y
            if x>=n: continue
            out.append((y,x))
    return out
G128=[[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27],[4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]]
G128=G128+[[l+32 for l in g] for g in G128]
G64=[list(range(32)),list(range(32,64))]
def cost(acc, width):   # acc: list of 64 dword addresses (or None); width in dwords (2: b64, 4: b128)
    groups = G64 if width==2 else G128
    tot=0
    for g in groups:
        banks={}
        for l in g:
            a=acc[l]
            if a is None: continue
            for k in range(width):
                banks.setdefault((a+k)%64,set()).add(a+k)
        tot+=max((len(v) for v in banks.values()), default=0)
    return tot, len(groups)
def lanes(N):
    nb=N//4; lpc=nb*nb; order=diag_order(nb)
    return [(l//lpc, order[l%lpc][0], order[l%lpc][1]) for l in range(64)]
def evaluate(N, padr, padt, padi, swz):
    NN=N*N; L=lanes(N); res={}
    nbm=N//4
    def s_t(by): return ((by % (N//4)) * 4) if swz else 0       # tmp: xor on dword column (multiples of 4)
    def s_i(by): return ((by % (N//4))) if swz else 0           # itmp: xor on 4-i16 unit index
    # mac_MX reads (res, i16): X[(k)*N + bx*4] b64 ; dword addr = (sl*(NN+padr) + k*N + bx*4)/2
    tot=0;base=0
    for k in range(N):
        acc=[(sl*(NN+padr)+k*N+bx*4)//2 for (sl,by,bx) in L]
        c,b=cost(acc,2); tot+=c; base+=b
    res['mac_MX(res)']=(tot,base)
    # mac_YM32 reads tmp i32: Y[(by*4+r)*N + k0] b128, physical col = k0 ^ s_t(by)
    tot=0;base=0
    for r in range(4):
        for k0 in range(0,N,4):
            acc=[sl*(NN+padt)+(by*4+r)*N+(k0^s_t(by)) for (sl,by,bx) in L]
            c,b=cost(acc,4); tot+=c; base+=b
    res['mac_YM32(tmp)']=(tot,base)
    # mac_YM16 reads itmp i16: Y[(by*4+r)*N + k0] b64: unit = k0/4 ^ s_i(by)
    tot=0;base=0
    for r in range(4):
        for k0 in range(0,N,4):
            acc=[(sl*(NN+padi)+(by*4+r)*N+(((k0//4)^s_i(by))*4))//2 for (sl,by,bx) in L]
            c,b=cost(acc,2); tot+=c; base+=b
    res['mac_YM16(itmp)']=(tot,base)
    return res
for N in (8,16,32):
    print("N",N,"baseline",evaluate(N,0,0,0,False))
    best=None
    for padr,padt,padi,swz in itertools.product((0,4,8,16),(0,4,8),(0,4,8,16),(False,True)):
        G=64//((N//4)**2)
        if G*(N*N+padr)*2 + G*(N*N+padt)*4 > 7168 or G*(N*N+padi)*2 > G*(N*N+padt)*4: continue
        r=evaluate(N,padr,padt,padi,swz); t=sum(v[0] for v in r.values())
        if best is None or t<best[0]: best=(t,(padr,padt,padi,swz),r)
    print("   best",best)

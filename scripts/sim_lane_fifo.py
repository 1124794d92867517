#!/usr/bin/env python
"""Design study (CPU only, no kernel code): would per-lane hit FIFOs in LDS beat the wave-wide compaction of k_rdf_pencil?
A per-lane FIFO needs 1 VALU per candidate column for the push (pointer bump under EXEC) instead of 3 (v_mbcnt x2 + address),
but its pop runs ragged rows.  This simulates one c2 work item (64 i atoms, half shell, x-sorted j streams) and counts VALU
instructions for both schemes.  Result: j atoms arrive sorted along x, so at any moment only the lanes near the current x
receive hits; even 32 rows per lane (8 KB of LDS per wave) leave the pop at 46 % lane utilisation and the total VALU count
no better than compaction (ratio 1.00; 1.22 at 8 rows).  Rejected - see DESIGN.md section 5."""
import numpy as np
rng=np.random.default_rng(0)
rho=33334/1e6; rc=12.0; edge=100.0/8   # 8x8 pencils of 12.5
def one_item():
    # i chunk: 64 atoms in own pencil, x-sorted contiguous
    L=64/(rho*edge*edge)
    xi=np.sort(rng.uniform(0,L,64)); yi=rng.uniform(0,edge,64); zi=rng.uniform(0,edge,64)
    I=np.stack([xi,yi,zi],1)
    cols=[]   # list of hit masks per column in stream order
    for (dy,dz) in [(0,0),(1,0),(-1,1),(0,1),(1,1)]:
        x0,x1=-rc,L+rc
        n=rng.poisson(rho*(x1-x0)*edge*edge)
        xj=np.sort(rng.uniform(x0,x1,n)); yj=rng.uniform(0,edge,n)+dy*edge; zj=rng.uniform(0,edge,n)+dz*edge
        J=np.stack([xj,yj,zj],1)
        d2=((I[None,:,:]-J[:,None,:])**2).sum(-1)   # [nj,64]
        m=d2<rc*rc
        if (dy,dz)==(0,0):
            # own pencil: only j beyond the chunk in x order (approx: xj > xi), half of pairs
            m&= (xj[:,None]>xi[None,:])
            keep=xj>0   # window starts at chunk start
            m=m[keep]
        cols.append(m)
    return np.concatenate(cols,0)
def simulate(D, items=40):
    colsA=0; hits=0; popsB=0; popLanes=0; cols=0
    for _ in range(items):
        M=one_item(); cols+=len(M); hits+=M.sum()
        cnt=np.zeros(64,int)
        for g in range(0,len(M),4):
            cnt+=M[g:g+4].sum(0)
            while cnt.max()>D-4:
                act=cnt>0
                popsB+=1; popLanes+=act.sum(); cnt[act]-=1
        # end of item: flush
        while cnt.max()>0:
            act=cnt>0; popsB+=1; popLanes+=act.sum(); cnt[act]-=1
    popsA=hits/64
    valuA=cols*(3+1+3)+popsA*9
    valuB=cols*(3+1+1+0.25)+popsB*10
    return cols/items, hits/items, hits/cols/64, popLanes/popsB/64, valuA/items, valuB/items
for D in (8,12,16,24,32):
    c,h,hr,util,a,b=simulate(D)
    print(f"D={D}: cols/item {c:.0f}, hits/item {h:.0f}, lane hit rate {hr:.3f}, pop utilisation {util:.2f}, VALU A {a:.0f} B {b:.0f} ratio {b/a:.3f}")

"""Host-side emulation of the index math of scripts/ubench/bhq32_probe.hip: every fragment read must land on the LDS granule the DMA
roles wrote for that (wave row, halo pixel, channel granule); ds_read_b128 groups of 16 lanes must be bank-conflict free; the counted
vmcnt table must equal the number of pieces issued after filter tile t + 1.  Runs anywhere:  python scripts/ubench/check_bhq32_indexing.py"""
RB=64; HP=20; NPX=200; NPIECE=13; WR_B=NPIECE*16*RB; HSLOT=4*WR_B; B_OFF=2*HSLOT; BSL=128*RB; PAD_OFF=B_OFF+4*BSL
lds={}   # byte address (16-B granule) -> tag
def put(addr, tag):
    assert addr%16==0
    assert addr not in lds or lds[addr]==tag, ("overwrite", addr, lds[addr], tag)
    lds[addr]=tag
chunk=0
# halo writer
for t in range(7):
    for wave in range(8):
        idp=8*t+wave; live=idp<4*NPIECE
        j=idp//NPIECE if live else 0; q=idp-j*NPIECE if live else 0
        for lane in range(64):
            hp=16*q+(lane>>2); hy=hp//HP; hx=hp-hy*HP
            g=(lane&3)^((hx>>2)&3)
            dst=((chunk&1)*HSLOT + j*WR_B + q*1024) if live else PAD_OFF
            addr=dst+lane*16
            if live:
                ok = hp<NPX and hx<18
                put(addr, ("H", j, hy, hx, g, ok))
# reader A
bad=0
for wave in range(8):
    wr=wave>>1
    for lane in range(64):
        l31=lane&31; half=lane>>5; tx=l31&15; tyl=l31>>4
        for ta in range(3):
            for tb in range(3):
                hx=tx+tb; sw=(hx>>2)&3
                base=(chunk&1)*HSLOT+wr*WR_B+((ta+tyl)*HP+hx)*RB
                for mb in range(4):
                    for ks in range(2):
                        a=base+(2*mb)*HP*RB+(((2*ks+half)^sw)<<4)
                        tag=lds.get(a)
                        want=("H", wr, ta+tyl+2*mb, hx, 2*ks+half, True)
                        if tag!=want:
                            bad+=1
                            if bad<5: print("A mismatch", wave,lane,ta,tb,mb,ks,a,tag,want)
print("A reads checked, mismatches:", bad)
# bank conflicts for A: groups of 16 lanes
def banks(addrs):
    used={}
    for a in addrs:
        for w in range(4):
            b=((a>>2)+w)%64
            used.setdefault(b,set()).add(a)
    return max(len(v) for v in used.values())
worst=0
for ta in range(3):
    for tb in range(3):
        for mb in range(4):
            for ks in range(2):
                for grp in range(4):
                    addrs=[]
                    for lane in range(16*grp,16*grp+16):
                        l31=lane&31; half=lane>>5; tx=l31&15; tyl=l31>>4
                        hx=tx+tb; sw=(hx>>2)&3
                        addrs.append(((ta+tyl)*HP+hx)*RB+(2*mb)*HP*RB+(((2*ks+half)^sw)<<4))
                    worst=max(worst,banks(addrs))
print("A worst bank multiplicity within a 16-lane group:", worst)
# filter writer/reader
ldsb={}
for wave in range(8):
    for lane in range(64):
        n=16*wave+(lane>>2); g=(lane&3)^((n>>2)&3)
        addr=B_OFF+wave*1024+lane*16
        ldsb[addr]=(n,g)
bad=0; worst=0
for wave in range(8):
    wc=wave&1
    for nb in range(2):
        for ks in range(2):
            for grp in range(4):
                addrs=[]
                for lane in range(16*grp,16*grp+16):
                    l31=lane&31; half=lane>>5
                    n=wc*64+32*nb+l31
                    a=B_OFF+n*RB+(((2*ks+half)^((n>>2)&3))<<4)
                    if ldsb.get(a)!=(n,2*ks+half): bad+=1
                    addrs.append(a)
                worst=max(worst,banks(addrs))
print("B mismatches:", bad, "worst bank multiplicity:", worst)
# wait-count table: pieces younger than filter tile t+1 with issue order (B, H) per k-tile, H in taps 0..6
seq=[]
for t in range(-3,0): pass
events=[]  # list of (kind, tile)
# prologue
for t in range(7): events.append(("H",-1))
for t in range(3): events.append(("B",t))
res={}
for T in range(27):
    tap=T%9
    events.append(("B",T+3))
    if tap<7: events.append(("H",T))
    idx=events.index(("B",T+1))
    res.setdefault(tap,set()).add(len(events)-1-idx)
print("younger-than-B(t+1) counts per tap:", {k:sorted(v) for k,v in res.items()})

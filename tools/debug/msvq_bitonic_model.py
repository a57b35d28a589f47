import numpy as np
rng=np.random.default_rng(1)
def dpp(v, kind):
    # v: array[64]; returns partner values per lane within 16-lane rows
    out=v.copy()
    for r in range(4):
        row=v[16*r:16*r+16]
        if kind=='x1': p=[i^1 for i in range(16)]
        elif kind=='x2': p=[i^2 for i in range(16)]
        elif kind=='x4': p=[i^4 for i in range(16)]
        elif kind=='x8': p=[i^8 for i in range(16)]
        elif kind=='f4': p=[(i&~3)|(3-(i&3)) for i in range(16)]
        elif kind=='f8': p=[(i&~7)|(7-(i&7)) for i in range(16)]
        elif kind=='f16': p=[15-i for i in range(16)]
        out[16*r:16*r+16]=row[p]
    return out
lane=np.arange(64)&15
def cx(v, kind, bit):
    p=dpp(v,kind)
    km=(lane&bit)==0
    lt=p<v
    take=(lt==km)
    return np.where(take,p,v)
SORT=[('x1',1),('f4',2),('x1',1),('f8',4),('x2',2),('x1',1),('f16',8),('x4',4),('x2',2),('x1',1)]
MERGE=[('x8',8),('x4',4),('x2',2),('x1',1)]
def rowsort(v):
    for k,b in SORT: v=cx(v,k,b)
    return v
def merge(a,b):
    c=np.minimum(a,dpp(b,'f16'))
    for k,bt in MERGE: c=cx(c,k,bt)
    return c
def p16swap(v):  # permlane16_swap(v,v): [0]=even rows in both rows of half, [1]=odd rows
    a=v.copy(); b=v.copy()
    for h in range(2):
        a[32*h+16:32*h+32]=v[32*h:32*h+16]
        b[32*h:32*h+16]=v[32*h+16:32*h+32]
    return a,b
def p32swap(v):
    a=v.copy(); b=v.copy()
    a[32:]=v[:32]; b[:32]=v[32:]
    return a,b
for trial in range(2000):
    R=rng.choice([1,2,4])
    total=rng.choice([32,64]) if R==1 else 64*R
    KEYMAX=2**62
    regs=[np.full(64,KEYMAX,dtype=np.int64) for _ in range(4)]
    keys=rng.integers(-2**40,2**40,size=total)
    keys=np.unique(keys)
    while keys.size<total: keys=np.unique(np.concatenate([keys,rng.integers(-2**40,2**40,size=total)]))[:total]
    rng.shuffle(keys)
    for t in range(total): regs[t//64][t%64]=keys[t]
    rs=[rowsort(regs[j]) for j in range(R)]
    if R==4: z=merge(merge(rs[0],rs[1]),merge(rs[2],rs[3]))
    elif R==2: z=merge(rs[0],rs[1])
    else: z=rs[0]
    a,b=p16swap(z); z=merge(a,b)
    a,b=p32swap(z); z=merge(a,b)
    want=np.sort(keys)[:16]
    for r in range(4):
        assert np.array_equal(z[16*r:16*r+16][:min(16,total)], want[:min(16,total)]), (trial,R,total,r)
print("network ok")

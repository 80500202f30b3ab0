"""Trips of the lane-paced region build (k_part_build_q: 512 lanes, every lane walks through its own items, a wave is through when its
slowest lane is) for a region of 4096 slots filled to a given load, by probe sequence: linear, double hashing, triangular.
Run under `timeout`; pure-Python loops."""
import numpy as np
rng=np.random.default_rng(1)
R=4096; T=512
def sim(n, probe="linear", reps=4, B=8):
    res=[]
    for _ in range(reps):
        home=rng.integers(0,R,n); step=rng.integers(0,R//2,n)*2+1
        tab=np.zeros(R,bool)
        q=[list(range(l,n,T)) for l in range(T)]
        cur=np.full(T,-1); off=np.zeros(T,int); k=np.zeros(T,int)
        trips=np.zeros(T,int); t=0
        while True:
            act=False
            for l in range(T):
                if cur[l]<0 and q[l]:
                    cur[l]=q[l].pop(0); off[l]=home[cur[l]]; k[l]=0
            claims={}
            for l in range(T):
                if cur[l]>=0:
                    act=True
                    s=off[l]
                    if not tab[s] and s not in claims: claims[s]=l
            if not act: break
            for l in range(T):
                if cur[l]>=0:
                    s=off[l]; i=cur[l]
                    if claims.get(s)==l: tab[s]=True; cur[l]=-1
                    else:
                        k[l]+=1
                        if probe=="linear": off[l]=(s+1)%R
                        elif probe=="double": off[l]=(s+step[i])%R
                        elif probe=="quad": off[l]=(home[i]+k[l]*(k[l]+1)//2)%R
                    trips[l]=t+1
            t+=1
        w=trips.reshape(-1,64).max(1)
        res.append((w.mean(), trips.mean(), t))
    return np.array(res).mean(0)
for n in (1815, 2720, 3277):
    print("n=%d load %.2f:"%(n,n/R), end="")
    for p in ("linear","double","quad"):
        a=sim(n,p)
        print("  %s wave-max %.1f (wg %.1f)"%(p,a[0],a[2]), end="")
    print()

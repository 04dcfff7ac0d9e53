import numpy as np, sys
sys.path.insert(0,'.')
from imcoalhmm_amd import synth
def compressible(n, seed, nsym=3):
    rng = np.random.default_rng(seed)
    p = np.full(nsym, 0.1 / max(nsym - 1, 1)); p[0] = 0.9
    return rng.choice(nsym, size=n, p=p / p.sum()).astype(np.uint8)
n=150
c = compressible(120000, seed=n+11)
for b in range(2):
    pi,T,E = synth.random_hmm(n,3,seed=9000+n+b,stay=0.9)
    C = [E[:,s][:,None]*T.T for s in range(3)]
    for seg in (1,5):
        P = np.eye(n)
        off = seg*4096
        for t in range(1024):
            P = C[c[off+t]] @ P
            if t%16==15: P /= P.max()
            if t in (63,127,255,511,1023):
                s = P.sum(0); cs = np.argmax(s)
                r = (P*s[cs])/(P[:,[cs]]*s[None,:])
                print(b,seg,t+1,np.abs(r-1).max())

import numpy as np, sys
pr = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8).astype(np.int64)
grid=int(sys.argv[2]) if len(sys.argv)>2 else 1024
ng=len(pr); per_xcd=(ng+7)//8; nbx=grid//8
L=np.zeros((8,nbx)); R=4 if len(sys.argv)<4 else int(sys.argv[3]); D=np.zeros((8,nbx,R)); 
for x in range(8):
    g=pr[x*per_xcd:min(ng,(x+1)*per_xcd)]
    for w in range(nbx):
        idx=list(range(w,len(g),nbx))
        if not idx: continue
        L[x,w]=g[idx[-1],6]-g[idx[0],0]
        for r,i in enumerate(idx[:R]): D[x,w,r]=g[i,6]-g[i,0]
m=L.mean(0)
print("lifetime mean %.1fk max %.1fk"%(L.mean()/1000,L.max()/1000))
step=max(8,nbx//16)
for w0 in range(0,nbx,step): print(w0, "life %.1f"%(m[w0:w0+step].mean()/1000), " ".join("r%d %.1f"%(r,D[:,w0:w0+step,r].mean()/1000) for r in range(R)))

"""dev: numpy prototype of the centre relocation of ivf_build_impl (csrc/ivf.hip, round 5) next to the reference call of web.py:522-536
(sklearn MiniBatchKMeans, init="random"): objective ratio with / without relocation on separated blobs and on a broad Gaussian, and that the
objective never increases.  python tools/proto_kmeans_relocation.py"""
import numpy as np, time
from sklearn.cluster import MiniBatchKMeans
def assign(x,c):
    d=(x*x).sum(1)[:,None]+(c*c).sum(1)[None,:]-2*x@c.T
    a=d.argmin(1); return a, np.maximum(d[np.arange(len(x)),a],0)
def lloyd(x,k,niter,reloc,seed=0,R_frac=0.05):
    rng=np.random.default_rng(seed)
    c=x[rng.choice(len(x),k,replace=False)].astype(np.float64)
    xs=x.astype(np.float64)
    objs=[]
    for it in range(niter+1):
        a,dist=assign(xs,c)
        objs.append(dist.sum())
        if it==niter: break
        n=np.bincount(a,minlength=k)
        for j in range(k):
            if n[j]: c[j]=xs[a==j].mean(0)
        if reloc:
            # distances to own (new) centres
            dist=((xs-c[a])**2).sum(1)
            S=np.bincount(a,weights=dist,minlength=k)
            cc=(c*c).sum(1)[:,None]+(c*c).sum(1)[None,:]-2*c@c.T
            np.fill_diagonal(cc,np.inf)
            nn=cc.argmin(1); dn=np.maximum(cc[np.arange(k),nn],0)
            cost=n*dn
            order_rm=np.argsort(cost)
            order_sp=np.argsort(-S)
            # used: moved centre / split cluster / receiver of a move of this iteration; moved: centres no longer at their old place -- a later
            # candidate whose nearest centre was moved has a stale cost and is skipped (same rule as csrc/ivf.hip)
            used=set(); moved=set(); nrel=0; R=max(1,int(k*R_frac)); si=0
            small=lambda o: n[o]<2
            for j in order_rm[:4*R]:
                if nrel>=R: break
                if j in used or nn[j] in moved: continue
                while si<k and (order_sp[si] in used or small(order_sp[si])): si+=1      # out for every later candidate
                sj=si
                while sj<k and (order_sp[sj] in used or small(order_sp[sj]) or order_sp[sj]==j or order_sp[sj]==nn[j]): sj+=1  # out for this one
                if sj>=k: break
                o=order_sp[sj]
                idx=np.nonzero(a==o)[0]
                p=xs[idx[dist[idx].argmax()]]
                dp=((xs[idx]-p)**2).sum(1)
                gain=np.maximum(dist[idx]-dp,0).sum()
                if gain>cost[j]:
                    c[j]=p; used.update([j,o,nn[j]]); moved.add(j); nrel+=1
                else:
                    break
    return c.astype(np.float32), objs
def obj(x,c):
    a,d=assign(x.astype(np.float64),c.astype(np.float64)); return d.sum()
rng=np.random.default_rng(0)
cent=(rng.standard_normal((200,64))*3).astype(np.float32)
blobs=(cent[rng.integers(0,200,30000)]+rng.standard_normal((30000,64))).astype(np.float32)
broad=rng.standard_normal((30000,64)).astype(np.float32)
for name,x in (("blobs",blobs),("broad",broad)):
    ref=MiniBatchKMeans(n_clusters=200,batch_size=2048,compute_labels=False,init="random",random_state=0).fit(x).cluster_centers_
    o_ref=obj(x,ref)
    for reloc in (False,True):
        t=time.time(); c,objs=lloyd(x,200,10,reloc); 
        print(name,"reloc",reloc,"ratio %.3f"%(obj(x,c)/o_ref),"mono",all(np.diff(objs)<=1e-9*objs[0]), "%.1fs"%(time.time()-t), ["%.3g"%o for o in objs[::2]])

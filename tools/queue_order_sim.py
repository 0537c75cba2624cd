#!/usr/bin/env python3
"""Developer probe (CPU): list-scheduling replay of the f64 search queues (8 queues x 256 waves, stealing) on the unit durations of a
tools/trace_units.py trace (IGT_TRACE_SAVE=gpurun_out/r04tr/lattice_trace.npz): span of the current classes, of LPT on the true
durations (the bound), and of orders built from per-group means / quantiles of the durations."""
import sys, numpy as np, heapq
sys.path.insert(0,'./igt-mpc-int_amd'); sys.path.insert(0,'./oracle')
from igtmpc.scenarios import make_batch
B=4096; W=4; N=20; dt=0.1
b=make_batch(B,dtype=np.float64)
t=np.load('./gpurun_out/r04tr/lattice_trace.npz')
ub,p,t0,t1,wave=t['b'],t['p'],t['t0'].astype(float),t['t1'].astype(float),t['wave']
D=np.zeros((B,W)); D[ub,p]=(t1-t0)/100
x0=b['x0']; kp=b['kparams']
s0=x0[:,2].astype(np.float32); v0=x0[:,5].astype(np.float32)
b0,b1,kv=kp[:,0].astype(np.float32),kp[:,1].astype(np.float32),kp[:,2].astype(np.float32)
reach=s0+np.float32(1.5)*np.maximum(v0,1)*np.float32(N*dt)
arc=(kv!=0)&(reach>=b0)&(s0<=b1)
def cls_current():
    c=np.zeros((B,W),int)
    for pp in range(W):
        frac=np.where((pp==0)|arc,0.95,np.clip(1.05-0.15*v0,0.4,0.95))
        cost=frac*np.where(arc,1.9,1.0)
        c[:,pp]=np.clip(((1.85-cost)*(8/1.5)).astype(int),0,7)
    return c
def simulate(key, label):
    # key[b,p]: sort key (ascending = first); stable by (j,p) inside the queue
    span=0; ends=[]
    # 8 queues, queue q owns scenarios b with (b%8 ... ) queue_scenario(q,j)=8j+((q-j)&7)
    queues=[]
    for q in range(8):
        items=[]
        for j in range(B//8):
            bb=8*j+((q-j)&7)
            for pp in range(W): items.append((key[bb,pp], j*W+pp, bb, pp))
        items.sort(key=lambda z:(z[0],z[1]))
        queues.append([(bb,pp) for _,_,bb,pp in items])
    # 256 waves per queue; list scheduling; stealing from next queues when dry
    pos=[0]*8
    heap=[(0.0,w,w%8) for w in range(2048)]
    heapq.heapify(heap)
    tmax=0
    while heap:
        tnow,w,q=heapq.heappop(heap)
        got=None
        for d in range(8):
            qq=(q+d)%8
            if pos[qq]<len(queues[qq]):
                got=queues[qq][pos[qq]]; pos[qq]+=1; break
        if got is None: tmax=max(tmax,tnow); continue
        heapq.heappush(heap,(tnow+D[got]+0.3,w,q))
    print(f'{label}: simulated span {tmax:.1f} us   (packed {D.sum()/2048:.1f})')
    return tmax
c=cls_current()
simulate(c,'current classes')
simulate(-D,'oracle LPT (true durations)')
simulate(np.zeros((B,W)),'index order')
# better predictors
Dm=D.copy()
# per-class mean duration check
for cc in range(8):
    m=c==cc
    if m.sum(): print('class',cc,'n',m.sum(),'mean',D[m].mean().round(1),'p90',np.percentile(D[m],90).round(1),'max',D[m].max().round(1))
np.savez('/tmp/lat.npz',D=D,c=c,arc=arc,v0=v0)
print('--- arc units by p')
for pp in range(W):
    m=arc
    print('arc p',pp,'mean',D[m,pp].mean().round(1),'p90',np.percentile(D[m,pp],90).round(1))
    print('straight p',pp,'mean',D[~m,pp].mean().round(1),'p90',np.percentile(D[~m,pp],90).round(1))
c2=c.copy()
for pp in range(W): c2[arc,pp]=pp
simulate(c2,'arc units split by p into classes 0-3')
c3=c.copy()
c3[arc,0]=0;c3[arc,1]=0;c3[arc,2]=1;c3[arc,3]=2
simulate(c3,'arc p0,p1 ->0, p2->1, p3->2')
# arc & slow vs fast
ey=np.abs(x0[:,3]); 
for lo,hi in ((0,1),(1,2),(2,3),(3,4),(4,5.1)):
    m=arc&(v0>=lo)&(v0<hi)
    if m.sum(): print('arc v0',lo,hi,'n',m.sum(),'u0',D[m,0].mean().round(1),'u1',D[m,1].mean().round(1),'u2',D[m,2].mean().round(1),'u3',D[m,3].mean().round(1))
# distance to arc start
dist=np.maximum(b0-s0,0)
for lo,hi in ((0,0.01),(0.01,3),(3,6),(6,10),(10,20)):
    m=arc&(dist>=lo)&(dist<hi)
    if m.sum(): print('arc dist',lo,hi,'n',m.sum(),'u0',D[m,0].mean().round(1),'u1',D[m,1].mean().round(1),'u2',D[m,2].mean().round(1),'u3',D[m,3].mean().round(1))
print('--- learned table predictor')
vb=np.minimum((v0).astype(int),4)
db=np.digitize(dist,[0.01,3,6,10])
pred=np.zeros((B,W))
for a in (0,1):
  for pp in range(W):
    for vv in range(5):
      for dd in range(5):
        m=(arc==a)&(vb==vv)&((db==dd)|(a==0))
        if m.sum(): pred[m,pp]=D[m,pp].mean()
simulate(-pred,'key = -E[dur | arc,p,v0 bin,dist bin] (continuous)')
# quantize into 8 classes with thresholds
for thr in ([45,35,27,20,15,10,6],[50,40,30,22,16,11,7],[40,30,24,18,13,9,5]):
    cq=np.digitize(-pred,-np.array(thr,float))
    simulate(cq,f'8 classes thresholds {thr}')
print('--- quantile predictor')
for qq in (90,97,99.5):
    predq=np.zeros((B,W))
    for a in (0,1):
      for pp in range(W):
        for vv in range(5):
            m=(arc==a)&(vb==vv)
            if m.sum(): predq[m,pp]=np.percentile(D[m,pp],qq)
    simulate(-predq,f'key = -p{qq}[dur | arc,p,v0 bin]')

"""Analysis (oracle side, CPU): how much of the compact list of feasible candidates (gt_mpc cost) survives a bound on the value
network -- global |V| bound, interval bound on the scenario's feature box, the same with one exact anchor (what value_bound_kernel
does), and an a-priori box with unit-local minima.       python tools/prune_probe.py [B=128]"""
import sys; import os; ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0]=[os.path.join(ROOT,'igt-mpc-int_amd'),os.path.join(ROOT,'oracle')]
import numpy as np, np_oracle as O
from igtmpc.scenarios import make_batch
from igtmpc.cinf import cinf_halfplanes
from igtmpc.value_nets import shipped_value_net
B=int(sys.argv[1]) if len(sys.argv)>1 else 128
P=O.Params(N=20)
b=make_batch(B, dtype=np.float64)
A,bb=cinf_halfplanes()
for sc in (1,3):
  layers=shipped_value_net(sc)['layers']
  net=dict(layers=layers, Wn=np.eye(6), mu_f=np.zeros(6), sigma_t=1.0, mu_t=0.0)
  Wl,bl=layers[-1]
  print('sc',sc,'sum|w_out|',np.abs(Wl).sum(),'b_out',bl, 'global dV', 2*np.abs(Wl).sum())
  for cand in ('lattice','track'):
    if cand=='lattice':
        r=O.solve_batch(b['x0'],b['u_prev'],b['kparams'],b['flags'],b['obs_xy'],A,bb,P,net=net,tv_sv=b['tv_sv'],enc=b['enc'],return_all=True)
    else:
        r=O.solve_batch_refined(b['x0'],b['u_prev'],b['kparams'],b['flags'],b['obs_xy'],A,bb,P,cand='track',net=net,tv_sv=b['tv_sv'],enc=b['enc'])[0]
    X,U,J,feas=r['X'],r['U'],r['J'],r['feas']
    sN,vN=X[...,2,-1],X[...,5,-1]
    V=O.terminal_value(net,sN,vN,b['tv_sv'],b['enc'])
    Jp=J+V            # J = Jp - V -> partial cost
    keep_ibp=0; keep_glob=0; tot=0; keep_lip=0
    for i in range(B):
        f=feas[i]
        if not f.any(): continue
        tot+=f.sum()
        jp=Jp[i][f]; v=V[i][f]
        # global bound
        dV=2*np.abs(Wl).sum()
        keep_glob+=(jp-jp.min()<=dV).sum()
        # IBP over the scenario's feature box
        feats=O.value_features(sN[i:i+1],vN[i:i+1],b['tv_sv'][i:i+1],b['enc'][i:i+1])[0][f]   # [n,6]
        lo,hi=feats.min(0),feats.max(0)
        c,rad=(lo+hi)/2,(hi-lo)/2
        for li,(W,bv) in enumerate(layers):
            c=W@c+bv; rad=np.abs(W)@rad
            if li+1<len(layers):
                l,h=np.tanh(c-rad),np.tanh(c+rad); c,rad=(l+h)/2,(h-l)/2
        Vlo,Vhi=(c-rad)[0],(c+rad)[0]
        assert (v>=Vlo-1e-9).all() and (v<=Vhi+1e-9).all()
        # candidate c survives if jp_c - Vhi <= min_c'(jp_c' - Vlo)
        keep_ibp+=(jp-Vhi<=(jp-Vlo).min()+0).sum()
        # two-anchor refinement: evaluate V exactly at the min-jp candidate: bound = jp_min - V(true at min)
        k=np.argmin(jp); ub=jp[k]-v[k]
        keep_lip+=(jp-Vhi<=ub).sum()
    print(f'  {cand}: feasible list {tot}  kept global {keep_glob/tot:.3f}  kept IBP {keep_ibp/tot:.3f}  kept IBP+exact anchor {keep_lip/tot:.3f}')

print('--- a-priori box (s0 + [0, 10.6], v in [0, 5.3]) with unit-local / scenario-local minimum of J - Vlo')
G=16
rank=np.empty(G,int); rank[np.argsort(np.abs(np.arange(G)-7.5),kind='stable')]=np.arange(G)
cidx=np.arange(256); unit_of=rank[cidx%G]//4
for sc in (1,3):
  layers=shipped_value_net(sc)['layers']
  net=dict(layers=layers, Wn=np.eye(6), mu_f=np.zeros(6), sigma_t=1.0, mu_t=0.0)
  for cand in ('lattice','track'):
    if cand=='lattice':
        r=O.solve_batch(b['x0'],b['u_prev'],b['kparams'],b['flags'],b['obs_xy'],A,bb,P,net=net,tv_sv=b['tv_sv'],enc=b['enc'],return_all=True)
    else:
        r=O.solve_batch_refined(b['x0'],b['u_prev'],b['kparams'],b['flags'],b['obs_xy'],A,bb,P,cand='track',net=net,tv_sv=b['tv_sv'],enc=b['enc'])[0]
    X,J,feas=r['X'],r['J'],r['feas']
    sN,vN=X[...,2,-1],X[...,5,-1]
    V=O.terminal_value(net,sN,vN,b['tv_sv'],b['enc']); Jp=J+V
    tot=ku=ks=0
    for i in range(B):
        f=feas[i]
        if not f.any(): continue
        s0=b['x0'][i,2]
        box_s=np.array([s0, s0+10.6]); box_v=np.array([0.0,5.3])
        f0=O.value_features(box_s[None,:1],box_v[None,:1],b['tv_sv'][i:i+1],b['enc'][i:i+1])[0,0]
        f1=O.value_features(box_s[None,1:],box_v[None,1:],b['tv_sv'][i:i+1],b['enc'][i:i+1])[0,0]
        lo,hi=np.minimum(f0,f1),np.maximum(f0,f1)
        c,rad=(lo+hi)/2,(hi-lo)/2
        for li,(W,bv) in enumerate(layers):
            c=W@c+bv; rad=np.abs(W)@rad
            if li+1<len(layers):
                l,h=np.tanh(c-rad),np.tanh(c+rad); c,rad=(l+h)/2,(h-l)/2
        Vlo,Vhi=(c-rad)[0],(c+rad)[0]
        tot+=f.sum()
        jp=np.where(f,Jp[i],np.inf)
        ks+=((jp-Vhi<=jp.min()-Vlo)&f).sum()
        for u in range(4):
            m=f&(unit_of==u)
            if m.any(): ku+=(jp[m]-Vhi<=jp[m].min()-Vlo).sum()
    print(f'  sc{sc} {cand}: kept with scenario-wide minimum {ks/tot:.3f}, with the unit\'s own minimum {ku/tot:.3f}')

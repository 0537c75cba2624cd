import sys, os
sys.path.insert(0,'/root/repo/oracle'); sys.path.insert(0,'/root/repo/igt-mpc-int_amd'); 
import numpy as np, np_oracle as O, closed_loop as CL
from igtmpc import routes as R
from igtmpc.cinf import cinf_halfplanes
from igtmpc.evaluate import initial_states
P=O.Params(N=20); cinf=cinf_halfplanes()
KE=float(sys.argv[2]); SPAN=float(sys.argv[3])
def solve_policy(x0,u_prev,kp,flags,obs,A,b,P,C=256,refine_iters=0,u_ws=None,**kw):
    x0a = O.apply_flags(x0, flags); B=x0.shape[0]; G=16
    c=np.arange(C); i,j=c//G,c%G
    ra, rd = P.dt*P.jerk, P.dt*P.steer_rate
    has_ws = (np.asarray(flags)&2)!=0 if u_ws is not None else np.zeros(B,bool)
    base_a = np.broadcast_to(u_prev[:,0,None],(B,P.N)).copy()
    if u_ws is not None: base_a[has_ws]=u_ws[has_ws,0]
    off_a = O.cand_m(i,G,True)*(P.N*ra)          # [C]
    off_b = O.cand_m(j,G,True)*SPAN               # beta offsets
    X=np.empty((B,C,7,P.N+1)); U=np.empty((B,C,2,P.N))
    x=np.broadcast_to(x0a[:,None,:],(B,C,7)).copy(); X[...,0]=x
    a=np.broadcast_to(u_prev[:,None,0],(B,C)).copy(); d=np.broadcast_to(u_prev[:,None,1],(B,C)).copy()
    for k in range(P.N):
        ta=np.clip(base_a[:,None,k]+off_a,P.a_min,P.a_max)
        a=np.clip(a+np.clip(ta-a,-ra,ra),P.a_min,P.a_max)
        beta=np.clip(-x[...,4]-KE*x[...,3]+off_b,-0.7,0.7)
        td=np.clip(np.arctan(2*np.tan(beta)),-P.df_max,P.df_max)
        d=np.clip(d+np.clip(td-d,-rd,rd),-P.df_max,P.df_max)
        U[...,0,k]=a; U[...,1,k]=d
        x=O.frenet_rk4_step(x,a,d,kp[:,None,:],P); X[...,k+1]=x
    J=O.stage_cost(X,U,P)
    g,mask=O.constraint_violation(X,U,u_prev[:,None,:],obs[:,None],A,b,P,check_rate=False)
    feas=(mask==0)&np.isfinite(J); Jm=np.where(feas,J,np.inf); arg=np.argmin(Jm,axis=1); ok=feas[np.arange(B),arg]
    out=dict(x=np.where(ok[:,None,None],X[np.arange(B),arg],np.nan),u=np.where(ok[:,None,None],U[np.arange(B),arg],np.nan),
             cost=np.where(ok,J[np.arange(B),arg],np.inf),argmin=np.where(ok,arg,-1),status=np.where(ok,0,1),X=X,U=U,J=J,g=g,mask=mask,feas=feas)
    return [out]
O.solve_batch_refined = solve_policy
tot=[]; dl=0
for sc in range(1,9):
    pairs=[R.SCENARIO_ROUTES[sc-1][int(sys.argv[4]) if len(sys.argv)>4 else 0]]
    x,_=initial_states(np.random.default_rng(2026+sc), pairs)
    r=CL.run_episode(x[0], pairs[0], P, cinf, M_sim=int(sys.argv[1]), cand_mode='ramp_hold')
    print('sc',sc,pairs,'infeasible',r['infeasible'],'final s %.1f %.1f'%(r['x_data'][2,-1],r['x_data'][9,-1]), 'max|ey| %.3f'%np.abs(r['x_data'][[3,10]]).max(), 'deadlock', r['deadlock'], flush=True)
    tot.append(r['infeasible'].sum()); dl+=r['deadlock']
print('total infeasible', sum(tot), 'of', 16*int(sys.argv[1]), 'deadlocks', dl)

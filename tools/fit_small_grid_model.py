"""Fit the launch cost model of the planner (csrc/gemm.hip: small_grid_plan) to the timings tools/sweep_small_m.py dumps
(gpurun_out/sweep_small_m_b1.json, ..._b16.json; round 4's are kept under profiles/r04_sweeps/): least squares on log(time) over
every measured (tile, stages, split) of the six candidates, then the regret of the model's argmin against the best measured variant
per shape.  CPU only.  usage: fit_small_grid_model.py profiles/r04_sweeps/sweep_small_m_b1.json profiles/r04_sweeps/sweep_small_m_b16.json"""
import json,sys,math,numpy as np
from scipy.optimize import least_squares
NCU=256
cdiv=lambda a,b:(a+b-1)//b
CANDS={(64,2):(64,64,4),(64,3):(64,64,3),(64,4):(64,64,2),(128,2):(128,128,2),(128,3):(128,128,1),(128,4):(128,128,1),(160,2):(128,160,2),(160,3):(128,160,1),(160,4):(128,160,1)}
USE=[(64,3),(64,4),(128,2),(128,4),(160,2),(160,3)]
names=['t1_64_3','t1_64_4','t1_128_2','t1_128_4','t1_160_2','t1_160_3','tk_64','tk_128','tk_160','launch','pro','epi64','epi128','epi160','red0','redbw','convmul']
x0=np.array([0.37,0.29,1.03,0.65,1.07,0.80,0.27,0.56,0.70,4.0,1.0,1.0,2.5,3.0,4.5,3.0,1.1])
def cost(x,kind,M,N,K,tile,st,sk):
    P=dict(zip(names,x))
    tm,tn,occ=CANDS[(tile,st)]
    nkt=cdiv(K,64); kps=cdiv(nkt,sk)
    wgs=cdiv(M,tm)*cdiv(N,tn)*sk
    n=cdiv(wgs,NCU)
    t1=P[f't1_{tile}_{st}']; tk=P[f'tk_{tile}']
    q,r=divmod(n,occ)
    full=(occ*tk if occ>1 else t1)
    rem=0 if r==0 else (t1 if r==1 else r*tk)
    per_kt=q*full+rem
    if kind=='conv': per_kt*=P['convmul']
    rounds=cdiv(n,occ)
    epi=P[f'epi{tile}']
    t=P['launch']+rounds*(P['pro']+epi)+kps*per_kt
    if sk>1: t+=P['red0']+(sk*M*N*4+M*N*2)/(P['redbw']*1e6)+rounds*epi*0.5
    return t
data=[]
for f in sys.argv[1:]:
    for e in json.load(open(f)):
        for tl,st,sk,t in e['variants']:
            if (tl,st) in USE: data.append((e['kind'],e['M'],e['N'],e['K'],tl,st,sk,t))
def resid(x):
    return [math.log(cost(x,*d[:7])/d[7]) for d in data]
r=least_squares(resid,x0,bounds=(x0*0.3,x0*3))
x=r.x
print({n:round(v,3) for n,v in zip(names,x)})
res=np.array(resid(x)); print('rms log err',res.std(),'max',np.abs(res).max())
# regret
tot=dict(auto=0,best=0,model=0,small=0)
for f in sys.argv[1:]:
    for e in json.load(open(f)):
        meas={(tl,st,sk):t for tl,st,sk,t in e['variants']}
        small={k:v for k,v in meas.items() if (k[0],k[1]) in USE}
        pred={k:cost(x,e['kind'],e['M'],e['N'],e['K'],*k) for k in small}
        km=min(pred,key=pred.get); kb=min(small,key=small.get); ka=min(meas,key=meas.get)
        tot['auto']+=e['auto']; tot['best']+=meas[ka]; tot['model']+=small[km]; tot['small']+=small[kb]
        flag='' if small[km]<=1.06*small[kb] else '  <<< %.2fx'%(small[km]/small[kb])
        print(f"{e['kind']} M{e['M']} N{e['N']} K{e['K']}".ljust(32),f"auto {e['auto']:6.1f} best {meas[ka]:6.1f} {str(ka):13s} small {small[kb]:6.1f} {str(kb):12s} model {str(km):12s} pred {pred[km]:6.1f} meas {small[km]:6.1f}{flag}")
print(tot)

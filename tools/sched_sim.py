"""Replays real per-lane period sequences (from the CPU oracle) through the generator/filter
scheduler of vs_synth_kernel and counts generator rounds / filter super-steps for a ring size and
a policy.  Cost model: a round or a super-step costs the same whether 1 or 64 lanes take part.
Usage: python tools/sched_sim.py <config index>   (numbers quoted in DESIGN.md section 4)"""
import sys, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import voice_synth_amd as vs
from voice_synth_amd import configs
from oracle import pyoracle as po

def get_T(index, nl, lane0=0):
    specs, fs, dur, _ = configs.config_specs(index, nl, lane0)
    lanes, d = vs.lanes_from_specs(specs)
    n = vs.num_samples(fs, d)
    Ts=[]
    for l in range(nl):
        f, recs, ncyc, nd = po.source_one(lanes[l], n, 4000)
        Ts.append(recs['T'][:ncyc].astype(np.int64))
    return Ts, n

def simulate(Ts, N, C, tbound, policy, SS=24):
    L=len(Ts)
    g=np.zeros(L,int); n=np.zeros(L,int); k=np.zeros(L,int)
    rounds=0; ssteps=0; gen_part=0; ss_part=0
    live=np.ones(L,bool)
    it=0
    while live.any():
        it+=1
        if it>10**6: raise RuntimeError('stuck')
        avail=g-n
        ready = live & ((avail>=SS) | (g>=N))
        cangen = live & (g<N) & (avail+tbound<=C)
        starving = live & (g<N) & (avail<SS)
        if policy=='eager':   # P1: generate whenever room, supersteps whenever any ready
            if cangen.any():
                idx=np.where(cangen)[0]
                for i in idx: g[i]+=Ts[i][k[i]]; k[i]+=1
                rounds+=1; gen_part+=len(idx); continue
            idx=np.where(ready)[0]
            n[idx]+=SS; ssteps+=1; ss_part+=len(idx)
            live &= n<N
        elif policy=='lazy':  # P2: supersteps only when all live ready; generate when someone starves
            if (ready==live).all():
                idx=np.where(live)[0]
                n[idx]+=SS; ssteps+=1; ss_part+=len(idx); live &= n<N
            else:
                idx=np.where(cangen)[0]
                assert len(idx)>0
                for i in idx: g[i]+=Ts[i][k[i]]; k[i]+=1
                rounds+=1; gen_part+=len(idx)
        elif policy.startswith('thr'):  # superstep if >= thr fraction ready, else generate; if nobody can generate, superstep
            thr=float(policy[3:])
            nlive=live.sum()
            if ready.sum()>=thr*nlive and ready.any() and not (starving.any() and ready.sum()<nlive and False):
                idx=np.where(ready)[0]
                n[idx]+=SS; ssteps+=1; ss_part+=len(idx); live &= n<N
            elif cangen.any():
                idx=np.where(cangen)[0]
                for i in idx: g[i]+=Ts[i][k[i]]; k[i]+=1
                rounds+=1; gen_part+=len(idx)
            else:
                idx=np.where(ready)[0]
                n[idx]+=SS; ssteps+=1; ss_part+=len(idx); live &= n<N
    return rounds, ssteps, gen_part/ (rounds*L), ss_part/(ssteps*L)

if __name__=='__main__':
    index=int(sys.argv[1]); 
    Ts,N=get_T(index,64)
    cyc=[len(t) for t in Ts]
    tb=max(int(t.max()) for t in Ts)
    P = int(np.median([t[0] for t in Ts]))
    tbound = int(np.floor(np.float32(1.2)*np.float32(P)))
    print('cycles min/mean/max',min(cyc),np.mean(cyc),max(cyc),'Tmax seen',tb,'tbound',tbound,'N',N,'ideal ss',N/24)
    Rg = 52.5*4*np.mean([t.mean() for t in Ts]); Rf=63*4*24
    for C in (192,216,240,264,288,312,336,360,408):
        for pol in ('eager','lazy','thr0.5','thr0.75','thr0.9'):
            try:
                r,s,gp,sp=simulate(Ts,N,C,tbound,pol)
                cost=(r*Rg+s*Rf)/N
                print('C=%d %-8s rounds %4d (part %.2f) ssteps %4d (part %.2f)  cost/sample %.0f'%(C,pol,r,gp,s,sp,cost))
            except Exception as e:
                print(C,pol,'ERR',e)

#!/usr/bin/env python3
"""dev: the ring walk of plan_device.hip (front entry + remaining weight, fixed point inside runs of equal step lengths,
run descriptors) emulated in Python against a port of RingSched::advance (plan.cpp / runmean.c:61-116) -- the algorithm
check that was run before the kernels were written.  usage: plan_walk_emulation.py"""
import struct

import numpy as np
S=250
def ref(lens):
    w=[0.0]*S; ins=[-1]*S; start=last=0; w[0]=5.0; ins[0]=-1
    out=[]; allops=[]
    for t,L in enumerate(lens):
        if L>=5.0:
            start=last=0; w[0]=5.0; ins[0]=t; out.append((-1,[])); continue
        left=L; i=start; ops=[]
        while left>0:
            if w[i]>left:
                w[i]-=left; ops.append((i,ins[i],left)); left=0
            else:
                ops.append((i,ins[i],w[i])); left-=w[i]; i=(i+1)%S
        start=i; i=(last+1)%S
        assert i!=start
        last=i; w[i]=L; ins[i]=t
        out.append((i,ops))
    return out
def dev(lens, minrun=48):
    n=len(lens); t=0; j=-1; r0=-1; wF=5.0; runStart=0; noFF=0
    out=[None]*n; runs=[]
    while t<n:
        L=lens[t]
        if t>0 and L!=lens[t-1]: runStart=t
        jB=j; wB=wF; ops=[]
        if L>=5.0:
            r0=t;j=t;wF=5.0;insSlot=-1
        else:
            left=L
            while left>0:
                assert j<t
                slot=(j-r0)%S; ins=j
                if wF>left: wop=left; wF-=left; left=0
                else:
                    wop=wF; left-=wF; j+=1; wF=lens[j] if j<t else 0.0
                ops.append((slot,ins,wop))
            insSlot=(t-r0)%S
        out[t]=(insSlot,ops)
        if t>=noFF and L<5.0 and jB>=runStart and j==jB+1 and struct.pack('d',wF)==struct.pack('d',wB) and t+1<n and lens[t+1]==L:
            t0=t; t+=1; e=t
            while e<n and lens[e]==L: e+=1
            K=e-t
            if K>=minrun:
                runs.append((t0,K))
                for i in range(1,K+1):
                    ins0,ops0=out[t0]
                    out[t0+i]=((ins0+i)%S,[((s+i)%S,a+i,w) for s,a,w in ops0])
                t+=K; j+=K
            else: noFF=e
            continue
        t+=1
    return out,runs
rng=np.random.default_rng(1)
cases={'half':[1/48]*17520,'niwot':list(np.tile([0.37,0.63],3000)+0), 'rand':list(rng.choice([1/48,1/24,0.3,0.5,1.0,2.0,3.0,6.0],5000)),
 'runs':sum([[float(rng.choice([1/48,1/24,0.25,1.0,2.5,0.1]))]*int(rng.integers(1,400)) for _ in range(80)],[]),
 'randf':list(rng.uniform(0.0202,1.5,4000)), 'three':[3.0]*200+[2.0]*200+[4.9]*100+[0.05]*300+[4.0]*5}
for k,l in cases.items():
    l=[float(x) for x in l]
    a=ref(l); b,runs=dev(l)
    bad=[t for t in range(len(l)) if a[t]!=b[t]]
    print(k,len(l),'runs',len(runs),'covered',sum(r[1] for r in runs),'bad',bad[:5])

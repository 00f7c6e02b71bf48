"""Closed-form replay of Adam decay-only steps (csrc/er_decay.h) against the fp32 step-by-step recurrence and its fp64
evaluation, in numpy: the error of each relative to the fp64 update, over gradient magnitudes from 1e-2 to 1e-10."""
import numpy as np
b1,b2,eps=np.float32(0.9),np.float32(0.999),np.float32(1e-8)
K=192;N=6
rng=np.random.default_rng(0)
# lr_t history with staircase + bias correction
T=5000
lr=np.maximum(1e-3*0.5**(np.arange(T)//1000),1e-5)
t=np.arange(1,T+1)
lrt=(lr*np.sqrt(1-0.999**t)/(1-0.9**t)).astype(np.float32)
q=np.sqrt(np.float64(b2)); B1=np.float64(b1)
s=np.arange(1,K+1)
coef=np.stack([B1**s*(-np.expm1(s*np.log(q)))**n for n in range(N)],1)  # [K,N]
def exact(var,m,v,t0,k):
    var=var.copy();m=m.copy();v=v.copy()
    for j in range(1,k+1):
        m=m*b1; v=v*b2
        var=var-(lrt[t0+j]*m)/(np.sqrt(v)+eps)
    return var,m,v
def exact64(var,m,v,t0,k):
    var=var.astype(np.float64);m=m.astype(np.float64);v=v.astype(np.float64)
    for j in range(1,k+1):
        m=m*np.float64(b1); v=v*np.float64(b2)
        var=var-(np.float64(lrt[t0+j])*m)/(np.sqrt(v)+np.float64(eps))
    return var,m,v
def closed(var,m,v,t0,k):
    kk=min(k,K)
    Tn=(lrt[t0+1:t0+kk+1].astype(np.float64)[:,None]*coef[:kk]).sum(0).astype(np.float32)
    a=np.sqrt(v); d=a+eps; z=a/d
    poly=np.zeros_like(var)+Tn[N-1]
    for n in range(N-2,-1,-1): poly=poly*z+Tn[n]
    var2=var-(m*poly)/d
    return var2,(m*np.float32(np.exp(k*np.log(B1)))),(v*np.float32(np.exp(k*np.log(np.float64(b2)))))
for gmag in [1e-2,1e-4,1e-5,1e-6,3e-7,1e-7,1e-8,1e-10]:
  worst=0;worst64=0
  for trial in range(20):
    g=(rng.standard_normal(16)*gmag).astype(np.float32)
    m=g*np.float32(0.1); v=g*g*np.float32(0.001)
    var=(rng.standard_normal(16)*0.0025).astype(np.float32)
    t0=int(rng.integers(0,3000)); k=int(rng.choice([1,2,5,20,64,150,300,1500]))
    e=exact(var,m,v,t0,k); c=closed(var,m,v,t0,k); e64=exact64(var,m,v,t0,k)
    du=np.abs((c[0]-var)-(e64[0]-var)).max()/ (np.abs(e64[0]-var).max()+1e-30)
    due=np.abs((e[0]-var)-(e64[0]-var)).max()/ (np.abs(e64[0]-var).max()+1e-30)
    worst=max(worst,du); worst64=max(worst64,due)
  print(gmag,'closed-vs-f64 rel upd err',worst,' fp32stepwise-vs-f64',worst64)

import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from infinite_video_amd import synth
from infinite_video_amd.engine import LTMEngine
import numpy as np
dev = torch.device("cuda:0")
N,H,DH,D,P,T,Q,L = 256,12,64,768,32,256,32,1
eng = LTMEngine(N,H,DH,D,P,tau=.75,sticky=True,n_layers=L,max_q=Q,device=dev)
projs=[tuple(torch.from_numpy(a).to(dev) for a in synth.layer_projections(0,D,H*DH))]
q=torch.randn(1,Q,H*DH,device=dev)
k=torch.randn(T*P,D,device=dev)
u=torch.rand(1,512,dtype=torch.float64,device=dev)
kbar=eng.pool(k)
eng.step(kbar,q,projs,None); eng.step(kbar,q,projs,u)
torch.cuda.synchronize()
def t(f,n=300):
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): f()
    t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
    return 1e6*(t1-t0)/n, 1e6*(t2-t0)/n
print("pool        host/e2e us", t(lambda: eng.pool(k)))
print("step        host/e2e us", t(lambda: eng.step(kbar,q,projs,u)))
print("has_memory  host us", t(lambda: eng.has_memory))
print("rand pair   host us", t(lambda: (torch.rand(512,dtype=torch.float64), torch.rand(512,dtype=torch.float64))))
pin=torch.empty(2,512,dtype=torch.float64).pin_memory(); dv=torch.empty(1,512,dtype=torch.float64,device=dev)
print("h2d copy    host/e2e us", t(lambda: dv.copy_(pin[0:1],non_blocking=True)))
print("empty ctx   host us", t(lambda: torch.empty(1,Q,768,device=dev)))
print("cuda.device ctx host us", t(lambda: torch.cuda.device(dev).__enter__()))
print("cur stream  host us", t(lambda: torch.cuda.current_stream(dev).cuda_stream))
import ctypes as C
from infinite_video_amd import _lib
from infinite_video_amd.engine import _ptr, _stream
arr = eng._proj_array(projs)
ctx = torch.empty(1, Q, 768, device=dev)
st = _stream(dev)
print("C step only host/e2e us", t(lambda: eng.lib.infv_ltm_step(eng._h, _ptr(kbar), T, _ptr(q), Q, arr, _ptr(u), _ptr(ctx), st)))
print("ensure_plan host us", t(lambda: eng.ensure_plan(T)))
print("_check_q    host us", t(lambda: eng._check_q(q)))
print("_proj_array host us", t(lambda: eng._proj_array(projs)))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(200): eng.step(kbar, q, projs, u)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(8)

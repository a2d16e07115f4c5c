import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import torch, bench, pioran_jl_amd as pj
from oracle import oracle as O
N=10000; B=4096
t,y,yerr=bench.synth_series(N)
dev=torch.device("cuda",0); stream=torch.cuda.current_stream(dev)
ctx=pj.Context(0, stream=stream.cuda_stream)
th,f_min,f_max=bench.synth_theta(B,t,y,seed=4321)
A,Bc,C,Dd=pj.approx_batch(pj.SingleBendingPowerLaw, th[:,:3], f_min,f_max,40,th[:,3],basis_function="SHO")
ds=pj.Dataset(t,y,yerr**2,ctx); ds.prepare(C,Dd)
d=[torch.from_numpy(np.ascontiguousarray(v)).to(dev) for v in (A,Bc,th[:,5].copy(),th[:,4].copy())]
dout=torch.empty(B,dtype=torch.float64,device=dev); dst=torch.zeros(B,dtype=torch.int32,device=dev)
def med_ms(f,reps=3):
    f(); torch.cuda.synchronize(); ts=[]
    for _ in range(reps):
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record(stream); f(); e1.record(stream); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))
ref=O.logl_batch(A[:8],Bc[:8],C,Dd,t,y,yerr**2,th[:8,5].copy(),th[:8,4].copy(),nthreads=8)
for nb in (4096,1024,512):
    go=lambda: ds.logl_batch_dev(nb,d[0].data_ptr(),d[1].data_ptr(),d[2].data_ptr(),d[3].data_ptr(),0,0,dout.data_ptr(),dst.data_ptr())
    ms=med_ms(go); got=dout[:8].cpu().numpy(); cfg=pj._lib.lib().pioran_celerite_config_name(0).decode()
    ctx.set_option("no_win2", True); msw=med_ms(go); gw=dout[:8].cpu().numpy(); ctx.set_option("no_win2", False)
    F=(N-1)*(5.5*80*80+18*80)*nb
    print(f"SHO-40 (80 rows) B={nb}: scan {cfg} {ms:.2f} ms = {nb/ms:.1f} k evals/s ({F/ms/1e9/78.6:.3f}); lean latency kernel {msw:.2f} ms = {nb/msw:.1f} k evals/s; maxrel vs oracle {np.max(np.abs(got-ref)/np.abs(ref)):.1e} / {np.max(np.abs(gw-ref)/np.abs(ref)):.1e}", flush=True)

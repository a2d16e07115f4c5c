import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import torch, bench, pioran_jl_amd as pj
N=10000; B=4096
t,y,yerr=bench.synth_series(N)
dev=torch.device("cuda",0); stream=torch.cuda.current_stream(dev)
ctx=pj.Context(0, stream=stream.cuda_stream)
th,f_min,f_max=bench.synth_theta(B,t,y,seed=4321)
def med_ms(f,reps=3):
    f(); torch.cuda.synchronize(); ts=[]
    for _ in range(reps):
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record(stream); f(); e1.record(stream); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))
for basis,J in (("SHO",39),("SHO",33),("DRWCelerite",24)):
    A,Bc,C,Dd=pj.approx_batch(pj.SingleBendingPowerLaw, th[:,:3], f_min,f_max,J,th[:,3],basis_function=basis)
    real=(Dd==0.0)&(Bc==0.0).all(axis=0); R=int(2*len(C)-real.sum())
    ds=pj.Dataset(t,y,yerr**2,ctx); ds.prepare(C,Dd,real.astype(np.int32))
    d=[torch.from_numpy(np.ascontiguousarray(v)).to(dev) for v in (A,Bc,th[:,5].copy(),th[:,4].copy())]
    dout=torch.empty(B,dtype=torch.float64,device=dev); dst=torch.zeros(B,dtype=torch.int32,device=dev)
    go=lambda: ds.logl_batch_dev(B,d[0].data_ptr(),d[1].data_ptr(),d[2].data_ptr(),d[3].data_ptr(),0,0,dout.data_ptr(),dst.data_ptr())
    res=[]
    for w in (False,True):
        ctx.set_option("win2", w); ms=med_ms(go); res.append((ms, dout.clone(), pj._lib.lib().pioran_celerite_config_name(0).decode()))
    ctx.set_option("win2", False)
    ok=torch.isfinite(res[0][1])&torch.isfinite(res[1][1])
    F=(N-1)*(5.5*R*R+18*R)*B
    print(basis,J,"rows",R,res[0][2],"step-by-step %.2f ms (%.3f)"%(res[0][0],F/res[0][0]/1e9/78.6),"two-step %.2f ms (%.3f)"%(res[1][0],F/res[1][0]/1e9/78.6),"maxrel %.1e"%float(((res[0][1][ok]-res[1][1][ok]).abs()/res[0][1][ok].abs()).max()),flush=True)

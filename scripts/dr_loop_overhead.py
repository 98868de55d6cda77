"""Host overhead of the in-library Douglas-Rachford loop (pg_dr_run): iterations/s with and without HIP-event profiling."""
import sys, time, numpy as np
sys.path.insert(0, '.')
import proximalalgorithms.jl_amd as pa
n=10_000_000; dtype=np.float32
rng=np.random.default_rng(0)
d=(0.1+rng.random(n)).astype(dtype); q=rng.standard_normal(n).astype(dtype); x0=np.zeros(n,dtype)
ctx=pa.get_context()
for block in (8,16,32):
    for prof in (False, True):
        itn=pa.DouglasRachfordIteration(f=pa.SeparableQuadratic(d,q), g=pa.IndBox(dtype(-0.5),dtype(0.25)), x0=x0, gamma=dtype(1.0), materialize=False)
        itn.device_run(2*block,0.0,block)
        ctx.profile(prof); ctx.profile_reset(); ctx.sync()
        steps=640
        t0=time.perf_counter(); s,k=itn.device_run(steps,0.0,block); ctx.sync(); dt=time.perf_counter()-t0
        ctx.profile(False)
        print(block, "profile", prof, round(steps/dt), "it/s", round(dt/ (steps/block)*1e6,1), "us per block")

export MSM_HIP_LIB=$PWD/ab_builds/libmsm_tune.so
for R in 0 1; do for B in 0 1; do echo "== MSM_RADIX=$R MSM_BINS=$B"; MSM_RADIX=$R MSM_BINS=$B python3 - <<'P'
import sys,time
sys.path.insert(0,'.')
from montgomery_amd.api import MsmContext
for curve in (0,1):
    ctx=MsmContext(curve); n=1<<20
    ctx.generate_points(n,seed=7); dev,_=ctx.generate_scalars(n,seed=9)
    for i in range(3): ctx.run_device(dev,n,no_tables=True)
    ts=[]
    for i in range(10):
        t=time.perf_counter(); r,info=ctx.run_device(dev,n,no_tables=True); ts.append((time.perf_counter()-t)*1e3)
    print(curve, round(min(ts),3), {k:round(v,3) for k,v in info['phase_ms'].items() if k in ('digits','sort','accumulate','reduce')})
    ctx.close()
P
done; done

# one-level sort against the radix split below 2^22 entries per window (tuning build: MSM_RADIX): tools/sortpath_sweep.sh
export MSM_HIP_LIB=$PWD/ab_builds/libmsm_tune.so
for R in 0 1; do echo "== MSM_RADIX=$R"; MSM_RADIX=$R python3 - <<'P'
import sys,time
sys.path.insert(0,'.')
from montgomery_amd.api import MsmContext
ctx=MsmContext(0)
for lg in (16,17,18,19,20,21):
    n=1<<lg
    ctx.generate_points(n,seed=7); dev,_=ctx.generate_scalars(n,seed=9)
    for i in range(3): ctx.run_device(dev,n,no_tables=True)
    ts=[]
    for i in range(12):
        t=time.perf_counter(); r,info=ctx.run_device(dev,n,no_tables=True); ts.append((time.perf_counter()-t)*1e3)
    print(lg, round(min(ts),3), 'sort', round(info['phase_ms']['sort'],3))
P
done

import sys, time, threading
sys.path.insert(0, "/root/repo")
from montgomery_amd.api import MsmContext
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 26
n = 1 << lg
ctxs = [MsmContext(), MsmContext()]
devs = []
for c in ctxs:
    c.generate_points(n, seed=7)
    d, _ = c.generate_scalars(n, seed=9)
    devs.append(d)
    c.run_device(d, n)
R = 6
t = time.perf_counter()
for i in range(2 * R): ctxs[0].run_device(devs[0], n)
one = (time.perf_counter() - t) / (2 * R) * 1e3
res = [None, None]
def work(k):
    for i in range(R): res[k] = ctxs[k].run_device(devs[k], n)[0]
th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
t = time.perf_counter()
for x in th: x.start()
for x in th: x.join()
two = (time.perf_counter() - t) / (2 * R) * 1e3
print(f"2^{lg}: one context {one:.2f} ms per MSM; two contexts in flight {two:.2f} ms per MSM; same result {res[0].as_tuple() == res[1].as_tuple()}")

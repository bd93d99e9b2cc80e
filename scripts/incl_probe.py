import sys,time
sys.path.insert(0,".")
import isvins_loader; isvins_loader.load()
from isvins_amd import backend, synth
ws = synth.make_windows(range(1024))
be = backend.Backend(11,5,max_landmarks=300,max_obs=max(w.n_obs for w in ws),max_batch=1024)
w2=[w.clone() for w in ws]; ptrs=be.marshal(w2)
for _ in range(5):
    t=time.perf_counter(); be.upload(w2, ptrs=ptrs); tu=time.perf_counter()-t; be.run_optimize(sync=True); to=time.perf_counter()-t-tu; be.download(w2, ptrs=ptrs, as_list=False); ti=time.perf_counter()-t
    print("incl: upload %.2f optimize %.2f download %.2f total %.2f ms -> %.0f windows/s" % (1e3*tu,1e3*to,1e3*(ti-tu-to),1e3*ti,1024/ti))

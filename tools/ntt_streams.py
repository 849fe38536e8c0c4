import ctypes, os, sys, threading, time
sys.path.insert(0, os.getcwd())
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import plonky2_demo_amd as p
from plonky2_demo_amd._lib import check, lib
B, lg = 64, 20
rng = np.random.default_rng(1)
x = rng.integers(0, 2**63, (B, 1 << lg), dtype=np.uint64)
for nctx in (1, 2, 4):
    ctxs = [p.Context(0) for _ in range(nctx)]
    per = B // nctx
    bufs = [c.alloc(per * (8 << lg)).upload(x[i * per:(i + 1) * per]) for i, c in enumerate(ctxs)]
    def work(i, reps):
        for _ in range(reps):
            check(lib.gl_ntt_forward(ctxs[i].handle, ctypes.c_void_p(bufs[i].ptr), lg, per))
        ctxs[i].synchronize()
    for reps in (2, 10):
        t0 = time.perf_counter()
        ths = [threading.Thread(target=work, args=(i, reps)) for i in range(nctx)]
        [t.start() for t in ths]; [t.join() for t in ths]
        dt = (time.perf_counter() - t0) / reps
    print("%d stream(s) x %d polynomials: %.4f ms per forward 2^20 x 64 -> %.1f GB/s algorithmic" % (nctx, per, dt * 1e3, 16.0 * B * (1 << lg) / dt / 1e9))
    for b in bufs: b.free()

"""Does freeing host arrays that were handed to HIP as pageable copy sources stall the next submission?  (tools, not product)"""
import gc, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, scenarios
from adaptive_matrix_solver_amd import Context
n = 8192
ctx = Context(0); ctx.set_matrix(scenarios.ginibre(2048, 1)[:1, :1].repeat(n, 0).repeat(n, 1) if False else np.zeros((n, n), dtype=np.complex128)); ctx.pop_reserve(200)
rng = np.random.default_rng(0)
def small_op(label):
    t0 = time.perf_counter(); ctx.pop_get(0, [1, 3, 5], n); print(f"{label}: next small read-back {(time.perf_counter() - t0) * 1e3:.2f} ms")
small_op("warm"); small_op("warm")
for trial in range(3):
    vecs = [rng.standard_normal(n) + 1j * rng.standard_normal(n) for _ in range(128)]       # 128 KB each: mmap'ed by malloc
    for i, v in enumerate(vecs):
        ctx.pop_put(0, [i], v[None, :])
    ctx.sync(); small_op("after 128 pop_put, arrays alive")
    del vecs, v; gc.collect()
    time.sleep(0.002)
    small_op("after freeing the 128 arrays")
    small_op("again")
big = rng.standard_normal((128, n)) + 1j * rng.standard_normal((128, n))
ctx.pop_put(0, list(range(128)), big); ctx.sync(); small_op("after one 16 MB pop_put, array alive")
del big; gc.collect(); small_op("after freeing the 16 MB array"); small_op("again")
out = ctx.pop_get(0, list(range(2)), n); del out; small_op("after freeing a 256 KB read-back target")
# which copy sizes leave a registration behind?  (mmap'ed buffers, so that every free is a munmap)
import mmap
from adaptive_matrix_solver_amd._cabi import _ptr
ctx1 = Context(0); ctx1.set_matrix(np.zeros((n, n), dtype=np.complex128)); ctx1.pop_reserve(130)
def small_op1():
    t0 = time.perf_counter(); ctx1.pop_get(0, [1, 3, 5], n); return (time.perf_counter() - t0) * 1e3
small_op1(); small_op1()
for direction in ("put", "get"):
    for kb in (4, 16, 64, 128, 256, 1024, 4096, 16384):
        rows = max(1, kb * 1024 // (16 * n)); ln = min(n, kb * 1024 // 16)
        sl = np.arange(rows, dtype=np.int32)
        worst = 0.0
        for rep in range(4):
            maps = []
            for k in range(16):
                m = mmap.mmap(-1, rows * ln * 16); a = np.frombuffer(m, dtype=np.complex128).reshape(rows, ln)
                f = ctx1.lib.maus_pop_put if direction == "put" else ctx1.lib.maus_pop_get
                assert f(ctx1.h, 0, _ptr(sl), rows, _ptr(a), ln) == 0
                del a
                maps.append(m)
            ctx1.sync()
            for m in maps:
                m.close()
            worst = max(worst, small_op1())
        print(f"{direction} {kb:6d} KB x 16 buffers, then munmap: next small operation, worst of 4: {worst:.2f} ms")

"""Do torch's bundled HIP runtime and libmaus_hip.so (linked against /opt/rocm's) coexist in one process, in either
initialisation order?  (The nccl path of bench.py needs torch.cuda AND the library's own context on the same device.)"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
A = r'''
import sys; sys.path.insert(0, %r)
import numpy as np
order = sys.argv[1]
def maus():
    from adaptive_matrix_solver_amd import Context
    c = Context(0); c.set_matrix(np.eye(8, dtype=np.complex128)); print("maus ctx ok:", c.device_info()["name"], flush=True); return c
def tch():
    import torch
    print("torch.cuda.is_available:", torch.cuda.is_available(), flush=True)
    t = torch.ones(4, device="cuda:0"); print("torch tensor ok:", float(t.sum()), flush=True)
if order == "maus_first": c = maus(); tch()
else: tch(); c = maus()
with open("/proc/self/maps") as f:
    libs = sorted({l.split()[-1] for l in f if "amdhip64" in l or "hsa-runtime" in l})
print("loaded:", libs)
''' % ROOT
for order in ("torch_first", "maus_first"):
    r = subprocess.run([sys.executable, "-c", A, order], capture_output=True, text=True)
    print("==", order, "rc", r.returncode); print(r.stdout[-1500:]); print(r.stderr[-800:])

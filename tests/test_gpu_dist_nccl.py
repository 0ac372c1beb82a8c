"""The RCCL collectives of libmaus_hip (csrc/comm.hip) behind dist.PopulationComm, executed on the GPU box.  One GPU is all
a test box has, and RCCL refuses two ranks on one device, so this is a one-rank communicator: every collective of a sharded
run (record all-gathers per phase, the device-to-device row exchange, the start-up broadcasts, the max over ranks of
bench.py) really goes through ncclAllGather / ncclBroadcast on the context's stream, and the run must reproduce the plain
single-process trajectory.  The partition / exchange logic for 2, 3 and 8 ranks is covered with gloo on the CPU
(tests/test_dist_gloo.py).  The process must not import torch: the product path has no such dependency."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

CHILD = r'''
import json, os, random, sys
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "tests")]
import numpy as np
import scenarios, snapshot
from adaptive_matrix_solver_amd.solver import MAUS_Solver, ProblemType, SolutionCandidate

def run(make_comm, name="eig64"):
    spec = scenarios.TRAJECTORIES[name]
    A, b = scenarios.build(spec)
    np.random.seed(spec["seed"]); random.seed(spec["seed"]); SolutionCandidate._candidate_id_counter = 0
    PT = {"eig": ProblemType.EIGENVALUE, "svd": ProblemType.SVD, "lin": ProblemType.SOLVE_LINEAR_SYSTEM}[spec["kind"]]
    comm = make_comm() if make_comm else None
    s = MAUS_Solver(A, PT, b_vector=b, initial_num_candidates=spec["P"], global_convergence_tol=spec["tol"],
                    quiet=True, comm=comm, pert_mode="mt19937")
    out = []
    for it in range(4):
        s.loop_body(it + 1)
        lam = (lambda c: c.sigma_k) if spec["kind"] == "svd" else (lambda c: c.lambda_k if spec["kind"] == "eig" else 0.0)
        out.append([[c.id, c.state.value, c.stuck_counter, c.local_psi_retries_needed, repr(complex(lam(c))), repr(float(c.residual_k))]
                    for c in s.candidates] + [snapshot.rng_digest()])
    # the rows themselves (they travelled device to device through RCCL in the sharded run)
    vec = (lambda c: c.right_v_k) if spec["kind"] == "svd" else ((lambda c: c.v_k) if spec["kind"] == "eig" else (lambda c: c.x_k))
    out.append([np.asarray(vec(c)).tobytes().hex()[:64] for c in s.candidates])
    return out, comm, s

from adaptive_matrix_solver_amd import dist as mdist
names = ("eig64", "svd5x4", "lin24", "herm16")
ref = [run(None, n)[0] for n in names]
got, stats = [], []
for n in names:
    o, comm, s = run(lambda: mdist.init_from_env("rccl"), n)      # one communicator per context
    assert comm is not None and comm.on_device and comm.transport == "rccl" and comm.world == 1
    got.append(o)
    # the start-up collectives on real RCCL: object / array broadcast, eigenvector broadcast, max over ranks
    assert comm.bcast_object({"cond": 12.5, "name": n}) == {"cond": 12.5, "name": n}
    a = np.arange(1000, dtype=np.float64) * 0.5
    assert np.array_equal(comm.bcast_array(a.copy()), a)
    assert comm.max_over_ranks(1.5) == 1.5
    if n == "herm16":
        V = np.linalg.qr(np.random.default_rng(1).standard_normal((16, 16)) + 0j)[0]
        comm.bcast_eigvecs(s.engine.ctx, V, 16)
        # the decomposition with the reduction / back-transformation on the device leaves V resident on the root: broadcast from there
        ev = s.engine.device_eigh(s.M)
        comm.bcast_eigvecs(s.engine.ctx, None, 16)
        Vd = s.engine.ctx.get_eigvecs()
        assert np.linalg.norm(s.M @ Vd - Vd * ev[None, :]) < 1e-12 and np.abs(Vd[0].imag).max() == 0.0
    if n == "eig64":
        # maus_comm_set_matrix (r04): the root uploads, or broadcasts the copy its device already holds; the matvec that follows
        # reads the broadcast matrix
        ctx = s.engine.ctx
        B = (np.random.default_rng(2).standard_normal((64, 64)) + 1j * np.random.default_rng(3).standard_normal((64, 64)))
        ctx.comm_set_matrix(B, comm.rank, 0)
        ctx.pop_reserve(4)
        X = np.eye(4, 64, dtype=np.complex128)
        ctx.pop_put(0, [0, 1, 2, 3], X)
        num, den = ctx.matvec_rayleigh([0, 1, 2, 3])
        assert np.allclose(num, np.diag(B)[:4]) and np.allclose(den, 1.0)
        ctx.comm_set_matrix(B, comm.rank, 0, resident_on_root=True)
        num2, _ = ctx.matvec_rayleigh([0, 1, 2, 3])
        assert np.array_equal(num, num2)
    comm.barrier()
    stats.append(comm.stats())
    lib_stats = s.engine.ctx.comm_stats()
    assert lib_stats["collectives"] >= stats[-1]["collectives"] - 1
    s.engine.ctx.close()
print(json.dumps({"equal": ref == got, "collectives": [st["collectives"] for st in stats], "bytes": [st["bytes"] for st in stats],
                  "torch_imported": "torch" in sys.modules}))
'''


def test_population_comm_over_rccl_world_size_one():
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533",
               MAUS_FORCE_COMM="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", "ROOT = %r\n" % ROOT + CHILD], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    rec = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["equal"], "the run through RCCL differs from the single-process run"
    # (r04: a direct-solver step is one record exchange + one row exchange, tests/test_dist_gloo.py::test_one_record_exchange_per_step)
    assert min(rec["collectives"][:3]) >= 2 * 4 and min(rec["bytes"]) > 0
    assert not rec["torch_imported"], "the sharded product path must not need torch"

"""The `nccl` (= RCCL) branch of dist.PopulationComm executed on the GPU box.  One GPU is all a test box has, and RCCL
refuses two ranks on one device, so this is a world-size-1 process group: every collective of a sharded run (record
all-gathers per phase, row sync, the MAX all-reduce of bench.py) really goes through RCCL with device tensors, and the
run must reproduce the plain single-process trajectory.  The 2-rank partition / exchange logic is covered with gloo on
the CPU (tests/test_dist_gloo.py)."""
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

def run(comm, name="eig64"):
    spec = scenarios.TRAJECTORIES[name]
    A, b = scenarios.build(spec)
    np.random.seed(spec["seed"]); random.seed(spec["seed"]); SolutionCandidate._candidate_id_counter = 0
    PT = {"eig": ProblemType.EIGENVALUE, "svd": ProblemType.SVD, "lin": ProblemType.SOLVE_LINEAR_SYSTEM}[spec["kind"]]
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
    return out

from adaptive_matrix_solver_amd import dist as mdist
comm = mdist.init_from_env("nccl")           # before the first device context: torch's HIP runtime must load first
assert comm is not None and comm.on_device and comm.dist.get_backend() == "nccl"
ref = [run(None, n) for n in ("eig64", "svd5x4", "lin24")]
got = [run(comm, n) for n in ("eig64", "svd5x4", "lin24")]
import torch
t = torch.tensor([1.5], dtype=torch.float64, device=comm.device)
comm.dist.all_reduce(t, op=comm.dist.ReduceOp.MAX)
comm.barrier()
print(json.dumps({"equal": ref == got, "collectives": comm.collectives, "bytes": comm.bytes_gathered, "max": float(t.item())}))
'''


def test_population_comm_over_rccl_world_size_one():
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533",
               MAUS_FORCE_COMM="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", "ROOT = %r\n" % ROOT + CHILD], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    rec = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["equal"], "the run through RCCL differs from the single-process run"
    assert rec["collectives"] >= 3 * 4 * 4 and rec["bytes"] > 0 and rec["max"] == 1.5

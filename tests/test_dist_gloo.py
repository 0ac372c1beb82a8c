"""N>1 path on CPU: two gloo ranks, population block-sharded, per-phase all-gather of records and
rows.  Each rank runs the replicated host logic over the NumPy device double and must end with
exactly the bookkeeping, RNG stream positions and vectors of the single-process run."""
import json
import os
import random
import socket
import subprocess
import sys
import tempfile

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


# populations smaller than the world size: in every step some ranks own no candidate at all
EXTRA = {"eig16_p5": dict(kind="eig", build=("ginibre", 16, 16, 1.0), P=5, iters=4, seed=1234, tol=1e-8),
         "svd3x2_p3": dict(kind="svd", build=("svd", 3, 2, 21, -3.0), P=3, iters=4, seed=11, tol=1e-6),
         "herm16_p3": dict(kind="eig", build=("hermitian", 16, 16), P=3, iters=3, seed=5, tol=1e-8)}


def _run(name, iters, comm, per_body=None, **kw):
    sys.path[:0] = [ROOT, os.path.join(HERE, "golden"), HERE]
    import scenarios
    import snapshot
    scenarios.TRAJECTORIES.update(EXTRA)
    from test_host_logic import make_solver, rows_of
    solver, spec = make_solver(name, comm=comm, **kw)
    out = []
    for it in range(iters):
        c0 = comm.collectives if comm is not None else 0
        solver.loop_body(it + 1)
        if per_body is not None:
            per_body.append((comm.collectives if comm is not None else 0) - c0)
        d = snapshot.digest_rows(rows_of(solver.candidates, spec["kind"]))
        out.append({"digest": d, "rng": snapshot.rng_digest(), "n": len(solver.candidates)})
    return out, (comm.collectives if comm is not None else 0)


def _worker(rank, world, port, name, iters, outdir, kw=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path[:0] = [ROOT]
    import scipy.linalg as sla
    from adaptive_matrix_solver_amd.dist import PopulationComm
    comm = PopulationComm("gloo")
    # what the sharding did: sizes of the partitions asked for, and who called the O(n^3) host routines
    asked, eigh_calls = [], [0]
    owners = comm.owners
    comm.owners = lambda n: (asked.append(int(n)), owners(n))[1]
    real_eigh = sla.eigh
    sla.eigh = lambda *a, **k: (eigh_calls.__setitem__(0, eigh_calls[0] + 1), real_eigh(*a, **k))[1]
    per_body = []
    out, ncoll = _run(name, iters, comm, per_body, **(kw or {}))
    with open(os.path.join(outdir, f"rank{rank}.json"), "w") as f:
        json.dump({"out": out, "collectives": ncoll, "asked": asked, "eigh_calls": eigh_calls[0],
                   "stats": comm.stats(), "per_body": per_body}, f)
    dist.barrier()
    dist.destroy_process_group()


def _spawn(world, name, iters, kw=None):
    import torch.multiprocessing as mp
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, _free_port(), name, iters, d, kw), nprocs=world, join=True)
        return [json.load(open(os.path.join(d, f"rank{r}.json"))) for r in range(world)]


@pytest.mark.parametrize("name,iters", [("eig16", 6), ("lap8", 12), ("svd5x4", 8), ("lin24", 6), ("herm16", 3)])
def test_two_ranks_equal_single_process(name, iters):
    ref, _ = _run(name, iters, None)
    got = _spawn(2, name, iters)
    for r in range(2):
        assert got[r]["out"] == ref, f"rank {r} diverged from the single-process run"
        assert got[r]["collectives"] > 0
    assert got[0]["collectives"] == got[1]["collectives"]


# world sizes that do not divide the population, and more ranks than active candidates: lin24 steps 10 -> 1 candidates,
# svd5x4 25 -> 13, herm16 12 (Hermitian shortcut: rank 0 alone decomposes the matrix and broadcasts lambda and V)
@pytest.mark.parametrize("world,name,iters", [(3, "eig16", 5), (3, "herm16", 3), (8, "lin24", 6), (8, "eig16_p5", 4), (8, "svd3x2_p3", 4),
                                              (8, "herm16_p3", 3)])
def test_three_and_eight_ranks_equal_single_process(world, name, iters):
    ref, _ = _run(name, iters, None)
    got = _spawn(world, name, iters)
    for r in range(world):
        assert got[r]["out"] == ref, f"rank {r} of {world} diverged from the single-process run"
        assert got[r]["collectives"] == got[0]["collectives"] > 0
        assert got[r]["asked"] == got[0]["asked"]
    steps = [n for n in got[0]["asked"] if n > 0]
    if name in EXTRA:
        assert min(steps) < world, "no step in which some ranks owned no candidate"
    if name.startswith("herm16"):
        # one eigh per matrix, on rank 0 only (the reference: one per candidate step, AMS:161)
        assert [g["eigh_calls"] for g in got] == [1] + [0] * (world - 1)


@pytest.mark.parametrize("world,name,iters", [(2, "eig16", 8), (3, "eig64", 5), (3, "lin24", 6), (8, "eig16_p5", 4)])
def test_one_record_exchange_per_step(world, name, iters):
    """SURVEY 8e / VERDICT r03 item 7: with a stream-independent host side (pert_mode 'none' here; 'mt19937' on the device) a
    sharded direct-solver step is this rank's share of the whole step, ONE all-gather of a 64-byte record per candidate, and the
    row exchange: two collectives per loop body where the phase-by-phase path needs five -- with the bookkeeping, the vectors
    and both RNG streams of the single-process run."""
    kw = dict(pert_mode="none", gmres_compat="rtol")
    ref, _ = _run(name, iters, None, **kw)
    got = _spawn(world, name, iters, kw)
    for r in range(world):
        assert got[r]["out"] == ref, f"rank {r} of {world} diverged from the single-process run"
        assert got[r]["per_body"] == got[0]["per_body"]
    pb = got[0]["per_body"]
    assert min(pb) == 2, pb                    # a loop body without an exceptional branch: record + rows
    assert sum(1 for x in pb if x == 2) >= len(pb) // 2, pb


def _failing_worker(rank, world, port, where, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path[:0] = [ROOT, os.path.join(HERE, "golden"), HERE]
    from adaptive_matrix_solver_amd.dist import PopulationComm, RootFailure
    from adaptive_matrix_solver_amd import solver as sv
    from adaptive_matrix_solver_amd import engine as en
    import scenarios
    from fake_ctx import FakeContext
    comm = PopulationComm("gloo")
    if rank == 0 and where == "diagnosis":
        def boom(self, M):
            raise MemoryError("no room for the (2n)^2 embedding")
        sv.MAUS_Solver._diagnose_matrix_initial = boom
    if rank == 0 and where == "eigh":
        import scipy.linalg as sla
        def boom(*a, **k):
            raise RuntimeError("hipMalloc failed: out of device memory")
        sla.eigh = boom
    what = "finished"
    try:
        spec = scenarios.TRAJECTORIES["herm16"]
        A, _b = scenarios.build(spec)
        np.random.seed(spec["seed"]); random.seed(spec["seed"]); sv.SolutionCandidate._candidate_id_counter = 0
        eng = en.DeviceEngine(ctx=FakeContext(), pert_mode="uniform", gmres_compat="scipy-legacy", comm=comm)
        solver = sv.MAUS_Solver(A, sv.ProblemType.EIGENVALUE, initial_num_candidates=spec["P"], global_convergence_tol=spec["tol"],
                                quiet=True, engine=eng, comm=comm)          # the sharded construction: rank 0 alone diagnoses
        solver.loop_body(1)
    except RootFailure as e:
        what = f"RootFailure: {e}"
    with open(os.path.join(outdir, f"rank{rank}.txt"), "w") as f:
        f.write(what)
    dist.barrier()                                     # every rank is still in step: nobody hangs in a broadcast
    dist.destroy_process_group()


@pytest.mark.parametrize("where", ["diagnosis", "eigh"])
def test_failure_of_rank0_only_work_reaches_every_rank(where):
    """rank 0 alone diagnoses the matrix and decomposes a Hermitian one; an exception there other than LinAlgError
    used to skip the broadcasts the other ranks were already waiting in.  Now every rank raises the same RootFailure."""
    import torch.multiprocessing as mp
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_failing_worker, args=(3, _free_port(), where, d), nprocs=3, join=True)
        got = [open(os.path.join(d, f"rank{r}.txt")).read() for r in range(3)]
    assert all(g.startswith("RootFailure: rank 0 failed") for g in got), got
    assert ("MemoryError" if where == "diagnosis" else "out of device memory") in got[1]
    assert got[0] == got[1] == got[2]


def test_owner_partition_is_contiguous_and_balanced():
    from adaptive_matrix_solver_amd.dist import PopulationComm

    class Dummy(PopulationComm):
        def __init__(self, world):
            self.world = world
            self.rank = 0
    for world in (1, 2, 3, 8):
        for n in (0, 1, 7, 8, 9, 256, 391):
            own = Dummy(world).owners(n)
            assert len(own) == n and list(own) == sorted(own)
            sizes = np.bincount(own, minlength=world)
            assert sizes.max() - sizes.min() <= 1


def _bench(args, env_extra=None, timeout=240):
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=timeout)


def test_bench_gpus_n_starts_n_ranks_itself():
    """`python bench.py --gpus N` outside torchrun: the parent starts N child ranks before touching any GPU and rank 0
    reports n_gpus = N (--launch-check stops after the process group's first all-gather; gloo on CPU here)."""
    r = _bench(["--gpus", "2", "--launch-check"], {"MAUS_DIST_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1, r.stdout                                   # ONE JSON line, from rank 0
    rec = json.loads(line[0])
    assert rec == {"launch_check": True, "n_gpus": 2, "ranks": [0, 1], "backend": "gloo"}
    r8 = _bench(["--gpus", "8", "--launch-check"], {"MAUS_DIST_BACKEND": "gloo"})
    assert r8.returncode == 0, r8.stderr[-2000:]
    rec8 = json.loads([l for l in r8.stdout.splitlines() if l.startswith("{")][0])
    assert rec8["n_gpus"] == 8 and rec8["ranks"] == list(range(8))
    # a single rank needs no launcher
    r1 = _bench(["--gpus", "1", "--launch-check"])
    assert r1.returncode == 0 and json.loads(r1.stdout.strip())["n_gpus"] == 1


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    r = _bench(["--gpus", "4", "--launch-check"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0",
                                                   "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port())})
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


_ID_CHILD = r'''
import os, sys
sys.path.insert(0, %r)
from adaptive_matrix_solver_amd.dist import _exchange_unique_id
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
uid = _exchange_unique_id(rank, world, lambda: bytes(range(128)), timeout=60.0)
assert len(uid) == 128
sys.stdout.write(uid.hex())
'''


@pytest.mark.parametrize("world", [2, 8])
def test_rccl_id_exchange_between_real_processes(world):
    """The 128-byte RCCL id travels from rank 0 to every other rank over a socket next to MASTER_PORT (dist._exchange_unique_id):
    exercised here with real processes and a stand-in id (the RCCL call itself needs one GPU per rank).  A foreign listener on
    the first candidate port must be skipped."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    squatter = socket.create_server(("127.0.0.1", port + 1))          # somebody else already listens where the id server would start
    try:
        procs = []
        for r in range(world):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                       TORCHELASTIC_RUN_ID="idtest")
            procs.append(subprocess.Popen([sys.executable, "-c", _ID_CHILD % ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
        outs = []
        for p in procs:
            out, err = p.communicate(timeout=120)
            assert p.returncode == 0, err[-2000:]
            outs.append(out.strip())
        assert all(o == bytes(range(128)).hex() for o in outs), outs
    finally:
        squatter.close()

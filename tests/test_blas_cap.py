"""Host BLAS pools are kept within the CPUs the process is allowed (engine.cap_blas_threads): an OpenBLAS pool sized for all
visible cores under a container CPU quota stalls the orchestration thread for tens of milliseconds at a time."""
import os

import pytest


def test_cpu_allowance_respects_affinity_and_quota(tmp_path, monkeypatch):
    from adaptive_matrix_solver_amd import engine
    n = engine.cpu_allowance()
    assert 1 <= n <= len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
    except OSError:
        return
    if quota != "max":
        assert n <= max(1, int(quota) // int(period))


@pytest.mark.parametrize("env,expect", [("0", None), ("2", 2), ("auto", "allowance")])
def test_cap_blas_threads(monkeypatch, env, expect):
    threadpoolctl = pytest.importorskip("threadpoolctl")
    import numpy as np                                           # noqa: F401  (loads the BLAS the pools belong to)
    from adaptive_matrix_solver_amd import engine
    pools = threadpoolctl.threadpool_info()
    if not pools:
        pytest.skip("no BLAS / OpenMP pool visible to threadpoolctl")
    before = max(p["num_threads"] for p in pools)
    with threadpoolctl.threadpool_limits(limits=before):         # restores the pools when the test is over
        monkeypatch.setenv("MAUS_BLAS_THREADS", env)
        monkeypatch.setattr(engine, "_blas_cap_done", False)
        engine.cap_blas_threads()
        after = max(p["num_threads"] for p in threadpoolctl.threadpool_info())
        if expect is None:
            assert after == before
        elif expect == "allowance":
            assert after <= max(1, min(before, engine.cpu_allowance()))
        else:
            assert after == min(before, expect) or before < expect
        monkeypatch.setattr(engine, "_blas_cap_done", False)
        engine.cap_blas_threads()                                # idempotent

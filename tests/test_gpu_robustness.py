"""Robustness of the batched LU path (GPU box only): no silent wrong answer, no workspace thrash.

 * The multi-workgroup base panel (small batches) makes the workgroups of one matrix rendezvous once per pivot
   column.  A rendezvous that times out leaves the matrix half factored (info = INT_MIN); the reference contract is that
   every failed solve surfaces (AMS:94-104), so the library must repeat the batch with one workgroup per matrix --
   and never report status 0 for it.  MAUS_PANEL_MW_FORCE_ABORT makes one workgroup skip an arrival.
 * The LU workspace is allocated at most twice per matrix size; at its limit further reserves are no-ops.
 * AMS:243-247 (tiny-sigma convergence of the SVD step with a collapsed right vector) on the device.
"""
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def crand(rng, *shape):
    return rng.standard_normal(shape) + 1j * rng.standard_normal(shape)


def test_panel_rendezvous_timeout_repeats_the_batch(monkeypatch):
    from adaptive_matrix_solver_amd import Context
    n, count = 2048, 4                      # m >= 1024 and count * W <= #CUs: the multi-workgroup panel is in use
    rng = np.random.default_rng(77)
    A = crand(rng, count, n, n) / np.sqrt(n)
    b = crand(rng, count, n)
    ref_ctx = Context(0)
    ref_ctx.set_shared_device(True)         # one workgroup per matrix, by declaration
    try:
        x_ref, st_ref, piv_ref = ref_ctx.lu_solve(A, b, want_ipiv=True)
        assert ref_ctx.lu_mw_aborts() == 0
    finally:
        ref_ctx.close()
    assert (st_ref == 0).all()
    ctx = Context(0)
    try:
        x0, st0, piv0 = ctx.lu_solve(A, b, want_ipiv=True)             # multi-workgroup panel, undisturbed
        assert ctx.lu_mw_aborts() == 0 and (st0 == 0).all()
        assert np.array_equal(piv0, piv_ref) and np.array_equal(x0, x_ref)      # same pivots, same bits
        monkeypatch.setenv("MAUS_PANEL_MW_FORCE_ABORT", "1")
        x1, st1, piv1 = ctx.lu_solve(A, b, want_ipiv=True)             # one arrival never happens -> time-out -> repeat
        assert ctx.lu_mw_aborts() == 1
        assert (st1 == 0).all()
        assert np.array_equal(piv1, piv_ref) and np.array_equal(x1, x_ref)
        # the kernel stays off for the rest of the context's life: no second abort although the hook is still set
        x2, st2 = ctx.lu_solve(A, b)
        assert ctx.lu_mw_aborts() == 1 and np.array_equal(x2, x_ref)
        monkeypatch.delenv("MAUS_PANEL_MW_FORCE_ABORT")
        # residual check against the inputs (the answer is right, not merely reproducible)
        r = np.einsum("gij,gj->gi", A, x1) - b
        assert np.max(np.linalg.norm(r, axis=1) / np.linalg.norm(b, axis=1)) < 1e-10
    finally:
        ctx.close()


def test_panel_timeout_through_the_candidate_step(monkeypatch):
    """The same event inside maus_shifted_lu_solve (the candidate step's entry point)."""
    from adaptive_matrix_solver_amd import Context
    from adaptive_matrix_solver_amd._cabi import PERT_NONE, POP_W, POP_X
    import scenarios
    n, P = 1536, 6
    A = scenarios.ginibre(n, 1536)
    rng = np.random.default_rng(3)
    V = crand(rng, P, n) / np.sqrt(n)
    lam = (rng.standard_normal(P) + 1j * rng.standard_normal(P)) * 0.3
    psi = np.full(P, 1e-20)
    out = []
    for force in (False, True):
        ctx = Context(0)
        try:
            ctx.set_matrix(A)
            ctx.pop_reserve(P)
            ctx.pop_put(POP_X, list(range(P)), V)
            if force:
                monkeypatch.setenv("MAUS_PANEL_MW_FORCE_ABORT", "1")
            st = ctx.shifted_lu_solve(list(range(P)), lam, psi, rhs_mode=0, pert_mode=PERT_NONE)
            if force:
                monkeypatch.delenv("MAUS_PANEL_MW_FORCE_ABORT")
            assert (st == 0).all()
            assert ctx.lu_mw_aborts() == (1 if force else 0)
            out.append(ctx.pop_get(POP_W, list(range(P)), n))
        finally:
            ctx.close()
    assert np.array_equal(out[0], out[1])
    H0 = A - lam[0] * np.eye(n)
    assert np.linalg.norm(H0 @ out[1][0] - V[0]) < 1e-9 * np.linalg.norm(out[1][0])


def test_lu_workspace_limit_sticks(monkeypatch):
    """once the workspace is at its limit (second allocation, cap, or a shrunk allocation) further reserves
    and oversized batches must not free and re-map it (seconds per step): they run in chunks."""
    from adaptive_matrix_solver_amd import Context
    monkeypatch.setenv("MAUS_LU_BATCH", "64")
    ctx = Context(0)
    try:
        n = 96
        assert ctx.lu_reserve(n, 20) == 32 and ctx.lu_workspace_allocations() == 1
        assert ctx.lu_reserve(n, 100) == 64 and ctx.lu_workspace_allocations() == 2      # second allocation: straight to the cap
        for want in (65, 200, 1000, 64, 3):
            assert ctx.lu_reserve(n, want) == 64
        assert ctx.lu_workspace_allocations() == 2
        rng = np.random.default_rng(1)
        A = crand(rng, 150, n, n)
        b = crand(rng, 150, n)
        x, st = ctx.lu_solve(A, b)                                                     # 150 systems through 64 slots
        assert (st == 0).all() and ctx.lu_workspace_allocations() == 2
        assert np.max(np.abs(np.einsum("gij,gj->gi", A, x) - b)) < 1e-9
        # a different matrix size starts over (and may allocate twice again)
        assert ctx.lu_reserve(200, 10) == 32 and ctx.lu_workspace_allocations() == 3
    finally:
        ctx.close()


def test_closed_context_and_stale_history_are_refused():
    from adaptive_matrix_solver_amd import Context
    from adaptive_matrix_solver_amd._cabi import POP_X, MausHipError
    from adaptive_matrix_solver_amd.solver import _HistRef
    rng = np.random.default_rng(2)
    ctx = Context(0)
    ctx.set_matrix(crand(rng, 600, 600))
    ctx.pop_reserve(4)
    X = crand(rng, 4, 600)
    ctx.pop_put(POP_X, [0, 1, 2, 3], X)
    first = ctx.hist_append(POP_X, [2, 3], 600)
    ref = _HistRef(ctx, 1.5, ((first + 1, 600),))
    got = ref.resolve()
    assert got[0] == 1.5 and np.array_equal(got[1], X[3])
    ctx.set_matrix(crand(rng, 700, 700))                 # another vector length: the store is dropped
    with pytest.raises(RuntimeError, match="history store"):
        ref.resolve()
    ctx.close()
    with pytest.raises(MausHipError, match="closed"):
        ctx.hist_get([0], 600)
    with pytest.raises(MausHipError, match="closed"):
        ctx.sync()


def test_svd_tiny_sigma_branch_on_the_device():
    """AMS:243-247.  A = 1e-10 * unitary: sigma < 1e-8 for every candidate, and ||A v||, ||A^H u|| land on either side of
    the 1e-10 tests of AMS:235/242 by rounding -- which side is decided by the last bit of a norm, so the device cannot be
    asked to take the oracle's side candidate by candidate.  What is checked: every candidate's outcome is the reference's
    rule applied to the norms the DEVICE computed (captured from the step's own maus_svd_power_propose calls), and the
    AMS:247 replacement actually occurs."""
    import scenarios
    from adaptive_matrix_solver_amd.solver import MAUS_Solver, ProblemType, SolutionCandidate
    S = SolutionCandidate.State
    n = 6
    A, _ = scenarios.build(scenarios.TRAJECTORIES["svdtiny"])
    np.random.seed(1); random.seed(1); SolutionCandidate._candidate_id_counter = 0
    solver = MAUS_Solver(A, ProblemType.SVD, initial_num_candidates=64, global_convergence_tol=1e-8, quiet=True)
    ctx = solver.engine.ctx
    calls = []
    real = ctx.svd_power_propose

    def spy(slots):
        norms = real(slots)
        calls.append((list(slots), norms.copy()))
        return norms
    ctx.svd_power_propose = spy
    ones = np.ones(n, dtype=np.complex128) / np.sqrt(n)
    fired = collapsed = 0
    for it in range(3):
        solver._update_global_diagnostics(it + 1)
        solver._adjust_global_strategy(it + 1)
        active = [c for c in solver.candidates if c.state not in (S.CONVERGED, S.RETIRED)]
        before = {c.id: (c.stuck_counter, c.num_resets) for c in active}
        calls.clear()
        solver.step_population()
        # the LAST device call that covered a candidate's slot as part of the accepted prefix decides its fate; the
        # engine accepts a run up to its first exceptional candidate, handles that one alone and restarts behind it
        last = {}
        for slots, norms in calls:
            for s, nr in zip(slots, norms):
                last[s] = nr
        for c in active:
            nr = last[c._slot]
            exceptional = nr[0] < 1e-10 or nr[2] < 1e-10
            if exceptional:
                collapsed += 1
                assert c.num_resets == before[c.id][1] + 1          # (AMS:309/312 may overwrite STUCK afterwards)
                continue
            sigma = max(nr[1], nr[3])
            assert sigma < 1e-8 and c.state == S.CONVERGED and c.stuck_counter == 0
            # u = t / (sigma1 if sigma1 > 1e-10 else 1), v = s / (sigma2 if sigma2 > 1e-10 else 1)   (AMS:235, 242)
            if nr[3] < 1e-10:
                fired += 1
                assert np.array_equal(c.right_v_k, ones)                       # AMS:247
            else:
                want = 1.0 if nr[3] > 1e-10 else nr[3]
                assert abs(np.linalg.norm(c.right_v_k) - want) <= 1e-12 * want
            want = 1.0 if nr[1] > 1e-10 else nr[1]                              # (||u|| < 1e-10 left through AMS:236-239)
            assert abs(np.linalg.norm(c.u_k) - want) <= 1e-12 * want
        solver._manage_candidates(it + 1)
    assert fired > 0, "AMS:247 never reached on the device: change the scale / population"
    assert collapsed >= 0

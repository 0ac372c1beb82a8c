"""The structure-of-arrays candidate store (adaptive_matrix_solver_amd/candstore.py) behind SolutionCandidate's attributes:
the attributes keep the reference's values AND scalar types (AMS:113-126, 295-331), objects a caller assigns come back as
they were, a reused slot starts from the constructor's state, and the histories that the batched step logs once per step
(n > 512) replay to exactly what the per-candidate appends of AMS:303-304 record.  CPU only (tests/fake_ctx.py)."""
import gc
import random
from fractions import Fraction

import numpy as np
import pytest

import scenarios
from fake_ctx import FakeContext


def _solver(name, lazy=False):
    """(solver, spec, undo): lazy=True makes every candidate constructed from now on take the device-backed history path of
    problems above n = 512 (undo() restores the constructor)."""
    from adaptive_matrix_solver_amd.engine import DeviceEngine
    from adaptive_matrix_solver_amd.solver import MAUS_Solver, ProblemType, SolutionCandidate
    spec = scenarios.TRAJECTORIES[name]
    A, b = scenarios.build(spec)
    np.random.seed(spec["seed"]); random.seed(spec["seed"]); SolutionCandidate._candidate_id_counter = 0
    PT = {"eig": ProblemType.EIGENVALUE, "lin": ProblemType.SOLVE_LINEAR_SYSTEM, "svd": ProblemType.SVD}[spec["kind"]]
    eng = DeviceEngine(ctx=FakeContext(), pert_mode="uniform", gmres_compat="scipy-legacy")
    undo = lambda: None
    if lazy:
        orig = SolutionCandidate.__init__

        def init(self, *a, **k):
            orig(self, *a, **k)
            self._lazy_hist = True
        SolutionCandidate.__init__ = init

        def undo():
            SolutionCandidate.__init__ = orig
    try:
        s = MAUS_Solver(A, PT, b_vector=b, initial_num_candidates=spec["P"], global_convergence_tol=spec["tol"],
                        quiet=True, engine=eng, record_history=True)
    except BaseException:
        undo()
        raise
    return s, spec, undo


def test_attribute_types_of_a_fresh_candidate():
    from adaptive_matrix_solver_amd.solver import SolutionCandidate
    s, spec, _ = _solver("eig16")
    c = s.candidates[0]
    assert type(c.residual_k) is float and c.residual_k == float("inf")
    assert type(c.prev_residual) is float and c.prev_residual == float("inf")
    assert type(c.alpha_local_step) is np.complex128 and c.alpha_local_step == 0.01
    assert type(c.w_k) is float and c.w_k == 0.01
    assert type(c.lambda_k) is complex                      # AMS:137: random.random() arithmetic
    assert c.sigma_k is None and c.b_vector is None
    assert c.state is SolutionCandidate.State.EXPLORING
    assert (c.stuck_counter, c.local_psi_retries_needed, c.num_resets) == (0, 0, 0)
    assert type(c.stuck_counter) is int
    assert c.residual_history == [float("inf")] and len(c.param_history) == 1
    s2, _, _ = _solver("svd5x4")
    c2 = s2.candidates[0]
    assert type(c2.sigma_k) is float and c2.sigma_k == 1.0 and c2.lambda_k is None


def test_assigned_objects_come_back_and_feed_the_array_code():
    s, spec, _ = _solver("eig16")
    c, d = s.candidates[0], s.candidates[1]
    st = c._st
    w = Fraction(1, 3)
    c.w_k = w
    assert c.w_k is w and st.w[c._slot] == float(w)
    lam = np.float64(2.5)                                    # what the Hermitian shortcut assigns (AMS:166: eigh's real eigenvalue)
    c.lambda_k = lam
    assert c.lambda_k is lam and st.lam[c._slot] == 2.5 + 0j
    c.alpha_local_step = 0.25                                # a clamp's Python float (AMS:308)
    assert type(c.alpha_local_step) is float and st.alpha[c._slot] == 0.25
    c.alpha_local_step = c.alpha_local_step * np.complex128(1.0)
    assert type(c.alpha_local_step) is np.complex128
    c.alpha_local_step = 3                                   # something else entirely: kept as it is
    assert c.alpha_local_step == 3 and type(c.alpha_local_step) is int and st.alpha[c._slot] == 3.0
    c.residual_k = np.float32(0.5)
    assert type(c.residual_k) is np.float32 and st.res[c._slot] == 0.5
    c.stuck_counter += 2
    assert c.stuck_counter == 2 and d.stuck_counter == 0     # neighbours are untouched
    c.state = type(c).State.STUCK
    assert c.state is type(c).State.STUCK and d.state is type(c).State.EXPLORING
    bvec = np.arange(3.0)
    c.b_vector = bvec
    assert c.b_vector is bvec


def test_a_reused_slot_starts_from_the_constructor_state():
    s, spec, _ = _solver("eig16")
    c = s.candidates.pop()
    slot = c._slot
    c.residual_k = np.float64(1e-3); c.stuck_counter = 5; c.w_k = 0.9; c.state = type(c).State.RETIRED
    s._view_cache = None
    del c
    gc.collect()
    n = s._new_candidate()
    assert n._slot == slot                                   # the freed slot is handed out again
    assert n.residual_k == float("inf") and n.stuck_counter == 0 and n.w_k == 0.01
    assert n.state is type(n).State.EXPLORING and n.residual_history == [float("inf")]


@pytest.mark.parametrize("name,iters", [("svd5x4", 8), ("eig16", 6), ("lin24", 5)])
def test_logged_histories_replay_to_the_eager_ones(name, iters):
    """The same trajectory twice: histories appended per candidate (n <= 512) and histories logged once per step and
    replayed on access (the path of n > 512).  Same lengths, same residual objects' values and types, same iterates."""
    from adaptive_matrix_solver_amd.solver import SolutionCandidate
    runs = []
    for lazy in (False, True):
        s, spec, undo = _solver(name, lazy=lazy)
        try:
            for it in range(iters):
                s.loop_body(it)
        finally:
            undo()
        runs.append(s)
    eager, lazy = runs
    assert [c.id for c in eager.candidates] == [c.id for c in lazy.candidates]
    assert len(lazy.candidates[0]._st.hist_log) > 0 and len(eager.candidates[0]._st.hist_log) == 0
    for ce, cl in zip(eager.candidates, lazy.candidates):
        assert len(cl._rh) < len(ce._rh) or len(ce._rh) == 1      # nothing was appended to the lazy candidate during the steps
        re_, rl = ce.residual_history, cl.residual_history
        assert len(re_) == len(rl) == len(ce.param_history) == len(cl.param_history)
        for a, b in zip(re_, rl):
            assert type(a) is type(b) and (a == b or (a != a and b != b))
        for pe, pl in zip(ce.param_history, cl.param_history):
            assert len(pe) == len(pl)
            for a, b in zip(pe, pl):
                if isinstance(a, np.ndarray):
                    assert np.array_equal(a, b)
                else:
                    assert type(a) is type(b) and a == b


def test_population_view_follows_the_candidate_list():
    """The solver keeps the slot array of self.candidates between calls; replacing the list, appending to it or
    shortening it is noticed."""
    s, spec, _ = _solver("eig16")
    st, slots, code = s._pop_view()
    assert list(slots) == [c._slot for c in s.candidates]
    s.candidates.append(s._new_candidate())
    assert list(s._pop_view()[1]) == [c._slot for c in s.candidates]
    s.candidates = list(reversed(s.candidates))
    assert list(s._pop_view()[1]) == [c._slot for c in s.candidates]
    s.candidates.pop(0)
    assert list(s._pop_view()[1]) == [c._slot for c in s.candidates]


def SolutionCandidateState():
    from adaptive_matrix_solver_amd.solver import SolutionCandidate
    return SolutionCandidate.State


def test_spawns_reach_the_device_in_one_transfer_and_the_gram_block_is_reused(monkeypatch):
    """AMS:533-549 constructs up to 15 candidates per loop body; their vectors are pushed together.  The Gram block of
    _manage_candidates serves the diagnostics of the next iteration (its CONVERGED set is a subset), and a host-side edit of a
    converged vector drops it."""
    s, spec, _ = _solver("herm16")                             # Hermitian: the shortcut converges every stepped candidate (AMS:155-181)
    ctx = s.engine.ctx
    puts, grams = [], []
    real_put, real_gram = ctx.pop_put, ctx.gram

    def put(which, slots, vecs):
        puts.append(len(list(slots)))
        return real_put(which, slots, vecs)

    def gram(which, slots, length):
        grams.append(len(list(slots)))
        return real_gram(which, slots, length)
    s.gram_min = 2
    CONV = SolutionCandidateState().CONVERGED
    it = 0
    while sum(c.state is CONV for c in s.candidates) < 3:      # loop bodies until a few candidates have converged
        it += 1
        assert it < 10
        s.loop_body(it)
    s._update_global_diagnostics(it + 1); s._adjust_global_strategy(it + 1); s.step_population()
    monkeypatch.setattr(ctx, "pop_put", put)
    monkeypatch.setattr(ctx, "gram", gram)
    ids_before = {c.id for c in s.candidates}
    s._manage_candidates(it + 1)
    spawned = [c for c in s.candidates if c.id not in ids_before]
    assert spawned and puts == [len(spawned)]                  # one transfer for all of them
    assert all(c._dev_valid for c in spawned)
    assert len(grams) == 1                                      # the block of the converged set ...
    s._update_global_diagnostics(it + 2)
    assert len(grams) == 1                                      # ... serves the next diagnostics (a subset of it)
    conv = [c for c in s.candidates if c.state is type(c).State.CONVERGED]
    conv[0].v_k = conv[0].v_k.copy()                            # a host-side edit: pushed again, the block is recomputed
    s._update_global_diagnostics(it + 3)
    assert len(grams) == 2

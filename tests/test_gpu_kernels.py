"""Kernel-level parity tests through the C ABI (GPU box only): the MFMA zgemm and the
batched LU solve against NumPy / SciPy on the same inputs."""
import numpy as np
import pytest
import scipy.linalg as sla

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from adaptive_matrix_solver_amd import Context
    c = Context(0)
    yield c
    c.close()


def crand(rng, *shape):
    return rng.standard_normal(shape) + 1j * rng.standard_normal(shape)


@pytest.mark.parametrize("M,N,K", [(16, 16, 4), (64, 64, 16), (5, 7, 3), (100, 130, 77), (256, 64, 512),
                                   (129, 65, 33), (300, 4096, 64), (1024, 1024, 256),
                                   # the LDS-DMA staged 3M kernel: 64 x 64 tiles (M, N >= 1536) with ragged edges and a K that
                                   # is a multiple of 8 only; 64 x 32 tiles with a ragged N
                                   (1600, 1700, 72), (1990, 1540, 128), (200, 1111, 200)])
@pytest.mark.parametrize("b_layout", [0, 1])
def test_zgemm_matches_numpy(ctx, M, N, K, b_layout):
    rng = np.random.default_rng(M * 1000 + N + K)
    A = crand(rng, M, K)
    B = crand(rng, N, K) if b_layout else crand(rng, K, N)
    Bm = B.T if b_layout else B
    C = ctx.zgemm(A, B, b_layout=b_layout)
    ref = A @ Bm
    scale = np.abs(A) @ np.abs(Bm)
    assert np.max(np.abs(C - ref) / scale) < 4e-16 * max(4, np.sqrt(K))


def test_zgemm_conj_alpha_beta(ctx):
    rng = np.random.default_rng(5)
    M, N, K = 96, 80, 48
    A, B, C0 = crand(rng, M, K), crand(rng, K, N), crand(rng, M, N)
    for ca in (False, True):
        for cb in (False, True):
            C = ctx.zgemm(A, B, C_in=C0, conj_a=ca, conj_b=cb, alpha=-1.0, beta=1)
            ref = C0 - (A.conj() if ca else A) @ (B.conj() if cb else B)
            assert np.max(np.abs(C - ref)) < 1e-12
    # asymmetric operand check of the MFMA lane maps: A = I must return B exactly
    I = np.eye(64, dtype=np.complex128)
    Bq = (np.arange(64 * 64).reshape(64, 64) % 97 + 1j * (np.arange(64 * 64).reshape(64, 64) % 89)).astype(np.complex128)
    assert np.array_equal(ctx.zgemm(I, Bq), Bq)
    assert np.array_equal(ctx.zgemm(Bq, I), Bq)


@pytest.mark.parametrize("n,count", [(5, 3), (8, 2), (31, 2), (32, 4), (33, 2), (64, 5), (96, 3), (100, 2),
                                     (256, 3), (300, 2), (512, 2), (1024, 2)])
def test_lu_solve_matches_scipy(ctx, n, count):
    rng = np.random.default_rng(n)
    A = crand(rng, count, n, n)
    b = crand(rng, count, n)
    x, status, ipiv = ctx.lu_solve(A, b, want_ipiv=True)
    assert np.all(status == 0)
    for g in range(count):
        ref = sla.solve(A[g], b[g])
        lu, piv = sla.lu_factor(A[g])
        assert np.array_equal(ipiv[g], piv), f"pivot sequence differs (n={n}, g={g})"
        err = np.linalg.norm(x[g] - ref) / np.linalg.norm(ref)
        assert err < 1e-13 * np.linalg.cond(A[g]), (n, g, err)
        assert np.linalg.norm(A[g] @ x[g] - b[g]) / np.linalg.norm(b[g]) < 1e-11


def test_lu_status_codes(ctx):
    rng = np.random.default_rng(1)
    n = 40
    A = crand(rng, 3, n, n)
    b = crand(rng, 3, n)
    A[0][:, 7] = 0.0                      # exactly singular: zero pivot at (1-based) column 8
    A[1][3, 4] = np.nan                   # non-finite input
    x, status = ctx.lu_solve(A, b)
    assert status[0] == 8
    assert status[1] == -1
    assert status[2] == 0
    with pytest.raises(np.linalg.LinAlgError):
        sla.solve(A[0], b[0])


def test_lu_tie_breaking_first_index(ctx):
    # equal |re|+|im| candidates: LAPACK izamax keeps the first
    n = 6
    A = np.eye(n, dtype=np.complex128)
    A[:, 0] = [1, 1j, -1, 0.5 + 0.5j, 1, 0]
    A += np.triu(np.ones((n, n)), 1) * 0.1
    b = np.arange(1, n + 1).astype(np.complex128)
    x, status, ipiv = ctx.lu_solve(A, b, want_ipiv=True)
    lu, piv = sla.lu_factor(A)
    assert status[0] == 0 and np.array_equal(ipiv[0], piv)
    assert np.allclose(x[0], sla.solve(A, b), rtol=1e-13)


@pytest.mark.parametrize("count,n", [(1, 7), (5, 33), (40, 300), (130, 1024)])
def test_gram_block_matches_numpy(ctx, count, n):
    """maus_gram: G[i, j] = np.vdot(x_i, x_j) over scattered population slots (SURVEY f-2)."""
    rng = np.random.default_rng(count * 1000 + n)
    A = crand(rng, n, n)
    ctx.set_matrix(A)
    ctx.pop_reserve(2 * count + 3)
    slots = list(rng.permutation(2 * count + 3)[:count])
    X = crand(rng, count, n)
    ctx.pop_put(0, slots, X)
    G = ctx.gram(0, slots, n)
    ref = X.conj() @ X.T
    assert np.abs(G - ref).max() <= 4e-16 * np.sqrt(n) * (np.abs(X) @ np.abs(X).T).max()
    assert np.allclose(np.diag(G).imag, 0.0, atol=1e-13 * n)


def test_empty_and_minimal_inputs(ctx):
    """count = 0 is a no-op for every batched entry point; n = 1 and n = 2 work (padding to 32, panels of height 32)."""
    from adaptive_matrix_solver_amd._cabi import KIND_EIG, PERT_NONE
    rng = np.random.default_rng(0)
    for n in (1, 2):
        A = crand(rng, n, n) + 3.0 * np.eye(n)
        ctx.set_matrix(A)
        ctx.pop_reserve(4)
        V = crand(rng, 3, n)
        ctx.pop_put(0, [0, 1, 2], V)
        empty = np.zeros(0, dtype=np.int32)
        num, den = ctx.matvec_rayleigh(empty)
        assert num.shape == (0,) and den.shape == (0,)
        assert ctx.shifted_lu_solve(empty, np.zeros(0, complex), np.zeros(0)).shape == (0,)
        assert ctx.relax_normalise(empty, np.zeros(0, complex)).shape == (0,)
        res, fin = ctx.residual(KIND_EIG, empty, np.zeros(0, complex))
        assert res.shape == (0,) and fin.shape == (0,)
        assert ctx.gram(0, empty, n).shape == (0, 0)
        lam = np.array([0.1, -0.2j, 0.3 + 0.1j])
        st = ctx.shifted_lu_solve([0, 1, 2], lam, np.zeros(3), 0, PERT_NONE)
        assert (st == 0).all()
        W = ctx.pop_get(2, [0, 1, 2], n)
        for k in range(3):
            ref = np.linalg.solve(A - lam[k] * np.eye(n), V[k])
            assert np.allclose(W[k], ref, rtol=1e-13, atol=1e-14)
        num, den = ctx.matvec_rayleigh([0, 1, 2])
        assert np.allclose(num, np.einsum("ki,ki->k", V.conj(), V @ A.T)) and np.allclose(den, np.einsum("ki,ki->k", V.conj(), V))


def test_population_product_rows_do_not_depend_on_the_batch(ctx):
    """A row of Y = X A^T has the same bits whether it is computed among 1600 rows (64 x 64 tiles of the DMA 3M kernel), among
    40 (64 x 32 tiles) or among 33: per-element arithmetic does not depend on the tile shape.  maus_svd_power_propose relies on it
    when it multiplies only the rows whose product is not at hand (AMS:228 / 295-298)."""
    from adaptive_matrix_solver_amd._cabi import POP_X
    POP_Y = 3
    rng = np.random.default_rng(7)
    n, P = 1536, 1600
    A = ((rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))) / np.sqrt(n)).astype(np.complex128)
    X = (rng.standard_normal((P, n)) + 1j * rng.standard_normal((P, n))).astype(np.complex128)
    ctx.set_matrix(A)
    ctx.pop_reserve(P)
    ctx.pop_put(POP_X, np.arange(P), X)
    ctx.matvec_rayleigh(np.arange(P))
    sel = np.array([5, 77, 640, 641, 1203, 1599] + list(range(100, 134)))          # 40 rows
    big = ctx.pop_get(POP_Y, sel, n)
    assert np.abs(big - X[sel] @ A.T).max() <= 1e-11
    from adaptive_matrix_solver_amd._cabi import POP_W
    for m in (40, 33):
        ctx.pop_copy(POP_W, POP_X, [])                    # drops the library's "Y = A X" stamps: the product below is really run
        ctx.pop_put(POP_Y, sel[:m], np.zeros((m, n), dtype=np.complex128))
        ctx.matvec_rayleigh(sel[:m])
        assert np.array_equal(ctx.pop_get(POP_Y, sel[:m], n), big[:m]), m
    # ... and with the stamps in place it is not: the rows come back as they are
    marker = np.full((3, n), 7.0 + 1.0j)
    ctx.pop_put(POP_W, sel[:3], marker)                   # (W is not X or Y: the stamps stay)
    ctx.matvec_rayleigh(sel[:33])
    assert np.array_equal(ctx.pop_get(POP_Y, sel[:33], n), big[:33])


def test_svd_power_step_reuses_the_residuals_product_bit_for_bit():
    """Two contexts walk the same SVD loop bodies (propose / commit / residual, AMS:227-255 + 295-298); in one of them every
    product is recomputed (the stamps that say 'Y = A X for this row' / 'S = A^H U for this row' are dropped before each propose
    and each residual), in the other the power step multiplies only the rows that changed since the residual -- here a few
    re-seeded rows per body, like the spawns of AMS:533-549 -- and the residual takes A^H u from the power step.  Norms,
    residuals and vectors must agree bit for bit."""
    from adaptive_matrix_solver_amd import Context
    from adaptive_matrix_solver_amd._cabi import KIND_SVD, POP_U, POP_W, POP_X
    rng = np.random.default_rng(11)
    rows, cols, P = 384, 320, 200
    A = ((rng.standard_normal((rows, cols)) + 1j * rng.standard_normal((rows, cols))) / 16).astype(np.complex128)
    V0 = (rng.standard_normal((P, cols)) + 1j * rng.standard_normal((P, cols))).astype(np.complex128)
    U0 = (rng.standard_normal((P, rows)) + 1j * rng.standard_normal((P, rows))).astype(np.complex128)
    fresh = [(rng.standard_normal((7, cols)) + 1j * rng.standard_normal((7, cols))).astype(np.complex128) for _ in range(4)]
    out = []
    for reuse in (False, True):
        c = Context(0)
        try:
            c.set_matrix(A)
            c.pop_reserve(P)
            sl = np.arange(P)
            c.pop_put(POP_X, sl, V0); c.pop_put(POP_U, sl, U0)
            rec = []
            for body in range(4):
                if not reuse:
                    c.pop_copy(POP_W, POP_X, [])                  # any writing entry point drops the stamps (no row is copied)
                norms = c.svd_power_propose(sl)
                c.svd_commit(sl)
                sig = np.maximum(norms[:, 1], norms[:, 3]).astype(np.complex128)
                if not reuse:
                    c.pop_copy(POP_W, POP_X, [])                  # ... and the stamps of the power step's own A^H u
                res, fin = c.residual(KIND_SVD, sl, sig)
                rec.append((norms.copy(), np.asarray(res).copy()))
                c.pop_put(POP_X, np.arange(3 + body, 3 + body + 7), fresh[body])     # seven rows change before the next body
            rec.append((c.pop_get(POP_X, sl, cols), c.pop_get(POP_U, sl, rows)))
            out.append(rec)
        finally:
            c.close()
    for a, b in zip(out[0], out[1]):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_kept_products_never_mix_kernel_families():
    """A product over up to 32 rows runs other kernels (another rounding) than one over more: the rows it leaves in Y must not
    be taken for the rows of a later, larger product.  Residual of 30 candidates, then the Rayleigh quotients of those 30 plus
    10 new ones: bit for bit what a context without any kept product computes."""
    from adaptive_matrix_solver_amd import Context
    from adaptive_matrix_solver_amd._cabi import KIND_EIG, POP_X
    rng = np.random.default_rng(5)
    n, P = 256, 40
    A = ((rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))) / np.sqrt(n)).astype(np.complex128)
    X = (rng.standard_normal((P, n)) + 1j * rng.standard_normal((P, n))).astype(np.complex128)
    lam = (rng.standard_normal(P) + 1j * rng.standard_normal(P)).astype(np.complex128)
    out = []
    for warm in (False, True):
        c = Context(0)
        try:
            c.set_matrix(A); c.pop_reserve(P); c.pop_put(POP_X, np.arange(P), X)
            if warm:
                c.residual(KIND_EIG, np.arange(30), lam[:30])         # 30 rows: the small-batch kernels
            out.append(c.matvec_rayleigh(np.arange(P)))
        finally:
            c.close()
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])

"""Host <-> device transfers go through a pinned buffer of the context, never straight from / into the caller's arrays
(csrc/capi.hip, "pinned staging"): large ones in two halves that alternate, small uploads through a pinned ring.  With a
1 MiB buffer every transfer below runs over many chunks -- odd and even counts, exact multiples of a half, a last partial
chunk -- and must come back bit for bit.  (GPU box only.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def crand(rng, *shape):
    return rng.standard_normal(shape) + 1j * rng.standard_normal(shape)


@pytest.fixture
def small_pin(monkeypatch):
    monkeypatch.setenv("MAUS_PIN_BYTES", str(1 << 20))
    from adaptive_matrix_solver_amd import Context
    ctx = Context(0)
    yield ctx
    ctx.close()


@pytest.mark.parametrize("n", [181, 256, 362, 600])          # 0.5 / 1.0 (one buffer exactly) / 2.0 / 5.5 MiB of complex128
def test_matrix_sized_transfers_round_trip(small_pin, n):
    ctx = small_pin
    rng = np.random.default_rng(n)
    A = crand(rng, n, n)
    V = crand(rng, n, n)
    ctx.set_matrix(A)
    ctx.set_eigvecs(V)
    assert np.array_equal(ctx.get_eigvecs(), V)
    # the matrix itself is not readable back; A @ x through the population product tells whether every chunk landed
    ctx.pop_reserve(8)
    X = crand(rng, 8, n)
    ctx.pop_put(0, list(range(8)), X)
    num, den = ctx.matvec_rayleigh(list(range(8)))
    want = np.einsum("pi,ij,pj->p", X.conj(), A, X)
    assert np.allclose(num, want, rtol=1e-11, atol=1e-9)
    assert np.allclose(den, np.einsum("pi,pi->p", X.conj(), X), rtol=1e-12)


def test_population_rows_in_chunks(small_pin):
    ctx = small_pin
    n, P = 1000, 300                                           # 16 000 B per row: 65 rows per 1 MiB buffer
    rng = np.random.default_rng(1)
    ctx.set_matrix(crand(rng, n, n))
    ctx.pop_reserve(P + 20)
    X = crand(rng, P, n)
    perm = rng.permutation(P + 20)[:P]                          # scattered slots
    ctx.pop_put(0, list(perm), X)
    assert np.array_equal(ctx.pop_get(0, list(perm), n), X)
    back = ctx.pop_get(0, list(range(P + 20)), n)               # contiguous read-back of everything
    assert np.array_equal(back[perm], X)
    for sl in ([5], [5, 6], [7, 3], [9, 2, 4], list(range(130)), list(perm[:131])):
        got = ctx.pop_get(0, sl, n - 7)                         # shorter than the row
        assert np.array_equal(got, back[sl, :n - 7])
    ctx.pop_put(0, [11, 12, 13], X[:3, :500])                   # partial rows: the tail of the row stays
    got = ctx.pop_get(0, [11, 12, 13], n)
    assert np.array_equal(got[:, :500], X[:3, :500]) and np.array_equal(got[:, 500:], back[[11, 12, 13], 500:])


def test_history_gram_and_host_solves_through_the_small_buffer(small_pin):
    ctx = small_pin
    n, P = 700, 90
    rng = np.random.default_rng(2)
    A = crand(rng, n, n) / np.sqrt(n)
    ctx.set_matrix(A)
    ctx.pop_reserve(P)
    X = crand(rng, P, n)
    ctx.pop_put(0, list(range(P)), X)
    first = ctx.hist_append(0, list(range(P)), n)
    idx = [first + k for k in (0, 17, 89, 3)]
    assert np.array_equal(ctx.hist_get(idx, n), X[[0, 17, 89, 3]])
    G = ctx.gram(0, list(range(P)), n)                          # 90 x 90 complex = 130 KB: staged
    assert np.allclose(G, X.conj() @ X.T, rtol=1e-11, atol=1e-9)
    B = crand(rng, 12, n)
    Hs = np.stack([A - (0.3 + 0.1j * k) * np.eye(n) for k in range(12)])       # 12 x 7.8 MB up, 12 x 11 KB back
    x, st = ctx.lu_solve(Hs, B)
    assert (st == 0).all()
    assert np.max(np.abs(np.einsum("gij,gj->gi", Hs, x) - B)) < 1e-9
    Ag, Bg = crand(rng, 300, 200), crand(rng, 200, 260)
    assert np.allclose(ctx.zgemm(Ag, Bg), Ag @ Bg, rtol=1e-11, atol=1e-9)

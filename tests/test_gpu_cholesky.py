"""svgp_potrf_batched / svgp_trsm_batched / svgp_potri_batched (north_star's "Cholesky of K_mm and the triangular
solves"; tf.linalg.cholesky + log diag_part of SVGPVAE_model.py:270-274) against torch.linalg on the same inputs,
m = 32 ... 2048 incl. sizes that are not multiples of the 64-block and the K + jitter spectrum the GP block factors."""
import pytest
import torch

from svgp_vae_amd import _lib

pytestmark = pytest.mark.gpu
DT = torch.float64


def _spd(m, batch, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    X = torch.randn(batch, m, m + 8, dtype=DT, device="cuda", generator=g)
    return X @ X.transpose(1, 2) / m + 0.05 * torch.eye(m, dtype=DT, device="cuda")


def _kernel_like(m, jitter, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    Q, _ = torch.linalg.qr(torch.randn(m, m, dtype=DT, device="cuda", generator=g))
    lam = 50 * torch.exp(-torch.arange(m, dtype=DT, device="cuda") / 8)
    K = (Q * lam) @ Q.T
    return 0.5 * (K + K.T), Q, lam, K + jitter * torch.eye(m, dtype=DT, device="cuda")


def _potrf(A):
    lib = _lib.load_library()
    batch, m = A.shape[0], A.shape[1]
    Lf = A.clone()
    logdet = torch.full((batch,), float("nan"), dtype=DT, device="cuda")
    work = torch.full((lib.svgp_potrf_workspace_elems(m, batch),), float("nan"), dtype=DT, device="cuda")   # poisoned: nothing may be read unwritten
    _lib.call("svgp_potrf_batched", m, batch, Lf.data_ptr(), m, m * m, logdet.data_ptr(), work.data_ptr(),
              torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return Lf, logdet, work


@pytest.mark.parametrize("m,batch", [(1, 2), (32, 1), (64, 3), (65, 2), (100, 1), (128, 2), (130, 3), (256, 17), (513, 2), (800, 3),
                                     (2048, 1)])
def test_potrf_potri_against_torch(m, batch):
    A = _spd(m, batch, m)
    Lf, logdet, work = _potrf(A)
    want = torch.linalg.cholesky(A)
    assert float((Lf - want).abs().max() / want.abs().max()) < 1e-11
    assert float(torch.triu(Lf, 1).abs().max()) == 0.0
    assert float((logdet - torch.linalg.slogdet(A)[1]).abs().max()) < 1e-10 * max(m, 8)
    lib = _lib.load_library()
    w2 = torch.full((lib.svgp_potri_workspace_elems(m, batch),), float("nan"), dtype=DT, device="cuda")
    nb = (m + 63) // 64
    for blocks in (work[:batch * nb * 4096], None):             # diagonal-block inverses from potrf / recomputed
        inv = Lf.clone()
        _lib.call("svgp_potri_batched", m, batch, inv.data_ptr(), None if blocks is None else blocks.data_ptr(), w2.data_ptr(),
                  torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        wi = torch.linalg.inv(A)
        assert float((inv - wi).abs().max() / wi.abs().max()) < 1e-9
        assert float((inv - inv.transpose(1, 2)).abs().max()) <= 1e-12 * float(wi.abs().max())
        eye = torch.eye(m, dtype=DT, device="cuda")
        assert float((inv @ A - eye).abs().max()) < 1e-8


def test_potrf_strided_and_padded_leading_dimension():
    """lda > m and a batch stride larger than the matrix (sub-matrices of a bigger buffer)."""
    m, batch, lda = 150, 3, 160
    A = _spd(m, batch, 9)
    buf = torch.full((batch, m + 5, lda), 7.0, dtype=DT, device="cuda")
    buf[:, :m, :m] = A
    logdet = torch.zeros(batch, dtype=DT, device="cuda")
    lib = _lib.load_library()
    work = torch.full((lib.svgp_potrf_workspace_elems(m, batch),), float("nan"), dtype=DT, device="cuda")   # poisoned: nothing may be read unwritten
    _lib.call("svgp_potrf_batched", m, batch, buf.data_ptr(), lda, (m + 5) * lda, logdet.data_ptr(), work.data_ptr(),
              torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert float((buf[:, :m, :m] - torch.linalg.cholesky(A)).abs().max()) < 1e-11
    assert float((buf[:, m:, :] - 7.0).abs().max()) == 0.0 and float((buf[:, :m, m:] - 7.0).abs().max()) == 0.0


@pytest.mark.parametrize("side,trans", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("m,n,batch,shared", [(32, 5, 2, False), (100, 64, 3, False), (256, 1, 16, True), (300, 700, 2, True),
                                              (800, 33, 2, False), (2048, 16, 1, False)])
def test_trsm_against_torch(side, trans, m, n, batch, shared):
    lib = _lib.load_library()
    A = _spd(m, 1 if shared else batch, m + n)
    Lt = torch.linalg.cholesky(A)
    # garbage above the diagonal must be ignored; .contiguous(): torch.linalg.cholesky returns column-major batches
    Lt = (Lt + torch.triu(torch.full_like(Lt, 3.0), 1)).contiguous()
    g = torch.Generator(device="cuda").manual_seed(n)
    B = torch.randn((batch, m, n) if side == 0 else (batch, n, m), dtype=DT, device="cuda", generator=g)
    X = B.clone()
    work = torch.zeros(lib.svgp_trsm_workspace_elems(m, n, batch), dtype=DT, device="cuda")
    _lib.call("svgp_trsm_batched", side, trans, m, n, Lt.data_ptr(), m, 0 if shared else m * m, X.data_ptr(), X.shape[-1],
              X[0].numel(), batch, work.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    Lc = torch.tril(Lt).expand(batch, m, m)
    op = Lc.transpose(1, 2) if trans else Lc
    want = torch.linalg.solve_triangular(op, B, upper=bool(trans), left=side == 0)
    assert float((X - want).abs().max() / want.abs().max()) < 1e-10
    res = (op @ X - B) if side == 0 else (X @ op - B)
    assert float(res.abs().max()) < 1e-10 * float(B.abs().max()) * m


@pytest.mark.parametrize("m,jitter", [(128, 1e-6), (256, 1e-6), (512, 1e-6), (800, 1e-4), (800, 1e-6), (2048, 1e-6)])
def test_cholesky_on_kernel_like_spectrum(m, jitter):
    """K + jitter I with a fast-decaying spectrum (cond up to 5e7): Cholesky is backward stable here -- the factor
    reproduces A to rounding, log det equals the spectrum's, and the inverse from the factor keeps the residuals and the
    K X K sandwich at torch.linalg.inv's level (the property SVGPVAE_model.py:339-341 needs)."""
    K, Q, lam, A = _kernel_like(m, jitter, m)
    Lf, logdet, work = _potrf(A[None].contiguous())
    assert float((Lf[0] @ Lf[0].T - A).abs().max()) < 1e-13 * float(A.abs().max()) * m
    assert abs(float(logdet[0]) - float(torch.log(lam + jitter).sum())) < 1e-8 * m
    lib = _lib.load_library()
    w2 = torch.full((lib.svgp_potri_workspace_elems(m, 1),), float("nan"), dtype=DT, device="cuda")
    X = Lf.clone()
    _lib.call("svgp_potri_batched", m, 1, X.data_ptr(), None, w2.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    X = X[0]
    T = torch.linalg.inv(A)
    eye = torch.eye(m, dtype=DT, device="cuda")
    res = lambda Z: max(float((A @ Z - eye).abs().max()), float((Z @ A - eye).abs().max()))
    sand = (Q * (lam * lam / (lam + jitter))) @ Q.T
    sw = lambda Z: float((K @ Z @ K - sand).abs().max() / sand.abs().max())
    assert res(X) <= 10 * res(T) + 1e-12
    assert sw(X) <= 10 * sw(T) + 1e-13
    # solves through the factor: t = Sigma^-1 v as two triangular solves (SVGPVAE_model.py:332-334 without the inverse)
    v = torch.randn(1, m, 3, dtype=DT, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    t = v.clone()
    wk = torch.zeros(lib.svgp_trsm_workspace_elems(m, 3, 1), dtype=DT, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    _lib.call("svgp_trsm_batched", 0, 0, m, 3, Lf.data_ptr(), m, m * m, t.data_ptr(), 3, m * 3, 1, wk.data_ptr(), s)
    _lib.call("svgp_trsm_batched", 0, 1, m, 3, Lf.data_ptr(), m, m * m, t.data_ptr(), 3, m * 3, 1, wk.data_ptr(), s)
    torch.cuda.synchronize()
    want = torch.cholesky_solve(v[0], torch.linalg.cholesky(A))
    assert float((A @ t[0] - v[0]).abs().max()) <= 10 * float((A @ want - v[0]).abs().max()) + 1e-12


def test_not_positive_definite_gives_nan_not_garbage():
    A = -torch.eye(40, dtype=DT, device="cuda")[None].contiguous()
    Lf, logdet, _ = _potrf(A)
    assert bool(torch.isnan(logdet).all()) and bool(torch.isnan(Lf[0, 0, 0]))

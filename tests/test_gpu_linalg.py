"""Batched float64 GEMM (MFMA) and blocked Gauss-Jordan SPD inverse against torch on the same inputs."""
import ctypes as C

import pytest
import torch

from svgp_vae_amd import _lib

pytestmark = pytest.mark.gpu
DT = torch.float64


def _gemm(ta, tb, M, N, K, alpha, A, B, beta, Cm, batch, sa, sb):
    s = torch.cuda.current_stream().cuda_stream
    _lib.call("svgp_dgemm_batched", ta, tb, M, N, K, alpha, A.data_ptr(), A.shape[-1], sa, B.data_ptr(), B.shape[-1], sb,
              beta, Cm.data_ptr(), Cm.shape[-1], M * N, batch, s)
    torch.cuda.synchronize()


@pytest.mark.parametrize("ta,tb", [(0, 0), (1, 0), (0, 1), (1, 1)])
@pytest.mark.parametrize("M,N,K,batch", [(64, 64, 64, 1), (100, 37, 250, 3), (256, 256, 1024, 2), (5, 1, 7, 4), (130, 200, 3, 1),
                                         (97, 129, 17, 2), (800, 800, 800, 2), (500, 800, 100, 1),
                                         (300, 260, 90, 24), (800, 800, 64, 9), (1024, 1024, 40, 40)])   # persistent tile walk
def test_dgemm_batched(ta, tb, M, N, K, batch):
    g = torch.Generator(device="cuda").manual_seed(M * 7 + N)
    A = torch.randn((batch, K, M) if ta else (batch, M, K), dtype=DT, device="cuda", generator=g)
    B = torch.randn((batch, N, K) if tb else (batch, K, N), dtype=DT, device="cuda", generator=g)
    C0 = torch.randn(batch, M, N, dtype=DT, device="cuda", generator=g)
    Cm = C0.clone()
    _gemm(ta, tb, M, N, K, 0.7, A, B, -0.3, Cm, batch, A[0].numel(), B[0].numel())
    opA = A.transpose(1, 2) if ta else A
    opB = B.transpose(1, 2) if tb else B
    want = 0.7 * opA @ opB - 0.3 * C0
    assert float((Cm - want).abs().max()) <= 1e-12 * K * float(want.abs().max() + 1)
    # shared B operand (stride 0) and beta = 0 must not read C
    Cn = torch.full((batch, M, N), float("nan"), dtype=DT, device="cuda")
    _gemm(ta, tb, M, N, K, 1.0, A, B[0].contiguous(), 0.0, Cn, batch, A[0].numel(), 0)
    want = opA @ (opB[0])
    assert float((Cn - want).abs().max()) <= 1e-12 * K * float(want.abs().max() + 1)


def test_dgemm_batched_random_tilings():
    """Seeded sweep over extents that take the 128 a + 96 b tilings, the mixed and the plain tiles, with contractions around the
    panel depth (steady-state loop / checked tail), padded leading dimensions, pointers that are not 16-byte aligned (scalar loads)
    and beta = 0 / != 0; the padding columns of C must stay untouched."""
    import random
    rnd = random.Random(7)
    st = torch.cuda.current_stream().cuda_stream
    for it in range(40):
        M = rnd.choice([192, 200, 224, 288, 300, 352, 416, 500, 544, 640, 800, 832, 1000, 1056])
        N = rnd.choice([192, 210, 256, 288, 320, 333, 480, 500, 736, 800, 928, 1024])
        K = rnd.choice([1, 3, 15, 16, 17, 31, 32, 33, 47, 48, 49, 64, 100, 257])
        ta, tb = rnd.randint(0, 1), rnd.randint(0, 1)
        nb = max(1, 200 // (((M + 127) // 128) * ((N + 127) // 128)) + rnd.randint(0, 2))
        pad_a, pad_b, pad_c, off = rnd.choice([0, 1, 2, 6]), rnd.choice([0, 1, 2, 6]), rnd.choice([0, 3]), rnd.choice([0, 1])
        ra, ca = (K, M) if ta else (M, K)
        rb, cb = (N, K) if tb else (K, N)
        g = torch.Generator(device="cuda").manual_seed(it)
        Abuf = torch.randn(nb * ra * (ca + pad_a) + 8, dtype=DT, device="cuda", generator=g)
        Bbuf = torch.randn(nb * rb * (cb + pad_b) + 8, dtype=DT, device="cuda", generator=g)
        Cbuf = torch.randn(nb * M * (N + pad_c) + 8, dtype=DT, device="cuda", generator=g)
        A = Abuf[off:off + nb * ra * (ca + pad_a)].view(nb, ra, ca + pad_a)
        B = Bbuf[off:off + nb * rb * (cb + pad_b)].view(nb, rb, cb + pad_b)
        Cm = Cbuf[off:off + nb * M * (N + pad_c)].view(nb, M, N + pad_c)
        C0 = Cm.clone()
        alpha, beta = 0.7, rnd.choice([0.0, -0.3])
        _lib.call("svgp_dgemm_batched", ta, tb, M, N, K, alpha, A.data_ptr(), ca + pad_a, ra * (ca + pad_a), B.data_ptr(),
                  cb + pad_b, rb * (cb + pad_b), beta, Cm.data_ptr(), N + pad_c, M * (N + pad_c), nb, st)
        torch.cuda.synchronize()
        opA = A[:, :, :ca].transpose(1, 2) if ta else A[:, :, :ca]
        opB = B[:, :, :cb].transpose(1, 2) if tb else B[:, :, :cb]
        want = alpha * opA @ opB + beta * C0[:, :, :N]
        case = (it, ta, tb, M, N, K, nb, pad_a, pad_b, pad_c, off)
        assert float((Cm[:, :, :N] - want).abs().max()) <= 1e-12 * K * float(want.abs().max() + 1), case
        assert pad_c == 0 or torch.equal(Cm[:, :, N:], C0[:, :, N:]), case


@pytest.mark.parametrize("ta,tb", [(0, 0), (1, 0), (0, 1), (1, 1)])
@pytest.mark.parametrize("M,N,K", [(1050, 500, 1024), (500, 1024, 1050), (1024, 500, 1050), (1050, 4, 500), (2, 500, 1050),
                                   (1050, 2, 500), (300, 300, 1031), (64, 64, 255), (2048, 2048, 512)])
def test_dgemm_splitk(ta, tb, M, N, K):
    """One GEMM, K cut into slices + fixed-order reduction; shapes of the moving-ball MLP layers, a contraction that is
    not a multiple of the slice count, a shape below the split threshold and one with enough tiles not to split."""
    lib = _lib.load_library()
    g = torch.Generator(device="cuda").manual_seed(M + 3 * N + K)
    A = torch.randn((K, M) if ta else (M, K), dtype=DT, device="cuda", generator=g)
    B = torch.randn((N, K) if tb else (K, N), dtype=DT, device="cuda", generator=g)
    C0 = torch.randn(M, N, dtype=DT, device="cuda", generator=g)
    need = lib.svgp_dgemm_splitk_scratch_elems(M, N, K)
    assert (need == 0) == (K < 256 or ((M + 63) // 64) * ((N + 63) // 64) >= 256)
    scratch = torch.full((max(need, 1),), float("nan"), dtype=DT, device="cuda")
    want_ab = (A.t() if ta else A) @ (B.t() if tb else B)
    for alpha, beta in ((0.7, -0.3), (1.0, 0.0)):
        Cm = C0.clone() if beta != 0 else torch.full((M, N), float("nan"), dtype=DT, device="cuda")
        _lib.call("svgp_dgemm_splitk", ta, tb, M, N, K, alpha, A.data_ptr(), A.shape[-1], B.data_ptr(), B.shape[-1], beta,
                  Cm.data_ptr(), N, scratch.data_ptr(), scratch.numel(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        want = alpha * want_ab + (beta * C0 if beta != 0 else 0)
        assert float((Cm - want).abs().max()) <= 1e-12 * K * float(want.abs().max() + 1)
    if need:
        with pytest.raises(_lib.SvgpError, match="scratch"):
            _lib.call("svgp_dgemm_splitk", ta, tb, M, N, K, 1.0, A.data_ptr(), A.shape[-1], B.data_ptr(), B.shape[-1], 0.0,
                      C0.data_ptr(), N, scratch.data_ptr(), need - 1, None)


@pytest.mark.parametrize("m,batch", [(32, 1), (64, 3), (72, 2), (100, 1), (128, 2), (130, 3), (256, 17), (513, 2), (640, 2), (800, 3)])
def test_spd_inverse_batched(m, batch):
    g = torch.Generator(device="cuda").manual_seed(m)
    X = torch.randn(batch, m, m + 8, dtype=DT, device="cuda", generator=g)
    A = X @ X.transpose(1, 2) / m + 0.05 * torch.eye(m, dtype=DT, device="cuda")
    inv = A.clone()
    logdet = torch.zeros(batch, dtype=DT, device="cuda")
    lib = _lib.load_library()
    work = torch.full((lib.svgp_spd_inverse_workspace_elems(m, batch),), float("nan"), dtype=DT, device="cuda")   # poisoned scratch
    _lib.call("svgp_spd_inverse_batched", m, batch, inv.data_ptr(), logdet.data_ptr(), work.data_ptr(),
              torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    want = torch.linalg.inv(A)
    assert float((inv - want).abs().max() / want.abs().max()) < 1e-9
    assert float((logdet - torch.linalg.slogdet(A)[1]).abs().max()) < 1e-9 * m
    eye = torch.eye(m, dtype=DT, device="cuda")
    assert float((inv @ A - eye).abs().max()) < 1e-8


@pytest.mark.parametrize("m,jitter", [(128, 1e-6), (512, 1e-6), (800, 1e-4), (800, 1e-6)])
def test_spd_inverse_residual_on_kernel_like_spectrum(m, jitter):
    """K + jitter I with a fast-decaying spectrum (what the GP block inverts, SVGPVAE_model.py:239,319,331): the
    residuals |A X - I|, |X A - I| and the sandwich K X K must stay at torch.linalg.inv's level.  (A blocked elimination
    with 128-wide explicit pivot-inverse products loses cond(P) here: 1e-2 residual at jitter 1e-6; the library uses the
    32-block sweep below m = 512 and potrf + potri from there on.)"""
    g = torch.Generator(device="cuda").manual_seed(m)
    Q, _ = torch.linalg.qr(torch.randn(m, m, dtype=DT, device="cuda", generator=g))
    lam = 50 * torch.exp(-torch.arange(m, dtype=DT, device="cuda") / 8)
    K = (Q * lam) @ Q.T
    K = 0.5 * (K + K.T)
    eye = torch.eye(m, dtype=DT, device="cuda")
    A = K + jitter * eye
    X = A.clone()[None].contiguous()
    logdet = torch.zeros(1, dtype=DT, device="cuda")
    lib = _lib.load_library()
    work = torch.full((lib.svgp_spd_inverse_workspace_elems(m, 1),), float("nan"), dtype=DT, device="cuda")
    _lib.call("svgp_spd_inverse_batched", m, 1, X.data_ptr(), logdet.data_ptr(), work.data_ptr(),
              torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    X = X[0]
    T = torch.linalg.inv(A)
    res = lambda Z: max(float((A @ Z - eye).abs().max()), float((Z @ A - eye).abs().max()))
    sand = (Q * (lam * lam / (lam + jitter))) @ Q.T
    sw = lambda Z: float((K @ Z @ K - sand).abs().max() / sand.abs().max())
    assert res(X) <= 10 * res(T) + 1e-12
    assert sw(X) <= 10 * sw(T) + 1e-13
    assert abs(float(logdet[0]) - float(torch.log(lam + jitter).sum())) < 1e-8 * m


@pytest.mark.parametrize("m", [1, 5, 16, 33, 100, 257, 800])
def test_lu_inverse_of_a_general_matrix(m):
    """svgp_lu_inverse (LU with partial pivoting + two triangular solves) against torch.linalg.inv on matrices that NEED row
    pivoting: non-symmetric, with a zero (1,1) entry and rows of very different scale."""
    from svgp_vae_amd.sprites import general_inverse
    g = torch.Generator().manual_seed(m)
    A = torch.randn(m, m, dtype=torch.float64, generator=g)
    if m > 1:
        A[0, 0] = 0.0
    A *= torch.logspace(-3, 3, m, dtype=torch.float64)[torch.randperm(m, generator=g)][:, None]
    dev = torch.device("cuda:0")
    X = general_inverse(A.to(dev)).cpu()
    ref = torch.linalg.inv(A)
    resid = float((A @ X - torch.eye(m, dtype=torch.float64)).abs().max())
    resid_ref = float((A @ ref - torch.eye(m, dtype=torch.float64)).abs().max())
    assert resid <= max(1e-9, 50 * resid_ref), (resid, resid_ref)
    assert float((X - ref).abs().max() / ref.abs().max()) < 1e-7


def test_lu_inverse_pivot_rule_and_permutation():
    """A permutation-like matrix: the inverse is its transpose exactly (no arithmetic, only the pivot search / swaps)."""
    from svgp_vae_amd.sprites import general_inverse
    m = 70
    perm = torch.randperm(m, generator=torch.Generator().manual_seed(0))
    A = torch.zeros(m, m, dtype=torch.float64)
    A[torch.arange(m), perm] = torch.arange(1, m + 1, dtype=torch.float64)
    X = general_inverse(A.to("cuda:0")).cpu()
    assert torch.equal(X, torch.linalg.inv(A))


def test_lu_inverse_on_the_sprites_inducing_kernel_matrix():
    """What SPRITES_experiment.py:178 inverts: K_mm WITHOUT jitter.  (a) SE x SE kernels (--K_SE), full rank: 1e-9 against
    torch.linalg.inv through the quantity the caller uses, diag(K_bm K_mm^-1 K_mb) (SVGPVAE_model.py:610-635).  (b) linear x linear
    kernels at m = 800: rank <= 128, the inverse is dominated by rounding in the null space and NO two LU implementations agree on it
    -- the bar is the reference computation's own response to a one-ulp perturbation of K_mm (the tolerance rule of
    tests/test_gpu_fullsize.py), plus: every entry finite."""
    from svgp_vae_amd.sprites import general_inverse
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    m, La, Lc, b = 800, 8, 16, 64
    Za, Zc = torch.randn(m, La, dtype=torch.float64, generator=g), torch.randn(m, Lc, dtype=torch.float64, generator=g)
    Xa, Xc = torch.randn(b, La, dtype=torch.float64, generator=g), torch.randn(b, Lc, dtype=torch.float64, generator=g)
    cos = lambda x, y: (x / x.norm(dim=1, keepdim=True)) @ (y / y.norm(dim=1, keepdim=True)).t()
    se = lambda x, y, l: torch.exp(-0.5 * torch.cdist(x, y) ** 2 / l ** 2)
    diag = lambda Kb, Ki: torch.einsum("bm,mn,bn->b", Kb, Ki, Kb)
    # (a) full rank
    K, Kb = se(Za, Za, 3.0) * se(Zc, Zc, 4.0), se(Xa, Za, 3.0) * se(Xc, Zc, 4.0)
    got, ref = diag(Kb, general_inverse(K.to(dev)).cpu()), diag(Kb, torch.linalg.inv(K))
    ulp = lambda: 1.0 + 2.0 ** -52 * (torch.randint(0, 2, K.shape, generator=g).to(torch.float64) * 2 - 1)
    pert = diag(Kb, torch.linalg.inv(K * ulp()))
    tol = max(1e-9, 20 * float((pert - ref).abs().max() / ref.abs().max()))
    assert float((got - ref).abs().max() / ref.abs().max()) < tol, tol
    # (b) rank 128 of 800
    K, Kb = cos(Za, Za) * cos(Zc, Zc), cos(Xa, Za) * cos(Xc, Zc)
    X = general_inverse(K.to(dev)).cpu()
    assert torch.isfinite(X).all()
    ref = torch.linalg.inv(K)
    pert = torch.linalg.inv(K * ulp())
    resp = float((diag(Kb, pert) - diag(Kb, ref)).abs().max() / diag(Kb, ref).abs().max())
    err = float((diag(Kb, X) - diag(Kb, ref)).abs().max() / diag(Kb, ref).abs().max())
    print(f"rank-128 K_mm: |diag(Kb X Kb^T) - torch| / max = {err:.2e}; torch's own one-ulp response {resp:.2e}")
    assert err < max(1e-6, 20 * resp)

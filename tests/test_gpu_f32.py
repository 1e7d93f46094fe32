"""float32 instantiation of the SPRITES training step: the reference runs SPRITES in float32 end to
end (VAE_utils.py:277 `dtype = tf.float32`, SVGPVAE_model.py:516 `dtype=np.float32`).

  * tap-table convolutions on v_mfma_f32_16x16x4_f32 (svgp_conv_taps_*_f32) against the float64 Keras-semantics oracle;
  * svgp_sgemm_batched (float32) and svgp_dgemm_f32c_batched (float64 storage, float32 MFMA arithmetic) against torch;
  * the whole step with float32 networks (and float32-MFMA GP products) against the float64 oracle.

Tolerances (float32 arithmetic against a float64 oracle): layer outputs / gradients 2e-5 of the tensor's max; the step's
ELBO and scalar members 1e-3 relative -- north_star's bar, stated there for exactly this comparison."""
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import sprites_oracle as SO
from oracle import svgpvae_oracle as O
from svgp_vae_amd import _lib
from tests.test_gpu_conv import CASES, _oracle

pytestmark = pytest.mark.gpu
DT = torch.float64
F32 = torch.float32


@pytest.mark.parametrize("Hi,Ci,Co,k,stride,padding,up,elu", CASES)
def test_conv_layer_float32(Hi, Ci, Co, k, stride, padding, up, elu):
    from svgp_vae_amd.conv import ConvLayer
    g = torch.Generator().manual_seed(Hi * 31 + Ci * 7 + Co + k)
    n = 3
    x = torch.randn(n, Hi, Hi, Ci, dtype=DT, generator=g)
    w = torch.randn(k, k, Ci, Co, dtype=DT, generator=g) * 0.3
    b = torch.randn(Co, dtype=DT, generator=g) * 0.1
    x, w, b = (t.float().double() for t in (x, w, b))                 # inputs exactly representable in float32
    xr, wr, br = (t.clone().requires_grad_() for t in (x, w, b))
    want = _oracle(xr, wr, br, k, stride, padding, up, elu)
    gout = torch.randn(*want.shape, dtype=DT, generator=g).float().double()
    gx, gw, gb = torch.autograd.grad((want * gout).sum(), (xr, wr, br))
    lay = ConvLayer(Hi, Ci, Co, k=k, stride=stride, padding=padding, up=up, elu=elu, dtype=F32)
    dev = "cuda"
    s = torch.cuda.current_stream().cuda_stream
    dx_, dw_, db_ = x.to(dev, F32), w.to(dev, F32), b.to(dev, F32)
    out = torch.full((n, lay.Ho, lay.Ho, Co), float("nan"), dtype=F32, device=dev)
    lay.forward(dx_, dw_, db_, out, s)
    torch.cuda.synchronize()
    rel = lambda a, c: float((a.cpu().double() - c).abs().max() / (c.abs().max() + 1e-300))
    assert rel(out, want.detach()) < 2e-6
    dout = gout.to(dev, F32).clone()
    ggw = torch.zeros(k, k, Ci, Co, dtype=F32, device=dev)
    ggb = torch.zeros(Co, dtype=F32, device=dev)
    scratch = torch.zeros(lay.scratch_elems(64), dtype=F32, device=dev)
    gdx = lay.backward(dx_, dw_, out, dout, ggw, ggb, scratch, s, nwg=64)
    torch.cuda.synchronize()
    assert rel(ggb, gb) < 2e-5 and rel(ggw, gw) < 2e-5 and rel(gdx, gx) < 2e-5


@pytest.mark.parametrize("ta,tb", [(0, 0), (1, 0), (0, 1), (1, 1)])
@pytest.mark.parametrize("M,N,K,batch", [(64, 64, 64, 1), (100, 37, 250, 3), (5, 1, 7, 4), (97, 129, 17, 2), (800, 800, 800, 2),
                                         (500, 128, 1024, 1), (256, 256, 1024, 2)])
def test_float32_gemms(ta, tb, M, N, K, batch):
    g = torch.Generator(device="cuda").manual_seed(M * 7 + N)
    A = torch.randn((batch, K, M) if ta else (batch, M, K), dtype=F32, device="cuda", generator=g)
    B = torch.randn((batch, N, K) if tb else (batch, K, N), dtype=F32, device="cuda", generator=g)
    C0 = torch.randn(batch, M, N, dtype=F32, device="cuda", generator=g)
    opA = (A.transpose(1, 2) if ta else A).double()
    opB = (B.transpose(1, 2) if tb else B).double()
    want = 0.7 * opA @ opB - 0.3 * C0.double()
    bound = 2e-7 * K ** 0.5 * float((opA.abs() @ opB.abs()).max()) + 1e-6 * float(want.abs().max())
    s = torch.cuda.current_stream().cuda_stream
    Cm = C0.clone()
    _lib.call("svgp_sgemm_batched", ta, tb, M, N, K, 0.7, A.data_ptr(), A.shape[-1], A[0].numel(), B.data_ptr(), B.shape[-1],
              B[0].numel(), -0.3, Cm.data_ptr(), N, M * N, batch, s)
    torch.cuda.synchronize()
    assert float((Cm.double() - want).abs().max()) <= bound
    # float64 storage, float32 MFMA arithmetic: same bound, float64 operands rounded to float32 on the way in
    Ad, Bd, Cd = A.double(), B.double(), C0.double().clone()
    _lib.call("svgp_dgemm_f32c_batched", ta, tb, M, N, K, 0.7, Ad.data_ptr(), Ad.shape[-1], Ad[0].numel(), Bd.data_ptr(),
              Bd.shape[-1], Bd[0].numel(), -0.3, Cd.data_ptr(), N, M * N, batch, s)
    torch.cuda.synchronize()
    assert float((Cd - want).abs().max()) <= bound
    assert float((Cd - Cm.double()).abs().max()) <= 4e-7 * float(want.abs().max()) + 1e-30    # same arithmetic, f64 epilogue


def _sprites_case(b, frames, L, La, Lc, m, n_act, seed, K_SE, norm):
    g = torch.Generator().manual_seed(seed)
    params = {k: torch.tensor(v, dtype=DT) for k, v in SO.glorot_init(L, Lc, seed).items()}
    for k in params:
        if k.endswith("_b"):
            params[k] = 0.05 * torch.randn(*params[k].shape, dtype=DT, generator=g)
    gp = dict(inducing_index_points=torch.randn(m, La + Lc, dtype=DT, generator=g) * 1.5,
              GPLVM_action=torch.randn(n_act, La, dtype=DT, generator=g) * 1.5,
              l_action=torch.tensor(5.0, dtype=DT), sigma_action=torch.tensor(1.4, dtype=DT),
              l_character=torch.tensor(7.0, dtype=DT), sigma_character=torch.tensor(1.2, dtype=DT))
    images = torch.rand(b, 64, 64, 3, dtype=DT, generator=g).float().double()
    ids = torch.randint(0, n_act, (b,), generator=g)
    eps = torch.randn(b, L, dtype=DT, generator=g)
    seg, rep = SO.aux_data_sprites_utils(b, frames, frames)
    return params, gp, images, ids, eps, seg, rep


_ORACLE_CACHE = {}


@pytest.mark.parametrize("m,b,L,K_SE,norm,gemm_f32", [(12, 8, 6, True, False, 0), (72, 8, 6, False, True, 1), (72, 8, 6, True, False, 1),
                                                      (800, 100, 64, False, True, 2), (800, 100, 64, False, True, 1)])
def test_sprites_step_float32(m, b, L, K_SE, norm, gemm_f32):
    """Whole SPRITES step with float32 networks against the float64 oracle.  gemm_f32 = 1: every product of the large-m
    GP block on the float32 MFMA; 2: only the statistics.  The report lists what float32 costs per quantity."""
    from svgp_vae_amd import sprites as S
    frames, La, Lc, n_act = (4 if b == 8 else 50), 8, 16, (9 if b == 8 else 72)
    params, gp, images, ids, eps, seg, rep = _sprites_case(b, frames, L, La, Lc, m, n_act, m + b, K_SE, norm)
    jitter, N_train = 0.01, (100.0 if b == 8 else 50000.0)
    kw = dict(beta=0.001, C_ma=torch.tensor(0.0, dtype=DT), lagrange_mult=torch.tensor(1.0, dtype=DT), alpha=0.0,
              kappa=math.sqrt(0.0075), L=L, L_action=La, jitter=jitter, N_train=N_train, segment_ids=seg, repeats=rep,
              clipping_qs=True, GECO=True, K_obj_normalize=norm, K_SE=K_SE, clip_grad=None, titsias=False)
    key = (m, b, L, K_SE, norm)               # (the two m = 800 cases differ in gemm_f32 only: one oracle evaluation, ~15 s of CPU)
    if key not in _ORACLE_CACHE:
        _ORACLE_CACHE[key] = SO.loss_and_grads(params, gp, (images, ids), eps, formulation="efficient", **kw)
    want, wgrads = _ORACLE_CACHE[key]
    svgp = S.spritesSVGP(False, False, gp["inducing_index_points"].numpy(), 'main', jitter, N_train, La,
                         gp["GPLVM_action"].numpy(), Lc, L, fixed_GP_params=False, fixed_GPLVM=False,
                         K_obj_normalize=norm, K_SE=K_SE)
    init = dict(params)
    init["se"] = torch.stack([gp["l_action"], gp["sigma_action"], gp["l_character"], gp["sigma_character"]])
    eng = S.SpritesStepEngine(S.spritesVAE(L), S.sprites_representation_network(Lc), svgp, b_max=b, seg_len=frames,
                              clip_qs=True, geco=True, kappa_squared=0.0075, beta=0.001, params=init,
                              net_dtype=torch.float32, gemm_f32=gemm_f32)
    eng.set_scalars(c_ma=0.0, lagrange=1.0, alpha=0.0)
    dev = eng.dev
    eng.step(images.to(dev), ids.to(dev, DT), eps.to(dev), adam=False)
    got = eng.outputs()
    rel = lambda a, c: float((torch.as_tensor(a, dtype=DT).cpu().reshape(-1) - torch.as_tensor(c, dtype=DT).reshape(-1)).abs().max()
                             / max(float(torch.as_tensor(c, dtype=DT).abs().max()), 1e-9))
    names = ("elbo", "recon_loss", "KL_term", "inside_elbo", "ce_term", "p_m", "p_v", "qnet_mu", "qnet_var", "recon_images",
             "inside_elbo_recon", "inside_elbo_kl", "latent_samples", "C_ma", "lagrange_mult")
    report = [f"{n}: rel {rel(got[i], want[i]):.2e}" for i, n in enumerate(names)]
    gr = eng.grads
    for k, w in wgrads.items():
        if k in gr:
            report.append(f"grad {k}: rel {rel(gr[k], w):.2e} (max|want| {float(w.abs().max()):.2e})")
    print("\n".join(report))
    assert rel(got[7], want[7]) < 1e-4 and rel(got[8], want[8]) < 1e-4          # encoder outputs: float32 level
    if m == 800 and gemm_f32 == 1:
        # Measured (r02f / r02l): with EVERY product of the GP block in float32 at m = 800 / jitter 0.01 the ELBO is off by
        # 2e-4 ... 9e-2 (depending on which triangle of a symmetric product is computed), p_m by 19 %, the reconstruction
        # by 4 % and the gradients are useless (encoder dense layer 114 %, inducing points 1e6): the K Sigma^-1 K
        # sandwiches and k^T Sigma^-1 k cancel terms of size 1 / jitter.  This is the arithmetic of the reference's
        # float32 graph; it is NOT a parity mode at this size (and float32 statistics alone, mode 2, pass this one-step
        # comparison but diverge within 11 Adam steps without --clip_qs: DESIGN.md section 10).  The SPRITES driver /
        # bench run float32 networks + a float64 GP block.
        assert all(bool(torch.isfinite(torch.as_tensor(got[i])).all()) for i in range(15))
        return
    # north_star: ELBO within 1e-3 relative
    assert rel(got[0], want[0]) < 1e-3 and rel(got[2], want[2]) < 1e-3 and rel(got[1], want[1]) < 1e-3
    assert rel(got[9], want[9]) < 1e-3
    assert rel(got[5], want[5]) < 1e-2 and rel(got[6], want[6]) < 1e-2
    for k in ("enc_c1_w", "dec_c7_w", "dec_c1_w", "repr_c1_w", "enc_d_w", "dec_d_w"):
        assert rel(gr[k], wgrads[k]) < 2e-2, k

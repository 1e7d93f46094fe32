"""Independent pins of the oracle's restated third-party formulas.

TensorFlow 1.15 / TFP 0.8 cannot run here and the reference holds no golden vectors, so parity stays "unpinned" in the
strict sense.  These tests lower the risk: every restated kernel / layer formula is compared with a SECOND, separately
written implementation that is in the image -- scikit-learn's Gaussian-process kernels (the same three covariance
functions: ExpSineSquared = TFP ExpSinSquared, RBF = TFP ExponentiatedQuadratic, DotProduct(sigma_0=0) = TFP Linear),
an explicit im2col convolution in numpy written from TensorFlow's documented SAME / VALID rules, scipy's correlate2d, and
scipy / numpy linear algebra for the Cholesky log-determinants and the Adam recursion."""
import math

import numpy as np
import pytest
import torch

from oracle import sprites_oracle as SO
from oracle import svgpvae_oracle as O

sk = pytest.importorskip("sklearn.gaussian_process.kernels")
DT = torch.float64


def _t(a):
    return torch.tensor(np.asarray(a), dtype=DT)


def test_exp_sin_squared_matches_sklearn_ExpSineSquared():
    """TFP ExpSinSquared(amplitude a, length_scale l, period T): a^2 exp(-2 sin^2(pi |x-y| / T) / l^2)
    (call site SVGPVAE_model.py:416, period 2 pi) == sklearn ConstantKernel(a^2) * ExpSineSquared(l, T)."""
    rs = np.random.RandomState(0)
    x, y = rs.uniform(0, 2 * np.pi, (13, 1)), rs.uniform(-3, 9, (7, 1))
    for a, l, T in ((1.0, 1.0, 2 * np.pi), (0.7, 1.3, 2 * np.pi), (1.9, 0.4, 3.0)):
        want = (sk.ConstantKernel(a * a) * sk.ExpSineSquared(length_scale=l, periodicity=T))(x, y)
        got = O.exp_sin_squared(_t(x[:, 0]), _t(y[:, 0]), _t(a), _t(l), period=T)
        np.testing.assert_allclose(got.numpy(), want, rtol=1e-13, atol=1e-15)
        gd = O.exp_sin_squared(_t(x[:7, 0]), _t(y[:, 0]), _t(a), _t(l), period=T, diag_only=True)
        np.testing.assert_allclose(gd.numpy(), np.diag(want[:7]), rtol=1e-13, atol=1e-15)


def test_linear_kernel_matches_sklearn_DotProduct_and_cosine():
    rs = np.random.RandomState(1)
    x, y = rs.normal(0, 1.5, (9, 8)), rs.normal(0, 1.5, (5, 8))
    want = sk.DotProduct(sigma_0=0.0)(x, y)
    np.testing.assert_allclose(O.linear_kernel(_t(x), _t(y)).numpy(), want, rtol=1e-13)
    from sklearn.metrics.pairwise import cosine_similarity                 # the normalised variant, :465-474 / :576-598
    np.testing.assert_allclose(O.linear_kernel(_t(x), _t(y), normalize=True).numpy(), cosine_similarity(x, y), rtol=1e-12)
    np.testing.assert_allclose(O.linear_kernel(_t(x[:5]), _t(y), normalize=True, diag_only=True).numpy(),
                               np.diag(cosine_similarity(x[:5], y)), rtol=1e-12)


def test_exponentiated_quadratic_matches_sklearn_RBF():
    """TFP ExponentiatedQuadratic(amplitude s, length_scale l): s^2 exp(-|x-y|^2 / (2 l^2)) (SVGPVAE_model.py:60,542-544)."""
    rs = np.random.RandomState(2)
    x, y = rs.normal(0, 1.5, (11, 16)), rs.normal(0, 1.5, (6, 16))
    for s, l in ((1.0, 1.0), (1.4, 5.0), (0.3, 0.7)):
        want = (sk.ConstantKernel(s * s) * sk.RBF(length_scale=l))(x, y)
        np.testing.assert_allclose(SO.exponentiated_quadratic(_t(x), _t(y), _t(s), _t(l)).numpy(), want, rtol=1e-12, atol=1e-300)
    from oracle import ball_oracle as BO
    t, z = rs.uniform(0, 30, (30, 1)), rs.uniform(0, 30, (15, 1))
    want = sk.RBF(length_scale=2.0)(t, z)
    np.testing.assert_allclose(BO.se_matrix(_t(t), _t(z), _t(2.0)).numpy(), want, rtol=1e-12)   # moving ball, :60


def test_product_kernels_of_both_models_match_sklearn_compositions():
    """mnistSVGP: view(angle) * object(dot), with table gather and inducing rows (SVGPVAE_model.py:427-476);
    spritesSVGP: action * character (:550-600).  sklearn evaluates each factor on its own columns."""
    rs = np.random.RandomState(3)
    M, n_obj = 8, 20
    ip = np.concatenate([np.arange(6)[:, None], rs.uniform(0, 6.28, (6, 1)), rs.normal(0, 1.5, (6, M))], 1)
    table = rs.normal(0, 1.5, (n_obj, M))
    aux = np.concatenate([rs.randint(0, n_obj, (10, 1)), rs.uniform(0, 6.28, (10, 1)), rs.normal(0, 1, (10, M))], 1)
    a, l = 0.9, 1.2
    sv = O.MnistSVGP(False, _t(ip), _t(table), _t(l), _t(a), 1e-6, 100.0)
    view = sk.ConstantKernel(a * a) * sk.ExpSineSquared(length_scale=l, periodicity=2 * np.pi)
    lin = sk.DotProduct(sigma_0=0.0)
    obj_rows = table[aux[:, 0].astype(int)]
    np.testing.assert_allclose(sv.kernel_matrix(_t(aux), _t(ip), x_inducing=False).numpy(),
                               view(aux[:, 1:2], ip[:, 1:2]) * lin(obj_rows, ip[:, 2:]), rtol=1e-12)
    np.testing.assert_allclose(sv.kernel_matrix(_t(ip), _t(ip)).numpy(), view(ip[:, 1:2]) * lin(ip[:, 2:]), rtol=1e-12)
    np.testing.assert_allclose(sv.kernel_matrix(_t(aux), _t(aux), False, False, diag_only=True).numpy(),
                               np.diag(view(aux[:, 1:2]) * lin(obj_rows)), rtol=1e-12)
    # SPRITES, SE x SE and linear x linear
    La, Lc, n_act = 8, 16, 7
    ips = rs.normal(0, 1.5, (5, La + Lc))
    act = rs.normal(0, 1.5, (n_act, La))
    x = np.concatenate([rs.randint(0, n_act, (9, 1)), rs.normal(0, 1.5, (9, Lc))], 1)
    se = dict(l_action=_t(5.0), sigma_action=_t(1.4), l_character=_t(7.0), sigma_character=_t(1.2))
    ssv = SO.SpritesSVGP(_t(ips), _t(act), 0.01, 100.0, La, K_SE=True, se_params=se)
    ka = sk.ConstantKernel(1.4 ** 2) * sk.RBF(5.0)
    kc = sk.ConstantKernel(1.2 ** 2) * sk.RBF(7.0)
    xa = act[x[:, 0].astype(int)]
    np.testing.assert_allclose(ssv.kernel_matrix(_t(x), _t(ips), x_inducing=False).numpy(),
                               ka(xa, ips[:, :La]) * kc(x[:, 1:], ips[:, La:]), rtol=1e-12)
    lsv = SO.SpritesSVGP(_t(ips), _t(act), 0.01, 100.0, La, K_obj_normalize=False)
    np.testing.assert_allclose(lsv.kernel_matrix(_t(x), _t(ips), x_inducing=False).numpy(),
                               lin(xa, ips[:, :La]) * lin(x[:, 1:], ips[:, La:]), rtol=1e-12)


# ---------------------------------------------------------------------------------------------------------------
# Keras Conv2D semantics: explicit im2col written from TensorFlow's documented padding rules
#   VALID: out = ceil((in - k + 1) / s), no padding
#   SAME : out = ceil(in / s); pad_total = max((out - 1) s + k - in, 0); pad_before = pad_total // 2 (extra goes after)
# ---------------------------------------------------------------------------------------------------------------
def _im2col_conv(x, w, bias, stride, padding):
    b, H, W, Ci = x.shape
    kh, kw, _, Co = w.shape
    if padding == "same":
        Ho, Wo = -(-H // stride), -(-W // stride)
        ph, pw = max((Ho - 1) * stride + kh - H, 0), max((Wo - 1) * stride + kw - W, 0)
        xp = np.zeros((b, H + ph, W + pw, Ci))
        xp[:, ph // 2:ph // 2 + H, pw // 2:pw // 2 + W] = x
    else:
        Ho, Wo = -(-(H - kh + 1) // stride), -(-(W - kw + 1) // stride)
        xp = x
    cols = np.empty((b, Ho, Wo, kh * kw * Ci))
    for i in range(Ho):
        for j in range(Wo):
            cols[:, i, j] = xp[:, i * stride:i * stride + kh, j * stride:j * stride + kw].reshape(b, -1)
    return cols @ w.reshape(kh * kw * Ci, Co) + bias


@pytest.mark.parametrize("H,k,Ci,Co,stride,padding", [
    (28, 3, 1, 8, 2, "valid"), (13, 3, 8, 8, 2, "valid"), (6, 3, 8, 8, 2, "valid"),       # mnistVAE encoder  VAE_utils.py:114-121
    (8, 3, 8, 8, 1, "same"), (16, 3, 8, 8, 1, "valid"), (28, 3, 8, 1, 1, "same"),         # mnistVAE decoder  :131-141
    (64, 3, 3, 16, 1, "same"), (64, 3, 16, 16, 2, "same"), (16, 3, 16, 16, 2, "same"),    # spritesVAE encoder :294-305
    (64, 2, 3, 16, 2, "same"), (32, 2, 16, 16, 2, "same"), (15, 2, 4, 4, 2, "same"),      # repr network :375-386 (+ odd size)
    (7, 3, 2, 3, 2, "same")])                                                             # odd input, stride 2: asymmetric pad
def test_conv2d_nhwc_matches_im2col(H, k, Ci, Co, stride, padding):
    rs = np.random.RandomState(H * 7 + k)
    x, w, bias = rs.normal(size=(2, H, H, Ci)), rs.normal(size=(k, k, Ci, Co)), rs.normal(size=Co)
    got = O._conv2d_nhwc(_t(x), _t(w), _t(bias), stride, padding).numpy()
    want = _im2col_conv(x, w, bias, stride, padding)
    assert got.shape == want.shape
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-12)


def test_conv2d_stride1_same_matches_scipy_correlate2d():
    from scipy.signal import correlate2d
    rs = np.random.RandomState(5)
    x, w = rs.normal(size=(1, 12, 12, 3)), rs.normal(size=(3, 3, 3, 2))
    got = O._conv2d_nhwc(_t(x), _t(w), None, 1, "same").numpy()
    for co in range(2):
        want = sum(correlate2d(x[0, :, :, ci], w[:, :, ci, co], mode="same") for ci in range(3))
        np.testing.assert_allclose(got[0, :, :, co], want, rtol=1e-12, atol=1e-12)


def test_upsampling_is_nearest_neighbour_kron():
    rs = np.random.RandomState(6)
    x = rs.normal(size=(2, 4, 4, 3))
    want = np.stack([np.stack([np.kron(x[n, :, :, c], np.ones((2, 2))) for c in range(3)], -1) for n in range(2)])
    np.testing.assert_allclose(O._upsample2_nhwc(_t(x)).numpy(), want)


def test_elu_and_dense_layer_conventions():
    """Keras ELU(alpha=1): x>0 ? x : exp(x)-1; Dense: x @ W(in,out) + b."""
    x = np.linspace(-5, 5, 41)
    import torch.nn.functional as F
    np.testing.assert_allclose(F.elu(_t(x)).numpy(), np.where(x > 0, x, np.expm1(x)), rtol=1e-14)


def test_cholesky_logdet_and_inverse_against_scipy():
    """tf.linalg.cholesky log-dets (SVGPVAE_model.py:270-274) / tf.linalg.inv: the oracle's torch.linalg calls vs
    scipy's LAPACK wrappers on a jittered kernel matrix."""
    import scipy.linalg as sl
    rs = np.random.RandomState(7)
    ip = np.concatenate([np.arange(24)[:, None], rs.uniform(0, 6.28, (24, 1)), rs.normal(0, 1.5, (24, 32))], 1)
    sv = O.MnistSVGP(False, _t(ip), None, _t(1.0), _t(1.0), 1e-6, 100.0)
    Kj = O.add_diagonal_jitter(sv.kernel_matrix(_t(ip), _t(ip)), 1e-6)
    c = sl.cholesky(Kj.numpy(), lower=True)
    ld_t = 2 * torch.sum(torch.log(torch.diagonal(torch.linalg.cholesky(Kj))))
    assert abs(float(ld_t) - 2 * np.log(np.diag(c)).sum()) < 1e-9 * abs(float(ld_t))
    np.testing.assert_allclose(torch.linalg.inv(Kj).numpy(), sl.inv(Kj.numpy()), rtol=1e-6, atol=1e-6 * float(torch.linalg.inv(Kj).abs().max()))


def test_adam_tf1_against_a_scalar_recursion():
    """TF1 AdamOptimizer: lr_t = lr sqrt(1 - b2^t) / (1 - b1^t); m, v EMAs; theta -= lr_t m / (sqrt(v) + eps)
    (epsilon OUTSIDE the bias correction -- the TF1 'epsilon hat' form, not Kingma's Algorithm 1)."""
    p = {"w": _t([0.5, -1.0])}
    ms, vs = {"w": torch.zeros(2, dtype=DT)}, {"w": torch.zeros(2, dtype=DT)}
    theta, m, v = np.array([0.5, -1.0]), np.zeros(2), np.zeros(2)
    rs = np.random.RandomState(8)
    for t in range(1, 6):
        g = rs.normal(size=2)
        O.adam_tf1_step(p, {"w": _t(g)}, ms, vs, t, 1e-3)
        m = 0.9 * m + 0.1 * g
        v = 0.999 * v + 0.001 * g * g
        theta = theta - 1e-3 * math.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t) * m / (np.sqrt(v) + 1e-8)
        np.testing.assert_allclose(p["w"].numpy(), theta, rtol=1e-13)

"""Host-side driver utilities against fixtures / an independent restatement (no GPU): the KDE inducing-point initialiser
(utils.py:691-744) against the committed golden inducing points, the SPRITES PCA initialiser (SPRITES_utils.py:217-279)
against a plain-numpy restatement on a synthetic training dict."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _train_aux(golden):
    """The 4 050 train aux rows of the reference's default dataset, rebuilt as tests/golden/make_golden.py does: PCA object
    vectors of the 360 train ids x the 15 train angles, filtered by the reference's train-id mask."""
    gin, _ = golden
    fx = np.load(os.path.join(ROOT, "tests", "golden", "mnist_train_ids_mask.npz"))
    mask, test_angle, ov = fx["train_ids_mask"], float(fx["test_angle"]), gin["object_vectors"]
    angles16 = np.linspace(0, 2 * np.pi, 17)[:-1]
    train_angles = np.array([a for a in angles16 if abs(a - test_angle) > 1e-9])
    rows = [np.concatenate(([i, train_angles[k]], ov[i])) for i in range(360) for k in range(15) if mask[i * 15 + k]]
    aux = np.array(rows)
    assert aux.shape == (4050, 10)
    return aux


def test_generate_init_inducing_points_reproduces_the_golden_inducing_points(golden):
    """utils.generate_init_inducing_points(train_aux, n=2, PCA=True) == golden['inducing_index_points']
    (computed independently by tests/golden/make_golden.py from the reference's data files): rows [id, angle, KDE draw]."""
    import importlib
    utils = importlib.import_module("svgp_vae_amd.utils")
    gin, _ = golden
    ip = utils.generate_init_inducing_points(None, n=2, nr_angles=16, seed_init=0, PCA=True, M=8, aux_data=_train_aux(golden))
    assert ip.shape == (32, 10)
    assert np.array_equal(ip, gin["inducing_index_points"])
    # n < 1: a random subset of the angles with one vector each (utils.py:713-717); Gaussian init without PCA
    sub = utils.generate_init_inducing_points(None, n=0.5, PCA=True, M=8, aux_data=_train_aux(golden), seed=3)
    assert sub.shape == (8, 10) and len(set(sub[:, 1].tolist())) == 8
    np.random.seed(0)
    g = utils.generate_init_inducing_points(None, n=3, PCA=False, M=8, aux_data=_train_aux(golden))
    assert g.shape == (48, 10) and abs(g[:, 2:].std() - 1.5) < 0.2


def test_sprites_PCA_init_matches_a_plain_numpy_restatement():
    """SPRITES_utils.py:217-279 on a synthetic training dict (the real one needs an external repository): GPLVM action vectors =
    principal components of the per-action mean frames, character part = KDE draws of the global principal components with
    the reference's re-used seed (identical draws for every action), rows [action vector tiled | character vectors]."""
    import importlib
    import scipy.stats
    SU = importlib.import_module("svgp_vae_amd.SPRITES_utils")
    rs = np.random.RandomState(0)
    n_char, N_action, m, La, Lc = 12, 6, 3, 4, 5
    base = rs.rand(n_char, 1, 8, 8, 3).repeat(8, 2).repeat(8, 3)
    act = rs.rand(1, N_action, 8, 8, 3).repeat(8, 2).repeat(8, 3)
    frames = (0.6 * base + 0.4 * act + 0.05 * rs.randn(n_char, N_action, 64, 64, 3)).reshape(-1, 64, 64, 3)
    aux = np.stack([np.repeat(np.arange(n_char), N_action), np.tile(np.arange(N_action), n_char)], 1)
    G, IP = SU.sprites_PCA_init(dict(frames=frames, aux_data=aux), m=m, L_action=La, L_character=Lc, seed=42, N_action=N_action)
    assert G.shape == (N_action, La) and IP.shape == (N_action * m, La + Lc)

    def pca(X, k):                       # centred SVD, components signed so that the largest |loading| is positive (svd_flip)
        Xc = X - X.mean(0)
        U, S, Vt = np.linalg.svd(Xc, full_matrices=False)
        sign = np.sign(U[np.abs(U).argmax(0), np.arange(U.shape[1])])
        return (U * sign)[:, :k] * S[:k]

    means = np.array([frames[aux[:, 1] == a].mean(0).reshape(-1) for a in range(N_action)])
    Gw = pca(means, La)
    # sklearn's default solver for wide data is the RANDOMISED SVD (as in the reference, which sets no random_state): the
    # components agree with the exact SVD to ~1e-2 relative and differ from run to run, so the comparison carries that tolerance
    assert np.allclose(np.abs(G), np.abs(Gw), atol=2e-2 * np.abs(Gw).max())              # (sign convention aside)
    glob = pca(frames.reshape(len(frames), -1), Lc)
    for i in range(N_action):
        rows = IP[i * m:(i + 1) * m]
        assert np.array_equal(rows[:, :La], np.tile(G[i], (m, 1)))
        assert np.array_equal(rows[:, La:], IP[:m, La:])             # the same KDE draws for every action (seed re-used)
    for ax in range(Lc):
        got = IP[:m, La + ax]            # (the sign of a principal axis is a convention: either orientation of the axis)
        wants = [scipy.stats.gaussian_kde(sg * glob[:, ax]).resample(m, seed=42).reshape(-1) for sg in (1.0, -1.0)]
        assert any(np.allclose(got, w, atol=2e-2 * np.abs(glob[:, ax]).max()) for w in wants), ax

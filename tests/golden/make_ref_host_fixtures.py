"""BUILD CONTAINER ONLY.  Runs the reference's own numpy / scipy / sklearn host functions (the parts of /root/reference/utils.py
and SPRITES_utils.py that never touch TensorFlow) and stores their inputs + outputs as tests/golden/ref_host_fixtures.npz.

    python tests/golden/make_ref_host_fixtures.py

TensorFlow, tensorflow_probability, seaborn, matplotlib and the external `load_sprites` module are absent from this image;
they are imported at module level by the reference but not used by the functions called here, so they are replaced in
sys.modules by inert stand-ins for the duration of the import.  Nothing of the reference's source travels: only the arrays
below are committed (the GPU box has no /root/reference), and tests/test_ref_host_fixtures_cpu.py compares the product's
`svgp_vae_amd.utils` / `SPRITES_utils` / `ball` functions with them.

Functions executed (reference file:line):
  utils.generate_init_inducing_points   utils.py:691-744   on MNIST data/eval_data3.p; (n=2, PCA=True), (n=3, PCA=False,
                                                           numpy seeded), (n=0.5, PCA=True, seed=3), remove_test_angle=7
  utils.Make_path_batch                 utils.py:29-56     seed=11
  utils.Make_Video_batch                utils.py:59-121    (it re-seeds numpy from the OS: np.random.seed(None); the fixture
                                                           pins that call to seed 1234 -- the test applies the same pin)
  utils.MSE_rotation                    utils.py:195-245   with VX (diagonal variances) and with full_cholesky=True
  utils.parse_opt_regime                utils.py:891-899
  SPRITES_utils.sprites_PCA_init        SPRITES_utils.py:217-279   on a small synthetic train dict (pickled to a temp file)
  SPRITES_utils.aux_data_sprites_utils  SPRITES_utils.py:317-332
"""
import importlib
import os
import pickle
import sys
import tempfile
import types
from unittest import mock

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_host_fixtures.npz")


class _Inert(types.ModuleType):
    """A module whose every attribute is another inert module / callable: enough for `import x.y as z`, `from x import y`
    and module-level expressions like `tfk = tfp.math.psd_kernels`."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        child = _Inert(self.__name__ + "." + name)
        setattr(self, name, child)
        return child

    def __call__(self, *a, **k):
        return _Inert(self.__name__ + "()")


def import_reference():
    stubs = ["tensorflow", "tensorflow.python", "tensorflow.python.ops", "tensorflow.python.ops.math_ops",
             "tensorflow_probability", "seaborn", "matplotlib", "matplotlib.pyplot", "matplotlib.patches", "load_sprites"]
    for name in stubs:
        sys.modules[name] = _Inert(name)
    sys.path.insert(0, REF)
    try:
        return importlib.import_module("utils"), importlib.import_module("SPRITES_utils")
    finally:
        sys.path.remove(REF)


def synthetic_sprites_dict(seed=0, n_char=12, N_action=6):
    """Character-specific base pattern + action-specific pattern + noise, rows ordered character-major; aux = [char id, action id]."""
    rs = np.random.RandomState(seed)
    base = rs.rand(n_char, 1, 8, 8, 3).repeat(8, 2).repeat(8, 3)
    act = rs.rand(1, N_action, 8, 8, 3).repeat(8, 2).repeat(8, 3)
    frames = (0.6 * base + 0.4 * act + 0.05 * rs.randn(n_char, N_action, 64, 64, 3)).reshape(-1, 64, 64, 3)
    aux = np.stack([np.repeat(np.arange(n_char), N_action), np.tile(np.arange(N_action), n_char)], 1)
    return dict(frames=frames, aux_data=aux)


def main():
    U, SU = import_reference()
    fx = {}
    ev = os.path.join(REF, "MNIST data", "eval_data3.p")
    with open(ev, "rb") as f:
        fx["ip_aux_data"] = np.asarray(pickle.load(f)["aux_data"], dtype=np.float64)       # the input the product test feeds
    fx["ip_n2_pca"] = U.generate_init_inducing_points(ev, n=2, nr_angles=16, seed_init=0, PCA=True, M=8)
    np.random.seed(5)
    fx["ip_n3_gauss_npseed5"] = U.generate_init_inducing_points(ev, n=3, nr_angles=16, PCA=False, M=8)
    fx["ip_nhalf_pca_seed3"] = U.generate_init_inducing_points(ev, n=0.5, nr_angles=16, PCA=True, M=8, seed=3)
    fx["ip_n1_pca_no_angle7"] = U.generate_init_inducing_points(ev, n=1, nr_angles=16, seed_init=100, remove_test_angle=7,
                                                                PCA=True, M=8)

    fx["path_b5_t12_lt3_seed11"] = U.Make_path_batch(batch=5, tmax=12, lt=3, seed=11)
    real_seed = np.random.seed
    with mock.patch("numpy.random.seed", lambda s=None: real_seed(1234 if s is None else s)):
        traj0, vid = U.Make_Video_batch(tmax=10, px=32, py=32, lt=5, batch=4, seed=1, r=3)
    fx["video_traj0"], fx["video_vid"] = traj0, np.asarray(vid, dtype=np.int64)

    rs = np.random.RandomState(2)
    X, Y = rs.randn(3, 9, 2), rs.randn(3, 9, 2)
    VX = rs.rand(3, 9, 2) + 0.1
    X_rot, W, MSE, VX_rot = U.MSE_rotation(X.copy(), Y.copy(), VX.copy())
    fx.update(rot_X=X, rot_Y=Y, rot_VX=VX, rot_X_rot=X_rot, rot_W=W, rot_MSE=np.float64(MSE), rot_VX_rot=VX_rot)
    Lfull = rs.randn(3, 9, 18)
    X_rot2, W2, MSE2, VX_rot2 = U.MSE_rotation(X.copy(), Y.copy(), Lfull.copy(), full_cholesky=True)
    fx.update(rot_fc_L=Lfull, rot_fc_X_rot=X_rot2, rot_fc_W=W2, rot_fc_MSE=np.float64(MSE2), rot_fc_VX_rot=VX_rot2)
    X_rot3, W3, MSE3, VX_rot3 = U.MSE_rotation(X.copy(), Y.copy())
    fx.update(rot_novx_VX_rot=VX_rot3, rot_novx_MSE=np.float64(MSE3))

    n1, r1 = U.parse_opt_regime(["joint-3"])
    n2, r2 = U.parse_opt_regime(["VAE-2", "GP-3", "joint-1"])
    fx["regime_1"] = np.array([str(n1)] + list(r1))
    fx["regime_2"] = np.array([str(n2)] + list(r2))

    d = synthetic_sprites_dict()
    with tempfile.NamedTemporaryFile(suffix=".p", delete=False) as tf_:
        pickle.dump(d, tf_)
    try:
        G, IP = SU.sprites_PCA_init(tf_.name, m=3, L_action=4, L_character=5, seed=42, N_action=6)
    finally:
        os.unlink(tf_.name)
    fx["sprites_pca_GPLVM_action"], fx["sprites_pca_inducing_points"] = G, IP
    seg, rep = SU.aux_data_sprites_utils(24, 8, 8)
    fx["sprites_aux_segment_ids"], fx["sprites_aux_repeats"] = np.asarray(seg), np.asarray(rep)

    np.savez_compressed(OUT, **fx)
    print("wrote", OUT, {k: np.asarray(v).shape for k, v in fx.items()})


if __name__ == "__main__":
    main()

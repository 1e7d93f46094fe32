"""Host-side data utilities of the rotated-MNIST driver, mirroring the reference's `utils.py`
(names and semantics; numpy / torch instead of tf.data):

  import_rotated_mnist(MNIST_path, ending, batch_size)   utils.py:799-875
  generate_init_inducing_points(train_data_path, n, ...) utils.py:691-744
  parse_opt_regime(arr)                                  utils.py:891-899
  gauss_cross_entropy(mu1, var1, mu2, var2)              utils.py:483-504 (element-wise; inside the step it is fused
                                                         into the HIP per-sample kernel, this is the stand-alone form)
  Make_Video_batch / build_video_batch_graph / MSE_rotation   utils.py:59-121,138-192,195-245 (moving ball; ball.py)
"""
import pickle
import random

import numpy as np


def _load(path):
    with open(path, "rb") as f:
        d = pickle.load(f)
    return {"images": np.asarray(d["images"], dtype=np.float64), "aux_data": np.asarray(d["aux_data"], dtype=np.float64)}


def batches(n_rows, batch_size):
    """Un-shuffled `.batch(batch_size)` without drop_remainder (utils.py:846-848): the final short batch
    is used (4050 = 15*256 + 210)."""
    return [(lo, min(lo + batch_size, n_rows)) for lo in range(0, n_rows, batch_size)]


def import_rotated_mnist(MNIST_path, ending, batch_size, train_file=None):
    """Returns (train_data_dict, eval_data_dict, test_data_dict, train_batches).  `train_file` overrides
    `train_data<ending>` (the reference checkout ships without train_data3.p, see .MISSING_LARGE_BLOBS)."""
    train = _load(train_file or (MNIST_path + "train_data" + ending))
    ev = _load(MNIST_path + "eval_data" + ending)
    te = _load(MNIST_path + "test_data" + ending)
    return train, ev, te, batches(len(train["images"]), batch_size)


def generate_init_inducing_points(train_data_path, n=5, nr_angles=16, seed_init=0, remove_test_angle=None,
                                  PCA=False, M=8, seed=0, aux_data=None):
    """utils.py:691-744: for each of `nr_angles` angles draw n object vectors from the empirical (KDE)
    distribution of the train PCA embeddings (PCA=True) or N(0,1.5^2); rows [id, angle, vector].
    n < 1 keeps a random subset of int(n*nr_angles) angles with one vector each."""
    import scipy.stats
    random.seed(seed)
    data = aux_data if aux_data is not None else _load(train_data_path)["aux_data"]
    angles = np.linspace(0, 2 * np.pi, nr_angles + 1)[:-1]
    if n < 1:
        indices = random.sample(list(range(nr_angles)), int(n * nr_angles))
        n = 1
    else:
        indices = range(nr_angles)
    pts = []
    for i in indices:
        if i == remove_test_angle:
            continue
        s = seed_init + i
        if PCA:
            obj = [scipy.stats.gaussian_kde(data[:, ax]).resample(int(n), seed=s) for ax in range(2, 2 + M)]
            obj = np.concatenate(tuple(obj)).T
        else:
            obj = np.random.normal(0, 1.5, int(n) * M).reshape(int(n), M)
        pts.append(np.hstack((np.full((int(n), 1), angles[i]), obj)))
    pts = np.concatenate(tuple(pts))
    return np.hstack((np.array([list(range(len(pts)))]).T, pts))


def parse_opt_regime(arr):
    """utils.py:891-899: ['joint-1000'] -> (1000, ['joint']*1000)."""
    arr = list(arr)
    for i in range(len(arr)):
        regime, nr_epochs = arr[i].split("-")
        arr[i] = (regime, int(nr_epochs))
    training_regime = [r for regime in arr for r in [regime[0]] * regime[1]]
    return len(training_regime), training_regime


def gauss_cross_entropy(mu1, var1, mu2, var2):
    """utils.py:483-504: element-wise cross-entropy H[N(mu1, var1), N(mu2, var2)] =
    -1/2 (log 2 pi + log var2 + (var1 + mu1^2 - 2 mu1 mu2 + mu2^2) / var2).  Runs as `svgp_gauss_cross_entropy` on
    float64 device tensors of any (equal) shape."""
    import torch
    from ._lib import call
    mu1, var1, mu2, var2 = (t.to(torch.float64).contiguous() for t in torch.broadcast_tensors(mu1, var1, mu2, var2))
    if not mu1.is_cuda:
        from ._lib import SvgpError
        raise SvgpError("gauss_cross_entropy needs device tensors; there is no CPU execution path")
    out = torch.empty_like(mu1)
    s = torch.cuda.current_stream(mu1.device).cuda_stream
    call("svgp_gauss_cross_entropy", mu1.numel(), mu1.data_ptr(), var1.data_ptr(), mu2.data_ptr(), var2.data_ptr(),
         out.data_ptr(), s)
    return out


def __getattr__(name):
    # the moving-ball data utilities live beside their device kernels in ball.py (BALL_experiment.py:11-12)
    if name in ("Make_path_batch", "Make_Video_batch", "MSE_rotation", "build_video_batch_graph"):
        from . import ball
        return ball.VideoBatchSource if name == "build_video_batch_graph" else getattr(ball, name)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")

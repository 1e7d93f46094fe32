"""Generates the committed golden fixtures under tests/golden/.

Run in the BUILD container only (needs /root/reference for the reference's DATA files;
no reference code is imported - TF is not installable here, so the expected outputs
come from oracle/svgpvae_oracle.py, the float64 restatement; parity unpinned, see its header).

    python tests/golden/make_golden.py

Inputs taken from the reference's data files (MNIST data/*.p):
  eval_data3.p      640 rotated-3 images + aux rows [id, angle, pca_1..8]
  pca_ov_init3.p    (400, 8) GPLVM/PCA object-vector table (--ov_joint --PCA init)
  train_ids_mask3.p (5400,) presence mask over 360 train ids x 15 train angles
                    (layout: GPVAE_Casale_model.py:24-38) -> the 4050 train aux rows,
                    which `generate_init_inducing_points` (utils.py:691-744) resamples
                    with a per-angle seeded gaussian_kde to initialise the inducing points.
"""
import os
import pickle
import sys

import numpy as np
import scipy.stats
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import svgpvae_oracle as O  # noqa: E402

REF_DATA = "/root/reference/MNIST data/"
DT = torch.float64


def reconstruct_train_aux():
    ov = pickle.load(open(REF_DATA + "pca_ov_init3.p", "rb"))
    mask = pickle.load(open(REF_DATA + "train_ids_mask3.p", "rb"))
    angles16 = np.linspace(0, 2 * np.pi, 17)[:-1]
    test_angle = pickle.load(open(REF_DATA + "test_data3.p", "rb"))["aux_data"][0, 1]
    train_angles = np.array([a for a in angles16 if abs(a - test_angle) > 1e-9])
    assert len(train_angles) == 15 and mask.shape == (360 * 15,)
    rows = []
    for i in range(360):
        for k in range(15):
            if mask[i * 15 + k]:
                rows.append(np.concatenate(([i, train_angles[k]], ov[i])))
    aux = np.array(rows)
    assert aux.shape == (4050, 10)
    return aux


def init_inducing_points(train_aux, n=2, nr_angles=16, seed_init=0, M=8):
    """utils.py:691-744 semantics with PCA=True (KDE resampling, seed = seed_init + angle idx)."""
    angles = np.linspace(0, 2 * np.pi, nr_angles + 1)[:-1]
    pts = []
    for i in range(nr_angles):
        seed = seed_init + i
        obj = [scipy.stats.gaussian_kde(train_aux[:, ax]).resample(int(n), seed=seed)
               for ax in range(2, 2 + M)]
        obj = np.concatenate(tuple(obj)).T
        pts.append(np.hstack((np.full((int(n), 1), angles[i]), obj)))
    pts = np.concatenate(tuple(pts))
    return np.hstack((np.arange(len(pts))[:, None].astype(float), pts))


def main():
    ev = pickle.load(open(REF_DATA + "eval_data3.p", "rb"))
    ov = pickle.load(open(REF_DATA + "pca_ov_init3.p", "rb"))
    images = ev["images"].astype(np.float64)            # (640,28,28,1)
    aux = ev["aux_data"].astype(np.float64)             # (640,10)
    train_aux = reconstruct_train_aux()
    ip = init_inducing_points(train_aux)                # (32,10)
    L = 16
    w = O.glorot_uniform_init(L=L, seed=0)
    eps = np.random.RandomState(1).randn(640, L)
    inputs = dict(images=images, aux=aux, object_vectors=ov, inducing_index_points=ip,
                  l_GP=np.array(1.0), amplitude=np.array(1.0), epsilon=eps,
                  train_aux=train_aux, **{"vae_" + k: v for k, v in w.items()})
    np.savez_compressed(os.path.join(HERE, "mnist_cfg2_inputs.npz"), **inputs)

    params = {k: torch.tensor(v, dtype=DT) for k, v in w.items()}
    params["inducing_index_points"] = torch.tensor(ip, dtype=DT)
    params["l_GP"] = torch.tensor(1.0, dtype=DT)
    params["amplitude"] = torch.tensor(1.0, dtype=DT)
    params["object_vectors"] = torch.tensor(ov, dtype=DT)
    timg, taux, teps = (torch.tensor(x, dtype=DT) for x in (images, aux, eps))
    common = dict(jitter=1e-6, N_train=4050.0, L=L, clipping_qs=True)
    names16 = ["elbo", "recon_loss", "KL_term", "inside_elbo", "ce_term", "p_m", "p_v", "qnet_mu",
               "qnet_var", "recon_images", "inside_elbo_recon", "inside_elbo_kl", "latent_samples",
               "C_ma", "lagrange_mult", "mean_vectors"]
    out = {}
    b = 256
    for mode, GECO in (("beta", False), ("geco", True)):
        kw = dict(beta=0.001, C_ma=torch.zeros((), dtype=DT), lagrange_mult=torch.ones((), dtype=DT),
                  alpha=0.0 if GECO else 0.99, kappa=float(np.sqrt(0.020)), GECO=GECO, **common)
        res, grads = O.loss_and_grads(params, timg[:b], taux[:b], teps[:b], formulation="literal", **kw)
        res_e, grads_e = O.loss_and_grads(params, timg[:b], taux[:b], teps[:b], formulation="efficient", **kw)
        for nm, a, a_e in zip(names16, res, res_e):
            a = torch.as_tensor(a, dtype=DT)
            a_e = torch.as_tensor(a_e, dtype=DT)
            rel = float((a - a_e).abs().max() / (a.abs().max() + 1e-300))
            assert rel < 1e-9, (nm, rel)
            out[f"{mode}_{nm}"] = a.numpy()
        for k in grads:
            rel = float((grads[k] - grads_e[k]).abs().max() / (grads[k].abs().max() + 1e-300))
            assert rel < 1e-7, (k, rel)
            out[f"{mode}_grad_{k}"] = grads[k].numpy()
        # 3-step trajectory: batches 256, 256, 128 (ragged last batch of the 640-row set)
        batches = [(timg[0:256], taux[0:256]), (timg[256:512], taux[256:512]), (timg[512:640], taux[512:640])]
        epss = [teps[0:256], teps[256:512], teps[512:640]]
        log, pfin, m_s, v_s = O.train_trajectory(params, batches, epss, beta=0.001, lr=1e-3, alpha_flag=0.99,
                                                 kappa=float(np.sqrt(0.020)), GECO=GECO, formulation="literal",
                                                 **common)
        for key in log[0]:
            out[f"{mode}_traj_{key}"] = np.array([s[key] for s in log])
        for k in pfin:
            out[f"{mode}_traj_param_{k}"] = pfin[k].numpy()
        print(mode, "elbo", float(res[0]), "traj", [s["elbo"] for s in log])
    np.savez_compressed(os.path.join(HERE, "mnist_cfg2_outputs.npz"), **out)
    for f in ("mnist_cfg2_inputs.npz", "mnist_cfg2_outputs.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)) / 1e6, "MB")


if __name__ == "__main__":
    main()

"""Deep SVIGP_Hensman baseline on rotated MNIST (`MNIST_experiment.py --elbo SVIGP_Hensman`), restated for CPU.

TEST INFRASTRUCTURE ONLY - imported by tests/, never by the product path.  Parity unpinned (TensorFlow 1.15 / TFP 0.8
not installable, no reference tests); pinned by the literal == efficient identity in tests/test_svigp_oracle.py.

Literal restatement of
  SVIGP_Hensman.__init__ / variational_loss / approximate_posterior_params     SVIGP_Hensman_model.py:14-227
  forward_pass_deep_SVIGP_Hensman / predict_deep_SVIGP_Hensman                 SVIGP_Hensman_model.py:230-339
  SVIGP_Hensman_decoder (= the mnistVAE decoder architecture)                   VAE_utils.py:394-431
  gradients of -elbo + TF1 Adam                                                 MNIST_experiment.py:633-640
The kernel is mnistSVGP's (SVIGP_Hensman_model.py:79-125 == SVGPVAE_model.py:427-476).
"""
import math

import torch

from .svgpvae_oracle import DT, MnistSVGP, MnistVAE, add_diagonal_jitter

LOG_2PI = math.log(2 * math.pi)


class SVIGPHensman(MnistSVGP):
    def __init__(self, inducing_index_points, object_vectors, l_GP, amplitude, jitter, N_train, L, loc, scale, noise,
                 K_obj_normalize=False):
        super().__init__(False, inducing_index_points, object_vectors, l_GP, amplitude, jitter, N_train, K_obj_normalize)
        self.L = L
        self.loc, self.scale, self.noise = loc, scale, noise         # (L,m), (L,m,m), scalar
        self.cov_mat = scale @ scale.transpose(1, 2)                 # :70-71

    def variational_loss(self, x, z, lat_channel):
        """:135-198 -> (L_3_sum_term, KL_term, mean_vector)."""
        ip = self.inducing_index_points
        m = float(ip.shape[0])
        K_mm = self.kernel_matrix(ip, ip)
        K_mm_inv = torch.linalg.inv(add_diagonal_jitter(K_mm, self.jitter))
        K_nn = self.kernel_matrix(x, x, False, False, diag_only=True)
        K_nm = self.kernel_matrix(x, ip, x_inducing=False)
        K_mn = K_nm.T
        loc, S = self.loc[lat_channel], self.cov_mat[lat_channel]
        mean_vector = K_nm @ (K_mm_inv @ loc)
        K_mm_chol = torch.linalg.cholesky(add_diagonal_jitter(K_mm, self.jitter))
        S_chol = torch.linalg.cholesky(add_diagonal_jitter(S, self.jitter))
        KL = 0.5 * (2 * torch.log(torch.diagonal(K_mm_chol)).sum() - 2 * torch.log(torch.diagonal(S_chol)).sum() - m +
                    torch.trace(K_mm_inv @ S) + (loc * (K_mm_inv @ loc)).sum())
        precision = 1 / self.noise
        K_tilde = precision * (K_nn - torch.diagonal(K_nm @ (K_mm_inv @ K_mn)))
        lam = K_nm[:, :, None] @ K_nm[:, None, :]
        lam = K_mm_inv @ (lam @ K_mm_inv)
        trace_terms = precision * torch.diagonal(S @ lam, dim1=-2, dim2=-1).sum(-1)
        return -0.5 * (K_tilde.sum() + trace_terms.sum()), KL, mean_vector

    def approximate_posterior_params(self, index_points_test, lat_channel):
        """:200-227 -> (mean, B diag)."""
        ip = self.inducing_index_points
        K_mm = self.kernel_matrix(ip, ip)
        K_mm_inv = torch.linalg.inv(add_diagonal_jitter(K_mm, self.jitter))
        K_xx = self.kernel_matrix(index_points_test, index_points_test, False, False, diag_only=True)
        K_xm = self.kernel_matrix(index_points_test, ip, x_inducing=False)
        A = K_xm @ K_mm_inv
        mean = A @ self.loc[lat_channel]
        # the reference subtracts an (x, x) matrix from the (x) diagonal by broadcasting (:225); only its diagonal is
        # meaningful and nothing downstream reads B
        Bm = K_xx - torch.diagonal(A @ ((K_mm - self.cov_mat[lat_channel]) @ A.T))
        return mean, Bm


def forward_pass_deep_SVIGP_Hensman(data_batch, vae, svgp):
    """:230-289 -> (elbo, recon_loss, KL_term, inside_elbo, recon_images, inside_elbo_recon, inside_elbo_kl, mean_vectors)."""
    images, aux = data_batch
    b = float(images.shape[0])
    K = float(images.shape[1] * images.shape[2] * images.shape[3])
    rec, kl, means = [], [], []
    for l in range(svgp.L):
        r_l, k_l, mean_l = svgp.variational_loss(aux, None, l)
        rec.append(r_l); kl.append(k_l); means.append(mean_l)
    inside_recon, inside_kl = torch.stack(rec).sum(), torch.stack(kl).sum()
    inside = inside_recon - (b / svgp.N_train) * inside_kl
    mean_vectors = torch.stack(means, 1)
    recon = vae.decode(mean_vectors)
    recon_loss = ((images - recon) ** 2).sum()
    elbo = -b * K * torch.log(svgp.noise) - 0.5 * b * K * LOG_2PI - (0.5 * svgp.noise ** (-2)) * recon_loss + inside
    return elbo, recon_loss / K, inside, inside, recon, inside_recon, inside_kl, mean_vectors


def predict_deep_SVIGP_Hensman(test_data_batch, vae, svgp):
    """:292-339 -> (recon_images_test, recon_loss per pixel count w*h)."""
    images, aux = test_data_batch
    p_m = torch.stack([svgp.approximate_posterior_params(aux, l)[0] for l in range(svgp.L)], 1)
    recon = vae.decode(p_m)
    return recon, ((images - recon) ** 2).sum() / float(images.shape[1] * images.shape[2])


GP_KEYS = ("inducing_index_points", "l_GP", "amplitude", "object_vectors", "loc", "scale", "noise")


def make_models(params, jitter, N_train, L, K_obj_normalize=False):
    vae = MnistVAE(params, L=L)
    svgp = SVIGPHensman(params["inducing_index_points"], params.get("object_vectors"), params["l_GP"], params["amplitude"],
                        jitter, N_train, L, params["loc"], params["scale"], params["noise"], K_obj_normalize)
    return vae, svgp


def loss_and_grads(params, images, aux, *, jitter, N_train, L, K_obj_normalize=False):
    """-elbo and its gradients wrt every parameter that takes part (decoder, GP parameters, variational parameters,
    likelihood noise); encoder entries of `params` are ignored (the SVIGP decoder object has none)."""
    leaf = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
    vae, svgp = make_models(leaf, jitter, N_train, L, K_obj_normalize)
    out = forward_pass_deep_SVIGP_Hensman((images, aux), vae, svgp)
    keys = [k for k in leaf if k.startswith("dec_") or k in GP_KEYS]
    gs = torch.autograd.grad(-out[0], [leaf[k] for k in keys], allow_unused=True)
    grads = {k: (torch.zeros_like(leaf[k]) if g is None else g) for k, g in zip(keys, gs)}
    return tuple(o.detach() for o in out), grads


def efficient_terms(K, Kn, knn, loc, scale, noise, jitter):
    """The O(b m^2 + L m^3) form the HIP path implements: W = K_nm Ki, H = W^T W, S_l = A_l A_l^T.
    Returns mean_vectors (b,L), sum_l L_3, sum_l KL."""
    m, L = K.shape[0], loc.shape[0]
    eye = torch.eye(m, dtype=DT)
    Kj = K + jitter * eye
    Ki = torch.linalg.inv(Kj)
    ldK = 2 * torch.log(torch.diagonal(torch.linalg.cholesky(Kj))).sum()
    W = Kn @ Ki
    q = (W * Kn).sum(1)
    S = scale @ scale.transpose(1, 2)
    H = W.T @ W
    p = 1 / noise
    L3 = -0.5 * p * (L * (knn - q).sum() + (S.sum(0) * H).sum())
    ldS = 2 * torch.log(torch.diagonal(torch.linalg.cholesky(S + jitter * eye[None]), dim1=-2, dim2=-1)).sum(-1)
    U = loc @ Ki.T
    KL = 0.5 * (L * ldK - ldS.sum() - L * m + (Ki[None] * S).sum() + (loc * U).sum())
    return W @ loc.T, L3, KL

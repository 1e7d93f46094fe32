"""CPU oracle for the SPRITES SVGPVAE_Hensman step.

TEST INFRASTRUCTURE ONLY; parity unpinned (see oracle/svgpvae_oracle.py header: TF/TFP not installable,
the reference has no tests).  The reference runs SPRITES in float32 (VAE_utils.py:277,365;
SVGPVAE_model.py:516); this restatement is float64 - a superset precision - so the HIP float64 path can be
compared tightly; the float32 reference is reproduced within float32 rounding by construction.

Restates: spritesSVGP.kernel_matrix (SVGPVAE_model.py:550-600, kernels :530-548), spritesVAE
(VAE_utils.py:275-360), sprites_representation_network (:363-391), aux_data_SVGPVAE_sprites
(SVGPVAE_model.py:1086-1115), aux_data_sprites_utils (SPRITES_utils.py:317-332), the repr_NN branch of
forward_pass_SVGPVAE (:861-863, 891-892), gradient clipping (SPRITES_experiment.py:234-235).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import svgpvae_oracle as O

DT = torch.float64


def exponentiated_quadratic(x, y, amplitude, length_scale, diag_only=False):
    """TFP ExponentiatedQuadratic: a^2 exp(-||x-y||^2 / (2 l^2))."""
    if diag_only:
        d2 = ((x - y) ** 2).sum(1)
    else:
        d2 = ((x[:, None, :] - y[None, :, :]) ** 2).sum(-1)
    return amplitude ** 2 * torch.exp(-d2 / (2 * length_scale ** 2))


class SpritesSVGP(O.MnistSVGP):
    """mainSVGP + spritesSVGP.kernel_matrix (SVGPVAE_model.py:487-600)."""

    def __init__(self, inducing_index_points, GPLVM_action, jitter, N_train, L_action, K_obj_normalize=False,
                 K_SE=False, se_params=None, titsias=False):
        super().__init__(titsias, inducing_index_points, None, None, None, jitter, N_train, K_obj_normalize)
        self.GPLVM_action = GPLVM_action
        self.L_action = L_action
        self.K_SE = K_SE
        self.se = se_params            # dict l_action, sigma_action, l_character, sigma_character

    def kernel_matrix(self, x, y, x_inducing=True, y_inducing=True, diag_only=False):
        La = self.L_action
        xa, xc = (x[:, :La], x[:, La:]) if x_inducing else (self.GPLVM_action[x[:, 0].detach().long()], x[:, 1:])
        ya, yc = (y[:, :La], y[:, La:]) if y_inducing else (self.GPLVM_action[y[:, 0].detach().long()], y[:, 1:])
        if self.K_SE:
            ka = exponentiated_quadratic(xa, ya, self.se["sigma_action"], self.se["l_action"], diag_only)
            kc = exponentiated_quadratic(xc, yc, self.se["sigma_character"], self.se["l_character"], diag_only)
        else:
            ka = O.linear_kernel(xa, ya, self.K_obj_normalize, diag_only)
            kc = O.linear_kernel(xc, yc, self.K_obj_normalize, diag_only)
        return ka * kc


SPRITES_VAE_SHAPES = lambda L: (
    [(f"enc_c{i}_w", (3, 3, 3 if i == 1 else 16, 16)) for i in range(1, 7)]
    + [("enc_d_w", (1024, 2 * L)), ("dec_d_w", (L, 1024))]
    + [(f"dec_c{i}_w", (3, 3, 16, 16 if i < 7 else 3)) for i in range(1, 8)])
ENC_STRIDES = (1, 2, 1, 2, 1, 2)
DEC_UP = (True, False, True, False, True, False, False)


def sprites_param_shapes(L, L_character=16):
    shp = []
    for name, s in SPRITES_VAE_SHAPES(L):
        shp.append((name, s))
        shp.append((name[:-1] + "b", (s[-1],)))
    for i, cin in ((1, 3), (2, L_character), (3, L_character)):
        shp.append((f"repr_c{i}_w", (2, 2, cin, L_character)))
        shp.append((f"repr_c{i}_b", (L_character,)))
    return shp


def glorot_init(L, L_character=16, seed=0):
    rng = np.random.RandomState(seed)
    out = {}
    for name, shp in sprites_param_shapes(L, L_character):
        if name.endswith("_b"):
            out[name] = np.zeros(shp)
            continue
        rf = shp[0] * shp[1] if len(shp) == 4 else 1
        fi, fo = (rf * shp[2], rf * shp[3]) if len(shp) == 4 else shp
        lim = math.sqrt(6.0 / (fi + fo))
        out[name] = rng.uniform(-lim, lim, size=shp)
    return out


class SpritesVAE:
    """VAE_utils.py:275-360."""

    def __init__(self, p, L):
        self.p, self.L = p, L

    def encode(self, images):
        h = images
        for i, s in enumerate(ENC_STRIDES, 1):
            h = F.elu(O._conv2d_nhwc(h, self.p[f"enc_c{i}_w"], self.p[f"enc_c{i}_b"], s, "same"))
        h = h.reshape(h.shape[0], -1)
        enc = h @ self.p["enc_d_w"] + self.p["enc_d_b"]
        return enc[:, :self.L], torch.exp(enc[:, self.L:])

    def decode(self, z):
        h = (z @ self.p["dec_d_w"] + self.p["dec_d_b"]).reshape(-1, 8, 8, 16)
        for i, up in enumerate(DEC_UP, 1):
            if up:
                h = O._upsample2_nhwc(h)
            h = F.elu(O._conv2d_nhwc(h, self.p[f"dec_c{i}_w"], self.p[f"dec_c{i}_b"], 1, "same"))
        return h


def repr_nn(p, images):
    """sprites_representation_network (VAE_utils.py:375-391): 3 x Conv2D(k=2, s=2, same, elu),
    AveragePooling2D(8, 'same') on the 8x8 map, Flatten."""
    h = images
    for i in (1, 2, 3):
        h = F.elu(O._conv2d_nhwc(h, p[f"repr_c{i}_w"], p[f"repr_c{i}_b"], 2, "same"))
    return h.mean(dim=(1, 2))


def pretrain_repr_nn_trajectory(params, frames, char_ids, *, nr_epochs, lr, batch_size, n_classes=1000, seed=0):
    """Pre-training of the representation network (SPRITES_experiment.py:139-151,214-224,325-357; SPRITES_utils.py:335-368):
    per batch loss = mean sparse softmax cross-entropy of Dense(n_classes)(repr_nn(frames)) against the character ids, TF1 Adam
    (ONE optimiser: step counter t = number of updates so far) on the representation network + the classification layer
    (Keras Dense default: glorot-uniform kernel, zero bias).  Un-shuffled batches (the in-batch shuffle of :346-351 permutes
    the rows of a mean), incomplete last batch dropped as the product does.  Returns (per-epoch (mean loss, accuracy),
    final repr_* parameters, dense kernel, dense bias, first moments, second moments of the repr_* parameters)."""
    names = [k for k in params if k.startswith("repr_")]
    p = {k: params[k].clone().requires_grad_(True) for k in names}
    Lc = p["repr_c3_b"].shape[0]
    lim = math.sqrt(6.0 / (Lc + n_classes))
    p["dense_w"] = torch.tensor(np.random.RandomState(seed).uniform(-lim, lim, (Lc, n_classes)), dtype=DT).requires_grad_(True)
    p["dense_b"] = torch.zeros(n_classes, dtype=DT).requires_grad_(True)
    ms = {k: torch.zeros_like(v) for k, v in p.items()}
    vs = {k: torch.zeros_like(v) for k, v in p.items()}
    n, t, hist = frames.shape[0], 0, []
    for _ in range(nr_epochs):
        tot, correct, seen, nb = 0.0, 0, 0, 0
        for lo in range(0, n - batch_size + 1, batch_size):
            x, lab = frames[lo:lo + batch_size], char_ids[lo:lo + batch_size].long()
            logits = repr_nn(p, x) @ p["dense_w"] + p["dense_b"]
            loss = F.cross_entropy(logits, lab)
            grads = dict(zip(p, torch.autograd.grad(loss, list(p.values()))))
            t += 1
            with torch.no_grad():
                O.adam_tf1_step(p, grads, ms, vs, t, lr)
            tot += float(loss); nb += 1; seen += batch_size
            correct += int((logits.argmax(1) == lab).sum())
        hist.append((tot / max(nb, 1), correct / max(seen, 1)))
    out = {k: v.detach() for k, v in p.items()}
    return hist, {k: out[k] for k in names}, out["dense_w"], out["dense_b"], {k: ms[k] for k in names}, {k: vs[k] for k in names}


def aux_data_sprites_utils(batch_size, N, repeats):
    """SPRITES_utils.py:317-332."""
    n_char = int(batch_size / N)
    return np.array([[i] * N for i in range(n_char)]).reshape(-1), [repeats for _ in range(n_char)]


def aux_data_SVGPVAE_sprites(data_batch, p, segment_ids, repeats):
    """SVGPVAE_model.py:1086-1115: repr CNN, segment_mean per character, repeat, prepend action id."""
    images, action_IDs = data_batch
    cv = repr_nn(p, images)
    seg = torch.as_tensor(segment_ids)
    n_seg = int(seg.max()) + 1
    means = torch.stack([cv[seg == g].mean(0) for g in range(n_seg)])
    cv = torch.repeat_interleave(means, torch.as_tensor(repeats), dim=0)
    return torch.cat([action_IDs.to(DT)[:, None], cv], dim=1)


def forward_pass_SVGPVAE_sprites(data_batch, beta, params, gp, C_ma, lagrange_mult, alpha, kappa, *, L,
                                 segment_ids, repeats, clipping_qs=False, GECO=False, epsilon=None,
                                 formulation="efficient", titsias=False):
    """forward_pass_SVGPVAE with repr_NN set (SVGPVAE_model.py:823-936): computed aux data, p_v clipped to
    [1e-4, 100] (:891-892).  `gp` = dict(ip, GPLVM_action, jitter, N_train, L_action, K_obj_normalize, K_SE, se)."""
    images, action_ids = data_batch
    _, w, h, c = images.shape
    Kpix = float(w * h * c)
    b = float(images.shape[0])
    vae = SpritesVAE(params, L)
    svgp = SpritesSVGP(gp["ip"], gp["GPLVM_action"], gp["jitter"], gp["N_train"], gp["L_action"],
                       gp.get("K_obj_normalize", False), gp.get("K_SE", False), gp.get("se"), titsias=titsias)
    if titsias:
        formulation = "literal"        # the b x b branch of variational_loss (:246-259)
    qnet_mu, qnet_var = vae.encode(images)
    if clipping_qs:
        qnet_var = O.clip_by_value(qnet_var, 1e-3, 10.0)
    aux = aux_data_SVGPVAE_sprites(data_batch, params, segment_ids, repeats)
    ip = svgp.inducing_index_points
    if formulation == "literal":
        rec, kl, pm, pv = [], [], [], []
        for l in range(L):
            a, bb, mu_hat, A_hat = svgp.approximate_posterior_params(aux, aux, qnet_mu[:, l], qnet_var[:, l])
            r_, k_ = svgp.variational_loss(aux, qnet_mu[:, l], mu_hat, A_hat, qnet_var[:, l])
            rec.append(r_); kl.append(k_); pm.append(a); pv.append(bb)
        inside_recon, inside_kl = torch.stack(rec).sum(), torch.stack(kl).sum()
        p_m, p_v = torch.stack(pm, 1), torch.stack(pv, 1)
    else:
        K = svgp.kernel_matrix(ip, ip)
        Kn = svgp.kernel_matrix(aux, ip, x_inducing=False)
        knn = svgp.kernel_matrix(aux, aux, False, False, diag_only=True)
        p_m, p_v, L3, KL = O.gp_block_efficient(K, Kn, knn, qnet_mu, qnet_var, gp["jitter"], gp["N_train"])
        inside_recon, inside_kl = L3.sum(), KL.sum()
    inside_elbo = inside_recon - inside_kl if titsias else inside_recon - (b / gp["N_train"]) * inside_kl   # :882-885
    p_v = O.clip_by_value(p_v, 1e-4, 100.0)
    ce_term = O.gauss_cross_entropy(p_m, p_v, qnet_mu, qnet_var).sum()
    KL_term = -ce_term + inside_elbo
    if epsilon is None:
        epsilon = torch.randn(p_m.shape, dtype=DT)
    z = p_m + epsilon * torch.sqrt(p_v)
    recon = vae.decode(z)
    if GECO:
        recon_loss = torch.sum(torch.mean((images - recon) ** 2, dim=(1, 2, 3)) - kappa ** 2)
        C_ma = alpha * C_ma + (1 - alpha) * recon_loss / b
        elbo = -KL_term + lagrange_mult * (recon_loss / b + (C_ma - recon_loss / b).detach())
        lagrange_mult = lagrange_mult * torch.exp(C_ma)
    else:
        recon_loss = torch.sum((images - recon) ** 2) / Kpix
        elbo = -recon_loss + (beta / float(L)) * KL_term
    return (elbo, recon_loss, KL_term, inside_elbo, ce_term, p_m, p_v, qnet_mu, qnet_var, recon, inside_recon,
            inside_kl, z, C_ma, lagrange_mult, aux)


def loss_and_grads(params, gp_params, data_batch, epsilon, *, beta, C_ma, lagrange_mult, alpha, kappa, L, L_action,
                   jitter, N_train, segment_ids, repeats, clipping_qs=False, GECO=False, K_obj_normalize=False,
                   K_SE=False, clip_grad=None, formulation="efficient", titsias=False):
    """Gradients of the minimised objective w.r.t. all network parameters and GP parameters
    (inducing points, GPLVM action table, SE hyper-parameters when K_SE); optional element-wise clipping."""
    leaf = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
    gleaf = {k: v.detach().clone().requires_grad_(True) for k, v in gp_params.items()}
    se = {k: gleaf[k] for k in ("l_action", "sigma_action", "l_character", "sigma_character")} if K_SE else None
    gp = dict(ip=gleaf["inducing_index_points"], GPLVM_action=gleaf["GPLVM_action"], jitter=jitter, N_train=N_train,
              L_action=L_action, K_obj_normalize=K_obj_normalize, K_SE=K_SE, se=se)
    out = forward_pass_SVGPVAE_sprites(data_batch, beta, leaf, gp, C_ma, lagrange_mult, alpha, kappa, L=L,
                                       segment_ids=segment_ids, repeats=repeats, clipping_qs=clipping_qs, GECO=GECO,
                                       epsilon=epsilon, formulation=formulation, titsias=titsias)
    objective = out[0] if GECO else -out[0]
    names = list(leaf) + list(gleaf)
    tensors = [leaf[k] for k in leaf] + [gleaf[k] for k in gleaf]
    gs = torch.autograd.grad(objective, tensors, allow_unused=True)
    grads = {k: (torch.zeros_like(t) if g is None else g) for k, t, g in zip(names, tensors, gs)}
    if clip_grad is not None:
        grads = {k: torch.clamp(g, -clip_grad, clip_grad) for k, g in grads.items()}
    return tuple(o.detach() if torch.is_tensor(o) else o for o in out), grads


# ---------------------------------------------------------------------------------------------
# conditional generation for a test character (test infrastructure, as everything in oracle/)
# ---------------------------------------------------------------------------------------------
def precompute_GP_params_SVGPVAE(means, vars_, aux_data, svgp):
    """SVGPVAE_model.py:989-1023, literal: per channel Sigma_l = K_mm + K_mn (K_nm / var_l), inverse WITHOUT jitter."""
    K_mm = svgp.kernel_matrix(svgp.inducing_index_points, svgp.inducing_index_points)
    K_nm = svgp.kernel_matrix(aux_data, svgp.inducing_index_points, x_inducing=False)
    mean_terms, inv = [], []
    for l in range(means.shape[1]):
        p = O.reciprocal_no_nan(vars_[:, l])
        Sigma_l = K_mm + K_nm.T @ (K_nm * p[:, None])
        Si = torch.linalg.inv(Sigma_l)
        mean_terms.append(Si @ (K_nm.T @ (p * means[:, l])))
        inv.append(Si)
    return torch.stack(mean_terms), torch.stack(inv)


def approximate_posterior_params_precomputed(svgp, index_points, mean_term, sigma_term, K_mm_inv):
    """SVGPVAE_model.py:610-635 for one channel."""
    K_bb = svgp.kernel_matrix(index_points, index_points, False, False, diag_only=True)
    K_bm = svgp.kernel_matrix(index_points, svgp.inducing_index_points, x_inducing=False)
    mean_vector = K_bm @ mean_term
    B = K_bb + torch.diagonal(-K_bm @ (K_mm_inv @ K_bm.T) + K_bm @ (sigma_term @ K_bm.T))
    return mean_vector, B


def predict_SVGPVAE_sprites_test_character(data_batch, params, svgp, mean_terms, var_terms, N_context, N_actions,
                                           batch_size_test, segment_ids, repeats, K_mm_inv, epsilon, L, context_draw=None):
    """SVGPVAE_model.py:1118-1195 with the N(0,1) draw as an input.  context_draw None: context_full_actions=True (:1146-1148);
    else (n_characters, N_context) offsets standing for the np.random.choice draw of context_full_actions=False (:1149-1151)."""
    images, action_ids = data_batch
    if context_draw is None:
        context = np.sort(np.array([list(range(i * N_actions, i * N_actions + N_context))
                                    for i in range(int(batch_size_test / N_actions))]).reshape(-1))
    else:
        context = np.sort(np.array([list(i * N_actions + np.asarray(context_draw[i]))
                                    for i in range(int(batch_size_test / N_actions))]).reshape(-1))
    target = np.array([x for x in range(batch_size_test) if x not in set(context.tolist())])
    images_context, images_t = images[context], images[target]
    aux_t = aux_data_SVGPVAE_sprites((images_context, action_ids[target]), params, segment_ids, repeats)
    p_m, p_v = [], []
    for l in range(L):
        a, b_ = approximate_posterior_params_precomputed(svgp, aux_t, mean_terms[l], var_terms[l], K_mm_inv)
        p_m.append(a); p_v.append(b_)
    p_m, p_v = torch.stack(p_m, 1), torch.stack(p_v, 1)
    p_v = torch.clamp(p_v, 1e-4, 100.0)
    z = p_m + epsilon * torch.sqrt(p_v)
    recon = SpritesVAE(params, L).decode(z)
    return recon, images_t, torch.sum((images_t - recon) ** 2) / float(64 * 64 * 3), p_m, p_v, aux_t

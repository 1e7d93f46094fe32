"""Stage-by-stage float64 restatement of the Hensman GP block and its hand-derived
reverse pass, cut exactly where the HIP kernels are cut (svgp-vae_amd/csrc).

TEST INFRASTRUCTURE ONLY (see oracle/svgpvae_oracle.py header; parity unpinned).
Each stage names the HIP entry point it checks.  The backward formulas are validated
against torch.autograd of `svgpvae_oracle.gp_block_efficient` in tests/test_oracle_kat.py.

Math: SURVEY.md Appendix A / I; reference lines SVGPVAE_model.py:220-343 (GP block),
:427-476 (kernel), :880-902 (assembly), utils.py:483-504 (cross entropy).
"""
import math

import torch

from .svgpvae_oracle import DT, LOG_2PI, reciprocal_no_nan


# ------------------------------------------------------------------ kernel matrices
def kernel_matrix_fwd(aux, ip, ov, ls, amp, normalize=False):
    """svgp_kernel_matrix_fwd: K_mm (m,m), K_nm (b,m), k_nn (b).  aux (b,2+M) batch rows
    [id, angle, pca...]; ip (m,2+M) inducing rows; ov (n_obj,M) GPLVM table or None."""
    th_n, th_m = aux[:, 1], ip[:, 1]
    o_m = ip[:, 2:]
    o_n = aux[:, 2:] if ov is None else ov[aux[:, 0].long()]
    if normalize:
        o_mh = o_m / torch.linalg.norm(o_m, dim=1, keepdim=True)
        o_nh = o_n / torch.linalg.norm(o_n, dim=1, keepdim=True)
    else:
        o_mh, o_nh = o_m, o_n

    def view(d):
        s = torch.sin(0.5 * d)
        return amp ** 2 * torch.exp(-2.0 * s * s / ls ** 2)

    K = view(th_m[:, None] - th_m[None, :]) * (o_mh @ o_mh.T)
    Kn = view(th_n[:, None] - th_m[None, :]) * (o_nh @ o_mh.T)
    knn = amp ** 2 * torch.sum(o_nh * o_nh, dim=1)
    return K, Kn, knn


def kernel_matrix_bwd(aux, ip, ov, ls, amp, gK, gKn, gknn, normalize=False):
    """svgp_kernel_matrix_bwd: VJP of kernel_matrix_fwd.  gK (m,m) is used as-is (not
    symmetrised).  Returns d_ip (m,2+M) (id column zero), d_ls, d_amp, d_ov (n_obj,M) or None."""
    m = ip.shape[0]
    th_n, th_m = aux[:, 1], ip[:, 1]
    o_m = ip[:, 2:]
    ids = aux[:, 0].long()
    o_n = aux[:, 2:] if ov is None else ov[ids]
    nm = torch.linalg.norm(o_m, dim=1, keepdim=True)
    nn_ = torch.linalg.norm(o_n, dim=1, keepdim=True)
    o_mh, o_nh = (o_m / nm, o_n / nn_) if normalize else (o_m, o_n)
    a2, l2 = amp ** 2, ls ** 2

    def parts(d, oa, ob):
        s = torch.sin(0.5 * d)
        V = a2 * torch.exp(-2.0 * s * s / l2)     # view kernel
        D = oa @ ob.T                             # object kernel
        return V, D, torch.sin(d), s * s

    # ---- K_mm (both arguments are inducing rows)
    V, D, sd, s2h = parts(th_m[:, None] - th_m[None, :], o_mh, o_mh)
    Kmat = V * D
    GK = gK * Kmat
    d_amp = 2.0 * GK.sum() / amp
    d_ls = (GK * 4.0 * s2h).sum() / ls ** 3
    T = GK * sd / l2
    d_th = -T.sum(1) + T.sum(0)
    GV = gK * V
    dD_a = GV @ o_mh          # d/d(o_mh row i) from first argument
    dD_b = GV.T @ o_mh        # second argument
    d_omh = dD_a + dD_b
    # ---- K_nm
    Vn, Dn, sdn, s2hn = parts(th_n[:, None] - th_m[None, :], o_nh, o_mh)
    Knm = Vn * Dn
    GKn = gKn * Knm
    d_amp = d_amp + 2.0 * GKn.sum() / amp
    d_ls = d_ls + (GKn * 4.0 * s2hn).sum() / ls ** 3
    d_th = d_th + (GKn * sdn / l2).sum(0)
    GVn = gKn * Vn
    d_omh = d_omh + GVn.T @ o_nh
    d_onh = GVn @ o_mh
    # ---- k_nn = a^2 |o_nh|^2
    knn = a2 * torch.sum(o_nh * o_nh, dim=1)
    d_amp = d_amp + 2.0 * (gknn * knn).sum() / amp
    d_onh = d_onh + 2.0 * a2 * gknn[:, None] * o_nh
    if normalize:  # o_h = o/|o|: d_o = (d_oh - <d_oh,o_h> o_h)/|o|
        d_om = (d_omh - (d_omh * o_mh).sum(1, keepdim=True) * o_mh) / nm
        d_on = (d_onh - (d_onh * o_nh).sum(1, keepdim=True) * o_nh) / nn_
    else:
        d_om, d_on = d_omh, d_onh
    d_ip = torch.zeros_like(ip)
    d_ip[:, 1] = d_th
    d_ip[:, 2:] = d_om
    d_ov = None
    if ov is not None:
        d_ov = torch.zeros_like(ov)
        d_ov.index_add_(0, ids, d_on)
    return d_ip, d_ls, d_amp, d_ov


# ------------------------------------------------------------------ weighted statistics
def gp_stats(Kn, w, a, bvec=None):
    """svgp_gp_stats: S[l] = Kn^T diag(w[:,l]) Kn (L,m,m); v1[l] = Kn^T a[:,l]; v2[l] = Kn^T b[:,l].
    Forward: w = p, a = p*y.  Backward: w = g_pv, a = mv_bar, b = c*g_pm.  These are the only
    cross-row reductions of the block, i.e. the tensors all-reduced under data parallelism."""
    S = torch.einsum('nl,ni,nj->lij', w, Kn, Kn)
    v1 = torch.einsum('nl,ni->li', a, Kn)
    v2 = None if bvec is None else torch.einsum('nl,ni->li', bvec, Kn)
    return S, v1, v2


# ------------------------------------------------------------------ m x m factor stage
def gp_factor_fwd(K, S, v, jitter, c, kl_form=0):
    """svgp_gp_factor_fwd: shared Ki, ldK and per-channel Si, t, mu_hat, A_hat, G, Aji, u, KL.
    kl_form 1 (moving-ball SVGP, SVGPVAE_model.py:135-137): KL's last summand is L tr(Ki A_l A_l) (stored as klq)."""
    m = K.shape[0]
    eye = torch.eye(m, dtype=DT)
    Kj = K + jitter * eye
    Ki = torch.linalg.inv(Kj)
    ldK = 2 * torch.log(torch.diagonal(torch.linalg.cholesky(Kj))).sum()
    Si = torch.linalg.inv(K[None] + c * S + jitter * eye[None])
    t = torch.einsum('lij,lj->li', Si, v)
    G = Si @ K[None]
    A = K[None] @ G
    mu = c * (t @ K.T)
    u = mu @ Ki.T
    Aj = A + jitter * eye[None]
    Aji = torch.linalg.inv(Aj)
    ldA = 2 * torch.log(torch.diagonal(torch.linalg.cholesky(Aj), dim1=-2, dim2=-1)).sum(-1)
    klq = torch.einsum('ij,ljk,lki->l', Ki, A, A)
    last = (mu * u).sum(1) if kl_form == 0 else S.shape[0] * klq
    KL = 0.5 * (ldK - ldA - m + torch.einsum('ij,lji->l', Ki, A) + last)
    return dict(Ki=Ki, ldK=ldK, Si=Si, t=t, G=G, A=A, mu=mu, u=u, Aji=Aji, ldA=ldA, KL=KL, klq=klq)


# ------------------------------------------------------------------ per-sample stage
def gp_posterior_fwd(Kn, knn, y, s2, eps, f, c):
    """svgp_gp_posterior_hensman_fwd: per (n,l) posterior moments, L3/CE integrands, sample z."""
    p = reciprocal_no_nan(s2)
    W = Kn @ f['Ki']
    q = (W * Kn).sum(1)
    p_m = c * (Kn @ f['t'].T)
    r = torch.einsum('ni,lij,nj->nl', Kn, f['Si'], Kn)
    p_v = (knn - q)[:, None] + r
    mv = Kn @ f['u'].T
    s = torch.einsum('ni,lij,nj->nl', W, f['A'], W)
    e = y - mv
    d = (knn - q)[:, None] + s + e * e                       # L3 integrand (times p)
    L3 = -0.5 * ((p * d).sum(0) + torch.log(s2).sum(0) + Kn.shape[0] * LOG_2PI)
    ce = -0.5 * (LOG_2PI + torch.log(s2) + (p_v + (p_m - y) ** 2) * p)
    z = p_m + eps * torch.sqrt(p_v)
    return dict(p=p, W=W, q=q, p_m=p_m, p_v=p_v, mv=mv, e=e, d=d, L3=L3, CE=ce.sum(), z=z)


# ------------------------------------------------------------------ backward stages
def gp_posterior_bwd_weights(y, s2, eps, ps, zbar, gT, c):
    """First half of svgp_gp_posterior_hensman_bwd (element-wise): upstream gradients of
    p_m, p_v and the weights of the backward statistics.  gT = d loss / d KL_term."""
    p = ps['p']
    g3 = gT
    g_pv = 0.5 * gT * p + zbar * eps / (2.0 * torch.sqrt(ps['p_v']))
    g_pm = gT * p * (ps['p_m'] - y) + zbar
    mvbar = g3 * p * ps['e']
    return g_pv, g_pm, mvbar


def gp_factor_bwd(K, S, v, f, A2, ud, td, gT, c, N_train, b_global, kl_form=0):
    """svgp_gp_factor_bwd: all m x m reverse algebra (identical on every rank once S, v, A2, ud, td
    are the global sums).  Returns K_bar (m,m), and per channel P = 2 Si, Q = Ssym - g3 Ki A Ki,
    v_bar, Ssym (for the per-sample stage)."""
    L, m = v.shape
    g3 = gT
    gK = -gT * (b_global / N_train)
    Ki, Si, A, G, Aji, mu, u, t = (f[k] for k in ('Ki', 'Si', 'A', 'G', 'Aji', 'mu', 'u', 't'))
    KiSKi = Ki[None] @ S @ Ki[None]
    Abar = -0.5 * g3 * KiSKi + 0.5 * gK * (Ki[None] - Aji)
    if kl_form == 0:
        ubar = ud + 0.5 * gK * mu
        mubar = 0.5 * gK * u + ubar @ Ki.T
        Kibar = 0.5 * gK * A.sum(0) + torch.einsum('li,lj->ij', ubar, mu)
    else:   # mu_hat enters only through mean_vector; the KL summand L tr(Ki A A) feeds Abar and Kibar
        KiA = Ki[None] @ A
        Abar = Abar + 0.5 * gK * L * (KiA + KiA.transpose(1, 2))
        ubar = ud
        mubar = ubar @ Ki.T
        Kibar = 0.5 * gK * A.sum(0) + torch.einsum('li,lj->ij', ubar, mu) + 0.5 * gK * L * (A @ A).sum(0)
    Gbar = K[None] @ Abar
    Kbar = (Abar @ G.transpose(1, 2)).sum(0) + (Si @ Gbar).sum(0)
    Sibar = Gbar @ K[None] + A2
    Kbar = Kbar + c * torch.einsum('li,lj->ij', mubar, t)
    tbar = td + c * (mubar @ K.T)
    Sibar = Sibar + torch.einsum('li,lj->lij', tbar, v)
    vbar = torch.einsum('lij,lj->li', Si, tbar)
    Sgbar = -(Si @ Sibar @ Si)
    Kbar = Kbar + Sgbar.sum(0)
    Sbar = c * Sgbar
    Ssym = Sbar + Sbar.transpose(1, 2)
    # Kn^T Wbar written through the all-reduced statistics
    KnTWbar = (-g3) * (S @ Ki[None] @ A).sum(0) + (0.5 * g3 * S - A2).sum(0)
    Kibar = Kibar + KnTWbar
    Kbar = Kbar - Ki @ Kibar @ Ki + (0.5 * gK * L) * Ki
    P = 2.0 * Si
    Q = Ssym - g3 * (Ki[None] @ A @ Ki[None])
    return dict(Kbar=Kbar, P=P, Q=Q, vbar=vbar, Ssym=Ssym)


def gp_posterior_bwd_rows(Kn, knn, y, s2, ps, f, fb, g_pv, g_pm, mvbar, gT, c):
    """Second half of svgp_gp_posterior_hensman_bwd: row-local gradients
    Kn_bar (b,m), knn_bar (b), y_bar (b,L), s2_bar (b,L)."""
    p = ps['p']
    g3 = gT
    qbar = (0.5 * g3 * p - g_pv).sum(1)
    knnbar = -qbar
    Mk = torch.einsum('nl,lij,nj->ni', g_pv, fb['P'], Kn) + torch.einsum('nl,lij,nj->ni', p, fb['Q'], Kn)
    Knbar = (Mk + mvbar @ f['u'] + c * (g_pm @ f['t']) + (p * y) @ fb['vbar']
             + 2.0 * qbar[:, None] * ps['W'])
    kv = Kn @ fb['vbar'].T                                     # (b,L)
    kSk = 0.5 * torch.einsum('ni,lij,nj->nl', Kn, fb['Ssym'], Kn)
    pbar = -0.5 * g3 * ps['d'] + kSk + y * kv
    ce_y = -gT * p * (ps['p_m'] - y)
    ce_s2 = 0.5 * gT * (p - (ps['p_v'] + (ps['p_m'] - y) ** 2) * p * p)
    ybar = ce_y - g3 * p * ps['e'] + p * kv
    s2bar = ce_s2 - 0.5 * g3 * p - pbar * p * p
    return Knbar, knnbar, ybar, s2bar


def gp_block_manual(K, Kn, knn, y, s2, eps, zbar, gT, jitter, N_train, b_global=None, kl_form=0):
    """Whole block forward + hand-derived backward (single rank).  Returns forward dicts and
    (K_bar, Kn_bar, knn_bar, y_bar, s2_bar)."""
    b = Kn.shape[0]
    bg = float(b if b_global is None else b_global)
    c = N_train / bg
    p = reciprocal_no_nan(s2)
    S, v, _ = gp_stats(Kn, p, p * y)
    f = gp_factor_fwd(K, S, v, jitter, c, kl_form)
    ps = gp_posterior_fwd(Kn, knn, y, s2, eps, f, c)
    g_pv, g_pm, mvbar = gp_posterior_bwd_weights(y, s2, eps, ps, zbar, gT, c)
    A2, ud, td = gp_stats(Kn, g_pv, mvbar, c * g_pm)
    fb = gp_factor_bwd(K, S, v, f, A2, ud, td, gT, c, N_train, bg, kl_form)
    Knbar, knnbar, ybar, s2bar = gp_posterior_bwd_rows(Kn, knn, y, s2, ps, f, fb, g_pv, g_pm,
                                                       mvbar, gT, c)
    return f, ps, fb, (fb['Kbar'], Knbar, knnbar, ybar, s2bar)


# ======================================================================================================================
# Large-m ("W form") staging of the same block: svgp-vae_amd/csrc/gp_large.hip from round 4 on.
#
# The L3 integrand's term k_n^T Ki A_l Ki k_n (SVGPVAE_model.py:281-284: K_mm_inv A_hat K_mm_inv inside the (b,m,m) lambda
# tensor) is evaluated as w_n^T Si_l w_n with the channel-INDEPENDENT rows w_n = K Ki k_n (W = Kn Ki K; Ki K is not
# simplified to I): identical algebra (A_l = K Si_l K), but the per-channel m^3 products Ki A_l Ki, and in the
# reverse pass S_l Ki, Ki S_l Ki, S_l Ki A_l, disappear -- their role is taken by one more (b, m, m) row product per channel
# (W Si_l beside Kn Si_l), by the weighted statistic W^T diag(p_l) W that joins the reverse statistic A2_l, and by a few
# channel-independent m x m products.  Row sums that enter the gradient of K_mm LINEARLY (Pbar = Kn^T Wbar, Qs = Kn^T diag(
# qbar) Kn) stay rank-local under data parallelism: the ranks' shares add up in the gradient all-reduce.
# ======================================================================================================================
def gp_posterior_fwd_w(Kn, knn, y, s2, eps, f, c, K):
    """svgp_big_posterior_fwd: as gp_posterior_fwd with s = rowdot(W Si_l, W), W = (Kn Ki) K."""
    p = reciprocal_no_nan(s2)
    KnKi = Kn @ f['Ki']
    W = KnKi @ K
    q = (KnKi * Kn).sum(1)
    p_m = c * (Kn @ f['t'].T)
    KnSi = torch.einsum('ni,lij->lnj', Kn, f['Si'])
    WSi = torch.einsum('ni,lij->lnj', W, f['Si'])
    r = torch.einsum('lnj,nj->nl', KnSi, Kn)
    s = torch.einsum('lnj,nj->nl', WSi, W)
    p_v = (knn - q)[:, None] + r
    mv = Kn @ f['u'].T
    e = y - mv
    d = (knn - q)[:, None] + s + e * e
    L3 = -0.5 * ((p * d).sum(0) + torch.log(s2).sum(0) + Kn.shape[0] * LOG_2PI)
    ce = -0.5 * (LOG_2PI + torch.log(s2) + (p_v + (p_m - y) ** 2) * p)
    z = p_m + eps * torch.sqrt(p_v)
    return dict(p=p, W=KnKi, Ww=W, KnSi=KnSi, WSi=WSi, q=q, p_m=p_m, p_v=p_v, mv=mv, e=e, d=d, L3=L3, CE=ce.sum(), z=z)


def gp_sw_rows(ps):
    """svgp_big_factor_bwd, early part, row form (batch local to the rank): SW_l = W^T diag(p_l) W."""
    return torch.einsum('nl,ni,nj->lij', ps['p'], ps['Ww'], ps['Ww'])


def gp_sw_mspace(S, K, Ki):
    """The same statistic from the (all-reduced) S_l: SW_l = P^T S_l P, P = Ki K -- the form under data parallelism (no
    exchange of its own) and whenever b >= 3 m."""
    P = Ki @ K
    return P.T[None] @ S @ P[None]


def gp_rows_local_w(Kn, ps, g_pv, gT, K, Ki):
    """The rank-local row sums of the reverse pass (forward quantities and loss seeds only, so they run early):
    Wbar = -g3 sum_l p_l * (W Si_l) (b, m); Pbar = Kn^T Wbar; Qs = Kn^T diag(qbar) Kn, qbar = sum_l (g3/2 p - g_pv)."""
    g3 = gT
    Wbar = -g3 * torch.einsum('nl,lnj->nj', ps['p'], ps['WSi'])
    Pbar = Kn.T @ Wbar
    qbar = (0.5 * g3 * ps['p'] - g_pv).sum(1)
    Qs = Kn.T @ (qbar[:, None] * Kn)
    return dict(Wbar=Wbar, Pbar=Pbar, Qs=Qs, qbar=qbar)


def gp_factor_bwd_w(K, v, f, A2, SW, ud, td, loc, gT, c, N_train, b_global, rep_weight=1.0):
    """svgp_big_factor_bwd: the m x m reverse algebra on a channel window (f, A2, SW, ud, td hold the window's channels).
    Returns Kbar = rep_weight x (window part) + (rank-local part from `loc`), and per channel Ssym, vbar.
    No per-channel Kibar / Abar arrays: Abar_l = gK/2 (Ki - Aji_l) enters through Gbar' = K Ki - K Aji_l (the scalar is applied
    where the products are consumed), the Ki-gradient is formed for the channel SUM only."""
    L, m = v.shape
    g3 = gT
    gK = -gT * (b_global / N_train)
    Ki, Si, A, Aji, mu, u, t = (f[k] for k in ('Ki', 'Si', 'A', 'Aji', 'mu', 'u', 't'))
    G = f['G']                                                    # Si K (forward product)
    D = Ki[None] - Aji                                            # Abar_l = gK/2 D_l (KL term only: the d-term went to W)
    H = G @ D                                                     # = Z' = Si K D
    HG = H @ G.transpose(1, 2)                                    # = Si K D K Si (symmetric)
    ubar = ud + 0.5 * gK * mu
    mubar = 0.5 * gK * u + ubar @ Ki.T
    tbar = td + c * (mubar @ K.T)
    # (the gradient of t = Si v is the rank-one tbar v^T; only the symmetric part of Sg is ever used -- Ssym below, Kbar + Kbar^T in
    # the kernel-matrix reverse pass -- so it is symmetrised where it is formed: gp_large.hip k_big_fb_sibar)
    tv = torch.einsum('li,lj->lij', tbar, v)
    X = A2 - 0.5 * g3 * SW + 0.5 * (tv + tv.transpose(1, 2))      # gradient of Si without the share through A_hat = K Si K
    vbar = torch.einsum('lij,lj->li', Si, tbar)
    Sg0 = -(Si @ X @ Si)
    Sg = Sg0 - 0.5 * gK * HG                                      # -Si (X + gK/2 K D K) Si
    Ssym = c * (Sg0 + Sg0.transpose(1, 2)) - c * gK * HG
    Zs = H.sum(0)
    Kb_win = 0.5 * gK * (Zs + Zs.T) + c * torch.einsum('li,lj->ij', mubar, t) + Sg.sum(0)
    Kib_win = 0.5 * gK * A.sum(0) + torch.einsum('li,lj->ij', ubar, mu)
    Kib_loc = loc['Qs'] + loc['Pbar'] @ K                         # q_n = k^T Ki k in d and p_v;  P = Ki K
    Kbar = rep_weight * (Kb_win - Ki @ Kib_win @ Ki + (0.5 * gK * L) * Ki) + (Ki @ loc['Pbar'] - Ki @ Kib_loc @ Ki)
    return dict(Kbar=Kbar, vbar=vbar, Ssym=Ssym)


def gp_posterior_bwd_rows_w(Kn, knn, y, s2, ps, f, fb, loc, g_pv, g_pm, mvbar, gT, c, K):
    """svgp_big_posterior_bwd: row-local gradients; the d-term reaches Kn through Wbar P^T = Wbar (Ki K)^T (one (b, m, m)
    product for all channels) instead of through Kn M2_l."""
    p = ps['p']
    g3 = gT
    qbar = loc['qbar']
    knnbar = -qbar
    R = torch.einsum('ni,lij->lnj', Kn, fb['Ssym'])
    part = (2.0 * g_pv.T[:, :, None] * ps['KnSi'] + p.T[:, :, None] * R + mvbar.T[:, :, None] * f['u'][:, None, :]
            + c * g_pm.T[:, :, None] * f['t'][:, None, :] + (p * y).T[:, :, None] * fb['vbar'][:, None, :])
    Knbar = part.sum(0) + 2.0 * qbar[:, None] * ps['W'] + loc['Wbar'] @ (K @ f['Ki'])
    kv = Kn @ fb['vbar'].T
    kSk = 0.5 * torch.einsum('lnj,nj->nl', R, Kn)
    pbar = -0.5 * g3 * ps['d'] + kSk + y * kv
    ce_y = -gT * p * (ps['p_m'] - y)
    ce_s2 = 0.5 * gT * (p - (ps['p_v'] + (ps['p_m'] - y) ** 2) * p * p)
    ybar = ce_y - g3 * p * ps['e'] + p * kv
    s2bar = ce_s2 - 0.5 * g3 * p - pbar * p * p
    return Knbar, knnbar, ybar, s2bar


def gp_block_manual_w(K, Kn, knn, y, s2, eps, zbar, gT, jitter, N_train, b_global=None):
    """gp_block_manual in the W form (single rank)."""
    b = Kn.shape[0]
    bg = float(b if b_global is None else b_global)
    c = N_train / bg
    p = reciprocal_no_nan(s2)
    S, v, _ = gp_stats(Kn, p, p * y)
    f = gp_factor_fwd(K, S, v, jitter, c)
    ps = gp_posterior_fwd_w(Kn, knn, y, s2, eps, f, c, K)
    g_pv, g_pm, mvbar = gp_posterior_bwd_weights(y, s2, eps, ps, zbar, gT, c)
    A2, ud, td = gp_stats(Kn, g_pv, mvbar, c * g_pm)
    loc = gp_rows_local_w(Kn, ps, g_pv, gT, K, f['Ki'])
    fb = gp_factor_bwd_w(K, v, f, A2, gp_sw_rows(ps), ud, td, loc, gT, c, N_train, bg)
    Knbar, knnbar, ybar, s2bar = gp_posterior_bwd_rows_w(Kn, knn, y, s2, ps, f, fb, loc, g_pv, g_pm, mvbar, gT, c, K)
    return f, ps, fb, (fb['Kbar'], Knbar, knnbar, ybar, s2bar)

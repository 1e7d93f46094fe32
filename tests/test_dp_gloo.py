"""Data-parallel schedule on CPU: world_size-2 (and 3, uneven shards) `gloo` processes run
svgp_vae_amd.engine.DataParallelStep over an ORACLE-backed stand-in for the HIP phases (test
infrastructure: same phase cuts, same exchange blocks statA / statB / gradC, same rank-0-only
rule for replicated gradient terms) and must reproduce the single-process gradients and scalars.

This pins the host logic of SURVEY 8e: row sharding, what is all-reduced, where, and the
`rep_weight` rule - the HIP engine plugs into the very same DataParallelStep on the GPU box.
"""
import math
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import staged_gp as SG
from oracle import svgpvae_oracle as O
from tests import helpers as H

DT = torch.float64
VAE_KEYS = [k for k, _ in O.mnist_vae_param_shapes(3)]
ORDER = VAE_KEYS + ["inducing_index_points", "l_GP", "amplitude", "object_vectors"]


class OracleBackend:
    """Same phase contract as MnistStepEngine, arithmetic from oracle/ (CPU, float64)."""

    def __init__(self, params, images, aux, eps, *, b_global, rank, N_train, jitter, geco, beta, lagrange):
        self.p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
        self.images, self.aux, self.eps = images, aux, eps
        self.bg, self.rank, self.N, self.j = float(b_global), rank, N_train, jitter
        self.geco, self.beta, self.lam = geco, beta, lagrange
        self.c = N_train / self.bg
        self.L = eps.shape[1]
        self.blocks = {}

    def block(self, name):
        return self.blocks[name]

    def phase(self, k):
        getattr(self, f"_phase{k}")()

    def _phase0(self):
        p = self.p
        self.vae = O.MnistVAE(p, self.L)
        self.mu, var_raw = self.vae.encode(self.images)
        self.var = O.clip_by_value(var_raw, 1e-3, 10.0)
        self.K, self.Kn, self.knn = SG.kernel_matrix_fwd(self.aux, p["inducing_index_points"].detach(),
                                                         p["object_vectors"].detach(), p["l_GP"].detach(),
                                                         p["amplitude"].detach())
        self.y, self.s2 = self.mu.detach(), self.var.detach()
        pr = O.reciprocal_no_nan(self.s2)
        S, v, _ = SG.gp_stats(self.Kn, pr, pr * self.y)
        self.shapeS, self.shapev = S.shape, v.shape
        self.blocks["statA"] = torch.cat([S.reshape(-1), v.reshape(-1)])

    def _phase1(self):
        nS = math.prod(self.shapeS)
        blk = self.blocks["statA"]
        self.S, self.v = blk[:nS].view(self.shapeS), blk[nS:].view(self.shapev)
        self.f = SG.gp_factor_fwd(self.K, self.S, self.v, self.j, self.c)
        self.ps = SG.gp_posterior_fwd(self.Kn, self.knn, self.y, self.s2, self.eps, self.f, self.c)
        z = self.ps["z"].clone().requires_grad_(True)
        recon = self.vae.decode(z)
        self.sq = torch.sum((self.images - recon) ** 2)
        gscale = (self.lam / self.bg if self.geco else 1.0) / 784.0
        dec_keys = [k for k in VAE_KEYS if k.startswith("dec_")]
        gs = torch.autograd.grad(gscale * self.sq, [z] + [self.p[k] for k in dec_keys])
        self.zbar = gs[0]
        self.g = {k: g for k, g in zip(dec_keys, gs[1:])}
        self.gT = -1.0 if self.geco else -self.beta / self.L
        self.gw = SG.gp_posterior_bwd_weights(self.y, self.s2, self.eps, self.ps, self.zbar, self.gT, self.c)
        A2, ud, td = SG.gp_stats(self.Kn, self.gw[0], self.gw[2], self.c * self.gw[1])
        self.blocks["statB"] = torch.cat([A2.reshape(-1), ud.reshape(-1), td.reshape(-1)])

    def _phase2(self):
        nS, nv = math.prod(self.shapeS), math.prod(self.shapev)
        blk = self.blocks["statB"]
        A2, ud, td = blk[:nS].view(self.shapeS), blk[nS:nS + nv].view(self.shapev), blk[nS + nv:].view(self.shapev)
        fb = SG.gp_factor_bwd(self.K, self.S, self.v, self.f, A2, ud, td, self.gT, self.c, self.N, self.bg)
        Knbar, knnbar, ybar, s2bar = SG.gp_posterior_bwd_rows(self.Kn, self.knn, self.y, self.s2, self.ps, self.f, fb,
                                                              self.gw[0], self.gw[1], self.gw[2], self.gT, self.c)
        rep = 1.0 if self.rank == 0 else 0.0          # replicated K_bar counted once across ranks
        p = self.p
        d_ip, d_ls, d_amp, d_ov = SG.kernel_matrix_bwd(self.aux, p["inducing_index_points"].detach(),
                                                       p["object_vectors"].detach(), p["l_GP"].detach(),
                                                       p["amplitude"].detach(), rep * fb["Kbar"], Knbar, knnbar)
        enc_keys = [k for k in VAE_KEYS if k.startswith("enc_")]
        gs = torch.autograd.grad((ybar * self.mu).sum() + (s2bar * self.var).sum(), [p[k] for k in enc_keys])
        self.g.update({k: g for k, g in zip(enc_keys, gs)})
        self.g.update(inducing_index_points=d_ip, l_GP=d_ls, amplitude=d_amp, object_vectors=d_ov)
        pr = O.reciprocal_no_nan(self.s2)
        l3data = -0.5 * ((pr * self.ps["d"]).sum() + torch.log(self.s2).sum())
        sums = torch.stack([l3data, self.ps["CE"], self.sq.detach(), torch.tensor(float(self.y.shape[0]), dtype=DT)])
        self.blocks["gradC"] = torch.cat([self.g[k].reshape(-1) for k in ORDER] + [sums])

    def _phase3(self):
        pass


class ShardedOracleBackend(OracleBackend):
    """Stage contract of engine.ChannelShardedStep over the oracle's "W form" staging (oracle/staged_gp.py, gp_*_w = the
    arithmetic of gp_large.hip): the m x m factor stages run on the rank's channel window only; Sigma^-1, t, u are
    all-gathered (no M2 / A: the row stage evaluates k^T Ki A Ki k as w^T Si w); the statistic SW_l = P^T S_l P comes from
    the reduce-scattered S_l, the row sums Pbar, Qs stay rank-local and reach the gradient through every rank's Kbar share."""

    def __init__(self, *a, world, packed=False, **kw):
        super().__init__(*a, **kw)
        self.world, self.packed = world, packed

    def stage(self, k):
        getattr(self, f"_stage{k}")()

    def ops(self, k):
        """Entries (kind, tensor) or (kind, tensor, avg): the three-element form marks an (L,m,m) SYMMETRIC block, which the
        packed schedule moves as its tile-packed lower triangle (engine.SymBlock; avg: symmetrised by averaging)."""
        from svgp_vae_amd.engine import ExchangeOp, SymBlock
        m = self.K.shape[0]
        out = []
        for e in self._ops[k]:
            sym = SymBlock(m, self.L, avg=e[2]) if (self.packed and len(e) == 3) else None
            out.append(ExchangeOp(e[0], e[1], sym))
        return out

    def _win(self):
        nl = self.L // self.world
        return self.rank * nl, (self.rank + 1) * nl

    def _stage0(self):
        self._phase0()
        nS = math.prod(self.shapeS)
        blk = self.blocks["statA"]
        self.S, self.v = blk[:nS].view(self.shapeS), blk[nS:].view(self.shapev)
        self._ops = {0: [("reduce_scatter", blk[:nS], False), ("reduce_scatter", blk[nS:])]}

    def _stage1(self):
        l0, l1 = self._win()
        fw = SG.gp_factor_fwd(self.K, self.S[l0:l1], self.v[l0:l1], self.j, self.c)
        self.fw = fw
        L = self.L
        self.full = {k: torch.full((L,) + tuple(fw[k].shape[1:]), float("nan"), dtype=DT) for k in ("Si", "t", "u", "KL")}
        for k in self.full:
            self.full[k][l0:l1] = fw[k]
        self._ops[1] = [("allgather", self.full["Si"].view(-1), False)] + [("allgather", self.full[k].view(-1)) for k in ("t", "u")]

    def _stage2(self):
        f = dict(self.fw)
        f.update({k: self.full[k] for k in ("Si", "t", "u")})
        self.f = f
        self.ps = SG.gp_posterior_fwd_w(self.Kn, self.knn, self.y, self.s2, self.eps, f, self.c, self.K)
        z = self.ps["z"].clone().requires_grad_(True)
        recon = self.vae.decode(z)
        self.sq = torch.sum((self.images - recon) ** 2)
        gscale = (self.lam / self.bg if self.geco else 1.0) / 784.0
        dec_keys = [k for k in VAE_KEYS if k.startswith("dec_")]
        gs = torch.autograd.grad(gscale * self.sq, [z] + [self.p[k] for k in dec_keys])
        self.zbar = gs[0]
        self.g = {k: g for k, g in zip(dec_keys, gs[1:])}
        self.gT = -1.0 if self.geco else -self.beta / self.L
        self.gw = SG.gp_posterior_bwd_weights(self.y, self.s2, self.eps, self.ps, self.zbar, self.gT, self.c)
        self.A2, self.ud, self.td = (t.contiguous() for t in SG.gp_stats(self.Kn, self.gw[0], self.gw[2], self.c * self.gw[1]))
        self.loc = SG.gp_rows_local_w(self.Kn, self.ps, self.gw[0], self.gT, self.K, self.fw["Ki"])      # rank-local row sums
        self._ops[2] = [("reduce_scatter", self.A2.view(-1), False)] + [("reduce_scatter", t.view(-1)) for t in (self.ud, self.td)]

    def _stage3(self):
        l0, l1 = self._win()
        fwin = dict(self.fw)                       # the window's own factors (G, A, Aji, mu are never exchanged)
        SW = SG.gp_sw_mspace(self.S[l0:l1], self.K, self.fw["Ki"])          # from the reduce-scattered S: no exchange of its own
        fb = SG.gp_factor_bwd_w(self.K, self.v[l0:l1], fwin, self.A2[l0:l1], SW, self.ud[l0:l1], self.td[l0:l1], self.loc,
                                self.gT, self.c, self.N, self.bg)
        self.Kbar_share = fb["Kbar"]               # window share + this rank's row-local share: every share counts
        L = self.L
        self.fbfull = {k: torch.full((L,) + tuple(fb[k].shape[1:]), float("nan"), dtype=DT) for k in ("Ssym", "vbar")}
        for k in self.fbfull:
            self.fbfull[k][l0:l1] = fb[k]
        self._ops[3] = [("allgather", self.fbfull["Ssym"].view(-1), False)] + \
            [("allgather", self.fbfull["vbar"].view(-1)), ("allgather", self.full["KL"].view(-1))]

    def _stage4(self):
        fb = dict(Ssym=self.fbfull["Ssym"], vbar=self.fbfull["vbar"])
        Knbar, knnbar, ybar, s2bar = SG.gp_posterior_bwd_rows_w(self.Kn, self.knn, self.y, self.s2, self.ps, self.f, fb, self.loc,
                                                                self.gw[0], self.gw[1], self.gw[2], self.gT, self.c, self.K)
        p = self.p
        # every rank's Kbar share counts (the kernel-matrix VJP is linear in Kbar and takes it unweighted)
        d_ip, d_ls, d_amp, d_ov = SG.kernel_matrix_bwd(self.aux, p["inducing_index_points"].detach(),
                                                       p["object_vectors"].detach(), p["l_GP"].detach(),
                                                       p["amplitude"].detach(), self.Kbar_share, Knbar, knnbar)
        enc_keys = [k for k in VAE_KEYS if k.startswith("enc_")]
        gs = torch.autograd.grad((ybar * self.mu).sum() + (s2bar * self.var).sum(), [p[k] for k in enc_keys])
        self.g.update({k: g for k, g in zip(enc_keys, gs)})
        self.g.update(inducing_index_points=d_ip, l_GP=d_ls, amplitude=d_amp, object_vectors=d_ov)
        pr = O.reciprocal_no_nan(self.s2)
        l3data = -0.5 * ((pr * self.ps["d"]).sum() + torch.log(self.s2).sum())
        sums = torch.stack([l3data, self.ps["CE"], self.sq.detach(), torch.tensor(float(self.y.shape[0]), dtype=DT)])
        self.blocks["gradC"] = torch.cat([self.g[k].reshape(-1) for k in ORDER] + [sums])
        self._ops[4] = [("allreduce", self.blocks["gradC"])]

    def _stage5(self):
        pass


def _worker_sharded(rank, world, port, b_global, geco, L, ret, packed=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from svgp_vae_amd.engine import ChannelShardedStep, shard_rows
        params, images, aux, eps = H.toy_problem(b=b_global, m=12, L=L, M=4, n_obj=20, seed=0)
        lo, hi = shard_rows(b_global, world, rank)
        be = ShardedOracleBackend(params, images[lo:hi], aux[lo:hi], eps[lo:hi], b_global=b_global, rank=rank, world=world,
                                  N_train=300.0, jitter=1e-6, geco=geco, beta=0.001, lagrange=1.7, packed=packed)
        ChannelShardedStep(be).step()
        if rank == 0:
            ret.put((be.block("gradC").clone().numpy(), be.full["KL"].clone().numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,b_global,geco,L,packed", [(2, 40, True, 4, False), (3, 41, False, 3, False), (2, 40, False, 6, False),
                                                          (2, 40, True, 4, True), (3, 41, False, 3, True)])
def test_channel_sharded_schedule_reproduces_single_process(world, b_global, geco, L, packed):
    """Reduce-scatter over the channels -> factor L / G channels per rank -> all-gather:
    gradients, scalar sums and the per-channel KL terms equal the single-process oracle.  packed: the symmetric (L,m,m)
    members of every exchange point travel as their symmetrised lower triangle (engine.SymBlock)."""
    ctx = mp.get_context("spawn")
    ret = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_sharded, args=(r, world, port, b_global, geco, L, ret, packed)) for r in range(world)]
    for p in procs:
        p.start()
    import time
    t0 = time.time()
    while ret.empty():                           # a failed worker must fail the test, not hang it on the queue
        if any(p.exitcode not in (None, 0) for p in procs) or time.time() - t0 > 300:
            for q in procs:
                if q.is_alive():
                    q.terminate()
            pytest.fail(f"worker exit codes {[p.exitcode for p in procs]}")
        time.sleep(0.05)
    got, KL = (torch.from_numpy(a) for a in ret.get())
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    params, images, aux, eps = H.toy_problem(b=b_global, m=12, L=L, M=4, n_obj=20, seed=0)
    out, grads = O.loss_and_grads(params, images, aux, eps, beta=0.001, C_ma=torch.zeros((), dtype=DT),
                                  lagrange_mult=torch.tensor(1.7, dtype=DT), alpha=0.9, kappa=math.sqrt(0.02),
                                  clipping_qs=True, GECO=geco, jitter=1e-6, N_train=300.0, L=L, formulation="efficient")
    keys = [k for k, _ in O.mnist_vae_param_shapes(L)] + ["inducing_index_points", "l_GP", "amplitude", "object_vectors"]
    want = torch.cat([grads[k].reshape(-1) for k in keys])
    n = want.numel()
    assert float((got[:n] - want).abs().max() / want.abs().max()) < 1e-9
    assert abs(float(KL.sum()) - float(out[11])) < 1e-9 * abs(float(out[11]))       # inside_elbo_kl = sum_l KL_l
    sums = got[n:]
    assert abs(float(sums[1]) - float(out[4])) < 1e-9 * abs(float(out[4]))          # ce_term
    assert float(sums[3]) == b_global


def _worker(rank, world, port, b_global, geco, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from svgp_vae_amd.engine import DataParallelStep, shard_rows
        params, images, aux, eps = H.toy_problem(b=b_global, m=12, L=3, M=4, n_obj=20, seed=0)
        lo, hi = shard_rows(b_global, world, rank)
        be = OracleBackend(params, images[lo:hi], aux[lo:hi], eps[lo:hi], b_global=b_global, rank=rank,
                           N_train=300.0, jitter=1e-6, geco=geco, beta=0.001, lagrange=1.7)
        DataParallelStep(be).step()
        if rank == 0:
            ret.put(be.block("gradC").clone())
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_shard_rows_partitions_exactly():
    from svgp_vae_amd.engine import shard_rows
    for b, w in ((256, 8), (210, 8), (5, 3), (1, 1), (7, 8)):
        spans = [shard_rows(b, w, r) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == b
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize("world,b_global,geco", [(2, 40, False), (2, 40, True), (3, 41, False)])
def test_dp_schedule_reproduces_single_process(world, b_global, geco):
    ctx = mp.get_context("spawn")
    ret = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, b_global, geco, ret)) for r in range(world)]
    for p in procs:
        p.start()
    got = ret.get()
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    params, images, aux, eps = H.toy_problem(b=b_global, m=12, L=3, M=4, n_obj=20, seed=0)
    out, grads = O.loss_and_grads(params, images, aux, eps, beta=0.001, C_ma=torch.zeros((), dtype=DT),
                                  lagrange_mult=torch.tensor(1.7, dtype=DT), alpha=0.9, kappa=math.sqrt(0.02),
                                  clipping_qs=True, GECO=geco, jitter=1e-6, N_train=300.0, L=3,
                                  formulation="efficient")
    want = torch.cat([grads[k].reshape(-1) for k in ORDER])
    n = want.numel()
    assert float((got[:n] - want).abs().max() / want.abs().max()) < 1e-9
    sums = got[n:]
    L3 = float(sums[0]) - 0.5 * 3 * b_global * O.LOG_2PI
    assert abs(L3 - float(out[10])) < 1e-9 * abs(float(out[10]))          # inside_elbo_recon
    assert abs(float(sums[1]) - float(out[4])) < 1e-9 * abs(float(out[4]))  # ce_term
    assert float(sums[3]) == b_global


def test_exchange_length_agreement_through_a_communicator_without_process_group():
    """engine.agree_on_lengths, communicator form (with only the library communicator attached, ranks built with
    different b_max ran all-reduces of different counts): three thread-ranks with a stand-in SUM all-reduce.  Equal lengths pass
    on every rank; ONE rank with a different length makes EVERY rank raise (nobody is left inside the next collective)."""
    import threading

    from svgp_vae_amd import _lib
    from svgp_vae_amd.engine import agree_on_lengths

    class FakeComm:
        def __init__(self, rank, world, shared):
            self.rank, self.world_size, self.sh = rank, world, shared

        def all_reduce(self, t, stream):
            sh = self.sh
            sh["slots"][self.rank] = t.clone()
            sh["bar"].wait()
            tot = sum(sh["slots"])
            sh["bar"].wait()
            t.copy_(tot)

    def run(lens_of_rank):
        W = len(lens_of_rank)
        shared = {"slots": [None] * W, "bar": threading.Barrier(W)}
        out = [None] * W

        def work(r):
            try:
                out[r] = agree_on_lengths(lens_of_rank[r], FakeComm(r, W, shared), "cpu", None)
            except _lib.SvgpError as e:
                out[r] = e

        th = [threading.Thread(target=work, args=(r,)) for r in range(W)]
        [t.start() for t in th]
        [t.join(30) for t in th]
        assert not any(t.is_alive() for t in th)
        return out

    assert run([[540672, 8, 74000]] * 3) == [True, True, True]
    res = run([[540672, 8, 74000], [540672, 8, 74000], [135168, 8, 74000]])
    assert all(isinstance(x, _lib.SvgpError) and "differ between ranks" in str(x) for x in res)

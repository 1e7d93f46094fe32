"""Host side of the SVGPVAE_Hensman step: owns the flat parameter / optimiser / workspace buffers
in HBM, enqueues the HIP phases (include/svgpvae_hip.h) on one stream, replays them as a hipGraph,
and inserts the three data-parallel all-reduces between phases.

Counterpart of the reference's graph construction + `sess.run([optim_step, ...])` loop body
(MNIST_experiment.py:111-134, 197-208, 327-355).  PyTorch is used for device memory, streams and
torch.distributed only; every kernel is in libsvgpvae_hip.so.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from ._lib import STATE, STATE_LEN, MnistCfg, ParamLayout, WsLayout, call

VAE_PARAM_NAMES = ("enc_c1_w", "enc_c1_b", "enc_c2_w", "enc_c2_b", "enc_c3_w", "enc_c3_b", "enc_d_w", "enc_d_b",
                   "dec_d_w", "dec_d_b", "dec_c1_w", "dec_c1_b", "dec_c2_w", "dec_c2_b", "dec_c3_w", "dec_c3_b")


def param_shapes(m, L, M, n_obj):
    """name -> shape, in flat-vector order (TF layouts; reference variable creation order)."""
    shp = {
        "enc_c1_w": (3, 3, 1, 8), "enc_c1_b": (8,), "enc_c2_w": (3, 3, 8, 8), "enc_c2_b": (8,),
        "enc_c3_w": (3, 3, 8, 8), "enc_c3_b": (8,), "enc_d_w": (32, 2 * L), "enc_d_b": (2 * L,),
        "dec_d_w": (L, 128), "dec_d_b": (128,), "dec_c1_w": (3, 3, 8, 8), "dec_c1_b": (8,),
        "dec_c2_w": (3, 3, 8, 8), "dec_c2_b": (8,), "dec_c3_w": (3, 3, 8, 1), "dec_c3_b": (1,),
        "inducing_index_points": (m, 2 + M), "l_GP": (), "amplitude": (),
    }
    if n_obj > 0:
        shp["object_vectors"] = (n_obj, M)
    return shp


_LAYOUT_NAME = {"inducing_index_points": "ip", "object_vectors": "ov"}


def shard_rows(b_global, world_size, rank):
    """Row partition of the global batch: rank r gets rows [lo, hi).  Remainder rows go to the
    lowest ranks (ragged last batch 210 = 27+27+26*6 on 8 ranks)."""
    base, rem = divmod(b_global, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class DataParallelStep:
    """The exchange schedule of one step, independent of what executes the phases.

    backend: object with `.phase(k)` for k in 0..3 and `.block(name)` returning the flat tensors
    'statA' (forward statistics S|v), 'statB' (backward statistics A2|ud|td), 'gradC' (gradients|sums).
    With world_size 1 no collective is issued."""

    def __init__(self, backend, group=None):
        self.backend = backend
        self.group = group
        import torch.distributed as dist
        self.dist = dist
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1

    def _allreduce(self, name):
        if self.world > 1:
            self.dist.all_reduce(self.backend.block(name), op=self.dist.ReduceOp.SUM, group=self.group)

    def step(self):
        be = self.backend
        be.phase(0)
        self._allreduce("statA")
        be.phase(1)
        self._allreduce("statB")
        if getattr(be, "split_grad_exchange", False):
            # cfg.split_grad_exchange through torch.distributed: the same two messages (no overlap with the encoder's reverse
            # pass here -- that is the in-library form's, svgp_mnist_train_step_dp)
            be.phase(4)
            self._allreduce("gradC_tail")
            be.phase(5)
            self._allreduce("gradC_head")
        else:
            be.phase(2)
            self._allreduce("gradC")
        be.phase(3)


class ChannelShardedStep:
    """Exchange schedule of one step with the factor stage sharded over the latent channels ("reduce-scatter
    over L -> factorize L/G channels per rank -> all-gather"), independent of what executes the stages.

    backend: `.stage(k)` for k in 0..5 and `.ops(k)` -> list of ExchangeOp to run after stage k (k in 0..4):
      stage 0  encoder, kernel matrices, forward statistics          ops: reduce_scatter S, v over channels
      stage 1  factor stage of the rank's channel window             ops: allgather Sigma^-1, t, u (KL at stage 3)
      stage 2  row stage, decoder fwd + bwd, backward statistics     ops: reduce_scatter A2, ud, td
      stage 3  reverse factor stage of the window                    ops: allgather Ssym, vbar
      stage 4  row gradients, kernel-matrix VJP (the rank's Kbar share counts on EVERY rank), encoder reverse pass
                                                                     ops: allreduce gradients + scalar sums
      stage 5  optimiser + epilogue
    SpritesStepEngine.phases yields exactly these op lists on the GPU; tests/test_dp_gloo.py runs the schedule with
    real processes over an oracle-backed backend."""

    def __init__(self, backend, group=None):
        self.backend, self.group = backend, group

    def step(self):
        for k in range(6):
            self.backend.stage(k)
            if k < 5:
                dist_exchange(self.backend.ops(k), self.group)


class ExchangeOp:
    """One collective of a data-parallel step on a flat, contiguous tensor.
      allreduce      : every rank ends with the sum;
      reduce_scatter : the tensor is `world` equal chunks, rank r ends with the SUM of chunk r in its chunk r (the other
                       chunks are then unspecified);
      allgather      : rank r contributes its chunk r, every rank ends with all chunks.
    In-place semantics of ncclAllReduce / ncclReduceScatter / ncclAllGather (svgp_comm_*).

    sym = SymBlock(...): the tensor is (L, m, m) SYMMETRIC matrices that travel tile-packed (svgp_sym_pack: 52 % of the
    bytes at m = 800).  What the exchange then DEFINES, whatever executes it: reduce_scatter -- chunk r = the lower triangle
    of the sum, mirrored; allgather -- every channel (the rank's own too) = the symmetrised block of its owner, lower
    triangle mirrored, or (X + X^T) / 2 with sym.avg (M2 = Ki A Ki is symmetric only up to rounding; the plain mirror would
    break the Ki E form of its rounding error)."""
    __slots__ = ("kind", "tensor", "sym")

    def __init__(self, kind, tensor, sym=None):
        assert kind in ("allreduce", "reduce_scatter", "allgather") and tensor.is_contiguous()
        assert sym is None or kind != "allreduce"
        self.kind, self.tensor, self.sym = kind, tensor, sym


class SymBlock:
    """Packing description of an (L, m, m) symmetric exchange block: xp = its tile-packed buffer (L * svgp_sym_packed_elems(m)
    doubles of the workspace's xpack region), avg = symmetrise by averaging, prepacked = the caller has already written the
    rank's window to xp (and, with avg, the symmetrised window back to the tensor) before forking work that reads it."""
    __slots__ = ("m", "L", "avg", "xp", "prepacked")

    def __init__(self, m, L, avg=False, xp=None, prepacked=False):
        self.m, self.L, self.avg, self.xp, self.prepacked = m, L, avg, xp, prepacked


def _symmetrise(x, m, avg):
    """(n, m, m) view of flat x -> the block as it comes out of svgp_sym_pack + svgp_sym_unpack."""
    X = x.view(-1, m, m)
    return (0.5 * (X + X.transpose(1, 2))) if avg else (torch.tril(X) + torch.tril(X, -1).transpose(1, 2))


def virtual_exchange(ops_per_rank):
    """Executes one exchange point for G virtual ranks that live on ONE device (the single-GPU data-parallel tests):
    ops_per_rank[r] = the list of ExchangeOp rank r reached.  Defines by construction what the collectives must do."""
    G = len(ops_per_rank)
    for ops in zip(*ops_per_rank):
        kind = ops[0].kind
        assert all(o.kind == kind and o.tensor.numel() == ops[0].tensor.numel() for o in ops)
        if kind == "allreduce":
            tot = sum(o.tensor for o in ops)
            for o in ops:
                o.tensor.copy_(tot)
            continue
        n = ops[0].tensor.numel()
        assert n % G == 0
        c = n // G
        sym = ops[0].sym
        if kind == "reduce_scatter":
            sums = [sum(o.tensor[r * c:(r + 1) * c] for o in ops) for r in range(G)]
            if sym is not None:
                sums = [_symmetrise(t, sym.m, False).reshape(-1) for t in sums]
            for r, o in enumerate(ops):
                o.tensor.fill_(float("nan"))                 # the other chunks are unspecified: poison them
                o.tensor[r * c:(r + 1) * c].copy_(sums[r])
        else:
            chunks = [ops[r].tensor[r * c:(r + 1) * c].clone() for r in range(G)]
            if sym is not None:
                chunks = [_symmetrise(t, sym.m, sym.avg).reshape(-1) for t in chunks]
            for o in ops:
                for r in range(G):
                    o.tensor[r * c:(r + 1) * c].copy_(chunks[r])


def dist_exchange(ops, group=None):
    """The same exchange point through torch.distributed (any backend; used by the gloo schedule tests)."""
    import torch.distributed as dist
    G, r = dist.get_world_size(group), dist.get_rank(group)
    for o in ops:
        if o.kind == "allreduce":
            dist.all_reduce(o.tensor, op=dist.ReduceOp.SUM, group=group)
        else:
            c = o.tensor.numel() // G
            chunks = [o.tensor[i * c:(i + 1) * c] for i in range(G)]
            if o.kind == "reduce_scatter":
                out = torch.empty_like(chunks[r])
                dist.reduce_scatter(out, [ch.clone() for ch in chunks], op=dist.ReduceOp.SUM, group=group) \
                    if dist.get_backend(group) != "gloo" else _gloo_reduce_scatter(out, chunks, r, group)
                chunks[r].copy_(_symmetrise(out, o.sym.m, False).reshape(-1) if o.sym is not None else out)
            else:
                if o.sym is not None:
                    chunks[r].copy_(_symmetrise(chunks[r].clone(), o.sym.m, o.sym.avg).reshape(-1))
                dist.all_gather([ch for ch in chunks], chunks[r].clone(), group=group)


def _gloo_reduce_scatter(out, chunks, r, group):
    """gloo has no reduce_scatter: all-reduce every chunk, keep one's own (test backend only)."""
    import torch.distributed as dist
    for i, ch in enumerate(chunks):
        t = ch.clone()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        if i == r:
            out.copy_(t)


def dp_pack_enabled(m):
    """Whether the channel-sharded exchange moves its (L,m,m) blocks tile-packed: SVGP_DP_PACK=0/1, default from m >= 512
    (the rule of svgp_mnist_train_step_dp)."""
    import os
    e = os.environ.get("SVGP_DP_PACK")
    return (e[0] != "0") if e else m >= 512


class RcclComm:
    """RCCL communicator owned by libsvgpvae_hip.so (svgp_comm_*), one per process / GPU.  Its all-reduces
    are enqueued on the engine's compute stream by svgp_mnist_train_step_dp, so a data-parallel step is one
    in-order queue with no host synchronisation.  torch.distributed is used only to hand rank 0's unique
    id to the other ranks (any backend)."""

    def __init__(self, rank, world_size, unique_id):
        self.lib = _lib.load_library()
        self.rank, self.world_size = rank, world_size
        self.handle = C.c_void_p()
        buf = C.create_string_buffer(bytes(unique_id), len(unique_id))
        call("svgp_comm_init", buf, len(unique_id), rank, world_size, C.byref(self.handle))

    @staticmethod
    def unique_id():
        lib = _lib.load_library()
        n = lib.svgp_comm_unique_id_bytes()
        buf = C.create_string_buffer(n)
        call("svgp_comm_unique_id", buf, n)
        return buf.raw

    @classmethod
    def from_process_group(cls, group=None):
        """Collective over `group`: rank 0 creates the id, broadcast_object_list distributes it."""
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [cls.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        return cls(rank, world, box[0])

    def all_reduce(self, tensor, stream):
        assert tensor.dtype in (torch.float64, torch.float32) and tensor.is_contiguous() and tensor.is_cuda
        fn = "svgp_allreduce_sum_f64" if tensor.dtype == torch.float64 else "svgp_allreduce_sum_f32"
        call(fn, self.handle, tensor.data_ptr(), tensor.numel(), stream)

    def reduce_scatter(self, tensor, stream):
        assert tensor.dtype == torch.float64 and tensor.is_contiguous() and tensor.numel() % self.world_size == 0
        call("svgp_reduce_scatter_sum_f64", self.handle, tensor.data_ptr(), tensor.numel() // self.world_size, stream)

    def all_gather(self, tensor, stream):
        assert tensor.dtype == torch.float64 and tensor.is_contiguous() and tensor.numel() % self.world_size == 0
        call("svgp_allgather_f64", self.handle, tensor.data_ptr(), tensor.numel() // self.world_size, stream)

    def run(self, ops, stream):
        """Enqueues one exchange point (list of ExchangeOp) on `stream` as ONE grouped RCCL launch (ncclGroupStart / End);
        symmetric (L,m,m) members travel tile-packed: pack -> collective on the packed buffer -> unpack (the sequence of
        svgp_mnist_train_step_dp)."""
        G, r = self.world_size, self.rank
        pe = lambda o: _lib.load_library().svgp_sym_packed_elems(o.sym.m)

        def win(o, lo, n, packed):          # data pointer of channels [lo, lo + n)
            return (o.sym.xp.data_ptr() + 8 * lo * pe(o)) if packed else (o.tensor.data_ptr() + 8 * lo * o.sym.m * o.sym.m)

        for o in ops:
            if o.sym is None or o.sym.prepacked:
                continue
            nl = o.sym.L // G
            if o.kind == "reduce_scatter":
                call("svgp_sym_pack", o.sym.m, o.sym.L, 0, o.tensor.data_ptr(), o.sym.xp.data_ptr(), stream)
            else:
                call("svgp_sym_pack", o.sym.m, nl, int(o.sym.avg), win(o, r * nl, nl, False), win(o, r * nl, nl, True), stream)
                if o.sym.avg:
                    call("svgp_sym_unpack", o.sym.m, nl, win(o, r * nl, nl, True), win(o, r * nl, nl, False), stream)
        call("svgp_comm_group_begin", self.handle)
        for o in ops:
            t = o.tensor if o.sym is None else o.sym.xp
            {"allreduce": self.all_reduce, "reduce_scatter": self.reduce_scatter, "allgather": self.all_gather}[o.kind](t, stream)
        call("svgp_comm_group_end", self.handle)
        for o in ops:
            if o.sym is None:
                continue
            nl = o.sym.L // G
            if o.kind == "reduce_scatter":
                call("svgp_sym_unpack", o.sym.m, nl, win(o, r * nl, nl, True), win(o, r * nl, nl, False), stream)
            else:
                for lo, n in ((0, r * nl), ((r + 1) * nl, o.sym.L - (r + 1) * nl)):
                    if n > 0:
                        call("svgp_sym_unpack", o.sym.m, n, win(o, lo, n, True), win(o, lo, n, False), stream)

    def timing(self, enable):
        call("svgp_comm_timing", self.handle, int(enable))

    def timing_read(self):
        """Microseconds of the exchange points of the last svgp_mnist_train_step_dp issued with timing on."""
        us, n = (C.c_float * 8)(), C.c_int(0)
        call("svgp_comm_timing_read", self.handle, us, 8, C.byref(n))
        return [float(us[k]) for k in range(n.value)]

    def close(self):
        if self.handle:
            self.lib.svgp_comm_destroy(self.handle)
            self.handle = C.c_void_p()


def agree_on_lengths(lens, comm, device, stream):
    """Collective: raises SvgpError on EVERY rank when the integer list `lens` (exchange block lengths) is not the same on all
    ranks; returns False when there is nothing to check with.  Through torch.distributed when a process group exists (MIN / MAX),
    otherwise through the library's own communicator (with only `attach_comm` ranks built with different b_max or
    SVGP_DP_STAT_PARTIALS ran all-reduces of different counts).  The communicator form uses two fixed-count SUM all-reduces:
    the lengths (every rank compares world * own with the sum -- a disagreement is seen by at least one rank), then the
    per-rank verdicts, so that all ranks raise together instead of one leaving the others in the next collective."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        t = torch.tensor(list(lens), dtype=torch.int64, device=device if dist.get_backend() == "nccl" else "cpu")
        lo, hi = t.clone(), t.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        if not torch.equal(lo, hi):
            raise _lib.SvgpError(f"exchange block lengths differ between ranks (min {lo.tolist()}, max {hi.tolist()}): "
                                 "build every rank's engine with the same b_max")
        return True
    if comm is None:
        return False
    import contextlib
    on_gpu = stream is not None            # (stream None: a host stand-in communicator, tests/test_dp_gloo.py)
    with (torch.cuda.stream(stream) if on_gpu else contextlib.nullcontext()):
        raw = stream.cuda_stream if on_gpu else None
        mine = torch.tensor([float(x) for x in lens], dtype=torch.float64, device=device)
        tot = mine.clone()
        comm.all_reduce(tot, raw)
        bad = (tot != mine * comm.world_size).any().to(torch.float64).reshape(1).contiguous()
        comm.all_reduce(bad, raw)
    if on_gpu:
        stream.synchronize()
    if float(bad.item()) > 0:
        raise _lib.SvgpError(f"exchange block lengths differ between ranks (this rank {list(lens)}, mean over ranks "
                             f"{(tot / comm.world_size).tolist()}): build every rank's engine with the same b_max")
    return True


def concurrent_streams(main, n, device):
    """n torch streams that run BESIDE `main` and beside each other.  HIP maps streams onto a small pool of hardware queues in
    creation order, and two streams on one queue execute one after the other whatever the events say (include/svgpvae_hip.h:
    svgp_streams_overlap; measured round 4: a side stream that shares the caller's queue hides nothing).  Candidates come from
    torch's stream pool and are probed with the library's spin / no-op test; if fewer than n concurrent ones exist, the first
    candidates fill up.  SVGP_STREAM_PROBE=0: the first n candidates."""
    probe = os.environ.get("SVGP_STREAM_PROBE", "1")[0] != "0"
    flag = C.c_int(0)

    def overlap(a, b):
        call("svgp_streams_overlap", a.cuda_stream, b.cuda_stream, C.byref(flag))
        return bool(flag.value)

    cands, picked = [], []
    for _ in range(12 if probe else n):
        s = torch.cuda.Stream(device=device)
        if any(s.cuda_stream == c.cuda_stream for c in cands) or s.cuda_stream == main.cuda_stream:
            continue
        cands.append(s)
        if not probe or (overlap(main, s) and all(overlap(p, s) for p in picked)):
            picked.append(s)
        if len(picked) == n:
            break
    for s in cands:
        if len(picked) < n and s not in picked:
            picked.append(s)
    return picked


class MnistStepEngine:
    """One rank's HIP execution state for the rotated-MNIST SVGPVAE_Hensman step."""

    def __init__(self, m, L=16, M=8, n_obj=400, *, N_train=4050.0, jitter=1e-6, clip_qs=True, geco=False,
                 K_obj_normalize=False, titsias=False, kappa_squared=0.020, alpha=0.99, beta=0.001, lr=1e-3,
                 train_ip=True, train_gp=True, train_ov=True, b_max=256, device="cuda:0",
                 rank=0, world_size=1, single_stat_block=None, split_grad_exchange=False):
        self.lib = _lib.load_library()          # raises if the HIP extension is missing
        if not torch.cuda.is_available():
            raise _lib.SvgpError("MnistStepEngine needs a HIP device (torch.cuda.is_available() is False); "
                                 "there is no CPU execution path")
        self.device = torch.device(device)
        self.rank, self.world_size = rank, world_size
        self.base = dict(m=m, L=L, M=M, n_obj=n_obj, normalize_obj=int(K_obj_normalize), clip_qs=int(clip_qs),
                         geco=int(geco), titsias=int(titsias), train_ip=int(train_ip), train_gp=int(train_gp), train_ov=int(train_ov),
                         N_train=float(N_train), jitter=float(jitter), kappa_squared=float(kappa_squared),
                         alpha=float(alpha), rep_weight=1.0 if rank == 0 else 0.0,
                         # sharded over ranks.  m > 64: the blocks travel (wire buffer of the packed exchange, SW from the
                         # exchanged S; gp_large.hip).  m <= 64 (round 4): the statistics launches KEEP their 4 row partials per
                         # channel and the all-reduce moves the partial blocks as one message (statA 135 -> 540 KB at config 2:
                         # still latency-bound; the consumers add the partials on load as on one GPU) -- one block per channel
                         # made the two statistics stages 12.8 + 14.2 us instead of 8.5 + 5.8.  SVGP_DP_STAT_PARTIALS=0: one block.
                         # Every rank must be built with the SAME b_max (the partial count is a function of the capacity);
                         # attach_comm / run check the block lengths across ranks.
                         # (single_stat_block on one rank: the kernel configuration of a multi-rank step, bench.py --force-comm)
                         single_stat_block=int((world_size > 1 and (m > 64 or os.environ.get("SVGP_DP_STAT_PARTIALS") == "0"))
                                               if single_stat_block is None else single_stat_block),
                         # row-sharded data parallelism: the gradient all-reduce in two parts, the first one (decoder + GP parameters
                         # + scalar sums) beside the encoder's reverse pass (include/svgpvae_hip.h: cfg.split_grad_exchange); off by
                         # default -- a fallback for slow small-message all-reduces, to be decided by the first multi-GPU run
                         split_grad_exchange=int(bool(split_grad_exchange)))
        self.b_max = b_max
        self.cfg = None
        self.pl = ParamLayout()
        call("svgp_mnist_param_layout_get", C.byref(self._make_cfg(b_max, b_max)), C.byref(self.pl))
        n = self.pl.n_total
        f64 = dict(dtype=torch.float64, device=self.device)
        self.theta = torch.zeros(n, **f64)
        self.adam_m = torch.zeros(n, **f64)
        self.adam_v = torch.zeros(n, **f64)
        self.state = torch.zeros(STATE_LEN, **f64)
        wl = WsLayout()
        call("svgp_mnist_ws_layout_get", C.byref(self._make_cfg(b_max, b_max)), C.byref(wl))
        self.ws = torch.zeros(wl.total, **f64)
        self.shapes = param_shapes(m, L, M, n_obj)
        self.params = {k: self._pview(self.theta, k) for k in self.shapes}
        self.stream = torch.cuda.Stream(device=self.device)
        # the zero-fills above were enqueued on torch's current stream; everything below writes on self.stream
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        self.reset_state(beta=beta, lr=lr)
        self._graphs = {}
        self._bound = None
        self.comm = None
        if m > 64:      # the library's side branches of this stream (forward tail / early reverse half; Cholesky look-ahead): picked
            call("svgp_side_streams_prepare", self.stream.cuda_stream)     # now, by the concurrency probe, not under a later capture
        self.set_batch_size(b_max, b_max * world_size)

    # ------------------------------------------------------------------ configuration
    def _make_cfg(self, b, b_global):
        return MnistCfg(b=b, b_global=b_global, b_cap=self.b_max, **self.base)

    def set_batch_size(self, b, b_global=None):
        """Local rows b of a global batch b_global (ragged last batch: MNIST_experiment.py:327,
        utils.py:846-848 `.batch()` without drop_remainder)."""
        if b > self.b_max:
            raise ValueError(f"b={b} exceeds b_max={self.b_max}")
        b_global = b if b_global is None else b_global
        self.cfg = self._make_cfg(b, b_global)
        self.wl = WsLayout()
        call("svgp_mnist_ws_layout_get", C.byref(self.cfg), C.byref(self.wl))
        assert self.wl.total <= self.ws.numel()

    def set_split_grad_exchange(self, flag):
        """Switches cfg.split_grad_exchange (the gradient all-reduce in two parts) on an existing engine."""
        self.base["split_grad_exchange"] = int(bool(flag))
        self.set_batch_size(self.cfg.b, self.cfg.b_global)

    def reset_state(self, beta=None, lr=None):
        """MNIST_experiment.py:313-315: first_step=True (alpha 0), C_ma_=0, lagrange_mult_=1."""
        st = torch.zeros(STATE_LEN, dtype=torch.float64)
        st[STATE["LAGRANGE"]] = 1.0
        st[STATE["ALPHA"]] = 0.0 if self.base["geco"] else self.base["alpha"]
        st[STATE["LR"]] = float(self.state[STATE["LR"]]) if lr is None else lr
        st[STATE["BETA"]] = float(self.state[STATE["BETA"]]) if beta is None else beta
        # on-device N(0,1): Philox(counter, local row * L + l).  Ranks hold different rows, so each gets its own
        # counter range (the epilogue advances it by 1 per step); 2^32 * rank is exact in float64
        st[STATE["RNG_CTR"]] = float(self.rank) * 4294967296.0
        with torch.cuda.stream(self.stream):
            self.state.copy_(st)
        self.stream.synchronize()

    def set_scalars(self, **kw):
        """Host writes into the device state vector (names of _lib.STATE, lower-case accepted)."""
        self.stream.synchronize()
        st = self.state.cpu()
        for k, v in kw.items():
            st[STATE[k.upper()]] = float(v)
        with torch.cuda.stream(self.stream):
            self.state.copy_(st)
        self.stream.synchronize()

    def scalars(self):
        self.stream.synchronize()      # the state vector is written on self.stream, .cpu() runs on torch's stream
        self.check_handoffs()
        st = self.state.cpu()
        return {k.lower(): float(st[i]) for k, i in STATE.items()}

    def check_handoffs(self):
        """The merged launches of the m <= 64 step hand data from producer to consumer workgroups of ONE launch through counters
        in ws.flags (include/svgpvae_hip.h); a consumer that gave up waiting (~1 s of polling) sets the sticky word flags[2] and
        goes on with whatever it read.  That must never pass for a result: raise, and re-arm the counters."""
        fl = self.ws_view("flags", (64,)).view(torch.int64)
        if int(fl[2].item()) != 0:
            fl.zero_()
            raise _lib.SvgpError("an intra-launch hand-off of the training step timed out (ws.flags[2] was set): "
                                 "the results of this step are invalid")

    # ------------------------------------------------------------------ views
    def _pview(self, flat, name):
        off = getattr(self.pl, _LAYOUT_NAME.get(name, name))
        shp = self.shapes[name]
        return flat[off:off + int(np.prod(shp, dtype=np.int64))].view(shp)

    def load_params(self, params):
        """params: dict name -> array-like (numpy / torch, any device)."""
        with torch.cuda.stream(self.stream):
            for k, v in params.items():
                if k not in self.params:
                    raise KeyError(k)
                self.params[k].copy_(torch.as_tensor(np.asarray(v) if not torch.is_tensor(v) else v,
                                                     dtype=torch.float64).reshape(self.shapes[k]))
        self.stream.synchronize()

    def ws_view(self, name, shape=None):
        """View of a workspace field.  The statistics fields S, v, A2, ud, td are stored as wl.stat_parts row partials;
        with a shape given, their sum (= the statistic itself) is returned."""
        off = getattr(self.wl, name)
        if shape is not None and name in ("S", "v", "A2", "ud", "td") and self.wl.stat_parts > 1:
            n, P = int(np.prod(shape, dtype=np.int64)), int(self.wl.stat_parts)
            return self.ws[off:off + P * n].view(P, *shape).sum(0)
        if shape is None:
            length = getattr(self.wl, name + "_len")
            return self.ws[off:off + length]
        return self.ws[off:off + int(np.prod(shape, dtype=np.int64))].view(shape)

    def grads(self):
        g = self.ws[self.wl.grad:self.wl.grad + self.pl.n_total]
        return {k: self._pview(g, k) for k in self.shapes}

    def block(self, name):
        if name in ("gradC_head", "gradC_tail"):      # the two messages of cfg.split_grad_exchange: encoder | everything else + sums
            g, n_enc = self.ws_view("gradC"), int(self.pl.n_enc)
            return g[:n_enc] if name == "gradC_head" else g[n_enc:]
        return self.ws_view(name)

    # ------------------------------------------------------------------ execution
    def bind(self, images, aux, eps=None):
        """Device tensors of the current batch (float64, contiguous).  eps=None -> on-device N(0,1)."""
        b = images.shape[0]
        assert images.dtype == torch.float64 and images.is_contiguous() and images.shape[1:] == (28, 28, 1)
        assert aux.dtype == torch.float64 and aux.is_contiguous() and aux.shape == (b, 2 + self.base["M"])
        assert eps is None or (eps.dtype == torch.float64 and eps.is_contiguous() and eps.shape == (b, self.base["L"]))
        if b != self.cfg.b:
            raise ValueError(f"batch rows {b} != configured {self.cfg.b}; call set_batch_size first")
        # the batch may have been produced on torch's current stream; kernels run on self.stream
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        self._bound = (images, aux, eps)

    def phase(self, k, adam=True):
        images, aux, eps = self._bound
        call("svgp_mnist_step_phase", C.byref(self.cfg), k, self.theta.data_ptr(), images.data_ptr(),
             aux.data_ptr(), eps.data_ptr() if eps is not None else None, self.ws.data_ptr(),
             self.state.data_ptr(), self.adam_m.data_ptr() if adam else None,
             self.adam_v.data_ptr() if adam else None, self.stream.cuda_stream)

    def full_step(self, adam=True):
        """svgp_mnist_train_step: the four phases issued back to back (single GPU), which lets the library
        keep its side-stream branches open across phase boundaries."""
        images, aux, eps = self._bound
        call("svgp_mnist_train_step", C.byref(self.cfg), self.theta.data_ptr(), images.data_ptr(),
             aux.data_ptr(), eps.data_ptr() if eps is not None else None, self.ws.data_ptr(),
             self.state.data_ptr(), self.adam_m.data_ptr() if adam else None,
             self.adam_v.data_ptr() if adam else None, self.stream.cuda_stream)

    def attach_comm(self, comm):
        """Use the library's RCCL communicator for the three exchanges (svgp_mnist_train_step_dp)."""
        assert comm.world_size == self.world_size and comm.rank == self.rank
        self.comm = comm
        self._check_block_lengths()

    def _check_block_lengths(self):
        """The exchange blocks must have the same length on every rank (the row-partial count of the statistics blocks is a
        function of the row capacity b_max and of SVGP_DP_STAT_PARTIALS): checked once -- see agree_on_lengths."""
        if getattr(self, "_blocks_checked", False) or self.world_size == 1:
            return
        if agree_on_lengths([self.wl.statA_len, self.wl.statB_len, self.wl.gradC_len], self.comm, self.device, self.stream):
            self._blocks_checked = True

    def full_step_dp(self, adam=True):
        images, aux, eps = self._bound
        call("svgp_mnist_train_step_dp", C.byref(self.cfg), self.comm.handle, self.theta.data_ptr(),
             images.data_ptr(), aux.data_ptr(), eps.data_ptr() if eps is not None else None, self.ws.data_ptr(),
             self.state.data_ptr(), self.adam_m.data_ptr() if adam else None,
             self.adam_v.data_ptr() if adam else None, self.stream.cuda_stream)

    def run(self, adam=True, group=None):
        """All four phases on self.stream, with all-reduces when world_size > 1 (in-stream RCCL when a
        communicator is attached, torch.distributed between the phases otherwise)."""
        with torch.cuda.stream(self.stream):
            if self.comm is not None:
                self.full_step_dp(adam)
            elif self.world_size == 1:
                self.full_step(adam)
            else:
                self._check_block_lengths()
                be = _PhaseAdapter(self, adam)
                DataParallelStep(be, group).step()

    def channel_sharded(self):
        """True when svgp_mnist_train_step_dp takes the channel-sharded schedule for this engine (large-m path, L
        divisible by the rank count, Hensman branch)."""
        return (self.world_size > 1 and self.base["m"] > 64 and self.base["L"] % self.world_size == 0
                and not self.base["titsias"] and not self.base.get("kl_form", 0))

    def sharded_stages(self, adam=True):
        """The channel-sharded data-parallel step (svgp_mnist_train_step_dp, m > 64) stage by stage: a generator that
        enqueues the stages between two exchange points and yields the list of ExchangeOp of that point.  The C entry
        point issues exactly this sequence with the RCCL collectives in between; this form exists so that the schedule
        can be executed with virtual ranks on one GPU (tests/test_gpu_dp_virtual.py)."""
        assert self.base["m"] > 64 and self.base["L"] % self.world_size == 0
        images, aux, eps = self._bound
        cfg = self._make_cfg(self.cfg.b, self.cfg.b_global)
        cfg.rep_weight = 1.0                      # every rank's Kbar share counts
        cp, th, ws, st, s = C.byref(cfg), self.theta.data_ptr(), self.ws.data_ptr(), self.state.data_ptr(), \
            self.stream.cuda_stream
        im, ax, ep = images.data_ptr(), aux.data_ptr(), (eps.data_ptr() if eps is not None else None)
        L, m, G = self.base["L"], self.base["m"], self.world_size
        nl, l0 = L // G, self.rank * (L // G)
        fld = lambda name, per: self.ws[getattr(self.wl, name):getattr(self.wl, name) + L * per]
        mm = m * m
        pack = dp_pack_enabled(m)
        pe = int(self.lib.svgp_sym_packed_elems(m))
        if pack and self.wl.xpack_len < L * pe:
            raise _lib.SvgpError("the packed exchange needs the workspace's wire buffer: build the engine with world_size > 1 "
                            "(cfg.single_stat_block) or set SVGP_DP_PACK=0")
        xp = [self.ws[self.wl.xpack:self.wl.xpack + L * pe]]            # ONE wire buffer: every point moves one symmetric block
        sym = lambda k, avg=False, pre=False: SymBlock(m, L, avg, xp[k], pre) if pack else None
        plain = lambda kind, *fields: [ExchangeOp(kind, fld(n, per)) for n, per in fields]
        fork = os.environ.get("SVGP_SIDE_STREAMS", "1")[0] != "0"
        side = self._side_stream() if fork else self.stream
        wptr = lambda name, packed_k=None: (xp[packed_k].data_ptr() + 8 * l0 * pe) if packed_k is not None else \
            (self.ws.data_ptr() + 8 * (getattr(self.wl, name) + l0 * mm))
        with torch.cuda.stream(self.stream):
            call("svgp_mnist_encoder_kernel_matrix_fwd", cp, th, im, ax, ws, s)
            call("svgp_gp_stats_fwd", cp, ws, s)
        yield [ExchangeOp("reduce_scatter", fld("S", mm), sym(0))] + plain("reduce_scatter", ("v", m))
        with torch.cuda.stream(self.stream):
            call("svgp_gp_factor_fwd_channels_part", cp, l0, nl, 1, ws, s)            # without the (A_hat + jI)^-1 tail
            if pack:      # the window in wire format BEFORE the side branch starts reading it
                call("svgp_sym_pack", m, nl, 0, wptr("Si"), wptr("Si", 0), s)
            # the tail and the early reverse half go to the side branch, beside the all-gather, the row stage, the decoder and
            # the reverse statistics: FORKED here, ISSUED behind the collective (below) -- enqueued first, the branch's GEMMs
            # fill the chip and the collective's kernel waits for a slot
            side.wait_stream(self.stream)
        # (round 4: M2 = Ki A Ki is no longer formed or exchanged -- the row stage evaluates k^T M2 k as w^T Si w, gp_large.hip)
        yield [ExchangeOp("allgather", fld("Si", mm), sym(0, pre=True))] + plain("allgather", ("t", m), ("u", m))
        with torch.cuda.stream(self.stream):
            call("svgp_gp_posterior_fwd", cp, ep, ws, st, s)              # (first: the caller's stream must not wait for the host)
            call("svgp_gp_factor_fwd_channels_part", cp, l0, nl, 2, ws, side.cuda_stream)
            if fork:
                call("svgp_gp_factor_bwd_channels_part", cp, l0, nl, 1, ws, st, side.cuda_stream)
            call("svgp_mnist_decoder_fwd", cp, th, im, ws, s)
            call("svgp_mnist_decoder_bwd", cp, th, im, ws, st, s)
            call("svgp_gp_stats_bwd", cp, ws, st, s)
        yield [ExchangeOp("reduce_scatter", fld("A2", mm), sym(0))] + plain("reduce_scatter", ("ud", m), ("td", m))
        with torch.cuda.stream(self.stream):
            self.stream.wait_stream(side)
            call("svgp_gp_factor_bwd_channels_part", cp, l0, nl, 2 if fork else 0, ws, st, s)
        # (KL_l comes out of the tail, joined above)
        yield [ExchangeOp("allgather", fld("Ssym", mm), sym(0))] + plain("allgather", ("vbar", m), ("KL", 1))
        with torch.cuda.stream(self.stream):
            call("svgp_gp_posterior_bwd", cp, ws, st, s)
            call("svgp_kernel_matrix_bwd_partials", cp, th, ax, ws, s)
            call("svgp_mnist_encoder_bwd", cp, th, im, ws, s)
            call("svgp_mnist_grad_reduce_all", cp, ax, ws, s)
        yield [ExchangeOp("allreduce", self.block("gradC"))]
        self.phase(3, adam)

    def _side_stream(self):
        if getattr(self, "_side", None) is None:
            self._side = concurrent_streams(self.stream, 1, self.device)[0]
        return self._side

    # hipGraph capture / replay through the library (not torch.cuda.graphs)
    def capture(self, key, adam=True):
        """Captures the 4 phases (single GPU) into a hipGraph stored under `key`."""
        assert self.world_size == 1, "graph capture of the full step is the single-GPU form"
        s = self.stream.cuda_stream
        self.stream.synchronize()
        call("svgp_graph_begin", s)
        try:
            self.full_step(adam)
        finally:
            exe = C.c_void_p()
            call("svgp_graph_end", s, C.byref(exe))
        old = self._graphs.get(key)
        if old is not None:
            self.lib.svgp_graph_destroy(old)
        self._graphs[key] = exe
        return exe

    def replay(self, key):
        call("svgp_graph_launch", self._graphs[key], self.stream.cuda_stream)

    def capture_phases(self, key, adam=True):
        """Four hipGraphs, one per phase; data parallelism replays them with the RCCL all-reduces in
        between (collectives themselves are not captured)."""
        s = self.stream.cuda_stream
        self.stream.synchronize()
        execs = []
        for k in range(4):
            call("svgp_graph_begin", s)
            try:
                self.phase(k, adam)
            finally:
                exe = C.c_void_p()
                call("svgp_graph_end", s, C.byref(exe))
            execs.append(exe)
            self._graphs[(key, k)] = exe
        return execs

    def run_phase_graphs(self, key, group=None):
        """One step from the per-phase graphs (+ all-reduces when world_size > 1)."""
        with torch.cuda.stream(self.stream):
            DataParallelStep(_GraphAdapter(self, key), group).step()

    def synchronize(self):
        self.stream.synchronize()

    def __del__(self):
        try:
            for exe in self._graphs.values():
                self.lib.svgp_graph_destroy(exe)
        except Exception:
            pass


class _GraphAdapter:
    def __init__(self, eng, key):
        self.eng, self.key = eng, key          # (per-phase graphs are the four phases 0..3: the unsplit exchange)

    def phase(self, k):
        self.eng.replay((self.key, k))

    def block(self, name):
        return self.eng.block(name)


class _PhaseAdapter:
    def __init__(self, eng, adam):
        self.eng, self.adam = eng, adam
        self.split_grad_exchange = bool(eng.base.get("split_grad_exchange"))

    def phase(self, k):
        self.eng.phase(k, self.adam)

    def block(self, name):
        return self.eng.block(name)

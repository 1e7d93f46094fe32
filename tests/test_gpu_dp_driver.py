"""Data-parallel training through the CLI driver (the reference is single-process, MNIST_experiment.py:299,308:
the sharded epoch loop is this build's, svgp_vae_amd/dp.py).
 (b) `python -m torch.distributed.run --nproc-per-node 1 -m svgp_vae_amd.MNIST_experiment` with SVGP_FORCE_DIST=1: process
     group, the library's RCCL communicator (one rank), in-stream collectives of svgp_mnist_train_step_dp -- the whole
     multi-rank code path of the driver on a 1-GPU box -- reproduces oracle.train_trajectory like the in-process driver test;
 (c) virtual ranks G = 2, 3 on one GPU: an epoch 256 + 256 + 210 cut with dp.shard_batch (the ragged batch 105 + 105 / 70 x 3)
     equals the single-engine epoch."""
import json
import math
import os
import pickle
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import svgpvae_oracle as O
from tests import helpers as H
from tests.test_gpu_dp_virtual import _lockstep

pytestmark = pytest.mark.gpu
DT = torch.float64
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("G", [2, 3])
def test_virtual_rank_epoch_with_ragged_batch_equals_single_engine(G):
    from svgp_vae_amd.dp import shard_batch
    from svgp_vae_amd.utils import batches
    N = 722                                           # 256 + 256 + 210: the ragged last batch of the reference's 4050-row epoch
    params, images, aux, eps = H.toy_problem(b=N, m=32, L=16, M=8, n_obj=400, seed=11)
    spans = batches(N, 256)
    assert [hi - lo for lo, hi in spans] == [256, 256, 210]
    cap = -(-256 // G)
    single = H.engine_for(params, 256, geco=True, N_train=float(N))
    dev = single.device
    di, da, de = images.to(dev), aux.to(dev), eps.to(dev)
    ranks = [H.engine_for(params, cap, geco=True, N_train=float(N), rank=r, world_size=G) for r in range(G)]
    for lo, hi in spans:
        single.set_batch_size(hi - lo)
        single.bind(di[lo:hi].contiguous(), da[lo:hi].contiguous(), de[lo:hi].contiguous())
        single.run(adam=True)
        single.synchronize()
        for r, e in enumerate(ranks):
            llo, lhi = shard_batch(lo, hi, G, r)
            e.set_batch_size(lhi - llo, hi - lo)
            e.bind(di[llo:lhi].contiguous(), da[llo:lhi].contiguous(), de[llo:lhi].contiguous())
        _lockstep(ranks, adam=True)
        ref = single.scalars()
        for e in ranks:
            sc = e.scalars()
            for k in ("elbo", "recon_loss", "kl_term", "c_ma", "lagrange", "adam_t"):
                assert abs(sc[k] - ref[k]) <= 1e-9 * max(1.0, abs(ref[k])), (hi - lo, k, sc[k], ref[k])
    for e in ranks:
        assert H.relerr(e.theta, single.theta) < 1e-9
        assert torch.equal(e.theta, ranks[0].theta)


@pytest.mark.parametrize("GECO,libcomm", [(True, True), (False, True), (True, False)])
def test_driver_under_torchrun_with_one_rank_takes_the_multi_rank_path_and_matches_the_oracle(golden, tmp_path, GECO, libcomm):
    gin, _ = golden
    d = str(tmp_path) + "/"
    pickle.dump({"images": gin["images"][:640], "aux_data": gin["aux"][:640]}, open(d + "train_data3.p", "wb"))
    for name, sl in (("eval_data3.p", slice(0, 64)), ("test_data3.p", slice(64, 128))):
        pickle.dump({"images": gin["images"][sl], "aux_data": gin["aux"][sl]}, open(d + name, "wb"))
    pickle.dump(gin["object_vectors"], open(d + "pca_ov_init3.p", "wb"))
    argv = ["--elbo", "SVGPVAE_Hensman", "--mnist_data_path", d, "--train_file", d + "train_data3.p", "--ip_joint", "--GP_joint",
            "--ov_joint", "--clip_qs", "--PCA", "--opt_regime", "joint-2", "--eval_every", "2", "--lr", "0.002", "--seed", "3",
            "--epsilon_seed", "7", "--log_json", d + "log.json", "--save", "--base_dir", d] + (["--GECO"] if GECO else [])
    env = dict(os.environ, SVGP_FORCE_DIST="1", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    if not libcomm:        # the library communicator refused on the rank: the collective vote sends the driver to torch.distributed
        env["SVGP_BENCH_FAIL_LIBCOMM"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "-m", "svgp_vae_amd.MNIST_experiment"] + argv
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert ("Data parallel over 1 ranks: " + ("RCCL communicator of the library" if libcomm else "torch.distributed all-reduces")) \
        in r.stdout, r.stdout[-2000:]
    log = json.load(open(d + "log.json"))
    assert log["rccl_ranks"] == (1 if libcomm else 0) and [s["rows"] for s in log["steps"]] == [256, 256, 128] * 2
    assert len(log["cgen_mse"]) == 1 and np.isfinite(log["cgen_mse"][0][1])          # rank 0 evaluated after the last epoch
    import glob
    files = glob.glob(d + "debug_MNIST/*/pics/test_metrics.txt")
    assert files and len(open(files[0]).read().strip().splitlines()) == 1
    # ---- the oracle on the same inputs (as tests/test_gpu_api.py::test_cli_driver_epoch_trajectory_matches_oracle)
    from svgp_vae_amd.utils import generate_init_inducing_points
    params = {k: torch.tensor(v, dtype=DT) for k, v in O.glorot_uniform_init(16, seed=3).items()}
    aux640 = np.asarray(gin["aux"][:640])
    params["inducing_index_points"] = torch.tensor(generate_init_inducing_points(None, n=2, PCA=True, M=8, aux_data=aux640), dtype=DT)
    params["l_GP"], params["amplitude"] = torch.tensor(1.0, dtype=DT), torch.tensor(1.0, dtype=DT)
    params["object_vectors"] = torch.tensor(gin["object_vectors"], dtype=DT)
    img, aux = torch.tensor(gin["images"][:640], dtype=DT), torch.tensor(gin["aux"][:640], dtype=DT)
    spans = [(0, 256), (256, 512), (512, 640)]
    eps_of = lambda epoch, i, b, L: np.random.RandomState(1000 * epoch + i + 7).randn(b, L)
    bts = [(img[lo:hi], aux[lo:hi]) for _ in range(2) for lo, hi in spans]
    epsilons = [torch.tensor(eps_of(e, i, hi - lo, 16), dtype=DT) for e in range(2) for i, (lo, hi) in enumerate(spans)]
    olog, oparams, _, _ = O.train_trajectory(params, bts, epsilons, beta=0.001, lr=0.002, alpha_flag=0.99,
                                             kappa=math.sqrt(0.020), clipping_qs=True, GECO=GECO, jitter=1e-6,
                                             N_train=640.0, L=16, formulation="efficient")
    for t, (got, want) in enumerate(zip(log["steps"], olog)):
        for k in ("elbo", "recon_loss", "C_ma", "lagrange_mult"):
            assert abs(got[k] - want[k]) <= 1e-8 * max(1.0, abs(want[k])), (t, k, got[k], want[k])
    theta = torch.tensor(log["theta"], dtype=DT)
    want = torch.cat([oparams[k].reshape(-1) for k in log["param_order"]])
    assert H.relerr(theta, want) < 1e-6 and log["adam_t"] == 6.0


def test_sprites_driver_under_torchrun_with_one_rank_equals_the_plain_run(tmp_path):
    """`python -m svgp_vae_amd.launch --nproc-per-node 1 -m svgp_vae_amd.SPRITES_experiment` with SVGP_FORCE_DIST=1 (process group,
    1-rank RCCL communicator, every exchange point of SpritesStepEngine.step issued, batches cut in whole character groups,
    rank-0 evaluation + barrier) against the same command without a launcher: per-step ELBO log and final parameters."""
    d = str(tmp_path) + "/"
    argv = ["--elbo", "SVGPVAE_Hensman", "--synthetic", "6,2", "--N_actions", "8", "--frames_per_character", "5",
            "--batch_size", "10", "--batch_size_test_char", "16", "--N_context", "3", "--L", "8", "--L_action", "8",
            "--L_character", "16", "--m", "2", "--K_SE", "--GECO", "--clip_qs", "--clip_grad", "--ip_joint", "--GPLVM_joint",
            "--GP_joint", "--opt_regime", "joint-2", "--eval_every", "2", "--lr", "0.002", "--epsilon_seed", "5", "--save", "--base_dir"]
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), HSA_ENABLE_IPC_MODE_LEGACY="0")
    logs = {}
    launch = lambda: [sys.executable, "-m", "svgp_vae_amd.launch", "--nproc-per-node", "1", "--master-port", str(_free_port())]
    for name, launcher, extra in (("plain", [sys.executable], {}),
                                  # (svgp_vae_amd.launch: torchrun's own parser rejects the reference's `--m` as ambiguous)
                                  ("dist", launch(), dict(SVGP_FORCE_DIST="1")),
                                  # the library communicator refused: every exchange point through dp.TorchDistComm
                                  ("fallback", launch(), dict(SVGP_FORCE_DIST="1", SVGP_BENCH_FAIL_LIBCOMM="1"))):
        out = d + name
        os.makedirs(out)
        cmd = launcher + ["-m", "svgp_vae_amd.SPRITES_experiment"] + argv + [out, "--log_json", out + "/log.json"]
        r = subprocess.run(cmd, env=dict(env, **extra), cwd=ROOT, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        if name == "dist":
            assert "Data parallel over 1 ranks: RCCL communicator of the library" in r.stdout, r.stdout[-2000:]
        if name == "fallback":
            assert "Data parallel over 1 ranks: torch.distributed collectives" in r.stdout, r.stdout[-2000:]
        logs[name] = json.load(open(out + "/log.json"))
        import glob
        files = glob.glob(out + "/debug_SPRITES/*/pics/test_metrics.txt")
        assert files and len(open(files[0]).read().strip().splitlines()) == 1
    a = logs["plain"]
    assert logs["dist"]["rccl_ranks"] == 1 and a["rccl_ranks"] == 0 and logs["fallback"]["rccl_ranks"] == 0
    for b in (logs["dist"], logs["fallback"]):
        assert len(a["steps"]) == len(b["steps"]) == 6
        for sa, sb in zip(a["steps"], b["steps"]):
            assert sb["local_rows"] == sb["rows"] == 10
            for k in ("elbo", "recon_loss", "C_ma", "lagrange_mult"):
                assert abs(sa[k] - sb[k]) <= 1e-9 * max(1.0, abs(sa[k])), (k, sa[k], sb[k])
        assert H.relerr(torch.tensor(b["theta"], dtype=DT), torch.tensor(a["theta"], dtype=DT)) < 1e-9
    b = logs["dist"]
    # (the evaluation draws its context frames and N(0,1) samples from generators whose state differs between the two processes)
    assert np.isfinite(b["cgen_mse"][0][1]) and abs(a["cgen_mse"][0][1] - b["cgen_mse"][0][1]) <= 0.05 * abs(a["cgen_mse"][0][1])

"""svgp_vae_amd.launch (the one-node launcher of the data-parallel drivers) without a GPU: rendezvous environment per rank,
verbatim pass-through of the driver's flags (incl. the reference's `--m`, which torchrun's parser rejects), exit codes."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tmp_path, body, args, n=2):
    (tmp_path / "probe_mod.py").write_text(textwrap.dedent(body))
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([str(tmp_path), ROOT, os.environ.get("PYTHONPATH", "")]))
    return subprocess.run([sys.executable, "-m", "svgp_vae_amd.launch", "--nproc-per-node", str(n), "-m", "probe_mod"] + args,
                          env=env, cwd=ROOT, capture_output=True, text=True, timeout=120)


def test_ranks_get_the_rendezvous_environment_and_the_flags_verbatim(tmp_path):
    r = _run(tmp_path, """
        import os, sys
        print("R", os.environ["RANK"], os.environ["LOCAL_RANK"], os.environ["WORLD_SIZE"], os.environ["MASTER_ADDR"],
              os.environ["MASTER_PORT"], os.environ["HSA_ENABLE_IPC_MODE_LEGACY"], " ".join(sys.argv[1:]), flush=True)
        """, ["--elbo", "SVGPVAE_Hensman", "--m", "3", "--master-port", "x"], n=3)
    assert r.returncode == 0, r.stderr
    rows = sorted(l.split() for l in r.stdout.splitlines() if l.startswith("R "))
    assert [x[1] for x in rows] == ["0", "1", "2"] and all(x[2] == x[1] and x[3] == "3" and x[4] == "127.0.0.1" for x in rows)
    assert len({x[5] for x in rows}) == 1 and all(x[6] == "0" for x in rows)
    assert all(x[7:] == ["--elbo", "SVGPVAE_Hensman", "--m", "3", "--master-port", "x"] for x in rows)


def test_a_failing_rank_ends_the_others_and_sets_the_exit_code(tmp_path):
    r = _run(tmp_path, """
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit(7)
        time.sleep(60)
        """, [])
    assert r.returncode == 7


def test_torchrun_rejects_the_reference_flag_that_the_launcher_passes(tmp_path):
    """Why the launcher exists: `--m` (SPRITES_experiment.py:52) is an ambiguous abbreviation for torch.distributed.run."""
    (tmp_path / "probe_mod.py").write_text("print('ok')\\n")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([str(tmp_path), os.environ.get("PYTHONPATH", "")]))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
                        "127.0.0.1", "--master-port", "29999", "-m", "probe_mod", "--m", "3"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert r.returncode != 0 and "ambiguous option: --m" in r.stderr

"""`python bench.py --gpus N` without a launcher: the process must start
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <same args>`
as a CHILD before anything touches the GPU, pass the ranks' stdout through, keep the result line as its own LAST stdout line
and exit with the launcher's return code.  The launcher module is replaced by a stub (SVGP_BENCH_LAUNCHER) that records its
command line and plays a scripted rank output, so the test needs neither a GPU nor a rendezvous."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

STUB = textwrap.dedent('''
    """Stand-in for torch.distributed.run: records argv + the environment the ranks would inherit, prints a script."""
    import json, os, sys
    rec = dict(argv=sys.argv[1:], world_size_in_env="WORLD_SIZE" in os.environ,
               ipc=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"))
    json.dump(rec, open(os.environ["STUB_RECORD"], "w"))
    for ln in json.loads(os.environ["STUB_LINES"]):
        print(ln, flush=True)
    sys.exit(int(os.environ.get("STUB_RC", "0")))
''')

STRONG = json.dumps({"metric": "m", "value": 1.0, "scaling": "strong"})
WEAK = json.dumps({"metric": "m", "value": 2.0, "scaling": "weak"})


def launch(tmp_path, args, lines, rc=0, env_extra=None):
    (tmp_path / "fake_torchrun.py").write_text(STUB)
    rec = tmp_path / "rec.json"
    env = dict(os.environ, SVGP_BENCH_LAUNCHER="fake_torchrun", STUB_RECORD=str(rec), STUB_LINES=json.dumps(lines),
               STUB_RC=str(rc), PYTHONPATH=str(tmp_path) + os.pathsep + os.environ.get("PYTHONPATH", ""),
               CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, capture_output=True, text=True,
                       timeout=300)
    return r, (json.load(open(rec)) if rec.exists() else None)


def test_gpus_2_spawns_the_launcher_with_the_same_arguments_and_keeps_the_json_last(tmp_path):
    args = ["--gpus", "2", "--steps", "7", "--warmup", "3", "--workload", "cfg2"]
    r, rec = launch(tmp_path, args, ["NCCL version 2.x banner", STRONG, WEAK, "trailing RCCL chatter"])
    assert r.returncode == 0, r.stderr
    a = rec["argv"]
    assert a[:2] == ["--nnodes=1", "--nproc-per-node=2"]
    assert a[2:4] == ["--master-addr", "127.0.0.1"] and a[4] == "--master-port" and 1024 < int(a[5]) < 65536
    assert os.path.samefile(a[6], os.path.join(ROOT, "bench.py")) and a[7:] == args
    assert rec["world_size_in_env"] is False and rec["ipc"] == "0"
    out = [ln for ln in r.stdout.splitlines() if ln.strip()]
    # both result lines passed through in order; the weak line is (again) the last one although chatter followed it
    assert out.index(STRONG) < out.index(WEAK)
    assert "trailing RCCL chatter" in out and out[-1] == WEAK
    assert json.loads(out[-1])["scaling"] == "weak"


def test_json_already_last_is_not_repeated(tmp_path):
    r, _ = launch(tmp_path, ["--gpus", "4"], ["banner", WEAK])
    assert r.returncode == 0
    assert [ln for ln in r.stdout.splitlines() if ln.strip()] == ["banner", WEAK]


@pytest.mark.parametrize("rc", [1, 3])
def test_return_code_of_the_ranks_is_the_parents(tmp_path, rc):
    r, rec = launch(tmp_path, ["--gpus", "8", "--steps", "5"], ["rank 3 died"], rc=rc)
    assert r.returncode == rc and rec["argv"][1] == "--nproc-per-node=8"


def test_ranks_that_print_no_result_are_a_failure(tmp_path):
    r, _ = launch(tmp_path, ["--gpus", "2"], ["nothing useful"], rc=0)
    assert r.returncode == 1 and "no result line" in r.stderr


def test_force_dist_on_one_gpu_takes_the_same_path(tmp_path):
    r, rec = launch(tmp_path, ["--gpus", "1", "--force-dist", "--steps", "20", "--warmup", "5"], [WEAK])
    assert r.returncode == 0 and rec["argv"][1] == "--nproc-per-node=1" and "--force-dist" in rec["argv"]


def test_under_a_launcher_nothing_is_spawned(tmp_path):
    # WORLD_SIZE present = we ARE a rank: no second launcher; a mismatch with --gpus is an argument error
    r, rec = launch(tmp_path, ["--gpus", "2"], [WEAK], env_extra={"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"})
    assert rec is None and r.returncode != 0 and "--gpus 2 but WORLD_SIZE=4" in r.stderr

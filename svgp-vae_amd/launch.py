"""One-node launcher of the data-parallel drivers: starts one process per GPU with the rendezvous environment
torch.distributed reads (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR = 127.0.0.1, MASTER_PORT), before anything has touched a GPU.

    python -m svgp_vae_amd.launch --nproc-per-node 8 -m svgp_vae_amd.SPRITES_experiment --elbo SVGPVAE_Hensman --m 3 ...

Equivalent to `python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 -m <module> ...`, which the
drivers accept as well -- but torchrun's argument parser scans the DRIVER's flags too and rejects the reference's `--m`
(SPRITES_experiment.py:52) as an ambiguous abbreviation of its own `--max-restarts / --master-addr / --module / ...`; everything
after `-m <module>` is passed through verbatim here.  The ranks are CHILD processes (never an exec of a process that holds a
GPU); the first failing rank ends the others; the exit code is the first non-zero one."""
import os
import signal
import socket
import subprocess
import sys
import time


def _free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def parse(argv):
    n, port, i = 1, None, 0
    while i < len(argv):
        a = argv[i]
        if a in ("--nproc-per-node", "--nproc_per_node"):
            n, i = int(argv[i + 1]), i + 2
        elif a.startswith("--nproc-per-node=") or a.startswith("--nproc_per_node="):
            n, i = int(a.split("=", 1)[1]), i + 1
        elif a in ("--master-port", "--master_port"):
            port, i = int(argv[i + 1]), i + 2
        elif a == "-m":
            if i + 1 >= len(argv):
                raise SystemExit("launch: -m needs a module name")
            return n, port, argv[i + 1], argv[i + 2:]
        else:
            raise SystemExit(f"launch: unknown option {a!r} (usage: --nproc-per-node G [--master-port P] -m module [module args])")
    raise SystemExit("launch: no -m <module> given")


def main(argv=None):
    n, port, module, rest = parse(list(sys.argv[1:] if argv is None else argv))
    port = port or _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: the only mode this pool's host driver supports
        env.setdefault("OMP_NUM_THREADS", "8")
        procs.append(subprocess.Popen([sys.executable, "-m", module, *rest], env=env))
    rc = 0
    try:
        while procs:
            for p in list(procs):
                code = p.poll()
                if code is None:
                    continue
                procs.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    for q in procs:                     # a rank failed: the others would wait in a collective for ever
                        q.send_signal(signal.SIGTERM)
            time.sleep(0.05)
    except KeyboardInterrupt:
        for q in procs:
            q.send_signal(signal.SIGTERM)
        rc = 130
    return rc


if __name__ == "__main__":
    raise SystemExit(main())

#!/usr/bin/env python3
"""`python -m root_digger_amd.cli ...`: a LAUNCHER of the native front end `bin/rd_amd`
(csrc/tools/rd_main.cpp -- the `rd --msa M --tree T [--exhaustive]` command line of the reference,
/root/reference/src/main.cpp:411-680, on the C ABI).  There is one search driver, the native one;
this module passes its arguments through and supplies the one thing a Python environment adds:
the rank environment.

  python -m root_digger_amd.cli --msa aln.fasta --tree t.nwk --prefix out --exhaustive \\
         [--rate-cats 4] [--lbfgsb /path/to/liblbfgsb.so] [--early-stop] [--gpus N [--site-shards G]]

* under `python -m torch.distributed.run --nproc-per-node N -m root_digger_amd.cli ...` every rank
  execs nothing and initialises nothing here: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR /
  MASTER_PORT go to rd_amd as they are (it splits the candidate roots like the reference's MPI
  ranks, src/model.cpp:1867-1911, and meets the others over TCP and in <prefix>.ckp);
* `--gpus N` without a launcher: N children, one per GPU, with that environment on 127.0.0.1;
* `--workers` is accepted as a synonym of the reference's `--threads`.
Everything else -- options, defaults, outputs -- is rd_amd's (`--help`)."""
import os
import socket
import subprocess
import sys

RD_AMD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "rd_amd")


def _run(args, env=None, relay=True):
    """rd_amd as a child (never exec: see the GPU box's rules on replacing a process that has
    touched the device); its stdout goes through sys.stdout so that callers may capture it"""
    p = subprocess.Popen([RD_AMD] + args, env=env, stdout=subprocess.PIPE if relay else subprocess.DEVNULL, text=True)
    if relay:
        for line in p.stdout:
            sys.stdout.write(line)
        sys.stdout.flush()
    return p


def main(argv=None):
    args = list(sys.argv[1:] if argv is None else argv)
    if not os.path.exists(RD_AMD):
        raise SystemExit("root_digger_amd: %s is missing (make -C root_digger_amd/csrc)" % RD_AMD)
    args = ["--threads" if a == "--workers" else a for a in args]
    gpus = 1
    if "--gpus" in args:
        i = args.index("--gpus")
        gpus = int(args[i + 1])
        del args[i:i + 2]
    if gpus <= 1 or "WORLD_SIZE" in os.environ:
        return _run(args).wait()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, WORLD_SIZE=str(gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    others = [_run(args, dict(env, RANK=str(r), LOCAL_RANK=str(r)), relay=False) for r in range(1, gpus)]
    codes = [_run(args, dict(env, RANK="0", LOCAL_RANK="0")).wait()] + [p.wait() for p in others]
    return next((c for c in codes if c), 0)


if __name__ == "__main__":
    sys.exit(main())

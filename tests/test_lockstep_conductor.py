"""The protocol of lock step in deterministic rounds (csrc/lockstep_conductor.hpp) on the CPU: G
simulated ranks of a site group x W candidates in flight each, the library and the HIP runtime
replaced by host stand-ins (tests/cpp/conductor_check.cpp), a reducer that REFUSES rounds whose
length differs between the ranks.  What a multi-GPU site group relies on -- north star: "site
blocks shard across the 8 GPUs ... with an RCCL all-reduce of per-block log-likelihoods" under a
batched outer loop (src/model.cpp:1154-1229) -- is that thread timing changes nothing: the same
rounds, the same collectives, the same values, the same candidate on the same worker, whatever
the delays.  Also run under ThreadSanitizer."""
import os
import shutil
import subprocess

import pytest

import util

SRC = os.path.join(util.ROOT, "tests", "cpp", "conductor_check.cpp")
INC = ["-I", os.path.join(util.ROOT, "root_digger_amd", "csrc"), "-I", os.path.join(util.ROOT, "include"),
       "-I", "/opt/rocm/include", "-D__HIP_PLATFORM_AMD__"]


def _build(tmp_path, name, extra):
    exe = str(tmp_path / name)
    out = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-pthread", "-Wall"] + extra + INC + [SRC, "-o", exe],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-3000:]
    return exe


def _run(exe, *args, timeout=120):
    out = subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0 and out.stdout.startswith("conductor OK"), out.stdout + out.stderr[-2000:]
    return dict(kv.split("=") for kv in out.stdout.split()[2:]), out.stderr


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
@pytest.mark.parametrize("ranks,workers,groups,candidates", [(2, 6, 2, 23), (4, 5, 1, 11), (3, 8, 2, 5), (2, 4, 2, 40),
                                                            (8, 4, 2, 9), (2, 1, 1, 3)])
def test_rounds_do_not_depend_on_thread_timing(tmp_path, ranks, workers, groups, candidates):
    exe = _build(tmp_path, "conductor_check", [])
    got = [_run(exe, ranks, workers, groups, candidates, seed)[0] for seed in (1, 2, 3)]
    assert got[0] == got[1] == got[2], got
    assert int(got[0]["collectives"]) <= int(got[0]["rounds"])
    # rounds that only hand out candidates carry no collective; every other one carries exactly one
    assert int(got[0]["rounds"]) - int(got[0]["collectives"]) <= 2 * (candidates // max(1, workers) + 2)


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_conductor_is_clean_under_thread_sanitizer(tmp_path):
    exe = _build(tmp_path, "conductor_check_tsan", ["-fsanitize=thread"])
    got, err = _run(exe, 2, 6, 2, 23, 5, timeout=300)
    assert "ThreadSanitizer" not in err, err[-3000:]
    plain = _run(_build(tmp_path, "conductor_check", []), 2, 6, 2, 23, 5)[0]
    assert got == plain

"""The protocol of lock step in deterministic rounds (csrc/lockstep_conductor.hpp) on the CPU: G
simulated ranks of a site group x W candidates in flight each, the library and the HIP runtime
replaced by host stand-ins (tests/cpp/conductor_check.cpp), a reducer that REFUSES rounds whose
length differs between the ranks.  What a multi-GPU site group relies on -- north star: "site
blocks shard across the 8 GPUs ... with an RCCL all-reduce of per-block log-likelihoods" under a
batched outer loop (src/model.cpp:1154-1229) -- is that thread timing changes nothing: the same
rounds, the same collectives, the same values, the same candidate on the same worker, whatever
the delays.  Also run under ThreadSanitizer.

The DEVICE path (what real RCCL runs take: stream-ordered batches whose second-pass flag travels
through the sum, the repeat of a collective in its worker group's turn, a reducer in two halves)
runs the same way with flags raised at random, differently on every rank; the rendezvous asserts
that all ranks issue the same sequence of collectives, repeats included.

The DIVERGENCE GUARD: a rank whose copy of one sum is off by one ulp (a reducer that does not
hand every rank the same bits) must fail the run on every rank at once, naming the round --
never leave the group in a collective that no longer matches."""
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


def _run_raw(exe, *args, timeout=120):
    return subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True, timeout=timeout)


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


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
@pytest.mark.parametrize("ranks,workers,groups,candidates", [(2, 6, 2, 23), (4, 5, 1, 11), (3, 8, 2, 9), (8, 4, 2, 9)])
def test_device_path_rounds_with_random_second_pass_flags(tmp_path, ranks, workers, groups, candidates):
    """flags differ per rank: every rank must repeat the collective together, in the same place of
    the order (A0 B0 A0' A1 ...), and use the values of the repeat"""
    exe = _build(tmp_path, "conductor_check", [])
    got = [_run(exe, ranks, workers, groups, candidates, seed, "device")[0] for seed in (1, 2, 3)]
    assert got[0] == got[1] == got[2], got
    assert int(got[0]["redos"]) > 0
    # the same candidates, the same values as the host path: where a sum is made changes nothing
    host = _run(exe, ranks, workers, groups, candidates, 1, "host")[0]
    assert host["digest"] == got[0]["digest"] and int(host["redos"]) == 0


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_device_path_is_clean_under_thread_sanitizer(tmp_path):
    exe = _build(tmp_path, "conductor_check_tsan", ["-fsanitize=thread"])
    got, err = _run(exe, 3, 6, 2, 17, 4, "device", timeout=300)
    assert "ThreadSanitizer" not in err, err[-3000:]
    assert int(got["redos"]) > 0


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
@pytest.mark.parametrize("mode", ["host", "device"])
@pytest.mark.parametrize("ranks,workers,groups,fault", [(2, 6, 2, "1:7"), (4, 5, 1, "2:3"), (3, 4, 2, "0:40")])
def test_one_ulp_on_one_rank_fails_every_rank_at_once(tmp_path, mode, ranks, workers, groups, fault):
    exe = _build(tmp_path, "conductor_check", [])
    out = _run_raw(exe, ranks, workers, groups, 23, 1, mode, fault, timeout=30)   # (seconds, not a comm timeout)
    assert out.returncode == 1 and "conductor FAILED" in out.stdout, out.stdout + out.stderr[-2000:]
    named = [ln for ln in out.stderr.splitlines() if "has diverged" in ln and "lock-step round" in ln]
    assert named and "did not receive the same bits" in named[0], out.stderr[-2000:]


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_failure_path_is_clean_under_address_sanitizer(tmp_path):
    """a round that fails while the OTHER worker group's round is under way: the posters whose requests
    that round holds must stay until it ends (their requests live on their stacks) -- round 6: one run in
    five corrupted the heap there.  Every fault position of the test above, repeated, under ASan."""
    exe = _build(tmp_path, "conductor_check_asan", ["-fsanitize=address"])
    for mode in ("host", "device"):
        for ranks, workers, groups, fault in [(2, 6, 2, "1:7"), (4, 5, 1, "2:3"), (3, 4, 2, "0:40")]:
            for _ in range(6):
                out = _run_raw(exe, ranks, workers, groups, 23, 1, mode, fault, timeout=60)
                assert "Sanitizer" not in out.stderr, out.stderr[-3000:]
                assert out.returncode == 1 and "conductor FAILED" in out.stdout, out.stdout + out.stderr[-2000:]

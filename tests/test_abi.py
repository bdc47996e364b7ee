"""The C-ABI library loads on a CPU-only box and exports every symbol that
include/root_digger_amd.h declares (no compute calls are made here)."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

import root_digger_amd as rd
import util

HEADER = os.path.join(util.ROOT, "include", "root_digger_amd.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    funcs = set(re.findall(r"\b(rdamd_[a-z0-9_]+)\s*\(", text))
    data = set(re.findall(r"extern\s+const\s+\w+\s+(rdamd_[a-z0-9_]+)\s*\[", text))
    return funcs, data


def test_every_declared_symbol_is_exported():
    funcs, data = declared_symbols()
    assert len(funcs) >= 50
    lib = ctypes.CDLL(rd.lib_path)
    missing = [s for s in sorted(funcs | data) if not hasattr(lib, s)]
    assert not missing, missing


def test_no_torch_types_in_the_abi():
    text = open(HEADER).read()
    assert "torch" not in text.lower().replace("no torch", "")
    assert "hipStream_t" not in text and "at::" not in text


def test_header_cites_reference_interfaces():
    text = open(HEADER).read()
    assert text.count("src/model.cpp") >= 15 and text.count("src/tree.cpp") >= 8


def test_partition_create_fails_loudly_without_gpu():
    if rd.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(rd.RdamdError) as e:
        rd.Partition(4, 6, 4, 10, 1, 6, 1, 6)
    assert "no CPU fallback" in str(e.value)


def test_product_does_not_link_the_oracle():
    import subprocess
    out = subprocess.run(["ldd", rd.lib_path], capture_output=True, text=True).stdout
    assert "oracle" not in out
    for root, _, files in os.walk(os.path.join(util.ROOT, "root_digger_amd")):
        if "/build" in root:
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h")):
                src = open(os.path.join(root, f), errors="ignore").read()
                assert "rd_oracle" not in src and "oracle_lib" not in src, f


def test_product_library_has_no_timing_only_switches():
    """the ablation variants (kernels with stores / loads removed, chosen by
    RDAMD_K20_VAR / RDAMD_FUSED_NS) exist only behind -DRDAMD_ABLATION in a second library;
    librdamd.so must not even know the variable names."""
    blob = open(rd.lib_path, "rb").read()
    assert os.path.basename(rd.lib_path) == "librdamd.so"
    for name in (b"RDAMD_K20_VAR", b"RDAMD_FUSED_NS", b"RDAMD_FUSED_DEPTH", b"RDAMD_FUSED_RL", b"RDAMD_FUSED_RW",
                 b"RDAMD_FUSED_SPILL_MIN"):
        assert name not in blob, name


def test_package_loader_knows_one_library():
    """no environment variable may swap the product library for another
    build (the ablation library computes garbage by design); A/B tooling hands the package a
    pre-loaded module instead (profiles/with_ablation.py)."""
    src = open(os.path.join(util.ROOT, "root_digger_amd", "_lib.py")).read()
    assert "environ" not in src and "getenv" not in src
    env = dict(os.environ, RDAMD_LIBRARY="/nonexistent/librdamd_ablation.so")
    out = subprocess.run([sys.executable, "-c", "import root_digger_amd as rd; print(rd.lib_path)"],
                         cwd=util.ROOT, env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert out.stdout.strip().endswith(os.path.join("root_digger_amd", "lib", "librdamd.so"))


def test_host_side_gamma_and_maps_work_without_gpu():
    r = rd.compute_gamma_cats(1.0, 4)
    assert abs(sum(r) / 4 - 1.0) < 1e-12
    assert rd.MAP_NT[ord("A")] == 1 and rd.MAP_NT[ord("t")] == 8 and rd.MAP_NT[ord("-")] == 15
    for g in util.golden("gamma.json"):
        got = rd.compute_gamma_cats(g["alpha"], g["cats"],
                                    rd.GAMMA_RATES_MEAN if g["mode"] == "mean" else rd.GAMMA_RATES_MEDIAN)
        assert max(abs(a - b) for a, b in zip(got, g["rates"])) < 2e-9


def test_k20_operand_layouts_are_consistent(tmp_path):
    """The 20-state CLV tile / tip-table row layouts of common.hpp (what keeps the
    matrix-core traversal kernel's loads and stores contiguous): bijective, and every
    lane's operands where the kernel's three instructions expect them.  Host code only."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    exe = str(tmp_path / "k20_layout_check")
    subprocess.check_call([hipcc, "-std=c++17", "-O1", "--offload-arch=gfx950", "-Wno-c99-designator", "-w",
                           "-I", os.path.join(util.ROOT, "root_digger_amd", "csrc"),
                           "-I", os.path.join(util.ROOT, "include"),
                           os.path.join(util.ROOT, "tests", "cpp", "k20_layout_check.cpp"), "-o", exe])
    out = subprocess.run([exe], stdout=subprocess.PIPE, text=True)
    assert out.returncode == 0 and "k20 layouts OK" in out.stdout, out.stdout


def test_two_hip_runtimes_are_detected():
    """bench.py hands torch device pointers to librdamd only when both sit on ONE libamdhip64
    (check_one_hip_runtime).  The helpers behind that check, in both import orders: torch
    first, librdamd binds to torch's bundled runtime (one instance); librdamd first, torch
    brings a second one."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n%s\n"
            "print(len(rd.mapped_hip_runtimes()), rd.hip_runtime_path() in rd.mapped_hip_runtimes())")
    first = subprocess.run([sys.executable, "-c", code % (util.ROOT, "import torch\nimport root_digger_amd as rd")],
                           capture_output=True, text=True, timeout=300)
    assert first.returncode == 0, first.stderr
    n, inside = first.stdout.split()
    assert inside == "True" and int(n) >= 1
    if int(n) == 1:   # (an image whose torch uses the system ROCm has one runtime either way)
        second = subprocess.run([sys.executable, "-c",
                                 code % (util.ROOT, "import root_digger_amd as rd\nimport torch")],
                                capture_output=True, text=True, timeout=300)
        assert second.returncode == 0, second.stderr
        assert second.stdout.split()[1] == "True"

"""Pure host logic of round 3, without a GPU: the subtree pattern-class census behind the
clade tables (csrc/clade_classes.hpp), the cutting of a 20-state operation list into
side-by-side pieces (csrc/k20_split.hpp) and the traversal compiler of the fused evaluators
(csrc/traversal_compiler.hpp: its programs replayed symbolically), checked by
tests/cpp/host_logic_check.cpp."""
import os
import subprocess

import util


def test_clade_classes_k20_split_and_traversal_compiler(tmp_path):
    exe = str(tmp_path / "host_logic_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(util.ROOT, "root_digger_amd", "csrc"),
                           "-I", os.path.join(util.ROOT, "include"),
                           # (fused.hpp includes the HIP runtime header for its prototypes: host mode)
                           "-I", "/opt/rocm/include", "-D__HIP_PLATFORM_AMD__",
                           os.path.join(util.ROOT, "tests", "cpp", "host_logic_check.cpp"), "-o", exe])
    out = subprocess.run([exe], stdout=subprocess.PIPE, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.startswith("host logic OK"), out.stdout
    assert int(out.stdout.split()[-1]) >= 1200


def test_host_logic_under_address_and_undefined_behaviour_sanitizers(tmp_path):
    """the same checks (the list cuts, their levels, the traversal compiler with its parks placed one by
    one) in an AddressSanitizer + UBSan build: the host code that decides what every kernel launch of
    the path looks like"""
    exe = str(tmp_path / "host_logic_check_san")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-I", os.path.join(util.ROOT, "root_digger_amd", "csrc"), "-I", os.path.join(util.ROOT, "include"),
                           "-I", "/opt/rocm/include", "-D__HIP_PLATFORM_AMD__",
                           os.path.join(util.ROOT, "tests", "cpp", "host_logic_check.cpp"), "-o", exe])
    out = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.startswith("host logic OK"), (out.stdout, out.stderr[-2000:])


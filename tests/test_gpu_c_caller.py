"""The drop-in boundary from the caller's side: tests/cpp/drop_in.c is compiled
with plain gcc against include/root_digger_amd.h, linked to librdamd.so and
run -- the call sequence of the reference's model_t, no Python in between."""
import os
import subprocess

import pytest

import root_digger_amd as rd

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def test_plain_c_program_links_and_matches_the_closed_form(tmp_path):
    exe = str(tmp_path / "drop_in")
    libdir = os.path.dirname(rd.lib_path)
    subprocess.run(["gcc", "-std=c11", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(HERE, "cpp", "drop_in.c"), "-o", exe, "-L", libdir, "-lrdamd",
                    "-Wl,-rpath," + libdir, "-lm"], check=True)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.startswith("lnL -")

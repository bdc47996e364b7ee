#!/usr/bin/env python3
"""Run a script of this repository against ANOTHER build of the library -- the ablation build
(`make -C root_digger_amd/csrc ablation`: timing-only kernel variants, results are garbage) or a
library built from another commit (lib_ab.sh).  Tooling only: the package's own loader
(root_digger_amd/_lib.py) knows one path, the product library; this launcher hands the package
a pre-loaded `_lib` module instead.
Usage: python3 profiles/with_ablation.py <library.so> <script.py> [args ...]"""
import ctypes
import os
import runpy
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    if len(sys.argv) < 3:
        sys.exit(__doc__)
    path = os.path.abspath(sys.argv[1])
    if not os.path.exists(path):
        sys.exit("with_ablation.py: %s does not exist" % path)
    try:   # (a script that uses torch must meet ONE HIP runtime: torch's comes first, as when
        import torch  # noqa: F401   the package is imported after it -- bench.py checks)
    except ImportError:
        pass
    mod = types.ModuleType("root_digger_amd._lib")

    class RdamdError(RuntimeError):
        pass

    mod.RdamdError = RdamdError
    mod.lib_path = path
    mod.lib = ctypes.CDLL(path)
    sys.modules["root_digger_amd._lib"] = mod
    sys.path.insert(0, ROOT)
    script = sys.argv[2]
    sys.argv = sys.argv[2:]
    print("with_ablation.py: root_digger_amd runs on %s" % path, file=sys.stderr)
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()

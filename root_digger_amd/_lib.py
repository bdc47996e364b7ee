"""Locate and load librdamd.so (built in-tree by __graft_entry__.build())."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
lib_path = os.path.join(_HERE, "lib", "librdamd.so")


class RdamdError(RuntimeError):
    pass


if not os.path.exists(lib_path):
    raise ImportError(
        "root_digger_amd: %s is missing. Build it with "
        "`python -c 'import __graft_entry__ as g; g.build()'` or "
        "`make -C root_digger_amd/csrc` (needs hipcc). There is no CPU fallback." % lib_path)

lib = ctypes.CDLL(lib_path)

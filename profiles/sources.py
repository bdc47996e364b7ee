"""Which source files decide what a profiled kernel does -- and a digest of them.

bench.py publishes counters (HBM traffic, issue-slot figures) that were measured by
profiles/collect.sh + profiles/summarize.py in separate rocprofv3 passes and committed as
profiles/r*_summary.json.  They describe the kernel as it was THEN: summarize.py stores the
digest of the kernel's sources next to them, bench.py recomputes it and refuses to publish
counters of a different kernel; tests/test_profiles.py holds the
committed summary to the committed sources on every CPU run."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "root_digger_amd", "csrc")

KERNEL_SOURCES = {
    "fused_dna_eval_kernel": ["kernels_fused.hip", "fused.hpp", "evaluate.hip", "kernels_clade.hip", "clades.hpp"],
    "clv_dna_traversal_kernel": ["kernels_clv.hip"],
    "fused20_eval_kernel": ["kernels_fused_k20.hip", "fused.hpp", "evaluate.hip"],
    "clv_k20_traversal_kernel": ["kernels_clv_mfma.hip"],
}


def source_digest(kernel):
    """16 hex digits over the files behind `kernel` (a key of KERNEL_SOURCES or a longer,
    templated name that starts with one)."""
    for prefix, files in KERNEL_SOURCES.items():
        if kernel.startswith(prefix):
            h = hashlib.sha256()
            for f in files:
                h.update(f.encode())
                h.update(open(os.path.join(CSRC, f), "rb").read())
            return h.hexdigest()[:16]
    return None

"""The committed counter summary (profiles/r*_summary.json) must describe the kernels as they
are NOW: bench.py publishes its HBM-traffic and issue-slot figures from it and refuses --
exits non-zero -- when the digest of the kernel's sources stored with the counters differs from
the current sources.  This test holds the two together on every CPU run,
so a kernel change without a new profiles/collect.sh pass is caught here, not by the driver's
bench run."""
import glob
import json
import os
import sys

import util

sys.path.insert(0, os.path.join(util.ROOT, "profiles"))
from sources import KERNEL_SOURCES, source_digest   # noqa: E402


def latest_summary():
    files = sorted(glob.glob(os.path.join(util.ROOT, "profiles", "r*_summary.json")))
    assert files, "no committed counter summary"
    return files[-1], json.load(open(files[-1]))


def test_committed_counters_belong_to_the_current_kernel_sources():
    name, d = latest_summary()
    checked = 0
    for kernel in ("fused_dna_eval_kernel", "clv_dna_traversal_kernel"):   # what the default bench line cites
        hits = [v for k, v in d.items() if k.startswith(kernel) and "hbm_bytes_per_launch" in v]
        assert hits, (name, kernel)
        v = max(hits, key=lambda v: v.get("avg_us", 0.0))
        assert v.get("source_digest") == source_digest(kernel), (
            "%s: counters of %s were taken from other sources; re-run profiles/collect.sh + "
            "profiles/summarize.py on a GPU box" % (os.path.basename(name), kernel))
        checked += 1
    assert checked == 2
    # the 20-state evaluator's counters (bench.py --config c3 publishes roofline.mfma_busy from them)
    hits = [v for k, v in d.items() if k.startswith("fused20_eval_kernel") and v.get("command") == "c3"]
    assert hits and all(v.get("source_digest") == source_digest("fused20_eval_kernel") for v in hits), name
    assert all(0.0 < v["derived"]["mfma_busy"] <= 1.0 for v in hits)


def test_digest_covers_existing_files():
    for kernel, files in KERNEL_SOURCES.items():
        for f in files:
            assert os.path.exists(os.path.join(util.ROOT, "root_digger_amd", "csrc", f)), (kernel, f)
        assert len(source_digest(kernel)) == 16

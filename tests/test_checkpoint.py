"""The <prefix>.ckp result log (root_digger_amd/csrc/checkpoint.cpp) against the
independent pure-Python restatement of the reference's format
(oracle/ckp_oracle.py): byte-for-byte layout, the checksum quirks, recovery
from a torn tail, and concurrent appends from several processes -- the cases
test/src/checkpoint.cpp of the reference covers, plus the byte layout it does
not pin.  No GPU needed: the log is host code."""
import multiprocessing as mp
import os
import struct
import sys

import numpy as np
import pytest

import root_digger_amd as rd

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import ckp_oracle as orc  # noqa: E402

OPTIONS = {
    "msa_filename": "data/10.fasta", "tree_filename": "data/10.tree", "prefix": "run/x",
    "prefix_dir": "run", "model_filename": "", "freqs_filename": "f.txt",
    "partition_filename": "", "data_type": "nt", "model_string": "UNREST+G4",
    "rate_cats": [{"type": 1, "rate_category_type": 1, "rate_cats": 4, "alpha_init": 0, "alpha": 1.0},
                  {"type": 3, "rate_category_type": 2, "rate_cats": 2, "alpha_init": 1, "alpha": 0.37}],
    "seed": 0xDEADBEEFCAFE, "min_roots": 3, "threads": 2, "root_ratio": 0.05,
    "abs_tolerance": 1e-6, "factor": 1e5, "br_tolerance": 1e-11, "bfgs_tol": 1e-8,
    "silent": 1, "exhaustive": 1, "echo": 0, "invariant_sites": 0, "early_stop": 2,
    "initial_root_strategy": 1,
}


def params(seed, partitions=1, free_rates=False):
    rng = np.random.default_rng(seed)
    return [{"subst_rates": rng.uniform(1e-4, 1, 12).tolist(),
             "freqs": rng.dirichlet(np.ones(4)).tolist(),
             "gamma_alpha": [float(rng.uniform(0.2, 5))],
             "gamma_weights": rng.dirichlet(np.ones(4)).tolist() if free_rates else []}
            for _ in range(partitions)]


def test_checksums_match_the_restatement():
    for k in range(20):
        rng = np.random.default_rng(100 + k)
        rid, llh, alpha = int(rng.integers(0, 1 << 40)), float(-rng.uniform(1, 1e7)), float(rng.uniform())
        assert rd.checkpoint_checksum_result(rid, llh, alpha) == orc.checksum_result(rid, llh, alpha)
        pp = params(k, partitions=1 + k % 3, free_rates=k % 2 == 1)
        assert rd.checkpoint_checksum_params(pp) == orc.checksum_params(pp)
    assert rd.checkpoint_checksum_params([]) == 1          # a = 1, b = 0 untouched
    # the quirk is visible: an empty parameter set still changes the sum
    assert rd.checkpoint_checksum_params([{}]) == orc.checksum_params([{}]) != 1


def test_file_bytes_equal_the_hand_assembled_layout(tmp_path):
    c = rd.Checkpoint(str(tmp_path / "x"))
    assert not c.existing_checkpoint()
    assert c.get_filename() == str(tmp_path / "x.ckp")
    assert c.load_options() is None
    c.save_options(OPTIONS)
    records = [(7, -123456.789, 0.25, params(1)), (0, -98765.4321, 1.0, params(2, 2, True)),
               (196, -1.5e6, 0.0, params(3))]
    for r in records:
        c.write(*r)
    expect = orc.put_header(OPTIONS) + b"".join(orc.put_record(*r) for r in records)
    assert open(c.get_filename(), "rb").read() == expect
    # one fixed value, assembled by hand, so the layout cannot drift together
    # with the restatement: u64 id, f64 lnL, f64 alpha
    at = len(orc.put_header(OPTIONS))
    assert expect[at:at + 24] == struct.pack("<Qdd", 7, -123456.789, 0.25)
    assert struct.unpack("<Q", expect[:8])[0] == len("data/10.fasta")


def test_round_trip_and_reopen(tmp_path):
    c = rd.Checkpoint(str(tmp_path / "run"))
    c.save_options(OPTIONS)
    records = [(i * 3, -1000.0 - i, i / 10, params(i, 1 + i % 2, i % 3 == 0)) for i in range(9)]
    for r in records:
        c.write(*r)
    got = c.read_results()
    assert [(r, l, a) for r, l, a, _ in got] == [(r, l, a) for r, l, a, _ in records]
    for (_, _, _, gp), (_, _, _, wp) in zip(got, records):
        assert gp == wp
    assert c.completed_indicies() == [r[0] for r in records]
    assert not c.needs_cleaning()
    c.close()
    d = rd.Checkpoint(str(tmp_path / "run"))     # a later run finds the file
    assert d.existing_checkpoint()
    back = d.load_options()
    for k, v in OPTIONS.items():
        assert back[k] == v, k
    d.save_options({"msa_filename": "other"})    # ignored for an existing file (checkpoint.cpp:212-217)
    assert d.load_options()["msa_filename"] == OPTIONS["msa_filename"]
    assert len(d.read_results()) == 9


@pytest.mark.parametrize("damage", ["truncate", "flip_result", "flip_params"])
def test_torn_tail_is_detected_and_cleaned(tmp_path, damage):
    c = rd.Checkpoint(str(tmp_path / "t"))
    c.save_options(OPTIONS)
    records = [(i, -50.0 * (i + 1), 0.5, params(i)) for i in range(4)]
    for r in records:
        c.write(*r)
    c.close()
    path = str(tmp_path / "t.ckp")
    raw = bytearray(open(path, "rb").read())
    last = len(raw) - len(orc.put_record(*records[-1]))
    if damage == "truncate":
        raw = raw[:-11]
    elif damage == "flip_result":
        raw[last + 9] ^= 0x40          # inside the lnL of the last record
    else:
        raw[-20] ^= 0x01               # inside its parameter block
    open(path, "wb").write(bytes(raw))
    d = rd.Checkpoint(str(tmp_path / "t"))
    assert d.needs_cleaning()
    assert d.completed_indicies() == [0, 1, 2]          # "resume with what we can"
    d.clean()
    assert not d.needs_cleaning()
    assert open(path, "rb").read() == orc.put_header(OPTIONS) + b"".join(
        orc.put_record(*r) for r in records[:3])
    d.write(*records[3])                                 # the handle follows the new file
    assert d.completed_indicies() == [0, 1, 2, 3]


def _appender(prefix, rank, count):
    c = rd.Checkpoint(prefix)
    for i in range(count):
        c.write(rank * 1000 + i, -float(rank * 1000 + i), 0.5, params(rank * 1000 + i))
    c.close()


def test_concurrent_appends_from_several_processes(tmp_path):
    """one log shared by the processes of a multi-GPU run, as the reference's MPI
    ranks share theirs: every record must come through whole."""
    prefix = str(tmp_path / "shared")
    c = rd.Checkpoint(prefix)
    c.save_options(OPTIONS)
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_appender, args=(prefix, r, 40)) for r in range(4)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert not c.needs_cleaning()
    got = c.read_results()
    assert sorted(r for r, _, _, _ in got) == sorted(r * 1000 + i for r in range(4) for i in range(40))
    for rid, llh, alpha, pp in got:
        assert llh == -float(rid) and pp == params(rid)


def test_header_is_required(tmp_path):
    path = tmp_path / "bad.ckp"
    path.write_bytes(b"\x00" * 10)
    c = rd.Checkpoint(str(tmp_path / "bad"))
    with pytest.raises(rd.RdamdError):
        c.read_results()


def test_struct_layouts_match_the_reference_header():
    """tests/golden/ref_layout.json is what the reference's own util.hpp says
    about the types it writes raw into the file (printed by oracle/ref_layout.cpp,
    compiled against /root/reference/src by `make -C oracle ref`).  The writer,
    the C ABI mirror and the Python restatement must agree with it."""
    import ctypes
    import json
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    gold = json.load(open(os.path.join(here, "golden", "ref_layout.json")))
    exe = os.path.join(here, "..", "oracle", "_ref", "ref_layout")
    if os.path.exists(exe):                       # the fixture is still what the header says
        assert json.loads(subprocess.run([exe], capture_output=True, text=True, check=True).stdout) == gold
    assert gold["sizeof_rd_result_t"] == struct.calcsize("<Qdd") == 24
    assert gold["offsetof_rd_result_t"] == [0, 8, 16]
    assert gold["sizeof_ratehet_opts_t"] == struct.calcsize("<iiQB7xd") == 32
    assert gold["offsetof_ratehet_opts_t"] == [0, 4, 8, 16, 24]
    assert gold["sizeof_enums"] == [4, 4, 4, 4]
    assert gold["sizeof_scalars"] == {"seed": 8, "min_roots": 8, "threads": 8, "bool": 1, "field_flags_t": 4}
    assert gold["enum_rate_category"] == {"MEDIAN": 0, "MEAN": 1, "FREE": 2}
    assert gold["enum_param_type"]["estimate"] == 1 and gold["enum_initial_root_strategy"]["modified_mad"] == 2
    # the ABI mirror has the reference's field order (its own padding is irrelevant: the
    # writer assembles the 32-byte image field by field)
    assert [f for f, _ in rd.api.RatehetOpts._fields_] == ["type", "rate_category_type", "rate_cats",
                                                          "alpha_init", "alpha"]
    # a header written with defaults reads back as the reference's defaults
    d = gold["cli_defaults"]
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        c = rd.Checkpoint(os.path.join(tmp, "d"))
        c.save_options({})
        c.close()
        back = rd.Checkpoint(os.path.join(tmp, "d")).load_options()
    for k in ("min_roots", "threads", "root_ratio", "abs_tolerance", "factor", "br_tolerance",
              "bfgs_tol", "initial_root_strategy"):
        assert back[k] == d[k], k
    assert len(back["rate_cats"]) == d["n_rate_cats"] and back["rate_cats"][0]["rate_cats"] == d["rate_cats0"]
    r = gold["ratehet_from_size_t"]
    assert (back["rate_cats"][0]["type"], back["rate_cats"][0]["rate_category_type"]) == (r["type"], r["rate_category_type"])
    assert back["early_stop"] == 0 and d["early_stop_initialized"] == 0

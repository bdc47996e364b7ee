"""TEST INFRASTRUCTURE -- not part of the product.

Pure-Python restatement of the reference's checkpoint file format
(/root/reference/src/checkpoint.hpp, src/checkpoint.cpp), used by
tests/test_checkpoint.py as the independent checker of
root_digger_amd/csrc/checkpoint.cpp.

PARITY UNPINNED: the reference's checkpoint code cannot be compiled here (it
includes tree.hpp -> coraxlib, absent) and its tests (test/src/checkpoint.cpp)
hold no golden file, only round trips.  What this file follows, line by line:

* write<T>(fd, val): the raw bytes of val              checkpoint.hpp:93-99
* write(fd, vector<T>): u64 size, then the elements    checkpoint.hpp:101-106
* write(fd, string): u64 size, then the characters     checkpoint.cpp:11-25
* write_with_success: value, then u32 flag 1           checkpoint.hpp:108-113
* write_with_checksum: value, then u32 checksum        checkpoint.hpp:115-133
* cli_options_t field order                            checkpoint.cpp:60-91
* partition_parameters_t field order                   checkpoint.cpp:126-133
* the Adler-32 variant and its overload quirk          checkpoint.hpp:33-91,
                                                       checkpoint.cpp:144-150
"""
import struct

MOD_ADLER = 65521
M32 = 0xFFFFFFFF


def fold(data, a=1, b=0):
    """compute_checksum_components(const T&, a, b), checkpoint.hpp:33-45.  The
    reference writes `b = b + a % MOD_ADLER`: b is never reduced and wraps."""
    for byte in data:
        a = (a + byte) % MOD_ADLER
        b = (b + a % MOD_ADLER) & M32
    return a, b


def checksum_result(root_id, llh, alpha):
    """compute_checksum(rd_result_t): one fold over the 24 raw bytes."""
    a, b = fold(struct.pack("<Qdd", root_id, llh, alpha))
    return ((b << 16) | a) & M32


def checksum_params(params):
    """compute_checksum(vector<partition_parameters_t>), checkpoint.hpp:66-77 with
    the specialisation at checkpoint.cpp:144-150.  The variadic helper
    (checkpoint.hpp:79-85) ends its recursion by calling the two-argument form
    compute_checksum_components(p.first, p.second), which resolves to the
    GENERIC overload: value = the 4 bytes of the running `a`, a = the running
    `b`, b = 0."""
    a, b = 1, 0
    for pp in params:
        for field in ("subst_rates", "freqs", "gamma_alpha", "gamma_weights"):
            for v in pp.get(field, ()):
                a, b = fold(struct.pack("<d", v), a, b)
        a, b = fold(struct.pack("<I", a), b, 0)
    return ((b << 16) | a) & M32


def put_string(s):
    raw = s.encode()
    return struct.pack("<Q", len(raw)) + raw


def put_doubles(v):
    return struct.pack("<Q", len(v)) + b"".join(struct.pack("<d", x) for x in v)


def put_ratehet(rc):
    """ratehet_opts_t as its raw struct (util.hpp:50-70): i32 type, i32
    rate_category_type, u64 rate_cats, bool alpha_init + 7 padding, f64 alpha."""
    return struct.pack("<iiQB7xd", rc.get("type", 1), rc.get("rate_category_type", 1),
                       rc.get("rate_cats", 1), int(rc.get("alpha_init", False)),
                       rc.get("alpha", 1.0))


def put_header(o):
    out = b"".join(put_string(o.get(k, "")) for k in (
        "msa_filename", "tree_filename", "prefix", "prefix_dir", "model_filename",
        "freqs_filename", "partition_filename", "data_type", "model_string"))
    cats = [{"rate_cats": c} if isinstance(c, int) else c for c in o.get("rate_cats", [1])]
    out += struct.pack("<Q", len(cats)) + b"".join(put_ratehet(c) for c in cats)
    out += struct.pack("<QQQ", o.get("seed", 0), o.get("min_roots", 1), o.get("threads", 0))
    out += struct.pack("<ddddd", o.get("root_ratio", 0.01), o.get("abs_tolerance", 1e-7),
                       o.get("factor", 1e4), o.get("br_tolerance", 1e-12), o.get("bfgs_tol", 1e-7))
    out += struct.pack("<BBBB", int(o.get("silent", 0)), int(o.get("exhaustive", 0)),
                       int(o.get("echo", 0)), int(o.get("invariant_sites", 0)))
    out += struct.pack("<ii", o.get("early_stop", 0), o.get("initial_root_strategy", 2))
    return out + struct.pack("<I", 1)   # CHECKPOINT_WRITE_SUCCESS_FLAG


def put_record(root_id, llh, alpha, params):
    out = struct.pack("<Qdd", root_id, llh, alpha)
    out += struct.pack("<I", checksum_result(root_id, llh, alpha))
    out += struct.pack("<Q", len(params))
    for pp in params:
        for field in ("subst_rates", "freqs", "gamma_alpha", "gamma_weights"):
            out += put_doubles(pp.get(field, ()))
    return out + struct.pack("<I", checksum_params(params))

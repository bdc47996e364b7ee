"""ctypes mirror of include/root_digger_amd.h (names follow the C ABI)."""
import ctypes as C
import os

import numpy as np

from ._lib import lib, RdamdError

GAMMA_RATES_MEAN = 0
GAMMA_RATES_MEDIAN = 1
ATTRIB_SITE_REPEATS = 1 << 10   # RDAMD_ATTRIB_SITE_REPEATS (CORAX_ATTRIB_SITE_REPEATS, src/model.cpp:145-149)
ATTRIB_NONREV = 1 << 11
ATTRIB_SPARSE_CLVS = 1 << 20    # RDAMD_ATTRIB_SPARSE_CLVS: CLV / scale buffers get memory when first named
SCALE_BUFFER_NONE = -1


class Operation(C.Structure):
    """rdamd_operation_t == corax_operation_t (src/tree.cpp:399-410)."""
    _fields_ = [
        ("parent_clv_index", C.c_uint),
        ("parent_scaler_index", C.c_int),
        ("child1_clv_index", C.c_uint),
        ("child1_matrix_index", C.c_uint),
        ("child1_scaler_index", C.c_int),
        ("child2_clv_index", C.c_uint),
        ("child2_matrix_index", C.c_uint),
        ("child2_scaler_index", C.c_int),
    ]

    def astuple(self):
        return tuple(getattr(self, f) for f, _ in self._fields_)


class RootLocation(C.Structure):
    """rdamd_root_location_t == root_location_t (src/tree.hpp:24-52)."""
    _fields_ = [("edge", C.c_int), ("id", C.c_uint64), ("saved_brlen", C.c_double),
                ("brlen_ratio", C.c_double)]

    def with_ratio(self, ratio):
        r = RootLocation(self.edge, self.id, self.saved_brlen, ratio)
        return r

    def brlen(self):
        return self.saved_brlen * self.brlen_ratio

    def brlen_compliment(self):
        return self.saved_brlen * (1 - self.brlen_ratio)


def _sig(name, restype, *argtypes):
    f = getattr(lib, name)
    f.restype = restype
    f.argtypes = list(argtypes)
    return f


_u = C.c_uint
_pu = C.POINTER(C.c_uint)
_pd = C.POINTER(C.c_double)
_vp = C.c_void_p
_pop = C.POINTER(Operation)
_prl = C.POINTER(RootLocation)

_errno = _sig("rdamd_errno", C.c_int)
_errmsg = _sig("rdamd_errmsg", C.c_char_p)
_sig("rdamd_version", C.c_char_p)
_sig("rdamd_hip_runtime_path", C.c_char_p)
_sig("rdamd_device_count", C.c_int)
_sig("rdamd_device_memory", C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64))
_sig("rdamd_set_device", C.c_int, C.c_int)

_sig("rdamd_partition_create", _vp, _u, _u, _u, _u, _u, _u, _u, _u, _u)
_sig("rdamd_partition_destroy", None, _vp)
_sig("rdamd_set_tip_states", C.c_int, _vp, _u, C.POINTER(C.c_uint64), C.c_char_p)
_sig("rdamd_set_pattern_weights", None, _vp, _pu)
_sig("rdamd_set_subst_params", None, _vp, _u, _pd)
_sig("rdamd_set_frequencies", None, _vp, _u, _pd)
_sig("rdamd_set_category_rates", None, _vp, _pd)
_sig("rdamd_set_category_weights", None, _vp, _pd)
_sig("rdamd_update_invariant_sites_proportion", C.c_int, _vp, _u, C.c_double)
_sig("rdamd_msa_empirical_frequencies", C.POINTER(C.c_double), _vp)
_sig("rdamd_compute_gamma_cats", C.c_int, C.c_double, _u, _pd, C.c_int)
_sig("rdamd_partition_states", _u, _vp)
_sig("rdamd_partition_rate_cats", _u, _vp)
_sig("rdamd_partition_sites", _u, _vp)
_sig("rdamd_partition_tips", _u, _vp)
_sig("rdamd_partition_subst_params", C.POINTER(C.c_double), _vp, _u)
_sig("rdamd_partition_frequencies", C.POINTER(C.c_double), _vp, _u)
_sig("rdamd_update_prob_matrices", C.c_int, _vp, _pu, _pu, _pd, _u)
_sig("rdamd_update_clvs", None, _vp, _pop, _u)
_sig("rdamd_compute_root_loglikelihood", C.c_double, _vp, _u, C.c_int, _pu, _pd)
_sig("rdamd_root_loglikelihood_fused", C.c_int, _vp, _pop, _pu, _pd, _pd, _u, _pd)
_sig("rdamd_root_loglikelihood_fused_multi", C.c_int, _u, C.POINTER(_vp), _pop, C.POINTER(_pu), _pd, _pd, _pu, _pd)
_sig("rdamd_schedule_create", _vp, _vp, _pop, _u, _pu, _pd, _u)
_sig("rdamd_schedule_destroy", None, _vp)
_sig("rdamd_schedule_stack_depth", _u, _vp)


class ScheduleStats(C.Structure):
    """rdamd_schedule_stats_t"""
    _fields_ = [(k, C.c_uint) for k in ("operations", "steps", "matvecs", "matvecs_plain", "pseudo_tips",
                                        "clade_nodes", "clade_rows", "stack_depth", "stack_depth_plain",
                                        "parks", "parks_in_registers", "parks_in_lds_slot")]


_sig("rdamd_schedule_stats", C.c_int, _vp, C.POINTER(ScheduleStats))
_sig("rdamd_partition_set_site_repeats", C.c_int, _vp, _u)
_sig("rdamd_partition_site_repeats", _u, _vp)
_sig("rdamd_evaluate_batch", C.c_int, _vp, _u, C.POINTER(_vp), _pd, _pd, _pd, _pd, _pd)
_sig("rdamd_evaluate_batch_device", C.c_int, _vp, _u, C.POINTER(_vp), _pd, _pd, _pd, _pd, _vp)
_sig("rdamd_evaluate_batch_submit", C.c_int, _vp, _u, _u, C.POINTER(_vp), _pd, _pd, _pd, _pd)
_sig("rdamd_evaluate_batch_wait", C.c_int, _vp, _u, _pd)
_sig("rdamd_evaluate_batch_submit_device", C.c_int, _vp, _u, _u, C.POINTER(_vp), _pd, _pd, _pd, _pd, _vp)
_sig("rdamd_evaluate_batch_redo_device", C.c_int, _vp, _u, _vp)
_sig("rdamd_evaluate_batch_finish_device", C.c_int, _vp, _u)
_sig("rdamd_partition_discard_clvs", None, _vp)
_sig("rdamd_partition_clv_bytes", C.c_uint64, _vp)
_sig("rdamd_update_clvs_launches", C.c_uint, _vp)
_sig("rdamd_partition_set_rescale_speculation", C.c_int, _vp, C.c_int)
_sig("rdamd_evaluate_second_passes", C.c_ulonglong, _vp)
_sig("rdamd_evaluate_root_children", C.c_int, _vp, _pop, _u, _pu, _pd, _u, _pd, _pd, _pd, _pd, _pd)
_sig("rdamd_get_clv", C.c_int, _vp, _u, _pd)
_sig("rdamd_get_scaler", C.c_int, _vp, _u, _pu)
_sig("rdamd_get_pmatrix", C.c_int, _vp, _u, _pd)
_sig("rdamd_partition_sync", None, _vp)
_sig("rdamd_profile_enable", None, _vp, C.c_int)
_sig("rdamd_profile_read", C.c_int, _vp, _pd, _pu)

_sig("rdamd_tree_from_file", _vp, C.c_char_p)
_sig("rdamd_tree_from_newick", _vp, C.c_char_p)
_sig("rdamd_tree_destroy", None, _vp)
for _n in ("tip_count", "inner_count", "branch_count", "root_count", "root_clv_index"):
    _sig("rdamd_tree_" + _n, _u, _vp)
_sig("rdamd_tree_root_scaler_index", C.c_int, _vp)
_sig("rdamd_tree_root_location", C.c_int, _vp, _u, _prl)
_sig("rdamd_tree_rank_midpoints", C.c_int, _vp, _pu)
_sig("rdamd_tree_rank_modified_mad", C.c_int, _vp, _pu)
_sig("rdamd_tree_root_location_by_label", C.c_int, _vp, C.c_char_p, _prl)
_sig("rdamd_tree_root_label", C.c_char_p, _vp, _u)
_sig("rdamd_tree_root_is_internal", C.c_int, _vp, _u)
_sig("rdamd_tree_tip_index", C.c_int, _vp, C.c_char_p)
_sig("rdamd_tree_tip_label", C.c_char_p, _vp, _u)
_sig("rdamd_tree_side_tips", _vp, _vp, _prl)
_sig("rdamd_tree_generate_operations", C.c_int, _vp, _prl, _pop, _pu, _pu, _pd, _pu)
_sig("rdamd_tree_generate_derivative_operations", C.c_int, _vp, _prl, _pop, _pu, _pd)
_sig("rdamd_tree_generate_root_update_operations", C.c_int, _vp, _prl, _pop, _pu, _pu, _pd, _pu)
_sig("rdamd_tree_root_by", C.c_int, _vp, _prl)
_sig("rdamd_tree_unroot", None, _vp)
_sig("rdamd_tree_rooted", C.c_int, _vp)
_sig("rdamd_tree_sanity_check", C.c_int, _vp)
_sig("rdamd_tree_newick", _vp, _vp, C.c_int)
_sig("rdamd_tree_annotate_branch", C.c_int, _vp, _prl, C.c_char_p, C.c_char_p)
_sig("rdamd_tree_annotate_branch_lr", C.c_int, _vp, _prl, C.c_char_p, C.c_char_p, C.c_char_p)

_sig("rdamd_model_create", _vp, _vp, _u, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), _pu, _u,
     C.POINTER(C.c_uint64), _u, C.c_uint64, C.c_int)
_sig("rdamd_model_create_from_file", _vp, _vp, C.c_char_p, _u, C.POINTER(C.c_uint64), _u,
     C.c_uint64, C.c_int, C.c_int, _pu)
_sig("rdamd_msa_probe", C.c_int, C.c_char_p, C.POINTER(C.c_uint64), C.c_int, _pu, _pu, _pu)
_sig("rdamd_model_destroy", None, _vp)
_sig("rdamd_model_initialize_partitions", C.c_int, _vp, C.c_int)
_sig("rdamd_model_set_subst_rates", C.c_int, _vp, _pd)
_sig("rdamd_model_set_subst_rates_uniform", C.c_int, _vp)
_sig("rdamd_model_set_freqs", C.c_int, _vp, _pd)
_sig("rdamd_model_set_empirical_freqs", C.c_int, _vp)
_sig("rdamd_model_set_gamma_alpha", C.c_int, _vp, C.c_double)
_sig("rdamd_model_compute_lh", C.c_double, _vp, _prl)
_sig("rdamd_model_compute_lh_root", C.c_double, _vp, _prl)
_sig("rdamd_model_compute_dlh", C.c_int, _vp, _prl, _pd)
_sig("rdamd_model_move_root", C.c_int, _vp, _prl)
_sig("rdamd_model_compute_all_root_lh", C.c_int, _vp, _pd)
_sig("rdamd_model_optimize_alpha", C.c_int, _vp, _prl, C.c_double, _prl)
_sig("rdamd_model_compute_all_root_lh_batched", C.c_int, _vp, _pd)
_sig("rdamd_model_search", C.c_int, _vp, _u, C.c_double, C.c_double, C.c_double, C.c_double,
     C.c_double, _prl, _pd)
_sig("rdamd_model_compute_lh_batch", C.c_int, _vp, _u, _prl, _pd, _pd, _pd, _pd)
class RatehetOpts(C.Structure):
    """rdamd_ratehet_opts_t == ratehet_opts_t (src/util.hpp:50-70)."""
    _fields_ = [("type", C.c_int32), ("rate_category_type", C.c_int32), ("rate_cats", C.c_uint64),
                ("alpha_init", C.c_int32), ("alpha", C.c_double)]


class CliOptions(C.Structure):
    """rdamd_cli_options_t: the fields of cli_options_t the checkpoint header
    holds (src/checkpoint.cpp:60-91)."""
    _fields_ = ([(n, C.c_char_p) for n in (
        "msa_filename", "tree_filename", "prefix", "prefix_dir", "model_filename",
        "freqs_filename", "partition_filename", "data_type", "model_string")] + [
        ("rate_cats", C.POINTER(RatehetOpts)), ("n_rate_cats", C.c_uint64),
        ("seed", C.c_uint64), ("min_roots", C.c_uint64), ("threads", C.c_uint64),
        ("root_ratio", C.c_double), ("abs_tolerance", C.c_double), ("factor", C.c_double),
        ("br_tolerance", C.c_double), ("bfgs_tol", C.c_double),
        ("silent", C.c_int32), ("exhaustive", C.c_int32), ("echo", C.c_int32),
        ("invariant_sites", C.c_int32), ("early_stop", C.c_int32),
        ("initial_root_strategy", C.c_int32)])


class PartitionInfo(C.Structure):
    """rdamd_partition_info_t: partition_info_t + model_info_t (src/util.hpp:86-100)."""
    _fields_ = [("model_name", C.c_char * 256), ("partition_name", C.c_char * 128),
                ("subst_str", C.c_char * 64), ("n_ranges", C.c_uint),
                ("ranges", (C.c_uint64 * 2) * 64), ("freq_type", C.c_int32),
                ("invar_present", C.c_int32), ("invar_type", C.c_int32),
                ("invar_user_prop", C.c_float), ("ratehet", RatehetOpts),
                ("asc_present", C.c_int32), ("asc_type", C.c_int32),
                ("asc_fels_weight", C.c_double), ("n_stam_weights", C.c_uint),
                ("stam_weights", C.c_double * 32)]


_pu64 = C.POINTER(C.c_uint64)
_sig("rdamd_parse_model_info", C.c_int, C.c_char_p, C.POINTER(PartitionInfo))
_sig("rdamd_parse_partition_info", C.c_int, C.c_char_p, C.POINTER(PartitionInfo))
_sig("rdamd_msa_partition_probe", C.c_int, C.c_char_p, C.c_void_p, _u, C.POINTER(C.c_char_p),
     C.c_int, _pu, _pu)
_sig("rdamd_model_create_partitioned", _vp, _vp, C.c_char_p, C.c_char_p, _u, C.c_void_p,
     C.c_uint64, C.c_int, _pu)
_sig("rdamd_model_create_from_file_ratehet", _vp, _vp, C.c_char_p, _u, C.c_void_p,
     C.POINTER(RatehetOpts), C.c_uint64, C.c_int, C.c_int, _pu)
_sig("rdamd_model_partition_count", C.c_int, _vp)
_sig("rdamd_checkpoint_open", _vp, C.c_char_p)
_sig("rdamd_checkpoint_close", None, _vp)
_sig("rdamd_checkpoint_existing", C.c_int, _vp)
_sig("rdamd_checkpoint_filename", C.c_char_p, _vp)
_sig("rdamd_checkpoint_save_options", C.c_int, _vp, C.POINTER(CliOptions))
_sig("rdamd_checkpoint_load_options", C.c_int, _vp, C.POINTER(CliOptions))
_sig("rdamd_checkpoint_write", C.c_int, _vp, C.c_uint64, C.c_double, C.c_double, _u, _pu64, _pd)
_sig("rdamd_checkpoint_read_results", C.c_int, _vp, _pu)
_sig("rdamd_checkpoint_result", C.c_int, _vp, _u, _pu64, _pd, _pd, _pu, _pu64)
_sig("rdamd_checkpoint_result_params", C.c_int, _vp, _u, _pu64, _pd)
_sig("rdamd_checkpoint_needs_cleaning", C.c_int, _vp)
_sig("rdamd_checkpoint_clean", C.c_int, _vp)
_sig("rdamd_checkpoint_checksum_result", C.c_uint32, C.c_uint64, C.c_double, C.c_double)
_sig("rdamd_checkpoint_checksum_params", C.c_uint32, _u, _pu64, _pd)
_sig("rdamd_model_assign_by_rank_search", C.c_int, _vp, _u, C.c_double, _u, _u, C.c_int, _vp)
_sig("rdamd_model_assigned", C.c_int, _vp, _pu64, _u)
_sig("rdamd_model_set_progress", C.c_int, _vp, C.c_int)
_sig("rdamd_model_set_checkpoint", C.c_int, _vp, _vp)
_sig("rdamd_model_assign_by_rank_checkpoint", C.c_int, _vp, _u, _u, _vp)
_sig("rdamd_model_compute_all_root_lh_directional", C.c_int, _vp, _pd, _pd)
_sig("rdamd_compute_root_loglikelihoods", C.c_int, _vp, _u, _pu, C.POINTER(C.c_int), _pu, _pd)
_sig("rdamd_tree_generate_directional_operations", C.c_int, _vp, _pd, _pop, _pu, _pu, _pd, _pu, _pu,
     C.POINTER(C.c_int), _pu)
_sig("rdamd_model_counters", None, C.c_void_p, C.POINTER(C.c_uint64))
_sig("rdamd_model_lockstep_stats", None, C.c_void_p, C.POINTER(C.c_uint64))
_sig("rdamd_model_set_lockstep_groups", None, C.c_void_p, _u)
_sig("rdamd_model_set_lockstep_rounds", None, C.c_void_p, C.c_int)
_sig("rdamd_model_round_stats", None, C.c_void_p, C.POINTER(C.c_uint64))
_sig("rdamd_model_round_seconds", None, C.c_void_p, _pd)
_sig("rdamd_model_set_lockstep_priority", None, C.c_void_p, C.c_int)
_sig("rdamd_model_set_root_children_only", None, C.c_void_p, C.c_int)
_sig("rdamd_partition_set_stream_priority", C.c_int, _vp, C.c_int)
_sig("rdamd_model_assign_by_rank", C.c_int, _vp, _u, _u)
_sig("rdamd_model_exhaustive_search_parallel", C.c_int, _vp, _u, C.c_double, C.c_double,
     C.c_double, C.c_double, C.POINTER(C.c_uint64), _pd, _pd, _pu, _prl, _pd)
_sig("rdamd_model_exhaustive_search_lockstep", C.c_int, _vp, _u, C.c_double, C.c_double,
     C.c_double, C.c_double, C.POINTER(C.c_uint64), _pd, _pd, _pu, _prl, _pd)
_sig("rdamd_model_set_lbfgsb", None, _vp, _vp)
_sig("rdamd_model_optimize_params", C.c_int, _vp, _prl, C.c_double, C.c_double, C.c_int, _pd, _pd,
     _pd, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64))
_sig("rdamd_model_exhaustive_search", C.c_int, _vp, C.c_double, C.c_double, C.c_double,
     C.c_double, C.POINTER(C.c_uint64), _pd, _pd, _pu, _prl, _pd)

# site-sharded runs: reduction hook + RCCL communicator (include/root_digger_amd.h)
LNL_REDUCER = C.CFUNCTYPE(C.c_int, _pd, _u, _vp, _vp)
_sig("rdamd_model_set_lnl_reducer", C.c_int, _vp, _vp, _vp, C.c_int)
_sig("rdamd_model_create_from_file_block", _vp, _vp, C.c_char_p, _u, C.c_void_p,
     C.POINTER(RatehetOpts), C.c_uint64, C.c_int, C.c_int, _u, _u, _pu, _pu)
_sig("rdamd_partition_weight_sum", C.c_double, _vp)
_sig("rdamd_partition_stream", _vp, _vp)
_sig("rdamd_comm_unique_id", C.c_int, C.c_char * 128)
_sig("rdamd_comm_create", _vp, C.c_char * 128, C.c_int, C.c_int)
_sig("rdamd_comm_allreduce_sum", C.c_int, _vp, _vp, _u, _vp)
_sig("rdamd_comm_destroy", None, _vp)
_sig("rdamd_comm_set_sum_mode", C.c_int, _vp, C.c_int)
_sig("rdamd_comm_sum_mode", C.c_int, _vp)
_sig("rdamd_rank_order_sum", C.c_int, _vp, _vp, _u, _u, _vp)
COMM_SUM_GATHER, COMM_SUM_ALLREDUCE = 0, 1

_sig("rdamd_partition_footprint", C.c_uint64, _u, _u, _u, _u, _u, _u, _u)
_sig("rdamd_model_max_replicas", _u, _vp, _u, C.POINTER(C.c_uint64))

_libc = C.CDLL(None)
_libc.free.argtypes = [_vp]
_libc.free.restype = None

MAP_NT = (C.c_uint64 * 256).in_dll(lib, "rdamd_map_nt")
MAP_BIN = (C.c_uint64 * 256).in_dll(lib, "rdamd_map_bin")


def _fail(where):
    raise RdamdError("%s: %s" % (where, (_errmsg() or b"").decode()))


def _take_string(ptr):
    if not ptr:
        _fail("string result")
    s = C.string_at(ptr).decode()
    _libc.free(ptr)
    return s


def hip_runtime_path():
    """libamdhip64 behind librdamd (rdamd_hip_runtime_path)."""
    return os.path.realpath(lib.rdamd_hip_runtime_path().decode())


def mapped_hip_runtimes():
    """every libamdhip64 mapped into this process (realpaths): more than one means two HIP
    runtime instances, between which device pointers must not travel"""
    found = set()
    for line in open("/proc/self/maps"):
        path = line.split()[-1] if "/" in line else ""
        if os.path.basename(path).startswith("libamdhip64.so"):
            found.add(os.path.realpath(path))
    return sorted(found)


def device_count():
    return lib.rdamd_device_count()


def device_memory():
    """(free, total) bytes of the current HIP device."""
    f, t = C.c_uint64(0), C.c_uint64(0)
    if lib.rdamd_device_memory(C.byref(f), C.byref(t)) != 1:
        _fail("device_memory")
    return int(f.value), int(t.value)


def set_device(device):
    if lib.rdamd_set_device(int(device)) != 1:
        _fail("set_device")


def msa_probe(path, cmap=None, compress=True):
    """(taxa, patterns, total weight) of an alignment file (msa_t(filename))."""
    a, b, c = C.c_uint(0), C.c_uint(0), C.c_uint(0)
    if lib.rdamd_msa_probe(str(path).encode(), cmap if cmap is not None else MAP_NT,
                           1 if compress else 0, C.byref(a), C.byref(b), C.byref(c)) != 1:
        _fail("msa_probe")
    return a.value, b.value, c.value


def compute_gamma_cats(alpha, cats, mode=GAMMA_RATES_MEAN):
    out = (C.c_double * cats)()
    if lib.rdamd_compute_gamma_cats(alpha, cats, out, mode) != 1:
        _fail("compute_gamma_cats")
    return list(out)


def _dptr(a):
    return a.ctypes.data_as(_pd)


def root_loglikelihood_fused_multi(parts, ops, lengths1, lengths2):
    """rdamd_root_loglikelihood_fused_multi: item i = partition parts[i], root operation ops[i],
    up to 8 positions with branch lengths lengths1[i] / lengths2[i]; -> list of lnL arrays."""
    n = len(parts)
    hs = (_vp * n)(*[p._h for p in parts])
    cops = (Operation * n)(*ops)
    pidx = (_pu * n)(*[_uptr(p.params_indices) for p in parts])
    l1 = np.zeros((n, 8), dtype=np.float64)
    l2 = np.zeros((n, 8), dtype=np.float64)
    npos = np.zeros(n, dtype=np.uint32)
    for i in range(n):
        npos[i] = len(lengths1[i])
        l1[i, :npos[i]] = lengths1[i]
        l2[i, :npos[i]] = lengths2[i]
    out = np.zeros((n, 8), dtype=np.float64)
    if lib.rdamd_root_loglikelihood_fused_multi(n, hs, cops, pidx, _dptr(l1), _dptr(l2), _uptr(npos),
                                                _dptr(out)) != 1:
        _fail("root_loglikelihood_fused_multi")
    return [out[i, :npos[i]].copy() for i in range(n)]


def _uptr(a):
    return a.ctypes.data_as(_pu)


class Tree:
    """rooted_tree_t (src/tree.hpp:54) through the C ABI."""

    def __init__(self, handle):
        if not handle:
            _fail("Tree")
        self._h = handle

    @classmethod
    def from_newick(cls, text):
        return cls(lib.rdamd_tree_from_newick(text.encode()))

    @classmethod
    def from_file(cls, path):
        return cls(lib.rdamd_tree_from_file(str(path).encode()))

    def __del__(self):
        if getattr(self, "_h", None):
            lib.rdamd_tree_destroy(self._h)
            self._h = None

    def tip_count(self):
        return lib.rdamd_tree_tip_count(self._h)

    def inner_count(self):
        return lib.rdamd_tree_inner_count(self._h)

    def branch_count(self):
        return lib.rdamd_tree_branch_count(self._h)

    def root_count(self):
        return lib.rdamd_tree_root_count(self._h)

    def root_clv_index(self):
        return lib.rdamd_tree_root_clv_index(self._h)

    def root_scaler_index(self):
        return lib.rdamd_tree_root_scaler_index(self._h)

    def root_location(self, key):
        rl = RootLocation()
        if isinstance(key, str):
            ok = lib.rdamd_tree_root_location_by_label(self._h, key.encode(), C.byref(rl))
        else:
            ok = lib.rdamd_tree_root_location(self._h, int(key), C.byref(rl))
        if ok != 1:
            _fail("root_location")
        return rl

    def generate_directional_operations(self, ratios=None):
        """All-directions schedule (csrc/tree.hpp): dict with ops, matrix_indices,
        branch_lengths, root_clv, root_scaler and the partition sizes it needs."""
        n, roots = self.tip_count(), self.root_count()
        cap_ops, cap_mat = 3 * n + roots + 8, 3 * roots + 8
        ops = (Operation * cap_ops)()
        pmi = np.zeros(cap_mat, dtype=np.uint32)
        brl = np.zeros(cap_mat, dtype=np.float64)
        rclv = np.zeros(roots, dtype=np.uint32)
        rsc = np.zeros(roots, dtype=np.int32)
        sizes = np.zeros(3, dtype=np.uint32)
        nops, nmat = C.c_uint(0), C.c_uint(0)
        r = None
        if ratios is not None:
            r = np.ascontiguousarray(ratios, dtype=np.float64)
        if lib.rdamd_tree_generate_directional_operations(
                self._h, _dptr(r) if r is not None else None, ops, C.byref(nops), _uptr(pmi),
                _dptr(brl), C.byref(nmat), _uptr(rclv), rsc.ctypes.data_as(C.POINTER(C.c_int)),
                _uptr(sizes)) != 1:
            _fail("generate_directional_operations")
        out = (Operation * nops.value)()
        for i in range(nops.value):
            out[i] = ops[i]
        return {"ops": out, "matrix_indices": pmi[:nmat.value].copy(),
                "branch_lengths": brl[:nmat.value].copy(), "root_clv": rclv, "root_scaler": rsc,
                "clv_buffers": int(sizes[0]), "scale_buffers": int(sizes[1]),
                "prob_matrices": int(sizes[2])}

    def _ranked(self, fn):
        ids = np.zeros(self.root_count(), dtype=np.uint32)
        if fn(self._h, _uptr(ids)) != 1:
            _fail("rank")
        return [int(i) for i in ids]

    def rank_midpoints(self):
        """root ids, best midpoint balance first (src/tree.cpp:863-901)."""
        return self._ranked(lib.rdamd_tree_rank_midpoints)

    def rank_modified_mad(self):
        """root ids ranked by the modified MAD score (src/tree.cpp:907-945)."""
        return self._ranked(lib.rdamd_tree_rank_modified_mad)

    def midpoint(self):
        return self.root_location(self.rank_midpoints()[0])

    def roots(self):
        return [self.root_location(i) for i in range(self.root_count())]

    def root_label(self, index):
        return lib.rdamd_tree_root_label(self._h, index).decode()

    def root_is_internal(self, index):
        return bool(lib.rdamd_tree_root_is_internal(self._h, index))

    def tip_index(self, label):
        return lib.rdamd_tree_tip_index(self._h, label.encode())

    def tip_label(self, clv_index):
        return lib.rdamd_tree_tip_label(self._h, clv_index).decode()

    def label_map(self):
        return {self.tip_label(i): i for i in range(self.tip_count())}

    def side_tips(self, rl):
        return _take_string(lib.rdamd_tree_side_tips(self._h, C.byref(rl))).split("\n")

    def _sched(self, fn, rl):
        n = self.tip_count()
        ops = (Operation * (2 * n))()
        pmi = np.zeros(2 * n, dtype=np.uint32)
        brl = np.zeros(2 * n, dtype=np.float64)
        nops, nmat = C.c_uint(0), C.c_uint(0)
        if fn(self._h, C.byref(rl), ops, C.byref(nops), _uptr(pmi), _dptr(brl),
              C.byref(nmat)) != 1:
            _fail("schedule")
        out = (Operation * nops.value)()
        for i in range(nops.value):
            out[i] = ops[i]
        return out, pmi[:nmat.value].copy(), brl[:nmat.value].copy()

    def generate_operations(self, rl):
        return self._sched(lib.rdamd_tree_generate_operations, rl)

    def generate_root_update_operations(self, rl):
        return self._sched(lib.rdamd_tree_generate_root_update_operations, rl)

    def generate_derivative_operations(self, rl):
        op = Operation()
        pmi = np.zeros(2, dtype=np.uint32)
        brl = np.zeros(2, dtype=np.float64)
        if lib.rdamd_tree_generate_derivative_operations(
                self._h, C.byref(rl), C.byref(op), _uptr(pmi), _dptr(brl)) != 1:
            _fail("generate_derivative_operations")
        return op, pmi, brl

    def root_by(self, rl):
        if lib.rdamd_tree_root_by(self._h, C.byref(rl)) != 1:
            _fail("root_by")

    def unroot(self):
        lib.rdamd_tree_unroot(self._h)

    def rooted(self):
        return bool(lib.rdamd_tree_rooted(self._h))

    def sanity_check(self):
        return bool(lib.rdamd_tree_sanity_check(self._h))

    def newick(self, annotations=True):
        return _take_string(lib.rdamd_tree_newick(self._h, 1 if annotations else 0))

    def annotate_branch(self, rl, key, value, right_value=None):
        right = value if right_value is None else right_value
        if lib.rdamd_tree_annotate_branch_lr(self._h, C.byref(rl), key.encode(),
                                             value.encode(), right.encode()) != 1:
            _fail("annotate_branch")


class Schedule:
    """A traversal compiled for the fused evaluator (rdamd_schedule_t)."""

    def __init__(self, part, ops, matrix_indices, branch_lengths):
        n = len(ops)
        if not isinstance(ops, C.Array):
            arr = (Operation * n)()
            for i, o in enumerate(ops):
                arr[i] = o
            ops = arr
        mi = np.ascontiguousarray(matrix_indices, dtype=np.uint32)
        bl = np.ascontiguousarray(branch_lengths, dtype=np.float64)
        self._part = part
        self._h = lib.rdamd_schedule_create(part.handle, ops, n, _uptr(mi), _dptr(bl), mi.size)
        if not self._h:
            _fail("schedule_create")

    def stack_depth(self):
        return lib.rdamd_schedule_stack_depth(self._h)

    def stats(self):
        """dict of rdamd_schedule_stats_t: what one (site, rate) executes per traversal."""
        st = ScheduleStats()
        if lib.rdamd_schedule_stats(self._h, C.byref(st)) != 1:
            _fail("schedule_stats")
        return {k: getattr(st, k) for k, _ in ScheduleStats._fields_}

    def destroy(self):
        if getattr(self, "_h", None) and getattr(self._part, "_h", None):
            lib.rdamd_schedule_destroy(self._h)
        self._h = None

    def __del__(self):
        self.destroy()


class Partition:
    """The coraxlib partition subset RootDigger uses, in HBM."""

    def __init__(self, tips, clv_buffers, states, sites, rate_matrices, prob_matrices,
                 rate_cats, scale_buffers, attributes=0):
        self._h = lib.rdamd_partition_create(tips, clv_buffers, states, sites,
                                             rate_matrices, prob_matrices, rate_cats,
                                             scale_buffers, attributes)
        if not self._h:
            _fail("partition_create")
        self.tips, self.clv_buffers, self.states, self.sites = tips, clv_buffers, states, sites
        self.rate_cats, self.prob_matrices, self.scale_buffers = rate_cats, prob_matrices, scale_buffers
        self.params_indices = np.zeros(rate_cats, dtype=np.uint32)

    @classmethod
    def for_tree(cls, tree, states, sites, rate_cats, attributes=0):
        """Sizes exactly as model_t's constructor picks them (src/model.cpp:159-168)."""
        b = tree.branch_count()
        return cls(tree.tip_count(), b, states, sites, 1, b, rate_cats, b, attributes)

    def destroy(self):
        if getattr(self, "_h", None):
            lib.rdamd_partition_destroy(self._h)
            self._h = None

    def __del__(self):
        self.destroy()

    @property
    def handle(self):
        return self._h

    def set_site_repeats(self, max_classes):
        """rdamd_partition_set_site_repeats: class limit of the pseudo-tips (0 = off)."""
        if lib.rdamd_partition_set_site_repeats(self._h, max_classes) != 1:
            _fail("partition_set_site_repeats")

    def site_repeats(self):
        """the pseudo-tips' class limit in force (0: no site repeats)."""
        return int(lib.rdamd_partition_site_repeats(self._h))

    def set_tip_states(self, tip_index, cmap, sequence):
        if isinstance(sequence, str):
            sequence = sequence.encode()
        if lib.rdamd_set_tip_states(self._h, tip_index, cmap, sequence) != 1:
            _fail("set_tip_states")

    def set_pattern_weights(self, w):
        w = np.ascontiguousarray(w, dtype=np.uint32)
        assert w.size == self.sites
        lib.rdamd_set_pattern_weights(self._h, _uptr(w))

    def set_subst_params(self, idx, params):
        a = np.ascontiguousarray(params, dtype=np.float64)
        assert a.size == self.states * self.states - self.states
        lib.rdamd_set_subst_params(self._h, idx, _dptr(a))

    def set_frequencies(self, idx, freqs):
        a = np.ascontiguousarray(freqs, dtype=np.float64)
        assert a.size == self.states
        lib.rdamd_set_frequencies(self._h, idx, _dptr(a))

    def set_category_rates(self, rates):
        a = np.ascontiguousarray(rates, dtype=np.float64)
        assert a.size == self.rate_cats
        lib.rdamd_set_category_rates(self._h, _dptr(a))

    def set_category_weights(self, w):
        a = np.ascontiguousarray(w, dtype=np.float64)
        assert a.size == self.rate_cats
        lib.rdamd_set_category_weights(self._h, _dptr(a))

    def update_invariant_sites_proportion(self, idx, p):
        if lib.rdamd_update_invariant_sites_proportion(self._h, idx, p) != 1:
            _fail("update_invariant_sites_proportion")

    def empirical_frequencies(self):
        ptr = lib.rdamd_msa_empirical_frequencies(self._h)
        out = [ptr[i] for i in range(self.states)]
        _libc.free(C.cast(ptr, _vp))
        return out

    def subst_params(self, idx=0):
        ptr = lib.rdamd_partition_subst_params(self._h, idx)
        return [ptr[i] for i in range(self.states * self.states - self.states)]

    def update_prob_matrices(self, matrix_indices, branch_lengths, params_indices=None):
        mi = np.ascontiguousarray(matrix_indices, dtype=np.uint32)
        bl = np.ascontiguousarray(branch_lengths, dtype=np.float64)
        pi = self.params_indices if params_indices is None else np.ascontiguousarray(
            params_indices, dtype=np.uint32)
        if lib.rdamd_update_prob_matrices(self._h, _uptr(pi), _uptr(mi), _dptr(bl),
                                          mi.size) != 1:
            _fail("update_prob_matrices")

    def update_clvs(self, ops):
        n = len(ops)
        if not isinstance(ops, C.Array):
            arr = (Operation * n)()
            for i, o in enumerate(ops):
                arr[i] = o
            ops = arr
        lib.rdamd_update_clvs(self._h, ops, n)
        if _errno():
            _fail("update_clvs")

    def compute_root_loglikelihood(self, clv_index, scaler_index, freqs_indices=None,
                                   persite=False):
        fi = self.params_indices if freqs_indices is None else np.ascontiguousarray(
            freqs_indices, dtype=np.uint32)
        ps = np.zeros(self.sites, dtype=np.float64) if persite else None
        v = lib.rdamd_compute_root_loglikelihood(
            self._h, clv_index, scaler_index, _uptr(fi), _dptr(ps) if persite else None)
        if _errno():
            _fail("compute_root_loglikelihood")
        return (v, ps) if persite else v

    def compute_root_loglikelihoods(self, clv_indices, scaler_indices, freqs_indices=None):
        """corax_compute_root_loglikelihood for many root CLVs in one launch."""
        ci = np.ascontiguousarray(clv_indices, dtype=np.uint32)
        si = np.ascontiguousarray(scaler_indices, dtype=np.int32)
        fi = (np.zeros(self.rate_cats, dtype=np.uint32) if freqs_indices is None
              else np.ascontiguousarray(freqs_indices, dtype=np.uint32))
        out = np.zeros(ci.size, dtype=np.float64)
        if lib.rdamd_compute_root_loglikelihoods(self._h, ci.size, _uptr(ci),
                                                 si.ctypes.data_as(C.POINTER(C.c_int)), _uptr(fi),
                                                 _dptr(out)) != 1:
            _fail("compute_root_loglikelihoods")
        return out

    def root_loglikelihood_fused(self, root_op, lengths1, lengths2, params_indices=None):
        l1 = np.ascontiguousarray(lengths1, dtype=np.float64)
        l2 = np.ascontiguousarray(lengths2, dtype=np.float64)
        out = np.zeros(l1.size, dtype=np.float64)
        pi = self.params_indices if params_indices is None else np.ascontiguousarray(
            params_indices, dtype=np.uint32)
        if lib.rdamd_root_loglikelihood_fused(self._h, C.byref(root_op), _uptr(pi),
                                              _dptr(l1), _dptr(l2), l1.size, _dptr(out)) != 1:
            _fail("root_loglikelihood_fused")
        return out

    def get_clv(self, idx):
        out = np.zeros((self.sites, self.rate_cats, self.states), dtype=np.float64)
        if lib.rdamd_get_clv(self._h, idx, _dptr(out)) != 1:
            _fail("get_clv")
        return out

    def get_scaler(self, idx):
        out = np.zeros(self.sites, dtype=np.uint32)
        if lib.rdamd_get_scaler(self._h, idx, _uptr(out)) != 1:
            _fail("get_scaler")
        return out

    def get_pmatrix(self, idx):
        out = np.zeros((self.rate_cats, self.states, self.states), dtype=np.float64)
        if lib.rdamd_get_pmatrix(self._h, idx, _dptr(out)) != 1:
            _fail("get_pmatrix")
        return out

    def sync(self):
        lib.rdamd_partition_sync(self._h)

    def profile_enable(self, on=True):
        lib.rdamd_profile_enable(self._h, 1 if on else 0)

    def profile_read(self):
        """-> {family: (kernel ms, launches)} measured with HIP events on the
        partition stream; resets the accumulators."""
        ms = (C.c_double * 8)()
        n = (C.c_uint * 8)()
        if lib.rdamd_profile_read(self._h, ms, n) != 1:
            _fail("profile_read")
        names = ("clv", "pmatrix", "root", "fused", "fused_pmatrix")
        return {k: (ms[i], n[i]) for i, k in enumerate(names)}

    # ---- batched fused evaluation (rdamd_evaluate_batch) ---------------------
    def schedule(self, ops, matrix_indices, branch_lengths):
        return Schedule(self, ops, matrix_indices, branch_lengths)

    @staticmethod
    def schedule_handles(schedules):
        """the handle array rdamd_evaluate_batch takes, for callers that evaluate the same list
        of schedules repeatedly (evaluate_batch accepts it in place of the list)"""
        return (_vp * len(schedules))(*[s._h for s in schedules])

    def _batch_args(self, schedules, subst, freqs, rates, rate_weights):
        n = len(schedules)
        hs = schedules if isinstance(schedules, C.Array) else (_vp * n)(*[s._h for s in schedules])
        k = self.states                       # 4, 2 (binary data on the 4-state kernels) or 20
        subst = np.ascontiguousarray(subst, dtype=np.float64).reshape(n, k * k - k)
        freqs = np.ascontiguousarray(freqs, dtype=np.float64).reshape(n, k)
        if rates is not None:
            rates = np.ascontiguousarray(rates, dtype=np.float64).reshape(n, self.rate_cats)
        if rate_weights is not None:
            rate_weights = np.ascontiguousarray(rate_weights, dtype=np.float64).reshape(
                n, self.rate_cats)
        return n, hs, subst, freqs, rates, rate_weights

    def evaluate_batch(self, schedules, subst, freqs, rates=None, rate_weights=None):
        """lnL of every (schedule, parameter set) job in one fused launch."""
        n, hs, subst, freqs, rates, rw = self._batch_args(schedules, subst, freqs, rates,
                                                          rate_weights)
        out = np.zeros(n, dtype=np.float64)
        if lib.rdamd_evaluate_batch(self._h, n, hs, _dptr(subst), _dptr(freqs),
                                    _dptr(rates) if rates is not None else None,
                                    _dptr(rw) if rw is not None else None, _dptr(out)) != 1:
            _fail("evaluate_batch")
        return out

    def set_stream_priority(self, level):
        """-1 high, 0 normal, +1 low (rdamd_partition_set_stream_priority): the lock-stepped search
        runs its long objective launches low and the replicas' short kernels high."""
        if lib.rdamd_partition_set_stream_priority(self._h, level) != 1:
            _fail("set_stream_priority")

    def evaluate_root_children(self, ops, matrix_indices, branch_lengths, subst, freqs, rates=None,
                               rate_weights=None):
        """lnL of the whole operation list through the fused evaluator; the root operation's two
        children are left in the partition's CLV / scaler buffers (rdamd_evaluate_root_children)."""
        n = len(ops)
        if not isinstance(ops, C.Array):
            arr = (Operation * n)()
            for i, o in enumerate(ops):
                arr[i] = o
            ops = arr
        mi = np.ascontiguousarray(matrix_indices, dtype=np.uint32)
        bl = np.ascontiguousarray(branch_lengths, dtype=np.float64)
        subst = np.ascontiguousarray(subst, dtype=np.float64)
        freqs = np.ascontiguousarray(freqs, dtype=np.float64)
        rates = None if rates is None else np.ascontiguousarray(rates, dtype=np.float64)
        rw = None if rate_weights is None else np.ascontiguousarray(rate_weights, dtype=np.float64)
        out = C.c_double(0.0)
        if lib.rdamd_evaluate_root_children(self._h, ops, n, _uptr(mi), _dptr(bl), mi.size, _dptr(subst),
                                            _dptr(freqs), _dptr(rates) if rates is not None else None,
                                            _dptr(rw) if rw is not None else None,
                                            C.cast(C.byref(out), _pd)) != 1:
            _fail("evaluate_root_children")
        return out.value

    def evaluate_batch_submit(self, slot, schedules, subst, freqs, rates=None, rate_weights=None):
        """Queue the batch on `slot` (0 or 1) and return (rdamd_evaluate_batch_submit); returns the
        number of jobs, which evaluate_batch_wait needs."""
        n, hs, subst, freqs, rates, rw = self._batch_args(schedules, subst, freqs, rates,
                                                          rate_weights)
        if lib.rdamd_evaluate_batch_submit(self._h, slot, n, hs, _dptr(subst), _dptr(freqs),
                                           _dptr(rates) if rates is not None else None,
                                           _dptr(rw) if rw is not None else None) != 1:
            _fail("evaluate_batch_submit")
        return n

    def evaluate_batch_submit_device(self, slot, schedules, subst, freqs, device_ptr, rates=None,
                                     rate_weights=None):
        """The stream-ordered form (rdamd_evaluate_batch_submit_device): results to device memory
        at `device_ptr` (n + 1 float64: the lnLs, then the second-pass flag as 0.0 / 1.0),
        written by the batch's finishing kernel; returns the number of jobs."""
        n, hs, subst, freqs, rates, rw = self._batch_args(schedules, subst, freqs, rates,
                                                          rate_weights)
        if lib.rdamd_evaluate_batch_submit_device(self._h, slot, n, hs, _dptr(subst), _dptr(freqs),
                                                  _dptr(rates) if rates is not None else None,
                                                  _dptr(rw) if rw is not None else None,
                                                  C.c_void_p(device_ptr)) != 1:
            _fail("evaluate_batch_submit_device")
        return n

    def evaluate_batch_redo_device(self, slot, device_ptr):
        if lib.rdamd_evaluate_batch_redo_device(self._h, slot, C.c_void_p(device_ptr)) != 1:
            _fail("evaluate_batch_redo_device")

    def evaluate_batch_finish_device(self, slot):
        if lib.rdamd_evaluate_batch_finish_device(self._h, slot) != 1:
            _fail("evaluate_batch_finish_device")

    def discard_clvs(self):
        """ATTRIB_SPARSE_CLVS partitions: every CLV / scale buffer gives its memory back."""
        lib.rdamd_partition_discard_clvs(self._h)

    def update_clvs_launches(self):
        """kernel launches the last update_clvs call took (a list that leaves the device empty is
        cut into independent subtrees, one launch per level of the cut)"""
        return int(lib.rdamd_update_clvs_launches(self._h))

    def set_rescale_speculation(self, mode):
        """-1 default (on up to 256 tips), 0 rescale tests on every step, 1 no tests in the first pass
        + a check of every site's sum at the root (include/root_digger_amd.h)"""
        if lib.rdamd_partition_set_rescale_speculation(self._h, int(mode)) != 1:
            _fail("set_rescale_speculation")

    def second_passes(self):
        """batches of this partition that needed their second evaluator pass so far"""
        return int(lib.rdamd_evaluate_second_passes(self._h))

    def clv_bytes(self):
        """device bytes the CLV and scale buffers hold right now"""
        return int(lib.rdamd_partition_clv_bytes(self._h))

    def evaluate_batch_wait(self, slot, n):
        """Results of the batch last submitted on `slot`."""
        out = np.zeros(n, dtype=np.float64)
        if lib.rdamd_evaluate_batch_wait(self._h, slot, _dptr(out)) != 1:
            _fail("evaluate_batch_wait")
        return out

    def evaluate_batch_device(self, schedules, subst, freqs, device_ptr, rates=None,
                              rate_weights=None):
        """Same, results left in device memory at `device_ptr` (n float64)."""
        n, hs, subst, freqs, rates, rw = self._batch_args(schedules, subst, freqs, rates,
                                                          rate_weights)
        if lib.rdamd_evaluate_batch_device(self._h, n, hs, _dptr(subst), _dptr(freqs),
                                           _dptr(rates) if rates is not None else None,
                                           _dptr(rw) if rw is not None else None,
                                           C.c_void_p(device_ptr)) != 1:
            _fail("evaluate_batch_device")


_OPTION_STRINGS = ("msa_filename", "tree_filename", "prefix", "prefix_dir", "model_filename",
                   "freqs_filename", "partition_filename", "data_type", "model_string")
_OPTION_SCALARS = ("seed", "min_roots", "threads", "root_ratio", "abs_tolerance", "factor",
                   "br_tolerance", "bfgs_tol", "silent", "exhaustive", "echo",
                   "invariant_sites", "early_stop", "initial_root_strategy")
PARAM_FIELDS = ("subst_rates", "freqs", "gamma_alpha", "gamma_weights")


def _flatten_params(params):
    """[{subst_rates, freqs, gamma_alpha, gamma_weights}, ...] -> (counts, values)"""
    counts = np.array([len(p.get(f, ())) for p in params for f in PARAM_FIELDS], dtype=np.uint64)
    values = np.array([v for p in params for f in PARAM_FIELDS for v in p.get(f, ())],
                      dtype=np.float64)
    return counts, np.ascontiguousarray(values if values.size else np.zeros(1))


PARAM_TYPES = ("emperical", "estimate", "equal", "user")          # src/util.hpp:37
RATE_CATEGORIES = ("MEDIAN", "MEAN", "FREE")                       # :48
ASC_TYPES = ("lewis", "fels", "stam")                              # :72


def _info_dict(pi, with_ranges):
    d = {"model_name": pi.model_name.decode(), "subst_str": pi.subst_str.decode(),
         "freq_type": PARAM_TYPES[pi.freq_type],
         "invar": ({"type": PARAM_TYPES[pi.invar_type], "user_prop": pi.invar_user_prop}
                   if pi.invar_present else None),
         "ratehet": {"type": PARAM_TYPES[pi.ratehet.type],
                     "rate_category_type": RATE_CATEGORIES[pi.ratehet.rate_category_type],
                     "rate_cats": int(pi.ratehet.rate_cats), "alpha_init": bool(pi.ratehet.alpha_init),
                     "alpha": pi.ratehet.alpha},
         "asc": ({"type": ASC_TYPES[pi.asc_type], "fels_weight": pi.asc_fels_weight,
                  "stam_weights": [pi.stam_weights[i] for i in range(pi.n_stam_weights)]}
                 if pi.asc_present else None)}
    if with_ranges:
        d["partition_name"] = pi.partition_name.decode()
        d["parts"] = [(int(pi.ranges[i][0]), int(pi.ranges[i][1])) for i in range(pi.n_ranges)]
    return d


def parse_model_info(model_string):
    """parse_model_info (src/msa.cpp:364-415) as a dict."""
    pi = PartitionInfo()
    if lib.rdamd_parse_model_info(model_string.encode(), C.byref(pi)) != 1:
        _fail("parse_model_info")
    return _info_dict(pi, False)


def parse_partition_info(line):
    """parse_partition_info (src/msa.cpp:417-506) as a dict."""
    pi = PartitionInfo()
    if lib.rdamd_parse_partition_info(line.encode(), C.byref(pi)) != 1:
        _fail("parse_partition_info")
    return _info_dict(pi, True)


def msa_partition_probe(path, lines, cmap=None, compress=True):
    """msa_t::partition on a file: [(length, total_weight)] per partition line."""
    arr = (C.c_char_p * len(lines))(*[l.encode() for l in lines])
    n = np.zeros(len(lines), dtype=np.uint32)
    w = np.zeros(len(lines), dtype=np.uint32)
    if lib.rdamd_msa_partition_probe(os.fsencode(path), cmap, len(lines), arr, int(compress),
                                     _uptr(n), _uptr(w)) != 1:
        _fail("msa_partition_probe")
    return [(int(a), int(b)) for a, b in zip(n, w)]


class Checkpoint:
    """checkpoint_t of the reference (src/checkpoint.hpp:231-300): the
    `<prefix>.ckp` result log, byte-compatible, shared between processes under
    an fcntl lock."""

    def __init__(self, prefix):
        self._h = lib.rdamd_checkpoint_open(os.fsencode(prefix))
        if not self._h:
            _fail("checkpoint_open")

    def close(self):
        if getattr(self, "_h", None):
            lib.rdamd_checkpoint_close(self._h)
            self._h = None

    def __del__(self):
        self.close()

    @property
    def handle(self):
        return self._h

    def existing_checkpoint(self):
        return bool(lib.rdamd_checkpoint_existing(self._h))

    def get_filename(self):
        return os.fsdecode(lib.rdamd_checkpoint_filename(self._h))

    def save_options(self, options):
        """options: dict of the cli_options_t fields (missing ones keep the
        reference's defaults); rate_cats = list of dicts or ints."""
        o = CliOptions()
        keep = []
        for k in _OPTION_STRINGS:
            b = os.fsencode(str(options.get(k, "")))
            keep.append(b)
            setattr(o, k, b)
        cats = options.get("rate_cats", [1])
        arr = (RatehetOpts * len(cats))()
        for i, c in enumerate(cats):
            c = {"rate_cats": c} if isinstance(c, int) else c
            arr[i] = RatehetOpts(c.get("type", 1), c.get("rate_category_type", 1),
                                 c.get("rate_cats", 1), int(c.get("alpha_init", False)),
                                 c.get("alpha", 1.0))
        o.rate_cats, o.n_rate_cats = arr, len(cats)
        defaults = dict(seed=0, min_roots=1, threads=0, root_ratio=0.01, abs_tolerance=1e-7,
                        factor=1e4, br_tolerance=1e-12, bfgs_tol=1e-7, silent=0, exhaustive=0,
                        echo=0, invariant_sites=0, early_stop=0, initial_root_strategy=2)
        for k in _OPTION_SCALARS:
            v = options.get(k, defaults[k])
            setattr(o, k, int(v) if isinstance(v, bool) else v)
        if lib.rdamd_checkpoint_save_options(self._h, C.byref(o)) != 1:
            _fail("checkpoint_save_options")

    def load_options(self):
        """the header of an existing checkpoint as a dict (None for a new file)."""
        if not self.existing_checkpoint():
            return None
        o = CliOptions()
        if lib.rdamd_checkpoint_load_options(self._h, C.byref(o)) != 1:
            _fail("checkpoint_load_options")
        out = {k: os.fsdecode(getattr(o, k) or b"") for k in _OPTION_STRINGS}
        out["rate_cats"] = [
            {f: getattr(o.rate_cats[i], f) for f, _ in RatehetOpts._fields_}
            for i in range(o.n_rate_cats)]
        out.update({k: getattr(o, k) for k in _OPTION_SCALARS})
        return out

    def write(self, root_id, llh, alpha, params):
        counts, values = _flatten_params(params)
        if lib.rdamd_checkpoint_write(self._h, root_id, llh, alpha, len(params),
                                      counts.ctypes.data_as(_pu64), _dptr(values)) != 1:
            _fail("checkpoint_write")

    def read_results(self):
        """[(root_id, llh, alpha, [params per partition]), ...] in file order."""
        n = C.c_uint(0)
        if lib.rdamd_checkpoint_read_results(self._h, C.byref(n)) != 1:
            _fail("checkpoint_read_results")
        out = []
        for i in range(n.value):
            rid, npart, nval = C.c_uint64(0), C.c_uint(0), C.c_uint64(0)
            llh, alpha = C.c_double(0), C.c_double(0)
            lib.rdamd_checkpoint_result(self._h, i, C.byref(rid), C.byref(llh), C.byref(alpha),
                                        C.byref(npart), C.byref(nval))
            counts = np.zeros(4 * npart.value, dtype=np.uint64)
            values = np.zeros(max(nval.value, 1), dtype=np.float64)
            lib.rdamd_checkpoint_result_params(self._h, i, counts.ctypes.data_as(_pu64),
                                               _dptr(values))
            params, at = [], 0
            for p in range(npart.value):
                d = {}
                for k, f in enumerate(PARAM_FIELDS):
                    c = int(counts[4 * p + k])
                    d[f] = values[at:at + c].tolist()
                    at += c
                params.append(d)
            out.append((int(rid.value), llh.value, alpha.value, params))
        return out

    def current_progress(self):
        return [(r, l, a) for r, l, a, _ in self.read_results()]

    def completed_indicies(self):
        return [r for r, _, _, _ in self.read_results()]

    def needs_cleaning(self):
        r = lib.rdamd_checkpoint_needs_cleaning(self._h)
        if r < 0:
            _fail("checkpoint_needs_cleaning")
        return bool(r)

    def clean(self):
        if lib.rdamd_checkpoint_clean(self._h) != 1:
            _fail("checkpoint_clean")


class Comm:
    """RCCL communicator of one site group (rdamd_comm_t)."""

    @staticmethod
    def unique_id():
        buf = (C.c_char * 128)()
        if lib.rdamd_comm_unique_id(buf) != 1:
            _fail("comm_unique_id")
        return bytes(buf)

    def __init__(self, unique_id, rank, n_ranks):
        buf = (C.c_char * 128).from_buffer_copy(unique_id)
        self._h = lib.rdamd_comm_create(buf, rank, n_ranks)
        if not self._h:
            _fail("comm_create")

    @property
    def handle(self):
        return self._h

    @property
    def reducer(self):
        """the C reducer to hand to Model.set_lnl_reducer(..., on_device=True, user=comm.handle)"""
        return C.cast(lib.rdamd_comm_reducer, _vp)

    def allreduce_sum(self, device_ptr, n, stream=None):
        if lib.rdamd_comm_allreduce_sum(self._h, device_ptr, n, stream) != 1:
            _fail("comm_allreduce_sum")

    def set_sum_mode(self, mode):
        """COMM_SUM_GATHER (default: ncclAllGather + a sum in rank order, the same bits on every
        rank by construction) or COMM_SUM_ALLREDUCE (one ncclAllReduce)"""
        if lib.rdamd_comm_set_sum_mode(self._h, mode) != 1:
            _fail("comm_set_sum_mode")

    @property
    def sum_mode(self):
        return int(lib.rdamd_comm_sum_mode(self._h))

    def destroy(self):
        if getattr(self, "_h", None):
            lib.rdamd_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        self.destroy()


def rank_order_sum(gathered_ptr, out_ptr, n, ranks, stream=None):
    """out[i] = ((g[0][i] + g[1][i]) + ...) over `ranks` device vectors of n doubles lying one
    behind the other (rdamd_rank_order_sum: the second half of COMM_SUM_GATHER)"""
    if lib.rdamd_rank_order_sum(gathered_ptr, out_ptr, n, ranks, stream) != 1:
        _fail("rank_order_sum")


def checkpoint_checksum_result(root_id, llh, alpha):
    return int(lib.rdamd_checkpoint_checksum_result(root_id, llh, alpha))


def checkpoint_checksum_params(params):
    counts, values = _flatten_params(params)
    return int(lib.rdamd_checkpoint_checksum_params(len(params), counts.ctypes.data_as(_pu64),
                                                    _dptr(values)))


class Model:
    """model_t (src/model.hpp:47) through the C ABI, one partition."""

    def __init__(self, tree, seqs, states=4, cmap=None, rate_cats=1, weights=None, seed=1,
                 early_stop=False):
        labels = list(seqs)
        n = len(labels)
        lab = (C.c_char_p * n)(*[l.encode() for l in labels])
        sq = (C.c_char_p * n)(*[seqs[l].encode() for l in labels])
        w = None
        if weights is not None:
            w = np.ascontiguousarray(weights, dtype=np.uint32)
        self._tree = tree
        self.states = states
        self._h = lib.rdamd_model_create(tree._h, n, lab, sq, _uptr(w) if w is not None else None,
                                         states, cmap if cmap is not None else MAP_NT, rate_cats,
                                         seed, 1 if early_stop else 0)
        if not self._h:
            _fail("model_create")

    @classmethod
    def from_file(cls, tree, msa_path, states=4, cmap=None, rate_cats=1, seed=1,
                  early_stop=False, compress=True, rate_category_type="mean"):
        """model_t over msa_t(filename): PHYLIP / FASTA ingest + pattern compression."""
        self = cls.__new__(cls)
        self._tree, self.states = tree, states
        n = C.c_uint(0)
        if rate_category_type != "mean":
            rc = RatehetOpts(1, RATE_CATEGORIES.index(rate_category_type.upper()), rate_cats, 0, 1.0)
            self._h = lib.rdamd_model_create_from_file_ratehet(
                tree._h, str(msa_path).encode(), states, cmap if cmap is not None else MAP_NT,
                C.byref(rc), seed, 1 if early_stop else 0, 1 if compress else 0, C.byref(n))
        else:
            self._h = lib.rdamd_model_create_from_file(
                tree._h, str(msa_path).encode(), states, cmap if cmap is not None else MAP_NT,
                rate_cats, seed, 1 if early_stop else 0, 1 if compress else 0, C.byref(n))
        if not self._h:
            _fail("model_create_from_file")
        self.patterns = n.value
        return self

    @classmethod
    def from_partition_file(cls, tree, msa_path, partition_path, states=4, cmap=None, seed=1,
                            early_stop=False):
        """The reference's partitioned set-up (src/main.cpp:512-555): one model
        partition per line of the partition file (column ranges + model string;
        rate categories come from each line's +G / +R option)."""
        self = cls.__new__(cls)
        self._tree, self.states = tree, states
        n = C.c_uint(0)
        self._h = lib.rdamd_model_create_partitioned(
            tree._h, os.fsencode(msa_path), os.fsencode(partition_path), states,
            cmap if cmap is not None else MAP_NT, seed, 1 if early_stop else 0, C.byref(n))
        if not self._h:
            _fail("model_create_partitioned")
        self.partitions = n.value
        return self

    @classmethod
    def from_file_block(cls, tree, msa_path, block, n_blocks, states=4, cmap=None, rate_cats=1,
                        seed=1, early_stop=False, compress=True):
        """One rank's model of a site-sharded run: column block `block` of `n_blocks`
        (rdamd_model_create_from_file_block); pair it with set_lnl_reducer."""
        self = cls.__new__(cls)
        self._tree, self.states = tree, states
        n, cols = C.c_uint(0), C.c_uint(0)
        rc = RatehetOpts(1, 1, rate_cats, 0, 1.0)
        self._h = lib.rdamd_model_create_from_file_block(
            tree._h, os.fsencode(msa_path), states, cmap if cmap is not None else MAP_NT,
            C.byref(rc), seed, 1 if early_stop else 0, 1 if compress else 0, block, n_blocks,
            C.byref(n), C.byref(cols))
        if not self._h:
            _fail("model_create_from_file_block")
        self.patterns, self.columns = n.value, cols.value
        return self

    def set_lnl_reducer(self, fn, on_device=False, user=None):
        """Site-group reduction hook (rdamd_model_set_lnl_reducer).  `fn` is either a
        Python callable(values: float64 numpy view, n) -> None summing a HOST
        array over the group in place, or a C function pointer (LNL_REDUCER /
        address, e.g. the library's rdamd_comm_reducer with on_device=True and
        `user` = the communicator handle)."""
        if fn is None:
            self._reducer = None
            self._ok(lib.rdamd_model_set_lnl_reducer(self._h, None, None, 0), "set_lnl_reducer")
            return
        if callable(fn) and not isinstance(fn, C._CFuncPtr):
            def trampoline(values, n, stream, user_, fn=fn):
                try:
                    fn(np.ctypeslib.as_array(values, shape=(n,)), n)
                    return 1
                except Exception:           # never let an exception cross the C frame
                    import traceback
                    traceback.print_exc()
                    return 0
            fn = LNL_REDUCER(trampoline)
        self._reducer = fn                   # keep the callback alive
        self._ok(lib.rdamd_model_set_lnl_reducer(self._h, C.cast(fn, _vp), user,
                                                 1 if on_device else 0), "set_lnl_reducer")

    def max_replicas(self, requested):
        """-> (replicas that fit the free device memory, bytes per replica)"""
        b = C.c_uint64(0)
        n = lib.rdamd_model_max_replicas(self._h, requested, C.byref(b))
        return int(n), int(b.value)

    def partition_count(self):
        return int(lib.rdamd_model_partition_count(self._h))

    def destroy(self):
        if getattr(self, "_h", None):
            lib.rdamd_model_destroy(self._h)
            self._h = None

    def __del__(self):
        self.destroy()

    def _ok(self, rc, what):
        if rc != 1:
            _fail(what)

    def initialize_partitions(self):
        self._ok(lib.rdamd_model_initialize_partitions(self._h, 0), "initialize_partitions")

    def initialize_partitions_uniform_freqs(self):
        self._ok(lib.rdamd_model_initialize_partitions(self._h, 1), "initialize_partitions")

    def set_subst_rates(self, rates):
        a = np.ascontiguousarray(rates, dtype=np.float64)
        self._ok(lib.rdamd_model_set_subst_rates(self._h, _dptr(a)), "set_subst_rates")

    def set_subst_rates_uniform(self):
        self._ok(lib.rdamd_model_set_subst_rates_uniform(self._h), "set_subst_rates_uniform")

    def set_freqs(self, freqs):
        a = np.ascontiguousarray(freqs, dtype=np.float64)
        self._ok(lib.rdamd_model_set_freqs(self._h, _dptr(a)), "set_freqs")

    def set_empirical_freqs(self):
        self._ok(lib.rdamd_model_set_empirical_freqs(self._h), "set_empirical_freqs")

    def set_gamma_alpha(self, alpha):
        self._ok(lib.rdamd_model_set_gamma_alpha(self._h, alpha), "set_gamma_alpha")

    def compute_lh(self, rl):
        v = lib.rdamd_model_compute_lh(self._h, C.byref(rl))
        if _errno():
            _fail("compute_lh")
        return v

    def compute_lh_root(self, rl):
        v = lib.rdamd_model_compute_lh_root(self._h, C.byref(rl))
        if _errno():
            _fail("compute_lh_root")
        return v

    def compute_dlh(self, rl):
        out = (C.c_double * 2)()
        self._ok(lib.rdamd_model_compute_dlh(self._h, C.byref(rl), out), "compute_dlh")
        return out[0], out[1]

    def move_root(self, rl):
        self._ok(lib.rdamd_model_move_root(self._h, C.byref(rl)), "move_root")

    def compute_all_root_lh(self):
        out = np.zeros(self._tree.root_count(), dtype=np.float64)
        self._ok(lib.rdamd_model_compute_all_root_lh(self._h, _dptr(out)), "compute_all_root_lh")
        return out

    def compute_all_root_lh_batched(self):
        out = np.zeros(self._tree.root_count(), dtype=np.float64)
        self._ok(lib.rdamd_model_compute_all_root_lh_batched(self._h, _dptr(out)),
                 "compute_all_root_lh_batched")
        return out

    def search(self, min_roots, root_ratio, atol, pgtol, brtol, factor):
        best = RootLocation()
        llh = C.c_double(0.0)
        self._ok(lib.rdamd_model_search(self._h, min_roots, root_ratio, atol, pgtol, brtol, factor,
                                        C.byref(best), C.byref(llh)), "search")
        return best, llh.value

    def optimize_alpha(self, rl, atol):
        out = RootLocation()
        self._ok(lib.rdamd_model_optimize_alpha(self._h, C.byref(rl), atol, C.byref(out)),
                 "optimize_alpha")
        return out

    def compute_lh_batch(self, rls, subst, freqs, gamma_alpha=None):
        n = len(rls)
        arr = (RootLocation * n)(*rls)
        k = self.states
        subst = np.ascontiguousarray(subst, dtype=np.float64).reshape(n, k * k - k)
        freqs = np.ascontiguousarray(freqs, dtype=np.float64).reshape(n, k)
        ga = None if gamma_alpha is None else np.ascontiguousarray(gamma_alpha, dtype=np.float64)
        out = np.zeros(n, dtype=np.float64)
        self._ok(lib.rdamd_model_compute_lh_batch(self._h, n, arr, _dptr(subst), _dptr(freqs),
                                                  _dptr(ga) if ga is not None else None,
                                                  _dptr(out)), "compute_lh_batch")
        return out

    def set_lbfgsb(self, setulb):
        """`setulb`: the caller's L-BFGS-B entry point (a ctypes function or address)."""
        addr = setulb if isinstance(setulb, int) else C.cast(setulb, _vp).value
        lib.rdamd_model_set_lbfgsb(self._h, C.c_void_p(addr))

    def optimize_params(self, rl, subst, freqs, gamma_alpha, pgtol, factor, optimize_gamma=True):
        subst = np.array(subst, dtype=np.float64)
        freqs = np.array(freqs, dtype=np.float64)
        ga = C.c_double(gamma_alpha)
        nb, ne = C.c_uint64(0), C.c_uint64(0)
        self._ok(lib.rdamd_model_optimize_params(self._h, C.byref(rl), pgtol, factor,
                                                 1 if optimize_gamma else 0, _dptr(subst),
                                                 _dptr(freqs), C.byref(ga), C.byref(nb),
                                                 C.byref(ne)), "optimize_params")
        return {"subst": subst, "freqs": freqs, "gamma_alpha": ga.value,
                "batches": nb.value, "evaluations": ne.value}

    def compute_all_root_lh_directional(self, ratios=None):
        """every root's lnL at the current parameters through the all-directions
        CLV cache (rdamd_model_compute_all_root_lh_directional)."""
        out = np.zeros(self._tree.root_count(), dtype=np.float64)
        r = None if ratios is None else np.ascontiguousarray(ratios, dtype=np.float64)
        self._ok(lib.rdamd_model_compute_all_root_lh_directional(
            self._h, _dptr(r) if r is not None else None, _dptr(out)), "compute_all_root_lh_directional")
        return out

    def counters(self):
        """work counters since creation (rdamd_model_counters)."""
        out = (C.c_uint64 * 6)()
        lib.rdamd_model_counters(self._h, out)
        names = ("objective_batches", "objective_evaluations", "full_traversals",
                 "root_positions", "move_root_calls", "setulb_calls")
        return dict(zip(names, (int(v) for v in out)))

    def set_lockstep_groups(self, groups):
        """0: the library's choice (two pipelined groups from four candidates in flight on);
        1: one group, blocking launches (rdamd_model_set_lockstep_groups)."""
        lib.rdamd_model_set_lockstep_groups(self._h, groups)

    def set_lockstep_rounds(self, mode):
        """-1 (default): a site-sharded model's lock-stepped search runs in deterministic rounds,
        others in arrival order; 0: arrival order; 1: rounds (rdamd_model_set_lockstep_rounds)."""
        lib.rdamd_model_set_lockstep_rounds(self._h, mode)

    def round_stats(self):
        """the last search in rounds + this model's own collectives (rdamd_model_round_stats)."""
        out = (C.c_uint64 * 4)()
        lib.rdamd_model_round_stats(self._h, out)
        sec = (C.c_double * 4)()
        lib.rdamd_model_round_seconds(self._h, sec)
        d = dict(zip(("rounds", "collectives", "redos", "own_collectives"), (int(v) for v in out)))
        d["seconds"] = dict(zip(("objective_queued", "root_launch", "sum_queued", "waiting"), (float(v) for v in sec)))
        return d

    def set_root_children_only(self, on):
        """the searches' compute_lh in front of the root-only steps: True (default) = one fused job
        that leaves only the root's children behind, False = the full traversal."""
        lib.rdamd_model_set_root_children_only(self._h, 1 if on else 0)

    def set_lockstep_priority(self, level):
        """stream priority of the shared objective partition during a lock-stepped search
        (+1 low, the default; 0 unchanged)."""
        lib.rdamd_model_set_lockstep_priority(self._h, level)

    def lockstep_stats(self):
        """combined launches of the last lock-stepped search (rdamd_model_lockstep_stats)."""
        out = (C.c_uint64 * 4)()
        lib.rdamd_model_lockstep_stats(self._h, out)
        return dict(zip(("objective_launches", "objective_jobs", "root_launches", "root_steps"),
                        (int(v) for v in out)))

    def assign_by_rank(self, rank, num_tasks, checkpoint=None):
        """assign_indicies_by_rank_exhaustive; with a Checkpoint, the roots it
        already holds are skipped (src/model.cpp:1867-1911)."""
        if checkpoint is not None:
            self._ok(lib.rdamd_model_assign_by_rank_checkpoint(self._h, rank, num_tasks,
                                                               checkpoint.handle), "assign_by_rank")
        else:
            self._ok(lib.rdamd_model_assign_by_rank(self._h, rank, num_tasks), "assign_by_rank")

    def assign_by_rank_search(self, min_roots, root_ratio, rank, num_tasks,
                              initial_root_strategy="modified_mad", checkpoint=None):
        """assign_indicies_by_rank_search (src/model.cpp:1809-1865)."""
        strategy = {"random": 0, "midpoint": 1, "modified_mad": 2}[initial_root_strategy]
        self._ok(lib.rdamd_model_assign_by_rank_search(
            self._h, min_roots, root_ratio, rank, num_tasks, strategy,
            checkpoint.handle if checkpoint else None), "assign_by_rank_search")

    def assigned(self):
        n = self._tree.root_count()
        ids = (C.c_uint64 * n)()
        k = lib.rdamd_model_assigned(self._h, ids, n)
        return [int(ids[i]) for i in range(min(k, n))]

    def set_progress(self, on=True):
        """print the reference's "Step i / n, ETC" lines during searches (call after assign)."""
        lib.rdamd_model_set_progress(self._h, 1 if on else 0)

    def set_checkpoint(self, checkpoint):
        """searches append every finished candidate to this Checkpoint (None detaches)."""
        self._checkpoint = checkpoint          # keep it alive
        lib.rdamd_model_set_checkpoint(self._h, checkpoint.handle if checkpoint else None)

    def exhaustive_search(self, atol, pgtol, brtol, factor, workers=0, lockstep=0):
        """workers > 0: that many host threads, each with its own model replica.
        lockstep > 0: that many candidates in flight, their objective batches
        combined into one launch (rdamd_model_exhaustive_search_lockstep)."""
        n = self._tree.root_count()
        ids = (C.c_uint64 * n)()
        llh = np.zeros(n, dtype=np.float64)
        alpha = np.zeros(n, dtype=np.float64)
        cnt = C.c_uint(0)
        best = RootLocation()
        best_llh = C.c_double(0.0)
        if lockstep > 0:
            self._ok(lib.rdamd_model_exhaustive_search_lockstep(
                self._h, lockstep, atol, pgtol, brtol, factor, ids, _dptr(llh), _dptr(alpha),
                C.byref(cnt), C.byref(best), C.byref(best_llh)), "exhaustive_search_lockstep")
        elif workers > 0:
            self._ok(lib.rdamd_model_exhaustive_search_parallel(
                self._h, workers, atol, pgtol, brtol, factor, ids, _dptr(llh), _dptr(alpha),
                C.byref(cnt), C.byref(best), C.byref(best_llh)), "exhaustive_search_parallel")
        else:
            self._ok(lib.rdamd_model_exhaustive_search(self._h, atol, pgtol, brtol, factor, ids,
                                                       _dptr(llh), _dptr(alpha), C.byref(cnt),
                                                       C.byref(best), C.byref(best_llh)),
                     "exhaustive_search")
        k = cnt.value
        return {"root_id": [int(ids[i]) for i in range(k)], "llh": llh[:k].copy(),
                "alpha": alpha[:k].copy(), "best": best, "best_llh": best_llh.value}

"""root_digger_amd -- MI355X-native likelihood core for RootDigger's root search.

The product is the C-ABI shared library ``lib/librdamd.so`` (hand-written HIP
for gfx950 + host C++ mirroring rooted_tree_t / model_t).  This package is a
thin ctypes binding over that ABI (see include/root_digger_amd.h); it contains
no arithmetic of its own and no CPU fallback: importing works without a GPU
(host-side tree/schedule logic is usable), but creating a partition without a
HIP device raises.
"""
from ._lib import lib, lib_path, RdamdError  # noqa: F401
from .api import (  # noqa: F401
    Operation,
    RootLocation,
    Tree,
    Partition,
    Schedule,
    Model,
    Checkpoint,
    Comm,
    LNL_REDUCER,
    parse_model_info,
    parse_partition_info,
    msa_partition_probe,
    checkpoint_checksum_result,
    checkpoint_checksum_params,
    MAP_NT,
    MAP_BIN,
    compute_gamma_cats,
    root_loglikelihood_fused_multi,
    GAMMA_RATES_MEAN,
    GAMMA_RATES_MEDIAN,
    ATTRIB_SITE_REPEATS,
    ATTRIB_NONREV,
    ATTRIB_SPARSE_CLVS,
    device_count,
    hip_runtime_path,
    mapped_hip_runtimes,
    set_device,
    device_memory,
    msa_probe,
    rank_order_sum,
    COMM_SUM_GATHER,
    COMM_SUM_ALLREDUCE,
)

__all__ = [
    "lib", "lib_path", "RdamdError", "Operation", "RootLocation", "Tree", "Partition", "Schedule", "Model", "Checkpoint", "Comm", "LNL_REDUCER", "parse_model_info", "parse_partition_info",
    "msa_partition_probe",
    "checkpoint_checksum_result", "checkpoint_checksum_params",
    "MAP_NT", "MAP_BIN", "compute_gamma_cats", "root_loglikelihood_fused_multi", "GAMMA_RATES_MEAN", "GAMMA_RATES_MEDIAN",
    "ATTRIB_SITE_REPEATS", "ATTRIB_NONREV", "ATTRIB_SPARSE_CLVS",
    "device_count", "hip_runtime_path", "mapped_hip_runtimes", "set_device", "device_memory", "msa_probe",
    "rank_order_sum", "COMM_SUM_GATHER", "COMM_SUM_ALLREDUCE",
]

"""ctypes binding of the CPU oracle (oracle/librd_oracle.so) with the same
Python surface as root_digger_amd.Partition, so parity tests drive both with
the same code.  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_so = os.path.join(ROOT, "oracle", "librd_oracle.so")
olib = C.CDLL(_so)


class OrcOperation(C.Structure):
    _fields_ = [
        ("parent_clv_index", C.c_uint), ("parent_scaler_index", C.c_int),
        ("child1_clv_index", C.c_uint), ("child1_matrix_index", C.c_uint),
        ("child1_scaler_index", C.c_int), ("child2_clv_index", C.c_uint),
        ("child2_matrix_index", C.c_uint), ("child2_scaler_index", C.c_int),
    ]


_u, _vp = C.c_uint, C.c_void_p
_pu, _pd = C.POINTER(C.c_uint), C.POINTER(C.c_double)


def _sig(name, res, *args):
    f = getattr(olib, name)
    f.restype, f.argtypes = res, list(args)


_sig("orc_partition_create", _vp, *([_u] * 9))
_sig("orc_partition_destroy", None, _vp)
_sig("orc_set_tip_states", C.c_int, _vp, _u, C.POINTER(C.c_uint64), C.c_char_p)
_sig("orc_set_pattern_weights", None, _vp, _pu)
_sig("orc_set_subst_params", None, _vp, _u, _pd)
_sig("orc_set_frequencies", None, _vp, _u, _pd)
_sig("orc_set_category_rates", None, _vp, _pd)
_sig("orc_set_category_weights", None, _vp, _pd)
_sig("orc_msa_empirical_frequencies", _pd, _vp)
_sig("orc_compute_gamma_cats", C.c_int, C.c_double, _u, _pd, C.c_int)
_sig("orc_update_prob_matrices", C.c_int, _vp, _pu, _pu, _pd, _u)
_sig("orc_update_clvs", None, _vp, C.POINTER(OrcOperation), _u)
_sig("orc_update_clvs_avx2", C.c_int, _vp, C.POINTER(OrcOperation), _u)
_sig("orc_compute_root_loglikelihood", C.c_double, _vp, _u, C.c_int, _pu, _pd)
_sig("orc_update_clvs_repeats", None, _vp, C.POINTER(OrcOperation), _u, C.c_int)
_sig("orc_compute_root_loglikelihood_repeats", C.c_double, _vp, _u, C.c_int, _pu)
_sig("orc_repeats_ratio", C.c_double, _vp)
_sig("orc_repeats_bytes", None, _vp, C.POINTER(C.c_double))
_sig("orc_stream_triad", C.c_double, C.c_int, C.POINTER(C.c_int), C.c_size_t, C.c_double)
_sig("orc_get_clv", _pd, _vp, _u)
_sig("orc_get_scaler", _pu, _vp, _u)
_sig("orc_get_pmatrix", _pd, _vp, _u)
_sig("orc_get_qmatrix", None, _vp, _u, _pd)
_sig("orc_expm", None, _pd, _u, _pd)

ORC_MAP_NT = (C.c_uint64 * 256).in_dll(olib, "orc_map_nt")
_libc = C.CDLL(None)
_libc.free.argtypes = [_vp]


def orc_expm(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    n = a.shape[0]
    out = np.zeros_like(a)
    olib.orc_expm(a.ctypes.data_as(_pd), n, out.ctypes.data_as(_pd))
    return out


def orc_gamma_cats(alpha, cats, mode=0):
    out = (C.c_double * cats)()
    assert olib.orc_compute_gamma_cats(alpha, cats, out, mode) == 1
    return list(out)


class OraclePartition:
    def __init__(self, tips, clv_buffers, states, sites, rate_matrices, prob_matrices,
                 rate_cats, scale_buffers, attributes=0):
        self._h = olib.orc_partition_create(tips, clv_buffers, states, sites,
                                            rate_matrices, prob_matrices, rate_cats,
                                            scale_buffers, attributes)
        self.tips, self.clv_buffers, self.states, self.sites = tips, clv_buffers, states, sites
        self.rate_cats, self.prob_matrices, self.scale_buffers = rate_cats, prob_matrices, scale_buffers
        self.params_indices = np.zeros(rate_cats, dtype=np.uint32)

    @classmethod
    def for_tree(cls, tree, states, sites, rate_cats, attributes=0):
        b = tree.branch_count()
        return cls(tree.tip_count(), b, states, sites, 1, b, rate_cats, b, attributes)

    def destroy(self):
        if getattr(self, "_h", None):
            olib.orc_partition_destroy(self._h)
            self._h = None

    def __del__(self):
        self.destroy()

    def set_tip_states(self, tip_index, cmap, sequence):
        if isinstance(sequence, str):
            sequence = sequence.encode()
        if olib.orc_set_tip_states(self._h, tip_index, cmap, sequence) != 1:
            raise RuntimeError("oracle: bad character in sequence")

    def set_pattern_weights(self, w):
        w = np.ascontiguousarray(w, dtype=np.uint32)
        olib.orc_set_pattern_weights(self._h, w.ctypes.data_as(_pu))

    def set_subst_params(self, idx, params):
        a = np.ascontiguousarray(params, dtype=np.float64)
        olib.orc_set_subst_params(self._h, idx, a.ctypes.data_as(_pd))

    def set_frequencies(self, idx, freqs):
        a = np.ascontiguousarray(freqs, dtype=np.float64)
        olib.orc_set_frequencies(self._h, idx, a.ctypes.data_as(_pd))

    def set_category_rates(self, rates):
        a = np.ascontiguousarray(rates, dtype=np.float64)
        olib.orc_set_category_rates(self._h, a.ctypes.data_as(_pd))

    def set_category_weights(self, w):
        a = np.ascontiguousarray(w, dtype=np.float64)
        olib.orc_set_category_weights(self._h, a.ctypes.data_as(_pd))

    def update_invariant_sites_proportion(self, idx, p):
        assert p == 0.0

    def empirical_frequencies(self):
        ptr = olib.orc_msa_empirical_frequencies(self._h)
        out = [ptr[i] for i in range(self.states)]
        _libc.free(C.cast(ptr, _vp))
        return out

    def update_prob_matrices(self, matrix_indices, branch_lengths, params_indices=None):
        mi = np.ascontiguousarray(matrix_indices, dtype=np.uint32)
        bl = np.ascontiguousarray(branch_lengths, dtype=np.float64)
        pi = self.params_indices if params_indices is None else np.ascontiguousarray(
            params_indices, dtype=np.uint32)
        if olib.orc_update_prob_matrices(self._h, pi.ctypes.data_as(_pu),
                                         mi.ctypes.data_as(_pu), bl.ctypes.data_as(_pd),
                                         mi.size) != 1:
            raise RuntimeError("oracle: update_prob_matrices failed")

    def update_clvs(self, ops, avx2=False):
        """avx2=True: the 256-bit-vector loop (4 states), bit-identical results; what
        bench.py's cpu_baseline leg times."""
        if isinstance(ops, C.Array) and ops._type_ is OrcOperation:
            arr, n = ops, len(ops)
        else:
            n = len(ops)
            arr = (OrcOperation * n)()
            for i, o in enumerate(ops):
                for f, _ in OrcOperation._fields_:
                    setattr(arr[i], f, getattr(o, f))
        if avx2:
            if olib.orc_update_clvs_avx2(self._h, arr, n) != 1:
                raise RuntimeError("oracle: the AVX2 loop takes 4-state data (and an AVX2 build)")
        else:
            olib.orc_update_clvs(self._h, arr, n)

    def update_clvs_repeats(self, ops, avx2=True):
        """the traversal with subtree site repeats (orc_update_clvs_repeats): per-class buffers,
        read back by compute_root_loglikelihood_repeats; bench.py's honest CPU comparator"""
        arr = ops if isinstance(ops, C.Array) and ops._type_ is OrcOperation else self.pack_ops(ops)
        olib.orc_update_clvs_repeats(self._h, arr, len(arr), 1 if avx2 else 0)

    def compute_root_loglikelihood_repeats(self, clv_index, scaler_index, freqs_indices=None):
        fi = self.params_indices if freqs_indices is None else np.ascontiguousarray(
            freqs_indices, dtype=np.uint32)
        return olib.orc_compute_root_loglikelihood_repeats(self._h, clv_index, scaler_index,
                                                           fi.ctypes.data_as(_pu))

    def repeats_bytes(self):
        """-> (written, read at least once, read if nothing is cached): bytes moved by the
        site-repeats traversals of this partition so far (orc_repeats_bytes)"""
        out = (C.c_double * 3)()
        olib.orc_repeats_bytes(self._h, out)
        return float(out[0]), float(out[1]), float(out[2])

    def repeats_ratio(self):
        """classes computed / columns a plain loop would have computed so far"""
        return olib.orc_repeats_ratio(self._h)

    @staticmethod
    def pack_ops(ops):
        """operations as the oracle's own array (build once, reuse in timed loops)"""
        arr = (OrcOperation * len(ops))()
        for i, o in enumerate(ops):
            for f, _ in OrcOperation._fields_:
                setattr(arr[i], f, getattr(o, f))
        return arr

    def compute_root_loglikelihood(self, clv_index, scaler_index, freqs_indices=None,
                                   persite=False):
        fi = self.params_indices if freqs_indices is None else np.ascontiguousarray(
            freqs_indices, dtype=np.uint32)
        ps = np.zeros(self.sites, dtype=np.float64) if persite else None
        v = olib.orc_compute_root_loglikelihood(
            self._h, clv_index, scaler_index, fi.ctypes.data_as(_pu),
            ps.ctypes.data_as(_pd) if persite else None)
        return (v, ps) if persite else v

    def root_loglikelihood_fused(self, root_op, lengths1, lengths2, params_indices=None):
        out = []
        for a, b in zip(lengths1, lengths2):
            self.update_prob_matrices([root_op.child1_matrix_index, root_op.child2_matrix_index],
                                      [a, b], params_indices)
            self.update_clvs([root_op])
            out.append(self.compute_root_loglikelihood(root_op.parent_clv_index,
                                                       root_op.parent_scaler_index,
                                                       params_indices))
        return np.array(out)

    def get_clv(self, idx):
        n = self.sites * self.rate_cats * self.states
        ptr = olib.orc_get_clv(self._h, idx)
        return np.ctypeslib.as_array(ptr, shape=(n,)).reshape(
            self.sites, self.rate_cats, self.states).copy()

    def get_scaler(self, idx):
        ptr = olib.orc_get_scaler(self._h, idx)
        return np.ctypeslib.as_array(ptr, shape=(self.sites,)).copy()

    def get_pmatrix(self, idx):
        n = self.rate_cats * self.states * self.states
        ptr = olib.orc_get_pmatrix(self._h, idx)
        return np.ctypeslib.as_array(ptr, shape=(n,)).reshape(
            self.rate_cats, self.states, self.states).copy()

    def get_qmatrix(self, idx=0):
        out = np.zeros((self.states, self.states))
        olib.orc_get_qmatrix(self._h, idx, out.ctypes.data_as(_pd))
        return out

    def sync(self):
        pass


def stream_triad(cpus, doubles=1 << 24, seconds=1.0):
    """STREAM triad on one pinned thread per entry of `cpus` (orc_stream_triad): aggregate GB/s"""
    arr = (C.c_int * len(cpus))(*cpus)
    return float(olib.orc_stream_triad(len(cpus), arr, doubles, seconds))

"""ctypes binding of oracle/libtd_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module
(the product package tagdust_amd never does).  The library is the CPU restatement of the
reference's per-read HMM decoding path (oracle/td_oracle.c, pinned against the reference by
tests/test_oracle_golden.py).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libtd_oracle.so")
MAX_SEG = 64


class _Model(C.Structure):
    _fields_ = [
        ("S", C.c_int32), ("H", C.c_int32), ("C", C.c_int32), ("avg_len", C.c_int32),
        ("bg", C.c_float * 5),
        ("n_hmm", C.c_int32 * MAX_SEG), ("n_col", C.c_int32 * MAX_SEG),
        ("col_off", C.c_int32 * MAX_SEG), ("hmm_off", C.c_int32 * MAX_SEG),
        ("skip", C.c_float * MAX_SEG), ("type", C.c_int8 * MAX_SEG),
        ("finger_len", C.c_int32 * MAX_SEG),
        ("trans", C.c_void_p), ("eM", C.c_void_p), ("eI", C.c_void_p),
        ("sM", C.c_void_p), ("sI", C.c_void_p), ("label", C.c_void_p), ("A", C.c_void_p),
    ]


class _Params(C.Structure):
    _fields_ = [("threshold", C.c_float), ("minlen", C.c_int32), ("dust", C.c_int32), ("matchstart", C.c_int32), ("matchend", C.c_int32)]


class _Artifacts(C.Structure):
    _fields_ = [("string", C.c_void_p), ("s_index", C.c_void_p), ("n_seq", C.c_int32), ("filter_error", C.c_int32)]


RESULT_DTYPE = np.dtype([
    ("b_score", "<f4"), ("f_score", "<f4"), ("r_score", "<f4"), ("bar_prob", "<f4"),
    ("Q", "<f4"), ("read_type", "<i4"), ("barcode", "<i4"), ("fingerprint", "<i4"),
])

_lib = None


def build(force=False):
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "td_oracle.c")):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libtd_oracle.so"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.tdo_init_logsum.restype = None
        _lib.tdo_logsum.restype = C.c_float
        _lib.tdo_logsum.argtypes = [C.c_float, C.c_float]
        _lib.tdo_logsum_table.restype = C.POINTER(C.c_float)
        _lib.tdo_qvalue.restype = C.c_float
        _lib.tdo_qvalue.argtypes = [C.c_float, C.c_float, C.c_float]
        _lib.tdo_label_batch.restype = C.c_int
        _lib.tdo_label_batch.argtypes = [C.POINTER(_Model), C.POINTER(_Params), C.c_int,
                                         C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        _lib.tdo_label_batch_art.restype = C.c_int
        _lib.tdo_label_batch_art.argtypes = [C.POINTER(_Model), C.POINTER(_Params), C.POINTER(_Artifacts), C.c_int,
                                             C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        _lib.tdo_init_logsum()
    return _lib


def bound_margins(model, seqs, offs, info):
    """Smallest margins by which the device kernel's pruning bounds (`info` = tagdust_amd.lib.spec_prune_info()) dominate the DP
    values the oracle computes over a batch: [forward, backward, entry term] of the leading pruned segments, then [forward,
    backward, exit term] of the trailing ones (1e30 where a side has nothing pruned); negative = a bound is violated."""
    L = lib()
    L.tdo_bound_margins.restype = C.c_int
    L.tdo_bound_margins.argtypes = [C.POINTER(_Model), C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
    offs = np.ascontiguousarray(offs, dtype=np.int64)
    names = ("fb", "bwb", "wa", "wb", "fbs", "bws", "wc", "wd")
    tab = np.ascontiguousarray(np.stack([info[k] for k in names]), dtype=np.float32)
    out = np.zeros(6, np.float64)
    rc = L.tdo_bound_margins(C.byref(model.c), seqs.ctypes.data, offs.ctypes.data, len(offs) - 1, int(info["n_seg"]),
                             int(info["sfx_first"]), tab.ctypes.data, tab.shape[1], out.ctypes.data)
    if rc != 0:
        raise RuntimeError("tdo_bound_margins failed (%d)" % rc)
    return out


def lead_profile(model, seqs, offs, n_seg, floor=-103.98):
    """(last position with a posterior term of the first n_seg segments above `floor`, per read; per position: the largest such
    term, the largest forward value, the largest backward value - b_score -- rows of the second result)"""
    L = lib()
    L.tdo_lead_profile.restype = C.c_int
    L.tdo_lead_profile.argtypes = [C.POINTER(_Model), C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_int]
    seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
    offs = np.ascontiguousarray(offs, dtype=np.int64)
    n = len(offs) - 1
    last = np.zeros(n, np.int32)
    plen = int(np.diff(offs).max()) + 2
    prof = np.zeros(3 * plen, np.float32)
    rc = L.tdo_lead_profile(C.byref(model.c), seqs.ctypes.data, offs.ctypes.data, n, int(n_seg), float(floor), last.ctypes.data, prof.ctypes.data, plen)
    if rc != 0:
        raise RuntimeError("tdo_lead_profile failed (%d)" % rc)
    return last, prof.reshape(3, plen)


def lead_class_profile(model, seqs, offs, n_seg):
    """Per state class of the first n_seg segments (column, M / I; HMMs of a segment together) and position: the largest forward
    value and the largest backward value - b_score over the reads -- array [classes][2][max_len + 2]."""
    L = lib()
    L.tdo_lead_class_profile.restype = C.c_int
    L.tdo_lead_class_profile.argtypes = [C.POINTER(_Model), C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int, C.c_int]
    seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
    offs = np.ascontiguousarray(offs, dtype=np.int64)
    plen = int(np.diff(offs).max()) + 2
    out = np.zeros(256 * 2 * plen, np.float32)
    nc = L.tdo_lead_class_profile(C.byref(model.c), seqs.ctypes.data, offs.ctypes.data, len(offs) - 1, int(n_seg), out.ctypes.data, plen, 256)
    if nc < 0:
        raise RuntimeError("tdo_lead_class_profile failed")
    return out[:nc * 2 * plen].reshape(nc, 2, plen).copy()


def restart_margins(model, seqs, offs, n_seg, sfx_first, gq, gf):
    """Smallest margins by which the start of the device kernel's restarted sweeps (tagdust_amd.lib.spec_restart_info()) dominates
    the backward values of the leading / the forward values of the trailing segments the oracle computes: [backward, forward]."""
    L = lib()
    L.tdo_restart_margins.restype = C.c_int
    L.tdo_restart_margins.argtypes = [C.POINTER(_Model), C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
    offs = np.ascontiguousarray(offs, dtype=np.int64)
    tab = np.ascontiguousarray(np.concatenate([gq, gf]), dtype=np.float32)
    out = np.zeros(2, np.float64)
    rc = L.tdo_restart_margins(C.byref(model.c), seqs.ctypes.data, offs.ctypes.data, len(offs) - 1, int(n_seg), int(sfx_first),
                               tab.ctypes.data, tab.shape[1], out.ctypes.data)
    if rc != 0:
        raise RuntimeError("tdo_restart_margins failed (%d)" % rc)
    return out


def logsum_table():
    p = lib().tdo_logsum_table()
    return np.ctypeslib.as_array(p, shape=(16000,)).copy()


class OracleModel:
    """Holds the flattened tables (numpy, kept alive) and the C struct that points at them."""

    def __init__(self, md):
        """md: dict with S,H,C,avg_len,bg,n_hmm,n_col,skip,seg_type,seg_len,trans,eM,eI,sM,sI,label,A."""
        self.S, self.H, self.Ccols = int(md["S"]), int(md["H"]), int(md["C"])
        f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
        self.trans = f32(md["trans"]).reshape(self.Ccols, 9)
        self.eM = f32(md["eM"]).reshape(self.Ccols, 5)
        self.eI = f32(md["eI"]).reshape(self.Ccols, 5)
        self.sM = f32(md["sM"]).reshape(self.Ccols)
        self.sI = f32(md["sI"]).reshape(self.Ccols)
        self.label = np.ascontiguousarray(md["label"], dtype=np.int32).reshape(self.H)
        self.A = f32(md["A"]).reshape(self.H, self.H)
        m = _Model()
        m.S, m.H, m.C, m.avg_len = self.S, self.H, self.Ccols, int(md["avg_len"])
        for i in range(5):
            m.bg[i] = float(np.float32(md["bg"][i]))
        co = ho = 0
        for j in range(self.S):
            m.n_hmm[j] = int(md["n_hmm"][j])
            m.n_col[j] = int(md["n_col"][j])
            m.col_off[j], m.hmm_off[j] = co, ho
            co += m.n_hmm[j] * m.n_col[j]
            ho += m.n_hmm[j]
            m.skip[j] = float(np.float32(md["skip"][j]))
            t = int(md["seg_type"][j])
            m.type[j] = t
            m.finger_len[j] = int(md["seg_len"][j]) if t == ord("F") else 0
        assert co == self.Ccols and ho == self.H
        for name in ("trans", "eM", "eI", "sM", "sI", "label", "A"):
            setattr(m, name, getattr(self, name).ctypes.data)
        self.c = m


def label_batch(model, seqs, offs, threshold, minlen=16, dust=100, n_threads=1, artifacts=None, window=None):
    """Run the oracle over a batch.  seqs: uint8 codes (0..4) concatenated; offs: int64 [n+1].
    artifacts: None or (string uint8, s_index int32 [n_seq+1], filter_error) -- struct fasta as read_fasta() leaves
    it; matching depends on n_threads (match_to_reference pairs reads in fours per thread range).
    window: None or (matchstart, matchend) of -start / -end.
    Returns (results structured array, labels int8 laid out at offs[i]+i with len+1 entries,
    seq_after uint8)."""
    L = lib()
    offs = np.ascontiguousarray(offs, dtype=np.int64)
    n = len(offs) - 1
    seq_after = np.array(seqs, dtype=np.uint8, copy=True)
    labels = np.zeros(int(offs[-1]) + n, dtype=np.int8)
    res = np.zeros(n, dtype=RESULT_DTYPE)
    p = _Params(float(threshold), int(minlen), int(dust), -1 if window is None else int(window[0]), -1 if window is None else int(window[1]))
    art = None
    if artifacts is not None:
        a_str = np.ascontiguousarray(artifacts[0], dtype=np.uint8)
        a_idx = np.ascontiguousarray(artifacts[1], dtype=np.int32)
        art = C.byref(_Artifacts(a_str.ctypes.data, a_idx.ctypes.data, len(a_idx) - 1, int(artifacts[2])))
    rc = L.tdo_label_batch_art(C.byref(model.c), C.byref(p), art, int(n_threads), seq_after.ctypes.data,
                               offs.ctypes.data, n, labels.ctypes.data, res.ctypes.data)
    if rc != 0:
        raise RuntimeError("tdo_label_batch failed (%d)" % rc)
    return res, labels, seq_after

"""ctypes mirror of include/tagdust_hip.h (the drop-in boundary for the reference's run_pHMM(),
src/barcode_hmm.h:342).  Fails loudly when the HIP library is missing or no MI355X is present."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TD_LIB_PATH", os.path.join(HERE, "libtagdust_hip.so"))   # TD_LIB_PATH: A/B another build

MODE_GET_LABEL = 1
MODE_GET_PROB = 4
MODE_ARCH_COMP = 5
NUM_OUTCOME_SLOTS = 8
NUM_BARCODE_BINS = 256
NUM_COUNTERS = NUM_OUTCOME_SLOTS + NUM_BARCODE_BINS
NUM_DIAG_COUNTERS = 64

# every symbol include/tagdust_hip.h declares
ABI_SYMBOLS = [
    "td_ctx_create", "td_ctx_destroy", "td_last_error", "td_logsum_table", "td_model_upload", "td_set_params",
    "td_batch_upload", "td_batch_upload_ascii", "td_run", "td_sync", "td_batch_download", "td_counts_reset",
    "td_counts_get", "td_diag_get", "td_counts_device_ptr", "td_last_kernel_ms", "td_timeline_origin", "td_last_kernel_times", "td_batch_info", "td_set_option", "td_get_option", "td_set_artifacts", "td_spec_source", "td_spec_prune_info", "td_spec_restart_info",
    "td_submit", "td_wait", "td_host_alloc", "td_host_free", "td_set_batch_window", "td_set_window", "td_arch_scores",
]
MULTI_ABI_SYMBOLS = ["td_shard_bounds", "td_count_outcomes", "td_multi_create", "td_multi_destroy", "td_multi_last_error",
                     "td_multi_size", "td_multi_ctx", "td_multi_model_upload", "td_multi_set_params", "td_multi_set_window", "td_multi_set_artifacts",
                     "td_multi_decode", "td_multi_counts", "td_multi_counts_reset", "td_multi_uses_rccl", "td_bind_host_to_device", "td_host_halves_bench"]
IO_ABI_SYMBOLS = ["td_io_last_error", "td_reads_parse", "td_reads_free", "td_writer_open", "td_writer_write", "td_writer_close",
                  "td_fasta_parse", "td_fasta_free", "td_stream_run", "td_stream_run_multi", "td_stream_release", "td_format_q"]
MODEL_ABI_SYMBOLS = ["td_arch_parse", "td_arch_free", "td_sequence_stats", "td_sequence_stats_window", "td_model_build", "td_model_tables_free",
                     "td_calibration_emit", "td_calibration_select", "td_calibration_free", "td_estimate_threshold",
                     "td_compare_architectures", "td_simreads", "td_text_free"]

RESULT_DTYPE = np.dtype([
    ("f_score", "<f4"), ("b_score", "<f4"), ("r_score", "<f4"), ("bar_prob", "<f4"), ("mapq", "<f4"),
    ("read_type", "<i4"), ("barcode", "<i4"), ("fingerprint", "<i4"),
])


class TdError(RuntimeError):
    pass


class _Reads(C.Structure):
    _fields_ = [("n_reads", C.c_int64), ("text", C.c_void_p), ("name_off", C.POINTER(C.c_int64)),
                ("name_len", C.POINTER(C.c_int32)), ("qual_off", C.POINTER(C.c_int64)), ("offs", C.POINTER(C.c_int64)),
                ("codes", C.POINTER(C.c_uint8))]


class _Calibration(C.Structure):
    _fields_ = [("n_reads", C.c_int64), ("codes", C.POINTER(C.c_uint8)), ("offs", C.POINTER(C.c_int64)),
                ("is_random", C.POINTER(C.c_uint8)), ("scoring", C.c_void_p)]


class _SeqStats(C.Structure):
    _fields_ = [("background", C.c_double * 5), ("expected_5_len", C.c_double), ("expected_3_len", C.c_double),
                ("mean_5_len", C.c_double), ("stdev_5_len", C.c_double), ("mean_3_len", C.c_double),
                ("stdev_3_len", C.c_double), ("average_length", C.c_double), ("max_seq_len", C.c_int32)]


class _ModelDesc(C.Structure):
    _fields_ = [
        ("S", C.c_int32), ("H", C.c_int32), ("C", C.c_int32), ("avg_len", C.c_int32), ("bg", C.c_float * 5),
        ("n_hmm", C.c_void_p), ("n_col", C.c_void_p), ("skip", C.c_void_p), ("seg_type", C.c_void_p),
        ("finger_len", C.c_void_p), ("trans", C.c_void_p), ("eM", C.c_void_p), ("eI", C.c_void_p),
        ("sM", C.c_void_p), ("sI", C.c_void_p), ("label", C.c_void_p), ("A", C.c_void_p),
    ]


_lib = None


def load_library():
    """dlopen libtagdust_hip.so (no GPU needed for this step)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TdError("%s is missing: run `python -m tagdust_amd.build` (or __graft_entry__.build()); "
                      "there is no CPU fallback" % LIB_PATH)
    if "TD_LIB_PATH" not in os.environ:
        try:   # a library older than its sources measures / tests the wrong code: say so (the GPU box only has the prebuilt file)
            from . import build as _b
            if _b.needs_build():
                import sys
                sys.stderr.write("tagdust_amd: %s is older than its sources -- run `python -m tagdust_amd.build`\n" % LIB_PATH)
        except Exception:
            pass
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # read by the HIP runtime at its first call (td_want_hw_queues, td_api.hip)
    lib = C.CDLL(LIB_PATH)
    lib.td_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    lib.td_ctx_destroy.argtypes = [C.c_void_p]
    lib.td_ctx_destroy.restype = None
    lib.td_last_error.argtypes = [C.c_void_p]
    lib.td_last_error.restype = C.c_char_p
    lib.td_logsum_table.restype = C.POINTER(C.c_float)
    lib.td_model_upload.argtypes = [C.c_void_p, C.POINTER(_ModelDesc)]
    lib.td_set_params.argtypes = [C.c_void_p, C.c_float, C.c_int32, C.c_int32]
    lib.td_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int32]
    if hasattr(lib, "td_set_artifacts"):
        lib.td_set_artifacts.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
    if hasattr(lib, "td_get_option"):   # absent from older builds loaded through TD_LIB_PATH for A/B runs
        lib.td_get_option.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int32)]
    lib.td_spec_source.argtypes = [C.POINTER(_ModelDesc), C.c_char_p, C.c_int64]
    lib.td_spec_source.restype = C.c_int64
    lib.td_spec_prune_info.argtypes = [C.POINTER(_ModelDesc), C.c_int32, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.td_batch_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
    lib.td_batch_upload_ascii.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
    lib.td_run.argtypes = [C.c_void_p, C.c_int]
    lib.td_sync.argtypes = [C.c_void_p]
    lib.td_batch_download.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.td_submit.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                              C.POINTER(C.c_int64)]
    lib.td_wait.argtypes = [C.c_void_p, C.c_int64]
    lib.td_host_alloc.argtypes = [C.c_size_t]
    lib.td_host_alloc.restype = C.c_void_p
    lib.td_host_free.argtypes = [C.c_void_p]
    lib.td_host_free.restype = None
    lib.td_set_batch_window.argtypes = [C.c_void_p, C.c_int64, C.c_int64]
    lib.td_set_window.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    lib.td_arch_scores.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    lib.td_shard_bounds.argtypes = [C.c_int64, C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.td_shard_bounds.restype = None
    lib.td_count_outcomes.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    lib.td_count_outcomes.restype = None
    lib.td_multi_create.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]
    lib.td_multi_destroy.argtypes = [C.c_void_p]
    lib.td_multi_destroy.restype = None
    lib.td_multi_last_error.argtypes = [C.c_void_p]
    lib.td_multi_last_error.restype = C.c_char_p
    lib.td_multi_size.argtypes = [C.c_void_p]
    lib.td_multi_ctx.argtypes = [C.c_void_p, C.c_int32]
    lib.td_multi_ctx.restype = C.c_void_p
    lib.td_multi_model_upload.argtypes = [C.c_void_p, C.POINTER(_ModelDesc)]
    lib.td_multi_set_params.argtypes = [C.c_void_p, C.c_float, C.c_int32, C.c_int32]
    lib.td_multi_set_window.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    lib.td_multi_set_artifacts.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
    lib.td_multi_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.td_multi_counts.argtypes = [C.c_void_p, C.c_void_p]
    lib.td_multi_counts_reset.argtypes = [C.c_void_p]
    lib.td_multi_uses_rccl.argtypes = [C.c_void_p]
    if hasattr(lib, "td_bind_host_to_device"):
        lib.td_bind_host_to_device.argtypes = [C.c_int32]
        lib.td_bind_host_to_device.restype = C.c_int32
    lib.td_counts_reset.argtypes = [C.c_void_p]
    lib.td_counts_get.argtypes = [C.c_void_p, C.c_void_p]
    if hasattr(lib, "td_diag_get"):
        lib.td_diag_get.argtypes = [C.c_void_p, C.c_void_p]
    lib.td_counts_device_ptr.argtypes = [C.c_void_p]
    lib.td_counts_device_ptr.restype = C.c_void_p
    lib.td_last_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    if hasattr(lib, "td_timeline_origin"):
        lib.td_timeline_origin.argtypes = [C.c_void_p]
        lib.td_last_kernel_times.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int32)]
    lib.td_batch_info.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
    lib.td_arch_parse.argtypes = [C.POINTER(C.c_char_p), C.c_int32, C.POINTER(C.c_void_p)]
    lib.td_arch_free.argtypes = [C.c_void_p]
    lib.td_arch_free.restype = None
    lib.td_sequence_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(_SeqStats)]
    lib.td_sequence_stats_window.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.POINTER(_SeqStats)]
    lib.td_model_build.argtypes = [C.c_void_p, C.POINTER(_SeqStats), C.c_float, C.c_float, C.POINTER(C.c_void_p)]
    lib.td_model_tables_free.argtypes = [C.c_void_p]
    lib.td_model_tables_free.restype = None
    lib.td_calibration_emit.argtypes = [C.c_void_p, C.POINTER(_SeqStats), C.c_float, C.c_uint32, C.c_int32, C.c_int32,
                                        C.POINTER(C.POINTER(_Calibration))]
    lib.td_calibration_select.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    lib.td_calibration_select.restype = C.c_float
    lib.td_calibration_free.argtypes = [C.POINTER(_Calibration)]
    lib.td_calibration_free.restype = None
    lib.td_estimate_threshold.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(_SeqStats), C.c_float, C.c_uint32, C.c_int32, C.c_int32,
                                          C.POINTER(C.c_float)]
    lib.td_compare_architectures.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int32, C.c_void_p, C.c_void_p, C.c_int64,
                                             C.c_float, C.c_float, C.c_int32, C.c_void_p, C.POINTER(C.c_int32)]
    lib.td_reads_parse.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.POINTER(C.POINTER(_Reads))]
    lib.td_reads_free.argtypes = [C.POINTER(_Reads)]
    lib.td_reads_free.restype = None
    lib.td_writer_open.argtypes = [C.c_char_p, C.c_void_p, C.POINTER(C.c_void_p)]
    lib.td_writer_write.argtypes = [C.c_void_p, C.POINTER(_Reads), C.c_void_p, C.c_void_p]
    lib.td_writer_close.argtypes = [C.c_void_p]
    _lib = lib
    return lib


def make_model_desc(md):
    """Build the C td_model_desc from a mapping of numpy tables; returns (desc, arrays to keep alive)."""
    S, H, Cc = int(md["S"]), int(md["H"]), int(md["C"])
    a = {
        "n_hmm": np.ascontiguousarray(md["n_hmm"], np.int32), "n_col": np.ascontiguousarray(md["n_col"], np.int32),
        "skip": np.ascontiguousarray(md["skip"], np.float32),
        "seg_type": np.ascontiguousarray(md["seg_type"], np.int32).astype(np.int8),
        "finger_len": np.where(np.asarray(md["seg_type"]) == ord("F"), np.asarray(md["seg_len"]), 0).astype(np.int32),
        "trans": np.ascontiguousarray(md["trans"], np.float32).reshape(Cc, 9),
        "eM": np.ascontiguousarray(md["eM"], np.float32).reshape(Cc, 5),
        "eI": np.ascontiguousarray(md["eI"], np.float32).reshape(Cc, 5),
        "sM": np.ascontiguousarray(md["sM"], np.float32).reshape(Cc),
        "sI": np.ascontiguousarray(md["sI"], np.float32).reshape(Cc),
        "label": np.ascontiguousarray(md["label"], np.int32).reshape(H),
        "A": np.ascontiguousarray(md["A"], np.float32).reshape(H, H),
    }
    d = _ModelDesc()
    d.S, d.H, d.C, d.avg_len = S, H, Cc, int(md["avg_len"])
    for i in range(5):
        d.bg[i] = float(np.float32(md["bg"][i]))
    for k, v in a.items():
        setattr(d, k, v.ctypes.data)
    return d, a


def build_model(segments, codes, offs, e=0.05, d=0.1, stats_override=None, window=None):
    """Host model construction through the C library (include/tagdust_model.h; no GPU needed):
    td_arch_parse -> td_sequence_stats -> td_model_build.  segments: ["B:ACGT,...", "R:N", ...].
    Returns (model mapping with the golden-fixture keys, stats dict)."""
    lib = load_library()
    arr = (C.c_char_p * len(segments))(*[s.encode() for s in segments])
    arch = C.c_void_p()
    if lib.td_arch_parse(arr, len(segments), C.byref(arch)) != 0:
        raise TdError("td_arch_parse failed for %r" % (segments,))
    try:
        codes = np.ascontiguousarray(codes, np.uint8)
        offs = np.ascontiguousarray(offs, np.int64)
        st = _SeqStats()
        ms, me = window if window else (-1, -1)
        if lib.td_sequence_stats_window(arch, codes.ctypes.data, offs.ctypes.data, len(offs) - 1, int(ms), int(me), C.byref(st)) != 0:
            raise TdError("td_sequence_stats failed")
        if stats_override:
            for k, v in stats_override.items():
                setattr(st, k, v)
        tab = C.c_void_p()
        if lib.td_model_build(arch, C.byref(st), float(e), float(d), C.byref(tab)) != 0:
            raise TdError("td_model_build failed")
        try:
            desc = C.cast(tab, C.POINTER(_ModelDesc)).contents
            S, H, Cc = desc.S, desc.H, desc.C

            def arr_of(ptr, ctype, n, dtype):
                return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(n,)).astype(dtype).copy()
            md = {
                "S": S, "H": H, "C": Cc, "avg_len": desc.avg_len, "bg": np.array(list(desc.bg), np.float32),
                "n_hmm": arr_of(desc.n_hmm, C.c_int32, S, np.int32), "n_col": arr_of(desc.n_col, C.c_int32, S, np.int32),
                "skip": arr_of(desc.skip, C.c_float, S, np.float32),
                "seg_type": arr_of(desc.seg_type, C.c_int8, S, np.int32),
                "trans": arr_of(desc.trans, C.c_float, Cc * 9, np.float32).reshape(Cc, 9),
                "eM": arr_of(desc.eM, C.c_float, Cc * 5, np.float32).reshape(Cc, 5),
                "eI": arr_of(desc.eI, C.c_float, Cc * 5, np.float32).reshape(Cc, 5),
                "sM": arr_of(desc.sM, C.c_float, Cc, np.float32), "sI": arr_of(desc.sI, C.c_float, Cc, np.float32),
                "label": arr_of(desc.label, C.c_int32, H, np.int32),
                "A": arr_of(desc.A, C.c_float, H * H, np.float32).reshape(H, H),
            }
            fl = arr_of(desc.finger_len, C.c_int32, S, np.int32)
            md["seg_len"] = np.where(md["seg_type"] == ord("F"), fl, md["n_col"]).astype(np.int32)
            stats = {k: (list(getattr(st, k)) if k == "background" else getattr(st, k)) for k, _ in _SeqStats._fields_}
            return md, stats
        finally:
            lib.td_model_tables_free(tab)
    finally:
        lib.td_arch_free(arch)


def _arch_and_stats(lib, segments, codes, offs):
    arr = (C.c_char_p * len(segments))(*[s.encode() for s in segments])
    arch = C.c_void_p()
    if lib.td_arch_parse(arr, len(segments), C.byref(arch)) != 0:
        raise TdError("td_arch_parse failed for %r" % (segments,))
    codes = np.ascontiguousarray(codes, np.uint8)
    offs = np.ascontiguousarray(offs, np.int64)
    st = _SeqStats()
    if lib.td_sequence_stats(arch, codes.ctypes.data, offs.ctypes.data, len(offs) - 1, C.byref(st)) != 0:
        lib.td_arch_free(arch)
        raise TdError("td_sequence_stats failed")
    return arch, st


def calibration_emit(segments, codes, offs, d=0.1, seed=42, n_reads=400000, rng=0):
    """The simulated reads of the reference's threshold calibration (calibrateQ.c:88-113) for this architecture and
    these sequence statistics: returns (codes, offs, is_random)."""
    lib = load_library()
    arch, st = _arch_and_stats(lib, segments, codes, offs)
    try:
        p = C.POINTER(_Calibration)()
        if lib.td_calibration_emit(arch, C.byref(st), float(d), int(seed), int(n_reads), int(rng), C.byref(p)) != 0:
            raise TdError("td_calibration_emit failed")
        c = p.contents
        n = int(c.n_reads)
        o = np.ctypeslib.as_array(c.offs, shape=(n + 1,)).copy()
        out = (np.ctypeslib.as_array(c.codes, shape=(int(o[-1]),)).copy(), o, np.ctypeslib.as_array(c.is_random, shape=(n,)).copy())
        lib.td_calibration_free(p)
        return out
    finally:
        lib.td_arch_free(arch)


def calibration_select(mapq, is_random):
    lib = load_library()
    q = np.ascontiguousarray(mapq, np.float32)
    r = np.ascontiguousarray(is_random, np.uint8)
    return float(np.float32(lib.td_calibration_select(q.ctypes.data, r.ctypes.data, len(q))))


def estimate_threshold(ctx, segments, codes, offs, d=0.1, seed=42, n_reads=400000, rng=0):
    """estimateQthreshold() with the scoring on the GPU (TD_MODE_GET_PROB)."""
    lib = load_library()
    arch, st = _arch_and_stats(lib, segments, codes, offs)
    try:
        thr = C.c_float()
        if lib.td_estimate_threshold(ctx.h, arch, C.byref(st), float(d), int(seed), int(n_reads), int(rng), C.byref(thr)) != 0:
            raise TdError(lib.td_last_error(ctx.h).decode() or "td_estimate_threshold failed")
        return float(np.float32(thr.value))
    finally:
        lib.td_arch_free(arch)


def compare_architectures(ctx, candidates, codes, offs, e=0.05, d=0.1, n_threads=8):
    """test_architectures(): candidates = list of segment-string lists; returns (posteriors float32[n], best index)."""
    lib = load_library()
    handles = []
    try:
        for segs in candidates:
            arr = (C.c_char_p * len(segs))(*[s.encode() for s in segs])
            h = C.c_void_p()
            if lib.td_arch_parse(arr, len(segs), C.byref(h)) != 0:
                raise TdError("td_arch_parse failed for %r" % (segs,))
            handles.append(h)
        arr = (C.c_void_p * len(handles))(*[h.value for h in handles])
        codes = np.ascontiguousarray(codes, np.uint8)
        offs = np.ascontiguousarray(offs, np.int64)
        post = np.zeros(len(handles), np.float32)
        best = C.c_int32(-1)
        if lib.td_compare_architectures(ctx.h, arr, len(handles), codes.ctypes.data, offs.ctypes.data, len(offs) - 1,
                                        float(e), float(d), int(n_threads), post.ctypes.data, C.byref(best)) != 0:
            raise TdError(lib.td_last_error(ctx.h).decode() or "td_compare_architectures failed")
        return post, int(best.value)
    finally:
        for h in handles:
            lib.td_arch_free(h)


class _Fasta(C.Structure):
    _fields_ = [("n_seq", C.c_int32), ("string", C.POINTER(C.c_uint8)), ("s_index", C.POINTER(C.c_int32)),
                ("names", C.POINTER(C.c_char_p))]


def parse_fasta(text):
    """-ref artifact sequences as read_fasta() (src/io.c:1912-2001) stores them: (string uint8, s_index int32, names)."""
    lib = load_library()
    buf = np.frombuffer(bytes(text), dtype=np.uint8)
    p = C.POINTER(_Fasta)()
    lib.td_fasta_parse.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
    lib.td_fasta_free.argtypes = [C.c_void_p]
    if lib.td_fasta_parse(buf.ctypes.data, len(buf), C.byref(p)) != 0:
        raise TdError("td_fasta_parse failed")
    try:
        f = p.contents
        n = int(f.n_seq)
        ix = np.ctypeslib.as_array(f.s_index, shape=(n + 1,)).copy()
        st = np.ctypeslib.as_array(f.string, shape=(max(int(ix[-1]), 1),))[:int(ix[-1])].copy()
        names = [f.names[j] for j in range(n)]
    finally:
        lib.td_fasta_free(p)
    return st, ix, names


class ParsedReads:
    """FASTQ/FASTA text parsed by the library (include/tagdust_io.h); keeps the text buffer alive."""

    def __init__(self, text, n_threads=0):
        lib = load_library()
        self._buf = np.frombuffer(text, dtype=np.uint8) if isinstance(text, (bytes, bytearray)) else np.ascontiguousarray(text, np.uint8)
        self._p = C.POINTER(_Reads)()
        if lib.td_reads_parse(self._buf.ctypes.data, len(self._buf), int(n_threads), C.byref(self._p)) != 0:
            self._p = None
            lib.td_io_last_error.restype = C.c_char_p
            raise TdError(lib.td_io_last_error().decode() or "td_reads_parse failed")
        r = self._p.contents
        self.n = int(r.n_reads)
        self.offs = np.ctypeslib.as_array(r.offs, shape=(self.n + 1,)).copy()
        self.codes = np.ctypeslib.as_array(r.codes, shape=(max(int(self.offs[-1]), 1),))[:int(self.offs[-1])].copy()
        self.name_off = np.ctypeslib.as_array(r.name_off, shape=(max(self.n, 1),))[:self.n].copy()
        self.name_len = np.ctypeslib.as_array(r.name_len, shape=(max(self.n, 1),))[:self.n].copy()
        self.qual_off = np.ctypeslib.as_array(r.qual_off, shape=(max(self.n, 1),))[:self.n].copy()

    def names(self):
        b = self._buf.tobytes()
        return [b[o:o + l] for o, l in zip(self.name_off, self.name_len)]

    def close(self):
        if self._p:
            load_library().td_reads_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def write_demultiplexed(prefix, segments, reads, res, seq_out):
    """print_all() for one input file through the library: res (RESULT_DTYPE) and seq_out as td_batch_download gives them."""
    lib = load_library()
    arr = (C.c_char_p * len(segments))(*[s.encode() for s in segments])
    arch = C.c_void_p()
    if lib.td_arch_parse(arr, len(segments), C.byref(arch)) != 0:
        raise TdError("td_arch_parse failed")
    try:
        w = C.c_void_p()
        if lib.td_writer_open(prefix.encode(), arch, C.byref(w)) != 0:
            raise TdError("td_writer_open failed for %s" % prefix)
        res = np.ascontiguousarray(res, RESULT_DTYPE)
        seq_out = np.ascontiguousarray(seq_out, np.uint8)
        rc = lib.td_writer_write(w, reads._p, res.ctypes.data, seq_out.ctypes.data)
        rc |= lib.td_writer_close(w)
        if rc != 0:
            raise TdError("td_writer_write/close failed")
    finally:
        lib.td_arch_free(arch)


class _StreamOpts(C.Structure):
    _fields_ = [("batch_reads", C.c_int32), ("n_threads", C.c_int32), ("block_bytes", C.c_int64)]


class _StreamStats(C.Structure):
    _fields_ = [("n_reads", C.c_int64), ("n_batches", C.c_int64), ("bytes_in", C.c_int64), ("bytes_out", C.c_int64),
                ("wall_s", C.c_double), ("read_s", C.c_double), ("parse_s", C.c_double), ("decode_s", C.c_double),
                ("write_s", C.c_double), ("codes_fnv", C.c_uint64)]


def stream_run(ctx, in_path, segments=None, out_prefix=None, batch_reads=0, n_threads=0, block_bytes=0):
    """td_stream_run: one input file through parse -> decode -> write as a pipeline; ctx None = parse-only run (no GPU).
    Returns the statistics as a dict."""
    lib = load_library()
    lib.td_stream_run.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_char_p, C.POINTER(_StreamOpts), C.POINTER(_StreamStats)]
    lib.td_io_last_error.restype = C.c_char_p
    arch = C.c_void_p()
    if segments is not None:
        arr = (C.c_char_p * len(segments))(*[s.encode() for s in segments])
        if lib.td_arch_parse(arr, len(segments), C.byref(arch)) != 0:
            raise TdError("td_arch_parse failed for %r" % (segments,))
    try:
        o = _StreamOpts(int(batch_reads), int(n_threads), int(block_bytes))
        st = _StreamStats()
        rc = lib.td_stream_run(ctx.h if ctx is not None else None, os.fsencode(in_path), arch if segments is not None else None,
                               os.fsencode(out_prefix) if out_prefix is not None else None, C.byref(o), C.byref(st))
        if rc != 0:
            raise TdError(lib.td_io_last_error().decode() or "td_stream_run failed")
        return {k: getattr(st, k) for k, _ in _StreamStats._fields_}
    finally:
        if arch:
            lib.td_arch_free(arch)


class _StreamFile(C.Structure):
    _fields_ = [("path", C.c_char_p), ("arch", C.c_void_p), ("ctx", C.POINTER(C.c_void_p))]


def stream_run_multi(files, out_prefix, n_devices=1, dust=100, batch_reads=0, n_threads=0, block_bytes=0):
    """td_stream_run_multi: the input files of one paired / multi-read run in lock-step.  files: a list of
    (path, segments, contexts) -- contexts = one TagdustHip per device holding that file's model and parameters, or None for a
    file whose architecture is a single read segment (not decoded: run_rna_dust).  Returns (statistics dict, counters int64[264])."""
    lib = load_library()
    lib.td_stream_run_multi.argtypes = [C.POINTER(_StreamFile), C.c_int32, C.c_int32, C.c_char_p, C.c_int32, C.POINTER(_StreamOpts),
                                        C.POINTER(_StreamStats), C.c_void_p]
    lib.td_io_last_error.restype = C.c_char_p
    arr = (_StreamFile * len(files))()
    archs, keep = [], []
    try:
        for k, (path, segments, ctxs) in enumerate(files):
            a = C.c_void_p()
            sa = (C.c_char_p * len(segments))(*[s.encode() for s in segments])
            if lib.td_arch_parse(sa, len(segments), C.byref(a)) != 0:
                raise TdError("td_arch_parse failed for %r" % (segments,))
            archs.append(a)
            arr[k].path = os.fsencode(path)
            arr[k].arch = a
            if ctxs is not None:
                if len(ctxs) != n_devices:
                    raise TdError("stream_run_multi: %d contexts for %d devices" % (len(ctxs), n_devices))
                cp = (C.c_void_p * n_devices)(*[c.h for c in ctxs])
                keep.append(cp)
                arr[k].ctx = C.cast(cp, C.POINTER(C.c_void_p))
        o = _StreamOpts(int(batch_reads), int(n_threads), int(block_bytes))
        st = _StreamStats()
        counts = np.zeros(264, np.int64)
        rc = lib.td_stream_run_multi(arr, len(files), int(n_devices), os.fsencode(out_prefix), int(dust), C.byref(o), C.byref(st), counts.ctypes.data)
        if rc != 0:
            raise TdError(lib.td_io_last_error().decode() or "td_stream_run_multi failed")
        return {k: getattr(st, k) for k, _ in _StreamStats._fields_}, counts
    finally:
        for a in archs:
            if a:
                lib.td_arch_free(a)


def stream_release():
    """td_stream_release: free the page-locked batch buffers td_stream_run keeps for the next run of the process (at most 1 GiB;
    the library also frees them when the last context is destroyed)."""
    lib = load_library()
    lib.td_stream_release.restype = None
    lib.td_stream_release()


def spec_source(md):
    """The HIP source of the model-specialised kernel for this model (no GPU needed)."""
    lib = load_library()
    d, keep = make_model_desc(md)
    n = lib.td_spec_source(C.byref(d), None, 0)
    buf = C.create_string_buffer(int(n) + 1)
    lib.td_spec_source(C.byref(d), buf, int(n) + 1)
    return buf.value.decode()


def spec_prune_info(md, lcap):
    """Position-pruning tables of the specialised kernel for this model: dict with n_seg (leading segments pruned, 0 = none),
    sfx_first (first trailing segment pruned, S = none), z, and the tables fb, bwb, wa, wb, fbs, bws, wc, wd."""
    lib = load_library()
    d, keep = make_model_desc(md)
    tab = np.zeros(8 * (lcap + 8), np.float32)
    z, ns, sf = C.c_float(0), C.c_int32(0), C.c_int32(0)
    if lib.td_spec_prune_info(C.byref(d), int(lcap), tab.ctypes.data, C.byref(z), C.byref(ns), C.byref(sf)) != 0:
        raise RuntimeError("td_spec_prune_info failed")
    t = tab.reshape(8, lcap + 8)
    out = dict(n_seg=int(ns.value), sfx_first=int(sf.value), S=int(md["S"]), z=float(z.value))
    for k, name in enumerate(("fb", "bwb", "wa", "wb", "fbs", "bws", "wc", "wd")):
        out[name] = t[k]
    return out


def spec_restart_info(md, lcap):
    """Impulse-response tables of the restarted sweeps for this model: (gq [4, lcap + 8] leading segments, gf [4, lcap + 8]
    trailing segments, restart flag)."""
    lib = load_library()
    d, keep = make_model_desc(md)
    tab = np.zeros(8 * (lcap + 8), np.float32)
    r = C.c_int32(0)
    lib.td_spec_restart_info.argtypes = [C.POINTER(_ModelDesc), C.c_int32, C.c_void_p, C.POINTER(C.c_int32)]
    if lib.td_spec_restart_info(C.byref(d), int(lcap), tab.ctypes.data, C.byref(r)) != 0:
        raise RuntimeError("td_spec_restart_info failed")
    t = tab.reshape(8, lcap + 8)
    return t[:4], t[4:], int(r.value)


class TagdustHip:
    """One context per GPU (the analogue of run_pHMM's per-thread model copies)."""

    def __init__(self, device=0):
        self.lib = load_library()
        h = C.c_void_p()
        if self.lib.td_ctx_create(int(device), C.byref(h)) != 0:
            raise TdError(self.lib.td_last_error(None).decode())
        self.h = h
        self._keep = None
        self.offs = None

    def close(self):
        if getattr(self, "h", None):
            self.lib.td_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise TdError(self.lib.td_last_error(self.h).decode())

    def set_option(self, name, value):
        self._chk(self.lib.td_set_option(self.h, name.encode(), int(value)))

    def set_artifacts(self, string=None, s_index=None, filter_error=2, n_threads=1):
        """-ref artifact filter (struct fasta's string / s_index); None switches it off."""
        if string is None:
            self._chk(self.lib.td_set_artifacts(self.h, None, None, 0, 0, 1))
            return
        a = np.ascontiguousarray(string, dtype=np.uint8)
        ix = np.ascontiguousarray(s_index, dtype=np.int32)
        self._chk(self.lib.td_set_artifacts(self.h, a.ctypes.data, ix.ctypes.data, len(ix) - 1, int(filter_error), int(n_threads)))

    def get_option(self, name):
        v = C.c_int32(0)
        self._chk(self.lib.td_get_option(self.h, name.encode(), C.byref(v)))
        return int(v.value)

    def upload_model(self, md):
        """md: mapping with S,H,C,avg_len,bg,n_hmm,n_col,skip,seg_type,seg_len,trans,eM,eI,sM,sI,label,A
        (the tables of struct model_bag; same keys as the golden fixtures)."""
        d, keep = make_model_desc(md)
        self._keep = keep
        self._chk(self.lib.td_model_upload(self.h, C.byref(d)))

    def set_params(self, threshold, minlen=16, dust=100):
        self._chk(self.lib.td_set_params(self.h, float(threshold), int(minlen), int(dust)))

    def set_window(self, matchstart=-1, matchend=-1):
        self._chk(self.lib.td_set_window(self.h, int(matchstart), int(matchend)))

    def upload_batch(self, codes, offs):
        codes = np.ascontiguousarray(codes, np.uint8)
        self.offs = np.ascontiguousarray(offs, np.int64)
        self._chk(self.lib.td_batch_upload(self.h, codes.ctypes.data, self.offs.ctypes.data, len(self.offs) - 1))

    def upload_batch_ascii(self, bases, offs):
        buf = np.frombuffer(bases, dtype=np.uint8) if isinstance(bases, (bytes, bytearray)) else np.ascontiguousarray(bases, np.uint8)
        self.offs = np.ascontiguousarray(offs, np.int64)
        self._chk(self.lib.td_batch_upload_ascii(self.h, buf.ctypes.data, self.offs.ctypes.data, len(self.offs) - 1))

    def run(self, mode=MODE_GET_LABEL):
        self._chk(self.lib.td_run(self.h, int(mode)))

    def sync(self):
        self._chk(self.lib.td_sync(self.h))

    def download(self, labels=True, seq=True):
        n = len(self.offs) - 1
        res = np.zeros(n, RESULT_DTYPE)
        lab = np.zeros(int(self.offs[-1]) + n, np.int8) if labels else None
        sq = np.zeros(int(self.offs[-1]), np.uint8) if seq else None
        self._chk(self.lib.td_batch_download(self.h, res.ctypes.data, lab.ctypes.data if labels else None,
                                             sq.ctypes.data if seq else None))
        return res, lab, sq

    def submit(self, bases, offs, mode=MODE_GET_LABEL, res=None, labels=None, seq_out=None, ascii=False):
        """td_submit: numpy arrays in (kept alive by the caller until wait()), result arrays named now; returns the ticket."""
        n = len(offs) - 1
        t = C.c_int64(0)
        self._chk(self.lib.td_submit(self.h, bases.ctypes.data, 1 if ascii else 0, offs.ctypes.data, n, int(mode),
                                     res.ctypes.data if res is not None else None,
                                     labels.ctypes.data if labels is not None else None,
                                     seq_out.ctypes.data if seq_out is not None else None, C.byref(t)))
        return int(t.value)

    def wait(self, ticket):
        self._chk(self.lib.td_wait(self.h, int(ticket)))

    def arch_scores(self, models, codes, offs):
        """td_arch_scores: backward score of every read under every candidate model, one launch; float32 [n_models, n]."""
        descs, keep = [], []
        for md in models:
            d, k = make_model_desc(md)
            descs.append(d)
            keep.append(k)
        ptrs = (C.c_void_p * len(descs))(*[C.addressof(d) for d in descs])
        codes = np.ascontiguousarray(codes, np.uint8)
        offs = np.ascontiguousarray(offs, np.int64)
        n = len(offs) - 1
        out = np.zeros((len(descs), n), np.float32)
        self._chk(self.lib.td_arch_scores(self.h, ptrs, len(descs), codes.ctypes.data, offs.ctypes.data, n, out.ctypes.data))
        return out

    def counts_reset(self):
        self._chk(self.lib.td_counts_reset(self.h))

    def counts(self):
        c = np.zeros(NUM_COUNTERS, np.int64)
        self._chk(self.lib.td_counts_get(self.h, c.ctypes.data))
        return c

    def diag(self):
        """td_diag_get: the 64 diagnostic words of the development knobs (word k = historical slot 192 + k)."""
        d = np.zeros(NUM_DIAG_COUNTERS, np.int64)
        self._chk(self.lib.td_diag_get(self.h, d.ctypes.data))
        return d

    def last_kernel_ms(self):
        ms = C.c_float()
        self._chk(self.lib.td_last_kernel_ms(self.h, C.byref(ms)))
        return float(ms.value)

    def timeline_origin(self):
        self._chk(self.lib.td_timeline_origin(self.h))

    def last_kernel_times(self):
        """(start_ms, stop_ms, stream index) of the last batch's decode kernel, from the origin set by timeline_origin()."""
        a, b, k = C.c_float(), C.c_float(), C.c_int32()
        self._chk(self.lib.td_last_kernel_times(self.h, C.byref(a), C.byref(b), C.byref(k)))
        return float(a.value), float(b.value), int(k.value)

    def batch_info(self):
        n, w, s = C.c_int64(), C.c_int64(), C.c_int32()
        self._chk(self.lib.td_batch_info(self.h, C.byref(n), C.byref(w), C.byref(s)))
        return int(n.value), int(w.value), int(s.value)


class PinnedArray:
    """A numpy view of page-locked host memory from td_host_alloc (batch inputs / outputs the DMA engines use directly)."""

    def __init__(self, shape, dtype):
        self.lib = load_library()
        dt = np.dtype(dtype)
        n = int(np.prod(shape))
        self.ptr = self.lib.td_host_alloc(max(n * dt.itemsize, 1))
        if not self.ptr:
            raise TdError("td_host_alloc(%d bytes) failed" % (n * dt.itemsize))
        buf = (C.c_uint8 * (n * dt.itemsize)).from_address(self.ptr)
        self.array = np.frombuffer(buf, dtype=dt, count=n).reshape(shape)

    def free(self):
        if getattr(self, "ptr", None):
            self.array = None
            self.lib.td_host_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def bind_host_to_device(device):
    """td_bind_host_to_device: pin the calling thread to the CPUs of the NUMA node next to `device`; returns the node or -1."""
    return int(load_library().td_bind_host_to_device(int(device)))


def shard_bounds(n_reads, world, rank):
    """td_shard_bounds: run_pHMM's contiguous split (barcode_hmm.c:1911-1922); no GPU needed."""
    lib = load_library()
    lo, hi = C.c_int64(), C.c_int64()
    lib.td_shard_bounds(int(n_reads), int(world), int(rank), C.byref(lo), C.byref(hi))
    return int(lo.value), int(hi.value)


def count_outcomes(res, lens=None):
    """td_count_outcomes: the device counters restated on the host from per-read results; no GPU needed."""
    lib = load_library()
    res = np.ascontiguousarray(res, RESULT_DTYPE)
    c = np.zeros(NUM_COUNTERS, np.int64)
    ln = None if lens is None else np.ascontiguousarray(lens, np.int32)
    lib.td_count_outcomes(res.ctypes.data, ln.ctypes.data if ln is not None else None, len(res), c.ctypes.data)
    return c


class TagdustMulti:
    """Several GPUs driven from one process (include/tagdust_multi.h)."""

    def __init__(self, devices):
        self.lib = load_library()
        d = np.ascontiguousarray(devices, np.int32)
        h = C.c_void_p()
        if self.lib.td_multi_create(d.ctypes.data, len(d), C.byref(h)) != 0:
            raise TdError(self.lib.td_multi_last_error(None).decode())
        self.h = h
        self._keep = None

    def _chk(self, rc):
        if rc != 0:
            raise TdError(self.lib.td_multi_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None):
            self.lib.td_multi_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload_model(self, md):
        d, keep = make_model_desc(md)
        self._keep = keep
        self._chk(self.lib.td_multi_model_upload(self.h, C.byref(d)))

    def set_params(self, threshold, minlen=16, dust=100):
        self._chk(self.lib.td_multi_set_params(self.h, float(threshold), int(minlen), int(dust)))

    def set_artifacts(self, string=None, s_index=None, filter_error=2, n_threads=1):
        if string is None:
            self._chk(self.lib.td_multi_set_artifacts(self.h, None, None, 0, 0, 1))
            return
        a = np.ascontiguousarray(string, np.uint8)
        ix = np.ascontiguousarray(s_index, np.int32)
        self._chk(self.lib.td_multi_set_artifacts(self.h, a.ctypes.data, ix.ctypes.data, len(ix) - 1, int(filter_error), int(n_threads)))

    def set_window(self, matchstart=-1, matchend=-1):
        self._chk(self.lib.td_multi_set_window(self.h, int(matchstart), int(matchend)))

    def decode(self, bases, offs, mode=MODE_GET_LABEL, labels=True, seq=True, ascii=False):
        bases = np.ascontiguousarray(bases, np.uint8)
        offs = np.ascontiguousarray(offs, np.int64)
        n = len(offs) - 1
        res = np.zeros(n, RESULT_DTYPE)
        lab = np.zeros(int(offs[-1]) + n, np.int8) if labels else None
        sq = np.zeros(int(offs[-1]), np.uint8) if seq else None
        self._chk(self.lib.td_multi_decode(self.h, bases.ctypes.data, 1 if ascii else 0, offs.ctypes.data, n, int(mode), res.ctypes.data,
                                           lab.ctypes.data if labels else None, sq.ctypes.data if seq else None))
        return res, lab, sq

    def counts(self):
        c = np.zeros(NUM_COUNTERS, np.int64)
        self._chk(self.lib.td_multi_counts(self.h, c.ctypes.data))
        return c

    def counts_reset(self):
        self._chk(self.lib.td_multi_counts_reset(self.h))

    def uses_rccl(self):
        return bool(self.lib.td_multi_uses_rccl(self.h))


class _SimParams(C.Structure):
    _fields_ = [("seed", C.c_uint32), ("rng", C.c_int32), ("barnum", C.c_int32), ("barlen", C.c_int32), ("readlen", C.c_int32),
                ("readlen_mod", C.c_int32), ("numseq", C.c_int32), ("end_loss", C.c_int32), ("random_frac", C.c_float),
                ("error_rate", C.c_float), ("indel_frac", C.c_float), ("seq5", C.c_char_p), ("seq3", C.c_char_p)]


def simreads(barcodes, seed=42, rng=0, barnum=0, barlen=0, readlen=0, readlen_mod=0, numseq=0, end_loss=0, random_frac=0.0,
             error_rate=0.0, indel_frac=0.0, seq5=None, seq3=None):
    """td_simreads: the FASTQ text the reference's simreads writes for these options (bytes)."""
    lib = load_library()
    p = _SimParams(int(seed), int(rng), int(barnum), int(barlen), int(readlen), int(readlen_mod), int(numseq), int(end_loss),
                   float(random_frac), float(error_rate), float(indel_frac), seq5.encode() if seq5 else None, seq3.encode() if seq3 else None)
    arr = (C.c_char_p * max(len(barcodes), 1))(*[b.encode() for b in barcodes])
    out, n = C.c_void_p(), C.c_int64()
    lib.td_simreads.argtypes = [C.POINTER(_SimParams), C.POINTER(C.c_char_p), C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
    lib.td_text_free.argtypes = [C.c_void_p]
    lib.td_text_free.restype = None
    if lib.td_simreads(C.byref(p), arr, len(barcodes), C.byref(out), C.byref(n)) != 0:
        raise TdError("td_simreads failed")
    try:
        return C.string_at(out, n.value)
    finally:
        lib.td_text_free(out)

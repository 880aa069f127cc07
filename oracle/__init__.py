"""CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package.  The product (``rust-birdnet-onnx_amd``) never
does and fails loudly when its HIP library is missing.

* ``postprocess.c`` / ``detection.c`` -- C restatement of the reference's
  top-K/sigmoid, model detection and chunking (file:line cited in each file),
  pinned by the reference's own known-answer tests.
* ``onnx_ref.py`` -- fp32/fp64 CPU interpreter of the ONNX operator subset the
  model files use (the network arithmetic lives in ONNX Runtime + an external
  .onnx file, neither of which is in /root/reference: PARITY UNPINNED for the
  network's numeric output).
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

MT_BIRDNET_V24, MT_BIRDNET_V30, MT_PERCH_V2 = 0, 1, 2


class OracleConfig(ctypes.Structure):
    _fields_ = [
        ("model_type", ctypes.c_int32),
        ("sample_rate", ctypes.c_uint32),
        ("segment_duration", ctypes.c_float),
        ("sample_count", ctypes.c_uint64),
        ("num_species", ctypes.c_uint64),
        ("has_embedding", ctypes.c_int32),
        ("embedding_dim", ctypes.c_uint64),
    ]


def build() -> str:
    """Compile liboracle.so with gcc (idempotent)."""
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("postprocess.c", "detection.c", "rangefilter.c")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        f32p = ctypes.POINTER(ctypes.c_float)
        u32p = ctypes.POINTER(ctypes.c_uint32)
        L.oracle_calculate_week.restype = ctypes.c_float
        L.oracle_calculate_week.argtypes = [ctypes.c_uint32, ctypes.c_uint32]
        L.oracle_validate_coordinates.restype = ctypes.c_int
        L.oracle_validate_coordinates.argtypes = [ctypes.c_float, ctypes.c_float]
        L.oracle_validate_date.restype = ctypes.c_int
        L.oracle_validate_date.argtypes = [ctypes.c_uint32, ctypes.c_uint32]
        L.oracle_location_scores.restype = ctypes.c_size_t
        L.oracle_location_scores.argtypes = [f32p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_float, u32p, f32p]
        L.oracle_filter_predictions.restype = ctypes.c_size_t
        L.oracle_filter_predictions.argtypes = [u32p, f32p, ctypes.c_size_t, u32p, f32p, ctypes.c_size_t, ctypes.c_float, ctypes.c_int, u32p, f32p]
        L.oracle_sigmoid.restype = ctypes.c_float
        L.oracle_sigmoid.argtypes = [ctypes.c_float]
        L.oracle_top_k.restype = ctypes.c_size_t
        L.oracle_top_k.argtypes = [f32p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int,
                                   ctypes.c_float, u32p, f32p]
        L.oracle_top_k_batch.restype = None
        L.oracle_top_k_batch.argtypes = [f32p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t,
                                         ctypes.c_int, ctypes.c_float, ctypes.c_size_t, u32p, f32p,
                                         u32p]
        L.oracle_score_entry_cmp.restype = ctypes.c_int
        L.oracle_score_entry_cmp.argtypes = [ctypes.c_float, ctypes.c_float]
        L.oracle_random_logits.restype = None
        L.oracle_random_logits.argtypes = [f32p, ctypes.c_size_t, ctypes.c_uint64]
        L.oracle_mock_embeddings.restype = None
        L.oracle_mock_embeddings.argtypes = [f32p, ctypes.c_size_t, ctypes.c_uint64]
        i64p = ctypes.POINTER(ctypes.c_int64)
        L.oracle_detect_model_type.restype = ctypes.c_int
        L.oracle_detect_model_type.argtypes = [i64p, ctypes.c_size_t, i64p,
                                               ctypes.POINTER(ctypes.c_size_t), ctypes.c_size_t,
                                               ctypes.c_int, ctypes.POINTER(OracleConfig)]
        L.oracle_chunk_plan.restype = ctypes.c_size_t
        L.oracle_chunk_plan.argtypes = [ctypes.c_size_t, ctypes.c_size_t, ctypes.c_float,
                                        ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint64), f32p,
                                        ctypes.c_size_t]
        L.oracle_chunk_fill.restype = None
        L.oracle_chunk_fill.argtypes = [f32p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_uint64,
                                        f32p]
        _LIB = L
    return _LIB


def _f32p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def sigmoid(x: float) -> float:
    return float(lib().oracle_sigmoid(ctypes.c_float(x)))


def top_k(logits, top_k: int, min_confidence=None):
    """postprocess.rs:40-87 -> list of (index, confidence) in reference order."""
    a = np.ascontiguousarray(logits, dtype=np.float32)
    n = a.shape[0]
    k = min(int(top_k), n)
    idx = np.zeros(max(k, 1), dtype=np.uint32)
    conf = np.zeros(max(k, 1), dtype=np.float32)
    tk = min(int(top_k), 2**63 - 1)
    m = lib().oracle_top_k(_f32p(a), n, tk, 0 if min_confidence is None else 1,
                           ctypes.c_float(0.0 if min_confidence is None else min_confidence),
                           idx.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), _f32p(conf))
    return [(int(idx[i]), np.float32(conf[i])) for i in range(m)]


def top_k_batch(logits2d, top_k: int, min_confidence=None):
    """Row-wise top_k over [B, N]; returns (idx[B,K], conf[B,K], count[B]) with K=min(top_k,N)."""
    a = np.ascontiguousarray(logits2d, dtype=np.float32)
    rows, n = a.shape
    k = max(min(int(top_k), n), 1)
    idx = np.zeros((rows, k), dtype=np.uint32)
    conf = np.zeros((rows, k), dtype=np.float32)
    cnt = np.zeros(rows, dtype=np.uint32)
    u32p = ctypes.POINTER(ctypes.c_uint32)
    lib().oracle_top_k_batch(_f32p(a), rows, n, min(int(top_k), 2**63 - 1),
                             0 if min_confidence is None else 1,
                             ctypes.c_float(0.0 if min_confidence is None else min_confidence), k,
                             idx.ctypes.data_as(u32p), _f32p(conf), cnt.ctypes.data_as(u32p))
    return idx, conf, cnt


def score_entry_cmp(a: float, b: float) -> int:
    return int(lib().oracle_score_entry_cmp(ctypes.c_float(a), ctypes.c_float(b)))


def random_logits(count: int, seed: int) -> np.ndarray:
    """testutil.rs:110-121."""
    out = np.zeros(count, dtype=np.float32)
    lib().oracle_random_logits(_f32p(out), count, ctypes.c_uint64(seed & (2**64 - 1)))
    return out


def mock_embeddings(dim: int, seed: int) -> np.ndarray:
    out = np.zeros(dim, dtype=np.float32)
    lib().oracle_mock_embeddings(_f32p(out), dim, ctypes.c_uint64(seed & (2**64 - 1)))
    return out


def detect_model_type(input_shape, output_shapes, override=None):
    """detection.rs:15-80. Returns OracleConfig or None on Error::ModelDetection."""
    ins = np.asarray(input_shape, dtype=np.int64)
    flat = np.asarray([d for s in output_shapes for d in s] or [0], dtype=np.int64)
    ranks = (ctypes.c_size_t * max(len(output_shapes), 1))(*[len(s) for s in output_shapes])
    cfg = OracleConfig()
    i64p = ctypes.POINTER(ctypes.c_int64)
    rc = lib().oracle_detect_model_type(ins.ctypes.data_as(i64p), len(ins),
                                        flat.ctypes.data_as(i64p), ranks, len(output_shapes),
                                        -1 if override is None else int(override),
                                        ctypes.byref(cfg))
    return cfg if rc == 0 else None


def chunk_plan(n_samples: int, segment_samples: int, overlap_secs: float, sample_rate: int):
    """birdnet-analyze.rs:707-743 -> (starts[u64], start_times[f32])."""
    n = lib().oracle_chunk_plan(n_samples, segment_samples, ctypes.c_float(overlap_secs),
                                sample_rate, None, None, 0)
    starts = np.zeros(max(n, 1), dtype=np.uint64)
    times = np.zeros(max(n, 1), dtype=np.float32)
    lib().oracle_chunk_plan(n_samples, segment_samples, ctypes.c_float(overlap_secs), sample_rate,
                            starts.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), _f32p(times), n)
    return starts[:n], times[:n]


def chunk_fill(samples: np.ndarray, segment_samples: int, start: int) -> np.ndarray:
    s = np.ascontiguousarray(samples, dtype=np.float32)
    out = np.empty(segment_samples, dtype=np.float32)
    lib().oracle_chunk_fill(_f32p(s), s.shape[0], segment_samples, ctypes.c_uint64(start),
                            _f32p(out))
    return out


# ---- range filter host logic (rangefilter.rs) ----
def calculate_week(month: int, day: int) -> float:
    return float(lib().oracle_calculate_week(month, day))


def validate_coordinates(latitude: float, longitude: float) -> int:
    """0 ok, 1 latitude, 2 longitude (rangefilter.rs:91-107)."""
    return int(lib().oracle_validate_coordinates(ctypes.c_float(latitude), ctypes.c_float(longitude)))


def validate_date(month: int, day: int) -> int:
    """0 ok, 1 month, 2 day (rangefilter.rs:118-133)."""
    return int(lib().oracle_validate_date(month, day))


def location_scores(scores: np.ndarray, n_labels: int, threshold: float):
    """rangefilter.rs:477-496 -> (indices, scores) sorted descending."""
    s = np.ascontiguousarray(scores, dtype=np.float32).reshape(-1)
    idx = np.zeros(max(len(s), 1), dtype=np.uint32)
    sc = np.zeros(max(len(s), 1), dtype=np.float32)
    m = lib().oracle_location_scores(_f32p(s), len(s), n_labels, ctypes.c_float(threshold),
                                     idx.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), _f32p(sc))
    return idx[:m], sc[:m]


def filter_predictions(pred_species, pred_conf, loc_species, loc_score, threshold: float, rerank: bool):
    """rangefilter.rs:333-386 on integer species ids -> (positions of the survivors in the input, confidences)."""
    ps = np.ascontiguousarray(pred_species, dtype=np.uint32)
    pc = np.ascontiguousarray(pred_conf, dtype=np.float32)
    ls = np.ascontiguousarray(loc_species, dtype=np.uint32)
    lc = np.ascontiguousarray(loc_score, dtype=np.float32)
    pos = np.zeros(max(len(ps), 1), dtype=np.uint32)
    conf = np.zeros(max(len(ps), 1), dtype=np.float32)
    u32p = ctypes.POINTER(ctypes.c_uint32)
    m = lib().oracle_filter_predictions(ps.ctypes.data_as(u32p), _f32p(pc), len(ps), ls.ctypes.data_as(u32p), _f32p(lc), len(ls),
                                        ctypes.c_float(threshold), 1 if rerank else 0, pos.ctypes.data_as(u32p), _f32p(conf))
    return pos[:m], conf[:m]

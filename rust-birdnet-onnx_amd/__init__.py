"""rust-birdnet-onnx_amd -- MI355X-native (gfx950 / HIP) BirdNET / Perch inference path.

Python here is a ctypes harness over the C ABI of ``libbirdnet_hip.so``
(``include/birdnet_hip.h`` = engine level, ``include/birdnet_host.h`` = the
compiled C++ mirror of the reference's ``Classifier`` API).  Nothing in this
package computes: if the shared library is missing the import fails loudly,
and if no gfx950 device is visible every load fails with ``NoDevice`` -- there
is no CPU fallback.

Names mirror the reference (tphakala/rust-birdnet-onnx ``src/lib.rs:93-108``):
``Classifier``, ``ClassifierBuilder``, ``BatchInferenceContext``,
``InferenceOptions``, ``CancellationToken``, ``ModelType``, ``ModelConfig``,
``Prediction``, ``PredictionResult``, ``Error``.
"""
from __future__ import annotations

import atexit
import ctypes as C
import weakref
import enum
import os
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np

# Several contexts in flight need hardware queues of their own: the ROCm runtime multiplexes streams onto
# GPU_MAX_HW_QUEUES (default 4) queues and the null stream takes one.  Read when the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

try:  # PyTorch, when present, must load ITS bundled HIP runtime first: the wheel's libamdhip64 and the
    # system one this library links against do not coexist in one process (torch.cuda then sees no GPU)
    import torch  # noqa: F401
except ImportError:  # the package itself needs no PyTorch
    pass

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BN_LIB") or os.path.join(_HERE, "libbirdnet_hip.so")  # BN_LIB: A/B builds of the same ABI

BN_MAX_OUTPUTS, BN_MAX_RANK, BN_NAME_LEN = 8, 6, 64
BN_CTX_DEFAULT, BN_CTX_ALL_OUTPUTS, BN_CTX_NO_GRAPH = 0, 1, 2
BN_ERR_MODEL_LOAD, BN_ERR_UNSUPPORTED_MODEL, BN_ERR_MODEL_DETECTION, BN_ERR_NO_DEVICE = 6, 7, 8, 9  # bn_status values the harness names

# every symbol include/birdnet_hip.h and include/birdnet_host.h declare
ENGINE_SYMBOLS = [
    "bn_abi_version", "bn_device_count", "bn_model_load", "bn_model_load_buffer", "bn_model_free",
    "bn_model_device", "bn_model_io_info", "bn_model_get_config", "bn_model_get_cost", "bn_detect_model_type", "bn_ctx_create", "bn_ctx_get_stats", "bn_ctx_input_device",
    "bn_ctx_destroy", "bn_ctx_max_batch", "bn_ctx_device_bytes", "bn_infer", "bn_infer_submit", "bn_infer_collect", "bn_infer_device",
    "bn_ctx_output_device", "bn_ctx_read_output", "bn_ctx_synchronize", "bn_ctx_stream", "bn_ctx_time_kernels", "bn_ctx_launch_costs",
    "bn_topk", "bn_topk_device", "bn_topk_host", "bn_step_device", "bn_step_results", "bn_plan_describe", "bn_model_survey", "bn_set_sharing_mode",
    "bn_recording_create", "bn_recording_create_async", "bn_recording_wait", "bn_recording_free", "bn_recording_samples", "bn_chunk_count", "bn_recording_windows",
    "bn_infer_windows", "bn_step_windows", "bn_ctx_step_device_rows", "bn_group_create", "bn_group_destroy", "bn_group_size",
    "bn_group_uses_rccl", "bn_group_get_stats", "bn_shard_range", "bn_group_analyze_recording", "bn_group_last_error", "bn_recording_create_resampled", "bn_resample_table", "bn_recording_read_f32", "bn_last_error",
]
HOST_SYMBOLS = [
    "bnh_classifier_build", "bnh_classifier_free", "bnh_classifier_config", "bnh_classifier_provider",
    "bnh_classifier_label_count", "bnh_classifier_label", "bnh_predict", "bnh_predict_batch",
    "bnh_create_batch_context", "bnh_create_native_batch_context", "bnh_context_read_output", "bnh_context_free", "bnh_context_max_batch_size", "bnh_context_sample_count",
    "bnh_context_input_buffer_capacity", "bnh_context_input_buffer_bytes", "bnh_context_model_type",
    "bnh_predict_batch_with_context", "bnh_predict_recording", "bnh_results_len", "bnh_result_model_type", "bnh_result_n_predictions",
    "bnh_result_species", "bnh_result_confidence", "bnh_result_index", "bnh_result_raw_scores",
    "bnh_result_embeddings", "bnh_results_free", "bnh_parse_labels", "bnh_parse_labels_format", "bnh_chunk_plan",
    "bnh_calculate_week", "bnh_validate_coordinates", "bnh_validate_date", "bnh_range_filter_build", "bnh_range_filter_free",
    "bnh_range_filter_predict", "bnh_range_filter_label", "bnh_filter_predictions",
]


class BnIoInfo(C.Structure):
    _fields_ = [("input_rank", C.c_int32), ("input_shape", C.c_int64 * BN_MAX_RANK),
                ("input_name", C.c_char * BN_NAME_LEN), ("n_outputs", C.c_int32),
                ("output_rank", C.c_int32 * BN_MAX_OUTPUTS),
                ("output_shape", (C.c_int64 * BN_MAX_RANK) * BN_MAX_OUTPUTS),
                ("output_name", (C.c_char * BN_NAME_LEN) * BN_MAX_OUTPUTS)]


class BnModelConfig(C.Structure):
    _fields_ = [("model_type", C.c_int32), ("sample_rate", C.c_uint32), ("segment_duration", C.c_float),
                ("sample_count", C.c_uint64), ("num_species", C.c_uint64), ("has_embedding", C.c_int32),
                ("embedding_dim", C.c_uint64), ("logits_output", C.c_int32), ("embedding_output", C.c_int32)]


class BnModelCost(C.Structure):
    _fields_ = [("macs_mfma", C.c_double), ("macs_valu", C.c_double), ("weight_bytes", C.c_double),
                ("activation_bytes", C.c_double), ("n_launches", C.c_int32), ("dft_gemm_macs", C.c_double), ("fft_flops", C.c_double),
                ("dft_performed_macs", C.c_double), ("dft_fft_equiv_flops", C.c_double),
                ("recompute_macs", C.c_double)]


class BnCtxStats(C.Structure):
    _fields_ = [("captures", C.c_uint64), ("instantiates", C.c_uint64), ("replays", C.c_uint64), ("eager_runs", C.c_uint64),
                ("capture_fallbacks", C.c_uint64), ("evictions", C.c_uint64), ("cached_graphs", C.c_uint64), ("last_fallback", C.c_char * 192),
                ("input_copies", C.c_uint64)]


BN_ABI_VERSION = 2  # include/birdnet_hip.h


class BnhError(C.Structure):
    _fields_ = [("kind", C.c_int32), ("index", C.c_uint64), ("expected", C.c_uint64), ("got", C.c_uint64),
                ("duration_ns", C.c_uint64), ("message", C.c_char * 512), ("latitude", C.c_float), ("longitude", C.c_float),
                ("month", C.c_uint32), ("day", C.c_uint32)]


def _load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` or `make`. "
            "This package has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    f32p, u32p, i64p = C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.POINTER(C.c_int64)
    vp, sz, i32 = C.c_void_p, C.c_size_t, C.c_int32
    sig = {
        "bn_abi_version": (i32, []),
        "bn_device_count": (i32, []),
        "bn_model_load": (i32, [C.c_char_p, i32, i32, C.POINTER(vp)]),
        "bn_model_load_buffer": (i32, [vp, sz, i32, i32, C.POINTER(vp)]),
        "bn_model_free": (None, [vp]),
        "bn_model_device": (i32, [vp]),
        "bn_model_io_info": (i32, [vp, C.POINTER(BnIoInfo)]),
        "bn_model_get_config": (i32, [vp, C.POINTER(BnModelConfig)]),
        "bn_model_get_cost": (i32, [vp, C.POINTER(BnModelCost), sz]),
        "bn_ctx_get_stats": (i32, [vp, C.POINTER(BnCtxStats), sz]),
        "bn_ctx_input_device": (i32, [vp, C.POINTER(vp), C.POINTER(sz)]),
        "bn_detect_model_type": (i32, [i64p, sz, i64p, C.POINTER(sz), sz, i32, C.POINTER(BnModelConfig)]),
        "bn_ctx_create": (i32, [vp, sz, C.c_uint32, C.POINTER(vp)]),
        "bn_ctx_destroy": (None, [vp]),
        "bn_ctx_max_batch": (sz, [vp]),
        "bn_ctx_device_bytes": (sz, [vp]),
        "bn_infer": (i32, [vp, C.POINTER(f32p), sz, f32p, f32p, C.POINTER(C.c_int32), C.c_uint64]),
        "bn_infer_submit": (i32, [vp, C.POINTER(f32p), sz, sz, i32, C.c_float, C.POINTER(C.c_uint64)]),
        "bn_infer_collect": (i32, [vp, C.c_uint64, f32p, f32p, sz, u32p, f32p, u32p, C.POINTER(C.c_int32), C.c_uint64]),
        "bn_infer_device": (i32, [vp, vp, sz, i32]),
        "bn_ctx_output_device": (i32, [vp, i32, C.POINTER(vp), C.POINTER(sz)]),
        "bn_ctx_read_output": (i32, [vp, i32, sz, f32p]),
        "bn_ctx_synchronize": (i32, [vp]),
        "bn_ctx_stream": (vp, [vp]),
        "bn_ctx_time_kernels": (sz, [vp, sz, vp, f32p, C.POINTER(C.c_double), C.POINTER(C.c_double), sz]),
        "bn_ctx_launch_costs": (sz, [vp, sz, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), sz]),
        "bn_topk": (i32, [vp, sz, sz, i32, C.c_float, sz, u32p, f32p, u32p]),
        "bn_topk_device": (i32, [i32, vp, sz, sz, sz, i32, C.c_float, sz, u32p, f32p, u32p]),
        "bn_topk_host": (i32, [i32, f32p, sz, sz, sz, i32, C.c_float, sz, u32p, f32p, u32p]),
        "bn_step_device": (i32, [vp, vp, sz, sz, i32, C.c_float, i32]),
        "bn_step_results": (i32, [vp, C.POINTER(f32p), C.POINTER(u32p), C.POINTER(f32p), C.POINTER(u32p), C.POINTER(sz)]),
        "bn_plan_describe": (sz, [C.c_char_p, i32, i32, C.c_char_p, sz, C.POINTER(i32)]),
        "bn_model_survey": (sz, [C.c_char_p, C.c_char_p, sz, C.POINTER(i32)]),
        "bn_set_sharing_mode": (None, [i32]),
        "bn_recording_create": (i32, [i32, vp, sz, i32, C.POINTER(vp)]),
        "bn_recording_create_async": (i32, [i32, vp, sz, i32, C.POINTER(vp)]),
        "bn_recording_wait": (i32, [vp]),
        "bn_recording_free": (None, [vp]),
        "bn_recording_samples": (sz, [vp]),
        "bn_chunk_count": (sz, [sz, sz]),
        "bn_recording_windows": (i32, [vp, sz, sz, sz, sz, f32p]),
        "bn_infer_windows": (i32, [vp, vp, sz, sz, sz, f32p, f32p, C.POINTER(C.c_int32), C.c_uint64]),
        "bn_step_windows": (i32, [vp, vp, sz, sz, sz, sz, i32, C.c_float, i32]),
        "bn_recording_create_resampled": (i32, [i32, vp, sz, i32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(vp)]),
        "bn_resample_table": (sz, [C.c_uint32, C.c_uint32, C.c_uint32, f32p, sz, u32p, u32p, u32p]),
        "bn_recording_read_f32": (i32, [vp, sz, sz, f32p]),
        "bn_last_error": (sz, [C.c_char_p, sz]),
        "bn_ctx_step_device_rows": (i32, [vp, C.POINTER(u32p)]),
        "bn_group_create": (i32, [C.POINTER(vp), C.POINTER(C.c_int32), i32, sz, i32, C.POINTER(vp)]),
        "bn_group_destroy": (None, [vp]),
        "bn_group_size": (i32, [vp]),
        "bn_group_uses_rccl": (i32, [vp]),
        "bn_group_get_stats": (i32, [vp, C.POINTER(BnCtxStats), sz]),
        "bn_shard_range": (None, [sz, i32, i32, C.POINTER(sz), C.POINTER(sz)]),
        "bn_group_analyze_recording": (i32, [vp, vp, sz, i32, sz, sz, i32, C.c_float, f32p, sz, u32p, f32p, u32p, C.POINTER(sz)]),
        "bn_group_last_error": (sz, [C.c_char_p, sz]),
        # host mirror
        "bnh_classifier_build": (i32, [C.c_char_p, C.c_char_p, C.POINTER(C.c_char_p), sz, i32, C.c_int64, i32,
                                       C.c_float, i32, C.POINTER(vp), C.POINTER(BnhError)]),
        "bnh_classifier_free": (None, [vp]),
        "bnh_classifier_config": (None, [vp, C.POINTER(BnModelConfig)]),
        "bnh_classifier_provider": (C.c_char_p, [vp]),
        "bnh_classifier_label_count": (sz, [vp]),
        "bnh_classifier_label": (C.c_char_p, [vp, sz]),
        "bnh_predict": (i32, [vp, f32p, sz, C.c_int64, C.POINTER(C.c_int32), C.POINTER(vp), C.POINTER(BnhError)]),
        "bnh_predict_batch": (i32, [vp, C.POINTER(f32p), C.POINTER(sz), sz, C.c_int64, C.POINTER(C.c_int32),
                                    C.POINTER(vp), C.POINTER(BnhError)]),
        "bnh_create_batch_context": (i32, [vp, sz, C.POINTER(vp), C.POINTER(BnhError)]),
        "bnh_create_native_batch_context": (i32, [vp, sz, i32, C.POINTER(vp), C.POINTER(BnhError)]),
        "bnh_context_read_output": (sz, [vp, i32, sz, f32p, sz, C.POINTER(sz), C.POINTER(BnhError)]),
        "bnh_context_free": (None, [vp]),
        "bnh_context_max_batch_size": (sz, [vp]),
        "bnh_context_sample_count": (sz, [vp]),
        "bnh_context_input_buffer_capacity": (sz, [vp]),
        "bnh_context_input_buffer_bytes": (sz, [vp]),
        "bnh_context_model_type": (i32, [vp]),
        "bnh_predict_batch_with_context": (i32, [vp, vp, C.POINTER(f32p), C.POINTER(sz), sz, C.c_int64,
                                                 C.POINTER(C.c_int32), C.POINTER(vp), C.POINTER(BnhError)]),
        "bnh_predict_recording": (i32, [vp, vp, vp, sz, i32, C.c_float, sz, sz, C.c_int64, C.POINTER(C.c_int32), C.POINTER(vp),
                                        f32p, sz, C.POINTER(BnhError)]),
        "bnh_parse_labels_format": (sz, [C.c_char_p, i32, C.c_char_p, sz, C.POINTER(BnhError)]),
        "bnh_calculate_week": (C.c_float, [C.c_uint32, C.c_uint32]),
        "bnh_validate_coordinates": (i32, [C.c_float, C.c_float, C.POINTER(BnhError)]),
        "bnh_validate_date": (i32, [C.c_uint32, C.c_uint32, C.POINTER(BnhError)]),
        "bnh_range_filter_build": (i32, [C.c_char_p, C.c_char_p, C.POINTER(C.c_char_p), sz, C.c_float, i32, C.POINTER(vp), C.POINTER(BnhError)]),
        "bnh_range_filter_free": (None, [vp]),
        "bnh_range_filter_predict": (i32, [vp, C.c_float, C.c_float, C.c_uint32, C.c_uint32, u32p, f32p, sz, C.POINTER(sz), C.POINTER(BnhError)]),
        "bnh_range_filter_label": (C.c_char_p, [vp, sz]),
        "bnh_filter_predictions": (sz, [C.POINTER(C.c_char_p), f32p, sz, C.POINTER(C.c_char_p), f32p, sz, C.c_float, i32, u32p, f32p]),
        "bnh_results_len": (sz, [vp]),
        "bnh_result_model_type": (i32, [vp, sz]),
        "bnh_result_n_predictions": (sz, [vp, sz]),
        "bnh_result_species": (C.c_char_p, [vp, sz, sz]),
        "bnh_result_confidence": (C.c_float, [vp, sz, sz]),
        "bnh_result_index": (sz, [vp, sz, sz]),
        "bnh_result_raw_scores": (sz, [vp, sz, C.POINTER(f32p)]),
        "bnh_result_embeddings": (sz, [vp, sz, C.POINTER(f32p)]),
        "bnh_results_free": (None, [vp]),
        "bnh_parse_labels": (sz, [C.c_char_p, i32, C.c_char_p, sz]),
        "bnh_chunk_plan": (sz, [sz, sz, C.c_float, C.c_uint32, C.POINTER(C.c_uint64), f32p, sz]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)  # AttributeError => the library does not export what the headers declare
        fn.restype = res
        fn.argtypes = args
    if L.bn_abi_version() != BN_ABI_VERSION:
        raise ImportError(f"{LIB_PATH} speaks ABI {L.bn_abi_version()}, this harness was written for ABI {BN_ABI_VERSION}: rebuild (make)")
    return L


lib = _load()


def last_error() -> str:
    buf = C.create_string_buffer(2048)
    lib.bn_last_error(buf, 2048)
    return buf.value.decode("utf-8", "replace")


# ------------------------------------------------------------------ boundary types
class ModelType(enum.IntEnum):
    BirdNetV24 = 0
    BirdNetV30 = 1
    PerchV2 = 2

    def sample_rate(self) -> int:
        return 48000 if self is ModelType.BirdNetV24 else 32000

    def segment_duration(self) -> float:
        return 3.0 if self is ModelType.BirdNetV24 else 5.0

    def sample_count(self) -> int:
        return 144000 if self is ModelType.BirdNetV24 else 160000

    def has_embeddings(self) -> bool:
        return self is not ModelType.BirdNetV24


@dataclass
class ModelConfig:
    model_type: ModelType
    sample_rate: int
    segment_duration: float
    sample_count: int
    num_species: int
    embedding_dim: Optional[int]

    @staticmethod
    def _from(c: BnModelConfig) -> "ModelConfig":
        return ModelConfig(ModelType(c.model_type), c.sample_rate, c.segment_duration, c.sample_count,
                           c.num_species, c.embedding_dim if c.has_embedding else None)


@dataclass
class Prediction:
    species: str
    confidence: np.float32
    index: int


@dataclass
class PredictionResult:
    model_type: ModelType
    predictions: list
    embeddings: Optional[np.ndarray]
    raw_scores: np.ndarray


class ErrorKind(enum.IntEnum):
    InputSize = 1
    BatchInputSize = 2
    ModelDetection = 3
    LabelCount = 4
    ModelPathRequired = 5
    LabelsRequired = 6
    ModelLoad = 7
    LabelLoad = 8
    LabelParse = 9
    Inference = 10
    Timeout = 11
    Cancelled = 12
    InvalidCoordinates = 13
    InvalidDate = 14
    RangeFilterInference = 15
    Other = 16


class Error(Exception):
    """reference src/error.rs: ``kind`` is the variant, ``str()`` the Display text."""

    def __init__(self, e: BnhError):
        super().__init__(e.message.decode("utf-8", "replace"))
        self.kind = ErrorKind(e.kind)
        self.index, self.expected, self.got, self.duration_ns = e.index, e.expected, e.got, e.duration_ns
        self.latitude, self.longitude, self.month, self.day = e.latitude, e.longitude, e.month, e.day


class EngineError(RuntimeError):
    def __init__(self, status: int):
        super().__init__(f"bn_status {status}: {last_error()}")
        self.status = status


class CancellationToken:
    """reference src/inference_options.rs:24-47 (an int32 flag shared with the C ABI)."""

    def __init__(self):
        self._flag = C.c_int32(0)

    def cancel(self):
        self._flag.value = 1

    def is_cancelled(self) -> bool:
        return self._flag.value != 0


class InferenceOptions:
    """reference src/inference_options.rs:73-114; timeout in seconds (float) or None."""

    def __init__(self, timeout: Optional[float] = None, cancellation_token: Optional[CancellationToken] = None):
        self.timeout = timeout
        self.cancellation_token = cancellation_token

    @staticmethod
    def with_timeout_of(seconds: float) -> "InferenceOptions":
        return InferenceOptions(timeout=seconds)

    def with_timeout(self, seconds: float) -> "InferenceOptions":
        self.timeout = seconds
        return self

    def with_cancellation_token(self, token: CancellationToken) -> "InferenceOptions":
        self.cancellation_token = token
        return self

    def needs_monitor(self) -> bool:
        return self.timeout is not None or self.cancellation_token is not None

    def _raw(self):
        t = -1 if self.timeout is None else int(round(self.timeout * 1e9))
        c = None if self.cancellation_token is None else C.byref(self.cancellation_token._flag)
        return C.c_int64(t), c


def _segments_arg(segments: Sequence[np.ndarray]):
    arrs = [np.ascontiguousarray(s, dtype=np.float32) for s in segments]
    n = len(arrs)
    f32p = C.POINTER(C.c_float)
    ptrs = (f32p * max(n, 1))(*[a.ctypes.data_as(f32p) for a in arrs])
    lens = (C.c_size_t * max(n, 1))(*[a.shape[0] for a in arrs])
    return arrs, ptrs, lens, n


def _collect(res: C.c_void_p) -> list:
    out = []
    try:
        for i in range(lib.bnh_results_len(res)):
            preds = [Prediction(lib.bnh_result_species(res, i, j).decode("utf-8"),
                                np.float32(lib.bnh_result_confidence(res, i, j)), lib.bnh_result_index(res, i, j))
                     for j in range(lib.bnh_result_n_predictions(res, i))]
            p = C.POINTER(C.c_float)()
            n = lib.bnh_result_raw_scores(res, i, C.byref(p))
            raw = np.ctypeslib.as_array(p, shape=(n,)).copy() if n else np.zeros(0, np.float32)
            n = lib.bnh_result_embeddings(res, i, C.byref(p))
            emb = np.ctypeslib.as_array(p, shape=(n,)).copy() if n else None
            out.append(PredictionResult(ModelType(lib.bnh_result_model_type(res, i)), preds, emb, raw))
    finally:
        lib.bnh_results_free(res)
    return out


class BatchInferenceContext:
    """reference src/batch_context.rs:70-165."""

    def __init__(self, handle):
        self._h = handle

    def __del__(self):
        if getattr(self, "_h", None):
            lib.bnh_context_free(self._h)
            self._h = None

    def max_batch_size(self) -> int:
        return lib.bnh_context_max_batch_size(self._h)

    def sample_count(self) -> int:
        return lib.bnh_context_sample_count(self._h)

    def input_buffer_capacity(self) -> int:
        return lib.bnh_context_input_buffer_capacity(self._h)

    def input_buffer_bytes(self) -> int:
        return lib.bnh_context_input_buffer_bytes(self._h)

    def model_type(self) -> ModelType:
        return ModelType(lib.bnh_context_model_type(self._h))

    def read_output(self, index: int, batch: int) -> np.ndarray:
        """Native extension: graph output `index` of the last batch as [batch, row_elems] (contexts made with
        create_native_batch_context(all_outputs=True) also hold the outputs the reference discards)."""
        err, row = BnhError(), C.c_size_t()
        need = lib.bnh_context_read_output(self._h, index, batch, None, 0, C.byref(row), C.byref(err))
        if err.kind:
            raise Error(err)
        out = np.zeros(need, dtype=np.float32)
        lib.bnh_context_read_output(self._h, index, batch, out.ctypes.data_as(C.POINTER(C.c_float)), need, C.byref(row), C.byref(err))
        if err.kind:
            raise Error(err)
        return out.reshape(batch, row.value)


class Classifier:
    """reference src/classifier.rs:436-867, backed by the compiled C++ mirror."""

    def __init__(self, handle):
        self._h = handle
        cfg = BnModelConfig()
        lib.bnh_classifier_config(handle, C.byref(cfg))
        self._config = ModelConfig._from(cfg)

    def __del__(self):
        if getattr(self, "_h", None):
            lib.bnh_classifier_free(self._h)
            self._h = None

    @staticmethod
    def builder() -> "ClassifierBuilder":
        return ClassifierBuilder()

    def config(self) -> ModelConfig:
        return self._config

    def labels(self) -> list:
        return [lib.bnh_classifier_label(self._h, i).decode("utf-8") for i in range(lib.bnh_classifier_label_count(self._h))]

    def requested_provider(self) -> str:
        return lib.bnh_classifier_provider(self._h).decode()

    def predict(self, segment, options: Optional[InferenceOptions] = None) -> PredictionResult:
        options = options or InferenceOptions()
        a = np.ascontiguousarray(segment, dtype=np.float32)
        t, c = options._raw()
        res, err = C.c_void_p(), BnhError()
        if lib.bnh_predict(self._h, a.ctypes.data_as(C.POINTER(C.c_float)), a.shape[0], t, c, C.byref(res), C.byref(err)):
            raise Error(err)
        return _collect(res)[0]

    def predict_batch(self, segments, options: Optional[InferenceOptions] = None) -> list:
        options = options or InferenceOptions()
        arrs, ptrs, lens, n = _segments_arg(segments)
        t, c = options._raw()
        res, err = C.c_void_p(), BnhError()
        if lib.bnh_predict_batch(self._h, ptrs, lens, n, t, c, C.byref(res), C.byref(err)):
            raise Error(err)
        return _collect(res)

    def create_batch_context(self, max_batch_size: int) -> BatchInferenceContext:
        h, err = C.c_void_p(), BnhError()
        if lib.bnh_create_batch_context(self._h, max_batch_size, C.byref(h), C.byref(err)):
            raise Error(err)
        return BatchInferenceContext(h)

    def create_native_batch_context(self, max_batch_size: int, all_outputs: bool = False) -> BatchInferenceContext:
        """Native extension: a batch context for every model family, Perch included (the reference refuses it)."""
        h, err = C.c_void_p(), BnhError()
        if lib.bnh_create_native_batch_context(self._h, max_batch_size, 1 if all_outputs else 0, C.byref(h), C.byref(err)):
            raise Error(err)
        return BatchInferenceContext(h)

    def predict_batch_with_context(self, context: BatchInferenceContext, segments,
                                   options: Optional[InferenceOptions] = None) -> list:
        options = options or InferenceOptions()
        arrs, ptrs, lens, n = _segments_arg(segments)
        t, c = options._raw()
        res, err = C.c_void_p(), BnhError()
        if lib.bnh_predict_batch_with_context(self._h, context._h, ptrs, lens, n, t, c, C.byref(res), C.byref(err)):
            raise Error(err)
        return _collect(res)


    def predict_recording(self, context: BatchInferenceContext, samples: np.ndarray, overlap_secs: float = 0.0, first_chunk: int = 0,
                          count: Optional[int] = None, options: Optional[InferenceOptions] = None) -> list:
        """The reference CLI's read_wav + chunk_audio + batch loop (src/bin/birdnet-analyze.rs:556-600, 683-743) on a mono
        int16 / float32 recording: uploaded once, windows cut on the device.  Returns [(start_time, PredictionResult)]."""
        options = options or InferenceOptions()
        a = np.ascontiguousarray(samples)
        if a.ndim != 1 or a.dtype not in (np.int16, np.float32):
            raise ValueError("mono int16 or float32 samples expected")
        t, c = options._raw()
        res, err = C.c_void_p(), BnhError()
        cap = 1 << 20
        times = np.zeros(cap, dtype=np.float32)
        if lib.bnh_predict_recording(self._h, context._h, a.ctypes.data_as(C.c_void_p), a.shape[0], 0 if a.dtype == np.int16 else 1,
                                     C.c_float(overlap_secs), first_chunk, (1 << 64) - 1 if count is None else count, t, c, C.byref(res),
                                     times.ctypes.data_as(C.POINTER(C.c_float)), cap, C.byref(err)):
            raise Error(err)
        results = _collect(res)
        return list(zip(times[:len(results)].tolist(), results))


class ClassifierBuilder:
    """reference src/classifier.rs:46-383."""

    def __init__(self):
        self._model_path = None
        self._labels_path = None
        self._labels = None
        self._model_type = -1
        self._top_k = 10
        self._min_conf = None
        self._device = 0

    def model_path(self, p: str):
        self._model_path = p
        return self

    def labels_path(self, p: str):
        self._labels_path, self._labels = p, None
        return self

    def labels(self, l: Sequence[str]):
        self._labels, self._labels_path = list(l), None
        return self

    def model_type(self, t: ModelType):
        self._model_type = int(t)
        return self

    def top_k(self, k: int):
        self._top_k = k
        return self

    def min_confidence(self, c: float):
        self._min_conf = c
        return self

    def with_rocm(self, device: int = 0):
        self._device = device
        return self

    def build(self) -> Classifier:
        h, err = C.c_void_p(), BnhError()
        labs = None
        n = 0
        if self._labels is not None:
            n = len(self._labels)
            labs = (C.c_char_p * max(n, 1))(*[s.encode("utf-8") for s in self._labels])
        top_k = self._top_k if self._top_k < 2 ** 63 else -1
        rc = lib.bnh_classifier_build(None if self._model_path is None else self._model_path.encode(),
                                      None if self._labels_path is None else self._labels_path.encode(), labs, n,
                                      self._model_type, top_k, 0 if self._min_conf is None else 1,
                                      C.c_float(self._min_conf or 0.0), self._device, C.byref(h), C.byref(err))
        if rc:
            raise Error(err)
        return Classifier(h)


# ------------------------------------------------------------------ engine-level wrappers (bench / tests)
class Model:
    """bn_model: what the Rust shim would hold in place of ort::Session."""

    def __init__(self, path: str, device: int = 0, model_type: int = -1):
        h = C.c_void_p()
        st = lib.bn_model_load(path.encode(), device, model_type, C.byref(h))
        if st:
            raise EngineError(st)
        self._h = h
        cfg = BnModelConfig()
        lib.bn_model_get_config(h, C.byref(cfg))
        self.config = cfg
        self.device = device

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:  # lib is gone at interpreter shutdown
            lib.bn_model_free(self._h)
            self._h = None

    def io_info(self) -> BnIoInfo:
        io = BnIoInfo()
        lib.bn_model_io_info(self._h, C.byref(io))
        return io

    def cost(self) -> BnModelCost:
        c = BnModelCost()
        lib.bn_model_get_cost(self._h, C.byref(c), C.sizeof(c))
        return c


class Context:
    """bn_ctx: buffers + stream for one in-flight batch."""

    def __init__(self, model: Model, max_batch: int, flags: int = BN_CTX_DEFAULT):
        h = C.c_void_p()
        st = lib.bn_ctx_create(model._h, max_batch, flags, C.byref(h))
        if st:
            raise EngineError(st)
        self._h, self.model, self.max_batch = h, model, max_batch

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:
            lib.bn_ctx_destroy(self._h)
            self._h = None

    def close(self):
        """Drain the context's stream and destroy it now (instead of whenever the object is collected).  Anything that wraps
        `stream()` -- torch.cuda.ExternalStream, events recorded on it, tensors copied on it -- must be released first; never
        `tensor.record_stream()` a context's stream: the allocator would record an event on it when the tensor is freed, possibly
        after bn_ctx_destroy has destroyed the hipStream_t (round 3's SIGSEGV)."""
        self.__del__()

    def input_device(self):
        """(device pointer, capacity in floats) of the context's own input buffer: a batch written here runs without a copy."""
        p, n = C.c_void_p(), C.c_size_t()
        st = lib.bn_ctx_input_device(self._h, C.byref(p), C.byref(n))
        if st:
            raise EngineError(st)
        return p.value, n.value

    def stats(self) -> dict:
        """bn_ctx_get_stats: captures / instantiates / replays / eager runs / capture fallbacks (must be 0) of this context."""
        st = BnCtxStats()
        r = lib.bn_ctx_get_stats(self._h, C.byref(st), C.sizeof(st))
        if r:
            raise EngineError(r)
        d = {k: int(getattr(st, k)) for k, _ in BnCtxStats._fields_ if k != "last_fallback"}
        d["last_fallback"] = st.last_fallback.decode(errors="replace")
        return d

    def infer(self, segments: np.ndarray, want_embeddings: bool = True, timeout_ns: int = 0, cancel=None):
        x = np.ascontiguousarray(segments, dtype=np.float32)
        b = x.shape[0]
        cfg = self.model.config
        f32p = C.POINTER(C.c_float)
        ptrs = (f32p * max(b, 1))(*[x[i].ctypes.data_as(f32p) for i in range(b)])
        logits = np.empty((b, self.output_device(cfg.logits_output)[1]), dtype=np.float32)
        emb = None
        if cfg.has_embedding and want_embeddings:
            emb = np.empty((b, self.output_device(cfg.embedding_output)[1]), dtype=np.float32)
        st = lib.bn_infer(self._h, ptrs, b, logits.ctypes.data_as(f32p),
                          None if emb is None else emb.ctypes.data_as(f32p), cancel, timeout_ns)
        if st:
            raise EngineError(st)
        return logits, emb

    def submit(self, segments, top_k: int = 0, min_confidence: Optional[float] = None) -> int:
        """bn_infer_submit: stage + upload + enqueue one batch of host segments ([b, S] array or a list of 1-D
        arrays); returns a ticket for collect().  At most two tickets may be outstanding per context."""
        f32p = C.POINTER(C.c_float)
        if isinstance(segments, np.ndarray):
            x = np.ascontiguousarray(segments, dtype=np.float32)
            rows = [x[i] for i in range(x.shape[0])]
        else:
            rows = [np.ascontiguousarray(r, dtype=np.float32) for r in segments]
        b = len(rows)
        ptrs = (f32p * max(b, 1))(*[r.ctypes.data_as(f32p) for r in rows])
        t = C.c_uint64(0)
        st = lib.bn_infer_submit(self._h, ptrs, b, top_k, 0 if min_confidence is None else 1, C.c_float(min_confidence or 0.0), C.byref(t))
        if st:
            raise EngineError(st)
        self._tickets = getattr(self, "_tickets", {})
        self._tickets[t.value] = (b, min(top_k, self.output_device(self.model.config.logits_output)[1]))
        return t.value

    def collect(self, ticket: int, want_embeddings: bool = True, timeout_ns: int = 0, cancel=None):
        """bn_infer_collect: (logits, embeddings or None, idx, conf, count) of a submitted batch (the top-K arrays
        are None when it was submitted with top_k = 0)."""
        f32p, u32p = C.POINTER(C.c_float), C.POINTER(C.c_uint32)
        b, k = self._tickets.pop(ticket, (0, 0))
        cfg = self.model.config
        logits = np.empty((b, self.output_device(cfg.logits_output)[1]), dtype=np.float32)
        emb = None
        if cfg.has_embedding and want_embeddings:
            emb = np.empty((b, self.output_device(cfg.embedding_output)[1]), dtype=np.float32)
        idx = conf = cnt = None
        if k:
            idx, conf, cnt = np.zeros((b, k), dtype=np.uint32), np.zeros((b, k), dtype=np.float32), np.zeros(b, dtype=np.uint32)
        st = lib.bn_infer_collect(self._h, ticket, logits.ctypes.data_as(f32p), None if emb is None else emb.ctypes.data_as(f32p), k,
                                  None if idx is None else idx.ctypes.data_as(u32p), None if conf is None else conf.ctypes.data_as(f32p),
                                  None if cnt is None else cnt.ctypes.data_as(u32p), cancel, timeout_ns)
        if st:
            raise EngineError(st)
        return logits, emb, idx, conf, cnt

    def infer_device(self, d_ptr: int, batch: int, sync: bool = False):
        st = lib.bn_infer_device(self._h, C.c_void_p(d_ptr), batch, 1 if sync else 0)
        if st:
            raise EngineError(st)

    def step_device(self, d_ptr: int, batch: int, top_k: int = 10, min_confidence: Optional[float] = None,
                    sync: bool = False):
        """One whole hot-path pass (plan + top-K + D2H of logits / top-K) on a device-resident batch."""
        st = lib.bn_step_device(self._h, C.c_void_p(d_ptr), batch, top_k, 0 if min_confidence is None else 1,
                                C.c_float(min_confidence or 0.0), 1 if sync else 0)
        if st:
            raise EngineError(st)

    def step_results(self, batch: int):
        lg, ix, cf, ct = C.POINTER(C.c_float)(), C.POINTER(C.c_uint32)(), C.POINTER(C.c_float)(), C.POINTER(C.c_uint32)()
        ks = C.c_size_t()
        st = lib.bn_step_results(self._h, C.byref(lg), C.byref(ix), C.byref(cf), C.byref(ct), C.byref(ks))
        if st:
            raise EngineError(st)
        n = self.output_device(self.model.config.logits_output)[1]
        k = ks.value
        return (np.ctypeslib.as_array(lg, shape=(batch, n)).copy(), np.ctypeslib.as_array(ix, shape=(batch, k)).copy(),
                np.ctypeslib.as_array(cf, shape=(batch, k)).copy(), np.ctypeslib.as_array(ct, shape=(batch,)).copy())

    def synchronize(self):
        st = lib.bn_ctx_synchronize(self._h)
        if st:
            raise EngineError(st)

    def output_device(self, index: int):
        p, n = C.c_void_p(), C.c_size_t()
        st = lib.bn_ctx_output_device(self._h, index, C.byref(p), C.byref(n))
        if st:
            raise EngineError(st)
        return p.value, n.value

    def read_output(self, index: int, batch: int) -> np.ndarray:
        _, n = self.output_device(index)
        out = np.empty((batch, n), dtype=np.float32)
        st = lib.bn_ctx_read_output(self._h, index, batch, out.ctypes.data_as(C.POINTER(C.c_float)))
        if st:
            raise EngineError(st)
        return out

    def topk(self, batch: int, top_k: int, min_confidence: Optional[float] = None):
        n = self.model.config.num_species
        k = max(min(top_k, n), 1)
        idx = np.zeros((batch, k), dtype=np.uint32)
        conf = np.zeros((batch, k), dtype=np.float32)
        cnt = np.zeros(batch, dtype=np.uint32)
        u32p = C.POINTER(C.c_uint32)
        st = lib.bn_topk(self._h, batch, top_k, 0 if min_confidence is None else 1,
                         C.c_float(min_confidence or 0.0), k, idx.ctypes.data_as(u32p),
                         conf.ctypes.data_as(C.POINTER(C.c_float)), cnt.ctypes.data_as(u32p))
        if st:
            raise EngineError(st)
        return idx, conf, cnt

    def infer_windows(self, rec: "Recording", step_samples: int, first: int, count: int, want_embeddings: bool = True,
                      timeout_ns: int = 0, cancel=None):
        """bn_infer_windows: windows [first, first+count) of an uploaded recording -> (logits, embeddings)."""
        cfg = self.model.config
        f32p = C.POINTER(C.c_float)
        logits = np.empty((count, self.output_device(cfg.logits_output)[1]), dtype=np.float32)
        emb = None
        if cfg.has_embedding and want_embeddings:
            emb = np.empty((count, self.output_device(cfg.embedding_output)[1]), dtype=np.float32)
        st = lib.bn_infer_windows(self._h, rec._h, step_samples, first, count, logits.ctypes.data_as(f32p),
                                  None if emb is None else emb.ctypes.data_as(f32p), cancel, timeout_ns)
        if st:
            raise EngineError(st)
        return logits, emb

    def step_windows(self, rec: "Recording", step_samples: int, first: int, count: int, top_k: int = 10,
                     min_confidence: Optional[float] = None, sync: bool = False):
        """bn_step_windows: one asynchronous hot-path pass over windows of an uploaded recording."""
        st = lib.bn_step_windows(self._h, rec._h, step_samples, first, count, top_k, 0 if min_confidence is None else 1,
                                 C.c_float(min_confidence or 0.0), 1 if sync else 0)
        if st:
            raise EngineError(st)

    def time_kernels(self, batch: int):
        cap = 1024
        names = C.create_string_buffer(cap * BN_NAME_LEN)
        usec = (C.c_float * cap)()
        macs = (C.c_double * cap)()
        byts = (C.c_double * cap)()
        n = lib.bn_ctx_time_kernels(self._h, batch, C.cast(names, C.c_void_p), usec, macs, byts, cap)
        out = []
        for i in range(min(n, cap)):
            nm = names.raw[i * BN_NAME_LEN:(i + 1) * BN_NAME_LEN].split(b"\0", 1)[0].decode()
            out.append((nm, float(usec[i]), float(macs[i]), float(byts[i])))
        return out

    def launch_costs(self, batch: int):
        """Per launch: (macs on the matrix cores, macs on the vector ALU, recompute share of the first, algorithmic bytes)."""
        cap = 1024
        a, b, r, by = ((C.c_double * cap)() for _ in range(4))
        n = lib.bn_ctx_launch_costs(self._h, batch, a, b, r, by, cap)
        return [(float(a[i]), float(b[i]), float(r[i]), float(by[i])) for i in range(min(n, cap))]

    def stream(self) -> int:
        return lib.bn_ctx_stream(self._h) or 0


_pending_uploads = weakref.WeakSet()


def _join_uploads():
    for r in list(_pending_uploads):
        try:
            if getattr(r, "_h", None):
                lib.bn_recording_wait(r._h)
        except Exception:  # noqa: BLE001 -- exit path
            pass


atexit.register(_join_uploads)


class Recording:
    """bn_recording: a mono recording uploaded once in its storage format (int16 or float32)."""

    def __init__(self, samples: np.ndarray, device: int = 0, src_rate: Optional[int] = None, dst_rate: Optional[int] = None,
                 zero_crossings: int = 0, async_upload: bool = False):
        """src_rate / dst_rate given and different: converted on the device by the polyphase resampler
        (bn_recording_create_resampled); the recording then holds f32 samples at dst_rate.
        async_upload: bn_recording_create_async -- returns at once, the windows' calls wait for the samples they read (this object
        keeps the array alive until the upload is done)."""
        a = np.ascontiguousarray(samples)
        if a.ndim != 1 or a.dtype not in (np.int16, np.float32):
            raise ValueError("mono int16 or float32 samples expected")
        h = C.c_void_p()
        fmt = 0 if a.dtype == np.int16 else 1
        if async_upload and src_rate and dst_rate:
            raise ValueError("async_upload cannot be combined with resampling: bn_recording_create_resampled converts during its own (synchronous) upload")
        if src_rate and dst_rate:
            st = lib.bn_recording_create_resampled(device, a.ctypes.data_as(C.c_void_p), a.shape[0], fmt, src_rate, dst_rate, zero_crossings, C.byref(h))
        elif async_upload:
            self._keep = a  # the uploader thread reads it until wait() / free
            st = lib.bn_recording_create_async(device, a.ctypes.data_as(C.c_void_p), a.shape[0], fmt, C.byref(h))
        else:
            st = lib.bn_recording_create(device, a.ctypes.data_as(C.c_void_p), a.shape[0], fmt, C.byref(h))
        if st:
            raise EngineError(st)
        self._h = h
        self.n_samples = int(lib.bn_recording_samples(h))
        if async_upload:
            _pending_uploads.add(self)  # joined at interpreter exit: the uploader thread must not outlive the array it reads

    def wait(self):
        """The whole recording is on the device (no-op for a synchronous upload)."""
        st = lib.bn_recording_wait(self._h)
        if st:
            raise EngineError(st)
        self._keep = None

    def read_f32(self, first: int = 0, count: Optional[int] = None) -> np.ndarray:
        count = self.n_samples - first if count is None else count
        out = np.zeros(max(count, 0), dtype=np.float32)
        st = lib.bn_recording_read_f32(self._h, first, count, out.ctypes.data_as(C.POINTER(C.c_float)))
        if st:
            raise EngineError(st)
        return out

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:  # lib is gone at interpreter shutdown
            lib.bn_recording_free(self._h)
            self._h = None

    def n_windows(self, step_samples: int) -> int:
        return int(lib.bn_chunk_count(self.n_samples, step_samples))

    def windows(self, segment_samples: int, step_samples: int, first: int, count: int) -> np.ndarray:
        """chunk_audio on the device, copied back: f32 [count, segment_samples]."""
        out = np.zeros((count, segment_samples), dtype=np.float32)
        st = lib.bn_recording_windows(self._h, segment_samples, step_samples, first, count, out.ctypes.data_as(C.POINTER(C.c_float)))
        if st:
            raise EngineError(st)
        return out


# ---- range filter (reference src/rangefilter.rs) ----
@dataclass
class LocationScore:
    species: str
    score: float
    index: int


def calculate_week(month: int, day: int) -> float:
    return float(lib.bnh_calculate_week(month, day))


def validate_coordinates(latitude: float, longitude: float) -> None:
    err = BnhError()
    if lib.bnh_validate_coordinates(C.c_float(latitude), C.c_float(longitude), C.byref(err)):
        raise Error(err)


def validate_date(month: int, day: int) -> None:
    err = BnhError()
    if lib.bnh_validate_date(month, day, C.byref(err)):
        raise Error(err)


def filter_predictions(predictions: list, location_scores: list, threshold: float, rerank: bool) -> list:
    """filter_predictions_impl (rangefilter.rs:333-386) through the compiled C++ mirror."""
    n, m = len(predictions), len(location_scores)
    ps = (C.c_char_p * max(n, 1))(*[p.species.encode() for p in predictions])
    pc = (C.c_float * max(n, 1))(*[p.confidence for p in predictions])
    ls = (C.c_char_p * max(m, 1))(*[s.species.encode() for s in location_scores])
    lc = (C.c_float * max(m, 1))(*[s.score for s in location_scores])
    pos = (C.c_uint32 * max(n, 1))()
    conf = (C.c_float * max(n, 1))()
    k = lib.bnh_filter_predictions(ps, pc, n, ls, lc, m, C.c_float(threshold), 1 if rerank else 0, pos, conf)
    return [Prediction(predictions[pos[i]].species, float(conf[i]), predictions[pos[i]].index) for i in range(k)]


class RangeFilter:
    """reference src/rangefilter.rs:395-580; the meta model runs on the MI355X (BN_MODEL_GENERIC)."""

    def __init__(self, handle, threshold: float):
        self._h = handle
        self.threshold = threshold

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:
            lib.bnh_range_filter_free(self._h)
            self._h = None

    @staticmethod
    def builder() -> "RangeFilterBuilder":
        return RangeFilterBuilder()

    def predict(self, latitude: float, longitude: float, month: int, day: int) -> list:
        err, n = BnhError(), C.c_size_t()
        cap = 1 << 16
        idx = (C.c_uint32 * cap)()
        sc = (C.c_float * cap)()
        if lib.bnh_range_filter_predict(self._h, C.c_float(latitude), C.c_float(longitude), month, day, idx, sc, cap, C.byref(n), C.byref(err)):
            raise Error(err)
        return [LocationScore(lib.bnh_range_filter_label(self._h, idx[i]).decode(), float(sc[i]), int(idx[i])) for i in range(min(n.value, cap))]

    def filter_predictions(self, predictions: list, location_scores: list, rerank: bool) -> list:
        return filter_predictions(predictions, location_scores, self.threshold, rerank)

    def filter_batch_predictions(self, predictions_batch: list, location_scores: list, rerank: bool) -> list:
        return [filter_predictions(p, location_scores, self.threshold, rerank) for p in predictions_batch]


class RangeFilterBuilder:
    """reference src/rangefilter.rs:142-277."""

    def __init__(self):
        self._model_path = None
        self._labels_path = None
        self._labels = None
        self._threshold = 0.01
        self._device = 0

    def model_path(self, p: str):
        self._model_path = p
        return self

    def labels_path(self, p: str):
        self._labels_path, self._labels = p, None
        return self

    def labels(self, l: list):
        self._labels, self._labels_path = list(l), None
        return self

    def from_classifier_labels(self, l: list):
        return self.labels(l)

    def threshold(self, t: float):
        self._threshold = t
        return self

    def with_rocm(self, device: int = 0):
        self._device = device
        return self

    def build(self) -> RangeFilter:
        h, err = C.c_void_p(), BnhError()
        labels = None
        if self._labels is not None:
            labels = (C.c_char_p * max(len(self._labels), 1))(*[x.encode() for x in self._labels])
        st = lib.bnh_range_filter_build(self._model_path.encode() if self._model_path else None,
                                        self._labels_path.encode() if self._labels_path else None, labels,
                                        len(self._labels) if self._labels is not None else 0, C.c_float(self._threshold), self._device,
                                        C.byref(h), C.byref(err))
        if st:
            raise Error(err)
        return RangeFilter(h, self._threshold)


def resample_table(src_rate: int, dst_rate: int, zero_crossings: int = 0):
    """The resampler's polyphase table [L, T] and (L, M, T) -- host arithmetic, needs no device."""
    L, M, T = C.c_uint32(), C.c_uint32(), C.c_uint32()
    n = lib.bn_resample_table(src_rate, dst_rate, zero_crossings, None, 0, C.byref(L), C.byref(M), C.byref(T))
    tab = np.zeros(n, dtype=np.float32)
    lib.bn_resample_table(src_rate, dst_rate, zero_crossings, tab.ctypes.data_as(C.POINTER(C.c_float)), n, C.byref(L), C.byref(M), C.byref(T))
    return tab.reshape(L.value, T.value), L.value, M.value, T.value


def topk_host(logits: np.ndarray, top_k: int, min_confidence: Optional[float] = None, device: int = 0):
    """top_k_predictions (reference src/postprocess.rs:40-87) on the GPU for host logits [rows, n]."""
    a = np.ascontiguousarray(logits, dtype=np.float32)
    if a.ndim == 1:
        a = a[None, :]
    rows, n = a.shape
    k = max(min(top_k, n), 1)
    idx = np.zeros((rows, k), dtype=np.uint32)
    conf = np.zeros((rows, k), dtype=np.float32)
    cnt = np.zeros(rows, dtype=np.uint32)
    u32p = C.POINTER(C.c_uint32)
    st = lib.bn_topk_host(device, a.ctypes.data_as(C.POINTER(C.c_float)), rows, n, min(top_k, 2 ** 63 - 1),
                          0 if min_confidence is None else 1, C.c_float(min_confidence or 0.0), k,
                          idx.ctypes.data_as(u32p), conf.ctypes.data_as(C.POINTER(C.c_float)), cnt.ctypes.data_as(u32p))
    if st:
        raise EngineError(st)
    return idx, conf, cnt


class Group:
    """bn_group: one model replica per device, a recording sharded by window across them, results all-gathered on the
    devices (RCCL between distinct devices, device copies between ranks that share one)."""

    def __init__(self, models, max_batch: int = 32, contexts_per_device: int = 4):
        self.models = list(models)
        n = len(self.models)
        hs = (C.c_void_p * n)(*[m._h for m in self.models])
        devs = (C.c_int32 * n)(*[m.device for m in self.models])
        h = C.c_void_p()
        st = lib.bn_group_create(hs, devs, n, max_batch, contexts_per_device, C.byref(h))
        if st:
            raise RuntimeError(f"bn_group_create: status {st}: {group_last_error()}")
        self._h = h

    def __del__(self):
        if getattr(self, "_h", None):
            lib.bn_group_destroy(self._h)
            self._h = None

    def size(self) -> int:
        return lib.bn_group_size(self._h)

    def uses_rccl(self) -> bool:
        return bool(lib.bn_group_uses_rccl(self._h))

    def stats(self) -> dict:
        """bn_group_get_stats: the contexts' counters summed (capture_fallbacks must be 0)."""
        st = BnCtxStats()
        if lib.bn_group_get_stats(self._h, C.byref(st), C.sizeof(st)):
            raise RuntimeError(group_last_error())
        d = {k: int(getattr(st, k)) for k, _ in BnCtxStats._fields_ if k != "last_fallback"}
        d["last_fallback"] = st.last_fallback.decode(errors="replace")
        return d

    def analyze_recording(self, samples: np.ndarray, step_samples: int, top_k: int = 10, min_confidence: Optional[float] = None,
                          want_logits: bool = True):
        """(logits or None, idx, conf, count) for every window of the recording, in time order.  The default gathers the
        [G, N] logits too (the reference's `raw_scores`, classifier.rs:907,948); want_logits=False is the lighter opt-in that
        gathers the packed top-K rows only (the logits slab is then not even allocated)."""
        x = np.ascontiguousarray(samples)
        fmt = {np.dtype(np.int16): 0, np.dtype(np.float32): 1}[x.dtype]
        cfg = self.models[0].config
        G = lib.bn_chunk_count(x.shape[0], step_samples)
        N = int(cfg.num_species)
        k = min(top_k, N)
        f32p, u32p = C.POINTER(C.c_float), C.POINTER(C.c_uint32)
        logits = np.empty((G, N), dtype=np.float32) if want_logits else None
        idx, conf, cnt = np.zeros((G, max(k, 1)), dtype=np.uint32), np.zeros((G, max(k, 1)), dtype=np.float32), np.zeros(G, dtype=np.uint32)
        ng = C.c_size_t(0)
        st = lib.bn_group_analyze_recording(self._h, x.ctypes.data_as(C.c_void_p), x.shape[0], fmt, step_samples, top_k,
                                            0 if min_confidence is None else 1, C.c_float(min_confidence or 0.0),
                                            None if logits is None else logits.ctypes.data_as(f32p), max(k, 1), idx.ctypes.data_as(u32p),
                                            conf.ctypes.data_as(f32p), cnt.ctypes.data_as(u32p), C.byref(ng))
        if st:
            raise RuntimeError(f"bn_group_analyze_recording: status {st}: {group_last_error()}")
        assert ng.value == G
        return logits, idx[:, :k], conf[:, :k], cnt


def group_last_error() -> str:
    buf = C.create_string_buffer(1024)
    lib.bn_group_last_error(buf, 1024)
    return buf.value.decode(errors="replace")


def plan_describe(path: str, model_type: int = -1, all_outputs: bool = False) -> str:
    st = C.c_int32()
    n = lib.bn_plan_describe(path.encode(), model_type, 1 if all_outputs else 0, None, 0, C.byref(st))
    if st.value:
        raise EngineError(st.value)
    buf = C.create_string_buffer(n + 1)
    lib.bn_plan_describe(path.encode(), model_type, 1 if all_outputs else 0, buf, n + 1, C.byref(st))
    return buf.value.decode()


SHARING_AUTO, SHARING_ALONE, SHARING_SHARED = -1, 0, 1


def set_sharing_mode(mode: int) -> None:
    """bn_set_sharing_mode: grids sized for a device of the caller's own (0, the default), a shared one (1), or by the number of live
    contexts (-1).  Results are bit-identical in every mode."""
    lib.bn_set_sharing_mode(int(mode))


def model_survey(path: str):
    """bn_model_survey: (status, text).  Needs no device.  status 0 = every plan accepted, BN_ERR_UNSUPPORTED_MODEL = a plan refused
    (the text says which node and why); an unreadable file raises."""
    st = C.c_int32()
    n = lib.bn_model_survey(path.encode(), None, 0, C.byref(st))
    if n == 0 and st.value:
        raise EngineError(st.value)
    buf = C.create_string_buffer(n + 1)
    lib.bn_model_survey(path.encode(), buf, n + 1, C.byref(st))
    return st.value, buf.value.decode()


def parse_labels(content: str, csv: bool) -> list:
    n = lib.bnh_parse_labels(content.encode("utf-8"), 1 if csv else 0, None, 0)
    buf = C.create_string_buffer(max(n, 1))
    lib.bnh_parse_labels(content.encode("utf-8"), 1 if csv else 0, buf, max(n, 1))
    s = buf.value.decode("utf-8")
    return s.split("\n") if s else []


class LabelFormat(enum.IntEnum):
    """reference src/types.rs LabelFormat."""
    Text = 0
    Csv = 1
    Json = 2


def parse_labels_format(content: str, fmt: "LabelFormat") -> list:
    """parse_labels (reference src/labels.rs:33-39) through the compiled C++ mirror; raises Error(LabelParse)."""
    err = BnhError()
    raw = content.encode("utf-8")
    n = lib.bnh_parse_labels_format(raw, int(fmt), None, 0, C.byref(err))
    if n == 0:
        raise Error(err)
    buf = C.create_string_buffer(n)
    lib.bnh_parse_labels_format(raw, int(fmt), buf, n, C.byref(err))
    s = buf.raw[:n - 1].decode("utf-8")
    return s.split("\x1f") if s else []


def chunk_plan(n_samples: int, segment_samples: int, overlap_secs: float, sample_rate: int):
    n = lib.bnh_chunk_plan(n_samples, segment_samples, C.c_float(overlap_secs), sample_rate, None, None, 0)
    starts = np.zeros(max(n, 1), dtype=np.uint64)
    times = np.zeros(max(n, 1), dtype=np.float32)
    lib.bnh_chunk_plan(n_samples, segment_samples, C.c_float(overlap_secs), sample_rate,
                       starts.ctypes.data_as(C.POINTER(C.c_uint64)), times.ctypes.data_as(C.POINTER(C.c_float)), n)
    return starts[:n], times[:n]


def device_count() -> int:
    return lib.bn_device_count()

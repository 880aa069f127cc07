"""Minimal ONNX file writer (protobuf wire format, no onnx/protobuf dependency).

Used to author the seeded synthetic-weight model files the tests and the
benchmark run (no real BirdNET / Perch .onnx file exists offline).  Field
numbers follow the published onnx.proto3 schema.
"""
from __future__ import annotations

import struct
from typing import Iterable, Sequence

import numpy as np

FLOAT, INT64, INT32 = 1, 7, 6


def _varint(v: int) -> bytes:
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _key(field: int, wt: int) -> bytes:
    return _varint((field << 3) | wt)


def _ld(field: int, payload: bytes) -> bytes:
    return _key(field, 2) + _varint(len(payload)) + payload


def _str(field: int, s: str) -> bytes:
    return _ld(field, s.encode("utf-8"))


def _int(field: int, v: int) -> bytes:
    return _key(field, 0) + _varint(v)


def tensor_proto(name: str, arr: np.ndarray, scalar: bool = False) -> bytes:
    """scalar: write a one-element array with NO dims (rank 0), as exporters write scalar indices / attributes-as-inputs; the default
    keeps the historical [1] so that the committed synthetic models do not change."""
    arr = np.ascontiguousarray(arr)
    if scalar:
        assert arr.size == 1
        arr = arr.reshape(())
    if arr.dtype == np.float32:
        dt = FLOAT
    elif arr.dtype == np.int64:
        dt = INT64
    elif arr.dtype == np.int32:
        dt = INT32
    else:
        raise TypeError(f"unsupported dtype {arr.dtype}")
    out = b"".join(_int(1, d) for d in arr.shape)
    out += _int(2, dt)
    out += _str(8, name)
    out += _ld(9, arr.tobytes())
    return out


def _attr(name: str, value) -> bytes:
    out = _str(1, name)
    if isinstance(value, float):
        out += _key(2, 5) + struct.pack("<f", value) + _int(20, 1)
    elif isinstance(value, (int, np.integer)):
        out += _int(3, int(value)) + _int(20, 2)
    elif isinstance(value, str):
        out += _ld(4, value.encode()) + _int(20, 3)
    elif isinstance(value, np.ndarray):
        out += _ld(5, tensor_proto("", value)) + _int(20, 4)
    elif isinstance(value, (list, tuple)):
        if all(isinstance(v, (int, np.integer)) for v in value):
            out += b"".join(_int(8, int(v)) for v in value) + _int(20, 7)
        else:
            out += b"".join(_key(7, 5) + struct.pack("<f", float(v)) for v in value) + _int(20, 6)
    else:
        raise TypeError(f"unsupported attribute {name}={value!r}")
    return out


def _value_info(name: str, shape: Sequence, elem_type: int = FLOAT) -> bytes:
    dims = b""
    for d in shape:
        if isinstance(d, str):
            dims += _ld(1, _str(2, d))  # dim_param
        elif d is None or d < 0:
            dims += _ld(1, _str(2, "batch"))
        else:
            dims += _ld(1, _int(1, d))
    tensor_type = _int(1, elem_type) + _ld(2, dims)
    return _str(1, name) + _ld(2, _ld(1, tensor_type))


class GraphBuilder:
    """Accumulates nodes / initializers; tensors are referred to by name."""

    def __init__(self, name: str = "g", opset: int = 17):
        self.name = name
        self.opset = opset
        self.nodes: list[bytes] = []
        self.inits: list[bytes] = []
        self.inputs: list[bytes] = []
        self.outputs: list[bytes] = []
        self._n = 0
        self.node_count = 0
        # every constant in creation order as (name, hint, array): lets tests rebuild the network from the raw weights
        # without ever going through the serialised file
        self.const_values: list = []

    def fresh(self, hint: str = "t") -> str:
        self._n += 1
        return f"{hint}_{self._n}"

    def const(self, arr, hint: str = "c", scalar: bool = False) -> str:
        name = self.fresh(hint)
        self.inits.append(tensor_proto(name, np.asarray(arr), scalar))
        self.const_values.append((name, hint, np.asarray(arr)))
        return name

    def add_input(self, name: str, shape: Sequence):
        self.inputs.append(_value_info(name, shape))

    def add_output(self, name: str, shape: Sequence | None):
        if shape is None:
            self.outputs.append(_str(1, name))
        else:
            self.outputs.append(_value_info(name, shape))

    def node(self, op: str, inputs: Iterable[str], n_out: int = 1, outputs: Sequence[str] | None = None,
             **attrs) -> str | list[str]:
        outs = list(outputs) if outputs is not None else [self.fresh(op.lower()) for _ in range(n_out)]
        body = b"".join(_str(1, i) for i in inputs)
        body += b"".join(_str(2, o) for o in outs)
        self.node_count += 1
        body += _str(3, f"{op}_{self.node_count}")
        body += _str(4, op)
        for k, v in attrs.items():
            body += _ld(5, _attr(k, v))
        self.nodes.append(body)
        return outs[0] if len(outs) == 1 else outs

    def serialize(self) -> bytes:
        g = b"".join(_ld(1, n) for n in self.nodes)
        g += _str(2, self.name)
        g += b"".join(_ld(5, t) for t in self.inits)
        g += b"".join(_ld(11, i) for i in self.inputs)
        g += b"".join(_ld(12, o) for o in self.outputs)
        model = _int(1, 8)  # ir_version
        model += _str(2, "birdnet-hip-synth")
        model += _ld(7, g)
        model += _ld(8, _str(1, "") + _int(2, self.opset))
        return model

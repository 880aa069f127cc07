"""The SURVEY.md 8(d) synthetic segments WITHOUT the package (tools/dump_ort_golden.py runs where libbirdnet_hip.so does not exist):
0.5 sin(2 pi f t) + 0.05 u(t), f cycling over 440 / 1000 / 2500 / 6000 Hz, u the LCG of the reference's src/testutil.rs:110-121 seeded
12345 + segment index, every 32nd segment silent (tests/integration_test.rs:52-54).  Bit-identical to
rust-birdnet-onnx_amd/synth.py::synthetic_segments (tests/test_real_model_plumbing.py compares the two)."""
import numpy as np


def segments(n: int, sample_count: int, sample_rate: int, first_index: int = 0) -> np.ndarray:
    out = np.empty((n, sample_count), dtype=np.float32)
    t = np.arange(sample_count, dtype=np.float64) / sample_rate
    freqs = (440.0, 1000.0, 2500.0, 6000.0)
    a, c, mask = 1103515245, 12345, (1 << 64) - 1
    for k in range(n):
        gi = first_index + k
        if gi % 32 == 31:
            out[k] = 0.0
            continue
        # state_{j+1} = a state_j + c (mod 2^64), state_0 = 12345 + gi; sample j reads bits 16..31 of state_{j+1}
        mult = np.empty(sample_count, dtype=np.uint64)
        add = np.empty(sample_count, dtype=np.uint64)
        with np.errstate(over="ignore"):
            mult[0], add[0] = a, c
            filled = 1
            while filled < sample_count:                                  # j-step maps by doubling: s -> M_j s + A_j
                take = min(filled, sample_count - filled)
                mult[filled:filled + take] = mult[:take] * mult[filled - 1]
                add[filled:filled + take] = mult[:take] * add[filled - 1] + add[:take]
                filled += take
            state = mult * np.uint64((12345 + gi) & mask) + add
        bits = ((state >> np.uint64(16)) & np.uint64(0xFFFF)).astype(np.float64)
        out[k] = (0.5 * np.sin(2.0 * np.pi * freqs[gi % 4] * t) + 0.05 * (bits * (2.0 / 65535.0) - 1.0)).astype(np.float32)
    return out

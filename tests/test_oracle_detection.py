"""Pins oracle/detection.c against /root/reference/src/detection.rs:183-284 and types.rs,
plus properties of chunk_audio (birdnet-analyze.rs:707-743; the reference has no test for it)."""
import numpy as np

import oracle


def test_detect_birdnet_v24():  # detection.rs:189-202
    c = oracle.detect_model_type([1, 144000], [[1, 6522]])
    assert c.model_type == oracle.MT_BIRDNET_V24
    assert (c.sample_rate, c.segment_duration, c.sample_count) == (48000, 3.0, 144000)
    assert c.num_species == 6522 and not c.has_embedding


def test_detect_birdnet_v30():  # :204-217
    c = oracle.detect_model_type([1, 160000], [[1, 1024], [1, 1000]])
    assert c.model_type == oracle.MT_BIRDNET_V30
    assert (c.sample_rate, c.segment_duration, c.sample_count) == (32000, 5.0, 160000)
    assert c.num_species == 1000 and c.has_embedding and c.embedding_dim == 1024


def test_detect_perch_v2():  # :219-238
    c = oracle.detect_model_type([1, 160000],
                                 [[1, 1536], [1, 16, 4, 1536], [1, 500, 128], [1, 14795]])
    assert c.model_type == oracle.MT_PERCH_V2
    assert c.num_species == 14795 and c.embedding_dim == 1536 and c.sample_rate == 32000


def test_detect_with_perch_override():  # :240-257
    c = oracle.detect_model_type([1, 160000], [[1, 512], [1, 16, 4, 512], [1, 500, 128], [1, 500]],
                                 oracle.MT_PERCH_V2)
    assert c.model_type == oracle.MT_PERCH_V2 and c.embedding_dim == 512 and c.num_species == 500


def test_detect_with_invalid_override():  # :259-268
    assert oracle.detect_model_type([1, 160000], [[1, 1024], [1, 1000]],
                                    oracle.MT_BIRDNET_V24) is None


def test_detect_unsupported_model():  # :270-281
    assert oracle.detect_model_type([1, 100000], [[1, 1000]]) is None


def test_extract_sample_count_2d_3d():  # :283-291
    assert oracle.detect_model_type([1, 144000], [[1, 10]]).sample_count == 144000
    assert oracle.detect_model_type([1, 1, 144000], [[1, 10]]).sample_count == 144000
    assert oracle.detect_model_type([-1, 144000], [[-1, 10]]).sample_count == 144000
    assert oracle.detect_model_type([144000], [[1, 10]]) is None
    assert oracle.detect_model_type([1, -1], [[1, 10]]) is None


def test_override_wrong_output_count():  # detection.rs:100-133
    assert oracle.detect_model_type([1, 144000], [[1, 5], [1, 6]], oracle.MT_BIRDNET_V24) is None
    assert oracle.detect_model_type([1, 160000], [[1, 5]], oracle.MT_BIRDNET_V30) is None
    assert oracle.detect_model_type([1, 160000], [[1, 5], [1, 6]], oracle.MT_PERCH_V2) is None


def test_chunk_plan_no_overlap():
    starts, times = oracle.chunk_plan(144000 * 3 + 10, 144000, 0.0, 48000)
    assert starts.tolist() == [0, 144000, 288000, 432000]  # trailing mostly-padding chunk emitted
    assert times.tolist() == [0.0, 3.0, 6.0, 9.0]


def test_chunk_plan_overlap_and_degenerate():
    starts, _ = oracle.chunk_plan(300000, 144000, 1.5, 48000)
    assert starts.tolist() == list(range(0, 300000, 72000))
    assert len(oracle.chunk_plan(1000, 144000, 3.0, 48000)[0]) == 0   # step == 0 -> empty
    assert len(oracle.chunk_plan(1000, 144000, 4.0, 48000)[0]) == 0   # saturating_sub
    assert len(oracle.chunk_plan(0, 144000, 0.0, 48000)[0]) == 0      # empty input


def test_chunk_fill_pads_with_zeros():
    x = np.arange(1, 11, dtype=np.float32)
    assert oracle.chunk_fill(x, 4, 8).tolist() == [9.0, 10.0, 0.0, 0.0]
    assert oracle.chunk_fill(x, 4, 0).tolist() == [1.0, 2.0, 3.0, 4.0]

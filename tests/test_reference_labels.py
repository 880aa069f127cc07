"""The reference's label-parser unit tests (src/labels.rs:127-404), test for test, through the compiled C++ mirror
(`parse_labels` with a LabelFormat).  Inputs and expected lists are the reference's known answers, restated as data."""
import pytest


@pytest.fixture()
def P(bn):
    return lambda content, fmt: bn.parse_labels_format(content, fmt)


def test_parse_text_labels(bn, P):
    assert P("American Robin\nNorthern Cardinal\n\nBlue Jay\n", bn.LabelFormat.Text) == ["American Robin", "Northern Cardinal", "Blue Jay"]


def test_parse_text_labels_with_whitespace(bn, P):
    assert P("  American Robin  \n  Northern Cardinal  ", bn.LabelFormat.Text) == ["American Robin", "Northern Cardinal"]


def test_parse_csv_labels_simple(bn, P):
    assert P("American Robin\nNorthern Cardinal\nBlue Jay", bn.LabelFormat.Csv) == ["American Robin", "Northern Cardinal", "Blue Jay"]


def test_parse_csv_labels_with_header(bn, P):
    content = "label,scientific_name\nAmerican Robin,Turdus migratorius\nNorthern Cardinal,Cardinalis cardinalis"
    assert P(content, bn.LabelFormat.Csv) == ["American Robin", "Northern Cardinal"]


@pytest.mark.parametrize("header", ["species", "inat2024_fsd50k", "dataset_fsd50k"])
def test_parse_csv_labels_header_variants(bn, P, header):      # species / perch_v2_inat / perch_v2_fsd50k headers
    assert P(f"{header}\nAmerican Robin\nNorthern Cardinal", bn.LabelFormat.Csv) == ["American Robin", "Northern Cardinal"]


def test_parse_json_array(bn, P):
    assert P('["American Robin", "Northern Cardinal", "Blue Jay"]', bn.LabelFormat.Json) == ["American Robin", "Northern Cardinal", "Blue Jay"]


def test_parse_json_object_with_labels(bn, P):
    assert P('{"labels": ["American Robin", "Northern Cardinal"]}', bn.LabelFormat.Json) == ["American Robin", "Northern Cardinal"]


@pytest.mark.parametrize("key", ["name", "label", "species"])
def test_parse_json_array_of_objects(bn, P, key):               # name / label / species keys
    assert P(f'[{{"{key}": "American Robin"}}, {{"{key}": "Northern Cardinal"}}]', bn.LabelFormat.Json) == ["American Robin", "Northern Cardinal"]


def test_parse_json_invalid(bn, P):
    with pytest.raises(bn.Error) as e:
        P('{"invalid": "format"}', bn.LabelFormat.Json)
    assert e.value.kind == bn.ErrorKind.LabelParse
    assert str(e.value) == "failed to parse labels: unrecognized JSON format: expected array of strings, {labels: [...]}, or [{name: ...}]"


def test_parse_labels_by_format(bn, P):
    assert len(P("American Robin\nNorthern Cardinal", bn.LabelFormat.Text)) == 2
    assert len(P('["American Robin", "Northern Cardinal"]', bn.LabelFormat.Json)) == 2


def test_load_labels_file_not_found(bn):
    with pytest.raises(bn.Error) as e:
        bn.Classifier.builder().model_path("/nonexistent/model.onnx").labels_path("/nonexistent/path.txt").build()
    # the builder reads the labels after the model; the label loader itself is reached through the range filter builder
    with pytest.raises(bn.Error) as e:
        bn.RangeFilter.builder().model_path("/nonexistent/meta.onnx").labels_path("/nonexistent/path.txt").build()
    assert "failed to load labels" in str(e.value)


def test_parse_text_labels_empty_lines(bn, P):
    assert P("Species 1\n\nSpecies 2\n\n\nSpecies 3", bn.LabelFormat.Text) == ["Species 1", "Species 2", "Species 3"]


def test_parse_text_labels_with_unicode(bn, P):
    assert P("Pingüino Emperador\n鸟类\nПтица\n🐦", bn.LabelFormat.Text) == ["Pingüino Emperador", "鸟类", "Птица", "🐦"]


def test_parse_text_labels_with_special_chars(bn, P):
    assert P("Species (Common)\nSpecies-rare\nSpecies_variant\nSpecies's", bn.LabelFormat.Text) == ["Species (Common)", "Species-rare", "Species_variant", "Species's"]


def test_parse_csv_labels_inconsistent_columns(bn, P):
    assert P("label,scientific\nSpecies 1,Name1,Extra\nSpecies 2,Name2", bn.LabelFormat.Csv) == ["Species 1", "Species 2"]


def test_parse_csv_labels_empty_values(bn, P):
    assert P("label\n\nSpecies 1\n\nSpecies 2", bn.LabelFormat.Csv) == ["Species 1", "Species 2"]


def test_parse_json_array_empty(bn, P):
    assert P("[]", bn.LabelFormat.Json) == []


def test_parse_json_array_with_unicode(bn, P):
    assert P('["Pingüino", "鸟类", "Птица"]', bn.LabelFormat.Json) == ["Pingüino", "鸟类", "Птица"]


def test_parse_json_array_of_objects_missing_keys(bn, P):
    assert P('[{"name": "Species 1"}, {"other": "Species 2"}]', bn.LabelFormat.Json) == ["Species 1"]


def test_parse_json_deeply_nested(bn, P):
    with pytest.raises(bn.Error):
        P('{"data": {"labels": ["Species 1"]}}', bn.LabelFormat.Json)


def test_parse_text_labels_only_whitespace(bn, P):
    assert P("   \n\t\n  \n", bn.LabelFormat.Text) == []


def test_parse_csv_labels_quoted_values(bn, P):
    content = 'label\n"Species, with comma"\n"Species with ""quotes"""\nSpecies normal'
    assert P(content, bn.LabelFormat.Csv) == ["Species, with comma", 'Species with "quotes"', "Species normal"]


def test_json_details_beyond_the_reference_tests(bn, P):
    """serde_json semantics the reference relies on implicitly."""
    assert P(r'["aé\n", "🐦"]', bn.LabelFormat.Json) == ["aé\n", "🐦"]      # escapes, surrogate pairs
    assert P('{"labels": ["x"], "version": 3}', bn.LabelFormat.Json) == ["x"]                  # unknown keys are ignored
    assert P('[{"name": null, "label": "L"}, {"species": "S", "rank": 1}]', bn.LabelFormat.Json) == ["L", "S"]
    for bad in ('["a", 1]', '[{"name": 5}]', '["a"] trailing', '', '{"labels": "x"}', '[{"other": 1}]'):
        with pytest.raises(bn.Error):
            P(bad, bn.LabelFormat.Json)

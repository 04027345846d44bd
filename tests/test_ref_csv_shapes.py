"""The fixture tests/golden/ref_benchmark_csv_shapes.json against the reference's shipped benchmark files (where the reference is present),
.  The GPU tests compare the CSV files the drivers write with the same fixture (tests/test_gpu_cpp_drivers.py:
assert_reference_shape -- ordered section labels per directory kind, rank header row, fields of the file name)."""
import importlib.util
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_benchmark_csv_shapes.json")))


def test_fixture_is_what_the_reference_ships():
    if not os.path.isdir("/root/reference/benchmarks"):
        pytest.skip("the reference is not here: the committed fixture stands")
    spec = importlib.util.spec_from_file_location("make_ref_csv_shapes", os.path.join(ROOT, "tests", "golden", "make_ref_csv_shapes.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    assert json.loads(json.dumps(m.shapes())) == SHAPES
    assert sum(e["files"] for e in SHAPES.values()) > 25000

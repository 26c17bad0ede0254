"""CPU: the host-only legs of bench.py -- the cpu_baseline object of the contract and the core count it
reports (the GPU legs need an MI355X and are exercised by running bench.py itself)."""
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def test_cpu_baseline_object(forest_paths):
    import bench
    assert 1 <= bench.host_cores() <= 16
    args = types.SimpleNamespace(forest=forest_paths["zero"], cpu_seconds=1.0)
    cb = bench.cpu_baseline(args, 256, 96)
    assert set(["value", "unit", "cores", "kind", "sample"]) <= set(cb)
    assert cb["unit"] == "Mpix/s" and cb["cores"] == 1 and cb["kind"] in ("port", "reference") and cb["value"] > 0
    if cb.get("all_cores"):
        assert cb["all_cores"]["cores"] == bench.host_cores() and cb["all_cores"]["value"] > 0


def test_default_workload_is_baseline_config():
    import bench
    old = sys.argv
    sys.argv = ["bench.py"]
    try:
        a = bench.parse_args()
    finally:
        sys.argv = old
    assert (a.gpus, a.width, a.height, a.batch) == (1, 1024, 436, 256)
    assert os.path.basename(a.forest) == "defaultZeroForest.txt" and a.steps > 0 and a.warmup >= 0

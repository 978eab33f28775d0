"""The driver parses bench.py's LAST stdout line: it must stay small (round 4's 29 KB line was not parsed) and keep the
contract keys.  The recorded full result of round 4 (profiles/r04_bench_default_final.json) is the fixture."""
import io
import json
from contextlib import redirect_stdout
from pathlib import Path

import pytest

from julius_amd import benchfmt

ROOT = Path(__file__).resolve().parent.parent
RECORDED = sorted((ROOT / "profiles").glob("r0[45]_bench_default*.json"))

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")


@pytest.mark.parametrize("path", RECORDED, ids=lambda p: p.name)
def test_final_line_is_small_and_round_trips(path):
    full = json.loads(path.read_text())
    if "roofline" not in full or "note" not in full["roofline"]:
        pytest.skip("already a compact record")
    s = benchfmt.final_line(full)
    assert "\n" not in s and len(s) < 6000, len(s)
    line = json.loads(s)
    for k in CONTRACT:
        assert k in line, k
    assert line["value"] == pytest.approx(full["value"], rel=1e-5)
    assert line["ms_per_step"] == pytest.approx(full["ms_per_step"], rel=1e-5)
    assert line["config"]["workload"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms"):
        assert k in line["roofline"], k
    assert line["roofline"]["frac"] == pytest.approx(full["roofline"]["frac"], rel=1e-5)
    for k in ("value", "cores", "kind"):
        assert k in line["cpu_baseline"], k
    nested = [k for k, v in full.items() if isinstance(v, dict) and "ms_per_step" in v]
    assert nested
    for k in nested:
        assert line[k]["ms_per_step"] == pytest.approx(full[k]["ms_per_step"], rel=1e-5)
        if "timing" in full[k]:                                   # the product's serving loop: its own clock instead of a roofline
            assert line[k]["timing"]["decode_s"] == pytest.approx(full[k]["timing"]["decode_s"], rel=1e-3)
            assert line[k]["parity"]["identical"] == full[k]["parity"]["result_lines_vs_in_process"]["identical"]
            continue
        assert "frac" in line[k]["roofline"]
        if "cpu_baseline" in full[k]:
            assert line[k]["cpu_baseline"]["rtf_inv"] == pytest.approx(full[k]["cpu_baseline"]["rtf_inv"], rel=1e-3)
        if "parity" in full[k] and "device_vs_compiled_reference" in full[k]["parity"]:
            vs = full[k]["parity"]["device_vs_compiled_reference"]
            assert line[k]["parity"]["utts"] == vs["utts"]
            assert line[k]["parity"]["identical"] <= vs["trellis_identical"]


def test_oversized_tree_still_fits():
    """Forty nested configurations: the fallback keeps the contract keys and shrinks the nested records."""
    full = json.loads(RECORDED[0].read_text())
    for i in range(40):
        full[f"extra_{i}"] = dict(full["e2e"])
    s = benchfmt.final_line(full)
    line = json.loads(s)
    for k in CONTRACT:
        assert k in line
    assert len(s) < 16000
    assert "hbm" in line["roofline"]


def test_incomplete_parity_is_flagged():
    full = json.loads(RECORDED[0].read_text())
    vs = full["e2e"]["parity"]["device_vs_compiled_reference"]
    vs["wanted"] = vs["utts"] + 2
    line = benchfmt.compact_line(full)
    assert line["e2e"]["parity"]["incomplete"] is True


def test_emit_prints_detail_lines_first_and_compact_line_last(tmp_path, monkeypatch):
    import bench
    full = json.loads(RECORDED[0].read_text())
    monkeypatch.setattr(bench, "ROOT", tmp_path)
    buf = io.StringIO()
    with redirect_stdout(buf):
        bench.emit(full)
    lines = buf.getvalue().strip().splitlines()
    assert len(lines) >= 3
    last = json.loads(lines[-1])
    assert len(lines[-1]) < 6000 and last["metric"] == full["metric"] and "bench_detail" not in last
    assert all("bench_detail" in json.loads(x) for x in lines[:-1])
    assert json.loads((tmp_path / "bench_detail.json").read_text())["e2e"]["ms_per_step"] == full["e2e"]["ms_per_step"]

"""bench.py prints ONE JSON line with the fields the driver reads (task contract): checked on a tiny configuration."""
import json
import os
import subprocess
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_line_has_the_contract_fields():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "c2", "--steps", "2", "--warmup", "1", "--scale", "0.25", "--spp", "16",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    j = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline", "secondary"):
        assert key in j, key
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["warmup"] == 1 and j["higher_is_better"] is True and j["scaling"] == "weak"
    assert j["vs_baseline"] is None and j["data"] == "synthetic" and j["unit"] == "Msamples/s" and j["dtype"] == "f32"
    assert "workload" in j["config"] and "model" not in j["config"]
    assert j["value"] > 0 and j["ms_per_step"] > 0
    r = j["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    # a fraction of a roofline is at most 1: against the HBM peak, or -- when a cache-resident tree serves more algorithmic
    # bytes per second than HBM could -- against the L2 gather ceiling, named as such
    assert r["unit"] == "GB/s" and (r["bound"] == "hbm" and r["peak"] == 8000.0 or r["bound"].startswith("l2-gather") and r["peak"] > 8000.0)
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0 < r["frac"] <= 1.0
    assert r["traffic"] is None          # no PMC summary matches this reduced configuration: nothing is borrowed from another run
    assert r["avg_launch_ms"] > 0 and r["launches"] > 0 and sum(r["stage_ms_per_step"].values()) > 0

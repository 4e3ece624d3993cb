"""bench.py's output contract, on a workload small enough for the test suite: one JSON line with the driver's keys, the
roofline object and the cpu_baseline object (the numbers themselves are the default run's business)."""
import contextlib
import io
import json

import pytest

pytestmark = pytest.mark.gpu


def test_bench_line_has_the_contract_keys():
    import bench
    out = io.StringIO()
    with contextlib.redirect_stdout(out):
        bench.main(["--gpus", "1", "--steps", "16", "--warmup", "1", "--games", "256", "--playouts", "40",
                    "--cohort", "64", "--cpu-budget", "1.0"])
    lines = [l for l in out.getvalue().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["metric"] == "selfplay_games_per_sec" and d["unit"] == "games/s" and d["value"] > 0
    assert (d["n_gpus"], d["steps"], d["warmup"]) == (1, 16, 1) and d["ms_per_step"] > 0
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"].startswith("f32") and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] in (157.3, 2500.0)
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["achieved"] > 0 and r["traffic"] > 0
    assert r["launches"] > 0 and r["avg_launch_ms"] > 0
    # value is the SURVEY 8d metric: games finished inside the window / wall, reproducible from the line itself
    c = d["config"]
    assert abs(d["value"] - c["games_finished_in_window"] / (d["ms_per_step"] * d["steps"] * 1e-3)) < 1e-6 * max(d["value"], 1)
    assert d["value_steady_state_estimate"] > 0 and c["preroll_steps_untimed"] > 0
    legs = [k for k in d if k.startswith("second_leg_")]
    assert len(legs) == 1 and d[legs[0]]["value"] >= 0 and d[legs[0]]["roofline"]["kernel"] != r["kernel"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and c["unit"] == "games/s" and c["sample"]
    assert d["config"]["plies_stored_in_replay"] > 0 and d["config"]["leaf_evals"] > 0

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
    # the evaluation cache is on in the headline and its work is accounted for; the leg without it evaluates every leaf
    k = d["config"]
    assert k["eval_cache_hits"] > 0 and 0 < k["eval_cache_hit_share_of_nonterminal_leaves"] < 1 and k["eval_cache_entries_per_gpu"] > 0
    off = d["leg_no_eval_cache"]
    assert off["value"] > 0 and off["eval_cache_hits"] == 0 and off["leaf_evals"] > 0


def test_bench_config4_line():
    """bench.py --config 4 (BASELINE configs[4]: NBTTT, fused tree search) at a size the suite can afford: an HBM-class roofline
    object for k_search_fused<NBTTT> and a cpu_baseline from the C port on the same game and evaluator."""
    import bench
    out = io.StringIO()
    with contextlib.redirect_stdout(out):
        bench.main(["--gpus", "1", "--config", "4", "--steps", "12", "--warmup", "2", "--games", "512", "--playouts", "100",
                    "--cohort", "128", "--cpu-budget", "1.0", "--preroll", "8"])
    d = json.loads([l for l in out.getvalue().splitlines() if l.startswith("{")][0])
    assert d["metric"] == "selfplay_games_per_sec" and d["value"] > 0 and d["config"]["baseline_config"] == 4
    assert "NBTTT" in d["config"]["workload"] and "configs[4]" in d["config"]["workload"] and d["config"]["mode"] == "lockstep"
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and r["kernel"] == "k_search_fused<NBTTT>"
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["achieved"] > 0 and r["launches"] == 12
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0 and "NBTTT" in d["cpu_baseline"]["sample"]
    assert d["config"]["plies_stored_in_replay"] > 0

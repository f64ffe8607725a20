"""The committed bench line (profiles/r1_bench.json, produced by bench.py on an MI355X) carries every field the driver's
contract names, and the rocprofv3 summary of the same command is committed next to it."""
import csv
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contract_fields():
    d = json.load(open(os.path.join(ROOT, "profiles", "r1_bench.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "audio-seconds/sec" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and d["vs_baseline"] is None and d["n_gpus"] == 1
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_us"):
        assert k in r, k
    assert r["bound"] in ("mfma", "hbm") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["unit"] == d["unit"]
    # value = clips * 3 s * steps / elapsed
    assert abs(d["value"] - 36 * 3.0 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-3


def test_rocprof_summary_of_the_same_command_is_committed_and_agrees():
    under = json.loads(open(os.path.join(ROOT, "profiles", "r1_bench_under_rocprof.json")).read())
    kern = under["roofline"]["kernel"]                       # e.g. mlp_fused[C=45]
    cp = {"45": "48", "72": "80"}.get(kern.split("=")[1].rstrip("]"), kern.split("=")[1].rstrip("]"))
    prefix = "mlp_fused_lds_kernel<" if kern.startswith("mlp") else "attn_"
    rows = list(csv.DictReader(open(os.path.join(ROOT, "profiles", "r1_kernel_stats.csv"))))
    match = [r for r in rows if prefix in r["Name"] and f"<{cp}," in r["Name"]]
    assert match, f"no kernel-trace row for {kern}"
    calls = sum(int(r["Calls"]) for r in match)
    avg_us = sum(float(r["TotalDurationNs"]) for r in match) / calls / 1e3
    assert abs(avg_us - under["roofline"]["avg_us"]) / avg_us < 0.10, (avg_us, under["roofline"]["avg_us"])


def test_round3_bench_lines_are_committed_with_roofline_traffic_and_cpu_baseline():
    """Round 3: the codec line (with the PMC traffic of the dominant kernel, taken on the committed kernel sources), the training line and the
    adversarial line (BASELINE configs[4]) with its per-kernel block."""
    d = json.load(open(os.path.join(ROOT, "profiles", "r3_bench_default_with_traffic.json")))
    assert d["n_gpus"] == 1 and d["scaling"] == "weak" and "configs[1]" in d["config"]["workload"] and d["dtype"] == "f32"
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["traffic"] and r["traffic"] > 1e8 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1 and d["value"] > 100 * d["cpu_baseline"]["value"]
    assert abs(d["value"] - 36 * 3.0 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-3
    t = json.load(open(os.path.join(ROOT, "profiles", "r3_bench_train.json")))
    assert t["config"]["tape_gb"] < 40 and t["roofline"]["frac"] > 0 and t["cpu_baseline"]["kind"] == "port"
    a = json.load(open(os.path.join(ROOT, "profiles", "r3_bench_train_adv.json")))
    assert "configs[4]" in a["config"]["workload"] and a["ms_per_step"] < 420
    rows = list(csv.DictReader(open(os.path.join(ROOT, "profiles", "r3_kernel_stats.csv"))))
    assert any("mlp_fused_lds_kernel" in r["Name"] for r in rows) and any("attn_" in r["Name"] for r in rows)

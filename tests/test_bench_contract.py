"""
bench.py's contract: the one JSON line the driver parses.  The CPU part checks the static side (workload table
against BASELINE.json, flags, defaults); the GPU part runs a small batch of every workload and checks the line's
schema, the roofline and cpu_baseline objects, and that a parity mismatch would be fatal (exit code 3 path exists).
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_workloads_cover_the_baseline_configs():
    sys.path.insert(0, ROOT)
    import bench
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    cfgs = base["configs"]
    assert "secp256k1 variable-base" in cfgs[1] and bench.WORKLOADS["secp256k1-var"][4] == "configs[1]"
    assert "Ed25519 fixed-base" in cfgs[2] and bench.WORKLOADS["ed25519-fixed"][4].startswith("configs[2]")
    assert "P-256 variable-base" in cfgs[3] and bench.WORKLOADS["p256-var"][4].startswith("configs[3]")
    assert "double-scalar-mul" in cfgs[4] and bench.WORKLOADS["secp256k1-double"][4].startswith("configs[4]")
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'default="secp256k1-var"' in src            # the headline is the default workload
    assert "raise SystemExit(3)" in src or "sys.exit(3)" in src   # a parity mismatch is fatal
    assert set(bench.WORKLOADS) <= set(__import__("forge_ec_amd.build", fromlist=["x"]).WORKLOAD_TU)


def test_strong_scaling_names_the_baseline_configs():
    """--scaling strong runs the two 8-GPU BASELINE configurations as quoted: ONE global batch (2^22 P-256, 2^20
    secp256k1 double-mul) split into contiguous shards; the line's config.workload quotes BASELINE.json verbatim."""
    sys.path.insert(0, ROOT)
    import bench
    from forge_ec_amd.dist import shard_range
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))["configs"]
    assert bench.BASELINE_LOG2_GLOBAL == {"p256-var": 22, "secp256k1-double": 20}
    assert "2^22 P-256" in base[3] and "2^20 secp256k1 ECDSA batch-verify" in base[4]
    for world in (1, 2, 4, 8):
        for wl, lg in bench.BASELINE_LOG2_GLOBAL.items():
            sizes = [shard_range(1 << lg, r, world)[1] - shard_range(1 << lg, r, world)[0] for r in range(world)]
            assert sum(sizes) == 1 << lg and max(sizes) - min(sizes) <= 1
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert '"scaling": args.scaling' in src and "base_cfgs[3]" in src and "base_cfgs[4]" in src
    assert set(bench.EXECUTED_MULS) == set(bench.WORKLOADS)
    # the two timed regions of --gather both: with the gather to rank 0 first (that one is `value`), then without
    assert 'modes = ["rank0", "none"] if gather_arg == "both"' in src


LINE_KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
             "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"}


@pytest.mark.gpu
@pytest.mark.parametrize("workload", ["secp256k1-var", "ed25519-fixed", "p256-var", "secp256k1-double", "ed25519-var"])
def test_bench_line_schema(workload):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--log2-batch", "13",
                        "--steps", "3", "--warmup", "1", "--cpu-seconds", "0.5"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    t = json.loads(lines[0])
    assert LINE_KEYS <= set(t)
    assert t["n_gpus"] == 1 and t["steps"] == 3 and t["warmup"] == 1 and t["higher_is_better"] is True
    assert t["scaling"] == "weak" and t["vs_baseline"] is None and t["data"] == "synthetic" and t["dtype"] == "u32"
    assert t["unit"] == "scalar-muls/s" and t["value"] > 0 and t["ms_per_step"] > 0
    assert abs(t["value"] - (1 << 13) / (t["ms_per_step"] * 1e-3)) / t["value"] < 1e-6
    assert "workload" in t["config"] and "model" not in t["config"]
    roof = t["roofline"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "executed_mul_insts_per_unit",
            "valu_issue_cycles_per_inst_per_simd"} <= set(roof)
    assert "valu_busy_pct" not in roof            # a busy fraction above 100 % is a derived-counter artefact: not printed
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9 and roof["kernel"]
    cpu = t["cpu_baseline"]
    assert {"value", "unit", "cores", "kind", "sample", "single_thread", "cpu_model", "build"} <= set(cpu)
    assert cpu["kind"] == "port" and cpu["cores"] >= 1 and cpu["single_thread"]["cores"] == 1 and cpu["single_thread"]["value"] > 0
    assert cpu["parity_sample_bit_exact"] is True and "bit-exact" in t["metric"]


@pytest.mark.gpu
def test_bench_strong_scaling_line_on_one_gpu():
    """The strong-scaling mode at N = 1 (the whole global batch is rank 0's shard) for the two 8-GPU configurations at
    a reduced global size: schema, scaling = strong, batch_per_gpu = global_batch."""
    for workload in ("p256-var", "secp256k1-double"):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--scaling", "strong",
                            "--log2-global-batch", "13", "--steps", "2", "--warmup", "1", "--cpu-seconds", "0.5"],
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        t = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
        assert t["scaling"] == "strong" and t["config"]["global_batch"] == 1 << 13 == t["config"]["batch_per_gpu"]
        assert abs(t["value"] - (1 << 13) / (t["ms_per_step"] * 1e-3)) / t["value"] < 1e-6
        assert t["cpu_baseline"]["parity_sample_bit_exact"] is True

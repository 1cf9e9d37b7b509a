"""bench.py's N > 1 path on a box without GPUs: the self-launch (`bench.py --gpus N` with no launcher around it) and
the whole control flow -- both timed regions, the double-buffered ResultGather, the max-reduce, rank 0's one JSON
line -- with TWO real ranks over gloo, the oracle standing in for the kernels (tests/bench_rehearsal.py)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra)
    return env


def _line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout   # exactly ONE JSON line, from rank 0
    return json.loads(lines[0])


def _rehearse(*flags):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "bench_rehearsal.py")] + list(flags), cwd=ROOT,
                       env=_env(FEC_BENCH_CHECK_GATHER="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    return _line(r.stdout)


def test_self_launch_command_is_built_before_torch_is_imported():
    """`python bench.py --gpus 4 ...` with no RANK in the environment becomes the torch.distributed.run command the
    driver would have typed, as a CHILD process, and this process has not imported torch (let alone touched a GPU)."""
    code = (
        "import json, subprocess, sys\n"
        "sys.path.insert(0, %r)\n"
        "import bench\n"
        "def fake(cmd, env=None):\n"
        "    print(json.dumps({'cmd': cmd, 'torch_loaded': 'torch' in sys.modules, 'ipc': env.get('HSA_ENABLE_IPC_MODE_LEGACY')}))\n"
        "    return 7\n"
        "subprocess.call = fake\n"
        "bench.main(['--gpus', '4', '--steps', '3', '--warmup', '1', '--workload', 'p256-var'])\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=120)
    assert r.returncode == 7, r.stdout + r.stderr   # the launcher's exit code is passed on
    t = _line(r.stdout)
    cmd = t["cmd"]
    assert t["torch_loaded"] is False
    assert t["ipc"] == "0"
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert int(cmd[cmd.index("--master-port") + 1]) > 0
    at = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[at + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1", "--workload", "p256-var"]


def test_a_rank_started_by_a_launcher_does_not_launch_again():
    """With RANK set the process is a rank: --gpus must match WORLD_SIZE (the torchrun form keeps working unchanged)."""
    code = ("import sys\nsys.path.insert(0, %r)\nimport bench\nbench.main(['--gpus', '4'])\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=_env(RANK="0", LOCAL_RANK="0", WORLD_SIZE="2"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "--gpus 4 but WORLD_SIZE=2" in r.stderr


def test_two_ranks_weak_scaling_both_gathers():
    t = _rehearse("--gpus", "2", "--steps", "3", "--warmup", "2", "--log2-batch", "6", "--gather", "both")
    assert t["n_gpus"] == 2 and t["steps"] == 3 and t["warmup"] == 2 and t["scaling"] == "weak"
    assert t["config"]["batch_per_gpu"] == 64 and t["config"]["global_batch"] == 128
    assert set(t["gather"]) == {"rank0", "none"}
    assert t["value"] == t["gather"]["rank0"]["value"] > 0          # `value` is the run WITH the gather
    assert abs(t["value"] - 128 * 3 / (t["ms_per_step"] * 3e-3)) < 1e-6 * t["value"]   # whole-job units / max-over-ranks time
    assert "cpu_baseline" not in t                                   # an N = 1 measurement


@pytest.mark.parametrize("workload,log2", [("p256-var", 7), ("secp256k1-double", 6)])
def test_two_ranks_strong_scaling_baseline_shapes(workload, log2):
    """The two 8-GPU BASELINE configurations' form: ONE global batch split into contiguous shards, the gathered batch
    checked against each rank's own shard (FEC_BENCH_CHECK_GATHER)."""
    t = _rehearse("--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", workload, "--scaling", "strong",
                  "--log2-global-batch", str(log2), "--gather", "both")
    assert t["scaling"] == "strong" and t["n_gpus"] == 2
    assert t["config"]["global_batch"] == 1 << log2 and t["config"]["batch_per_gpu"] == 1 << (log2 - 1)
    assert set(t["gather"]) == {"rank0", "none"} and t["value"] > 0
    if workload == "secp256k1-double":
        assert t["roofline"]["wavefront_rounds"]["wavefronts"] == 2 * (((1 << (log2 - 1)) + 63) // 64)   # two launches side by side
        assert t["roofline"]["executed_share_of_reference_steps"] == (512 - 24) / 512.0   # 24 of multiply(G, u1)'s steps are a table fetch
        assert abs(t["roofline"]["frac_executed_steps"] - t["roofline"]["frac"] * (512 - 24) / 512.0) < 1e-12


def test_clock_probe_helper_parses_rocm_smi_and_survives_its_absence(tmp_path):
    """bench.ClockSampler: the helper process (started before the GPU is touched) runs `rocm-smi --showclocks --json` between
    `start` and `stop`; with a stand-in rocm-smi on PATH that prints what the GPU box's prints, the summary carries the
    sclk / mclk readings; with no rocm-smi at all the summary is None and nothing raises."""
    import time
    sys.path.insert(0, ROOT)
    import bench
    fake = tmp_path / "rocm-smi"
    fake.write_text("#!/bin/sh\necho 'WARNING: some banner'\n"
                    "echo '{\"card0\": {\"fclk clock speed:\": \"(1250Mhz)\", \"mclk clock speed:\": \"(2000Mhz)\", \"mclk clock level:\": \"0\", "
                    "\"sclk clock speed:\": \"(2391Mhz)\", \"sclk clock level:\": \"1\"}, \"card1\": {\"sclk clock speed:\": \"(95Mhz)\"}}'\n")
    fake.chmod(0o755)
    old_path = os.environ["PATH"]
    try:
        os.environ["PATH"] = str(tmp_path) + os.pathsep + old_path
        with bench.ClockSampler(0) as c:
            time.sleep(0.4)
        s = c.summary()
        assert s and s["sclk_mhz"]["min"] == s["sclk_mhz"]["max"] == 2391 and s["mclk_mhz"]["max"] == 2000 and s["sclk_mhz"]["samples"] >= 1
        os.environ["PATH"] = str(tmp_path / "nothing-here")
        with bench.ClockSampler(0) as c:
            time.sleep(0.1)
        assert c.summary() is None
    finally:
        os.environ["PATH"] = old_path

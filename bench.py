#!/usr/bin/env python3
"""
bench.py -- headline benchmark of the batched scalar-multiplication hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W        (N > 1, one rank per GPU)

With N > 1 and no RANK in the environment the first form starts the second one itself: N fresh rank processes through
torch.distributed.run on 127.0.0.1 and a free port, spawned BEFORE this process has imported torch or touched a GPU
(a process that has initialised the GPU is never replaced or re-used), and rank 0's JSON line is the output.

Default workload (BASELINE.json configs[1], the headline): 2^20 secp256k1 variable-base scalar
multiplications per GPU (--workload selects the other BASELINE configurations) on synthetic seeded inputs (forge_ec_amd/synth.py), inputs resident in HBM before the timed
region.  One "step" = one pass of the hot path over one batch.

N > 1: no collective on the compute path; the result shards are gathered to rank 0 over RCCL/xGMI, overlapped
with the next step's kernel.  --scaling weak (default): every rank runs its own 2^--log2-batch shard.
--scaling strong --log2-global-batch G: ONE global batch of 2^G elements split into contiguous shards
(forge_ec_amd.dist.shard_range) -- how BASELINE.json quotes its two 8-GPU configurations:
    --workload p256-var --scaling strong --log2-global-batch 22          configs[3]
    --workload secp256k1-double --scaling strong --log2-global-batch 20  configs[4]
--gather both (the default when N > 1) times the K steps twice, with the gather to rank 0 (`value`) and without it
(`gather.none`), so that the gather's cost is visible in one line.

Prints ONE JSON line on rank 0: metric/value (whole-job scalar-muls/s), roofline (integer-VALU:
algorithmic 32x32 multiply-adds per second against the chip's peak; kernel time from HIP events
on the launch stream) and cpu_baseline (the C oracle timed on this box's host cores on a bounded
sample of the same inputs, which doubles as a parity spot-check of the GPU output).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# workload -> (curve, kind, algorithmic MAD32 per unit, algorithmic HBM bytes per unit, BASELINE config)
# MAD32 counts are SURVEY.md section 8(d)'s "needed" column (popcount-128 averages for P-256/Ed25519)
WORKLOADS = {
    "secp256k1-var": ("secp256k1", "var", 693248, 224, "configs[1]"),
    "ed25519-fixed": ("ed25519", "fixed", 82944, 160, "configs[2]"),
    "p256-var": ("p256", "var", 278528, 224, "configs[3] (per-GPU shard)"),
    "secp256k1-double": ("secp256k1", "double", 1388384, 256, "configs[4] (per-GPU shard)"),
    "ed25519-var": ("ed25519", "var", 248832, 288, "-"),
    "secp256k1-fixed": ("secp256k1", "fixed", 693248, 128, "north_star fixed-base target"),
}
# for the workloads whose work depends on the scalars' set bits the per-run figure is computed from
# the actual batch (mean popcount pc of this rank's scalars), not from the popcount-128 average of the table above
ALG_FROM_POPCOUNT = {
    "p256-var": lambda pc: 256 * 576 + pc * 1024,        # 256 doublings (9 Mul) + one addition (16 Mul) per set bit
    "ed25519-fixed": lambda pc: pc * 648,                # one addition (9 Mul of 72 MAD32) per set bit
    "ed25519-var": lambda pc: (256 + pc) * 648,          # 256 doublings of the addend + one addition per set bit
}
CURVE_ID = {"secp256k1": 0, "p256": 1, "ed25519": 2}
# where the PMC traffic is far above the algorithmic bytes ON PURPOSE
TRAFFIC_NOTES = {
    "ed25519-fixed": "batches of 2^16 elements and more are walked in popcount order (a permutation of the whole batch): each "
                     "lane gathers its 32-byte scalar from its own line, so the fetch side counts whole lines (about 90 MB "
                     "over the 160 MB algorithmic); the kernel moves 47 GB/s, the sort is worth 6 % of its time",
    "ed25519-var": "the running result of every in-flight element lives in its slot of the output array and is read and "
                   "rewritten by ~128 additions per element (L2 / Infinity Cache working set, 1.4 TB/s): LDS holds the "
                   "addends of 1024 elements per CU instead, which is what lets the kernel run three wavefronts per SIMD; "
                   "with both in LDS (592 slots, two wavefronts per SIMD) it moved 305 MB and took 22.4 ms against 19.5",
}
# guide-derived integer-VALU peak: 256 CUs x 4 SIMDs x 64 lanes x 2.4 GHz / 4 cycles per
# v_mad_u64_u32 wave-instruction (MI355X_MICROARCH.md chip parameters; issue cost measured,
# profiles/valu_rates2_r01.txt) = 39.3e12 MAD32/s
PEAK_MAD32_FORMULA = 256 * 4 * 64 * 2.4e9 / 4.0


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="secp256k1-var", choices=list(WORKLOADS),
                    help="default = the headline (BASELINE.json configs[1])")
    ap.add_argument("--log2-batch", type=int, default=20, help="weak scaling: scalar-muls per GPU per step = 2^this")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: 2^--log2-batch per GPU; strong: 2^--log2-global-batch in all, split into contiguous shards")
    ap.add_argument("--log2-global-batch", type=int, default=None,
                    help="strong scaling: scalar-muls per step over ALL GPUs = 2^this (default: the size BASELINE.json "
                         "quotes for the workload, else 20)")
    ap.add_argument("--gather", default=None, choices=["both", "rank0", "all", "none"],
                    help="N>1: `both` (default) = time the K steps with the gather of the result shards to rank 0 over RCCL "
                         "(overlapped with the next step; this is `value`) and again without it (`gather.none`); or one of "
                         "rank0 / all (all-gather to every rank) / none only")
    ap.add_argument("--scalars-below-l", action="store_true",
                    help="Ed25519 workloads: scalars below 2^252 < l (the reference's Scalar::random leaves them below "
                         "2^255 - 19, the default here; SURVEY.md section 8d asks for this run as well)")
    ap.add_argument("--fixed-prefix-bits", type=int, default=None,
                    help="fixed-base workloads: size of the generator's prefix table (fec_ctx_set_fixed_prefix_bits; "
                         "0 = off; default: the library's, 24)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU-baseline duration (all legs together)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-clock-probe", action="store_true", help="skip the ~1 s rocm-smi clock sampling under load (N = 1)")
    return ap.parse_args(argv)


def host_cores():
    """Usable host cores: the affinity mask, capped by the cgroup CPU quota (a one-GPU box exposes
    every logical CPU but grants a share of them)."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            cores = min(cores, max(1, int(int(quota) / int(period) + 0.5)))
    except Exception:
        pass
    return max(1, min(cores, 64))


def cpu_baseline(workload, inputs, gpu_out, target_s):
    """Time the C oracle (the reference's full work, discarded doublings included) on this box's host cores over a
    bounded prefix of the same inputs -- on ONE thread and on all granted cores (SURVEY.md section 8d asks for both) --
    and compare the all-core leg's output with the GPU output (the parity sample).  The timed library is the oracle's
    source rebuilt on this box with -O3 -march=native when gcc is here (c_oracle.native_lib), else the shipped
    x86-64-v2 build; `build` says which."""
    c_oracle = load_checker()
    curve_name, kind = WORKLOADS[workload][0], WORKLOADS[workload][1]
    cid = CURVE_ID[curve_name]
    cores = host_cores()
    L, build = c_oracle.native_lib()

    def run(m, threads):
        if kind == "var":
            return c_oracle.batch_mul(cid, inputs[0][:m], inputs[1][:m], nthreads=threads, L=L)
        if kind == "fixed":
            return c_oracle.batch_mul_fixed(cid, inputs[0][:m], c_oracle.generator(cid), nthreads=threads, L=L)
        return c_oracle.batch_double_mul(cid, inputs[0][:m], inputs[1][:m], inputs[2][:m], nthreads=threads, L=L)

    def leg(threads, seconds):
        probe = 64 * threads
        t0 = time.perf_counter()
        run(probe, threads)
        rate = probe / max(time.perf_counter() - t0, 1e-9)
        m = int(min(inputs[0].shape[0], max(probe, rate * seconds)))
        t0 = time.perf_counter()
        ref = run(m, threads)
        return m, time.perf_counter() - t0, ref

    n1, dt1, ref1 = leg(1, target_s / 3.0)
    n, dt, ref = leg(cores, target_s * 2.0 / 3.0) if cores > 1 else (n1, dt1, ref1)
    ok = bool(np.array_equal(ref, gpu_out[:n])) and bool(np.array_equal(ref1, gpu_out[:n1]))
    return {"value": n / dt, "unit": "scalar-muls/s", "cores": cores, "kind": "port",
            "single_thread": {"value": n1 / dt1, "unit": "scalar-muls/s", "cores": 1,
                              "sample": "first %d of the rank-0 batch, %.1f s" % (n1, dt1)},
            "cpu_model": c_oracle.cpu_model(), "build": build,
            "sample": "first %d of the rank-0 batch (%s), C oracle oracle/forge_ec_oracle.c (a restatement of the "
                      "reference's Rust, not rustc output), %d threads, %.1f s" % (n, workload, cores, dt),
            "parity_sample_bit_exact": ok}


def load_checker():
    """The CPU checker of the cpu_baseline leg -- the only place this file touches oracle/.  main() calls it once BEFORE
    the GPU is touched when the leg will run, so that the first use on a box compiles the checker (gcc, a child process)
    while this process holds no GPU state; cpu_baseline() then finds the library built."""
    from oracle import c_oracle
    return c_oracle


def committed_pmc(workload, n):
    """HBM bytes per launch and the VALU issue interval from a committed rocprofv3 PMC pass (bench.py cannot collect
    PMC itself).  Only a pass taken on THIS build counts: tools/pmc_summarize.py records the library's
    source hash beside the counters, and a pass whose hash differs from the loaded library's is
    ignored -- the fields are then null and say why."""
    import glob
    from forge_ec_amd import build as fbuild
    tu = fbuild.WORKLOAD_TU.get(workload)
    here = fbuild.tu_closure_hash(tu, kernel_code_only=True) if tu else fbuild.source_hash()   # the kernel's translation unit + its includes
    stale = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "pmc_r*", "*", "pmc.json")), reverse=True):
        try:
            t = json.load(open(path))
        except Exception:
            continue
        if t.get("workload") != workload or t.get("units_per_launch") != n:
            continue
        rel = os.path.relpath(path, ROOT)
        if t.get("kernel_source_hash", t.get("source_hash")) != here:
            stale = stale or rel
            continue
        d = t.get("derived", {})
        return {"traffic": d.get("hbm_bytes_per_launch"),
                "valu_issue_cycles": valu_issue_cycles(t),
                "source": "%s: committed rocprofv3 PMC pass of this kernel's sources (%s + includes, hash %s), not measured by this run"
                          % (rel, tu, here[:12])}
    why = ("only a stale PMC pass exists (%s, other source hash)" % stale) if stale else "no committed PMC pass for this workload"
    return {"traffic": None, "valu_issue_cycles": None, "source": why}


# Multiply instructions the kernels EXECUTE per unit (v_mad_u64_u32 + v_mul_lo_u32 on the hot path, static count from
# tools/isa_mix.py: profiles/r02_isa_mix_k_secp_mul.txt, profiles/r03_isa_mix_sched_kernels.txt) next to the ALGORITHMIC
# MAD32 of the reference's op sequence that `roofline.achieved` is priced with: the kernels need fewer multiplies than the
# reference performs (closed-form Montgomery recurrence, exact squarings, z2z2 computed once per element), so `frac` is
# NOT the share of issue slots spent multiplying -- carries outnumber multiplies in every one of them.
# Fixed-base workloads: `b` = bits of the generator's prefix table (fecgpu.h: fec_ctx_set_fixed_prefix_bits) -- the first
# b ladder steps (Ed25519: the additions of the b low bits, b / 2 on average) are a table fetch, not executed.
EXECUTED_MULS = {
    "secp256k1-var": lambda b: 256 * 1600, "secp256k1-fixed": lambda b: (256 - b) * 1600,
    "secp256k1-double": lambda b: (512 - b) * 1600 + 1600,
    "p256-var": lambda b: 256 * 324 + 128 * 848, "ed25519-var": lambda b: 255 * 536 + 128 * 648,
    "ed25519-fixed": lambda b: (128 - b // 2) * 648,
}
# the size BASELINE.json quotes for a workload's configuration (strong scaling default)
BASELINE_LOG2_GLOBAL = {"p256-var": 22, "secp256k1-double": 20}


def valu_issue_cycles(pmc_json):
    """Cycles per VALU wave-instruction per SIMD from a committed PMC pass: (GRBM_GUI_ACTIVE / 8 XCDs) / (SQ_INSTS_VALU /
    1024 SIMDs).  This -- not the derived VALUBusy counter, which reads 102-104 % on these kernels -- is the statement
    "the kernel is VALU-issue bound": a 64-lane instruction occupies its SIMD for about four cycles."""
    c = pmc_json.get("counters", {})
    try:
        return (c["GRBM_GUI_ACTIVE"]["per_launch"] / 8.0) / (c["SQ_INSTS_VALU"]["per_launch"] / 1024.0)
    except (KeyError, ZeroDivisionError):
        return None


class CudaPlatform:
    """Where the bench runs: rank `local_rank`'s MI355X, RCCL between the ranks.  (tests/bench_rehearsal.py substitutes
    a CPU / gloo platform whose context is the oracle, to drive THIS file's N > 1 control flow with two real ranks on a
    box without GPUs; nothing here knows about it.)"""
    backend = "nccl"

    def __init__(self, local_rank):
        import torch
        self.torch, self.local_rank = torch, local_rank
        torch.cuda.set_device(local_rank)
        self.device = torch.device("cuda", local_rank)

    def init_process_group(self, dist):
        dist.init_process_group(backend=self.backend, device_id=self.device)

    def context(self):
        if os.environ.get("FEC_AB_LIB"):  # same-box A/B of another build of the library (tools/pmc_quick.sh, tools/quick_perf.py)
            from forge_ec_amd import _lib
            _lib.SO_PATH = os.path.abspath(os.environ["FEC_AB_LIB"])
        import forge_ec_amd as F
        return F.Context(self.local_rank)

    def new_stream(self):
        """An explicit (non-default) stream, made torch's current one; returns its raw handle."""
        self.tstream = self.torch.cuda.Stream()
        self.torch.cuda.set_stream(self.tstream)
        return self.tstream.cuda_stream

    def synchronize(self):
        self.torch.cuda.synchronize()


CLOCK_HELPER = r"""
import json, re, subprocess, sys, threading
card = sys.argv[1]
samples = {"sclk": [], "mclk": []}
state = {"on": False}
def once():
    out = subprocess.run(["rocm-smi", "--showclocks", "--json"], capture_output=True, text=True, timeout=20).stdout
    c = json.loads(out[out.index("{"):]).get(card, {})
    for key, val in c.items():
        m = re.search(r"(\d+)\s*mhz", str(val).lower())
        for clk in ("sclk", "mclk"):
            if m and key.lower().startswith(clk + " clock speed"):
                samples[clk].append(int(m.group(1)))
def loop():
    while state["on"]:
        try:
            once()
        except Exception:
            return
t = None
for line in sys.stdin:
    cmd = line.strip()
    if cmd == "start":
        state["on"] = True
        t = threading.Thread(target=loop, daemon=True)
        t.start()
    elif cmd == "stop":
        state["on"] = False
        if t:
            t.join(timeout=30)
        print(json.dumps(samples), flush=True)
        break
"""


class ClockSampler:
    """sclk / mclk of the bench's GPU under load, so that a box-to-box spread of a kernel time can be pinned on clocks or
    ruled out.  The amdgpu sysfs tables of a shared host cannot be matched to the one GPU a container is given (its PCI
    numbering is virtual), so the readings come from `rocm-smi --showclocks --json`, which sees exactly the visible
    GPU(s).  rocm-smi is run by a HELPER process that main() starts before this process has touched the GPU (a process
    that holds the GPU never forks a program); the helper samples, again and again, between `start` and `stop`, while the
    caller keeps the kernel running back to back (about a second, outside every timed region)."""

    def __init__(self, card_index):
        import subprocess
        self.proc = None
        try:
            self.proc = subprocess.Popen([sys.executable, "-c", CLOCK_HELPER, "card%d" % card_index], stdin=subprocess.PIPE,
                                         stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
        except Exception:
            self.proc = None
        self.result = None

    def _send(self, word):
        try:
            self.proc.stdin.write(word + "\n")
            self.proc.stdin.flush()
            return True
        except Exception:
            return False

    def __enter__(self):
        if self.proc:
            self._send("start")
        return self

    def __exit__(self, *a):
        if self.proc and self._send("stop"):
            try:
                self.result = json.loads(self.proc.stdout.readline() or "null")
            except Exception:
                self.result = None
        self.close()

    def close(self):
        if self.proc:
            try:
                self.proc.stdin.close()
                self.proc.wait(timeout=30)
            except Exception:
                self.proc.kill()
            self.proc = None

    def summary(self):
        v = self.result or {}
        if not v.get("sclk") and not v.get("mclk"):
            return None
        out = {"source": "rocm-smi --showclocks, sampled by a helper process while the kernel ran back to back"}
        for key in ("sclk", "mclk"):
            if v.get(key):
                out[key + "_mhz"] = {"min": min(v[key]), "max": max(v[key]), "samples": len(v[key])}
        return out


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(n_ranks, script, argv):
    """`bench.py --gpus N` without a launcher: N rank processes of `script` through torch.distributed.run, as children.
    The caller has not imported torch or touched a GPU at this point.  Returns the launcher's exit code."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), script] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL across processes needs on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def main(argv=None, platform_factory=None, script=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse(argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "RANK" not in os.environ:
        # no launcher around us: be the launcher (before torch is imported or a GPU is touched in this process)
        raise SystemExit(self_launch(args.gpus, script or os.path.abspath(__file__), argv))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.gpus > 1 and world == 1:
        raise SystemExit("--gpus %d but this launcher started one rank" % args.gpus)
    if not args.no_cpu_baseline and world == 1 and platform_factory is None:
        # build / load the CPU checker before anything touches the GPU (its first use on a box compiles it with gcc)
        load_checker().native_lib()
    # the clock probe's helper process, likewise started while this process holds no GPU state
    clock_sampler = ClockSampler(local_rank) if (world == 1 and platform_factory is None and not args.no_clock_probe) else None
    import torch
    plat = (platform_factory or CudaPlatform)(local_rank)
    import forge_ec_amd as F
    from forge_ec_amd import synth
    from forge_ec_amd.dist import shard_range

    dist = None
    # FEC_BENCH_FORCE_DIST=1 rehearses the N>1 code path (RCCL init, overlapped gather, max-reduce)
    # with a single rank on a one-GPU box
    force_dist = os.environ.get("FEC_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ
    if world > 1 or force_dist:
        import torch.distributed as dist
        plat.init_process_group(dist)

    workload = args.workload
    curve, kind, alg, hbm_bytes, cfg = WORKLOADS[workload]
    cid = CURVE_ID[curve]
    limbs = F.POINT_LIMBS[cid]
    ctx = plat.context()
    strong = args.scaling == "strong"
    if strong:
        log2_global = args.log2_global_batch if args.log2_global_batch is not None else BASELINE_LOG2_GLOBAL.get(workload, 20)
        n_global = 1 << log2_global
        lo, hi = shard_range(n_global, rank, world)
        n = hi - lo
        if n == 0:
            raise SystemExit("a rank has an empty shard")
    else:
        n = 1 << args.log2_batch
        n_global, lo = n * world, None

    # synthetic seeded inputs, resident in HBM before timing.  Weak scaling: one stream set per rank.  Strong scaling:
    # ONE global batch (the same data whatever N is), every rank generates it and keeps its contiguous shard.
    def gen(count, seed_off):
        k = synth.scalars(count, cid, 1000 + seed_off)
        if args.scalars_below_l and curve == "ed25519":
            k[:, 3] &= np.uint64((1 << 60) - 1)   # < 2^252 < l; a scalar that became zero is as rare as 2^-252
        arrs = [k]
        if kind in ("var", "double"):
            arrs.append(synth.points(count, cid, 1001 + seed_off) if kind == "var" else synth.scalars(count, cid, 1001 + seed_off))
        if kind == "double":
            arrs.append(synth.points(count, cid, 1002 + seed_off))
        return arrs
    if strong:
        inputs = [np.ascontiguousarray(a[lo:lo + n]) for a in gen(n_global, 0)]
    else:
        inputs = gen(n, 3 * rank)
    mean_popcount = float(np.unpackbits(np.ascontiguousarray(inputs[0]).view(np.uint8)).sum()) / n
    if workload in ALG_FROM_POPCOUNT:
        alg = ALG_FROM_POPCOUNT[workload](mean_popcount)
    d_in = [torch.from_numpy(a.view(np.int64)).to(plat.device) for a in inputs]
    d_out = [torch.empty((n, limbs), dtype=torch.int64, device=plat.device) for _ in range(2)]
    # All launches and collectives are ordered on ONE explicit (non-default) torch stream: the library
    # launches on the stream it is handed (a null handle would mean its own ctx stream, which torch's
    # collectives know nothing about), RCCL orders its internal stream against torch's current stream
    # at enqueue, and handle.wait() makes that stream wait for the collective -- so the gather reads a
    # finished shard and the next kernel cannot overwrite a buffer the gather is still reading.
    stream = plat.new_stream()
    assert stream != 0

    gather_arg = args.gather if args.gather is not None else ("both" if dist is not None else "none")
    if dist is None:
        gather_arg = "none"
    modes = ["rank0", "none"] if gather_arg == "both" else [gather_arg]   # the first mode's time is `value`
    gathers = {}
    if dist is not None:
        from forge_ec_amd.dist import ResultGather
        for m in modes:
            if m != "none":
                gathers[m] = [ResultGather(n_global, limbs, plat.device,
                                           dst=0 if m == "rank0" else None) for _ in range(2)]

    launched = {"name": None}  # the kernel(s) the library reports for the timed launch

    # Fixed-base prefix table of the generator (fixed and double workloads): built by the first launch that can use it.
    # Timed here, outside every timed region: one call that builds it, one that finds it (both over the whole batch, so
    # that every launch of the workload's kernel in a profile of this command is a full-size one).
    prefix = None
    if kind in ("fixed", "double"):
        # asked for explicitly, so that the table exists before the timed region whatever the batch size (a ctx left to
        # its defaults would build it in the launch that takes it past 2^21 multiplications by the generator -- for a
        # small per-GPU shard that is somewhere inside the timed steps)
        want_bits = args.fixed_prefix_bits
        if want_bits is None:
            want_bits = int(os.environ.get("FEC_FIXED_PREFIX_BITS") or 24)
        ctx.set_fixed_prefix_bits(want_bits)
        ctx.build_fixed_prefix(cid)    # attach / build now: the table exists before anything is timed

        def small_call():
            t0 = time.perf_counter()
            if kind == "fixed":
                ctx.batch_mul_fixed_dev(cid, d_in[0].data_ptr(), ctx.generator_dev(cid), d_out[0].data_ptr(), n, stream)
            else:
                ctx.batch_double_mul_dev(cid, d_in[0].data_ptr(), d_in[1].data_ptr(), d_in[2].data_ptr(), d_out[0].data_ptr(),
                                         n, stream)
            plat.synchronize()
            return (time.perf_counter() - t0) * 1e3
        first_ms, second_ms = small_call(), small_call()
        bits = ctx.fixed_prefix_bits(cid)
        entry_bytes = {0: 192, 1: 96, 2: 128}[cid]
        prefix = {"bits": bits, "table_bytes": (entry_bytes << bits) if bits else 0,
                  "build_ms_once_per_ctx": round(max(first_ms - second_ms, 0.0), 3),
                  "note": "state of multiply(G, k) after its first `bits` steps for every pattern of those bits, computed one step "
                          "of the reference's loop per entry and level by the launch that takes the ctx past 2^21 "
                          "multiplications by the generator, kept in HBM; results are identical without it "
                          "(--fixed-prefix-bits 0)"}

    def step(i, timed, gather):
        buf = i & 1
        if gather is not None:
            gather[buf].finish()  # the collective that last read d_out[buf] has completed
        if kind == "var":
            ctx.batch_mul_dev(cid, d_in[0].data_ptr(), d_in[1].data_ptr(), d_out[buf].data_ptr(), n, stream)
        elif kind == "fixed":
            ctx.batch_mul_fixed_dev(cid, d_in[0].data_ptr(), ctx.generator_dev(cid), d_out[buf].data_ptr(), n, stream)
        else:
            ctx.batch_double_mul_dev(cid, d_in[0].data_ptr(), d_in[1].data_ptr(), d_in[2].data_ptr(),
                                     d_out[buf].data_ptr(), n, stream)
        ms = None
        if timed:
            ms, launched["name"] = ctx.last_kernel_ms()  # HIP events on the launch stream (syncs this launch)
        if gather is not None:
            gather[buf].start(d_out[buf])
        return ms

    def barrier():
        if dist is not None:
            dist.barrier()
        plat.synchronize()

    def timed_region(mode):
        """W untimed warm-up steps, then exactly K steps bracketed by barrier + synchronize; MAX over ranks."""
        gather = gathers.get(mode)
        ctx.set_timing(False)
        for i in range(args.warmup):
            step(i, False, gather)
        if gather is not None:
            for g in gather:
                g.finish()
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i, False, gather)
        if gather is not None:
            for g in gather:
                g.finish()
        barrier()
        elapsed = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device=plat.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed

    elapsed_by_mode = {m: timed_region(m) for m in modes}
    elapsed = elapsed_by_mode[modes[0]]
    # the *_dev launches only enqueue: a fault a kernel reported is visible now (the host-pointer calls check themselves)
    ctx.check()

    # kernel duration from HIP events on the launch stream (separate passes, not in `value`)
    gather0 = gathers.get(modes[0])
    ctx.set_timing(True)
    kms = [step(i, True, gather0) for i in range(min(args.steps, 5))]
    if gather0 is not None:
        for g in gather0:
            g.finish()
    ctx.set_timing(False)
    plat.synchronize()
    if gather0 is not None and os.environ.get("FEC_BENCH_CHECK_GATHER") == "1":
        # rehearsal aid: the gathered block of this rank must equal the shard the kernel just produced
        step(0, False, gather0)
        full = gather0[0].finish()
        plat.synchronize()
        off = lo if strong else rank * n
        if full is not None and not torch.equal(full[off:off + n], d_out[0]):
            raise SystemExit("gathered shard differs from the kernel output")
    kernel_ms = float(np.mean(kms))
    clocks_under_load = None
    if clock_sampler is not None:
        # about a second of the kernel back to back (outside every timed region) while the helper asks rocm-smi for the clocks
        with clock_sampler as clocks:
            t_end = time.perf_counter() + 1.2
            i = 0
            while time.perf_counter() < t_end:
                step(i, False, None)
                plat.synchronize()
                i += 1
        clocks_under_load = clocks.summary()
    peak_measured = ctx.measure_peak_mad32()
    info = ctx.device_info()

    if rank == 0:
        total = n_global * args.steps
        value = total / elapsed
        achieved = n * alg / (kernel_ms * 1e-3)
        kname = launched["name"]  # as the library names the launch (the rocprofv3 kernel-stats row has the same stem)
        pmc = committed_pmc(workload, n)
        cpu = None
        if not args.no_cpu_baseline and world == 1:  # the CPU leg is an N = 1 measurement
            gpu_out = d_out[(min(args.steps, 5) - 1) & 1].cpu().numpy().view(np.uint64)
            cpu = cpu_baseline(workload, inputs, gpu_out, args.cpu_seconds)
            if not cpu["parity_sample_bit_exact"]:
                # a kernel whose output differs from the oracle has no throughput worth reporting
                print(json.dumps({"error": "GPU output differs from the CPU oracle on the parity sample",
                                  "workload": workload, "cpu_baseline": cpu}), flush=True)
                raise SystemExit(3)
        base_cfgs = json.load(open(os.path.join(ROOT, "BASELINE.json")))["configs"]
        if strong:
            quoted = {"p256-var": base_cfgs[3], "secp256k1-double": base_cfgs[4]}.get(workload)
            wl = ("%s -- run as ONE global batch of 2^%d split into %d contiguous shard(s) of %d"
                  % (quoted if quoted and n_global == 1 << BASELINE_LOG2_GLOBAL[workload] else
                     "2^%d %s scalar-muls (global)" % (n_global.bit_length() - 1, workload), n_global.bit_length() - 1, world, n))
        else:
            wl = "2^%d %s scalar-muls per GPU per step (BASELINE.json %s)" % (args.log2_batch, workload, cfg)
        if args.scalars_below_l and curve == "ed25519":
            wl += "; scalars below 2^252 < l"
        bits = prefix["bits"] if prefix else 0
        # Fixed-base and double workloads: `frac` prices the REFERENCE's operation count, but the first `bits` steps of
        # multiply(G, .) come out of the prefix table.  frac_executed_steps scales it by the share of the steps (Ed25519:
        # of the additions) the kernels still execute -- the number to compare with a variable-base row.
        executed_share = None
        if kind == "fixed" and curve == "ed25519":
            executed_share = max(mean_popcount - bits / 2.0, 0.0) / mean_popcount if mean_popcount else None
        elif kind == "fixed":
            executed_share = (256 - bits) / 256.0
        elif kind == "double":
            executed_share = (512 - bits) / 512.0
        table_fetch = ({0: 192, 1: 96, 2: 128}[cid] if bits else 0) if kind in ("fixed", "double") else 0
        # The lock-step ladder (secp256k1) runs one wavefront per 64 elements, three resident per SIMD: a launch whose
        # wavefronts are not a whole number of rounds of the chip's resident slots pays for the thin last round (a lone
        # wavefront issues at ~4.6 cycles per instruction, three at ~3.8 each).  This is the term that bounds a small
        # per-GPU shard (BASELINE configs[4] on 8 GPUs: 2^17 per GPU, two launches side by side).
        rounds = None
        if curve == "secp256k1":
            slots = info["compute_units"] * 4 * 3
            waves = ((n + 63) // 64) * (2 if kind == "double" else 1)
            full, rest = divmod(waves, slots)
            per_simd_last = rest / float(info["compute_units"] * 4)
            rounds = {"wavefronts": waves, "resident_wavefront_slots": slots, "rounds": waves / float(slots),
                      "last_round_wavefronts_per_simd": per_simd_last,
                      "bound": ("whole rounds of three wavefronts per SIMD" if rest == 0 else
                                "%d full round(s) at three wavefronts per SIMD + a last round at %.2f per SIMD: one ladder "
                                "(256 steps) takes ~2.8 ms alone on a SIMD and ~5.4 ms for three, so the time is set by rounds "
                                "of resident wavefronts, not by the element count" % (full, per_simd_last))}
        out = {
            "metric": "%s scalar-muls/sec (batched, %s)" % (
                workload, "bit-exact vs CPU oracle on the parity sample" if cpu else "parity check not run in this invocation"),
            "value": value, "unit": "scalar-muls/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": wl,
                       "curve": curve, "kind": kind, "batch_per_gpu": n, "global_batch": n_global,
                       "result_gather": ("n/a" if dist is None else modes[0] + (
                           " (this backend has no gather: all-gather used)" if modes[0] == "rank0" and gather0 is not None and gather0[0].dst is None else "")),
                       "device": info["name"], "compute_units": info["compute_units"]},
            "roofline": {
                "bound": "int-valu", "achieved": achieved / 1e12, "peak": PEAK_MAD32_FORMULA / 1e12,
                "unit": "TMAD32/s", "frac": achieved / PEAK_MAD32_FORMULA,
                "frac_executed_steps": (achieved / PEAK_MAD32_FORMULA * executed_share) if executed_share is not None else None,
                "executed_share_of_reference_steps": executed_share,
                "wavefront_rounds": rounds,
                "clocks_under_load": clocks_under_load,
                "traffic": pmc["traffic"],
                "traffic_source": pmc["source"],
                "traffic_note": TRAFFIC_NOTES.get(workload),
                "kernel": kname, "kernel_ms": kernel_ms,
                "algorithmic_mad32_per_unit": alg, "mean_scalar_popcount": mean_popcount, "executed_mul_insts_per_unit": EXECUTED_MULS[workload](bits),
                "fixed_base_prefix_table": prefix,
                "units_per_launch": n,
                "peak_measured": peak_measured / 1e12, "frac_of_measured_peak": achieved / peak_measured,
                "valu_issue_cycles_per_inst_per_simd": pmc["valu_issue_cycles"],
                "hbm": {"achieved_GBps": n * (hbm_bytes + table_fetch) / (kernel_ms * 1e-3) / 1e9, "peak_GBps": 8000.0,
                        "algorithmic_bytes_per_unit": hbm_bytes,
                        "prefix_table_fetch_bytes_per_unit": table_fetch},
            },
        }
        if dist is not None:
            # the gather's cost, both ways in one line: throughput with the result shards gathered
            # to rank 0 (overlapped with the next step's kernel) and with the shards left where they are
            out["gather"] = {m: {"value": total / elapsed_by_mode[m], "ms_per_step": elapsed_by_mode[m] / args.steps * 1e3}
                             for m in modes}
        if cpu:
            out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

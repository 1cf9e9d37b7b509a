timeout -k 10 300 python bench.py --workload ed25519-var --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_r04i_edvar.json 2> gpurun_out/bench_r04i_edvar.err || { tail -5 gpurun_out/bench_r04i_edvar.err; exit 1; }
python - <<'PY'
import json
t = json.loads([l for l in open("gpurun_out/bench_r04i_edvar.json") if l.startswith("{")][0])
print(t["value"], t["roofline"]["kernel_ms"], t["roofline"]["clocks_under_load"])
PY
ls /sys/bus/pci/devices/*/pp_dpm_sclk 2>/dev/null | head -3
timeout -k 10 600 python tools/next_rows_perf.py 20 > gpurun_out/r04_next_rows_perf.jsonl 2>&1 || { tail -5 gpurun_out/r04_next_rows_perf.jsonl; exit 1; }
grep -E "ecdsa_verify|ECDH|eddsa|multi_scalar|schnorr" gpurun_out/r04_next_rows_perf.jsonl | cut -c1-200

#!/bin/bash
# Summarise the passes of tools/pmc_refresh.sh into profiles/pmc_r04/<workload>/ (run in the container, sources
# unchanged since the passes were taken: the summary records the kernel-source hash bench.py matches against).
set -eu
TAG=${1:?tag}
python tools/pmc_summarize.py gpurun_out/pmc_${TAG}_secp "k_secp_mul<0>" --workload secp256k1-var --copy-to profiles/pmc_r04/secp256k1-var > /dev/null
python tools/pmc_summarize.py gpurun_out/pmc_${TAG}_p256 k_p256_mul_sched --workload p256-var --copy-to profiles/pmc_r04/p256-var > /dev/null
python tools/pmc_summarize.py gpurun_out/pmc_${TAG}_edvar k_ed_mul_pers --workload ed25519-var --copy-to profiles/pmc_r04/ed25519-var > /dev/null
python tools/pmc_summarize.py gpurun_out/pmc_${TAG}_edfixed k_ed_fixed_sorted --workload ed25519-fixed --copy-to profiles/pmc_r04/ed25519-fixed > /dev/null
python tools/pmc_summarize.py gpurun_out/pmc_${TAG}_secpfixed "k_secp_mul<2>" --workload secp256k1-fixed --copy-to profiles/pmc_r04/secp256k1-fixed > /dev/null
python -c "
import bench
for w in ('secp256k1-var','p256-var','ed25519-var','ed25519-fixed','secp256k1-fixed'): print(w, bench.committed_pmc(w, 1<<20)['traffic'])"

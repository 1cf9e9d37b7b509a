#!/bin/bash
# Re-take the PMC + kernel-stats passes of the four dominant kernels (run ON the GPU box from the repo root);
# summarise afterwards in the container with tools/pmc_refresh_summarize.sh.
set -u
TAG=${1:?tag}
bash tools/pmc_passes.sh secp256k1-var gpurun_out/pmc_${TAG}_secp &&
bash tools/pmc_passes.sh p256-var gpurun_out/pmc_${TAG}_p256 &&
bash tools/pmc_passes.sh ed25519-var gpurun_out/pmc_${TAG}_edvar &&
bash tools/pmc_passes.sh ed25519-fixed gpurun_out/pmc_${TAG}_edfixed &&
bash tools/pmc_passes.sh secp256k1-fixed gpurun_out/pmc_${TAG}_secpfixed && echo PMCDONE

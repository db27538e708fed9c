#!/bin/bash
# Records PyTorch TunableOp's GEMM selections for the BASELINE configurations on the GPU box (run through gpurun);
# copy gpurun_out/tuned/gemm0.csv to mm-unet_amd/tuned/gemm_gfx950.csv afterwards (mm-unet_amd/tuned_gemms.py reads it).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/tuned
rm -f gpurun_out/tuned/gemm.csv
export PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_FILENAME=$PWD/gpurun_out/tuned/gemm.csv
common="--no-cpu-baseline --no-roofline --steps 3 --warmup 2"
python3 bench.py --gpus 1 $common > gpurun_out/tuned/c1.json 2> gpurun_out/tuned/c1.err || exit 1
python3 bench.py --infer $common > gpurun_out/tuned/c2.json 2> gpurun_out/tuned/c2.err || exit 1
# (float32 only: tuning the bfloat16 shapes of configs 3 / 5 faulted the GPU inside a candidate library kernel -- do not add them)
ls -la gpurun_out/tuned/; wc -l gpurun_out/tuned/gemm*.csv

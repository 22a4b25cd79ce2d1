set -o pipefail
for t in VQAE_NO_WINOGRAD VQAE_NO_TRUNK_FUSION VQAE_NO_UP_TAIL_FUSION VQAE_NO_SMALL_K VQAE_C8_MFMA VQAE_NO_UP_REORDER VQAE_NO_DOWN_FUSION; do
  env $t=1 timeout -k 10 600 python -m pytest tests/test_model_gpu.py tests/test_autocast_gpu.py tests/test_driver_gpu.py -x -q -k "not winograd and not fusion" > gpurun_out/tog_$t.log 2>&1
  echo "$t rc=$? $(tail -1 gpurun_out/tog_$t.log)"
done
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 2 --warmup 1 --batch 512 2>&1 | tail -1 | cut -c1-120

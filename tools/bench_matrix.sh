#!/bin/bash
# Secondary configurations of DESIGN.md section 5 (one GPU): prints "<label> <patches/s> <ms/step>".
run() { label=$1; shift
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" > gpurun_out/bm_$label.log 2>&1
  python - "$label" <<'PY'
import json, sys
lab = sys.argv[1]
try:
    l = [x for x in open(f'gpurun_out/bm_{lab}.log') if x.startswith('{')][-1]
    d = json.loads(l); print(lab, d['value'], d['ms_per_step'])
except Exception as e:
    print(lab, 'FAILED', e)
PY
}
run B_encode --mode encode
run B_bf16 --dtype bf16
run B_f16 --dtype f16
run A_f32 --config A --batch 64
run A_bf16 --config A --dtype bf16 --batch 64
run C_f32 --config C --batch 64
run C_f16 --config C --dtype f16 --batch 256
run C_f16_encode --config C --dtype f16 --batch 256 --mode encode
run BM_f32 --config BM --batch 64

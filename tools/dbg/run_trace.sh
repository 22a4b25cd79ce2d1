#!/bin/bash
mkdir -p gpurun_out/r03
L=$PWD/2d-vq-ae-2_amd/build/var/libvqae_trace.so
for st in ${STAGS:-0 14}; do
  VQAE_W43_STAG=$st VQAE_HIP_LIB=$L timeout -k 10 300 python3 tools/dbg/w43_trace.py > gpurun_out/r03/w43_trace_s$st.log 2>&1 || { tail -5 gpurun_out/r03/w43_trace_s$st.log; exit 1; }
  echo "=== stag $st"; tail -${LINES_OUT:-12} gpurun_out/r03/w43_trace_s$st.log | cut -c1-420
done

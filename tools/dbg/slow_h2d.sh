#!/bin/bash
sed -i 's/loader="ring", encode_batch/loader="auto", encode_batch/' tools/dbg/slow_h2d.py
for w in "" "pinalloc"; do
  timeout -k 10 200 python tools/dbg/slow_h2d.py $w 2>&1 | grep -E "patches/s|Warning|warn" | cut -c1-260
done

#!/bin/bash
cd /root/repo
for lib in "" ""; do
  if [ -n "$lib" ]; then export PHA_LIB_OVERRIDE=$PWD/$lib; else unset PHA_LIB_OVERRIDE; fi
  python bench.py --steps 20 --warmup 5 --no-c5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('lib=${lib:-product}', round(d['keyswitch_c4']['value']), round(d['hommul_relin_rescale']['gpu_ms_per_op']['mean_ms'],4), round(d['hommul_relin_rescale']['batched']['ms_per_op'],4), round(d['next_rows']['bfv_multiply']['behz_multiply_ms'],3), round(d['next_rows']['bfv_multiply']['hps_multiply_ms'],3))"
done

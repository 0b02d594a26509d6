#!/bin/bash
cd /root/repo
for lib in "" phantom-fhe_amd/phantom_fhe_amd/libphantom_amd_nolanes.so "" phantom-fhe_amd/phantom_fhe_amd/libphantom_amd_nolanes.so; do
  if [ -n "$lib" ]; then export PHA_LIB_OVERRIDE=$PWD/$lib; else unset PHA_LIB_OVERRIDE; fi
  python bench.py --steps 20 --warmup 5 --no-c5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('lib=${lib:-product}', 'batched', round(d['hommul_relin_rescale']['batched']['ms_per_op'],4), 'single', round(d['hommul_relin_rescale']['gpu_ms_per_op']['mean_ms'],4), 'c4', round(d['keyswitch_c4']['value']))"
done
python -m pytest tests/test_gpu_rns.py -q -m gpu -x -k "rescale" 2>&1 | tail -2

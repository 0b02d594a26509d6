#!/bin/bash
# f3 / f4 rows: timings and kernel traces of the BFV multiply variants and of the batched modular GEMM
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
python tools/time_gemm.py > $O/r03m_gemm.txt 2>&1
python tools/time_bfv_mul.py > $O/r03m_bfvmul.txt 2>&1
cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/pg -o g -- python /root/repo/tools/time_gemm.py > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/pb -o b -- python /root/repo/tools/time_bfv_mul.py > /dev/null 2>&1
cd /root/repo
python tools/summarize_prof.py /tmp/pg > $O/r03m_gemm_kernels.txt 2>&1 || cp $(find /tmp/pg -name '*kernel_stats.csv' | head -1) $O/r03m_gemm_kernel_stats.csv
python tools/summarize_prof.py /tmp/pb > $O/r03m_bfvmul_kernels.txt 2>&1 || cp $(find /tmp/pb -name '*kernel_stats.csv' | head -1) $O/r03m_bfvmul_kernel_stats.csv
cp $(find /tmp/pb -name '*kernel_stats.csv' | head -1) $O/r03m_bfvmul_kernel_stats.csv 2>/dev/null
cp $(find /tmp/pg -name '*kernel_stats.csv' | head -1) $O/r03m_gemm_kernel_stats.csv 2>/dev/null
cat $O/r03m_gemm.txt $O/r03m_bfvmul.txt

#!/bin/bash
# C++ host mirror with device-resident ciphertexts: tests, then multiply_and_relin/s through the fHE::Evaluator API at C3 and C5
set -o pipefail
OUT=gpurun_out/${1:-hostapi}; mkdir -p $OUT
timeout -k 10 600 python -u -m pytest tests/test_gpu_host_mirror.py -m gpu -x -q --timeout 300 --timeout-method thread > $OUT/tests.log 2>&1 || { tail -40 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
B=fhe_amd/host/bin/host_roundtrip
timeout -k 10 300 $B bench 15 16 1 64 5 > $OUT/api_c5.json 2> $OUT/api_c5.err || { tail $OUT/api_c5.err; exit 1; }
cat $OUT/api_c5.json
timeout -k 10 400 $B bench 16 30 1 16 5 > $OUT/api_c3.json 2> $OUT/api_c3.err || { tail $OUT/api_c3.err; exit 1; }
cat $OUT/api_c3.json

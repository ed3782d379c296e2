#!/bin/bash
# Round 5, first GPU contact of the call-long role S: small parity tests, then same-box timing of the three chain forms.
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_timed_path_gpu.py -x -q -m gpu -k "bench_call_matches and (64 or 256)" 2>&1 | tail -15
timeout 900 python -m pytest tests/test_timed_path_gpu.py -x -q -m gpu -k "odd_call_lengths or chain_timeout or placement_independent" 2>&1 | tail -15
export INFV_LTM_LIBRARY=exp
for r in 1 2; do
  echo "-- call-long + mailboxes (default)";  timeout 300 python tools/one_pass.py 2048 5 2>&1 | grep "^pass" | tail -3 | tr '\n' ' '; echo
  echo "-- per-sub-batch + mailboxes";        INFV_CHAIN_CALL=0 timeout 300 python tools/one_pass.py 2048 5 2>&1 | grep "^pass" | tail -3 | tr '\n' ' '; echo
  echo "-- per-sub-batch + atomics (round 4)"; INFV_CHAIN_CALL=0 INFV_CHAIN_XCD=0 timeout 300 python tools/one_pass.py 2048 5 2>&1 | grep "^pass" | tail -3 | tr '\n' ' '; echo
  echo "-- call-long + atomics";              INFV_CHAIN_XCD=0 timeout 300 python tools/one_pass.py 2048 5 2>&1 | grep "^pass" | tail -3 | tr '\n' ' '; echo
done
echo "-- shard 256: default / round-4 form"
timeout 300 python tools/one_pass.py 256 8 2>&1 | grep "^pass" | tail -4 | tr '\n' ' '; echo
INFV_CHAIN_CALL=0 INFV_CHAIN_XCD=0 timeout 300 python tools/one_pass.py 256 8 2>&1 | grep "^pass" | tail -4 | tr '\n' ' '; echo
unset INFV_LTM_LIBRARY
timeout 600 bash tools/quick_bench.sh r05a 10

#!/bin/bash
# Round-4 evidence for the role-S step (profiles/r04_stamps.txt, profiles/r04_ubench_exchange.txt): in-kernel phase stamps and the
# per-launch average step of chain_batch3_kernel -- atomics exchange (shipped) and XCD-local mailboxes (INFV_CHAIN_XCD=1) -- alone
# (no pooling / GEMM / UC launches) and in situ, then the exchange / scan micro-benchmarks.
export INFV_LTM_LIBRARY=exp
run() { # label, env...
  local label=$1; shift
  echo "== $label"
  env "$@" INFV_CHAIN_STAMPS=1 python tools/one_pass.py 2048 4 2>&1 | grep -E "batch-S stamps|batch-S avg|pass 3" | tail -7
}
{
echo "# stamps x10ns of workgroup 0 at step 5 of a launch: wait+read = loop top -> probabilities normalised (exchange wait, partial sums,"
echo "# two wave reductions), draw = scan + cdf + barrier 1, tab = search + barrier 2, recurrence = row phase + barrier 3,"
echo "# row-phase+add = row sums + deposit + poll issue;  [batch-S avg] = (top of last step - top of step 1) / (n - 2) of a launch"
run "atomics exchange (shipped form), alone" INFV_SKIP=7
run "atomics exchange (shipped form), in situ"
run "XCD-local mailboxes, alone" INFV_CHAIN_XCD=1 INFV_SKIP=7
run "XCD-local mailboxes, in situ (the pass time shows the placement wait of the XCD-aware launches)" INFV_CHAIN_XCD=1
run "mailboxes written sc1 (placement handshake overridden: the cross-XCD form of the same exchange), alone" INFV_CHAIN_XCD=1 INFV_S_FLAGS=32 INFV_SKIP=7
run "8-row tiles (96 workgroups), atomics, alone" INFV_CHAIN_RPW=1 INFV_SKIP=7
run "8-row tiles (96 workgroups), atomics, in situ" INFV_CHAIN_RPW=1
} > gpurun_out/r04_stamps.txt 2>&1
./tools/ubench/ubench_exchange > gpurun_out/r04_ubench_exchange.txt 2>&1
tail -5 gpurun_out/r04_stamps.txt; tail -3 gpurun_out/r04_ubench_exchange.txt

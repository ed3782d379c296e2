mkdir -p gpurun_out
for m in 7 1 2 4 3; do
INFV_CHAIN_ROLES=$m timeout 300 python bench.py --steps 2 --warmup 1 --chunks 512 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('roles=$m', round(d['ms_per_step'],2), d['roofline']['kernel_ms_per_pass'])"
done

#!/bin/bash
export INFV_LTM_LIBRARY=${INFV_LTM_LIBRARY:-exp}   # experiment knobs passed in by the caller only exist in the experiments build (csrc/knobs.h)
# usage (on the GPU box): tools/quick_bench.sh <tag> [steps]  -- short headline bench line + per-kernel ms, no CPU legs
tag=$1; steps=${2:-10}
python bench.py --steps $steps --warmup 3 --no-cpu-baseline --no-encode-video --no-secondary > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err
python - <<PY
import json
d = json.load(open("gpurun_out/bench_$tag.json"))
print("$tag", round(d["value"]), "chunks/s", round(d["ms_per_step"], 3), "ms/video; pool frac", round(d["roofline"]["frac"], 3),
      d["roofline"]["kernel_ms_per_pass"], "selfcheck", d.get("selfcheck_max_abs_err"), "shard256", d.get("shard256_ms"))
PY

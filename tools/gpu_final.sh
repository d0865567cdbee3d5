#!/bin/bash
# final bench lines of the round, AFTER tools/collect_evidence.sh has put the counter files of this library build under
# profiles/ (bench.py replays them when the source hash matches): ROUND=r06 bash tools/gpu_final.sh -> gpurun_out/$ROUND_final/
ROUND=${ROUND:-r06}; G=gpurun_out/${ROUND}_final; mkdir -p $G
timeout 600 python bench.py > $G/bench.log 2>/dev/null
timeout 600 python bench.py --steps 20 --warmup 5 > $G/bench_driver_like.log 2>/dev/null
timeout 600 python bench.py --shard rows --molecule Li2O --samples 50000 --steps 100 --warmup 10 > $G/bench_li2o.log 2>/dev/null
ROUND=$ROUND bash tools/n2_sweep.sh > $G/n2_sweep.log 2>&1
cp gpurun_out/$ROUND/n2_sweep.txt $G/n2_sweep.txt
python - <<PY
import json
for f in ("bench.log", "bench_driver_like.log", "bench_li2o.log"):
    d = json.loads(open("$G/" + f).read().strip().splitlines()[-1])
    o = d["roofline"]["other_kernels"][0] if d["roofline"]["bound"] == "mfma" else d["roofline"]
    print(f, round(d["value"] / 1e6, 1), "M/s", round(d["ms_per_step"] * 1e3, 2), "us | traffic", d["roofline"].get("traffic"), "| eloc issue frac", o.get("frac"))
PY

#!/bin/bash
R=$PWD
python -m pytest tests/test_nade_gpu.py tests/test_variants_gpu.py tests/test_optimizer_gpu.py -m gpu -q -x 2>&1 | tail -15 > gpurun_out/r02_pytest5.log
NAQS_DEBUG_CLOCKS=1 python tools/clock_probe.py > gpurun_out/r02_clocks5.log 2>&1
python bench.py --no-cpu-baseline --no-config4 > gpurun_out/r02_bench5.log 2>&1
python bench.py --no-cpu-baseline --no-config4 --pipeline 1 > gpurun_out/r02_bench5s.log 2>&1
tail -3 gpurun_out/r02_pytest5.log; tail -9 gpurun_out/r02_clocks5.log; python - <<'PY'
import json
for f in ("gpurun_out/r02_bench5.log","gpurun_out/r02_bench5s.log"):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    r=d["roofline"]; o=r["other_kernels"][0]
    print(f, round(d["value"]/1e6,1), "M/s", round(d["ms_per_step"]*1e3,2), "us/step;", r["kernel"][:20], round(r["kernel_us"],1), o["kernel"], round(o["kernel_us"],1), d.get("serial"))
PY

#!/bin/bash
python - <<'PY'
import json, os, subprocess, sys
def run(env, extra=()):
    e = dict(os.environ, **env)
    r = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--no-config4", "--steps", "400", *extra], env=e, capture_output=True, text=True)
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])
    except Exception:
        print(env, extra, "FAILED", r.stderr[-500:]); return
    rf = d["roofline"]; o = rf["other_kernels"][0]
    ph, el = (rf, o) if "phase" in rf["kernel"] else (o, rf)
    s = d.get("serial") or {}
    print(env, extra, f"{d['value']/1e6:7.1f} M/s {d['ms_per_step']*1e3:6.2f} us/step | pipelined phase {ph['kernel_us']:.1f} eloc {el['kernel_us']:.1f} | serial step {s.get('ms_per_step',0)*1e3:.1f} phase {s.get('logpsi_kernel_us',0):.1f} eloc {s.get('eloc_kernel_us',0):.1f}", flush=True)
for env in ({}, {"NAQS_STAGE": "1"}, {"NAQS_STAGE": "0"}, {"NAQS_STAGE": "1", "NAQS_BLOCK": "256"}, {"NAQS_STAGE": "1", "NAQS_BLOCK": "512"}, {"NAQS_BLOCK": "256"}, {"NAQS_BLOCK": "512"}):
    run(env)
run({}, ("--pipeline", "3"))
run({"NAQS_STAGE": "1"}, ("--pipeline", "3"))
PY

#!/bin/bash
# Per-kernel durations of the one-call training step (rocprofv3 --kernel-trace --stats over tools/train_loop_profile.py, 340 steps),
# interleaved over two rounds, for A/B decisions that a wall-clock loop on a shared box cannot resolve (boxes of the pool differ by
# several per cent from run to run; a kernel's own duration does not):
#   tools/train_kernel_times.sh env "NAQS_FUSE_SUMS=0" "NAQS_FUSE_SUMS=1"          environment settings of the in-tree library
#   tools/train_kernel_times.sh lib build/ab/libnaqs_a.so build/ab/libnaqs_b.so      library builds (tools/build_variant.sh)
# MOL=H2O selects the molecule (default N2); KERNELS="a|b" the kernel-name patterns to list (default: all above 4 us).
mode=$1; shift
R=$PWD; cd /tmp; export TMPDIR=/tmp
for rep in 1 2; do
for setting in "$@"; do
  (
  if [ "$mode" = lib ]; then export NAQS_LOADER_LAX=1 NAQS_HIP_LIB=$R/$setting; else for e in $setting; do export $e; done; fi
  rm -rf /tmp/tkt
  timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tkt -o t -- python3 $R/tools/train_loop_profile.py $R/tests/golden/ham_${MOL:-N2}.npz 1000000 300 40 > /tmp/tkt.log 2>&1
  f=$(find /tmp/tkt -name "t_kernel_stats.csv" | head -1)
  echo "== $setting: $(grep 'ms/step' /tmp/tkt.log | tail -1)"
  KERNELS="$KERNELS" python3 - "$f" <<'PY'
import csv, os, re, sys
pat = os.environ.get("KERNELS") or ""
tot = 0.0
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]; tot += float(r["TotalDurationNs"])
    avg = float(r["AverageNs"]) / 1000
    if "at::native" in n or int(r["Calls"]) < 10:
        continue
    if (pat and re.search(pat, n)) or (not pat and avg > 4.0):
        print("   %-64s calls %6s avg %7.2f us" % (n.replace("(anonymous namespace)::", "")[:64], r["Calls"], avg))
print("   all kernels: %.1f us per step (340 steps)" % (tot / 340e3))
PY
  )
done
done

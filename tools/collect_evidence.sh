#!/bin/bash
# copy what tools/gpu_evidence_pass.sh left under gpurun_out/$SRC into profiles/ as round $ROUND (run in the repo root):
#   SRC=r06 ROUND=r06 bash tools/collect_evidence.sh
ROUND=${ROUND:-r06}; G=gpurun_out/${SRC:-$ROUND}; P=profiles
for c in fetch write issue_n2 issue_li2o wait_n2 mfma_n2; do cp $G/pmc_$c/bench_counter_collection.csv $P/${ROUND}_pmc_${c}_counter_collection.csv; done
cp $G/prof_pipeline2/bench_kernel_stats.csv $P/${ROUND}_bench_n2_10k_kernel_stats_pipeline2.csv
cp $G/prof_serial/bench_kernel_stats.csv $P/${ROUND}_bench_n2_10k_kernel_stats_serial.csv
cp $G/prof_li2o/bench_kernel_stats.csv $P/${ROUND}_bench_li2o_50k_rowshard_kernel_stats.csv
cp $G/prof_train/train_kernel_stats.csv $P/${ROUND}_train_step_n2_kernel_stats.csv
cp $G/train_step_timing.txt $P/${ROUND}_train_step_timing.txt
cp $G/step_timeline_train.txt $P/${ROUND}_train_step_n2_timeline.txt
cp $G/step_timeline_train_h2o.txt $P/${ROUND}_train_step_h2o_timeline.txt
cp $G/train_onecall_ab.txt $P/${ROUND}_train_step_onecall_ab.txt
cp $G/train_scaling_n2.json $P/${ROUND}_train_step_scaling_model_n2.json
cp $G/train_scaling_li2o_pub.json $P/${ROUND}_train_step_scaling_model_li2o_published.json
[ -f $G/pmc_phase_mem.txt ] && grep -v "^\[" $G/pmc_phase_mem.txt > $P/${ROUND}_pmc_phase_mem.txt
cp $G/source_hash.txt $P/${ROUND}_library_source_hash.txt
tail -4 $G/pytest.log > $P/${ROUND}_gpu_test_suite.txt; tail -2 $G/smoke.log >> $P/${ROUND}_gpu_test_suite.txt
cp $G/n2_sweep.txt $P/${ROUND}_n2_sweep.txt
rm -f $P/${ROUND}_pmc_traffic.json $P/${ROUND}_pmc_issue.json
python tools/collect_pmc.py traffic $G/pmc_fetch $G/pmc_write $P/${ROUND}_pmc_traffic.json > /dev/null
python tools/collect_pmc.py issue $G/pmc_issue_n2 N2_10000 $P/${ROUND}_pmc_issue.json > /dev/null
python tools/collect_pmc.py issue $G/pmc_issue_li2o Li2O_50000 $P/${ROUND}_pmc_issue.json > /dev/null
for pair in "H2O:h2o" "N2:n2" "N2_default:n2_default" "Li2O:li2o" "N2_1.95:n2_1.95_fullmask" "N2_2.25:n2_2.25_fullmask"; do
  m=${pair%%:*}; n=${pair##*:}
  cp $G/train_${m}_summary.txt $P/${ROUND}_${n}_sto3g_training_summary.txt
  (head -12 $G/train_$m.log; echo "..."; tail -14 $G/train_$m.log) > $P/${ROUND}_${n}_sto3g_training_log_excerpt.txt
done
python - <<PY
import json
def last_json(p):
    return json.loads(open(p).read().strip().splitlines()[-1])
for src, dst in (("$G/bench.log", "$P/${ROUND}_bench_n2_10k.json"), ("$G/bench_driver_like.log", "$P/${ROUND}_bench_n2_10k_steps20.json"),
                 ("$G/bench_li2o.log", "$P/${ROUND}_bench_li2o_50k_rowshard.json"),
                 ("$G/bench_dist1.log", "$P/${ROUND}_bench_n2_10k_rccl_world1.json"),
                 ("$G/emulate_li2o.log", "$P/${ROUND}_config4_scaling_model.json")):
    d = last_json(src)
    json.dump(d, open(dst, "w"), indent=1)
    print(dst, round(d["value"] / 1e6, 1), "M/s", round(d["ms_per_step"] * 1e3, 2), "us/step")
PY

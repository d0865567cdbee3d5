#!/bin/bash
# copy what tools/gpu_evidence_pass.sh + tools/gpu_final.sh left under gpurun_out/ into profiles/ (run in the repo root)
G=gpurun_out; P=profiles
cp $G/pmc_fetch/bench_counter_collection.csv $P/r02_pmc_fetch_counter_collection.csv
cp $G/pmc_write/bench_counter_collection.csv $P/r02_pmc_write_counter_collection.csv
cp $G/pmc_issue_n2/bench_counter_collection.csv $P/r02_pmc_issue_n2_counter_collection.csv
cp $G/pmc_issue_li2o/bench_counter_collection.csv $P/r02_pmc_issue_li2o_counter_collection.csv
cp $G/pmc_wait_n2/bench_counter_collection.csv $P/r02_pmc_wait_n2_counter_collection.csv
cp $G/pmc_mfma_n2/bench_counter_collection.csv $P/r02_pmc_mfma_n2_counter_collection.csv
cp $G/prof_r02/bench_kernel_stats.csv $P/r02_bench_n2_10k_kernel_stats_pipeline2.csv
cp $G/prof_r02_serial/bench_kernel_stats.csv $P/r02_bench_n2_10k_kernel_stats_serial.csv
cp $G/prof_r02_li2o/bench_kernel_stats.csv $P/r02_bench_li2o_50k_rowshard_kernel_stats.csv
cp $G/prof_train_r02/train_kernel_stats.csv $P/r02_train_step_n2_kernel_stats.csv
cp $G/r02_train_step_n2_timing.txt $P/r02_train_step_n2_timing.txt
python tools/collect_pmc.py traffic $G/pmc_fetch $G/pmc_write $P/r02_pmc_traffic.json > /dev/null
rm -f $P/r02_pmc_issue.json
python tools/collect_pmc.py issue $G/pmc_issue_n2 N2_10000 $P/r02_pmc_issue.json > /dev/null
python tools/collect_pmc.py issue $G/pmc_issue_li2o Li2O_50000 $P/r02_pmc_issue.json > /dev/null
for pair in "H2O:h2o" "N2:n2" "N2_default:n2_default" "Li2O:li2o"; do
  m=${pair%%:*}; n=${pair##*:}
  cp $G/r02_train_${m}_summary.txt $P/r02_${n}_sto3g_training_summary.txt
  (head -12 $G/r02_train_$m.log; echo "..."; tail -14 $G/r02_train_$m.log) > $P/r02_${n}_sto3g_training_log_excerpt.txt
done
python - <<'PY'
import json
def last_json(p):
    return json.loads(open(p).read().strip().splitlines()[-1])
for src, dst in (("gpurun_out/r02_bench_final.log", "profiles/r02_bench_n2_10k.json"),
                 ("gpurun_out/r02_bench_final_li2o.log", "profiles/r02_bench_li2o_50k_rowshard.json"),
                 ("gpurun_out/r02_bench_final_dist1.log", "profiles/r02_bench_n2_10k_rccl_world1.json")):
    d = last_json(src)
    json.dump(d, open(dst, "w"), indent=1)
    print(dst, round(d["value"] / 1e6, 1), "M/s", round(d["ms_per_step"] * 1e3, 2), "us/step")
PY

#!/bin/bash
# A/B of environment settings on the training step (tools/train_loop_profile.py), interleaved, 3 rounds; prints ms/step.
# usage: tools/train_ab.sh <molecule npz> "VAR=a" "VAR=b,VAR2=c" ...   ("-" = no setting)
mol=$1; shift
for rep in 1 2 3; do
for setting in "$@"; do
  (
  if [ "$setting" != "-" ]; then IFS=',' read -ra kv <<< "$setting"; for e in "${kv[@]}"; do export "$e"; done; fi
  echo "$setting: $(timeout 200 python tools/train_loop_profile.py $mol 1000000 300 40 2>&1 | grep "ms/step" | tail -1)"
  )
done
done

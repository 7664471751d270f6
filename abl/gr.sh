#!/bin/bash
# retry gpurun while the pod's GPU slots are busy (nothing is charged for those attempts)
# usage: abl/gr.sh <timeout_s> '<command>'
for i in $(seq 1 40); do
  out=$(gpurun --timeout "$1" -- "$2" 2>&1)
  if echo "$out" | grep -q "status=transient"; then sleep 45; continue; fi
  echo "$out"; exit 0
done
echo "gave up: no GPU slot"; exit 3

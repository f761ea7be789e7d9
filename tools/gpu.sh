#!/bin/bash
# usage: tools/gpu.sh <tag> <timeout_s> '<command>'   -- runs gpurun, retrying only while no box/slot is free (exit 3: nothing ran)
tag=$1; to=$2; shift 2
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $to -- "$@" > /root/repo/gpurun_out/call_$tag.log 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then echo "rc=$rc" >> /root/repo/gpurun_out/call_$tag.log; exit $rc; fi
  sleep 45
done
exit 3

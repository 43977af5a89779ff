#!/bin/bash
# Usage: tools/power_probe.sh <tag> <command...>: runs the command while sampling rocm-smi power / sclk twice a second.
tag=$1; shift
mkdir -p gpurun_out
"$@" > gpurun_out/probe_$tag.out 2>&1 &
pid=$!
: > gpurun_out/probe_$tag.smi
while kill -0 $pid 2>/dev/null; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|Power|fclk|mclk" | tr '\n' ' ' >> gpurun_out/probe_$tag.smi
  echo >> gpurun_out/probe_$tag.smi
  sleep 0.4
done
wait $pid

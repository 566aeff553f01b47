#!/bin/bash
# Samples rocm-smi (power, clocks) while a command runs; prints the samples.  Usage: tools/power_probe.sh <cmd...>
"$@" > gpurun_out/power_probe_cmd.log 2>&1 &
PID=$!
sleep 25
for i in $(seq 1 12); do
  if ! kill -0 $PID 2>/dev/null; then break; fi
  /opt/rocm/bin/rocm-smi --showpower --showclocks --showperflevel 2>/dev/null | grep -E "Power|sclk|mclk|fclk" | tr '\n' ';'
  echo
  sleep 1.5
done
wait $PID
tail -3 gpurun_out/power_probe_cmd.log | cut -c1-400

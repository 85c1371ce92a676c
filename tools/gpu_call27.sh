#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
python tools/config_probe.py 2>&1 | grep configs > gpurun_out/config_probe_r02.log; cat gpurun_out/config_probe_r02.log
cd /tmp
for k in 2 3 4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk$k -o c -- python $GRAFT_REPO_ROOT/tools/config_probe.py $k > /dev/null 2>&1
  f=$(find /tmp/pk$k -name "*kernel_stats.csv" | head -n 1)
  echo "== configs[$k] kernel stats (12 forward + step passes profiled)" >> $GRAFT_REPO_ROOT/gpurun_out/config_probe_r02.log
  head -n 9 $f | cut -c 1-230 >> $GRAFT_REPO_ROOT/gpurun_out/config_probe_r02.log
done
tail -n 32 $GRAFT_REPO_ROOT/gpurun_out/config_probe_r02.log | cut -c1-200

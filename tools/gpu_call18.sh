#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --tb=short -p no:cacheprovider --timeout 200 -k "fbank or wave or ragged" 2>&1 | tail -n 5
echo new; python tools/fbank_probe.py 20 2>&1 | tail -n 2
echo old; VPMI_FBANK_OLD=1 python tools/fbank_probe.py 20 2>&1 | tail -n 2
cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf -o fb -- python $GRAFT_REPO_ROOT/tools/fbank_probe.py 20 > /dev/null 2>&1; grep -E "fbank" /tmp/pf/*/fb_kernel_stats.csv | cut -c1-200

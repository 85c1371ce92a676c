export TMPDIR=/tmp
for d in ${DBGS:-0 1 2 3}; do
 (cd /tmp && VPMI_ASP_DBG=$d timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pa_$d -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --graph 0 --streams 1 --no-cpu-baseline --no-roofline --no-train-line > /tmp/pa_$d.log 2>&1)
 f=$(find /tmp/pa_$d -name "*kernel_stats.csv" | head -n 1); echo "dbg=$d: $(grep asp_utt $f | cut -d, -f2-4)"
done

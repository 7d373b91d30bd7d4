set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
rm -rf $O/prof && mkdir -p $O/prof
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof/stats -o st -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-runner > $O/prof/stats.log 2>&1
echo "stats done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/prof/pmc_fetch -o f -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-runner > $O/prof/f.log 2>&1
echo "fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/prof/pmc_write -o w -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-runner > $O/prof/w.log 2>&1
echo "write done"
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/prof/pmc_l2 -o l -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-runner > $O/prof/l.log 2>&1
echo "l2 done"
python3 tools/prof_summary.py $O/prof $O/prof_summary.json
cp "$(find $O/prof/stats -name "*kernel_stats.csv" | head -1)" $O/kernel_stats.csv
rm -rf $O/prof/stats $O/prof/pmc_fetch $O/prof/pmc_write $O/prof/pmc_l2
ls -la $O

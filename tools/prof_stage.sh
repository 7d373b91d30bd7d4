# usage (on the GPU box): bash tools/prof_stage.sh <stage> [<stage> ...]   stage = detector | encoder | decoder
# rocprofv3 kernel stats of ONE stage of the config-3 step run in a loop (tools/stage_times.py --only <stage>);
# the summary lands in gpurun_out/prof_<stage>_kernel_stats.csv
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for st in "$@"; do
  rm -rf $O/prof_$st && mkdir -p $O/prof_$st
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$st -o st -- python3 $R/tools/stage_times.py --only $st --iters 5 > $O/prof_$st.log 2>&1
  cp "$(find $O/prof_$st -name "*kernel_stats.csv" | head -1)" $O/prof_${st}_kernel_stats.csv
  python3 $R/tools/prof_trace.py "$(find $O/prof_$st -name "*kernel_trace.csv" | head -1)" 5 60 > $O/prof_${st}_table.txt
  rm -rf $O/prof_$st
  echo "$st done"
done

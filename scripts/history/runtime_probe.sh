export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for rt in system torch; do
  export GOMEL_HIP_RUNTIME=$rt
  python3 $R/scripts/h2h_trace.py 6
  (cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $R/gpurun_out/rt/$rt -- python3 $R/scripts/h2h_trace.py 1 > $R/gpurun_out/rt_$rt.log 2>&1)
  cat $R/gpurun_out/rt/$rt/*/*kernel_stats.csv | cut -d, -f1-4 | grep -i "copyBuffer\|mel_to_mag"
  cat $R/gpurun_out/rt/$rt/*/*memory_copy_stats.csv | cut -d, -f1-4 | grep MEMORY
done

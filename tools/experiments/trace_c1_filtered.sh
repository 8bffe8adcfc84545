cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/c1f
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/c1f -o t -- python3 $GRAFT_REPO_ROOT/tools/run_kernels.py c1 20 step filter > $GRAFT_REPO_ROOT/gpurun_out/c1f.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/trace_by_grid.py $GRAFT_REPO_ROOT/gpurun_out/c1f

cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c2 -o c2 -- python3 -m pytest tests/test_gpu_fullsize.py -x -q -k "hessian_identities and config2" > gpurun_out/c2.log 2>&1
tail -2 gpurun_out/c2.log
head -8 gpurun_out/c2/c2_kernel_stats.csv | cut -c1-170

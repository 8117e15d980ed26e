# rocprofv3 kernel stats of the native training step (tools/train_bench.py, torch baselines skipped)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TB_ITERS=3 TB_NATIVE_ONLY=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_train -- python3 tools/train_bench.py > gpurun_out/prof_train.log 2>&1
echo rc=$?
find gpurun_out/prof_train -name "*kernel_trace.csv" -delete
ls gpurun_out/prof_train/*/

# summaries for profiles/: native training step + full fine-tune iteration, with rocprofv3 kernel stats
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/train_bench.py > gpurun_out/train_bench.json 2> /dev/null
python tools/finetune_bench.py 2> /dev/null | tail -1 > gpurun_out/finetune_bench.json
TB_ITERS=3 TB_NATIVE_ONLY=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_train -- python3 tools/train_bench.py > gpurun_out/prof_train.log 2>&1
FB_ITERS=3 FB_NATIVE_ONLY=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ft -- python3 tools/finetune_bench.py > gpurun_out/prof_ft.log 2>&1
find gpurun_out/prof_train gpurun_out/prof_ft -name "*kernel_trace.csv" -delete
cat gpurun_out/train_bench.json gpurun_out/finetune_bench.json

# Round 6, final build: the whole GPU suite in one process, then the record run (tools/r6_profiles.sh)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/gpu_tests.sh > gpurun_out/r6_final_tests.txt 2>&1 || { tail -30 gpurun_out/r6_final_tests.txt; exit 1; }
tail -3 gpurun_out/r6_final_tests.txt
bash tools/r6_profiles.sh > gpurun_out/r6_final_record.txt 2>&1 || { tail -30 gpurun_out/r6_final_record.txt; exit 1; }
tail -60 gpurun_out/r6_final_record.txt

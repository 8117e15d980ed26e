# Round 6, final build: the reducer paths on the GPU -- the late chain join through a reducer (one process), then the two-rank equality
# rehearsal on one card over gloo (tools/dp_rehearsal.py: reduced buckets == full-batch gradients, ragged sharded sampling)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_boundary.py -x -q -m gpu -k "late_chain_join or switches or two_objective" > gpurun_out/r6_dp_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r6_dp_tests.log
[ $rc = 0 ] || { grep -E "^E |Error|assert" gpurun_out/r6_dp_tests.log | head -30; exit $rc; }
bash tools/dp_rehearsal.sh && cp gpurun_out/dp_rehearsal.json gpurun_out/r06_dp_rehearsal_2ranks_one_gpu.json

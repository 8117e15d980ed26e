# Round 4: the parity / edge / bench-path tests under the switches that select the older kernels of the small-launch path (every path stays green)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4s
T="tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_boundary.py tests/test_gpu_train.py"
run() { echo "== $1"; env $1 timeout -k 10 900 python -m pytest $T -m gpu -q -x 2>&1 | tail -2; }
run "MST_SMALL_FAST=0"
run "MST_SMALL_LN=0"
run "MST_SMALL_NTB1_M=0 MST_SMALL_NTB2_FROM=100000"
run "MST_SMALL_NTB1_M=100000 MST_SMALL_LN_M=100000"
# (MST_SMALL_M=100000 also changes the slice policy, which tests assert: 68 numerical tests pass under it up to the first policy assertion)

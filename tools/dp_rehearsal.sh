# Two ranks on one card (gloo): tools/dp_rehearsal.py; output -> gpurun_out/dp_rehearsal.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 tools/dp_rehearsal.py > gpurun_out/dp_rehearsal.log 2>&1; rc=$?
tail -5 gpurun_out/dp_rehearsal.log | cut -c1-1500
grep '^{' gpurun_out/dp_rehearsal.log | tail -1 > gpurun_out/dp_rehearsal.json
exit $rc

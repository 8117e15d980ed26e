# Round 4: the small-launch path after its GEMMs became resident-tile kernels (mst_small.h): clips/s over the batch size, 200-step loops.
#   default | LayerNorms as launches (MST_SMALL_LN=0) | slab ring (MST_SMALL_FAST=0) | large tiles only (MST_SMALL_M=0) | small path forced (MST_SMALL_M=100000)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { timeout -k 10 300 python bench.py --steps 2 --warmup 1 --denoise-steps 200 --batch $1 --no-cpu-baseline --no-boundary > gpurun_out/ab.log 2>&1; tail -1 gpurun_out/ab.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('batch $1 $2', d['value'], round(1e6*$1/d['value']/200,1), 'us/step')"; }
for b in ${BATCHES:-1 2 4 6 8 9 10 12}; do
  run $b "default               "
  MST_SMALL_LN=0 run $b "LayerNorm launches    "
  MST_SMALL_FAST=0 run $b "slab ring             "
  MST_SMALL_M=0 run $b "large tiles           "
  MST_SMALL_M=100000 MST_SMALL_LN_M=100000 run $b "small path forced     "
  MST_SMALL_M=100000 MST_SMALL_LN=0 run $b "small forced, LN launch"
done

# Round 4: 16-token tiles for the plain (no LayerNorm prologue) small-launch GEMMs, single-clip latency
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for r in 1 2; do
  echo "64-token tiles"; python tools/latency_b1.py 2>&1 | grep "^F="
  echo "16-token tiles up to 512 rows"; MST_SMALL_NTB1_M=512 python tools/latency_b1.py 2>&1 | grep "^F="
done
MST_SMALL_NTB1_M=512 python tools/single_clip_families.py 2>&1 | tail -9

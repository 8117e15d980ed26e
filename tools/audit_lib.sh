# ISA audit of the PRODUCT build: compiles csrc/mst_engine.hip with the library's own flag list (mst_amd._native.HIPCC_FLAGS -- the one
# compile recipe, hashed into the source hash) + -save-temps into gpurun_out/isa (scratch) and walks every kernel that streams weight
# fragments with hand-counted waits (tools/audit_stream_isa.py: no instruction touches a register of a load still in flight, no
# spill, no AGPR park).  CPU only (hipcc cross-compiles).   bash tools/audit_lib.sh
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $ROOT/gpurun_out/isa
FLAGS=$(cd $ROOT && python3 -c 'import mst_amd; from mst_amd import _native; print(" ".join(_native.HIPCC_FLAGS))')
cd $ROOT/diffusion-based-motion-style-transfer_amd/csrc
hipcc $FLAGS '-DMST_SRC_HASH="audit"' -save-temps=obj -o $ROOT/gpurun_out/isa/lib.so mst_engine.hip 2>&1 | tee $ROOT/gpurun_out/isa/build.log | grep -E "error|warning: v" || true
S=$ROOT/gpurun_out/isa/mst_engine-hip-amdgcn-amd-amdhsa-gfx950.s
test -s $S || { echo "hipcc produced no ISA: see gpurun_out/isa/build.log"; exit 1; }
rc=0
for k in k_layer_tailILi2E k_layer_tailILi3E k_layer_tailILi4E k_qkv_attention2ILi2E k_qkv_attention2ILi4E k_qkv_attention2ILi6E k_qkv_attention2ILi8E k_qkv_attention2ILi10E k_qkv_attention2ILi12E k_qkv_attention2ILi13E; do
  python3 $ROOT/tools/audit_stream_isa.py $S $k || rc=1
done
# every instantiation of the round-4 streaming kernels (mst_small.h, mst_embed.h)
for k in $(grep -oE "^_ZN3mst1[01]k_(rows_gemm|embed_out|embed_in)I[A-Za-z0-9]*E" $S | sort -u | sed -E 's/_ZN3mst1[01]//'); do
  python3 $ROOT/tools/audit_stream_isa.py $S $k || rc=1
done
exit $rc

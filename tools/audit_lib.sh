# ISA audit of the PRODUCT build: compiles csrc/mst_engine.hip with the library's own flag list (mst_amd._native.HIPCC_FLAGS -- the one
# compile recipe, hashed into the source hash) + -save-temps into gpurun_out/isa (scratch) and walks every kernel that streams weight
# fragments with hand-counted waits (tools/audit_stream_isa.py: no instruction touches a register of a load still in flight, no
# spill, no AGPR park), then EVERY kernel of the library for MFMA -> inline-asm register hazards (tools/audit_asm_hazards.py: hipcc pads
# nothing inside an asm statement).  CPU only (hipcc cross-compiles).   bash tools/audit_lib.sh [--selftest]
#   --selftest: also compile the layer tail WITHOUT tail_acc_settle() (tools/audit_selftest.hip, -DMST_AUDIT_SELFTEST_NO_SETTLE: the
#   build round 4 found 7 % wrong at 48 tokens) and require that the hazard audit REPORTS it -- the proof that the audit fires.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $ROOT/gpurun_out/isa
if [ "$1" = "--selftest" ]; then
  mkdir -p $ROOT/gpurun_out/isa/selftest
  for v in broken good; do
    D=""; [ $v = broken ] && D="-DMST_AUDIT_SELFTEST_NO_SETTLE"
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize $D -I$ROOT/diffusion-based-motion-style-transfer_amd/csrc -S --cuda-device-only \
      -o $ROOT/gpurun_out/isa/selftest/$v.s $ROOT/tools/audit_selftest.hip 2> /dev/null
  done
  echo "== selftest: the layer tail without tail_acc_settle() (must be reported)"
  if python3 $ROOT/tools/audit_asm_hazards.py $ROOT/gpurun_out/isa/selftest/broken.s; then echo "SELFTEST FAILED: the broken build passed the audit"; exit 1; fi
  echo "== selftest: the same translation unit as shipped (must be clean)"
  python3 $ROOT/tools/audit_asm_hazards.py $ROOT/gpurun_out/isa/selftest/good.s || { echo "SELFTEST FAILED: the shipped header is reported"; exit 1; }
  python3 $ROOT/tools/audit_asm_hazards.py --calibrate
fi
FLAGS=$(cd $ROOT && python3 -c 'import mst_amd; from mst_amd import _native; print(" ".join(_native.HIPCC_FLAGS))')
cd $ROOT/diffusion-based-motion-style-transfer_amd/csrc
hipcc $FLAGS '-DMST_SRC_HASH="audit"' -save-temps=obj -o $ROOT/gpurun_out/isa/lib.so mst_engine.hip 2>&1 | tee $ROOT/gpurun_out/isa/build.log | grep -E "error|warning: v" || true
S=$ROOT/gpurun_out/isa/mst_engine-hip-amdgcn-amd-amdhsa-gfx950.s
test -s $S || { echo "hipcc produced no ISA: see gpurun_out/isa/build.log"; exit 1; }
rc=0
for k in k_layer_tail_trainILi4E k_layer_tail_bwdILb0ELb0E k_layer_tail_bwdILb1ELb0E k_layer_tail_bwdILb0ELb1E; do      # training kernels: spills are reported, hazards fail
  timeout 300 python3 $ROOT/tools/audit_stream_isa.py $S $k --allow-spills || rc=1
done
for k in k_layer_tailILi2E k_layer_tailILi3E k_layer_tailILi4E k_qkv_attention2ILi2E k_qkv_attention2ILi4E k_qkv_attention2ILi6E k_qkv_attention2ILi8E k_qkv_attention2ILi10E k_qkv_attention2ILi12E k_qkv_attention2ILi13E; do
  python3 $ROOT/tools/audit_stream_isa.py $S $k || rc=1
done
# every instantiation of the round-4 streaming kernels (mst_small.h, mst_embed.h)
for k in $(grep -oE "^_ZN3mst1[01]k_(rows_gemm|embed_out|embed_in)I[A-Za-z0-9]*E" $S | sort -u | sed -E 's/_ZN3mst1[01]//'); do
  python3 $ROOT/tools/audit_stream_isa.py $S $k || rc=1
done
echo "== MFMA -> inline-asm register hazards, every kernel of the library"
python3 $ROOT/tools/audit_asm_hazards.py $S | tail -3 || rc=1
exit $rc

// tools/audit_lib.sh --selftest: the layer tail alone, every tile height, compiled WITHOUT tail_acc_settle() (mst_tail.h) -- the
// build round 4 found 7 % wrong at 48 tokens.  tools/audit_asm_hazards.py must report it.
#include "mst_tail.h"
template __global__ void mst::k_layer_tail<2>(const f16*, const f16*, const float*, const float*, const float*, const float*, const float*, const float*, const float*, f16*, f16*, const float*, int);
template __global__ void mst::k_layer_tail<3>(const f16*, const f16*, const float*, const float*, const float*, const float*, const float*, const float*, const float*, f16*, f16*, const float*, int);
template __global__ void mst::k_layer_tail<4>(const f16*, const f16*, const float*, const float*, const float*, const float*, const float*, const float*, const float*, f16*, f16*, const float*, int);

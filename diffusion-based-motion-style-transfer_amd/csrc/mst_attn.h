// K5: softmax(Q K^T / sqrt(128)) V for one (clip, head) per workgroup, S <= 224 tokens, no mask
// (the denoiser passes no key-padding mask, mdm_forstyledataset.py:622).
//
// The whole K and V of one head live in LDS (2 x S x 128 f16 <= 112 KB); each wave owns one 32-query
// tile and keeps the full score row block in registers, so softmax is exact two-pass (no online
// rescaling).  Both products keep the QUERY on the MFMA lane:
//     St[key][q] = K . Q^T        A = K fragment (ds_read_b128),  B = Q fragment (global -> regs)
//     Ot[d][q]   = V^T . P^T      A = V^T fragment (ds_read_b64_tr_b16), B = P straight from the
//                                 St accumulators (rows = keys sit in registers, cdna guide section 3
//                                 "accumulator tile as the next MFMA's operand")
// so row max / row sum are in-lane reductions plus one cross-half shuffle, 1/l is a per-lane
// scalar, and the output store is 8 bytes per lane (4 consecutive d of one query).
#pragma once
#include "mst_common.h"

namespace mst {

// K image: 256-B rows, 16-B chunk ch of row `row` at chunk ch ^ (row & 15)  (ds_read_b128, T2)
__device__ __forceinline__ int k_off(int row, int ch) { return row * 256 + ((ch ^ (row & 15)) << 4); }
// V image: 256-B rows cut into four 64-B pieces (one 32-wide d tile each); piece p of key `key`
// at piece p ^ (key & 3): the 4 keys x 64 B a half-wave's transposed read touches cover all 64 banks.
__device__ __forceinline__ int v_off(int key, int d) { return key * 256 + ((((d >> 5) ^ (key & 3))) << 6) + ((d & 31) << 1); }

template <int NKT>   // number of 32-key tiles = ceil(S / 32), 1..7
__global__ __launch_bounds__(512) void k_attention(const f16* __restrict__ qkv, f16* __restrict__ out, int S) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KEYS = NKT * 32;
    char* ks = smem;
    char* vs = smem + KEYS * 256;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int clip = blockIdx.x / MST_H, head = blockIdx.x % MST_H;
    const f16* base = qkv + (size_t)clip * S * (3 * MST_D) + head * MST_HD;

    // ---- stage K and V (zero rows beyond S so P = 0 meets V = 0, never NaN)
    for (int q = tid; q < KEYS * 16; q += 512) {
        int row = q >> 4, ch = q & 15;
        uint4 kv = {0, 0, 0, 0}, vv = {0, 0, 0, 0};
        if (row < S) {
            const f16* p = base + (size_t)row * (3 * MST_D) + ch * 8;
            kv = *reinterpret_cast<const uint4*>(p + MST_D);
            vv = *reinterpret_cast<const uint4*>(p + 2 * MST_D);
        }
        *reinterpret_cast<uint4*>(ks + k_off(row, ch)) = kv;
        *reinterpret_cast<uint4*>(vs + v_off(row, ch * 8)) = vv;
    }
    __syncthreads();
    if (wave >= NKT) return;        // query tiles = key tiles; no barrier below

    const int hh = lane >> 5;
    const int q_idx = wave * 32 + (lane & 31);
    const int q_ld = q_idx < S ? q_idx : S - 1;
    f16x8 qf[8];
    {
        const f16* qp = base + (size_t)q_ld * (3 * MST_D) + 8 * hh;
#pragma unroll
        for (int s = 0; s < 8; s++) qf[s] = *reinterpret_cast<const f16x8*>(qp + s * 16);
    }

    // ---- scores St[key][q]
    f32x16 sc[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; kt++) {
#pragma unroll
        for (int r = 0; r < 16; r++) sc[kt][r] = 0.f;
        const int row = kt * 32 + (lane & 31);
#pragma unroll
        for (int s = 0; s < 8; s++) {
            f16x8 kf = *reinterpret_cast<const f16x8*>(ks + k_off(row, 2 * s + hh));
            sc[kt] = mfma_f16(kf, qf[s], sc[kt]);
        }
    }

    // ---- softmax over keys (rows of St): registers, then the other half-wave
    const float scale = 0.08838834764831845f;   // 1/sqrt(128)
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; kt++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            float v = sc[kt][r] * scale;
            if (kt == NKT - 1) {
                int key = kt * 32 + mfma_row(r, lane);
                if (key >= S) v = -INFINITY;
            }
            sc[kt][r] = v;
            m = fmaxf(m, v);
        }
    m = fmaxf(m, __shfl_xor(m, 32));
    float l = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; kt++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            float p = __expf(sc[kt][r] - m);
            sc[kt][r] = p;
            l += p;
        }
    l += __shfl_xor(l, 32);
    const float inv_l = 1.0f / l;

    // P as f16 B-operand fragments: k-step s2 of key tile kt = registers 8*s2 .. 8*s2+7
    f16x8 pf[NKT][2];
#pragma unroll
    for (int kt = 0; kt < NKT; kt++)
#pragma unroll
        for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
            for (int j = 0; j < 8; j++) pf[kt][s2][j] = (f16)sc[kt][8 * s2 + j];

    // ---- Ot[d][q] = sum_key V[key][d] * P[q][key]
    // transposed read: lane L of 16-lane group g supplies row (L&15)>>2, columns 4*(L&3).. of a
    // 4-key x 16-d block and receives column (L&15) of the 4 keys.  Block for (kt, s2, half):
    // keys kt*32 + 16*s2 + 4*hh (+8 for fragment elements 4..7), d = dt*32 + 16*(g&1) ..
    const int i16 = lane & 15, g = lane >> 4;
    const int key_lane = 4 * hh + (i16 >> 2);
    const int d_lane = 16 * (g & 1) + 4 * (i16 & 3);
    f16* orow = out + ((size_t)clip * S + q_ld) * MST_D + head * MST_HD;
#pragma unroll
    for (int dt = 0; dt < 4; dt++) {
        f32x16 o;
#pragma unroll
        for (int r = 0; r < 16; r++) o[r] = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; kt++)
#pragma unroll
            for (int s2 = 0; s2 < 2; s2++) {
                int key = kt * 32 + 16 * s2 + key_lane;
                int d = dt * 32 + d_lane;
                s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (s16x4 __attribute__((address_space(3)))*)(vs + v_off(key, d)));
                s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (s16x4 __attribute__((address_space(3)))*)(vs + v_off(key + 8, d)));
                // whole-vector bit casts only: __builtin_bit_cast on an ext-vector ELEMENT (lo[j])
                // makes clang (ROCm 7.2) read element 0 for every j.
                const f16x4 lo_h = __builtin_bit_cast(f16x4, lo), hi_h = __builtin_bit_cast(f16x4, hi);
                f16x8 vf = __builtin_shufflevector(lo_h, hi_h, 0, 1, 2, 3, 4, 5, 6, 7);
                o = mfma_f16(vf, pf[kt][s2], o);
            }
        if (q_idx < S) {
#pragma unroll
            for (int gq = 0; gq < 4; gq++) {
                int dd = dt * 32 + 8 * gq + 4 * hh;
                *reinterpret_cast<uint2*>(orow + dd) =
                    pack4_f16(o[4 * gq] * inv_l, o[4 * gq + 1] * inv_l, o[4 * gq + 2] * inv_l, o[4 * gq + 3] * inv_l);
            }
        }
    }
}

}  // namespace mst

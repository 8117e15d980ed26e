// The encoder stack of one denoise step as ONE launch of resident workgroup groups (round 5).
//
// A sampling step's trunk was 16 dependent launches (8 x fused QKV + attention, 8 x fused layer tail), each a single round of workgroups:
// a launch lasts as long as one workgroup, and ~9 us per layer went into launch ramps and boundaries.  Here the four
// workgroups of a clip stay resident for the whole stack: member i runs head i of the attention phase (qa2_body, mst_attn.h), then token
// tile i of the layer tail (tail_body, mst_tail.h; a 197-token clip = 13 blocks of 16 tokens, split 4 | 3 | 3 | 3), layer after layer.  Clips
// never interact (model/mdm_forstyledataset.py:602-625 has no cross-sample op), so the ONLY synchronisation is among those four
// workgroups: an arrival counter per clip, two hand-offs per layer (group_wait / group_signal, mst_common.h).  The arithmetic is the two
// kernels' own code, instruction for instruction: results are bit-identical to the two-kernel path on 64- and 48-token tail tiles.
//
//   * blocks b, b + 8, b + 16, b + 24 of a 32-block chunk form a group (one XCD under round-robin placement: speed only);
//   * more clips than resident groups: a group walks clips g, g + G, ... (batch 128, classifier-free guidance: two rounds, no idle tail);
//   * a clip's counter counts its 8 arrivals per layer; the arrival that completes the launch's last phase puts it back to 0;
//   * every spin is bounded; a give-up sets the host-visible error word and lets the launch drain (mst_trunk_check()).
#pragma once
#include "mst_attn.h"
#include "mst_tail.h"

namespace mst {

struct TrunkLayer {
    const f16* wqkv; const float* b_in;
    const f16* wtail; const float *b_out, *g1, *be1, *b1, *b2, *g2, *be2;
};
struct TrunkArgs {
    const TrunkLayer* L;            // [nlayers] in device memory: a phase loads its ten pointers when it starts (kept in the kernel arguments,
                                    // 80 pointers stayed alive in scalar registers across every phase and spilled)
    f16 *hx, *hl, *att;
    const float* gelu_tab;
    unsigned* cnt;                  // [clips][32]: one 128-byte line per clip of this launch
    unsigned* err;
    int S, nclips, nlayers;
};

template <int NT16> struct TrunkTile {
    static constexpr int HI = (NT16 + 3) / 4, LO = NT16 / 4, NHI = NT16 - 4 * LO;      // members < NHI take HI blocks (NHI = 0: every member LO)
    static_assert(LO >= 2 && HI <= 4, "tail tiles of 32 .. 64 tokens");
    static constexpr int SMEM = QA2Tile<NT16>::SMEM > TailCfg::SMEM ? QA2Tile<NT16>::SMEM : TailCfg::SMEM;
    __host__ __device__ static constexpr int first_block(int m) { return m * LO + (m < NHI ? m : NHI); }
};

template <int NT16>
__global__ __launch_bounds__(512) void k_trunk_groups(TrunkArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using TT = TrunkTile<NT16>;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // the only use of threadIdx: v0 is free from here on
    const int G = gridDim.x >> 2, full = (G >> 3) << 5;
    int grp, member;
    if ((int)blockIdx.x < full) {
        const int within = blockIdx.x & 31;
        grp = ((blockIdx.x >> 5) << 3) + (within & 7);
        member = within >> 3;
    } else {
        const int r = blockIdx.x - full;
        grp = ((G >> 3) << 3) + (r >> 2);
        member = r & 3;
    }
    const int S = a.S;
#pragma unroll 1
    for (int clip = grp; clip < a.nclips; clip += G) {
        unsigned* cnt = a.cnt + clip * 32;
        const int row0 = clip * S, tok0 = row0 + 16 * TT::first_block(member);
#pragma unroll 1
        for (int l = 0; l < a.nlayers; l++) {
            // a phase loads ITS pointers when it starts (scalar loads through the constant address space: the table is never written while a
            // launch runs): nothing is hoisted or kept across phases
            typedef const TrunkLayer __attribute__((address_space(4))) * LayerTab;
            LayerTab wp = (LayerTab)(unsigned long long)opaque_uniform_ptr(a.L + l);
            {
                const f16* wqkv = wp->wqkv;
                const float* b_in = wp->b_in;
                const GroupSync sa{cnt, 8u * (unsigned)l, a.err};
                qa2_body<NT16, true>(smem, a.hx, wqkv, b_in, a.att, S, clip, member, sa, l > 0, wave);
            }
            group_signal(cnt, wave, 0u);
            wp = (LayerTab)(unsigned long long)opaque_uniform_ptr((const TrunkLayer*)(unsigned long long)wp);
            {
                const TrunkLayer w{nullptr, nullptr, wp->wtail, wp->b_out, wp->g1, wp->be1, wp->b1, wp->b2, wp->g2, wp->be2};
                const GroupSync st{cnt, 8u * (unsigned)l + 4u, a.err};
                if (TT::NHI > 0 && member < TT::NHI)
                    tail_body<TT::HI, true>(smem, a.att, w.wtail, w.b_out, w.g1, w.be1, w.b1, w.b2, w.g2, w.be2, a.hx, a.hl, a.gelu_tab, row0 + S, tok0, st, wave);
                else
                    tail_body<TT::LO, true>(smem, a.att, w.wtail, w.b_out, w.g1, w.be1, w.b1, w.b2, w.g2, w.be2, a.hx, a.hl, a.gelu_tab, row0 + S, tok0, st, wave);
            }
            group_signal(cnt, wave, l + 1 == a.nlayers ? 8u * (unsigned)a.nlayers : 0u);
        }
    }
}

}  // namespace mst

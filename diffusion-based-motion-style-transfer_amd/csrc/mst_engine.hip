// libmst_engine.so -- host side: engine state, launch sequence, C ABI (include/mst_engine.h).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include "../../include/mst_engine.h"
#include "mst_attn.h"
#include "mst_train.h"
#include "mst_common.h"
#include "mst_elem.h"
#include "mst_gemm_dma.h"
#include "mst_tail.h"
#include "mst_trunk.h"
#include "mst_tail_bwd.h"
#ifdef EMB_PROBE            // diagnostic build only (tools/experiments/r4_embed_stamps.py): wave 0..7 of every workgroup stamp the 100 MHz clock at the phase marks
__device__ unsigned long long g_emb_stamp[512][8][8];
#define EMB_MARK(i) if ((threadIdx.x & 63) == 0 && blockIdx.x < 512) g_emb_stamp[blockIdx.x][threadIdx.x >> 6][i] = __builtin_amdgcn_s_memrealtime();
extern "C" int mst_probe_read(void* dst) { return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_emb_stamp), sizeof(g_emb_stamp)) == hipSuccess ? 0 : 1; }
#endif
#include "mst_embed.h"
#include "mst_small.h"

using namespace mst;

// wide (QKV / FFN1) tile: WIDE_BT tokens x 256 features, 8 waves of (WIDE_BT/64) x 2 MFMA tiles
#define WIDE_BT 128   // same-box A/B (round 1, script since removed): 128/3-slot/XCD 52.4 clips/s, 128/4-slot 47.2, 256/4-slot 47.5
#define WIDE_NS 3     // 3 x 24 KB ring: two blocks share a CU
#define FFN1_BF 256   // same-box A/B: 128x512 tiles (one 8-wave block per CU) 40 us vs 35 us for two co-resident 128x256 blocks
#define WIDE_XCD 1
// LayerNorm GEMMs (64 x 512 tiles, one block per CU): slab depth / ring slots.  64-deep slabs move whole
// 128-B cache lines per DMA segment; 32-deep slabs fetch every line twice, one slab apart (gemm_bench: 24.3 -> 19.5 us at K=512).
#define LN_BK 64
#define LN_NS (LN_BK == 64 ? 2 : 4)
// (These are fixed constants, not -D knobs: a stray compile flag must not be able to change what the product library computes or how
// fast -- _native.build_in_place passes no -D but the source hash, and folds its flag list into that hash.)
// launch geometry of a wide GEMM over nx token tiles x ny feature tiles (1-D when XCD-aware)
static dim3 wide_grid(int M, int ny) {
    const int nx = (M + WIDE_BT - 1) / WIDE_BT;
    return WIDE_XCD ? dim3(((nx + 7) / 8) * 8 * ny, 1, 1) : dim3(nx, ny, 1);
}

// dgrad GEMMs that end in the fp32 gradient stream (N = 512): token tile
#define DG_BT 128     // same-box A/B of the backward pass at batch 64: 128 x 256 tiles 2.99 ms, 64 x 256 (394 blocks, two per CU) 3.10 ms
static dim3 dg_grid(int M, int ny) {
    const int nx = (M + DG_BT - 1) / DG_BT;
    return WIDE_XCD ? dim3(((nx + 7) / 8) * 8 * ny, 1, 1) : dim3(nx, ny, 1);
}

// ------------------------------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";
static int fail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return 1;
}
#define HIPCHECK(x)                                                                              \
    do {                                                                                         \
        hipError_t _e = (x);                                                                     \
        if (_e != hipSuccess) return fail("%s failed: %s (%s:%d)", #x, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)
#define CHECK(x)                 \
    do {                         \
        int _r = (x);            \
        if (_r) return _r;       \
    } while (0)

extern "C" const char* mst_last_error(void) { return g_err; }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a PER-DEVICE opt-in: remember (kernel, device) pairs, so a second
// engine on another GPU of the same process gets its own (a process-wide "done" flag would launch there without it).
static int ensure_dyn_lds(const void* kern, int bytes) {
    static std::mutex mu;
    static std::set<std::pair<const void*, int>> done;
    int dev = 0;
    HIPCHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    if (done.count({kern, dev})) return 0;
    HIPCHECK(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    done.insert({kern, dev});
    return 0;
}

// Every ABI entry runs on its engine's / schedule's device and leaves the caller's current device as it found it.
struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess) cur = -1;
        if (cur != dev) {
            ok = hipSetDevice(dev) == hipSuccess;
            prev = cur;
        }
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};
#define ON_DEVICE(dev)                                                    \
    DeviceGuard _dev_guard(dev);                                          \
    if (!_dev_guard.ok) return fail("hipSetDevice(%d) failed (%s:%d)", (int)(dev), __FILE__, __LINE__)
extern "C" int mst_version(void) { return 2; }
#ifndef MST_SRC_HASH
#define MST_SRC_HASH "unhashed"
#endif
extern "C" const char* mst_source_hash(void) { return MST_SRC_HASH; }

// ------------------------------------------------------------------------------------------ state
struct mst_schedule {
    int n = 0, device = 0;
    float* tab = nullptr;          // [NTAB][n] float32
    long long* tmap = nullptr;     // [n] int64 original-process timesteps
};

struct LayerW {
    f16 *w_in = nullptr, *w_out = nullptr, *w1 = nullptr, *w2 = nullptr;
    float *b_in = nullptr, *b_out = nullptr, *b1 = nullptr, *b2 = nullptr;
    float *g1 = nullptr, *be1 = nullptr, *g2 = nullptr, *be2 = nullptr;
    // [in][out] f16 copies: the "weights" operand of the dgrad GEMMs (training path)
    f16 *w_inT = nullptr, *w_outT = nullptr, *w1T = nullptr, *w2T = nullptr;
    f16* wtail = nullptr;           // W_out | W1 | W2 as the fused layer tail's slab stream (mst_tail.h, k_pack_tail)
    f16* wtail_bwd = nullptr;       // W2^T | W1^T | W_out^T as the fused BACKWARD tail's stream (mst_tail_bwd.h, k_pack_tail_bwd): allocated and packed on first use
    bool tailb_dirty = true;
    f16 *w_in_lo = nullptr, *w_out_lo = nullptr, *w1_lo = nullptr, *w2_lo = nullptr;   // f16(w - f16(w)): the weights' lo halves (precise mode)
    bool tail_dirty = true, qkv_dirty = true;   // wtail / wqkv are older than w_out | w1 | w2 / w_in: repacked by ensure_packed() before the next sampling launch
    f16* wqkv = nullptr;            // W_in as the fused QKV+attention kernel's per-(head, wave) fragment streams (mst_attn.h, k_pack_qkv)
    f16 *wsm_in = nullptr, *wsm_out = nullptr, *wsm_1 = nullptr, *wsm_2 = nullptr;   // the four matrices as [16-row block][k-step] fragments: the small-launch GEMMs (mst_small.h)
    bool small_dirty = true;        // ... older than the plain matrices (ensure_packed / the training forward repack them, all stale layers in one launch)
};

// engine-owned scratch of the backward pass, allocated on the first training call
struct TrainWS {
    bool ready = false;
    float *g0 = nullptr, *g1 = nullptr;                 // fp32 gradient stream, ping-pong
    // operands the wgrad stream reads while the dgrad chain moves on: double-buffered by layer parity
    f16 *dbr2[2] = {nullptr, nullptr}, *dbr1[2] = {nullptr, nullptr}, *dpre[2] = {nullptr, nullptr}, *dqkv[2] = {nullptr, nullptr};
    f16* datt = nullptr;
    f16* hidr[2] = {nullptr, nullptr};                  // hid = dropout(GELU(pre)) regenerated by the FFN2 dgrad epilogue: the operand of dW2 (by layer parity)
    hipEvent_t ev_ready = nullptr, ev_side[2] = {nullptr, nullptr};
    hipEvent_t ev_layer[16] = {nullptr};                // recorded when layer l's parameter gradients of the LAST backward call are enqueued
    bool layer_done[16] = {false};
    float* part = nullptr;                              // split-K partial products
    float* ln_part = nullptr;                           // k_ln_bwd's per-block [dgamma | dbeta | dbias] partials (dgrad stream)
    float* ln_part2 = nullptr;                          // ... LayerNorm2's, when both LayerNorms of a layer are finished by one launch (round 6)
    // round 6: a layer's split-K reduces wait for ONE batched launch at the layer's end (train_stack_backward sets red_defer around its layers;
    // each weight gradient then gets a region of `part` of its own).  MST_WGRAD_REDUCE_BATCH=0: a reduce behind every weight gradient.
    RedJobs red_jobs{};
    int n_red = 0;
    bool red_defer = false;
    size_t part_used = 0, part_cap = 0;
    float* cs_part = nullptr;                           // k_colsum_f16's per-row-block bias-gradient partials (wgrad stream)
    float* zeros = nullptr;                             // zero bias
    float* gscale = nullptr;                            // [0] scale applied to the incoming gradient, [1] its inverse
    unsigned* amax = nullptr;
    size_t split_cap = 0;
};

enum Family { FAM_COND = 0, FAM_EMBED_IN, FAM_QKV, FAM_ATTN, FAM_OUTPROJ_LN, FAM_FFN1, FAM_FFN2_LN, FAM_EMBED_OUT, FAM_QKV_ATTN, FAM_TAIL, FAM_WGRAD, FAM_COUNT };
static const char* kFamilyNames[FAM_COUNT] = {"cond_token", "embed_in", "qkv_gemm", "attention", "outproj_ln_gemm",
                                              "ffn1_gelu_gemm", "ffn2_ln_gemm", "embed_out_step", "qkv_attention_fused",
                                              "layer_tail_fused", "wgrad_tr"};

struct ProfPoint { int fam; hipEvent_t a, b; };

struct mst_engine {
    mst_config cfg;
    int S_max = 0, M_pad = 0, kin_pad = 0, nt_out = 0, fout_pad = 0;
    LayerW L[16];
    f16 *w_pose_in = nullptr, *w_pose_out = nullptr;
    f16 *w_pose_in_lo = nullptr, *w_pose_out_lo = nullptr;      // lo halves (precise mode)
    f16 *w_pose_inT = nullptr, *w_pose_outT = nullptr;   // [fout_pad][512] and [512][kin_pad]: the projections' dgrad operands
    float *b_pose_in = nullptr, *b_pose_out = nullptr;
    f16 *w_pose_in_pk = nullptr, *w_pose_out_pk = nullptr;      // the two projections as per-wave fragment streams (mst_embed.h, k_pack_wave_blocks)
    bool pose_in_dirty = true, pose_out_dirty = true;          // ... older than w_pose_in / w_pose_out: repacked by ensure_packed()
    int small_ln = 1, small_ln_m = 512;   // ... with the LayerNorms inside the consuming GEMMs, up to this many stream rows (MST_SMALL_LN, MST_SMALL_LN_M; tools/experiments/r4_small_sweep.sh: ahead through 2 clips x 197 rows, behind from 4)
    int train_fuse_ln2_bwd = 1;           // ... with LayerNorm2's backward at the head of the same launch (frozen stacks; MST_TRAIN_FUSE_LN2_BWD=0: a launch of its own; LAB_NOTES R6.9)
    int train_fuse_bwd_tail = 1;          // training backward at batch size: FFN2 dgrad + GELU' + FFN1 dgrad + LayerNorm1 backward + out-proj dgrad as k_layer_tail_bwd
                                          // (MST_TRAIN_FUSE_BWD_TAIL: 0 = three dgrad launches, 1 = frozen stacks (no parameter gradients: the motion encoder), 2 = every stack)
    int train_small_ln = 1;               // training at a clip or two: the LayerNorms inside the GEMMs behind them (MST_TRAIN_SMALL_LN=0: k_ln_rows_train launches)
    int train_fuse_tail = 1;              // training forward at batch size: out-proj + LN1 + FFN + LN2 as k_layer_tail_train, writing the tape (MST_TRAIN_FUSE_TAIL=0: three ring GEMMs)
    int fuse_ln_bwd = 1;                  // training at batch size: LayerNorm1's backward in the epilogue of the dgrad GEMM in front of it (MST_FUSE_LN_BWD=0: two launches)
    int small_fast = 1;                   // small launches: the layer GEMMs as the kernels of mst_small.h (MST_SMALL_FAST=0: the slab ring)
    int fuse_embed = 1;                   // a sampling step's output projection also embeds the next step (MST_FUSE_EMBED=0: two launches)
    int embed_fast = 1;                   // K3 / K9 as the latency kernels of mst_embed.h; MST_EMBED_FAST=0: the ring GEMMs of rounds 1-3
    float *w_t0 = nullptr, *b_t0 = nullptr, *w_t2 = nullptr, *b_t2 = nullptr, *w_text = nullptr, *b_text = nullptr;
    float* pe = nullptr;
    // workspace
    f16* hl = nullptr;        // lo half of the stream (hx is the hi half)
    f16 *hx2 = nullptr, *hl2 = nullptr;   // second stream buffer: small launches with the LayerNorms inside the GEMMs (mst_small.h, LnRows)
    f16 *hx = nullptr, *qkv = nullptr, *att = nullptr, *hid = nullptr, *xt = nullptr;
    float* gelu_tab = nullptr;      // Phi(x) interpolation table of the fused layer tail's GELU stage (TailCfg::GELU_N entries {Phi, dPhi})
    f16* xt_lo = nullptr;     // lo half of the frame rows: the pose embedding multiplies x_t as hi + lo (RowsDirect::Xlo)
    float *temb_hid = nullptr, *temb = nullptr, *textproj = nullptr;
    int temb_cap = 0;
    // round 6: the timestep embedding of EVERY timestep (pe_len rows), built once per load of the (frozen) timestep MLP: a training call reads its
    // clips' rows through their timestep indices inside the embedding kernel instead of running the two-layer MLP on them first -- two dependent
    // launches less at the head of every model call, six of the seven per fine-tune iteration on the chained steps' critical path.  MST_TEMB_TABLE=0: off.
    float *temb_table = nullptr, *temb_table_hid = nullptr;
    bool temb_table_valid = false;
    int temb_table_on = 1;
    std::vector<std::string> loaded;
    std::set<std::string> lo_missing;     // GEMM weights uploaded while precise mode was off: their lo halves f16(w - f16(w)) were not written
                                          // (a fine-tune iteration re-uploads all 96 tensors and never reads them); precise mode refuses to run on those
    int text_batch = 0, text_cfg = 0;
    TrainWS tw;
    int dbg_layer = -1, dbg_stage = -1;   // stop the trunk after (layer, stage); -1 = run everything
    int fuse_qkv_attn = 1;                // K4 + K5 as one kernel per (clip, head); MST_FUSE_QKV_ATTN=0 keeps them apart
    int fuse_tail = 1;                    // K6 + K7 + K8 as one kernel per 64-token tile (mst_tail.h); MST_FUSE_TAIL=0 keeps them apart
    int tail_ntb = 0;                     // fused layer tail: 16-token blocks per tile; 0 = per launch (launch_tail), MST_TAIL_NTB=2..4 fixes it
    int cur_slices = 1;                   // clip slices the launches being enqueued share the chip with (mst_sample_loop; 1 = a lone launch sequence)
    int trunk_groups = 0;                 // MST_TRUNK=1: the encoder stack of a sampling step as ONE launch of resident workgroup groups (mst_trunk.h)
    unsigned* trunk_cnt = nullptr;        // [max_rows][32]: a clip's arrival counter (one 128-byte line each); every launch finds it at 0 and leaves it at 0
    unsigned* trunk_err = nullptr;        // pinned host word the kernel sets when a bounded spin gives up (mst_trunk_check)
    bool trunk_used = false;              // a resident-group launch was enqueued since the last check: mst_forward / mst_sample_loop synchronise and read trunk_err
    void* trunk_layers = nullptr;         // TrunkLayer[num_layers] in device memory: the layers' pointers (fixed at creation), uploaded at the first such launch
    int fuse_frames = 1;                  // sampling loop: a step's epilogue writes the next step's f16 frame rows; MST_FUSE_FRAMES=0 runs k_frames_f16 every step
    int precise = 0;                      // mst_set_precise / MST_PRECISE=1: every layer GEMM of the sampling path multiplies its activation as hi + lo (the small-tile
                                          // kernels at any size, ~2x their MFMA work): for checkpoints whose outlier channels put f16 operands above the 1e-3 bar
    int small_m = 1900;                   // launches of at most this many token rows take the small-tile path (MST_SMALL_M, 0 = never).
                                          // Round 4, with 32-token tiles in the fused tail of a lone launch (tools/experiments/r4_batch_sweep.sh, clips/s of a
                                          // 200-step loop, small / large tiles): 4 clips 46.2 / 41.0, 8: 85.6 / 83.2, 12: 105.8 / 125.1, 16: 139.2 / 167.5
                                          // -- hand-over between 9 and 10 clips (it was 16 clips = 3200 rows with 64-token tiles only)
    float* zacc = nullptr;                // fp32 GEMM result feeding k_ln_rows on that path
    int ln128_min_m = 1 << 30;            // MST_LN128_M=n: launches of >= n token rows use 128-token LayerNorm tiles (half the weight
                                          // re-streaming).  Off by default: wins 16-21 % in gemm_bench, nothing in the pipeline (CFG 39.7 vs
                                          // 39.9, batch 128 77.2 vs 77.1 clips/s; forced at batch 64: 58.7 vs 68.3) -- kept, parity-tested
    int wgrad_stream_on = 1;              // training: wgrads on a second stream beside the dgrad chain (MST_WGRAD_STREAM=0: one stream)
    int nsplit = 0;                       // clip slices of a sampling loop on separate streams: 0 = chosen per call (loop_slices_for), 1..3 = MST_STREAMS
    static constexpr int MAX_SLICES = 8;
    hipStream_t aux_stream[MAX_SLICES - 1] = {nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[MAX_SLICES - 1] = {nullptr};
    // sampling-loop state in device memory + the captured step graph (mst_sample_loop)
    LoopDev* ld_dev = nullptr;
    unsigned char* rowflag = nullptr;     // [max_rows][feats] summary of the loop's inpainting mask (k_mask_rowflags)
    LoopDev* ld_pin = nullptr;            // pinned staging ring for the per-call upload
    static constexpr int LD_SLOTS = 8;
    hipEvent_t ld_ev[LD_SLOTS] = {nullptr};
    int ld_next = 0;
    int graph_on = 0;                     // MST_GRAPH=1: long loops replay a captured hipGraph of graph_steps steps (host enqueue ~0).
                                          // Off by default: on ROCm 7.2 the replay is not faster than host-enqueued launches
                                          // (same box, batch 64: 73.7 vs 74.1 clips/s; batch 1: 381 vs 367 us per step) -- the loop
                                          // is GPU-bound either way (tools/host_bound.py)
    int graph_steps = 20;                 // denoise steps per captured graph (MST_GRAPH_STEPS)
    hipStream_t loop_stream = nullptr;    // the loop runs on the engine's own streams (a caller's legacy default stream cannot be
    hipEvent_t ev_in = nullptr, ev_out = nullptr;   //  captured); it is ordered behind / before the caller's stream by these two events
    hipGraphExec_t gexec = nullptr;
    std::vector<long long> gkey;          // what the instantiated graph was captured for
    std::vector<long long> warm_key;      // configuration whose kernels have all been launched eagerly once (per-kernel LDS opt-ins done)
    // profiling
    int prof_on = 0, prof_now = 0, prof_period = 16;
    std::vector<ProfPoint> prof_pts;
    size_t prof_used = 0;
    float prof_overhead_us = -1.f;        // median duration an empty event pair reports (calibrated at mst_profile_enable)
    double prof_ms[FAM_COUNT] = {0};
    int prof_n[FAM_COUNT] = {0};
};

template <class T>
static int dmalloc(T** p, size_t count) {
    HIPCHECK(hipMalloc((void**)p, count * sizeof(T)));
    HIPCHECK(hipMemset(*p, 0, count * sizeof(T)));
    return 0;
}

// ------------------------------------------------------------------------------------------ schedule
extern "C" int mst_schedule_create(int32_t num_steps, const float* tables_host, const int32_t* timestep_map_host,
                                   int32_t device, mst_schedule** out) {
    if (!out || !tables_host || !timestep_map_host || num_steps <= 0) return fail("mst_schedule_create: bad arguments");
    ON_DEVICE(device);
    mst_schedule* s = new mst_schedule();
    s->n = num_steps;
    s->device = device;
    HIPCHECK(hipMalloc((void**)&s->tab, sizeof(float) * NTAB * num_steps));
    HIPCHECK(hipMalloc((void**)&s->tmap, sizeof(long long) * num_steps));
    HIPCHECK(hipMemcpy(s->tab, tables_host, sizeof(float) * NTAB * num_steps, hipMemcpyHostToDevice));
    std::vector<long long> tm(num_steps);
    for (int i = 0; i < num_steps; i++) tm[i] = timestep_map_host[i];
    HIPCHECK(hipMemcpy(s->tmap, tm.data(), sizeof(long long) * num_steps, hipMemcpyHostToDevice));
    *out = s;
    return 0;
}

extern "C" void mst_schedule_destroy(mst_schedule* s) {
    if (!s) return;
    (void)hipFree(s->tab);
    (void)hipFree(s->tmap);
    delete s;
}

// ------------------------------------------------------------------------------------------ engine
extern "C" int mst_engine_create(const mst_config* c, mst_engine** out) {
    if (!c || !out) return fail("mst_engine_create: null argument");
    if (c->latent_dim != MST_D || c->num_heads != MST_H || c->ff_size != MST_FF)
        return fail("mst_engine_create: kernels are built for latent_dim 512 / 4 heads / ff 1024 (got %d/%d/%d)",
                    c->latent_dim, c->num_heads, c->ff_size);
    if (c->num_layers < 1 || c->num_layers > 16) return fail("mst_engine_create: num_layers must be 1..16");
    if (c->feats < 1 || c->feats > 512) return fail("mst_engine_create: feats must be 1..512 (got %d)", c->feats);
    if (c->max_frames < 1 || c->max_frames > 223) return fail("mst_engine_create: max_frames must be 1..223 (got %d)", c->max_frames);
    if (c->max_rows < 1) return fail("mst_engine_create: max_rows must be >= 1");
    if (c->clip_dim < 1 || c->pe_len < c->max_frames + 1) return fail("mst_engine_create: bad clip_dim / pe_len");
    ON_DEVICE(c->device);
    mst_engine* e = new mst_engine();
    e->cfg = *c;
    e->S_max = c->max_frames + 1;
    size_t M = (size_t)c->max_rows * e->S_max;
    // rows of every token-row buffer: whole 256-row tiles plus one spare tile -- a slice starts at an arbitrary row and
    // its last tile may over-READ up to a tile beyond the slice (never write)
    e->M_pad = (int)(((M + 255) / 256) * 256) + 256;
    e->kin_pad = ((c->feats + 31) / 32) * 32;           // K of the pose-embedding GEMM: whole 32-deep slabs,
    if (e->kin_pad < 96) e->kin_pad = 96;               // and at least the 3 slabs the ring keeps in flight
    e->nt_out = (c->feats + 255) / 256;              // output-projection tile = 256 * nt_out features
    e->fout_pad = e->nt_out * 256;
    for (int l = 0; l < c->num_layers; l++) {
        LayerW& w = e->L[l];
        CHECK(dmalloc(&w.w_in, (size_t)3 * MST_D * MST_D));
        CHECK(dmalloc(&w.w_out, (size_t)MST_D * MST_D));
        CHECK(dmalloc(&w.w1, (size_t)MST_FF * MST_D));
        CHECK(dmalloc(&w.w2, (size_t)MST_D * MST_FF));
        CHECK(dmalloc(&w.b_in, 3 * MST_D));
        CHECK(dmalloc(&w.b_out, MST_D));
        CHECK(dmalloc(&w.b1, MST_FF));
        CHECK(dmalloc(&w.b2, MST_D));
        CHECK(dmalloc(&w.g1, MST_D));
        CHECK(dmalloc(&w.be1, MST_D));
        CHECK(dmalloc(&w.g2, MST_D));
        CHECK(dmalloc(&w.be2, MST_D));
        CHECK(dmalloc(&w.w_inT, (size_t)3 * MST_D * MST_D));
        CHECK(dmalloc(&w.w_outT, (size_t)MST_D * MST_D));
        CHECK(dmalloc(&w.w1T, (size_t)MST_FF * MST_D));
        CHECK(dmalloc(&w.w2T, (size_t)MST_D * MST_FF));
        CHECK(dmalloc(&w.wtail, TailCfg::LAYER_BYTES / 2));
        CHECK(dmalloc(&w.wqkv, (size_t)3 * MST_D * MST_D));
        CHECK(dmalloc(&w.wsm_in, (size_t)3 * MST_D * MST_D));
        CHECK(dmalloc(&w.wsm_out, (size_t)MST_D * MST_D));
        CHECK(dmalloc(&w.wsm_1, (size_t)MST_FF * MST_D));
        CHECK(dmalloc(&w.wsm_2, (size_t)MST_D * MST_FF));
        CHECK(dmalloc(&w.w_in_lo, (size_t)3 * MST_D * MST_D));
        CHECK(dmalloc(&w.w_out_lo, (size_t)MST_D * MST_D));
        CHECK(dmalloc(&w.w1_lo, (size_t)MST_FF * MST_D));
        CHECK(dmalloc(&w.w2_lo, (size_t)MST_D * MST_FF));
    }
    CHECK(dmalloc(&e->w_pose_in, (size_t)MST_D * e->kin_pad));
    CHECK(dmalloc(&e->w_pose_in_lo, (size_t)MST_D * e->kin_pad));
    CHECK(dmalloc(&e->w_pose_out_lo, (size_t)e->fout_pad * MST_D));
    CHECK(dmalloc(&e->b_pose_in, MST_D));
    CHECK(dmalloc(&e->w_pose_in_pk, (size_t)8 * 16 * 4 * 512));
    CHECK(dmalloc(&e->w_pose_out_pk, (size_t)8 * 16 * 4 * 512));
    CHECK(dmalloc(&e->w_pose_out, (size_t)e->fout_pad * MST_D));
    CHECK(dmalloc(&e->b_pose_out, e->fout_pad));
    CHECK(dmalloc(&e->w_pose_inT, (size_t)e->fout_pad * MST_D));
    CHECK(dmalloc(&e->w_pose_outT, (size_t)MST_D * e->kin_pad));
    CHECK(dmalloc(&e->w_t0, (size_t)MST_D * MST_D));
    CHECK(dmalloc(&e->b_t0, MST_D));
    CHECK(dmalloc(&e->w_t2, (size_t)MST_D * MST_D));
    CHECK(dmalloc(&e->b_t2, MST_D));
    CHECK(dmalloc(&e->w_text, (size_t)MST_D * c->clip_dim));
    CHECK(dmalloc(&e->b_text, MST_D));
    CHECK(dmalloc(&e->pe, (size_t)c->pe_len * MST_D));
    CHECK(dmalloc(&e->hl, (size_t)e->M_pad * MST_D));
    CHECK(dmalloc(&e->hx, (size_t)e->M_pad * MST_D));
    CHECK(dmalloc(&e->hx2, (size_t)e->M_pad * MST_D));
    CHECK(dmalloc(&e->hl2, (size_t)e->M_pad * MST_D));
    CHECK(dmalloc(&e->qkv, (size_t)e->M_pad * 3 * MST_D));
    CHECK(dmalloc(&e->att, (size_t)e->M_pad * MST_D));
    CHECK(dmalloc(&e->hid, (size_t)e->M_pad * MST_FF));
    CHECK(dmalloc(&e->xt, ((size_t)c->max_rows * c->max_frames + 128) * e->kin_pad));
    CHECK(dmalloc(&e->xt_lo, ((size_t)c->max_rows * c->max_frames + 128) * e->kin_pad));
    {   // Phi in double on the host: entry i = {Phi(x_i), Phi(x_i+1) - Phi(x_i)}, x_i = -6 + i / 128
        std::vector<float> tab(TailCfg::GELU_TAB_BYTES / 4, 0.f);
        auto phi = [](double x) { return 0.5 * std::erfc(-x * 0.70710678118654752440); };
        for (int i = 0; i <= TailCfg::GELU_N; i++) {
            const double x0 = -6.0 + i / 128.0, p0 = phi(x0), p1 = phi(x0 + 1.0 / 128.0);
            tab[2 * i] = (float)p0;
            tab[2 * i + 1] = (float)(p1 - p0);
        }
        CHECK(dmalloc(&e->gelu_tab, tab.size()));
        HIPCHECK(hipMemcpy(e->gelu_tab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
    }
    e->temb_cap = c->max_rows > 1024 ? c->max_rows : 1024;
    CHECK(dmalloc(&e->temb_hid, (size_t)e->temb_cap * MST_D));
    CHECK(dmalloc(&e->temb, (size_t)e->temb_cap * MST_D));
    CHECK(dmalloc(&e->textproj, (size_t)c->max_rows * MST_D));
    HIPCHECK(hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
    for (int i = 0; i < mst_engine::MAX_SLICES - 1; i++) {
        HIPCHECK(hipStreamCreateWithFlags(&e->aux_stream[i], hipStreamNonBlocking));
        HIPCHECK(hipEventCreateWithFlags(&e->ev_join[i], hipEventDisableTiming));
    }
    if (const char* v = getenv("MST_STREAMS")) {
        // At most three concurrent slices: ROCm maps streams onto four hardware queues by default and the caller's stream holds
        // one, so a fourth slice stream shares a queue with another and the two serialise (measured: 77.1 clips/s at 3, 40.9 at 4).
        int n = atoi(v);
        e->nsplit = n < 1 ? 1 : (n > mst_engine::MAX_SLICES ? mst_engine::MAX_SLICES : n);
    }
    if (const char* v = getenv("MST_FUSE_QKV_ATTN")) e->fuse_qkv_attn = atoi(v);
    if (const char* v = getenv("MST_FUSE_TAIL")) e->fuse_tail = atoi(v) != 0;
    if (const char* v = getenv("MST_FUSE_FRAMES")) e->fuse_frames = atoi(v) != 0;
    if (const char* v = getenv("MST_EMBED_FAST")) e->embed_fast = atoi(v) != 0;
    if (const char* v = getenv("MST_FUSE_EMBED")) e->fuse_embed = atoi(v) != 0;
    if (const char* v = getenv("MST_SMALL_FAST")) e->small_fast = atoi(v) != 0;
    if (const char* v = getenv("MST_FUSE_LN_BWD")) e->fuse_ln_bwd = atoi(v) != 0;
    if (const char* v = getenv("MST_TRAIN_FUSE_TAIL")) e->train_fuse_tail = atoi(v) != 0;
    if (const char* v = getenv("MST_TRAIN_SMALL_LN")) e->train_small_ln = atoi(v) != 0;
    if (const char* v = getenv("MST_TRAIN_FUSE_BWD_TAIL")) e->train_fuse_bwd_tail = atoi(v);
    if (const char* v = getenv("MST_TRAIN_FUSE_LN2_BWD")) e->train_fuse_ln2_bwd = atoi(v);
    if (const char* v = getenv("MST_TEMB_TABLE")) e->temb_table_on = atoi(v);
    if (const char* v = getenv("MST_SMALL_LN")) e->small_ln = atoi(v) != 0;
    if (const char* v = getenv("MST_SMALL_LN_M")) e->small_ln_m = atoi(v);
    if (const char* v = getenv("MST_TAIL_NTB")) { int n = atoi(v); e->tail_ntb = (n >= 2 && n <= 4) ? n : 0; }
    if (const char* v = getenv("MST_WGRAD_STREAM")) e->wgrad_stream_on = atoi(v) != 0;
    if (const char* v = getenv("MST_SMALL_M")) e->small_m = atoi(v);
    if (const char* v = getenv("MST_PRECISE")) e->precise = atoi(v) != 0;
    if (const char* v = getenv("MST_LN128_M")) e->ln128_min_m = atoi(v);
    if (const char* v = getenv("MST_TRUNK")) e->trunk_groups = atoi(v) != 0;
    CHECK(dmalloc(&e->trunk_cnt, (size_t)c->max_rows * 32));
    HIPCHECK(hipHostMalloc((void**)&e->trunk_err, 64, hipHostMallocDefault));
    e->trunk_err[0] = 0;
    CHECK(dmalloc(&e->zacc, (size_t)e->M_pad * MST_D));
    CHECK(dmalloc(&e->ld_dev, 1));
    CHECK(dmalloc(&e->rowflag, (size_t)c->max_rows * c->feats));
    HIPCHECK(hipStreamCreateWithFlags(&e->loop_stream, hipStreamNonBlocking));
    HIPCHECK(hipEventCreateWithFlags(&e->ev_in, hipEventDisableTiming));
    HIPCHECK(hipEventCreateWithFlags(&e->ev_out, hipEventDisableTiming));
    HIPCHECK(hipHostMalloc((void**)&e->ld_pin, sizeof(LoopDev) * mst_engine::LD_SLOTS, hipHostMallocDefault));
    for (int i = 0; i < mst_engine::LD_SLOTS; i++) HIPCHECK(hipEventCreateWithFlags(&e->ld_ev[i], hipEventDisableTiming));
    if (const char* v = getenv("MST_GRAPH")) e->graph_on = atoi(v) != 0;
    if (const char* v = getenv("MST_GRAPH_STEPS")) { int n = atoi(v); e->graph_steps = n < 2 ? 2 : (n > 100 ? 100 : n); }
    *out = e;
    return 0;
}

extern "C" void mst_engine_destroy(mst_engine* e) {
    if (!e) return;
    for (int l = 0; l < e->cfg.num_layers; l++) {
        LayerW& w = e->L[l];
        void* p[] = {w.w_in, w.w_out, w.w1, w.w2, w.b_in, w.b_out, w.b1, w.b2, w.g1, w.be1, w.g2, w.be2,
                     w.w_inT, w.w_outT, w.w1T, w.w2T, w.wtail, w.wtail_bwd, w.wqkv, w.w_in_lo, w.w_out_lo, w.w1_lo, w.w2_lo, w.wsm_in, w.wsm_out, w.wsm_1, w.wsm_2};
        for (void* q : p) (void)hipFree(q);
    }
    {
        TrainWS& t = e->tw;
        void* p[] = {t.g0, t.g1, t.dbr2[0], t.dbr2[1], t.dbr1[0], t.dbr1[1], t.dpre[0], t.dpre[1], t.dqkv[0], t.dqkv[1],
                     t.datt, t.part, t.zeros, t.gscale, t.amax, t.ln_part, t.cs_part, t.hidr[0], t.hidr[1], t.ln_part2};
        if (t.ev_ready) (void)hipEventDestroy(t.ev_ready);
        for (int i = 0; i < 16; i++) if (t.ev_layer[i]) (void)hipEventDestroy(t.ev_layer[i]);
        for (int i = 0; i < 2; i++) if (t.ev_side[i]) (void)hipEventDestroy(t.ev_side[i]);
        for (void* q : p) (void)hipFree(q);
    }
    void* p[] = {e->w_pose_in_pk, e->w_pose_out_pk, e->w_pose_in_lo, e->w_pose_out_lo, e->w_pose_in, e->b_pose_in, e->w_pose_out, e->b_pose_out, e->w_pose_inT, e->w_pose_outT, e->w_t0, e->b_t0, e->w_t2, e->b_t2,
                 e->w_text, e->b_text, e->pe, e->hl, e->hx, e->hx2, e->hl2, e->qkv, e->att, e->hid, e->xt, e->xt_lo, e->gelu_tab, e->temb_hid, e->temb, e->textproj, e->zacc, e->temb_table, e->temb_table_hid};
    for (void* q : p) (void)hipFree(q);
    for (int i = 0; i < mst_engine::MAX_SLICES - 1; i++) {
        if (e->aux_stream[i]) (void)hipStreamDestroy(e->aux_stream[i]);
        if (e->ev_join[i]) (void)hipEventDestroy(e->ev_join[i]);
    }
    if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
    if (e->trunk_cnt) (void)hipFree(e->trunk_cnt);
    if (e->trunk_layers) (void)hipFree(e->trunk_layers);
    if (e->trunk_err) (void)hipHostFree(e->trunk_err);
    if (e->gexec) (void)hipGraphExecDestroy(e->gexec);
    if (e->loop_stream) (void)hipStreamDestroy(e->loop_stream);
    if (e->ev_in) (void)hipEventDestroy(e->ev_in);
    if (e->ev_out) (void)hipEventDestroy(e->ev_out);
    (void)hipFree(e->ld_dev);
    (void)hipFree(e->rowflag);
    if (e->ld_pin) (void)hipHostFree(e->ld_pin);
    for (int i = 0; i < mst_engine::LD_SLOTS; i++) if (e->ld_ev[i]) (void)hipEventDestroy(e->ld_ev[i]);
    for (auto& pp : e->prof_pts) {
        (void)hipEventDestroy(pp.a);
        (void)hipEventDestroy(pp.b);
    }
    delete e;
}

// ------------------------------------------------------------------------------------------ weights
static int put_matrix(const float* src, int N, int K, f16* dst, int Npad, int Kpad, hipStream_t st, f16* dst_lo = nullptr) {
    hipLaunchKernelGGL(k_convert_pad, dim3(1024), dim3(256), 0, st, src, N, K, dst, Npad, Kpad, dst_lo);
    HIPCHECK(hipGetLastError());
    return 0;
}
static int put_matrix_t(const float* src, int N, int K, f16* dst, hipStream_t st) {      // [N][K] f32 -> [K][N] f16
    hipLaunchKernelGGL(k_convert_transpose, dim3((K + 31) / 32, (N + 31) / 32), dim3(256), 0, st, src, N, K, dst, N);
    HIPCHECK(hipGetLastError());
    return 0;
}
static int put_vector(const float* src, int n, float* dst, int npad, hipStream_t st) {
    hipLaunchKernelGGL(k_copy_pad_f32, dim3((npad + 255) / 256), dim3(256), 0, st, src, n, dst, npad);
    HIPCHECK(hipGetLastError());
    return 0;
}
static bool shape_is(const int64_t* s, int nd, int64_t a, int64_t b = -1) {
    if (b < 0) return nd == 1 && s[0] == a;
    return nd == 2 && s[0] == a && s[1] == b;
}

extern "C" int mst_load_weight(mst_engine* e, const char* name, const float* src, const int64_t* shape, int32_t ndim,
                               void* stream) {
    if (!e || !name || !src || !shape) return fail("mst_load_weight: null argument");
    hipStream_t st = (hipStream_t)stream;
    ON_DEVICE(e->cfg.device);
    const int F = e->cfg.feats, C = e->cfg.clip_dim;
    std::string n(name);
    int rc = -1;
    int layer = -1;
    char rest[128] = "";
    if (sscanf(name, "seqTransEncoder.layers.%d.%127s", &layer, rest) == 2) {
        if (layer < 0 || layer >= e->cfg.num_layers) return fail("mst_load_weight: layer %d out of range", layer);
        LayerW& w = e->L[layer];
        std::string r(rest);
#define MAT(key, N_, K_, dst) if (r == key) { if (!shape_is(shape, ndim, N_, K_)) return fail("mst_load_weight: %s: bad shape", name); rc = put_matrix(src, N_, K_, dst, N_, K_, st, e->precise ? dst##_lo : nullptr); if (!rc) rc = put_matrix_t(src, N_, K_, dst##T, st); repack = dst != w.w_in; is_gemm_w = true; }
        bool repack = false, is_gemm_w = false;
#define VEC(key, N_, dst) if (r == key) { if (!shape_is(shape, ndim, N_)) return fail("mst_load_weight: %s: bad shape", name); rc = put_vector(src, N_, dst, N_, st); }
        MAT("self_attn.in_proj_weight", 3 * MST_D, MST_D, w.w_in)
        VEC("self_attn.in_proj_bias", 3 * MST_D, w.b_in)
        MAT("self_attn.out_proj.weight", MST_D, MST_D, w.w_out)
        VEC("self_attn.out_proj.bias", MST_D, w.b_out)
        MAT("linear1.weight", MST_FF, MST_D, w.w1)
        VEC("linear1.bias", MST_FF, w.b1)
        MAT("linear2.weight", MST_D, MST_FF, w.w2)
        VEC("linear2.bias", MST_D, w.b2)
        VEC("norm1.weight", MST_D, w.g1)
        VEC("norm1.bias", MST_D, w.be1)
        VEC("norm2.weight", MST_D, w.g2)
        VEC("norm2.bias", MST_D, w.be2)
#undef MAT
#undef VEC
        // The fused kernels' packed copies (k_pack_qkv, k_pack_tail) are made once per re-upload and layer, and only if a sampling launch
        // follows (ensure_packed): a fine-tune iteration re-uploads all 96 tensors and its training node reads the plain matrices.
        if (rc == 0 && r == "self_attn.in_proj_weight") w.qkv_dirty = true;
        if (rc == 0 && repack) w.tail_dirty = w.tailb_dirty = true;
        if (rc == 0 && is_gemm_w) w.small_dirty = true;
        if (rc == 0 && is_gemm_w) { if (e->precise) e->lo_missing.erase(n); else e->lo_missing.insert(n); }
    } else if (n == "input_process.poseEmbedding.weight") {
        if (!shape_is(shape, ndim, MST_D, F)) return fail("mst_load_weight: %s: bad shape", name);
        rc = put_matrix(src, MST_D, F, e->w_pose_in, MST_D, e->kin_pad, st, e->precise ? e->w_pose_in_lo : nullptr);
        if (!rc) { if (e->precise) e->lo_missing.erase(n); else e->lo_missing.insert(n); e->pose_in_dirty = true; }
        if (!rc) {      // [512][F] -> [F (padded to fout_pad)][512]
            hipLaunchKernelGGL(k_convert_transpose, dim3((F + 31) / 32, (MST_D + 31) / 32), dim3(256), 0, st, src, MST_D, F, e->w_pose_inT, MST_D);
            HIPCHECK(hipGetLastError());
        }
    } else if (n == "input_process.poseEmbedding.bias") {
        if (!shape_is(shape, ndim, MST_D)) return fail("mst_load_weight: %s: bad shape", name);
        rc = put_vector(src, MST_D, e->b_pose_in, MST_D, st);
    } else if (n == "output_process.poseFinal.weight") {
        if (!shape_is(shape, ndim, F, MST_D)) return fail("mst_load_weight: %s: bad shape", name);
        rc = put_matrix(src, F, MST_D, e->w_pose_out, e->fout_pad, MST_D, st, e->precise ? e->w_pose_out_lo : nullptr);
        if (!rc) { if (e->precise) e->lo_missing.erase(n); else e->lo_missing.insert(n); e->pose_out_dirty = true; }
        if (!rc) {      // [F][512] -> [512][F (padded to kin_pad)]
            hipLaunchKernelGGL(k_convert_transpose, dim3((MST_D + 31) / 32, (F + 31) / 32), dim3(256), 0, st, src, F, MST_D, e->w_pose_outT, e->kin_pad);
            HIPCHECK(hipGetLastError());
        }
    } else if (n == "output_process.poseFinal.bias") {
        if (!shape_is(shape, ndim, F)) return fail("mst_load_weight: %s: bad shape", name);
        rc = put_vector(src, F, e->b_pose_out, e->fout_pad, st);
    } else if (n == "embed_timestep.time_embed.0.weight" || n == "embed_timestep.time_embed.2.weight") {
        if (!shape_is(shape, ndim, MST_D, MST_D)) return fail("mst_load_weight: %s: bad shape", name);
        rc = put_vector(src, MST_D * MST_D, n[26] == '0' ? e->w_t0 : e->w_t2, MST_D * MST_D, st);
        e->temb_table_valid = false;
    } else if (n == "embed_timestep.time_embed.0.bias" || n == "embed_timestep.time_embed.2.bias") {
        if (!shape_is(shape, ndim, MST_D)) return fail("mst_load_weight: %s: bad shape", name);
        rc = put_vector(src, MST_D, n[26] == '0' ? e->b_t0 : e->b_t2, MST_D, st);
        e->temb_table_valid = false;
    } else if (n == "embed_text.weight") {
        if (!shape_is(shape, ndim, MST_D, C)) return fail("mst_load_weight: %s: bad shape", name);
        rc = put_vector(src, MST_D * C, e->w_text, MST_D * C, st);
    } else if (n == "embed_text.bias") {
        if (!shape_is(shape, ndim, MST_D)) return fail("mst_load_weight: %s: bad shape", name);
        rc = put_vector(src, MST_D, e->b_text, MST_D, st);
    } else if (n == "sequence_pos_encoder.pe") {
        int64_t rows = ndim == 3 ? shape[0] : (ndim == 2 ? shape[0] : -1);
        int64_t cols = ndim == 3 ? shape[1] * shape[2] : (ndim == 2 ? shape[1] : -1);
        if (cols != MST_D || rows < e->cfg.pe_len) return fail("mst_load_weight: %s: bad shape", name);
        rc = put_vector(src, e->cfg.pe_len * MST_D, e->pe, e->cfg.pe_len * MST_D, st);
        e->temb_table_valid = false;
    }
    if (rc == -1) return fail("mst_load_weight: unknown tensor name '%s'", name);
    if (rc) return rc;
    bool seen = false;
    for (auto& s : e->loaded) seen |= (s == n);
    if (!seen) e->loaded.push_back(n);
    return 0;
}

// All 12 tensors of every encoder layer in one launch per 8 layers (k_upload_layers): what a fine-tune iteration does after every
// optimizer step.  srcs: num_layers x 12 device pointers in the order of mst_load_weight's layer list (in_proj_weight, in_proj_bias,
// out_proj.weight, out_proj.bias, linear1.weight, linear1.bias, linear2.weight, linear2.bias, norm1.weight, norm1.bias, norm2.weight,
// norm2.bias).  Same results as 12 x num_layers mst_load_weight calls (precise mode: use those -- this entry writes no lo halves).
extern "C" int mst_load_layers(mst_engine* e, const float* const* srcs, void* stream) {
    if (!e || !srcs) return fail("mst_load_layers: null argument");
    if (e->precise) return fail("mst_load_layers: precise mode needs the weights' lo halves: upload with mst_load_weight");
    static_assert(MST_D == 512 && MST_FF == 1024, "k_upload_layers' tile map");
    static const char* kNames[12] = {"self_attn.in_proj_weight", "self_attn.in_proj_bias", "self_attn.out_proj.weight", "self_attn.out_proj.bias",
                                     "linear1.weight", "linear1.bias", "linear2.weight", "linear2.bias", "norm1.weight", "norm1.bias",
                                     "norm2.weight", "norm2.bias"};
    hipStream_t st = (hipStream_t)stream;
    ON_DEVICE(e->cfg.device);
    const int nl = e->cfg.num_layers;
    for (int i = 0; i < 12 * nl; i++) if (!srcs[i]) return fail("mst_load_layers: null tensor %d", i);
    for (int l0 = 0; l0 < nl; l0 += 8) {
        UpArgs a{};
        const int n = nl - l0 < 8 ? nl - l0 : 8;
        for (int j = 0; j < n; j++) {
            LayerW& w = e->L[l0 + j];
            UpLayer& u = a.L[j];
            for (int k = 0; k < 12; k++) u.src[k] = srcs[(l0 + j) * 12 + k];
            u.w[0] = w.w_in; u.w[1] = w.w_out; u.w[2] = w.w1; u.w[3] = w.w2;
            u.wT[0] = w.w_inT; u.wT[1] = w.w_outT; u.wT[2] = w.w1T; u.wT[3] = w.w2T;
            u.v[0] = w.b_in; u.v[1] = w.b_out; u.v[2] = w.b1; u.v[3] = w.b2; u.v[4] = w.g1; u.v[5] = w.be1; u.v[6] = w.g2; u.v[7] = w.be2;
        }
        hipLaunchKernelGGL(k_upload_layers, dim3(2049, n), dim3(256), 0, st, a);
        HIPCHECK(hipGetLastError());
    }
    for (int l = 0; l < nl; l++) {
        e->L[l].qkv_dirty = e->L[l].tail_dirty = e->L[l].tailb_dirty = e->L[l].small_dirty = true;
        for (int k = 0; k < 12; k++) {
            char name[160];
            snprintf(name, sizeof(name), "seqTransEncoder.layers.%d.%s", l, kNames[k]);
            std::string n(name);
            if ((k & 1) == 0 && k < 8) e->lo_missing.insert(n);
            bool seen = false;
            for (auto& s2 : e->loaded) seen |= (s2 == n);
            if (!seen) e->loaded.push_back(n);
        }
    }
    return 0;
}

extern "C" int mst_weights_complete(const mst_engine* e) {
    if (!e) return fail("mst_weights_complete: null engine");
    size_t want = (size_t)e->cfg.num_layers * 12 + 11;
    if (e->loaded.size() != want) return fail("mst_weights_complete: %zu of %zu tensors loaded", e->loaded.size(), want);
    return 0;
}

// ------------------------------------------------------------------------------------------ profiling
extern "C" int mst_profile_enable(mst_engine* e, int32_t on) {
    if (!e) return fail("mst_profile_enable: null engine");
    e->prof_on = on > 0 ? 1 : 0;          // on = N > 0: instrument every N-th step of a loop
    e->prof_period = on > 0 ? on : 16;
    if (e->prof_on && e->prof_pts.empty()) {
        e->prof_pts.resize(8192);
        for (auto& p : e->prof_pts) {
            HIPCHECK(hipEventCreate(&p.a));
            HIPCHECK(hipEventCreate(&p.b));
        }
    }
    e->prof_used = 0;
    for (int i = 0; i < FAM_COUNT; i++) { e->prof_ms[i] = 0; e->prof_n[i] = 0; }
    if (e->prof_on && e->prof_overhead_us < 0.f) {
        // what an event pair measures around NOTHING on an otherwise busy-free stream: the fixed part of every timed launch
        // (rocprofv3's kernel durations are shorter than event-bracketed ones by about this much)
        ON_DEVICE(e->cfg.device);
        float v[16];
        for (int i = 0; i < 16; i++) {
            HIPCHECK(hipEventRecord(e->prof_pts[0].a, e->loop_stream));
            HIPCHECK(hipEventRecord(e->prof_pts[0].b, e->loop_stream));
            HIPCHECK(hipEventSynchronize(e->prof_pts[0].b));
            HIPCHECK(hipEventElapsedTime(&v[i], e->prof_pts[0].a, e->prof_pts[0].b));
        }
        std::sort(v, v + 16);
        e->prof_overhead_us = 1e3f * v[8];
    }
    return 0;
}

extern "C" float mst_profile_event_overhead_us(const mst_engine* e) { return e ? e->prof_overhead_us : -1.f; }

extern "C" int mst_profile_read(mst_engine* e, const char** names, float* total_ms, int32_t* launches, int32_t cap) {
    if (!e) return -1;
    // caller has synchronised the stream
    for (size_t i = 0; i < e->prof_used; i++) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e->prof_pts[i].a, e->prof_pts[i].b) == hipSuccess) {
            e->prof_ms[e->prof_pts[i].fam] += ms;
            e->prof_n[e->prof_pts[i].fam] += 1;
        }
    }
    e->prof_used = 0;
    int n = FAM_COUNT < cap ? FAM_COUNT : cap;
    for (int i = 0; i < n; i++) {
        names[i] = kFamilyNames[i];
        total_ms[i] = (float)e->prof_ms[i];
        launches[i] = e->prof_n[i];
    }
    return n;
}

struct ProfScope {
    mst_engine* e; hipStream_t st; ProfPoint* p = nullptr;
    ProfScope(mst_engine* e_, int fam, hipStream_t st_) : e(e_), st(st_) {
        if (e->prof_on && e->prof_now && e->prof_used < e->prof_pts.size()) {
            p = &e->prof_pts[e->prof_used++];
            p->fam = fam;
            (void)hipEventRecord(p->a, st);
        }
    }
    ~ProfScope() { if (p) (void)hipEventRecord(p->b, st); }
};

// ------------------------------------------------------------------------------------------ launches
template <int BT, int BF, int MT, int NT, int NS, int NX, int BK = 32, int XS = 1, class SRC, class EPI>
static int launch_gemm_dma(dim3 grid, const SRC& xs, const f16* W, int ldw, int K, const EPI& epi, hipStream_t st, int xcd_ny = 0) {
    using TL = DTile<BT, BF, MT, NT, NS, NX, BK, XS>;
    static_assert(EPI::template smem_bytes<BT, BF>() <= TL::SMEM, "epilogue tile must fit the ring");
    static_assert(TL::SMEM <= 163840, "ring exceeds the 160 KiB LDS");
    auto kern = k_gemm_dma<BT, BF, MT, NT, NS, NX, SRC, EPI, BK, XS>;
    if (XS == 2 && !xs.Xlo) return fail("gemm: the split-activation tile needs the lo tensor");
    CHECK(ensure_dyn_lds((const void*)kern, TL::SMEM));
    if (K < BK * (NS - 1) || (K % BK)) return fail("gemm: K=%d unsupported by the %d-slot ring", K, NS);
    hipLaunchKernelGGL(kern, grid, dim3(512), TL::SMEM, st, xs, W, ldw, K, xcd_ny, epi);
    HIPCHECK(hipGetLastError());
    return 0;
}

// Wide (128 x 256) GEMM launch.  Few token rows = few workgroups, each walking the whole K loop alone: the loop is then
// bound by the DMA round trip per slab, not by bandwidth, so small launches use 64-deep slabs (half as many round trips,
// one workgroup per CU is plenty); large launches keep 32-deep slabs and two co-resident workgroups per CU.
#define SMALL_M 2048
#define LN_BWD_WIDE_M 4096        // k_ln_bwd: sixteen waves per block above this many rows (four below: a clip or a few, one row per wave)
template <class SRC, class EPI>
static int launch_wide(int M, int ny, const SRC& xs, const f16* W, int ldw, int K, const EPI& epi, hipStream_t st) {
    if (M <= SMALL_M && (K % 64) == 0 && K >= 128)
        return launch_gemm_dma<WIDE_BT, 256, WIDE_BT / 64, 2, 3, 1, 64>(wide_grid(M, ny), xs, W, ldw, K, epi, st, WIDE_XCD ? ny : 0);
    return launch_gemm_dma<WIDE_BT, 256, WIDE_BT / 64, 2, WIDE_NS, 1>(wide_grid(M, ny), xs, W, ldw, K, epi, st, WIDE_XCD ? ny : 0);
}

// Few token rows (a clip or two): 64 x 128 tiles over a 2-D grid so that tens of workgroups share the weight matrix
// instead of 4-12 of them each streaming all of it; LayerNorm then runs as its own row-wise kernel (k_ln_rows).
template <class SRC, class EPI>
static int launch_small(int M, int N, const SRC& xs, const f16* W, int ldw, int K, const EPI& epi, hipStream_t st) {
    return launch_gemm_dma<64, 128, 1, 1, 3, 1, 64>(dim3((M + 63) / 64, N / 128), xs, W, ldw, K, epi, st, 0);
}

// The small-launch GEMMs of mst_small.h: 64 x 128 tiles, the token tile resident in LDS, the weights streamed as fragments.
static int g_rows_ntb1_m = [] { const char* v = getenv("MST_SMALL_NTB1_M"); return v ? atoi(v) : 800; }();
// (tools/experiments/r4_ntb1_sweep.sh, r4_ntb2_sweep.sh: 16-token tiles ahead through 4 clips x 197 rows, behind at 6; 32-token tiles 4 % ahead at 8 and 9 clips, level at 6, behind at 5)
static int g_rows_ntb2_from = [] { const char* v = getenv("MST_SMALL_NTB2_FROM"); return v ? atoi(v) : 1300; }();
template <int KS, int MODE>
static int launch_rows_gemm(int M, int N, const f16* X, const f16* wpk, const float* bias, void* out, int ldo, hipStream_t st, const LnRows* ln = nullptr,
                            const FfnTrain* ft = nullptr) {
    constexpr int smem = 64 * (KS / 16) * 1024;
    if constexpr (KS == 16 && (MODE == 0 || MODE == 1 || MODE == 3)) {
        if (ln) {                                                  // 16-token tiles (mst_small.h)
            hipLaunchKernelGGL((k_rows_gemm<KS, MODE, 1, 1>), dim3((M + 15) / 16, N / 128), dim3(512), 16 * 1024, st, X, wpk, bias, out, ldo, M, *ln, ft ? *ft : FfnTrain{});
            HIPCHECK(hipGetLastError());
            return 0;
        }
    }
    {
        if (M <= g_rows_ntb1_m) {                                  // a clip or two: 16-token tiles here too (a quarter of the DMA burst in front of the first MFMA)
            hipLaunchKernelGGL((k_rows_gemm<KS, MODE, 0, 1>), dim3((M + 15) / 16, N / 128), dim3(512), smem / 4, st, X, wpk, bias, out, ldo, M, LnRows{}, ft ? *ft : FfnTrain{});
            HIPCHECK(hipGetLastError());
            return 0;
        }
        if (M > g_rows_ntb2_from) {
            hipLaunchKernelGGL((k_rows_gemm<KS, MODE, 0, 2>), dim3((M + 31) / 32, N / 128), dim3(512), smem / 2, st, X, wpk, bias, out, ldo, M, LnRows{}, ft ? *ft : FfnTrain{});
            HIPCHECK(hipGetLastError());
            return 0;
        }
    }
    CHECK(ensure_dyn_lds((const void*)k_rows_gemm<KS, MODE>, smem));
    hipLaunchKernelGGL((k_rows_gemm<KS, MODE>), dim3((M + 63) / 64, N / 128), dim3(512), smem, st, X, wpk, bias, out, ldo, M, LnRows{}, ft ? *ft : FfnTrain{});
    HIPCHECK(hipGetLastError());
    return 0;
}
template <int NKT>
static int launch_attn_n(const f16* qkv, f16* out, int S, int rows, hipStream_t st, int qsplit, f16* out_lo) {
    auto kern = k_attention<NKT>;
    const int smem = NKT * 32 * 256 * 2;
    CHECK(ensure_dyn_lds((const void*)kern, smem));
    hipLaunchKernelGGL(kern, dim3(rows * MST_H, qsplit ? NKT : 1), dim3(512), smem, st, qkv, out, S, qsplit, out_lo);
    HIPCHECK(hipGetLastError());
    return 0;
}

template <int NKT>
static int launch_qkv_attn_n(const f16* hx, const f16* w_in, const float* b_in, f16* out, int S, int rows, hipStream_t st) {
    auto kern = k_qkv_attention<NKT>;
    constexpr int smem = QATile<NKT>::SMEM;
    static_assert(smem <= 163840, "fused QKV+attention exceeds the 160 KiB LDS");
    CHECK(ensure_dyn_lds((const void*)kern, smem));
    hipLaunchKernelGGL(kern, dim3(rows * MST_H), dim3(512), smem, st, hx, w_in, b_in, out, S);
    HIPCHECK(hipGetLastError());
    return 0;
}

static int launch_qkv_attn(const f16* hx, const f16* w_in, const float* b_in, f16* out, int S, int rows, hipStream_t st) {
    switch ((S + 31) / 32) {
        case 1: return launch_qkv_attn_n<1>(hx, w_in, b_in, out, S, rows, st);
        case 2: return launch_qkv_attn_n<2>(hx, w_in, b_in, out, S, rows, st);
        case 3: return launch_qkv_attn_n<3>(hx, w_in, b_in, out, S, rows, st);
        case 4: return launch_qkv_attn_n<4>(hx, w_in, b_in, out, S, rows, st);
        case 5: return launch_qkv_attn_n<5>(hx, w_in, b_in, out, S, rows, st);
        case 6: return launch_qkv_attn_n<6>(hx, w_in, b_in, out, S, rows, st);
        case 7: return launch_qkv_attn_n<7>(hx, w_in, b_in, out, S, rows, st);
    }
    return fail("attention: S=%d unsupported", S);
}

template <int NT16>
static int launch_qkv_attn2_n(const f16* hx, const f16* wq, const float* b_in, f16* out, int S, int rows, hipStream_t st) {
    auto kern = k_qkv_attention2<NT16>;
    constexpr int smem = QA2Tile<NT16>::SMEM;
    CHECK(ensure_dyn_lds((const void*)kern, smem));
    hipLaunchKernelGGL(kern, dim3(rows * MST_H), dim3(512), smem, st, hx, wq, b_in, out, S);
    HIPCHECK(hipGetLastError());
    return 0;
}
// The role-swapped kernel holds K, V and Q images of 16 ceil(S / 16) rows in LDS: up to S = 208 (the model's 196 frames + 1).
static bool qkv_attn2_fits(int S) { return S <= 208; }
static int launch_qkv_attn2(const f16* hx, const f16* wq, const float* b_in, f16* out, int S, int rows, hipStream_t st) {
    const int n16 = (S + 15) / 16;
    switch (n16 == 13 ? 13 : (n16 + 1) / 2 * 2) {
        case 2: return launch_qkv_attn2_n<2>(hx, wq, b_in, out, S, rows, st);
        case 4: return launch_qkv_attn2_n<4>(hx, wq, b_in, out, S, rows, st);
        case 6: return launch_qkv_attn2_n<6>(hx, wq, b_in, out, S, rows, st);
        case 8: return launch_qkv_attn2_n<8>(hx, wq, b_in, out, S, rows, st);
        case 10: return launch_qkv_attn2_n<10>(hx, wq, b_in, out, S, rows, st);
        case 12: return launch_qkv_attn2_n<12>(hx, wq, b_in, out, S, rows, st);
        case 13: return launch_qkv_attn2_n<13>(hx, wq, b_in, out, S, rows, st);
    }
    return fail("attention: S=%d unsupported", S);
}

static int launch_attn(const f16* qkv, f16* out, int S, int rows, hipStream_t st, int qsplit = 0, f16* out_lo = nullptr) {
    switch ((S + 31) / 32) {
        case 1: return launch_attn_n<1>(qkv, out, S, rows, st, qsplit, out_lo);
        case 2: return launch_attn_n<2>(qkv, out, S, rows, st, qsplit, out_lo);
        case 3: return launch_attn_n<3>(qkv, out, S, rows, st, qsplit, out_lo);
        case 4: return launch_attn_n<4>(qkv, out, S, rows, st, qsplit, out_lo);
        case 5: return launch_attn_n<5>(qkv, out, S, rows, st, qsplit, out_lo);
        case 6: return launch_attn_n<6>(qkv, out, S, rows, st, qsplit, out_lo);
        case 7: return launch_attn_n<7>(qkv, out, S, rows, st, qsplit, out_lo);
    }
    return fail("attention: S=%d unsupported", S);
}

static int rowwise_linear(const float* in, int ldin, const long long* gather, const float* rowscale, int rows_zero_from,
                          const float* W, const float* b, int K, int N, int act, float* out, int in_row_mod, int rows,
                          hipStream_t st, int rowscale_is_drop = 0) {
    // a wave per output where the rows are few (64 rows x 128 = 8 K blocks; the hoisted timestep rows of a 1000-step loop keep 16 per row)
    const int gy = rows <= 64 ? (N + 3) / 4 : 16;
    hipLaunchKernelGGL(k_rowwise_linear, dim3(rows, gy), dim3(256), 0, st, in, ldin, gather, rowscale, rows_zero_from, W, b,
                       K, N, act, out, in_row_mod, rowscale_is_drop);
    HIPCHECK(hipGetLastError());
    return 0;
}

// the timestep embedding of every timestep there is (rows of the positional table), once per load of the timestep MLP
static int ensure_temb_table(mst_engine* e, hipStream_t st) {
    if (e->temb_table_valid) return 0;
    const int rows = e->cfg.pe_len;
    if (!e->temb_table) {
        CHECK(dmalloc(&e->temb_table, (size_t)rows * MST_D));
        CHECK(dmalloc(&e->temb_table_hid, (size_t)rows * MST_D));
    }
    CHECK(rowwise_linear(e->pe, MST_D, nullptr, nullptr, rows, e->w_t0, e->b_t0, MST_D, MST_D, 1, e->temb_table_hid, 0, rows, st));
    CHECK(rowwise_linear(e->temb_table_hid, MST_D, nullptr, nullptr, rows, e->w_t2, e->b_t2, MST_D, MST_D, 0, e->temb_table, 0, rows, st));
    e->temb_table_valid = true;
    return 0;
}

// timestep embedding rows: temb[j] = W2 silu(W0 pe[tidx[j]] + b0) + b2   (mdm :415-422)
static int timestep_rows(mst_engine* e, const long long* tidx_dev, int rows, hipStream_t st) {
    if (rows > e->temb_cap) return fail("timestep_rows: %d rows exceed capacity %d", rows, e->temb_cap);
    CHECK(rowwise_linear(e->pe, MST_D, tidx_dev, nullptr, rows, e->w_t0, e->b_t0, MST_D, MST_D, 1, e->temb_hid, 0, rows, st));
    CHECK(rowwise_linear(e->temb_hid, MST_D, nullptr, nullptr, rows, e->w_t2, e->b_t2, MST_D, MST_D, 0, e->temb, 0, rows, st));
    return 0;
}

extern "C" int mst_set_text(mst_engine* e, const float* text_emb, const float* keep, int32_t batch, int32_t cfg, void* stream) {
    if (!e || !text_emb) return fail("mst_set_text: null argument");
    int rows = cfg ? 2 * batch : batch;
    if (batch < 1 || rows > e->cfg.max_rows) return fail("mst_set_text: %d rows exceed max_rows %d", rows, e->cfg.max_rows);
    hipStream_t st = (hipStream_t)stream;
    ON_DEVICE(e->cfg.device);
    CHECK(rowwise_linear(text_emb, e->cfg.clip_dim, nullptr, keep, cfg ? batch : rows, e->w_text, e->b_text, e->cfg.clip_dim,
                         MST_D, 0, e->textproj, batch, rows, st));
    e->text_batch = batch;
    e->text_cfg = cfg ? 1 : 0;
    return 0;
}

extern "C" int mst_set_text_dropped(mst_engine* e, const float* text_emb, const float* drop, int32_t batch, void* stream) {
    if (!e || !text_emb || !drop) return fail("mst_set_text_dropped: null argument");
    if (batch < 1 || batch > e->cfg.max_rows) return fail("mst_set_text_dropped: %d rows exceed max_rows %d", batch, e->cfg.max_rows);
    hipStream_t st = (hipStream_t)stream;
    ON_DEVICE(e->cfg.device);
    CHECK(rowwise_linear(text_emb, e->cfg.clip_dim, nullptr, drop, batch, e->w_text, e->b_text, e->cfg.clip_dim, MST_D, 0, e->textproj, batch, batch, st, 1));
    e->text_batch = batch;
    e->text_cfg = 0;
    return 0;
}

// A slice of the workspace: clips [r0, r0 + n) of the batch get their own rows of every buffer, so slices can
// run concurrently on different streams (tiles may over-READ into a neighbour's rows; they never write them).
struct WS {
    f16 *hl, *hx, *qkv, *att, *hid, *xt; float* textproj; float* zacc; f16* xt_lo; f16 *hx2, *hl2;
};
static WS ws_slice(const mst_engine* e, int r0, int T) {
    const size_t row = (size_t)r0 * (T + 1);
    return WS{e->hl + row * MST_D, e->hx + row * MST_D, e->qkv + row * 3 * MST_D, e->att + row * MST_D, e->hid + row * MST_FF,
              e->xt + (size_t)r0 * T * e->kin_pad, e->textproj + (size_t)r0 * MST_D, e->zacc + row * MST_D,
              e->xt_lo + (size_t)r0 * T * e->kin_pad, e->hx2 + row * MST_D, e->hl2 + row * MST_D};
}

// Packed weight copies of the two fused kernels, refreshed where stale: on the stream the sampling launch is about to use (the caller
// orders that stream behind its weight uploads, as for the plain matrices).
static int embed_out_nbw(const mst_engine* e) { return (e->cfg.feats + 127) / 128; }     // 16-feature blocks per wave of k_embed_out
// (layers = false: the training node's model calls -- they run the two projections' fast kernels but read the layers' plain matrices,
// and every fine-tune iteration re-uploads those: their packed copies are made only when a sampling launch follows)
// (on the stream every slice of a loop forks from: mst_sample_loop packs before the fork, so no slice reads fragments another stream is still writing)
static int ensure_small_packed(mst_engine* e, int l, hipStream_t st) {
    LayerW& w = e->L[l];
    if (!w.small_dirty) return 0;
    hipLaunchKernelGGL(k_pack_blocks, dim3(192), dim3(256), 0, st, w.w_in, MST_D, 3 * MST_D, MST_D, w.wsm_in);
    hipLaunchKernelGGL(k_pack_blocks, dim3(64), dim3(256), 0, st, w.w_out, MST_D, MST_D, MST_D, w.wsm_out);
    hipLaunchKernelGGL(k_pack_blocks, dim3(128), dim3(256), 0, st, w.w1, MST_D, MST_FF, MST_D, w.wsm_1);
    hipLaunchKernelGGL(k_pack_blocks, dim3(128), dim3(256), 0, st, w.w2, MST_FF, MST_D, MST_FF, w.wsm_2);
    HIPCHECK(hipGetLastError());
    w.small_dirty = false;
    return 0;
}

// every stale layer's four matrices in ONE launch (the fine-tune loop re-uploads all of them each iteration)
static int ensure_small_packed_all(mst_engine* e, hipStream_t st) {
    PackJobs jobs{};
    int n = 0;
    for (int l = 0; l < e->cfg.num_layers && l < 8; l++) {
        LayerW& w = e->L[l];
        if (!w.small_dirty) continue;
        jobs.j[n++] = PackJob{w.w_in, w.wsm_in, MST_D, 3 * MST_D, MST_D, 0};
        jobs.j[n++] = PackJob{w.w_out, w.wsm_out, MST_D, MST_D, MST_D, 0};
        jobs.j[n++] = PackJob{w.w1, w.wsm_1, MST_D, MST_FF, MST_D, 0};
        jobs.j[n++] = PackJob{w.w2, w.wsm_2, MST_FF, MST_D, MST_FF, 0};
    }
    if (n) {
        hipLaunchKernelGGL(k_pack_blocks_multi, dim3(48, n), dim3(256), 0, st, jobs);
        HIPCHECK(hipGetLastError());
        for (int l = 0; l < e->cfg.num_layers && l < 8; l++) e->L[l].small_dirty = false;
    }
    for (int l = 8; l < e->cfg.num_layers; l++) CHECK(ensure_small_packed(e, l, st));
    return 0;
}
static int ensure_packed(mst_engine* e, hipStream_t st, bool layers = true) {
    if (e->pose_in_dirty) {
        hipLaunchKernelGGL(k_pack_wave_blocks, dim3(256), dim3(256), 0, st, e->w_pose_in, e->kin_pad, MST_D, e->kin_pad / 32, 4, e->w_pose_in_pk);
        HIPCHECK(hipGetLastError());
        e->pose_in_dirty = false;
    }
    if (e->pose_out_dirty) {
        hipLaunchKernelGGL(k_pack_wave_blocks, dim3(256), dim3(256), 0, st, e->w_pose_out, MST_D, e->fout_pad, MST_D / 32, embed_out_nbw(e), e->w_pose_out_pk);
        HIPCHECK(hipGetLastError());
        e->pose_out_dirty = false;
    }
    if (!layers) return 0;
    for (int l = 0; l < e->cfg.num_layers; l++) {
        LayerW& w = e->L[l];
        if (w.qkv_dirty) {
            hipLaunchKernelGGL(k_pack_qkv, dim3(384), dim3(256), 0, st, w.w_in, w.wqkv);
            HIPCHECK(hipGetLastError());                  // a flag is cleared only behind a launch that was accepted
            w.qkv_dirty = false;
        }
        if (w.tail_dirty) {
            hipLaunchKernelGGL(k_pack_tail, dim3(640), dim3(256), 0, st, w.w_out, w.w1, w.w2, w.wtail);
            HIPCHECK(hipGetLastError());
            w.tail_dirty = false;
        }
    }
    if (e->small_fast && e->small_m > 0) CHECK(ensure_small_packed_all(e, st));
    return 0;
}

// K6 + K7 + K8 of one layer as one launch (mst_tail.h): one workgroup per 64-token tile
static int launch_tail(const mst_engine* e, const LayerW& w, const WS& ws, int M, hipStream_t st) {
    static_assert(TailCfg::SMEM <= 163840, "fused layer tail exceeds the 160 KiB LDS");
    // Tile height (16 NTB tokens; mst_tail.h).  A launch that has the chip to itself and does not fill it runs on more, lower tiles; the
    // clip slices of a sampling loop share the chip (3 x 68 tiles of 64 tokens at the headline batch) and keep the 64-token tile: 48-token
    // tiles measured 99.4 against 105.3 clips/s there, 32-token tiles 88.3 (tools/experiments/r4_ntb_ab.sh).
    int ntb = e->tail_ntb;
    if (ntb == 0) {
        ntb = 4;
        if (e->cur_slices == 1) {
            if ((M + 31) / 32 <= 256) ntb = 2;
            else if ((M + 47) / 48 <= 256) ntb = 3;
        }
    }
    const int grid = (M + 16 * ntb - 1) / (16 * ntb);
#define TAIL_LAUNCH(N_)                                                                                                          \
    do {                                                                                                                         \
        CHECK(ensure_dyn_lds((const void*)k_layer_tail<N_>, TailCfg::SMEM));                                                     \
        hipLaunchKernelGGL(k_layer_tail<N_>, dim3(grid), dim3(512), TailCfg::SMEM, st, ws.att, w.wtail, w.b_out, w.g1, w.be1,     \
                           w.b1, w.b2, w.g2, w.be2, ws.hx, ws.hl, e->gelu_tab, M);                                               \
    } while (0)
    if (ntb == 2) TAIL_LAUNCH(2);
    else if (ntb == 3) TAIL_LAUNCH(3);
    else TAIL_LAUNCH(4);
#undef TAIL_LAUNCH
    HIPCHECK(hipGetLastError());
    return 0;
}

// The whole stack of a sampling step as one launch of resident groups (mst_trunk.h): frame counts whose token count is 13 blocks of 16
// (193 .. 208 tokens: the model's 196 frames), the default fused kernels, no debug stop, no instrumented step, at most 8 layers.
static bool trunk_groups_fit(const mst_engine* e, int S) {
    return e->trunk_groups && (S + 15) / 16 == 13 && e->fuse_qkv_attn == 1 && e->fuse_tail && e->tail_ntb == 0 && e->dbg_stage < 0 && !e->precise &&
           e->cfg.num_layers <= 8;
}
static int launch_trunk_groups(mst_engine* e, const WS& ws, int S, int rows, hipStream_t st) {
    using TT = TrunkTile<13>;
    if (e->trunk_err[0]) return fail("resident-group trunk: a hand-off wait gave up in an earlier launch (results of that loop are invalid)");
    TrunkArgs a{};
    if (!e->trunk_layers) {
        TrunkLayer tab[8];
        for (int l = 0; l < e->cfg.num_layers; l++) {
            const LayerW& w = e->L[l];
            tab[l] = TrunkLayer{w.wqkv, w.b_in, w.wtail, w.b_out, w.g1, w.be1, w.b1, w.b2, w.g2, w.be2};
        }
        HIPCHECK(hipMalloc(&e->trunk_layers, sizeof(tab)));
        HIPCHECK(hipMemcpy(e->trunk_layers, tab, sizeof(tab), hipMemcpyHostToDevice));
    }
    a.L = static_cast<const TrunkLayer*>(e->trunk_layers);
    a.hx = ws.hx; a.hl = ws.hl; a.att = ws.att; a.gelu_tab = e->gelu_tab;
    a.cnt = e->trunk_cnt + (size_t)((ws.hx - e->hx) / ((size_t)S * MST_D)) * 32;       // the slice's first clip
    a.err = e->trunk_err;
    a.S = S; a.nclips = rows; a.nlayers = e->cfg.num_layers;
    const int groups = rows < 64 ? rows : 64;
    CHECK(ensure_dyn_lds((const void*)k_trunk_groups<13>, TT::SMEM));
    hipLaunchKernelGGL(k_trunk_groups<13>, dim3(4 * groups), dim3(512), TT::SMEM, st, a);
    HIPCHECK(hipGetLastError());
    e->trunk_used = true;
    return 0;
}
// The error word is written by the kernel when it RUNS; a loop enqueues hundreds of steps ahead, so reading it at enqueue time is always
// stale.  Every ABI call that may have used the resident launch therefore ends with a synchronisation of its stream and
// this check: a give-up (the four blocks of a group not co-resident -- the launch rests on in-order dispatch of at most 256 one-per-CU
// blocks, which another process or a busy foreign stream on the same GPU can break) is an ERROR of that call, never a wrong sample.
static int trunk_settle(mst_engine* e, hipStream_t st, int rc) {
    if (!e->trunk_used) return rc;
    e->trunk_used = false;
    if (hipStreamSynchronize(st) != hipSuccess) return rc ? rc : fail("resident-group trunk: synchronisation failed");
    if (e->trunk_err[0]) return fail("resident-group trunk: a hand-off wait gave up (groups not co-resident?); the results of this call are invalid -- "
                                     "mst_set_trunk_groups(e, 0) returns to one launch per kernel");
    return rc;
}
extern "C" int mst_trunk_check(mst_engine* e) {          // after a synchronisation: did every hand-off of the resident-group launches arrive?
    if (!e) return fail("mst_trunk_check: null engine");
    if (e->trunk_err[0]) return fail("resident-group trunk: a hand-off wait gave up");
    return 0;
}

// K3 .. K8: token stream through the encoder stack.  rows = clips through the transformer.
// K1-K3: conditioning token + pose embedding of the frames -> token stream rows (ws.hx / ws.hl)
struct LoopRef { const LoopDev* ld = nullptr; int joff = 0; unsigned long long eo = 0; bool frames_ready = false; bool stream_ready = false; };   // stream_ready: the previous step's k_embed_out already embedded this step (token stream and conditioning tokens are in ws.hx / ws.hl)   // frames_ready: the previous step's epilogue already wrote ws.xt   // loop mode of a step's kernels (see LoopDev)

template <int KS>
static int launch_embed_in_n(mst_engine* e, const WS& ws, int tot, const DEpiEmbedIn& epi, hipStream_t st) {
    CHECK(ensure_dyn_lds((const void*)k_embed_in<KS>, EmbCfg::SMEM));
    hipLaunchKernelGGL(k_embed_in<KS>, dim3((tot + EmbCfg::BT - 1) / EmbCfg::BT), dim3(512), EmbCfg::SMEM, st, ws.xt, ws.xt_lo, e->kin_pad,
                       e->w_pose_in_pk, epi);
    HIPCHECK(hipGetLastError());
    return 0;
}
static int launch_embed_in(mst_engine* e, const WS& ws, int tot, const DEpiEmbedIn& epi, hipStream_t st) {
    switch (e->kin_pad / 32) {
#define EI(K_) case K_: return launch_embed_in_n<K_>(e, ws, tot, epi, st);
        EI(3) EI(4) EI(5) EI(6) EI(7) EI(8) EI(9) EI(10) EI(11) EI(12) EI(13) EI(14) EI(15) EI(16)
#undef EI
    }
    return fail("pose embedding: %d input columns unsupported", e->kin_pad);
}

// the pose embedding's epilogue of a model call / loop step: positional rows, stream rows, conditioning tokens
static DEpiEmbedIn embed_in_epi(const mst_engine* e, const WS& ws, int clips_x, int rows, int T, int temb_uniform_row, int temb_mod, int tp_uncond,
                                const LoopDev* ld, int joff, const long long* tidx_table = nullptr, const Drop* pe_dropout = nullptr) {
    const int S = T + 1, tot = clips_x * T;
    DEpiEmbedIn epi{e->b_pose_in, e->pe, ws.hx, ws.hl, T, S, tot, rows > clips_x ? (size_t)clips_x * S * MST_D : 0};
    epi.ct.temb = e->temb; epi.ct.textproj = ws.textproj; epi.ct.ld = ld;
    epi.ct.uniform_row = temb_uniform_row; epi.ct.temb_mod = temb_mod; epi.ct.joff = joff; epi.ct.rows = rows;
    epi.ct.tp_half = rows > clips_x ? clips_x : 0; epi.ct.tp_uncond = rows > clips_x ? tp_uncond : 0;
    if (tidx_table) { epi.ct.temb = e->temb_table; epi.ct.tidx = tidx_table; }
    if (pe_dropout) epi.pd = *pe_dropout;
    return epi;
}

static int assemble_stream(mst_engine* e, const WS& ws, const float* x, int clips_x, int rows, int T, int temb_uniform_row, int temb_mod,
                           hipStream_t st, int tp_uncond, LoopRef lr = LoopRef(), const long long* tidx_table = nullptr, const Drop* pe_dropout = nullptr) {
    if (lr.stream_ready) return 0;
    const int S = T + 1;
    (void)S;
    // (the conditioning token of every clip -- timestep embedding + text projection + positional row 0, formerly a launch of its
    // own, k_cond_token -- is written by the pose-embedding GEMM's epilogue: CondTok in mst_gemm_dma.h)
    {
        // frames -> f16 rows (transpose + convert), then the pose-embedding GEMM on the DMA ring.  The doubled
        // CFG batch feeds the same x to both halves: embed once, store twice (dup).
        ProfScope ps(e, FAM_EMBED_IN, st);
        const int F = e->cfg.feats, tot = clips_x * T;
        if (!lr.frames_ready) {
            hipLaunchKernelGGL(k_frames_f16, dim3((T + 31) / 32, e->kin_pad / 32, clips_x), dim3(256), 0, st, x, F, T, e->kin_pad, ws.xt, (const float*)nullptr, lr.ld, lr.eo, ws.xt_lo);
            HIPCHECK(hipGetLastError());
        }
        const DEpiEmbedIn epi = embed_in_epi(e, ws, clips_x, rows, T, temb_uniform_row, temb_mod, tp_uncond, lr.ld, lr.joff, tidx_table, pe_dropout);
        if (e->embed_fast && !e->precise) return launch_embed_in(e, ws, tot, epi, st);
        // x_t as hi + lo (RowsDirect::Xlo): both halves of a k-slab beside ONE copy of the weight slab
        CHECK((launch_gemm_dma<64, 512, 2, 2, 4, 1, 32, 2>(dim3((tot + 63) / 64, 1), RowsDirect{ws.xt, e->kin_pad, ws.xt_lo, e->precise ? e->w_pose_in_lo : nullptr},
                                                          e->w_pose_in, e->kin_pad, e->kin_pad, epi, st)));
    }
    return 0;
}

// K3 .. K8: token stream through the encoder stack.  rows = clips through the transformer.
static int run_trunk(mst_engine* e, const WS& ws, const float* x, int clips_x, int rows, int T, int temb_uniform_row, int temb_mod,
                     hipStream_t st, int tp_uncond = 0, LoopRef lr = LoopRef()) {
    const int S = T + 1, M = rows * S;
    CHECK(assemble_stream(e, ws, x, clips_x, rows, T, temb_uniform_row, temb_mod, st, tp_uncond, lr));
    if (e->dbg_stage == 0) return 0;
#define DBG_STOP(stage) if (e->dbg_layer == l && e->dbg_stage == stage) return 0;
    const bool small = e->precise || (e->small_m > 0 && M <= e->small_m);
    // Clips of at most 16 frames: so few values are averaged per output that the f16 rounding of the ACTIVATION operands shows at the
    // 1e-3 bar (oracle rounding model, classifier-free guidance: 1.07e-3 mean over seeds at 1 frame, 9.1e-4 at 5 frames, 7.6e-4 at
    // 196).  Those launches -- a handful of tiles, nowhere near a throughput regime -- multiply every activation as hi + lo
    // (RowsDirect::Xlo; the lo halves of att and hid borrow the idle hid / qkv buffers): 6.5e-4 mean in the same model.
    const bool precise = e->precise || (small && T <= 16);
    f16* const att_lo = precise ? ws.hid : nullptr;
    f16* const hid_lo = precise ? ws.qkv : nullptr;
    const f16* const hl_in = precise ? ws.hl : nullptr;
    // round 4: without split operands the four GEMMs run as resident-tile / streamed-weight kernels (mst_small.h; MST_SMALL_FAST=0: the ring)
    const bool fast = small && !precise && e->small_fast;
    // ... and the LayerNorms inside the GEMM behind them (LnRows): LN1 in FFN1, which leaves the stream in (hx2, hl2); LN2 in the next
    // layer's QKV GEMM, which brings it back to (hx, hl); the last LN2 as the rows kernel (MST_SMALL_LN=0: every LayerNorm a launch)
    const bool lnf = fast && e->small_ln && e->dbg_stage < 0 && M <= e->small_ln_m;
    const int NL = e->cfg.num_layers;
    for (int l = 0; small && l < NL; l++) {
        const LayerW& w = e->L[l];
        {
            ProfScope ps(e, FAM_QKV, st);
            if (lnf && l > 0) {
                const LayerW& p = e->L[l - 1];
                const LnRows ln{ws.zacc, p.b2, p.g2, p.be2, ws.hx2, ws.hl2, ws.hx, ws.hl};
                CHECK((launch_rows_gemm<16, 0>(M, 3 * MST_D, nullptr, w.wsm_in, w.b_in, ws.qkv, 3 * MST_D, st, &ln)));
            } else if (fast) CHECK((launch_rows_gemm<16, 0>(M, 3 * MST_D, ws.hx, w.wsm_in, w.b_in, ws.qkv, 3 * MST_D, st)));
            else {
                DEpiBiasF16<false> epi{w.b_in, ws.qkv, 3 * MST_D, M};
                CHECK(launch_small(M, 3 * MST_D, RowsDirect{ws.hx, MST_D, hl_in, e->precise ? w.w_in_lo : nullptr}, w.w_in, MST_D, MST_D, epi, st));
            }
        }
        DBG_STOP(1)
        {
            ProfScope ps(e, FAM_ATTN, st);
            CHECK(launch_attn(ws.qkv, ws.att, S, rows, st, 1, att_lo));
        }
        DBG_STOP(2)
        {
            ProfScope ps(e, FAM_OUTPROJ_LN, st);
            if (fast) CHECK((launch_rows_gemm<16, 2>(M, MST_D, ws.att, w.wsm_out, nullptr, ws.zacc, MST_D, st)));
            else {
                DEpiPlainF32 epi{ws.zacc, MST_D, M};
                CHECK(launch_small(M, MST_D, RowsDirect{ws.att, MST_D, att_lo, e->precise ? w.w_out_lo : nullptr}, w.w_out, MST_D, MST_D, epi, st));
            }
            if (!lnf) hipLaunchKernelGGL(k_ln_rows, dim3((M + 3) / 4), dim3(256), 0, st, ws.zacc, w.b_out, w.g1, w.be1, ws.hx, ws.hl, M, (f16*)nullptr, (f16*)nullptr);
            HIPCHECK(hipGetLastError());
        }
        DBG_STOP(3)
        {
            ProfScope ps(e, FAM_FFN1, st);
            if (lnf) {
                const LnRows ln{ws.zacc, w.b_out, w.g1, w.be1, ws.hx, ws.hl, ws.hx2, ws.hl2};
                CHECK((launch_rows_gemm<16, 1>(M, MST_FF, nullptr, w.wsm_1, w.b1, ws.hid, MST_FF, st, &ln)));
            } else if (fast) CHECK((launch_rows_gemm<16, 1>(M, MST_FF, ws.hx, w.wsm_1, w.b1, ws.hid, MST_FF, st)));
            else {
                DEpiBiasF16<true> epi{w.b1, ws.hid, MST_FF, M, hid_lo};
                CHECK(launch_small(M, MST_FF, RowsDirect{ws.hx, MST_D, hl_in, e->precise ? w.w1_lo : nullptr}, w.w1, MST_D, MST_D, epi, st));
            }
        }
        DBG_STOP(4)
        {
            ProfScope ps(e, FAM_FFN2_LN, st);
            if (fast) CHECK((launch_rows_gemm<32, 2>(M, MST_D, ws.hid, w.wsm_2, nullptr, ws.zacc, MST_D, st)));
            else {
                DEpiPlainF32 epi{ws.zacc, MST_D, M};
                CHECK(launch_small(M, MST_D, RowsDirect{ws.hid, MST_FF, hid_lo, e->precise ? w.w2_lo : nullptr}, w.w2, MST_FF, MST_FF, epi, st));
            }
            if (!lnf) hipLaunchKernelGGL(k_ln_rows, dim3((M + 3) / 4), dim3(256), 0, st, ws.zacc, w.b2, w.g2, w.be2, ws.hx, ws.hl, M, (f16*)nullptr, (f16*)nullptr);
            else if (l == NL - 1) hipLaunchKernelGGL(k_ln_rows, dim3((M + 3) / 4), dim3(256), 0, st, ws.zacc, w.b2, w.g2, w.be2, ws.hx2, ws.hl2, M, ws.hx, ws.hl);
            HIPCHECK(hipGetLastError());
        }
        DBG_STOP(5)
    }
    if (!small && !e->prof_now && trunk_groups_fit(e, S)) {
        CHECK(launch_trunk_groups(e, ws, S, rows, st));
        return 0;
    }
    for (int l = 0; !small && l < e->cfg.num_layers; l++) {
        const LayerW& w = e->L[l];
        if (e->fuse_qkv_attn && !(e->dbg_layer == l && e->dbg_stage == 1)) {
            ProfScope ps(e, FAM_QKV_ATTN, st);
            // 1 (default): weights streamed to registers, tokens through the ring; 2: round 2's kernel (both operands through the ring),
            // which also takes S = 209..224
            if (e->fuse_qkv_attn == 1 && qkv_attn2_fits(S)) CHECK(launch_qkv_attn2(ws.hx, w.wqkv, w.b_in, ws.att, S, rows, st));
            else CHECK(launch_qkv_attn(ws.hx, w.w_in, w.b_in, ws.att, S, rows, st));
        } else {
            {
                ProfScope ps(e, FAM_QKV, st);
                DEpiBiasF16<false> epi{w.b_in, ws.qkv, 3 * MST_D, M};
                CHECK((launch_wide(M, 3 * MST_D / 256, RowsDirect{ws.hx, MST_D}, w.w_in, MST_D, MST_D, epi, st)));
            }
            DBG_STOP(1)
            {
                ProfScope ps(e, FAM_ATTN, st);
                CHECK(launch_attn(ws.qkv, ws.att, S, rows, st));
            }
        }
        DBG_STOP(2)
        if (e->fuse_tail && !(e->dbg_layer == l && (e->dbg_stage == 3 || e->dbg_stage == 4))) {
            ProfScope ps(e, FAM_TAIL, st);
            CHECK(launch_tail(e, w, ws, M, st));
            DBG_STOP(5)
            continue;
        }
        {
            ProfScope ps(e, FAM_OUTPROJ_LN, st);
            DEpiResidLN epi{w.b_out, w.g1, w.be1, ws.hx, ws.hl, M};
            if (M >= e->ln128_min_m)
                CHECK((launch_gemm_dma<128, 512, 2, 4, 2, 1, 64>(dim3((M + 127) / 128, 1), RowsDirect{ws.att, MST_D}, w.w_out, MST_D, MST_D, epi, st)));
            else
                CHECK((launch_gemm_dma<64, 512, 2, 2, LN_NS, 1, LN_BK>(dim3((M + 63) / 64, 1), RowsDirect{ws.att, MST_D}, w.w_out, MST_D, MST_D, epi, st)));
        }
        DBG_STOP(3)
        {
            ProfScope ps(e, FAM_FFN1, st);
            DEpiBiasF16<true> epi{w.b1, ws.hid, MST_FF, M};
#if FFN1_BF == 512      // 128 x 512 tiles: 198 blocks at batch 64, one per CU, 640 KB staged per block
            const int nx = (M + 127) / 128;
            CHECK((launch_gemm_dma<128, 512, 2, 4, 3, 1>(dim3(((nx + 7) / 8) * 8 * 2, 1, 1), RowsDirect{ws.hx, MST_D}, w.w1, MST_D, MST_D, epi, st, 2)));
#else
            CHECK((launch_wide(M, MST_FF / 256, RowsDirect{ws.hx, MST_D}, w.w1, MST_D, MST_D, epi, st)));
#endif
        }
        DBG_STOP(4)
        {
            ProfScope ps(e, FAM_FFN2_LN, st);
            DEpiResidLN epi{w.b2, w.g2, w.be2, ws.hx, ws.hl, M};
            if (M >= e->ln128_min_m)
                CHECK((launch_gemm_dma<128, 512, 2, 4, 2, 1, 64>(dim3((M + 127) / 128, 1), RowsDirect{ws.hid, MST_FF}, w.w2, MST_FF, MST_FF, epi, st)));
            else
                CHECK((launch_gemm_dma<64, 512, 2, 2, LN_NS, 1, LN_BK>(dim3((M + 63) / 64, 1), RowsDirect{ws.hid, MST_FF}, w.w2, MST_FF, MST_FF, epi, st)));
        }
        DBG_STOP(5)
    }
#undef DBG_STOP
    return 0;
}

// output projection tiles: 64 frames x BF features.  8 waves as 1 x 8 with 2 x NTO MFMA tiles each (BF = 256 NTO), or -- for
// 257..384 features, the HumanML / bandai widths -- as 2 x 4 with 1 x 3 tiles (BF = 384): the 512-row tile streamed and multiplied
// 37 % zero-padding rows per slab.
template <int MODE, int BF, int MT, int NT, int NX, int XS = 1>
static int launch_out(mst_engine* e, const WS& ws, int batch, int T, float* out, const StepArgs& sa, hipStream_t st,
                      const f16* w_override = nullptr, const float* b_override = nullptr, int tok_off = 1, bool frames_next = false,
                      bool hi_lo = false) {
    const int S = T + tok_off;
    RowsFrames xs{ws.hx, MST_D, T, S, batch * T, 64, (size_t)batch * S, tok_off, hi_lo ? ws.hl : nullptr,   // hi_lo: ws.hx / ws.hl are the last stream
                  hi_lo && e->precise && !w_override ? e->w_pose_out_lo : nullptr};
    DEpiEmbedOut<MODE> epi{b_override ? b_override : e->b_pose_out, e->cfg.feats, T, batch * T, out, sa};
    if (frames_next) { epi.xt_next = ws.xt; epi.kpad = e->kin_pad; epi.xt_next_lo = ws.xt_lo; }
    return launch_gemm_dma<64, BF, MT, NT, 4, NX, 32, XS>(dim3((batch * T + 63) / 64, 1), xs, w_override ? w_override : e->w_pose_out, MST_D, MST_D, epi, st);
}
template <int MODE, int BF, int MT, int NT>
static int launch_out_nx(mst_engine* e, const WS& ws, int cfg, int batch, int T, float* out, const StepArgs& sa, hipStream_t st,
                         const f16* wo = nullptr, const float* bo = nullptr, int tok_off = 1, bool frames_next = false, bool hi_lo = false) {
    // hi + lo stream: both halves of a k-slab beside one copy of the weight slab where that ring fits the LDS (<= 384 rows),
    // otherwise the K range twice (gemm_mainloop_dma)
    if constexpr (BF <= 384) {
        if (hi_lo) return cfg ? launch_out<MODE, BF, MT, NT, 2, 2>(e, ws, batch, T, out, sa, st, wo, bo, tok_off, frames_next, true)
                              : launch_out<MODE, BF, MT, NT, 1, 2>(e, ws, batch, T, out, sa, st, wo, bo, tok_off, frames_next, true);
    }
    return cfg ? launch_out<MODE, BF, MT, NT, 2>(e, ws, batch, T, out, sa, st, wo, bo, tok_off, frames_next, hi_lo)
               : launch_out<MODE, BF, MT, NT, 1>(e, ws, batch, T, out, sa, st, wo, bo, tok_off, frames_next, hi_lo);
}
template <int MODE, int NBW, int NX>
static int launch_embed_out_n(mst_engine* e, const RowsFrames& xs, const DEpiEmbedOut<MODE>& epi, int tiles, hipStream_t st) {
    CHECK(ensure_dyn_lds((const void*)k_embed_out<NBW, MODE, NX>, EmbCfg::SMEM));
    hipLaunchKernelGGL((k_embed_out<NBW, MODE, NX>), dim3(tiles), dim3(512), EmbCfg::SMEM, st, xs, e->w_pose_out_pk, epi);
    HIPCHECK(hipGetLastError());
    return 0;
}
// Can a sampling step's output projection also embed the NEXT step (k_embed_out<.., KSN>)?  The shapes the reference's datasets have
// (150 / 181 / 190 / 263 features: 5 / 6 / 6 / 9 k-steps), and the tile plus the two frame images must fit the LDS.
static bool embed_next_fits(const mst_engine* e, int T) {
    const int ks = e->kin_pad / 32, nbw = (e->cfg.feats + 127) / 128, F = e->cfg.feats;
    if (!e->embed_fast || e->precise || (T & 3)) return false;
    if (!((ks == 5 && nbw == 2) || (ks == 6 && nbw == 2) || (ks == 9 && nbw == 3))) return false;
    return (size_t)F * (EmbCfg::BT + 4) * 4 + 16 + 2 * (size_t)EmbCfg::BT * (ks * 64 + 16) <= (size_t)EmbCfg::SMEM;
}
template <int MODE, int NBW, int KSN>
static int launch_embed_step_n(mst_engine* e, const RowsFrames& xs, const DEpiEmbedOut<MODE>& epi, const DEpiEmbedIn& next, int tiles, hipStream_t st) {
    CHECK(ensure_dyn_lds((const void*)k_embed_out<NBW, MODE, 1, KSN>, EmbCfg::SMEM));
    hipLaunchKernelGGL((k_embed_out<NBW, MODE, 1, KSN>), dim3(tiles), dim3(512), EmbCfg::SMEM, st, xs, e->w_pose_out_pk, epi, e->w_pose_in_pk, next);
    HIPCHECK(hipGetLastError());
    return 0;
}
template <int MODE>
static int launch_embed_out(mst_engine* e, const WS& ws, int cfg, int batch, int T, float* out, const StepArgs& sa, hipStream_t st,
                            int tok_off, bool frames_next, const DEpiEmbedIn* next = nullptr) {
    const int S = T + tok_off, tiles = (batch * T + EmbCfg::BT - 1) / EmbCfg::BT;
    RowsFrames xs{ws.hx, MST_D, T, S, batch * T, EmbCfg::BT, (size_t)batch * S, tok_off, ws.hl, nullptr};
    DEpiEmbedOut<MODE> epi{e->b_pose_out, e->cfg.feats, T, batch * T, out, sa};
    if (frames_next) { epi.xt_next = ws.xt; epi.kpad = e->kin_pad; epi.xt_next_lo = ws.xt_lo; }
    const int nbw = embed_out_nbw(e);
    if constexpr (MODE != 0) {
        if (next) {                                         // (mst_sample_loop asked embed_next_fits first)
            if (cfg || tok_off != 1) return fail("output projection: the fused next-step embedding is for plain sampling steps");
            epi.xt_next = ws.xt; epi.kpad = e->kin_pad;     // (marks the tile as reusable: OutItems::apply writes x_{t-1} back into it)
            switch (e->kin_pad / 32) {
                case 5: return launch_embed_step_n<MODE, 2, 5>(e, xs, epi, *next, tiles, st);
                case 6: return launch_embed_step_n<MODE, 2, 6>(e, xs, epi, *next, tiles, st);
                case 9: return launch_embed_step_n<MODE, 3, 9>(e, xs, epi, *next, tiles, st);
            }
            return fail("output projection: no fused instantiation for %d input columns", e->kin_pad);
        }
    }
#define EO(N_) case N_: return cfg ? launch_embed_out_n<MODE, N_, 2>(e, xs, epi, tiles, st) : launch_embed_out_n<MODE, N_, 1>(e, xs, epi, tiles, st);
    switch (nbw) { EO(1) EO(2) EO(3) case 4: return launch_embed_out_n<MODE, 4, 1>(e, xs, epi, tiles, st); }
#undef EO
    return fail("output projection: %d features unsupported", e->cfg.feats);
}
template <int MODE>
static int launch_out_nt(mst_engine* e, const WS& ws, int cfg, int batch, int T, float* out, const StepArgs& sa, hipStream_t st,
                         const f16* wo = nullptr, const float* bo = nullptr, int tok_off = 1, bool frames_next = false, bool hi_lo = false,
                         const DEpiEmbedIn* next = nullptr) {
    ProfScope ps(e, FAM_EMBED_OUT, st);
    if (e->embed_fast && !e->precise && hi_lo && !wo && !bo && !(cfg && embed_out_nbw(e) == 4))      // (385+ features under CFG: its two accumulator sets spill)
        return launch_embed_out<MODE>(e, ws, cfg, batch, T, out, sa, st, tok_off, frames_next, next);
    if (next) return fail("output projection: the fused next-step embedding needs the fast embed kernels");
    const int rows_out = e->cfg.feats;                   // (also for the pose embedding's backward, which runs this kernel with W_in^T)
    if (rows_out <= 256) return launch_out_nx<MODE, 256, 2, 1>(e, ws, cfg, batch, T, out, sa, st, wo, bo, tok_off, frames_next, hi_lo);
    if (rows_out <= 384) return launch_out_nx<MODE, 384, 1, 3>(e, ws, cfg, batch, T, out, sa, st, wo, bo, tok_off, frames_next, hi_lo);
    return launch_out_nx<MODE, 512, 2, 2>(e, ws, cfg, batch, T, out, sa, st, wo, bo, tok_off, frames_next, hi_lo);
}

static int check_ready(mst_engine* e, int batch, int frames, int cfg) {
    if (!e) return fail("null engine");
    CHECK(mst_weights_complete(e));
    if (frames < 1 || frames > e->cfg.max_frames) return fail("frames %d outside 1..%d", frames, e->cfg.max_frames);
    int rows = cfg ? 2 * batch : batch;
    if (batch < 1 || rows > e->cfg.max_rows) return fail("batch %d (rows %d) exceeds max_rows %d", batch, rows, e->cfg.max_rows);
    if (e->precise && !e->lo_missing.empty())
        return fail("precise mode: %zu weight matrices (first: %s) were uploaded while it was off, without their lo halves: upload them again",
                    e->lo_missing.size(), e->lo_missing.begin()->c_str());
    if (e->text_batch != batch || e->text_cfg != (cfg ? 1 : 0))
        return fail("mst_set_text was not called for batch %d cfg %d (have %d/%d)", batch, cfg, e->text_batch, e->text_cfg);
    return 0;
}

extern "C" int mst_forward(mst_engine* e, const float* x, const int64_t* t, const float* scale, int32_t batch, int32_t frames,
                           int32_t cfg, float* out, void* stream) {
    CHECK(check_ready(e, batch, frames, cfg));
    if (!x || !t || !out || (cfg && !scale)) return fail("mst_forward: null argument");
    hipStream_t st = (hipStream_t)stream;
    ON_DEVICE(e->cfg.device);
    e->prof_now = e->prof_on;
    e->cur_slices = 1;
    CHECK(ensure_packed(e, st));
    CHECK(timestep_rows(e, (const long long*)t, batch, st));
    const int rows = cfg ? 2 * batch : batch;
    const WS ws = ws_slice(e, 0, frames);
    CHECK(run_trunk(e, ws, x, batch, rows, frames, -1, batch, st, batch));
    StepArgs sa{};
    sa.scale = scale;
    return trunk_settle(e, st, launch_out_nt<0>(e, ws, cfg, batch, frames, out, sa, st, nullptr, nullptr, 1, false, true));
}

// How many independent clip slices a loop over `batch` clips of `frames` frames runs as.  Measured, same box, interleaved
// (tools/experiments/streams_ab.sh, tools/experiments/streams_ab_configs.sh), round-2 kernels at 196 frames: a batch whose tiles are all resident at
// once on the large-tile path wanted ONE slice (batch 64: 82.0 / 81.6 / 80.6 clips/s at 1 / 2 / 3 slices; batch 32: 43.8 vs 39.7
// at 3 -- slices would drop to the small-tile kernels); more tiles than CUs want one slice per round of tiles (batch 128 = 394
// tiles: 78.5 / 91.2 / 88.0 at 1 / 2 / 3; CFG at 64 clips: 39.8 / 45.5 / 44.6); the small-tile path (batch 16: 22.1 vs 28.8) up to three.
static int loop_slices_for(const mst_engine* e, int batch, int cfg, int frames) {
    if (!e || e->dbg_stage >= 0) return 1;
    const int rows = (cfg ? 2 : 1) * batch;
    int n = e->nsplit;
    if (n == 0 && trunk_groups_fit(e, frames + 1) && !(e->small_m > 0 && (long long)rows * (frames + 1) <= e->small_m)) return 1;   // every clip is a chain of its own inside ONE launch
    if (n == 0) {
        const long long M = (long long)rows * (frames + 1);
        const bool small = e->precise || (e->small_m > 0 && M <= e->small_m);
        const long long tiles = (M + 63) / 64, waves = (tiles + 255) / 256;      // rounds of 64-token tiles over the 256 CUs
        n = small ? 3 : (int)(waves < 3 ? waves : 3);
        // Round 3: a batch that fills most of the chip in ONE round (the headline: 64 clips = 197 tiles) runs every workgroup through the
        // same phase at the same time; three slices of it (each still on the large-tile path) decorrelate them.  Same-box, interleaved,
        // 1 vs 3 slices at 64 clips: 96.9 vs 98.9 clips/s on a slow box (three rounds), 103.2 vs 104.1 on a fast one; at 48 clips
        // (148 tiles) two slices are a wash (84.0 vs 83.7) and three fall to the small-tile kernels.
        if (!small && waves == 1 && tiles >= 192) n = 3;
    }
    while (n > 1 && rows / n < 8) n--;                       // at least 8 rows through the transformer per slice
    return n;
}
extern "C" int mst_loop_slices(const mst_engine* e, int32_t batch, int32_t cfg, int32_t frames) {
    return loop_slices_for(e, batch, cfg, frames > 0 ? frames : (e ? e->cfg.max_frames : 0));
}

// One denoise step of every slice, enqueued (or captured): slice sl on streams[sl]; step = *ld.jbase + joff.
struct LoopPlan {
    const mst_schedule* s; const mst_loop_args* a; int nsl; size_t per_clip, clip_elems;
    hipStream_t streams[mst_engine::MAX_SLICES];
};
// frames_ready / frames_next: the step's frame rows (ws.xt) were written by the previous step's epilogue / this step's epilogue
// writes them for the next one (mst_sample_loop decides; see DEpiEmbedOut::xt_next)
// stream_ready / embed_next: the previous step's output projection embedded this step / this step's embeds the next (k_embed_out<.., KSN>)
static int enqueue_step(mst_engine* e, const LoopPlan& p, int joff, int nsj, bool frames_ready = false, bool frames_next = false,
                        bool stream_ready = false, bool embed_next = false) {
    const mst_loop_args* a = p.a;
    e->cur_slices = nsj;
    for (int sl = 0; sl < nsj; sl++) {
        const int per = (a->batch + nsj - 1) / nsj;          // clips per slice (last one may be short)
        const int c0 = sl * per;
        const int nb = (c0 + per <= a->batch) ? per : a->batch - c0;
        if (nb <= 0) continue;
        const size_t eo = (size_t)c0 * p.per_clip;
        // workspace rows of the slice: its clips (cond + uncond twins under CFG); text projections are indexed
        // in the full-batch layout [cond 0..B | uncond 0..B]
        WS ws = ws_slice(e, a->cfg ? 2 * c0 : c0, a->frames);
        ws.textproj = e->textproj + (size_t)c0 * MST_D;
        hipStream_t ss = p.streams[sl];
        LoopRef lr{e->ld_dev, joff, eo, frames_ready, stream_ready};
        CHECK(run_trunk(e, ws, nullptr, nb, a->cfg ? 2 * nb : nb, a->frames, 0, 0, ss, a->batch, lr));
        const DEpiEmbedIn next_epi = embed_in_epi(e, ws, nb, nb, a->frames, 0, 0, a->batch, e->ld_dev, joff + 1);
        const DEpiEmbedIn* next = embed_next ? &next_epi : nullptr;
        StepArgs sa{};
        sa.tab = p.s->tab;
        sa.nsteps = p.s->n;
        // presence flags: the kernel resolves the pointers from *ld (step_resolve)
        const float* const yes = reinterpret_cast<const float*>(1);
        sa.mask = a->inpainting_mask_dev ? yes : nullptr;
        sa.motion = a->inpainted_motion_dev ? yes : nullptr;
        sa.noise = a->noise_mode == MST_NOISE_BUFFER ? yes : nullptr;
        sa.scale = a->scale_dev ? yes : nullptr;
        sa.xstart = a->xstart_dump_dev ? reinterpret_cast<float*>(1) : nullptr;
        sa.clip0 = (unsigned)c0;
        sa.mask_noise = a->mask_noise;
        sa.clip = a->clip_denoised;
        sa.philox = a->noise_mode == MST_NOISE_PHILOX;
        sa.ld = e->ld_dev;
        sa.joff = joff;
        sa.eo = eo;
        sa.step_stride = p.clip_elems;
        sa.rowflag = (a->inpainting_mask_dev && a->inpainted_motion_dev) ? e->rowflag + (size_t)c0 * e->cfg.feats : nullptr;
        if (a->sampler == MST_SAMPLER_DDPM) CHECK(launch_out_nt<1>(e, ws, a->cfg, nb, a->frames, nullptr, sa, ss, nullptr, nullptr, 1, frames_next, true, next));
        else CHECK(launch_out_nt<2>(e, ws, a->cfg, nb, a->frames, nullptr, sa, ss, nullptr, nullptr, 1, frames_next, true, next));
    }
    return 0;
}
static int fork_slices(mst_engine* e, const LoopPlan& p) {
    HIPCHECK(hipEventRecord(e->ev_fork, p.streams[0]));
    for (int i = 1; i < p.nsl; i++) HIPCHECK(hipStreamWaitEvent(p.streams[i], e->ev_fork, 0));
    return 0;
}
static int join_slices(mst_engine* e, const LoopPlan& p) {
    for (int i = 1; i < p.nsl; i++) {
        HIPCHECK(hipEventRecord(e->ev_join[i - 1], p.streams[i]));
        HIPCHECK(hipStreamWaitEvent(p.streams[0], e->ev_join[i - 1], 0));
    }
    return 0;
}

// The reference's per-step Python loop (gaussian_diffusion.py:775-794) as ONE call.  Steps are either enqueued from the host
// (short loops, instrumented runs) or replayed from a captured hipGraph of `graph_steps` steps: every kernel reads its
// tensors and its step index through the LoopDev block in device memory, which a one-thread kernel at the end of the graph
// advances, so the instantiated graph is reused by every replay and by every later call with the same shapes.
extern "C" int mst_sample_loop(mst_engine* e, const mst_schedule* s, const mst_loop_args* a, void* stream) {
    if (!s || !a) return fail("mst_sample_loop: null argument");
    CHECK(check_ready(e, a->batch, a->frames, a->cfg));
    if (a->t_start >= s->n || a->t_end < 0 || a->t_start < a->t_end)
        return fail("mst_sample_loop: bad index range %d..%d for %d steps", a->t_start, a->t_end, s->n);
    if (!a->x_dev || (a->cfg && !a->scale_dev)) return fail("mst_sample_loop: null x / scale");
    if (a->noise_mode == MST_NOISE_BUFFER && !a->noise_dev) return fail("mst_sample_loop: noise buffer missing");
    if (a->sampler != MST_SAMPLER_DDPM && a->sampler != MST_SAMPLER_DDIM) return fail("mst_sample_loop: bad sampler");
    hipStream_t caller = (hipStream_t)stream;
    ON_DEVICE(e->cfg.device);
    // The loop runs on engine-owned streams: behind everything the caller has enqueued (ev_in), and the caller's stream
    // continues behind the loop (ev_out).  A captured graph needs that: torch's current stream is normally the legacy
    // default stream, which cannot be captured.
    hipStream_t st = e->loop_stream;
    HIPCHECK(hipEventRecord(e->ev_in, caller));
    HIPCHECK(hipStreamWaitEvent(st, e->ev_in, 0));
    const int nrun = a->t_start - a->t_end + 1;
    // K1 hoisted: the timestep MLP for every index this loop visits (row j <-> index t_end + j)
    e->prof_now = 0;
    CHECK(ensure_packed(e, st));
    CHECK(timestep_rows(e, s->tmap + a->t_end, nrun, st));
    // Clips are independent, so the batch runs as `nsplit` slices on separate streams: one slice's kernels fill
    // the CUs the other leaves idle in its prologues, tails and launch gaps (per-launch time is per-CU bound and
    // flat in the block count at this size).  CFG batches are sliced the same way (cond + uncond twins stay together).
    LoopPlan p{s, a, loop_slices_for(e, a->batch, a->cfg, a->frames), (size_t)e->cfg.feats * a->frames, (size_t)a->batch * e->cfg.feats * a->frames,
               {st, e->aux_stream[0], e->aux_stream[1], e->aux_stream[2], e->aux_stream[3], e->aux_stream[4], e->aux_stream[5], e->aux_stream[6]}};
    // per-call arguments -> device (pinned staging slot; the slot's previous upload has long completed when it comes round again)
    {
        const int slot = e->ld_next;
        e->ld_next = (slot + 1) % mst_engine::LD_SLOTS;
        HIPCHECK(hipEventSynchronize(e->ld_ev[slot]));
        LoopDev& h = e->ld_pin[slot];
        h = LoopDev{a->x_dev, a->inpainting_mask_dev, a->inpainted_motion_dev, a->noise_dev, a->scale_dev, a->xstart_dump_dev,
                    a->seed, a->eta, a->t_start, nrun, 0, 0};
        HIPCHECK(hipMemcpyAsync(e->ld_dev, &h, sizeof(LoopDev), hipMemcpyHostToDevice, st));
        HIPCHECK(hipEventRecord(e->ld_ev[slot], st));
    }
    if (a->inpainting_mask_dev && a->inpainted_motion_dev) {      // once per loop: which mask rows the step kernel may skip
        const int rows_m = a->batch * e->cfg.feats;
        hipLaunchKernelGGL(k_mask_rowflags, dim3((rows_m + 3) / 4), dim3(256), 0, st, a->inpainting_mask_dev, rows_m, a->frames, e->rowflag);
        HIPCHECK(hipGetLastError());
    }
    // what a captured graph depends on (everything else reaches the kernels through LoopDev)
    const int U = e->graph_steps;
    std::vector<long long> key = {a->batch, a->frames, a->cfg, a->sampler, a->noise_mode, a->mask_noise, a->clip_denoised,
                                  a->inpainting_mask_dev != nullptr, a->inpainted_motion_dev != nullptr, a->xstart_dump_dev != nullptr,
                                  a->scale_dev != nullptr, p.nsl, U, (long long)(size_t)s->tab, s->n, e->small_m, e->fuse_tail,
                                  e->fuse_qkv_attn, e->ln128_min_m, e->precise, e->tail_ntb, e->embed_fast, e->small_fast, e->small_ln};      // every switch run_trunk / loop_slices_for branch on
    const bool use_graph = e->graph_on && !e->prof_on && e->dbg_stage < 0 && nrun >= 2 * U;
    bool forked = false;
    auto steps = [&]() -> int {
        int j = 0;
        if (use_graph) {
            // head of the loop from the host: the remainder, plus one graph's worth of steps the first time a configuration is
            // seen (every kernel's per-device LDS opt-in must have happened before a capture)
            int pre = nrun % U;
            if (e->warm_key != key) pre += U;
            if (p.nsl > 1) { CHECK(fork_slices(e, p)); forked = true; }
            for (; j < pre; j++) CHECK(enqueue_step(e, p, j, p.nsl));
            if (forked) { CHECK(join_slices(e, p)); forked = false; }
            e->warm_key = key;
            if (pre > 0) {
                hipLaunchKernelGGL(k_loop_advance, dim3(1), dim3(1), 0, st, e->ld_dev, pre);
                HIPCHECK(hipGetLastError());
            }
            if (e->gkey != key || !e->gexec) {
                if (e->gexec) { (void)hipGraphExecDestroy(e->gexec); e->gexec = nullptr; }
                e->gkey.clear();
                HIPCHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
                int rc = 0;
                if (p.nsl > 1) rc = fork_slices(e, p);
                for (int k = 0; k < U && !rc; k++) rc = enqueue_step(e, p, k, p.nsl);
                if (!rc && p.nsl > 1) rc = join_slices(e, p);
                if (!rc) {
                    hipLaunchKernelGGL(k_loop_advance, dim3(1), dim3(1), 0, st, e->ld_dev, U);
                    if (hipGetLastError() != hipSuccess) rc = fail("mst_sample_loop: capture of the step-counter kernel failed");
                }
                hipGraph_t graph = nullptr;
                const hipError_t ce = hipStreamEndCapture(st, &graph);          // always end the capture: the stream must come back
                if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
                if (ce != hipSuccess || !graph) return fail("mst_sample_loop: stream capture failed: %s", hipGetErrorString(ce));
                const hipError_t ie = hipGraphInstantiate(&e->gexec, graph, nullptr, nullptr, 0);
                (void)hipGraphDestroy(graph);
                if (ie != hipSuccess) { e->gexec = nullptr; return fail("mst_sample_loop: hipGraphInstantiate failed: %s", hipGetErrorString(ie)); }
                e->gkey = key;
            }
            for (; j < nrun; j += U) HIPCHECK(hipGraphLaunch(e->gexec, st));
            return 0;
        }
        // Host-enqueued steps chain their frame rows: step j's epilogue writes the f16 rows step j + 1 embeds, so only step 0 runs
        // the transpose kernel.  Not under CFG (there a slice's rows sit at 2 x its first clip, which changes when an instrumented
        // step runs as one slice), not for frame counts the epilogue walks element-wise, not in captured graphs (their first
        // step would have to differ from call to call).
        const bool chain = e->fuse_frames && !a->cfg && (a->frames & 3) == 0 && e->dbg_stage < 0;
        // ... and where the shapes allow, step j's output projection embeds step j + 1 in the same launch (MST_FUSE_EMBED=0: frame rows
        // only).  Instrumented steps and their neighbours keep the two kernels apart, so that the event-timed families stay what they say.
        const bool fuse_embed = chain && e->fuse_embed && embed_next_fits(e, a->frames);
        auto instrumented = [&](int jj) { return e->prof_on && (jj % e->prof_period == 0); };
        bool stream_ready = false;
        for (; j < nrun; j++) {
            e->prof_now = e->prof_on && (j % e->prof_period == 0);
            // instrumented steps run as ONE full-batch slice so the HIP-event durations are those of isolated
            // full-batch launches (the roofline figures); all other steps use the concurrent slices
            const int nsj = e->prof_now ? 1 : p.nsl;
            if (nsj > 1 && !forked) { CHECK(fork_slices(e, p)); forked = true; }
            else if (nsj == 1 && forked) { CHECK(join_slices(e, p)); forked = false; }
            const bool embed_next = fuse_embed && j + 1 < nrun && !instrumented(j) && !instrumented(j + 1);
            CHECK(enqueue_step(e, p, j, nsj, chain && j > 0 && !stream_ready, chain && j + 1 < nrun && !embed_next, stream_ready, embed_next));
            stream_ready = embed_next;
        }
        return 0;
    };
    const int rc = steps();
    // join the slice streams whatever happened: after an error mid-loop the slices' work is still in flight on the
    // caller's tensors, so the caller's stream must not run ahead of it (best effort: errors here are not reported twice)
    if (forked) {
        for (int i = 1; i < p.nsl; i++) {
            if (hipEventRecord(e->ev_join[i - 1], p.streams[i]) == hipSuccess) (void)hipStreamWaitEvent(st, e->ev_join[i - 1], 0);
        }
    }
    e->prof_now = 0;
    if (hipEventRecord(e->ev_out, st) == hipSuccess) (void)hipStreamWaitEvent(caller, e->ev_out, 0);
    return trunk_settle(e, st, rc);
}

// ------------------------------------------------------------------------------------------ elementwise ABI
extern "C" int mst_q_sample(const mst_schedule* s, const float* x0, const float* noise, const float* mask, const int64_t* t,
                            int32_t batch, int64_t per_clip, float* out, void* stream) {
    if (!s || !x0 || !noise || !t || !out || batch < 1 || per_clip < 1) return fail("mst_q_sample: bad arguments");
    ON_DEVICE(s->device);
    int gx = (int)((per_clip + 255) / 256);
    if (gx > 2048) gx = 2048;
    hipLaunchKernelGGL(k_q_sample, dim3(gx, batch), dim3(256), 0, (hipStream_t)stream, s->tab, s->n, x0, noise, mask,
                       (const long long*)t, (long long)per_clip, out);
    HIPCHECK(hipGetLastError());
    return 0;
}

extern "C" int mst_step_epilogue_mt(const mst_schedule* s, const float* model_out, const float* x, const float* noise,
                                    const float* mask, const float* motion, const int64_t* t, int32_t batch, int64_t per_clip,
                                    int32_t sampler, int32_t mean_type, float eta, int32_t mask_noise, int32_t clip_denoised, float* sample,
                                    float* xstart, void* stream) {
    if (!s || !model_out || !x || !t || batch < 1 || per_clip < 1) return fail("mst_step_epilogue: bad arguments");
    if (sampler != MST_SAMPLER_DDPM && sampler != MST_SAMPLER_DDIM) return fail("mst_step_epilogue: bad sampler %d", sampler);
    if (mean_type < 0 || mean_type > 2) return fail("mst_step_epilogue: bad mean type %d (0 = x_start, 1 = epsilon, 2 = previous x)", mean_type);
    ON_DEVICE(s->device);
    int gx = (int)((per_clip + 255) / 256);
    if (gx > 2048) gx = 2048;
#define STEP_LAUNCH(S_, M_)                                                                                                              \
    hipLaunchKernelGGL((k_step_epilogue<S_, M_>), dim3(gx, batch), dim3(256), 0, (hipStream_t)stream, s->tab, s->n, eta, model_out, x,   \
                       noise, mask, motion, (const long long*)t, (long long)per_clip, mask_noise, clip_denoised, sample, xstart)
    const bool ddim = sampler == MST_SAMPLER_DDIM;
    switch (mean_type) {
        case 0: if (ddim) STEP_LAUNCH(1, 0); else STEP_LAUNCH(0, 0); break;
        case 1: if (ddim) STEP_LAUNCH(1, 1); else STEP_LAUNCH(0, 1); break;
        default: if (ddim) STEP_LAUNCH(1, 2); else STEP_LAUNCH(0, 2); break;
    }
#undef STEP_LAUNCH
    HIPCHECK(hipGetLastError());
    return 0;
}
extern "C" int mst_step_epilogue(const mst_schedule* s, const float* model_out, const float* x, const float* noise,
                                 const float* mask, const float* motion, const int64_t* t, int32_t batch, int64_t per_clip,
                                 int32_t sampler, float eta, int32_t mask_noise, int32_t clip_denoised, float* sample,
                                 float* xstart, void* stream) {
    return mst_step_epilogue_mt(s, model_out, x, noise, mask, motion, t, batch, per_clip, sampler, 0, eta, mask_noise, clip_denoised, sample,
                                xstart, stream);
}

extern "C" int mst_step_backward(const mst_schedule* s, const float* g_sample, const float* g_pred, const float* mask,
                                 int32_t has_blend, const int64_t* t, int32_t batch, int64_t per_clip, int32_t sampler, float eta,
                                 const float* pred_clipped, float* d_model_out, void* stream) {
    if (!s || !t || !d_model_out || batch < 1 || per_clip < 1 || (!g_sample && !g_pred) || (has_blend && !mask))
        return fail("mst_step_backward: bad arguments");
    ON_DEVICE(s->device);
    int gx = (int)((per_clip + 255) / 256);
    if (gx > 2048) gx = 2048;
    if (sampler == MST_SAMPLER_DDPM)
        hipLaunchKernelGGL(k_step_backward<0>, dim3(gx, batch), dim3(256), 0, (hipStream_t)stream, s->tab, s->n, eta, g_sample, g_pred, mask,
                           has_blend, (const long long*)t, (long long)per_clip, pred_clipped, d_model_out);
    else if (sampler == MST_SAMPLER_DDIM)
        hipLaunchKernelGGL(k_step_backward<1>, dim3(gx, batch), dim3(256), 0, (hipStream_t)stream, s->tab, s->n, eta, g_sample, g_pred, mask,
                           has_blend, (const long long*)t, (long long)per_clip, pred_clipped, d_model_out);
    else
        return fail("mst_step_backward: bad sampler %d", sampler);
    HIPCHECK(hipGetLastError());
    return 0;
}

extern "C" int mst_masked_l2(const float* a, int64_t a_stride, const float* b, const float* mask, int64_t mask_stride, int32_t n,
                             int32_t feats, int32_t frames, const float* g, float* out, void* stream) {
    if (!a || !b || !mask || !out || n < 1 || feats < 1 || frames < 1) return fail("mst_masked_l2: bad arguments");
    if (!g)
        hipLaunchKernelGGL(k_masked_l2_fwd, dim3(n), dim3(1024), 0, (hipStream_t)stream, a, (long long)a_stride, b, mask, (long long)mask_stride,
                           feats, frames, out);
    else {
        int gx = (feats * frames + 255) / 256;
        if (gx > 64) gx = 64;
        hipLaunchKernelGGL(k_masked_l2_bwd, dim3(gx, n), dim3(256), 0, (hipStream_t)stream, a, (long long)a_stride, b, mask,
                           (long long)mask_stride, feats, frames, g, out);
    }
    HIPCHECK(hipGetLastError());
    return 0;
}

extern "C" int mst_text_cosine(const float* f, const float* m, int32_t batch, int32_t dim, const float* g, float* out, void* stream) {
    if (!f || !m || !out || batch < 1 || batch > 1024 || dim < 1) return fail("mst_text_cosine: bad arguments (batch 1..1024)");
    hipLaunchKernelGGL(k_text_cosine, dim3(1), dim3(1024), 0, (hipStream_t)stream, f, m, batch, dim, g ? 1 : 0, g, out);
    HIPCHECK(hipGetLastError());
    return 0;
}

extern "C" int mst_philox_normal(float* out, int32_t batch, int32_t feats, int32_t frames, uint64_t seed, uint32_t step,
                                 void* stream) {
    if (!out || batch < 1 || feats < 1 || frames < 1) return fail("mst_philox_normal: bad arguments");
    int n = feats * ((frames + 3) / 4);
    hipLaunchKernelGGL(k_philox_normal, dim3((n + 255) / 256, batch), dim3(256), 0, (hipStream_t)stream, out, feats, frames,
                       (unsigned long long)seed, (unsigned)step);
    HIPCHECK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------ training ABI
// The trainable encoder stack (seqTransEncoder, mdm_forstyledataset.py:539-546) in training mode.
// Tape: per layer the activations the backward pass needs, in caller-owned memory.
struct TapeL { f16 *qkv, *att, *z1h, *z1l, *x1h, *x1l, *pre, *hid, *z2h, *z2l; float* lse; };
struct Tape { f16* sh[17]; f16* sl[17]; TapeL L[16]; };
static size_t tape_rows(int M) { return ((size_t)(M + 255) / 256) * 256 + 256; }     // whole tiles + one spare (over-read)
static size_t tape_layout(char* base, int nl, size_t Mp, Tape* t) {
    size_t off = 0;
    auto take = [&](size_t width) { f16* p = reinterpret_cast<f16*>(base + off); off += Mp * width * 2; return p; };
    for (int l = 0; l <= nl; l++) { t->sh[l] = take(MST_D); t->sl[l] = take(MST_D); }
    for (int l = 0; l < nl; l++) {
        TapeL& a = t->L[l];
        a.qkv = take(3 * MST_D); a.att = take(MST_D);
        a.z1h = take(MST_D); a.z1l = take(MST_D); a.x1h = take(MST_D); a.x1l = take(MST_D);
        a.pre = take(MST_FF); a.hid = take(MST_FF);
        a.z2h = take(MST_D); a.z2l = take(MST_D);
        a.lse = reinterpret_cast<float*>(take(2 * MST_H));            // softmax row statistics [rows][4][S] (4 floats per token)
    }
    return off;
}

// elem_off: the site's element counter starts there (drop_mul hashes idx * phi + key, so an offset is a shifted key): a forward pass that
// writes clips [clip0, ..) of a LARGER tape draws exactly the masks a backward pass over the whole tape regenerates
// the rows of clips [clip0, ..) inside a tape laid out for more clips (every slot is [rows][width], clip-major)
static void tape_shift(Tape& t, int nl, size_t row0) {
    for (int l = 0; l <= nl; l++) { t.sh[l] += row0 * MST_D; t.sl[l] += row0 * MST_D; }
    for (int l = 0; l < nl; l++) {
        TapeL& a = t.L[l];
        a.qkv += row0 * 3 * MST_D; a.att += row0 * MST_D;
        a.z1h += row0 * MST_D; a.z1l += row0 * MST_D; a.x1h += row0 * MST_D; a.x1l += row0 * MST_D;
        a.pre += row0 * MST_FF; a.hid += row0 * MST_FF;
        a.z2h += row0 * MST_D; a.z2l += row0 * MST_D;
        a.lse += row0 * MST_H;
    }
}

static Drop make_drop(uint64_t seed, int layer, int site, float p, uint32_t elem_off = 0) {
    Drop d{0u, 0u, 1.0f};
    if (p <= 0.f) return d;
    d.key = mix32((uint32_t)seed ^ mix32((uint32_t)(seed >> 32) + 0x9E3779B9u * (uint32_t)(layer * 4 + site + 1))) + elem_off * 0x9E3779B9u;
    const double t = (double)p * 4294967296.0;
    d.thr = t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
    d.inv = 1.0f / (1.0f - p);
    return d;
}

static int train_check(mst_engine* e, int rows, int S, float p) {
    if (!e) return fail("null engine");
    CHECK(mst_weights_complete(e));
    if (S < 2 || S > e->S_max) return fail("train: S=%d outside 2..%d", S, e->S_max);
    if (rows < 1 || (size_t)rows * S > (size_t)e->cfg.max_rows * e->S_max) return fail("train: %d rows x %d tokens exceed the engine capacity", rows, S);
    if (!(p >= 0.f && p < 1.f)) return fail("train: dropout probability %g outside [0, 1)", (double)p);
    if ((size_t)rows * S * MST_FF >= 0xFFFFFFFFull || (size_t)rows * MST_H * S * S >= 0xFFFFFFFFull)
        return fail("train: batch too large for the 32-bit dropout counters");
    return 0;
}

extern "C" int64_t mst_train_tape_bytes(const mst_engine* e, int32_t rows, int32_t S) {
    if (!e || rows < 1 || S < 1) return -1;
    Tape t;
    return (int64_t)tape_layout(nullptr, e->cfg.num_layers, tape_rows(rows * S), &t);
}

template <int NKT>
static int launch_attn_train_n(const f16* qkv, f16* out, int S, int rows, Drop d, const unsigned char* keep, int qsplit, float* lse, hipStream_t st) {
    auto kern = k_attention_train<NKT>;
    const int smem = NKT * 32 * 256 * 2 + NKT * 32 * 4;
    CHECK(ensure_dyn_lds((const void*)kern, smem));
    hipLaunchKernelGGL(kern, dim3(rows * MST_H, qsplit ? NKT : 1), dim3(512), smem, st, qkv, out, S, d, keep, qsplit, lse);
    HIPCHECK(hipGetLastError());
    return 0;
}
template <int NKT>
static int launch_attn_bwd_n(const f16* qkv, const f16* att, const f16* datt, f16* dqkv, int S, int rows, Drop d,
                             const unsigned char* keep, const float* lse, int split, hipStream_t st) {
    auto kern = k_attention_bwd<NKT>;
    const int smem = NKT * 32 * 256 * 2 + NKT * 32 * 12;
    CHECK(ensure_dyn_lds((const void*)kern, smem));
    hipLaunchKernelGGL(kern, dim3(rows * MST_H, split ? 2 * NKT : 1), dim3(512), smem, st, qkv, att, datt, dqkv, S, d, keep, lse, split);
    HIPCHECK(hipGetLastError());
    return 0;
}
#define NKT_SWITCH(fn, S, ...)                                    \
    switch (((S) + 31) / 32) {                                    \
        case 1: return fn<1>(__VA_ARGS__);                        \
        case 2: return fn<2>(__VA_ARGS__);                        \
        case 3: return fn<3>(__VA_ARGS__);                        \
        case 4: return fn<4>(__VA_ARGS__);                        \
        case 5: return fn<5>(__VA_ARGS__);                        \
        case 6: return fn<6>(__VA_ARGS__);                        \
        case 7: return fn<7>(__VA_ARGS__);                        \
    }                                                             \
    return fail("attention: S=%d unsupported", S);
static int launch_attn_train(const f16* qkv, f16* out, int S, int rows, Drop d, const unsigned char* keep, int qsplit, float* lse, hipStream_t st) {
    NKT_SWITCH(launch_attn_train_n, S, qkv, out, S, rows, d, keep, qsplit, lse, st)
}
static int launch_attn_bwd(const f16* qkv, const f16* att, const f16* datt, f16* dqkv, int S, int rows, Drop d,
                           const unsigned char* keep, const float* lse, int split, hipStream_t st) {
    NKT_SWITCH(launch_attn_bwd_n, S, qkv, att, datt, dqkv, S, rows, d, keep, lse, split, st)
}

// h_in / h_out: [rows][S][512] float32 (clip-major token rows).  mdm_forstyledataset.py:622 `self.seqTransEncoder(xseq)`
// with nn.TransformerEncoderLayer semantics (post-norm, erf GELU, dropout p at the four sites of the layer).
// the eight layers from tape slot 0 to slot num_layers
static int train_stack_forward(mst_engine* e, const Tape& t, int rows, int S, float p_drop, uint64_t seed, const uint8_t* key_keep,
                               hipStream_t st, int clip0 = 0) {
    const int M = rows * S, nl = e->cfg.num_layers;
    // dropout counters of the four sites start at clip `clip0` of the tape (attention probabilities, 512-wide rows, hidden rows)
    const uint32_t o0 = (uint32_t)clip0 * MST_H * S * S, o1 = (uint32_t)clip0 * S * MST_D, o2 = (uint32_t)clip0 * S * MST_FF;
    e->prof_now = 0;
    const bool small = e->small_m > 0 && M <= e->small_m;
    const bool fast = small && e->small_fast;            // the four GEMMs as resident-tile kernels (mst_small.h), as in the sampling path
    if (fast) CHECK(ensure_small_packed_all(e, st));
    // round 6: at a clip or two (the objective's chained single-clip calls) the two LayerNorms of a layer ride in the GEMM behind them
    // (k_rows_gemm LNF = 1 with LnRows' training fields: dropout, z -> tape), as on the sampling path: 16 launches fewer per call.
    // LayerNorm1 in FFN1; LayerNorm2 in the NEXT layer's QKV GEMM, the last one as the rows kernel.  MST_TRAIN_SMALL_LN=0: every LayerNorm a launch.
    const bool lnf = fast && e->train_small_ln && M <= e->small_ln_m;
    for (int l = 0; small && l < nl; l++) {              // few token rows: 64 x 128 tiles, row-wise LayerNorm, query-split attention
        const LayerW& w = e->L[l];
        const TapeL& a = t.L[l];
        if (lnf && l > 0) {
            const LayerW& p = e->L[l - 1];
            const TapeL& pa = t.L[l - 1];
            LnRows ln{e->zacc, p.b2, p.g2, p.be2, pa.x1h, pa.x1l, t.sh[l], t.sl[l]};
            ln.z_hi = pa.z2h; ln.z_lo = pa.z2l; ln.d = make_drop(seed, l - 1, 3, p_drop, o1);
            CHECK((launch_rows_gemm<16, 0>(M, 3 * MST_D, nullptr, w.wsm_in, w.b_in, a.qkv, 3 * MST_D, st, &ln)));
        } else if (fast) CHECK((launch_rows_gemm<16, 0>(M, 3 * MST_D, t.sh[l], w.wsm_in, w.b_in, a.qkv, 3 * MST_D, st)));
        else {
            DEpiBiasF16<false> epi{w.b_in, a.qkv, 3 * MST_D, M};
            CHECK(launch_small(M, 3 * MST_D, RowsDirect{t.sh[l], MST_D}, w.w_in, MST_D, MST_D, epi, st));
        }
        CHECK(launch_attn_train(a.qkv, a.att, S, rows, make_drop(seed, l, 0, p_drop, o0), key_keep, 1, a.lse, st));
        {
            if (fast) CHECK((launch_rows_gemm<16, 2>(M, MST_D, a.att, w.wsm_out, nullptr, e->zacc, MST_D, st)));
            else {
                DEpiPlainF32 epi{e->zacc, MST_D, M};
                CHECK(launch_small(M, MST_D, RowsDirect{a.att, MST_D}, w.w_out, MST_D, MST_D, epi, st));
            }
            if (!lnf) {
                hipLaunchKernelGGL(k_ln_rows_train, dim3((M + 3) / 4), dim3(256), 0, st, e->zacc, w.b_out, w.g1, w.be1, t.sh[l], t.sl[l],
                                   a.z1h, a.z1l, a.x1h, a.x1l, M, make_drop(seed, l, 1, p_drop, o1));
                HIPCHECK(hipGetLastError());
            }
        }
        if (lnf) {
            const FfnTrain ft{a.pre, make_drop(seed, l, 2, p_drop, o2)};
            LnRows ln{e->zacc, w.b_out, w.g1, w.be1, t.sh[l], t.sl[l], a.x1h, a.x1l};
            ln.z_hi = a.z1h; ln.z_lo = a.z1l; ln.d = make_drop(seed, l, 1, p_drop, o1);
            CHECK((launch_rows_gemm<16, 3>(M, MST_FF, nullptr, w.wsm_1, w.b1, a.hid, MST_FF, st, &ln, &ft)));
        } else if (fast) {
            const FfnTrain ft{a.pre, make_drop(seed, l, 2, p_drop, o2)};
            CHECK((launch_rows_gemm<16, 3>(M, MST_FF, a.x1h, w.wsm_1, w.b1, a.hid, MST_FF, st, nullptr, &ft)));
        } else {
            DEpiRowOp<OpFfn1Train> epi{w.b1, M, OpFfn1Train{a.pre, a.hid, MST_FF, make_drop(seed, l, 2, p_drop, o2)}};
            CHECK(launch_small(M, MST_FF, RowsDirect{a.x1h, MST_D}, w.w1, MST_D, MST_D, epi, st));
        }
        {
            if (fast) CHECK((launch_rows_gemm<32, 2>(M, MST_D, a.hid, w.wsm_2, nullptr, e->zacc, MST_D, st)));
            else {
                DEpiPlainF32 epi{e->zacc, MST_D, M};
                CHECK(launch_small(M, MST_D, RowsDirect{a.hid, MST_FF}, w.w2, MST_FF, MST_FF, epi, st));
            }
            if (!lnf || l == nl - 1) {
                hipLaunchKernelGGL(k_ln_rows_train, dim3((M + 3) / 4), dim3(256), 0, st, e->zacc, w.b2, w.g2, w.be2, a.x1h, a.x1l,
                                   a.z2h, a.z2l, t.sh[l + 1], t.sl[l + 1], M, make_drop(seed, l, 3, p_drop, o1));
                HIPCHECK(hipGetLastError());
            }
        }
    }
    // round 6: everything of a layer behind the attention as ONE launch (k_layer_tail_train, mst_tail.h) that writes the tape slots
    // itself -- the sampling path's fused tail with dropout at the three sites, instead of three ring GEMMs moving 52-103 MB each
    const bool fused_tail = !small && e->train_fuse_tail && e->fuse_tail;
    if (fused_tail) {
        for (int l = 0; l < nl; l++) {
            LayerW& w = e->L[l];
            if (!w.tail_dirty) continue;
            hipLaunchKernelGGL(k_pack_tail, dim3(640), dim3(256), 0, st, w.w_out, w.w1, w.w2, w.wtail);
            HIPCHECK(hipGetLastError());
            w.tail_dirty = false;
        }
        CHECK(ensure_dyn_lds((const void*)k_layer_tail_train<4>, TailCfg::SMEM));
    }
    for (int l = 0; !small && l < nl; l++) {
        const LayerW& w = e->L[l];
        const TapeL& a = t.L[l];
        {
            DEpiBiasF16<false> epi{w.b_in, a.qkv, 3 * MST_D, M};
            CHECK((launch_wide(M, 3 * MST_D / 256, RowsDirect{t.sh[l], MST_D}, w.w_in, MST_D, MST_D, epi, st)));
        }
        CHECK(launch_attn_train(a.qkv, a.att, S, rows, make_drop(seed, l, 0, p_drop, o0), key_keep, 0, a.lse, st));
        if (fused_tail) {
            auto td = [&](int site, uint32_t off) { const Drop d = make_drop(seed, l, site, p_drop, off); return TailDrop{d.key, d.thr, d.inv}; };
            const TailTrain tt{t.sh[l], t.sl[l], a.z1h, a.z1l, a.x1h, a.x1l, a.pre, a.hid, a.z2h, a.z2l, td(1, o1), td(2, o2), td(3, o1)};
            hipLaunchKernelGGL(k_layer_tail_train<4>, dim3((M + 63) / 64), dim3(512), TailCfg::SMEM, st, a.att, w.wtail, w.b_out, w.g1, w.be1,
                               w.b1, w.b2, w.g2, w.be2, t.sh[l + 1], t.sl[l + 1], e->gelu_tab, M, tt);
            HIPCHECK(hipGetLastError());
            continue;
        }
        {
            DEpiResidLNTrain epi{w.b_out, w.g1, w.be1, t.sh[l], t.sl[l], a.z1h, a.z1l, a.x1h, a.x1l, M, make_drop(seed, l, 1, p_drop, o1)};
            CHECK((launch_gemm_dma<64, 512, 2, 2, LN_NS, 1, LN_BK>(dim3((M + 63) / 64, 1), RowsDirect{a.att, MST_D}, w.w_out, MST_D, MST_D, epi, st)));
        }
        {
            DEpiRowOp<OpFfn1Train> epi{w.b1, M, OpFfn1Train{a.pre, a.hid, MST_FF, make_drop(seed, l, 2, p_drop, o2)}};
            CHECK((launch_wide(M, MST_FF / 256, RowsDirect{a.x1h, MST_D}, w.w1, MST_D, MST_D, epi, st)));
        }
        {
            DEpiResidLNTrain epi{w.b2, w.g2, w.be2, a.x1h, a.x1l, a.z2h, a.z2l, t.sh[l + 1], t.sl[l + 1], M, make_drop(seed, l, 3, p_drop, o1)};
            CHECK((launch_gemm_dma<64, 512, 2, 2, LN_NS, 1, LN_BK>(dim3((M + 63) / 64, 1), RowsDirect{a.hid, MST_FF}, w.w2, MST_FF, MST_FF, epi, st)));
        }
    }
    return 0;
}

extern "C" int mst_train_forward(mst_engine* e, const float* h_in, int32_t rows, int32_t S, float p_drop, uint64_t seed,
                                 const uint8_t* key_keep, void* tape, float* h_out, void* stream) {
    CHECK(train_check(e, rows, S, p_drop));
    if (!h_in || !tape || !h_out) return fail("mst_train_forward: null argument");
    hipStream_t st = (hipStream_t)stream;
    ON_DEVICE(e->cfg.device);
    const int M = rows * S, nl = e->cfg.num_layers;
    Tape t;
    tape_layout((char*)tape, nl, tape_rows(M), &t);
    const size_t n = (size_t)M * MST_D;
    hipLaunchKernelGGL(k_split_stream, dim3(1024), dim3(256), 0, st, h_in, n, t.sh[0], t.sl[0]);
    HIPCHECK(hipGetLastError());
    CHECK(train_stack_forward(e, t, rows, S, p_drop, seed, key_keep, st));
    hipLaunchKernelGGL(k_join_stream, dim3(1024), dim3(256), 0, st, t.sh[nl], t.sl[nl], h_out, n);
    HIPCHECK(hipGetLastError());
    return 0;
}

static int train_ws(mst_engine* e) {
    TrainWS& t = e->tw;
    if (t.ready) return 0;
    const size_t Mp = (size_t)e->M_pad;
    t.split_cap = 64;
    CHECK(dmalloc(&t.g0, Mp * MST_D));
    CHECK(dmalloc(&t.g1, Mp * MST_D));
    for (int i = 0; i < 2; i++) {
        CHECK(dmalloc(&t.dbr2[i], Mp * MST_D));
        CHECK(dmalloc(&t.dbr1[i], Mp * MST_D));
        CHECK(dmalloc(&t.dpre[i], Mp * MST_FF));
        CHECK(dmalloc(&t.hidr[i], Mp * MST_FF));
        CHECK(dmalloc(&t.dqkv[i], Mp * 3 * MST_D));
        HIPCHECK(hipEventCreateWithFlags(&t.ev_side[i], hipEventDisableTiming));
    }
    HIPCHECK(hipEventCreateWithFlags(&t.ev_ready, hipEventDisableTiming));
    for (int i = 0; i < 16; i++) HIPCHECK(hipEventCreateWithFlags(&t.ev_layer[i], hipEventDisableTiming));
    CHECK(dmalloc(&t.datt, Mp * MST_D));
    t.part_cap = t.split_cap * (size_t)3 * MST_D * MST_D;
    CHECK(dmalloc(&t.part, t.part_cap));
    CHECK(dmalloc(&t.ln_part, (size_t)512 * 3 * MST_D));
    CHECK(dmalloc(&t.ln_part2, (size_t)512 * 3 * MST_D));
    CHECK(dmalloc(&t.cs_part, (Mp / 128 + 2) * (size_t)3 * MST_D));
    CHECK(dmalloc(&t.zeros, 3 * MST_D + MST_D));          // zero bias [1536] + a 512-float dump for unwanted reductions
    CHECK(dmalloc(&t.gscale, 2));
    CHECK(dmalloc(&t.amax, 1));
    t.ready = true;
    return 0;
}

static int wgrad_flush(mst_engine* e, hipStream_t st) {
    TrainWS& t = e->tw;
    if (t.n_red) {
        hipLaunchKernelGGL(k_splitk_reduce_multi, dim3(768, t.n_red), dim3(256), 0, st, t.red_jobs, t.gscale);
        HIPCHECK(hipGetLastError());
    }
    t.n_red = 0;
    t.part_used = 0;
    return 0;
}

// dW[n_out][k_in] += unscale * dY^T X   (dY: [M][n_out] f16, X: [M][k_in] f16), db += unscale * colsum(dY) if db
static int wgrad(mst_engine* e, const f16* dY, int n_out, const f16* X, int k_in, int M, float* dW, float* db, hipStream_t st) {
    TrainWS& t = e->tw;
    // tiles x splits ~ 128 workgroups (MST_WGRAD_WGS): the partials' round trip grows with the splits -- stack backward at 64 clips 3.11 / 2.96 / 2.85 /
    // 2.95 / 3.11 / 3.37 ms at 64 / 96 / 128 / 256 / 384 / 512, and the dgrad chain on the other stream uses the CUs left; every split contracts a
    // whole number of 32-token slabs
    const int tiles = (n_out / 128) * (k_in / 256);
    static const int wg_target = [] { const char* v = getenv("MST_WGRAD_WGS"); return v ? atoi(v) : 128; }();
    // (round 6, measured and removed: twice the workgroups for layer 0's weight gradients -- they run when the dgrad chain beside them has
    // ended -- 10.11 / 10.15 / 10.20 ms per iteration without, 10.15 / 10.18 / 10.22 ms with: the wgrad stream's tail is not short of CUs)
    int nsplit = (wg_target + tiles - 1) / tiles;
    const int slabs = (M + 31) / 32;
    // round 6: splits a multiple of 8 and every split's tiles on ONE XCD (k_wgrad_tr's xcd_tiles): the slabs of dY and X a split walks are
    // requested by 4 + 4 (W1 / W2) tiles, and those used to sit on all eight L2s.  Same-box A/Bs of the fine-tune line, three alternating rounds
    // on two boxes: 9.76 / 9.86 / 10.04 -> 9.67 / 9.66 / 9.75 ms and 9.95 / 9.94 / 9.89 -> 9.82 / 9.89 / 9.84 ms (the launch itself 49.7 -> 47.8 us:
    // most of the gain is the dgrad stream's, which shares the memory system with it).  A deeper slab ring (5 / 6 slabs: 120 / 144 KB of LDS)
    // was measured with it and is slower -- the 72 KB ring leaves the CU's LDS to a dgrad workgroup.  MST_WGRAD_XCD=0: the 3-D grid of round 5.
    static const int xcd_on = [] { const char* v = getenv("MST_WGRAD_XCD"); return v ? atoi(v) : 1; }();
    bool xcd = xcd_on && M > SMALL_M && slabs >= 64;
    if (xcd) nsplit = ((nsplit + 7) / 8) * 8;
    if (nsplit > slabs) nsplit = slabs;
    if ((size_t)nsplit > t.split_cap) nsplit = (int)t.split_cap;
    if (M <= SMALL_M) nsplit = 1;        // few tokens: one workgroup per tile adds its product straight into dW (no partials, no reduce launch)
    const int kchunk = ((slabs + nsplit - 1) / nsplit) * 32;
    nsplit = (M + kchunk - 1) / kchunk;
    if (nsplit % 8) xcd = false;
    auto kern = k_wgrad_tr;
    CHECK(ensure_dyn_lds((const void*)kern, WgTile::SMEM));
    static_assert(DEpiF32::smem_bytes<128, 256>() <= WgTile::SMEM, "epilogue tile must fit the ring");
    const size_t nelem = (size_t)n_out * k_in;
    // profiling on (mst_profile_enable): every k_wgrad_tr launch of a training pass is bracketed by HIP events on ITS stream (bench.py --mode
    // finetune reads the family "wgrad_tr": the training path's dominant kernel, timed in the run that reports it)
    const int prof_keep = e->prof_now;
    e->prof_now = e->prof_on;
    if (nsplit == 1) {
        DEpiF32 epi{dW, dW, k_in, n_out, t.gscale};
        {
            ProfScope ps(e, FAM_WGRAD, st);
            hipLaunchKernelGGL(kern, dim3(n_out / 128, k_in / 256, 1), dim3(512), WgTile::SMEM, st, dY, n_out, X, k_in, M, kchunk, nelem, epi, 0);
        }
        e->prof_now = prof_keep;
        HIPCHECK(hipGetLastError());
    } else {
        const bool defer = t.red_defer && t.n_red < 4 && t.part_used + (size_t)nsplit * nelem <= t.part_cap;
        if (!defer) CHECK(wgrad_flush(e, st));               // (pending regions start at offset 0 as well)
        float* part = t.part + (defer ? t.part_used : 0);
        DEpiF32 epi{nullptr, part, k_in, n_out};
        {
            ProfScope ps(e, FAM_WGRAD, st);
            if (xcd) hipLaunchKernelGGL(kern, dim3(tiles * nsplit), dim3(512), WgTile::SMEM, st, dY, n_out, X, k_in, M, kchunk, nelem, epi, tiles);
            else hipLaunchKernelGGL(kern, dim3(n_out / 128, k_in / 256, nsplit), dim3(512), WgTile::SMEM, st, dY, n_out, X, k_in, M, kchunk, nelem, epi, 0);
        }
        e->prof_now = prof_keep;
        HIPCHECK(hipGetLastError());
        if (defer) {
            t.red_jobs.j[t.n_red++] = RedJob{part, dW, nelem, nsplit, 0};
            t.part_used += (size_t)nsplit * nelem;
        } else {
            hipLaunchKernelGGL(k_splitk_reduce, dim3((unsigned)((nelem / 4 + 255) / 256)), dim3(256), 0, st, t.part, nsplit, nelem, t.gscale, dW);
            HIPCHECK(hipGetLastError());
        }
    }
    if (db) {
        const int rpb = 128, nrb = (M + rpb - 1) / rpb;      // bias gradient: row-block partials, then an ordered sum (no float atomics)
        hipLaunchKernelGGL(k_colsum_f16, dim3(n_out / 256, nrb), dim3(256), 0, st, dY, n_out, M, rpb, t.gscale, db, t.cs_part);
        HIPCHECK(hipGetLastError());
        if (nrb > 1) {
            hipLaunchKernelGGL(k_sum_partials, dim3((n_out + kFinOut - 1) / kFinOut), dim3(256), 0, st, t.cs_part, nrb, n_out, t.gscale, db);
            HIPCHECK(hipGetLastError());
        }
    }
    return 0;
}

// grads: HOST array of num_layers * 12 device pointers in nn.TransformerEncoderLayer parameter order
// (in_proj_weight, in_proj_bias, out_proj.weight, out_proj.bias, linear1.weight, linear1.bias, linear2.weight,
//  linear2.bias, norm1.weight, norm1.bias, norm2.weight, norm2.bias); every buffer is ACCUMULATED into.
// device-side gradient scaling: gscale <- the power of two that brings max |g| into [16, 32)
static int grad_scale_from(mst_engine* e, const float* g, size_t n, hipStream_t st) {
    TrainWS& w_ = e->tw;
    HIPCHECK(hipMemsetAsync(w_.amax, 0, sizeof(unsigned), st));
    hipLaunchKernelGGL(k_amax, dim3(512), dim3(256), 0, st, g, n, w_.amax);
    hipLaunchKernelGGL(k_grad_scale, dim3(1), dim3(1), 0, st, w_.amax, w_.gscale);
    HIPCHECK(hipGetLastError());
    return 0;
}

// the eight layers backwards: on entry tw.g1 holds the SCALED gradient wrt the stack output, on exit the scaled gradient
// wrt tape slot 0
static int train_stack_backward(mst_engine* e, const Tape& t, int rows, int S, float p_drop, uint64_t seed, const uint8_t* key_keep,
                                float* const* grads, hipStream_t st) {
    TrainWS& w_ = e->tw;
    const int M = rows * S, nl = e->cfg.num_layers;
    float* gA = w_.g0;      // dz buffers
    float* gB = w_.g1;      // gradient wrt the current layer output
    // one block per CU at most; their partial sums are added in block order (k_ln_bwd_finish).  Sixteen waves per block above LN_BWD_WIDE_M rows
    const bool ln_wide = M > LN_BWD_WIDE_M;
    const int ln_wpb = ln_wide ? 16 : 4;
    const int ln_blocks = (M + ln_wpb - 1) / ln_wpb < 256 ? (M + ln_wpb - 1) / ln_wpb : 256;
    auto ln_bwd = ln_wide ? k_ln_bwd<16> : k_ln_bwd<4>;
    const bool small = e->small_m > 0 && M <= e->small_m;
    // at batch size LayerNorm1's backward rides in the epilogue of the dgrad GEMM in front of it (DEpiLnBwd; MST_FUSE_LN_BWD=0: two launches)
    const int ln_tiles = (M + 63) / 64;
    const bool fuse_ln = e->fuse_ln_bwd && !small && ln_tiles <= 512;   // (512: the partial-sum buffer's rows)
    e->prof_now = 0;
    // The dgrad chain (LayerNorm / GELU / attention backward and the four dgrad GEMMs) is serial; the four wgrads of a
    // layer only consume its by-products, so they run on a second stream beside it.  Their f16 operands are
    // double-buffered by layer parity: the chain of layer l waits for the wgrads of layer l + 2, not l + 1.
    hipStream_t sw = e->wgrad_stream_on ? e->aux_stream[0] : st;
    const bool two = sw != st;
    bool side_used[2] = {false, false};
#define TO_SIDE()                                                   \
    if (two && grads) {                                                \
        HIPCHECK(hipEventRecord(w_.ev_ready, st));                  \
        HIPCHECK(hipStreamWaitEvent(sw, w_.ev_ready, 0));           \
    }
    // with parameter gradients the fused launch also writes dpre, hid and dbr1 as 8-byte pieces from the accumulator layout and is no faster
    // than the three launches yet (113 against 128-155 us beside the wgrad stream: LAB_NOTES R6.9); a frozen stack writes none of them
    const bool fused_bwd = (e->train_fuse_bwd_tail >= 2 || (e->train_fuse_bwd_tail == 1 && !grads)) && !small && ln_tiles <= 512;
    if (fused_bwd) {
        for (int l = 0; l < nl; l++) {
            LayerW& w = e->L[l];
            if (!w.wtail_bwd) CHECK(dmalloc(&w.wtail_bwd, TailCfg::LAYER_BYTES / 2));
            if (!w.tailb_dirty) continue;
            hipLaunchKernelGGL(k_pack_tail_bwd, dim3(640), dim3(256), 0, st, w.w_outT, w.w2T, w.w1T, w.wtail_bwd);
            HIPCHECK(hipGetLastError());
            w.tailb_dirty = false;
        }
        CHECK(ensure_dyn_lds((const void*)k_layer_tail_bwd<true>, TailBwdCfg::SMEM));
        CHECK(ensure_dyn_lds((const void*)k_layer_tail_bwd<false>, TailBwdCfg::SMEM));
        CHECK(ensure_dyn_lds((const void*)k_layer_tail_bwd<false, true>, TailBwdCfg::SMEM));
    }
    const bool fused_ln2 = fused_bwd && e->train_fuse_ln2_bwd && !grads;      // (frozen stacks: with gradients dbr2 and LayerNorm2's sums must reach HBM anyway)
    static const int red_batch = [] { const char* v = getenv("MST_WGRAD_REDUCE_BATCH"); return v ? atoi(v) : 1; }();
    w_.red_defer = red_batch != 0 && grads != nullptr;
    w_.n_red = 0;
    w_.part_used = 0;
    for (int l = nl - 1; l >= 0; l--) {
        const LayerW& w = e->L[l];
        const TapeL& a = t.L[l];
        const int par = l & 1;
        f16 *dbr2 = w_.dbr2[par], *dbr1 = w_.dbr1[par], *dpre = w_.dpre[par], *dqkv = w_.dqkv[par];
        // grads == NULL (frozen stack): no wgrads; the LayerNorm kernel's small reductions land in a scratch vector
        float* dump[12];
        for (int i = 0; i < 12; i++) dump[i] = w_.zeros + 3 * MST_D;
        float* const* G = grads ? grads + (size_t)l * 12 : dump;
        const bool wg = grads != nullptr;
        for (int i = 0; i < 12; i++) if (!G[i]) return fail("mst_train_backward: null gradient buffer (layer %d, tensor %d)", l, i);
        if (two && side_used[par]) HIPCHECK(hipStreamWaitEvent(st, w_.ev_side[par], 0));
        // LayerNorm2 backward: gB -> dz2 (gA, fp32) and the branch gradient dbr2 (f16); dgamma2, dbeta2, db2
        // (round 6, MST_LN_FINISH_MERGE: with parameter gradients and the unfused tail, LayerNorm2's partial sums wait in a buffer of their own and
        // ONE launch behind LayerNorm1's backward finishes both -- a launch less per layer on the dgrad chain, same sums in the same order)
        static const int merge_on = [] { const char* v = getenv("MST_LN_FINISH_MERGE"); return v ? atoi(v) : 1; }();
        const bool merge_fin = merge_on && wg && !fused_bwd;
        if (!fused_ln2) {
        hipLaunchKernelGGL(ln_bwd, dim3(ln_blocks), dim3(64 * ln_wpb), 0, st, gB, a.z2h, a.z2l, w.g2, M, make_drop(seed, l, 3, p_drop), w_.gscale,
                           gA, dbr2, merge_fin ? w_.ln_part2 : w_.ln_part);
        if (wg && !merge_fin) hipLaunchKernelGGL(k_ln_bwd_finish, dim3(3 * MST_D / kFinOut), dim3(256), 0, st, w_.ln_part, ln_blocks, w_.gscale, G[10], G[11], G[7]);
        HIPCHECK(hipGetLastError());
        }
        // d pre = (dbr2 W2) * mask * gelu'(pre); the same epilogue regenerates hid = dropout(GELU(pre)), dW2's operand, from the pre it reads
        // anyway (round 6: the fused training forward no longer writes hid to the tape -- 2 KB per token and layer less, and a frozen stack
        // never needs it)
        f16* hidr = wg ? w_.hidr[par] : nullptr;
        if (fused_bwd) {
            // round 6: FFN2 dgrad + GELU' + FFN1 dgrad + LayerNorm1 backward + out-proj dgrad in ONE launch (mst_tail_bwd.h): dz1 -> gA in place,
            // d att; with gradients wanted also dpre, hid, dbr1 and the tiles' [dgamma1 | dbeta1 | db_out] sums
            auto td = [&](int site) { const Drop d = make_drop(seed, l, site, p_drop); return TailDrop{d.key, d.thr, d.inv}; };
            const TailBwdOut o{wg ? dpre : nullptr, hidr, wg ? dbr1 : nullptr, wg ? w_.ln_part : nullptr};
            const TailBwdLn2 n2{gB, a.z2h, a.z2l, w.g2, td(3)};
            if (fused_ln2) hipLaunchKernelGGL((k_layer_tail_bwd<false, true>), dim3(ln_tiles), dim3(512), TailBwdCfg::SMEM, st, dbr2, w.wtail_bwd, a.pre, a.z1h, a.z1l, w.g1, gA, w_.datt, o, td(1), td(2), M, e->gelu_tab, n2);
            else if (wg) hipLaunchKernelGGL(k_layer_tail_bwd<true>, dim3(ln_tiles), dim3(512), TailBwdCfg::SMEM, st, dbr2, w.wtail_bwd, a.pre, a.z1h, a.z1l, w.g1, gA, w_.datt, o, td(1), td(2), M, e->gelu_tab);
            else hipLaunchKernelGGL(k_layer_tail_bwd<false>, dim3(ln_tiles), dim3(512), TailBwdCfg::SMEM, st, dbr2, w.wtail_bwd, a.pre, a.z1h, a.z1l, w.g1, gA, w_.datt, o, td(1), td(2), M, e->gelu_tab);
            if (wg) hipLaunchKernelGGL(k_ln_bwd_finish, dim3(3 * MST_D / kFinOut), dim3(256), 0, st, w_.ln_part, ln_tiles, w_.gscale, G[8], G[9], G[3]);
            HIPCHECK(hipGetLastError());
            TO_SIDE()
            if (wg) CHECK(wgrad(e, dbr2, MST_D, hidr, MST_FF, M, G[6], nullptr, sw));                  // dW2 += dbr2^T hid
            if (wg) CHECK(wgrad(e, dpre, MST_FF, a.x1h, MST_D, M, G[4], G[5], sw));                    // dW1 += dpre^T x1, db1
            if (wg) CHECK(wgrad(e, dbr1, MST_D, a.att, MST_D, M, G[2], nullptr, sw));                  // dW_out += dbr1^T att
        } else {
        {
            DEpiRowOp<OpGeluBwd> epi{nullptr, M, OpGeluBwd{a.pre, dpre, MST_FF, make_drop(seed, l, 2, p_drop), hidr}};
            CHECK(small ? launch_small(M, MST_FF, RowsDirect{dbr2, MST_D}, w.w2T, MST_D, MST_D, epi, st)
                        : launch_wide(M, MST_FF / 256, RowsDirect{dbr2, MST_D}, w.w2T, MST_D, MST_D, epi, st));
        }
        TO_SIDE()
        if (wg) CHECK(wgrad(e, dbr2, MST_D, hidr, MST_FF, M, G[6], nullptr, sw));                  // dW2 += dbr2^T hid
        if (wg) CHECK(wgrad(e, dpre, MST_FF, a.x1h, MST_D, M, G[4], G[5], sw));                    // dW1 += dpre^T x1, db1
        if (fuse_ln) {
            // g(x1) = dpre W1 + dz2 and LayerNorm1's backward behind it in ONE launch (DEpiLnBwd): dz1 -> gA in place, dbr1, the tiles' sums
            DEpiLnBwd epi{gA, a.z1h, a.z1l, w.g1, M, make_drop(seed, l, 1, p_drop), gA, dbr1, w_.ln_part};
            CHECK((launch_gemm_dma<64, 512, 2, 2, LN_NS, 1, LN_BK>(dim3(ln_tiles, 1), RowsDirect{dpre, MST_FF}, w.w1T, MST_FF, MST_FF, epi, st)));
            if (merge_fin) hipLaunchKernelGGL(k_ln_bwd_finish2, dim3(3 * MST_D / kFinOut, 2), dim3(256), 0, st, LnFinJob{w_.ln_part2, ln_blocks, G[10], G[11], G[7]},
                                              LnFinJob{w_.ln_part, ln_tiles, G[8], G[9], G[3]}, w_.gscale);
            else if (wg) hipLaunchKernelGGL(k_ln_bwd_finish, dim3(3 * MST_D / kFinOut), dim3(256), 0, st, w_.ln_part, ln_tiles, w_.gscale, G[8], G[9], G[3]);
            HIPCHECK(hipGetLastError());
        } else {
        // g(x1) = dpre W1 + dz2  -> gB
        {
            DEpiF32 epi{gA, gB, MST_D, M};
            CHECK(small ? launch_small(M, MST_D, RowsDirect{dpre, MST_FF}, w.w1T, MST_FF, MST_FF, epi, st)
                        : launch_wide(M, MST_D / 256, RowsDirect{dpre, MST_FF}, w.w1T, MST_FF, MST_FF, epi, st));
        }
        // LayerNorm1 backward: gB -> dz1 (gA), dbr1 = d(out-proj output); dgamma1, dbeta1, db_out
        hipLaunchKernelGGL(ln_bwd, dim3(ln_blocks), dim3(64 * ln_wpb), 0, st, gB, a.z1h, a.z1l, w.g1, M, make_drop(seed, l, 1, p_drop), w_.gscale,
                           gA, dbr1, w_.ln_part);
        if (merge_fin) hipLaunchKernelGGL(k_ln_bwd_finish2, dim3(3 * MST_D / kFinOut, 2), dim3(256), 0, st, LnFinJob{w_.ln_part2, ln_blocks, G[10], G[11], G[7]},
                                          LnFinJob{w_.ln_part, ln_blocks, G[8], G[9], G[3]}, w_.gscale);
        else if (wg) hipLaunchKernelGGL(k_ln_bwd_finish, dim3(3 * MST_D / kFinOut), dim3(256), 0, st, w_.ln_part, ln_blocks, w_.gscale, G[8], G[9], G[3]);
        HIPCHECK(hipGetLastError());
        }
        TO_SIDE()
        if (wg) CHECK(wgrad(e, dbr1, MST_D, a.att, MST_D, M, G[2], nullptr, sw));                  // dW_out += dbr1^T att
        // d att = dbr1 W_out
        {
            DEpiBiasF16<false> epi{w_.zeros, w_.datt, MST_D, M};
            CHECK(small ? launch_small(M, MST_D, RowsDirect{dbr1, MST_D}, w.w_outT, MST_D, MST_D, epi, st)
                        : launch_wide(M, MST_D / 256, RowsDirect{dbr1, MST_D}, w.w_outT, MST_D, MST_D, epi, st));
        }
        }   // !fused_bwd
        // attention backward -> d qkv
        CHECK(launch_attn_bwd(a.qkv, a.att, w_.datt, dqkv, S, rows, make_drop(seed, l, 0, p_drop), key_keep, a.lse, small ? 1 : 0, st));
        TO_SIDE()
        if (wg) CHECK(wgrad(e, dqkv, 3 * MST_D, t.sh[l], MST_D, M, G[0], G[1], sw));               // dW_in += dqkv^T x_in, db_in
        if (wg) CHECK(wgrad_flush(e, sw));                                                         // the layer's split-K reduces, one launch
        if (two && wg) {
            HIPCHECK(hipEventRecord(w_.ev_side[par], sw));
            side_used[par] = true;
        }
        if (wg) {
            // everything that writes layer l's 12 gradient tensors is enqueued: the wgrads on `sw` (which waited for the
            // dgrad chain up to the attention backward, i.e. for both LayerNorm-backward kernels of this layer too).
            // A data-parallel reducer makes its communication stream wait for THIS event (mst_train_wait_layer_grads) and
            // starts the layer's all-reduce while the layers below are still being differentiated.
            HIPCHECK(hipEventRecord(w_.ev_layer[l], sw));
            w_.layer_done[l] = true;
        }
        // g(x_in) = dqkv W_in + dz1 -> gB
        {
            DEpiF32 epi{gA, gB, MST_D, M};
            CHECK(small ? launch_small(M, MST_D, RowsDirect{dqkv, 3 * MST_D}, w.w_inT, 3 * MST_D, 3 * MST_D, epi, st)
                        : launch_wide(M, MST_D / 256, RowsDirect{dqkv, 3 * MST_D}, w.w_inT, 3 * MST_D, 3 * MST_D, epi, st));
        }
    }
    w_.red_defer = false;
#undef TO_SIDE
    if (two)
        for (int i = 0; i < 2; i++)
            if (side_used[i]) HIPCHECK(hipStreamWaitEvent(st, w_.ev_side[i], 0));
    return 0;
}

extern "C" int mst_train_backward(mst_engine* e, const void* tape, const float* d_out, int32_t rows, int32_t S, float p_drop,
                                  uint64_t seed, const uint8_t* key_keep, float* d_in, float* const* grads, void* stream) {
    CHECK(train_check(e, rows, S, p_drop));
    if (!tape || !d_out) return fail("mst_train_backward: null argument");
    if (!grads && !d_in) return 0;                       // nothing asked for
    hipStream_t st = (hipStream_t)stream;
    ON_DEVICE(e->cfg.device);
    CHECK(train_ws(e));
    TrainWS& w_ = e->tw;
    const int M = rows * S, nl = e->cfg.num_layers;
    Tape t;
    tape_layout((char*)const_cast<void*>(tape), nl, tape_rows(M), &t);
    const size_t n = (size_t)M * MST_D;
    CHECK(grad_scale_from(e, d_out, n, st));
    hipLaunchKernelGGL(k_scale_f32, dim3(1024), dim3(256), 0, st, d_out, n, w_.gscale, 0, w_.g1);
    HIPCHECK(hipGetLastError());
    CHECK(train_stack_backward(e, t, rows, S, p_drop, seed, key_keep, grads, st));
    if (d_in) {
        hipLaunchKernelGGL(k_scale_f32, dim3(1024), dim3(256), 0, st, w_.g1, n, w_.gscale, 1, d_in);
        HIPCHECK(hipGetLastError());
    }
    return 0;
}

extern "C" int mst_train_wait_layer_grads(mst_engine* e, int32_t layer, void* stream) {
    if (!e || layer < 0 || layer >= e->cfg.num_layers) return fail("mst_train_wait_layer_grads: bad arguments");
    if (!e->tw.ready || !e->tw.layer_done[layer]) return fail("mst_train_wait_layer_grads: no backward pass has produced gradients of layer %d yet", layer);
    ON_DEVICE(e->cfg.device);
    HIPCHECK(hipStreamWaitEvent((hipStream_t)stream, e->tw.ev_layer[layer], 0));
    return 0;
}

// ---- the whole denoiser inside the native graph: conditioning token, pose embedding, positional-encoding dropout, the
// trainable stack, output projection (StyleDiffusion.forward / MDM.forward, mdm_forstyledataset.py:602-625 / :315-364, in
// train mode).  Text conditioning as for mst_forward: mst_set_text(batch, cfg = 0) first.
static Drop pe_drop(uint64_t seed, float p, uint32_t elem_off = 0) { return make_drop(seed, 32, 0, p, elem_off); }       // site of its own, beyond any layer

extern "C" int mst_train_model_forward(mst_engine* e, const float* x, const int64_t* t_idx, int32_t batch, int32_t frames,
                                       float p_drop, float p_pe, uint64_t seed, void* tape, float* out, int32_t clip0, int32_t tape_clips,
                                       void* stream) {
    CHECK(check_ready(e, batch, frames, 0));
    const int S = frames + 1;
    if (tape_clips <= 0) { tape_clips = batch; clip0 = 0; }
    if (clip0 < 0 || clip0 + batch > tape_clips) return fail("mst_train_model_forward: clips %d..%d outside a tape of %d", clip0, clip0 + batch, tape_clips);
    CHECK(train_check(e, tape_clips, S, p_drop));
    if (!(p_pe >= 0.f && p_pe < 1.f)) return fail("mst_train_model_forward: positional-encoding dropout %g outside [0, 1)", (double)p_pe);
    if (!x || !t_idx || !tape || !out) return fail("mst_train_model_forward: null argument");
    hipStream_t st = (hipStream_t)stream;
    ON_DEVICE(e->cfg.device);
    const int M = batch * S, nl = e->cfg.num_layers;
    Tape t;
    tape_layout((char*)tape, nl, tape_rows(tape_clips * S), &t);
    tape_shift(t, nl, (size_t)clip0 * S);
    e->prof_now = 0;
    CHECK(ensure_packed(e, st, false));
    const bool table = e->temb_table_on != 0;
    if (table) CHECK(ensure_temb_table(e, st));
    else CHECK(timestep_rows(e, (const long long*)t_idx, batch, st));
    WS ws = ws_slice(e, 0, frames);
    ws.hx = t.sh[0];                                      // the token stream is assembled straight into tape slot 0
    ws.hl = t.sl[0];
    // round 6: PositionalEncoding's dropout rides in the embedding kernel's epilogue (MST_TRAIN_FUSE_PE_DROP=0: k_dropout_stream behind it)
    static const int fuse_pe = [] { const char* v = getenv("MST_TRAIN_FUSE_PE_DROP"); return v ? atoi(v) : 1; }();
    const Drop ped = pe_drop(seed, p_pe, (uint32_t)clip0 * S * MST_D);
    CHECK(assemble_stream(e, ws, x, batch, batch, frames, -1, batch, st, batch, LoopRef(), table ? (const long long*)t_idx : nullptr,
                          (fuse_pe && p_pe > 0.f) ? &ped : nullptr));
    if (p_pe > 0.f && !fuse_pe) {
        hipLaunchKernelGGL(k_dropout_stream, dim3(1024), dim3(256), 0, st, t.sh[0], t.sl[0], (size_t)M * MST_D,
                           pe_drop(seed, p_pe, (uint32_t)clip0 * S * MST_D));
        HIPCHECK(hipGetLastError());
    }
    CHECK(train_stack_forward(e, t, batch, S, p_drop, seed, nullptr, st, clip0));
    ws.hx = t.sh[nl];
    ws.hl = t.sl[nl];
    StepArgs sa{};
    return launch_out_nt<0>(e, ws, 0, batch, frames, out, sa, st, nullptr, nullptr, 1, false, true);
}

extern "C" int mst_train_model_backward(mst_engine* e, const void* tape, const float* d_out, int32_t batch, int32_t frames,
                                        float p_drop, float p_pe, uint64_t seed, float* d_x, float* const* grads, void* stream) {
    const int S = frames + 1;
    CHECK(train_check(e, batch, S, p_drop));
    if (!tape || !d_out) return fail("mst_train_model_backward: null argument");
    if (!grads && !d_x) return 0;
    hipStream_t st = (hipStream_t)stream;
    ON_DEVICE(e->cfg.device);
    CHECK(train_ws(e));
    TrainWS& w_ = e->tw;
    const int M = batch * S, nl = e->cfg.num_layers, F = e->cfg.feats, tot = batch * frames;
    Tape t;
    tape_layout((char*)const_cast<void*>(tape), nl, tape_rows(M), &t);
    const size_t n_out = (size_t)batch * F * frames, n = (size_t)M * MST_D;
    e->prof_now = 0;
    CHECK(grad_scale_from(e, d_out, n_out, st));
    // output projection backward: token-stream gradient rows = (scaled d_out as frame rows) x W_out; token 0 gets none
    hipLaunchKernelGGL(k_frames_f16, dim3((frames + 31) / 32, e->kin_pad / 32, batch), dim3(256), 0, st, d_out, F, frames, e->kin_pad, e->xt,
                       (const float*)w_.gscale, (const LoopDev*)nullptr, 0ull);
    hipLaunchKernelGGL(k_zero_token0, dim3((batch * MST_D + 255) / 256), dim3(256), 0, st, w_.g1, S, batch);
    HIPCHECK(hipGetLastError());
    {
        DEpiFramesGrad epi{w_.g1, frames, S, tot};
        CHECK((launch_gemm_dma<64, 512, 2, 2, 4, 1>(dim3((tot + 63) / 64, 1), RowsDirect{e->xt, e->kin_pad}, e->w_pose_outT, e->kin_pad,
                                                   e->kin_pad, epi, st)));
    }
    CHECK(train_stack_backward(e, t, batch, S, p_drop, seed, nullptr, grads, st));
    if (d_x) {
        if (p_pe > 0.f) {
            hipLaunchKernelGGL(k_mask_f32, dim3(1024), dim3(256), 0, st, w_.g1, n, pe_drop(seed, p_pe));
            HIPCHECK(hipGetLastError());
        }
        // pose embedding backward: d x[b][f][t] = sum_k W_in[k][f] g[b, 1 + t, k]  == the output-projection kernel with W_in^T
        hipLaunchKernelGGL(k_f32_to_f16, dim3(1024), dim3(256), 0, st, w_.g1, n, w_.datt);
        HIPCHECK(hipGetLastError());
        WS ws = ws_slice(e, 0, frames);
        ws.hx = w_.datt;
        StepArgs sa{};
        CHECK(launch_out_nt<0>(e, ws, 0, batch, frames, d_x, sa, st, e->w_pose_inT, w_.zeros));
        hipLaunchKernelGGL(k_scale_f32, dim3(1024), dim3(256), 0, st, d_x, n_out, w_.gscale, 1, d_x);
        HIPCHECK(hipGetLastError());
    }
    return 0;
}

// ---- MotionEncoder.forward (mdm_forstyledataset.py:90-124) as one native call: [muQuery | sigmaQuery | pose embedding of the
// frames] + positional rows (+ its dropout), the 8 key-padding-masked layers of THIS engine, token 0 of the result.
// key_keep: [batch][frames + 2] (1 = real key; the two query tokens are always 1).  The engine holds the encoder's own layers
// and the prior's input projection / positional table (engine.load_state_dict with prior_prefix "mdm_model.").
static int menc_check(mst_engine* e, int batch, int frames, float p_drop, float p_pe) {
    if (!e) return fail("null engine");
    const int S = frames + 2;
    if (frames < 1 || S > e->S_max) return fail("motion encoder: %d frames + 2 query tokens exceed the engine's %d tokens", frames, e->S_max);
    CHECK(train_check(e, batch, S, p_drop));
    if (!(p_pe >= 0.f && p_pe < 1.f)) return fail("motion encoder: positional-encoding dropout %g outside [0, 1)", (double)p_pe);
    return 0;
}

extern "C" int mst_motion_encoder_forward(mst_engine* e, const float* x, const float* mu_query, const float* sigma_query,
                                          const uint8_t* key_keep, int32_t batch, int32_t frames, float p_drop, float p_pe,
                                          uint64_t seed, void* tape, float* mu_out, void* stream) {
    CHECK(menc_check(e, batch, frames, p_drop, p_pe));
    if (!x || !mu_query || !sigma_query || !tape || !mu_out) return fail("mst_motion_encoder_forward: null argument");
    hipStream_t st = (hipStream_t)stream;
    ON_DEVICE(e->cfg.device);
    const int S = frames + 2, M = batch * S, nl = e->cfg.num_layers, F = e->cfg.feats, tot = batch * frames;
    Tape t;
    tape_layout((char*)tape, nl, tape_rows(M), &t);
    e->prof_now = 0;
    hipLaunchKernelGGL(k_query_tokens, dim3((batch * 2 * MST_D + 255) / 256), dim3(256), 0, st, mu_query, sigma_query, e->pe, S, batch, t.sh[0], t.sl[0]);
    hipLaunchKernelGGL(k_frames_f16, dim3((frames + 31) / 32, e->kin_pad / 32, batch), dim3(256), 0, st, x, F, frames, e->kin_pad, e->xt,
                       (const float*)nullptr, (const LoopDev*)nullptr, 0ull);
    HIPCHECK(hipGetLastError());
    {
        DEpiEmbedIn epi{e->b_pose_in, e->pe, t.sh[0], t.sl[0], frames, S, tot, 0, 2};
        CHECK((launch_gemm_dma<64, 512, 2, 2, 4, 1>(dim3((tot + 63) / 64, 1), RowsDirect{e->xt, e->kin_pad}, e->w_pose_in, e->kin_pad, e->kin_pad, epi, st)));
    }
    if (p_pe > 0.f) {
        hipLaunchKernelGGL(k_dropout_stream, dim3(1024), dim3(256), 0, st, t.sh[0], t.sl[0], (size_t)M * MST_D, pe_drop(seed, p_pe));
        HIPCHECK(hipGetLastError());
    }
    CHECK(train_stack_forward(e, t, batch, S, p_drop, seed, key_keep, st));
    hipLaunchKernelGGL(k_gather_token0, dim3((batch * MST_D + 255) / 256), dim3(256), 0, st, t.sh[nl], t.sl[nl], S, batch, mu_out);
    HIPCHECK(hipGetLastError());
    return 0;
}

extern "C" int mst_motion_encoder_backward(mst_engine* e, const void* tape, const float* d_mu, const uint8_t* key_keep, int32_t batch,
                                           int32_t frames, float p_drop, float p_pe, uint64_t seed, float* d_x, void* stream) {
    CHECK(menc_check(e, batch, frames, p_drop, p_pe));
    if (!tape || !d_mu || !d_x) return fail("mst_motion_encoder_backward: null argument");
    hipStream_t st = (hipStream_t)stream;
    ON_DEVICE(e->cfg.device);
    CHECK(train_ws(e));
    TrainWS& w_ = e->tw;
    const int S = frames + 2, M = batch * S, nl = e->cfg.num_layers, F = e->cfg.feats;
    Tape t;
    tape_layout((char*)const_cast<void*>(tape), nl, tape_rows(M), &t);
    const size_t n = (size_t)M * MST_D, n_out = (size_t)batch * F * frames;
    e->prof_now = 0;
    CHECK(grad_scale_from(e, d_mu, (size_t)batch * MST_D, st));
    hipLaunchKernelGGL(k_seed_token0, dim3(2048), dim3(256), 0, st, d_mu, (const float*)w_.gscale, S, batch, w_.g1);   // only token 0 of every clip carries a gradient
    HIPCHECK(hipGetLastError());
    CHECK(train_stack_backward(e, t, batch, S, p_drop, seed, key_keep, nullptr, st));      // frozen stack: input gradient only
    if (p_pe > 0.f) {
        hipLaunchKernelGGL(k_mask_f32, dim3(1024), dim3(256), 0, st, w_.g1, n, pe_drop(seed, p_pe));
        HIPCHECK(hipGetLastError());
    }
    // pose embedding backward on the frame tokens (rows 2..): d x[b][f][t] = sum_k W_in[k][f] g[b, 2 + t, k]
    hipLaunchKernelGGL(k_f32_to_f16, dim3(1024), dim3(256), 0, st, w_.g1, n, w_.datt);
    HIPCHECK(hipGetLastError());
    WS ws = ws_slice(e, 0, frames);
    ws.hx = w_.datt;
    StepArgs sa{};
    CHECK(launch_out_nt<0>(e, ws, 0, batch, frames, d_x, sa, st, e->w_pose_inT, w_.zeros, 2));
    hipLaunchKernelGGL(k_scale_f32, dim3(1024), dim3(256), 0, st, d_x, n_out, w_.gscale, 1, d_x);
    HIPCHECK(hipGetLastError());
    return 0;
}

// keep-multipliers (0 or 1/(1-p)) of the first n elements of dropout site (layer, site) under `seed`:
// site 0 attention probabilities [(clip*4+head)][q][key], 1 out-proj output [tok][512], 2 FFN hidden [tok][1024],
// 3 FFN output [tok][512].  Lets a test rebuild the exact masked forward in PyTorch.
extern "C" int mst_dropout_mask(uint64_t seed, int32_t layer, int32_t site, float p, uint64_t n, float* out, void* stream) {
    if (!out || site < 0 || site > 3 || layer < 0 || !(p >= 0.f && p < 1.f) || n >= 0xFFFFFFFFull) return fail("mst_dropout_mask: bad arguments");
    hipLaunchKernelGGL(k_dropout_mask, dim3(1024), dim3(256), 0, (hipStream_t)stream, make_drop(seed, layer, site, p), (size_t)n, out);
    HIPCHECK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------ optimizer ABI
extern "C" int64_t mst_adamw_workspace_bytes(int32_t n_tensors, const int64_t* numel) {
    if (n_tensors < 1 || !numel) return -1;
    int64_t chunks = 0;
    for (int i = 0; i < n_tensors; i++) chunks += (numel[i] + kAdamChunk - 1) / kAdamChunk;
    return (int64_t)n_tensors * (int64_t)sizeof(AdamTensor) + chunks * (int64_t)sizeof(AdamChunk) + 512 + chunks * 2 * (int64_t)sizeof(float);
}

extern "C" int mst_adamw_step(int32_t n_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
                              float* const* exp_avg_sq, const int64_t* numel, float lr, float beta1, float beta2, float eps,
                              float weight_decay, int32_t step, float* norms_dev, void* workspace_dev, int64_t workspace_bytes,
                              int32_t upload_tables, void* stream) {
    if (n_tensors < 1 || !params || !grads || !exp_avg || !exp_avg_sq || !numel || !workspace_dev || step < 1)
        return fail("mst_adamw_step: bad arguments");
    if (workspace_bytes < mst_adamw_workspace_bytes(n_tensors, numel)) return fail("mst_adamw_step: workspace too small");
    size_t nchunks = 0;
    for (int i = 0; i < n_tensors; i++) {
        if (!params[i] || !grads[i] || !exp_avg[i] || !exp_avg_sq[i] || numel[i] < 1) return fail("mst_adamw_step: null tensor %d", i);
        nchunks += (size_t)((numel[i] + kAdamChunk - 1) / kAdamChunk);
    }
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace_dev;
    const size_t tb = sizeof(AdamTensor) * (size_t)n_tensors;
    const size_t tb_al = (tb + 255) / 256 * 256;
    // The tables change only when a tensor moves (a new gradient buffer) or the workspace is new: the owner of the
    // workspace tracks that (optim.FusedAdamW keeps the pointer tuple the tables were written from) and says so; the
    // library keeps no per-workspace state (a process-global cache keyed by the workspace address would survive the
    // workspace itself and hand a later optimizer stale tables).
    if (upload_tables) {
        std::vector<AdamTensor> tt(n_tensors);
        std::vector<AdamChunk> cc;
        cc.reserve(nchunks);
        for (int i = 0; i < n_tensors; i++) {
            tt[i] = AdamTensor{params[i], grads[i], exp_avg[i], exp_avg_sq[i], (long long)numel[i]};
            for (long long s0 = 0; s0 < numel[i]; s0 += kAdamChunk) cc.push_back(AdamChunk{i, 0, s0});
        }
        HIPCHECK(hipStreamSynchronize(st));                 // a previous step may still be reading the old tables
        HIPCHECK(hipMemcpy(ws, tt.data(), tb, hipMemcpyHostToDevice));
        HIPCHECK(hipMemcpy(ws + tb_al, cc.data(), sizeof(AdamChunk) * cc.size(), hipMemcpyHostToDevice));
    }
    const float bias1 = 1.0f - (float)pow((double)beta1, (double)step);
    const float bias2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    // grad / param norms: per-chunk partials behind the tables, summed in chunk order (no float atomics: the logged norms and
    // anything derived from them are bit-reproducible)
    float* norm_part = norms_dev ? (float*)(ws + tb_al + ((sizeof(AdamChunk) * nchunks + 255) / 256) * 256) : nullptr;
    hipLaunchKernelGGL(k_adamw_multi, dim3((unsigned)nchunks), dim3(256), 0, st, (const AdamTensor*)ws, (const AdamChunk*)(ws + tb_al),
                       lr, beta1, beta2, eps, weight_decay, bias1, bias2_sqrt, norm_part);
    HIPCHECK(hipGetLastError());
    if (norms_dev) {
        hipLaunchKernelGGL(k_adamw_norms, dim3(1), dim3(256), 0, st, norm_part, (int)nchunks, norms_dev);
        HIPCHECK(hipGetLastError());
    }
    return 0;
}

// ------------------------------------------------------------------------------------------ post-sampling ABI
extern "C" int mst_recover_from_ric(const float* sample, const float* mean, const float* stdv, int32_t batch, int32_t feats,
                                    int32_t frames, int32_t joints, float* out, void* stream) {
    if (!sample || !mean || !stdv || !out || batch < 1 || frames < 1 || joints < 1) return fail("mst_recover_from_ric: bad arguments");
    if (feats < 4 + 3 * (joints - 1)) return fail("mst_recover_from_ric: %d features cannot hold %d joints", feats, joints);
    if (frames > 4096) return fail("mst_recover_from_ric: frames %d > 4096", frames);
    hipLaunchKernelGGL(k_recover_from_ric, dim3(batch), dim3(256), sizeof(float) * 5 * frames, (hipStream_t)stream, sample, mean, stdv,
                       feats, frames, joints, out);
    HIPCHECK(hipGetLastError());
    return 0;
}

extern "C" int mst_set_precise(mst_engine* e, int32_t on) {
    if (!e) return fail("mst_set_precise: null engine");
    e->precise = on != 0;
    // 2 = switched on, but weights uploaded so far lack their lo halves: the caller uploads them again (DenoiserEngine.set_precise does)
    return (e->precise && !e->lo_missing.empty()) ? 2 : 0;
}

// ------------------------------------------------------------------------------------------ debug ABI
extern "C" int mst_set_trunk_groups(mst_engine* e, int32_t on) {
    if (!e) return fail("mst_set_trunk_groups: null engine");
    e->trunk_groups = on != 0;
    if (e->trunk_err[0]) {                   // after a give-up: the counters hold whatever the abandoned launch left -- start clean
        ON_DEVICE(e->cfg.device);
        HIPCHECK(hipDeviceSynchronize());
        HIPCHECK(hipMemset(e->trunk_cnt, 0, (size_t)e->cfg.max_rows * 32 * sizeof(unsigned)));
        e->trunk_err[0] = 0;
    }
    return 0;
}
extern "C" int mst_debug_stop_after(mst_engine* e, int32_t layer, int32_t stage) {
    if (!e) return fail("mst_debug_stop_after: null engine");
    e->dbg_layer = layer;
    e->dbg_stage = stage;
    return 0;
}

extern "C" int mst_debug_copy(mst_engine* e, const char* which, void* dst_dev, uint64_t nbytes, void* stream) {
    if (!e || !which || !dst_dev) return fail("mst_debug_copy: null argument");
    std::string w(which);
    const void* src = nullptr;
    size_t cap = 0;
    if (w == "hs") {           // the stream as float32: joined from its f16 hi/lo halves on the fly
        const size_t n = nbytes / 4;
        if (nbytes > (size_t)e->M_pad * MST_D * 4) return fail("mst_debug_copy: hs holds %zu bytes", (size_t)e->M_pad * MST_D * 4);
        hipLaunchKernelGGL(k_join_stream, dim3(1024), dim3(256), 0, (hipStream_t)stream, e->hx, e->hl, (float*)dst_dev, n);
        HIPCHECK(hipGetLastError());
        return 0;
    }
    if (w == "hx") { src = e->hx; cap = (size_t)e->M_pad * MST_D * 2; }
    else if (w == "qkv") { src = e->qkv; cap = (size_t)e->M_pad * 3 * MST_D * 2; }
    else if (w == "att") { src = e->att; cap = (size_t)e->M_pad * MST_D * 2; }
    else if (w == "hid") { src = e->hid; cap = (size_t)e->M_pad * MST_FF * 2; }
    else if (w == "temb") { src = e->temb; cap = (size_t)e->temb_cap * MST_D * 4; }
    else if (w == "textproj") { src = e->textproj; cap = (size_t)e->cfg.max_rows * MST_D * 4; }
    else return fail("mst_debug_copy: unknown buffer '%s'", which);
    if (nbytes > cap) return fail("mst_debug_copy: %llu bytes requested, buffer holds %zu", (unsigned long long)nbytes, cap);
    HIPCHECK(hipMemcpyAsync(dst_dev, src, nbytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
}

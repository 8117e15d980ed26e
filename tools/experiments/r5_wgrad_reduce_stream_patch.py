p='/root/repo/diffusion-based-motion-style-transfer_amd/csrc/mst_engine.hip'
s=open(p).read()

# TrainWS
old="    float* part = nullptr;                              // split-K partial products\n"
new='''    float* part = nullptr;                              // split-K partial products
    // MST_WGRAD_STREAM=2: the ordered reduce of a wgrad's partials (and the bias-gradient sums) run on a THIRD stream while the next wgrad is
    // already multiplying: two partial buffers taking turns, ev_wg[i] = buffer i is written, ev_red[i] = buffer i has been read
    float* part2 = nullptr;
    int part_idx = 0;
    hipEvent_t ev_wg[2] = {nullptr, nullptr}, ev_red[2] = {nullptr, nullptr}, ev_rdone = nullptr;
    bool red_used[2] = {false, false};
'''
assert old in s; s=s.replace(old,new,1)

old="    CHECK(dmalloc(&t.part, t.split_cap * (size_t)3 * MST_D * MST_D));\n"
new='''    CHECK(dmalloc(&t.part, t.split_cap * (size_t)3 * MST_D * MST_D));
    if (e->wgrad_stream_on >= 2) {
        CHECK(dmalloc(&t.part2, t.split_cap * (size_t)3 * MST_D * MST_D));
        for (int i = 0; i < 2; i++) {
            HIPCHECK(hipEventCreateWithFlags(&t.ev_wg[i], hipEventDisableTiming));
            HIPCHECK(hipEventCreateWithFlags(&t.ev_red[i], hipEventDisableTiming));
        }
        HIPCHECK(hipEventCreateWithFlags(&t.ev_rdone, hipEventDisableTiming));
    }
'''
assert old in s; s=s.replace(old,new,1)

old="t.datt, t.part, t.zeros,"
new="t.datt, t.part, t.part2, t.zeros,"
assert old in s; s=s.replace(old,new,1)
old="        if (t.ev_ready) (void)hipEventDestroy(t.ev_ready);\n"
new='''        if (t.ev_ready) (void)hipEventDestroy(t.ev_ready);
        if (t.ev_rdone) (void)hipEventDestroy(t.ev_rdone);
        for (int i = 0; i < 2; i++) { if (t.ev_wg[i]) (void)hipEventDestroy(t.ev_wg[i]); if (t.ev_red[i]) (void)hipEventDestroy(t.ev_red[i]); }
'''
assert old in s; s=s.replace(old,new,1)

old='    if (const char* v = getenv("MST_WGRAD_STREAM")) e->wgrad_stream_on = atoi(v) != 0;\n'
new='    if (const char* v = getenv("MST_WGRAD_STREAM")) e->wgrad_stream_on = atoi(v);          // 0: one stream; 1: wgrads beside the dgrad chain; 2: and their reduces on a third stream\n'
assert old in s; s=s.replace(old,new,1)

# wgrad signature + body
old="static int wgrad(mst_engine* e, const f16* dY, int n_out, const f16* X, int k_in, int M, float* dW, float* db, hipStream_t st) {\n    TrainWS& t = e->tw;\n"
new="static int wgrad(mst_engine* e, const f16* dY, int n_out, const f16* X, int k_in, int M, float* dW, float* db, hipStream_t st, hipStream_t sr = nullptr) {\n    TrainWS& t = e->tw;\n    if (!sr || !t.part2) sr = st;                         // sr: the stream of the reduce / bias-sum launches (MST_WGRAD_STREAM=2: a third one)\n"
assert old in s; s=s.replace(old,new,1)

old='''    } else {
        DEpiF32 epi{nullptr, t.part, k_in, n_out};
        {
            ProfScope ps(e, FAM_WGRAD, st);
            hipLaunchKernelGGL(kern, dim3(n_out / 128, k_in / 256, nsplit), dim3(512), WgTile::SMEM, st, dY, n_out, X, k_in, M, kchunk, nelem, epi);
        }
        e->prof_now = prof_keep;
        HIPCHECK(hipGetLastError());
        hipLaunchKernelGGL(k_splitk_reduce, dim3((unsigned)((nelem / 4 + 255) / 256)), dim3(256), 0, st, t.part, nsplit, nelem, t.gscale, dW);
        HIPCHECK(hipGetLastError());
    }
    if (db) {
        const int rpb = 128, nrb = (M + rpb - 1) / rpb;      // bias gradient: row-block partials, then an ordered sum (no float atomics)
        hipLaunchKernelGGL(k_colsum_f16, dim3(n_out / 256, nrb), dim3(256), 0, st, dY, n_out, M, rpb, t.gscale, db, t.cs_part);
        HIPCHECK(hipGetLastError());
        if (nrb > 1) {
            hipLaunchKernelGGL(k_sum_partials, dim3((n_out + 63) / 64), dim3(256), 0, st, t.cs_part, nrb, n_out, t.gscale, db);
            HIPCHECK(hipGetLastError());
        }
    }
    return 0;'''
new='''    } else {
        const bool third = sr != st;
        const int pi = third ? t.part_idx : 0;
        float* const part = pi ? t.part2 : t.part;
        if (third && t.red_used[pi]) HIPCHECK(hipStreamWaitEvent(st, t.ev_red[pi], 0));      // the reduce that read this buffer two wgrads ago
        DEpiF32 epi{nullptr, part, k_in, n_out};
        {
            ProfScope ps(e, FAM_WGRAD, st);
            hipLaunchKernelGGL(kern, dim3(n_out / 128, k_in / 256, nsplit), dim3(512), WgTile::SMEM, st, dY, n_out, X, k_in, M, kchunk, nelem, epi);
        }
        e->prof_now = prof_keep;
        HIPCHECK(hipGetLastError());
        if (third) {
            HIPCHECK(hipEventRecord(t.ev_wg[pi], st));
            HIPCHECK(hipStreamWaitEvent(sr, t.ev_wg[pi], 0));
        }
        hipLaunchKernelGGL(k_splitk_reduce, dim3((unsigned)((nelem / 4 + 255) / 256)), dim3(256), 0, sr, part, nsplit, nelem, t.gscale, dW);
        HIPCHECK(hipGetLastError());
        if (third) {
            HIPCHECK(hipEventRecord(t.ev_red[pi], sr));
            t.red_used[pi] = true;
            t.part_idx ^= 1;
        }
    }
    if (db) {
        // (on sr behind this wgrad's reduce: dY is complete -- the wgrad that read it was enqueued behind the dgrad chain's event)
        hipStream_t sb = nsplit == 1 ? st : sr;
        const int rpb = 128, nrb = (M + rpb - 1) / rpb;      // bias gradient: row-block partials, then an ordered sum (no float atomics)
        hipLaunchKernelGGL(k_colsum_f16, dim3(n_out / 256, nrb), dim3(256), 0, sb, dY, n_out, M, rpb, t.gscale, db, t.cs_part);
        HIPCHECK(hipGetLastError());
        if (nrb > 1) {
            hipLaunchKernelGGL(k_sum_partials, dim3((n_out + 63) / 64), dim3(256), 0, sb, t.cs_part, nrb, n_out, t.gscale, db);
            HIPCHECK(hipGetLastError());
        }
    }
    return 0;'''
assert old in s; s=s.replace(old,new,1)

# backward: third stream
old="    hipStream_t sw = e->wgrad_stream_on ? e->aux_stream[0] : st;\n    const bool two = sw != st;\n"
new="    hipStream_t sw = e->wgrad_stream_on ? e->aux_stream[0] : st;\n    const bool two = sw != st;\n    hipStream_t sr = (two && e->wgrad_stream_on >= 2 && !small && w_.part2) ? e->aux_stream[1] : sw;      // the wgrads' reduces (wgrad())\n"
assert old in s; s=s.replace(old,new,1)
for a in ("G[6], nullptr, sw)", "G[4], G[5], sw)", "G[2], nullptr, sw)", "G[0], G[1], sw)"):
    assert a in s
    s=s.replace(a, a[:-1] + ", sr)",1)
old='''        if (two && wg) {
            HIPCHECK(hipEventRecord(w_.ev_side[par], sw));'''
new='''        if (wg && sr != sw) {                                       // the layer's reduces: everything recorded on sw below covers them
            HIPCHECK(hipEventRecord(w_.ev_rdone, sr));
            HIPCHECK(hipStreamWaitEvent(sw, w_.ev_rdone, 0));
        }
        if (two && wg) {
            HIPCHECK(hipEventRecord(w_.ev_side[par], sw));'''
assert old in s; s=s.replace(old,new,1)
open(p,'w').write(s)
print("patched")

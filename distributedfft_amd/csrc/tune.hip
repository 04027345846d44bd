// tune.hip -- the two tuners of a plan (C ABI: dfft_tune_variants, dfft_tune_placement): kernel configuration / workgroup order per
// pass and the physical backing of the buffers, both by MEASUREMENT of the plan's own passes on the caller's data.  Host code on top
// of the public entry points and the plan object; its own translation unit since round 6.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <functional>
#include <string>
#include <vector>

#include "alloc.hpp"
#include "host_common.hpp"
#include "plan.hpp"

using namespace dfft;

extern "C" {

// device time of the FFT passes (exchanges excluded) of one forward (+ inverse, if back != nullptr) execution on the given
// buffers, best of `reps` after one untimed execution
static int placement_measure(dfft_plan *p, const void *in, void *out, void *back, int reps, float *ms)
{
    auto once = [&](float *sum) -> int {
        float ph[5];
        if (p->c2c) TRY(dfft_exec_c2c(p, out, const_cast<void *>(in), DFFT_FORWARD)); else TRY(dfft_exec_r2c(p, out, in));
        int n = dfft_get_phase_times(p, ph, 5);
        float acc = 0;
        for (int i = 0; i < n; i += 2) acc += ph[i];
        if (back) {
            if (p->c2c) TRY(dfft_exec_c2c(p, back, out, DFFT_INVERSE)); else TRY(dfft_exec_c2r(p, back, out));
            n = dfft_get_phase_times(p, ph, 5);
            for (int i = 0; i < n; i += 2) acc += ph[i];
        }
        *sum = acc;
        return 0;
    };
    float best = 1e30f, cur = 0;
    TRY(once(&cur));
    for (int r = 0; r < reps; r++) { TRY(once(&cur)); best = std::min(best, cur); }
    *ms = best;
    return 0;
}

// The y / x passes of a plan on given buffers: the streaming (nontemporal) sibling of their kernel configuration where one
// exists and measures faster HERE.  Whether the hints pay depends on the pass, the layout and the physical backing of the
// buffers: on plain hipMalloc buffers they gained nothing repeatable on the 128-byte-run stores of 1024^3 fp64 (round 2), on
// tuned backings the x pass goes 5.66 -> 5.44 ms (profiles/r3_yx_variants_on_tuned_buffers.txt); at fp32 2048 points they take
// a quarter off two passes of the 8-GPU plan and double another (profiles/r3_f32_2048_tiled_variants.txt).
// One trial of the tuners below: the plan executes forward in -> o and, if b, inverse o -> b three times; t[0..5] receive the
// smallest time of every pass (fz fy fx ix iy iz, from the phase timers), *total their smallest sum.
static int tune_trial(dfft_plan *p, const void *in, void *o, void *b, float t[6], float *total)
{
    for (int k = 0; k < 6; k++) t[k] = 1e30f;
    *total = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        float ph[5], sum = 0;
        if (p->c2c) TRY(dfft_exec_c2c(p, o, const_cast<void *>(in), DFFT_FORWARD)); else TRY(dfft_exec_r2c(p, o, in));
        int n = dfft_get_phase_times(p, ph, 5);
        for (int i = 0; i < n && i < 5; i += 2) { t[i / 2] = std::min(t[i / 2], ph[i]); sum += ph[i]; }
        if (b) {
            if (p->c2c) TRY(dfft_exec_c2c(p, b, o, DFFT_INVERSE)); else TRY(dfft_exec_c2r(p, b, o));
            n = dfft_get_phase_times(p, ph, 5);
            for (int i = 0; i < n && i < 5; i += 2) { t[3 + i / 2] = std::min(t[3 + i / 2], ph[i]); sum += ph[i]; }
        }
        *total = std::min(*total, sum);
    }
    return 0;
}

// The workgroup -> tile order of every pass (PassArgs::a_fastest, xcd_swizzle: which tiles are in flight together, and on which
// XCD's L2 neighbours meet) and its kernel configuration (the variants of its length: streaming siblings, other lane mappings,
// other tile shapes), chosen by measurement on the buffers the plan will run on.  The rules of build_pipeline / dfft_init were
// fitted on the single-GPU 1024^3 plans; on the per-GPU plans of the 8-GPU grids other choices win some passes (rank 0 of 2 x 4 at
// 1024^3 fp64: y 1.02 -> 0.88 ms with a-fastest tiles; 2048^3 fp32: y 4.89 -> 3.86 ms with the streaming configuration, y^-1
// 5.59 -> 4.73 with the point-fastest store mapping on half tiles; profiles/r3_pass_orders_8gpu_plans.txt,
// r3_pass_variants_8gpu_plans.txt) -- and whether the nontemporal hints pay depends on the pass, the layout and the physical backing
// (profiles/r3_yx_variants_on_tuned_buffers.txt, r3_f32_2048_tiled_variants.txt).
// A trial sets EVERY pass to order d (to variant v) at once and reads the per-pass times from the phase timers, so each pass
// picks for itself from the same few executions: 4 order settings, then one setting per variant number that any axis length of
// the plan has.  Which trials run depends on the global grid only, never on the rank (every trial executes the plan, exchanges
// included: collective safety); the choices are each rank's own.  Passes the caller pinned (order_* / variant_*) are left alone.
static int tune_variants(dfft_plan *p, const void *in, void *o, void *b, float &best, const std::function<void(float)> &note)
{
    if (p->zyx || p->yzx) return 0;                 // the slab sequences keep their rules
    Pipeline &pl = p->pl;
    const bool shared = p->nranks == 1 && !p->opt.mirror && !p->spectral_mirror && p->c2c;      // a single rank's complex inverse runs the forward launches
    const bool single = pl.single && shared;                            // ... in the z, x, y order (three launches)
    std::vector<Launch> *vecs[6] = {&pl.fz, &pl.fy, nullptr, &pl.ix, &pl.iy, &pl.iz};
    auto launches = [&](int k, const std::function<void(Launch &)> &f) {
        if (single) { if (k < 3) f(k == 0 ? pl.sz : k == 1 ? pl.sy : pl.sx); return; }
        if (k == 2) f(pl.fx); else for (auto &L : *vecs[k]) f(L);
    };
    auto axis_of = [](int k) { return k < 3 ? k : 5 - k; };
    auto slot = [&](int k) -> int & { return k < 3 ? p->vfwd[k] : p->vinv[5 - k]; };
    auto usable = [&](int k) { return !(k >= 3 && (!b || shared)); };
    float t[6], total = 0;
    // judge a pass by both directions where they share its launches
    auto cost = [&](const float *tt, int k) { return shared && b ? tt[k] + tt[5 - k] : tt[k]; };

    // ---- orders
    {
        int keep[6];
        float tb[6] = {0, 0, 0, 0, 0, 0}, td[4][6];
        for (int k = 0; k < 6; k++) {
            keep[k] = -1;
            launches(k, [&](Launch &L) { if (keep[k] < 0) keep[k] = (L.args.a_fastest ? 1 : 0) + (L.args.xcd_swizzle ? 2 : 0); });
        }
        auto apply = [&](int k, int d) { launches(k, [&](Launch &L) { L.args.a_fastest = d & 1; L.args.xcd_swizzle = (d >> 1) & 1; }); };
        auto tunable = [&](int k) { return keep[k] >= 0 && p->opt.order[k] < 0 && usable(k); };
        for (int d = 0; d < 4; d++) {
            for (int k = 0; k < 6; k++) if (tunable(k)) apply(k, d);
            TRY(tune_trial(p, in, o, b, td[d], &total));
            note(total);
        }
        for (int k = 0; k < 6; k++) {
            if (!tunable(k)) continue;
            int pick = keep[k];
            for (int d = 0; d < 4; d++)
                if (cost(td[d], k) < 1e29f && cost(td[d], k) < 0.99f * cost(td[pick], k)) pick = d;
            apply(k, pick);
            tb[k] = td[pick][k];
        }
        (void)tb;
    }
    // ---- kernel configurations
    TRY(tune_trial(p, in, o, b, t, &total));        // the chosen orders with the rule-based configurations: the reference of the trials
    note(total);
    float cur[6];
    for (int k = 0; k < 6; k++) cur[k] = t[k];
    auto exists = [&](int k, int v) {
        PassInfo pi;
        const Axis &a = p->ax[axis_of(k)];
        return !a.bluestein && (p->prec == DFFT_F64 ? pass_info_f64((int)a.N, v, &pi) : pass_info_f32((int)a.N, v, &pi));
    };
    auto tunable = [&](int k) { return usable(k) && p->opt.variant[k] < 0 && !p->ax[axis_of(k)].bluestein && !(p->c2c == false && axis_of(k) == 0); };
    // only the role variants that the parity suite runs on every pass and address form (tests/test_gpu_variants.py); an A/B build
    // (-DDFFT_EXPERIMENTS) carries further configuration numbers that are measured by hand, never picked here
    auto validated = [&](int v) {
        if (p->prec == DFFT_F64) return (v >= 0 && v <= 3) || v == 7 || v == 8;
        return v == 0 || v == 1 || (v >= 3 && v <= 7) || v == 9 || v == 14 || v == 15;
    };
    for (int v = 0; v < 16; v++) {
        if (!validated(v)) continue;
        // does any axis length of the plan have this variant?  (global lengths: the same answer on every rank)
        bool any = false;
        for (int k = 0; k < 3; k++) any = any || exists(k, v);
        if (!any) continue;
        int old[6];
        bool tried[6];
        for (int k = 0; k < 6; k++) {
            old[k] = slot(k);
            tried[k] = tunable(k) && exists(k, v) && old[k] != v;
            if (tried[k]) slot(k) = v;
        }
        if (shared) for (int k = 0; k < 3; k++) if (tried[k]) tried[5 - k] = false;      // (vinv is not used; judged through cost())
        float tv[6];
        TRY(tune_trial(p, in, o, b, tv, &total));
        note(total);
        for (int k = 0; k < 6; k++) {
            if (!tried[k]) continue;
            float c0[6], c1[6];
            for (int q = 0; q < 6; q++) { c0[q] = cur[q]; c1[q] = tv[q]; }
            if (cost(c1, k) < 1e29f && cost(c1, k) < 0.99f * cost(c0, k)) { cur[k] = tv[k]; if (shared && b) cur[5 - k] = tv[5 - k]; }
            else slot(k) = old[k];
        }
    }
    // ---- address forms: scalar base + 32-bit lane offset (the default) against per-point 64-bit vector addresses.  The scalar
    // form saves a 64-bit multiply-add and a register pair per point and wins wherever instruction issue matters (fp32 passes
    // 4-9 %, the fp32 strided read 23 %, profiles/r3_scalar_base_addresses.txt); the y and z passes of 1024^3 fp64 on one rank
    // run 1-1.5 % faster with the old form (their accesses leave in one burst after all addresses are known)
    {
        TRY(tune_trial(p, in, o, b, t, &total));
        float tv[6];
        for (int k = 0; k < 6; k++) if (usable(k)) launches(k, [&](Launch &L) { L.args.addr64 = 1; });
        TRY(tune_trial(p, in, o, b, tv, &total));
        note(total);
        for (int k = 0; k < 6; k++) {
            if (!usable(k)) continue;
            const bool keep64 = cost(tv, k) < 1e29f && cost(tv, k) < 0.995f * cost(t, k);      // (the two forms differ by 0.5 - 1.5 % where the old one wins)
            launches(k, [&](Launch &L) { L.args.addr64 = keep64 ? 1 : 0; });
        }
    }
    TRY(placement_measure(p, in, o, b, 2, &best));
    note(best);
    return 0;
}

// physical chunk sizes (MiB) tried in turn; 0 = plain hipMalloc
static const size_t kPlacementRecipes[] = {1024, 64, 2, 256, 0, 16, 512, 128};      // ([0] is never used: the first candidate is the default recipe)

int dfft_tune_variants(dfft_plan *p, const void *in, void *out, void *back, float *report_ms, int max_report, int *n_report)
{
    TRY(check_ready(p));
    if (!in || !out) return fail(ERR_ARG, "null buffer");
    const bool was_timing = p->timing;
    TRY(dfft_enable_phase_timing(p, 1));
    int nrep = 0;
    auto note = [&](float v) { if (report_ms && nrep < max_report) report_ms[nrep] = v; nrep++; };
    float best = 0;
    int rc = placement_measure(p, in, out, back, 2, &best);
    note(best);
    if (rc == 0) rc = tune_variants(p, in, out, back, best, note);
    p->timing = was_timing;
    graphs_clear(p);
    if (n_report) *n_report = nrep < max_report ? nrep : max_report;
    return rc;
}

int dfft_tune_placement(dfft_plan *p, const void *in, int tries, void **out, void **back, float *report_ms, int max_report, int *n_report)
{
    TRY(check_ready(p));
    if (!in || !out) return fail(ERR_ARG, "null buffer");
    if (tries < 1) tries = 1;
    // COLLECTIVE on a multi-rank plan: every trial executes the plan, exchanges included.  How many candidates fit depends on the
    // rank's own buffer sizes and free memory, so a search there could run a different number of trials on different ranks and
    // strand the peers in an exchange.  Multi-rank plans therefore get NO candidate search: out / back come from the default
    // recipe (like the work area) and only the variant trials run, whose count depends on the global grid alone.
    if (p->nranks > 1) tries = 1;
    size_t isz[3];
    TRY(dfft_get_in_size(p, isz));
    const size_t in_bytes = isz[0] * isz[1] * isz[2] * (p->c2c ? p->esz : p->esz / 2);
    const size_t out_bytes = p->domainsize, work_bytes = p->worksize_d;
    const bool own_work = p->work_owned;      // a caller-provided work area stays as it is
    const bool was_timing = p->timing;
    TRY(dfft_enable_phase_timing(p, 1));
    int nrep = 0;
    auto note = [&](float v) { if (report_ms && nrep < max_report) report_ms[nrep] = v; nrep++; };
    auto room_for = [&](size_t bytes) {
        size_t free_b = 0, total_b = 0;
        return hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b > bytes + ((size_t)1 << 30);
    };
    const size_t nrec = sizeof(kPlacementRecipes) / sizeof(kPlacementRecipes[0]);
    void *o = nullptr, *b = nullptr;
    // first candidate of every buffer: the default recipe (what a caller gets from dfft_malloc(DFFT_CHUNK_DEFAULT) without a search)
    int rc = dev_alloc_default(out_bytes, &o);
    if (rc == 0 && back) rc = dev_alloc_default(in_bytes, &b);
    float best = 0;
    if (rc == 0) rc = placement_measure(p, in, o, b, 2, &best);
    note(best);
    // One buffer at a time (the passes' sensitivities to their buffers are independent): work area, out, back.  All candidates
    // of a buffer are allocated BEFORE any is measured and the losers are freed afterwards: a freed candidate's physical pages
    // would simply be handed out again to the next one, and it is the physical pages that differ.
    for (int which = 0; which < 3 && rc == 0; which++) {
        if (which == 0 && !own_work) continue;
        if (which == 2 && !back) continue;
        const size_t bytes = which == 0 ? work_bytes : which == 1 ? out_bytes : in_bytes;
        std::vector<void *> cands;
        for (int t = 1; t < tries; t++) {
            void *cand = nullptr;
            // (dev_alloc clamps a chunk larger than the buffer to the buffer's own size: still another physical allocation to try)
            const size_t rec_chunk = std::min(kPlacementRecipes[(size_t)t % nrec] << 20, (bytes + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1));
            const size_t rounded = rec_chunk ? (bytes + rec_chunk - 1) / rec_chunk * rec_chunk : bytes;
            if (!room_for(rounded) || dev_alloc(bytes, kPlacementRecipes[(size_t)t % nrec], &cand) != 0) break;      // out of memory: fewer candidates
            cands.push_back(cand);
        }
        void *keep = which == 0 ? p->work_d : which == 1 ? o : b;
        for (void *cand : cands) {
            if (rc != 0) { (void)dev_free(cand); continue; }
            if (which == 0) p->work_d = cand;
            float ms = 0;
            rc = placement_measure(p, in, which == 1 ? cand : o, which == 2 ? cand : b, 2, &ms);
            note(ms);
            if (rc == 0 && ms < best) {
                best = ms;
                (void)dev_free(keep);
                keep = cand;
            } else {
                (void)dev_free(cand);
            }
            if (which == 0) p->work_d = keep;
        }
        if (which == 1) o = keep; else if (which == 2) b = keep;
    }
    if (rc == 0) rc = tune_variants(p, in, o, b, best, note);
    p->timing = was_timing;
    graphs_clear(p);
    if (rc != 0) { (void)dev_free(o); (void)dev_free(b); return rc; }
    *out = o;
    if (back) *back = b;
    if (n_report) *n_report = nrep < max_report ? nrep : max_report;
    return 0;
}

}  // extern "C"

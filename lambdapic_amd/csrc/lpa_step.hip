// lpa_step.hip -- the no-callback stage sequence of one time step enqueued by ONE host call.
//
// The reference's Simulation.run walks its stages from Python, one facade call each
// (simulation/simulation.py:937-1122); on the device every facade call is a kernel launch of 5-40 us, and a Python
// stage loop issues a 2-D laser-target step (config C3: ~20 launches) no faster than the GPU executes it.  lpa_step
// enqueues the same launches, in the same order, from C: the host cost drops from ~15 us to ~3 us per launch and the
// step becomes GPU bound.  It calls the public entry points of this library -- nothing is re-implemented here.
//
// Slab ranks (d->slab): the guard stages also move the x faces through the slab's transport (lpa_comm.hip), from C, on
// the same stream -- where the reference brackets its intra-rank work with mpi.sync_*_start / _wait
// (simulation.py:948-960, 1043-1080, 1104-1118).  x planes are contiguous in the field arrays, so E / B guard planes are
// sent from and received into the arrays themselves (no pack / unpack launch); the J / rho guard planes leave from the
// arrays and are added by one launch; four message rounds per step (B1 | J + rho + every species' leavers | B2 + the jx
// plane of the continuity update | E2, E1 of the next step when the caller defers the E2 guards) -- or two, with
// LPA_STEP_B_EXT_*: the B sweeps advance the x guard planes themselves (exact: they read only E, whose guard planes are
// current) and the jx plane of the continuity update is formed from what travels with J.
#include "lpa_common.hpp"
#include "lpa_fold.hpp"
#include "lpa_migrate.hpp"
#include "lpa_tail.hpp"

static long plane_of(const lpa_grid *g) { return (long)(g->ny + 2 * g->ng) * (g->nz > 1 ? g->nz + 2 * g->ng : 1); }

// E (which = 1) or B (which = 2) guard planes between slabs: my interior edge planes become the neighbours' guard planes
static int slab_exchange_guards(const lpa_step_desc *d, int which, bool with_jx, void *st) {
    const lpa_step_slab *sl = d->slab;
    const lpa_grid *g = &d->grid;
    const long plane = plane_of(g), n = (long)g->ng * plane;
    double *f[3] = {which == 1 ? g->ex : g->bx, which == 1 ? g->ey : g->by, which == 1 ? g->ez : g->bz};
    lpa_face_msg m[4];
    for (int c = 0; c < 3; c++) {
        m[c].send_lo = f[c] + (long)g->ng * plane;            // interior rows [0, ng)
        m[c].send_hi = f[c] + (long)g->nx * plane;            // interior rows [nx - ng, nx)
        m[c].recv_lo = f[c];                                  // my low guard
        m[c].recv_hi = f[c] + (long)(g->nx + g->ng) * plane;  // my high guard
        m[c].n_send_lo = m[c].n_send_hi = m[c].n_recv_lo = m[c].n_recv_hi = n;
    }
    int nm = 3;
    if (with_jx) {
        // rho continuity: the backward difference of jx at my node 0 needs the LEFT neighbour's folded jx at its last node
        m[3].send_lo = nullptr; m[3].recv_hi = nullptr; m[3].n_send_lo = m[3].n_recv_hi = 0;
        m[3].send_hi = g->jx + (long)(g->ng + g->nx - 1) * plane;
        m[3].recv_lo = sl->jx_left_plane;
        m[3].n_send_hi = m[3].n_recv_lo = plane;
        nm = 4;
    }
    return lpa_comm_exchange(sl->comm, m, nm, st);
}

// (B sweeps of a slab with LPA_STEP_B_EXT_*: the x guard planes at a face with a neighbour are advanced in place)
static bool local_b(const lpa_step_desc *d) { return (d->flags & (LPA_STEP_B_EXT_LO | LPA_STEP_B_EXT_HI)) != 0; }

// (`b_part`: see lpai_fdtd -- 1 / 2 = the part of a B sweep that reads no E guard plane / the part that does)
static int step_fields(const lpa_step_desc *d, bool efield, int wrap, void *st, bool twice = false, int b_part = 0,
                       const lpai_tail *tail = nullptr) {
    const int ng = d->grid.ng > 3 ? 3 : d->grid.ng;
    const int lo = (!efield && (d->flags & LPA_STEP_B_EXT_LO)) ? ng : 0, hi = (!efield && (d->flags & LPA_STEP_B_EXT_HI)) ? ng - 1 : 0;
    return lpai_fdtd(&d->grid, d->dim, efield, 0.5 * d->dt, d->eps0, efield ? d->e_axes : d->b_axes, wrap, twice, lo, hi, b_part,
                     tail, st);
}

static int stream_fork(void *from, void *to, void *ev) {
    if (hipEventRecord((hipEvent_t)ev, (hipStream_t)from) != hipSuccess ||
        hipStreamWaitEvent((hipStream_t)to, (hipEvent_t)ev, 0) != hipSuccess) {
        lpa_set_error("lpa_step: cannot order the side stream");
        return LPA_ERR_HIP;
    }
    return LPA_OK;
}

static lpa_push_params species_params(const lpa_step_desc *d, const lpa_step_species *sp) {
    lpa_push_params pp = sp->pp;
    pp.dt = d->dt;
    pp.flags = (sp->pp.flags & ~LPA_PUSH_NO_RHO) | (d->continuity ? LPA_PUSH_NO_RHO : 0);   // (LPA_PUSH_NO_IG: per species)
    pp.absorbed = d->absorbed; pp.absorbed_count = d->absorbed_count; pp.absorbed_capacity = d->absorbed_capacity;
    pp.absorbed_spill = d->absorbed ? d->absorbed_spill : nullptr;
    return pp;
}

// every per-step device counter: the overflow-list counters of the tiled pushes and, on slab ranks, the count headers of
// the particle messages (send side; receive side of a face without a neighbour: nothing arrives).  Zeroed by ONE launch --
// the current reset's when LPA_STAGE_RESET runs in the same call, else one of their own
static int step_counters(const lpa_step_desc *d, uint32_t **w, bool overflow) {
    int n = 0;
    const lpa_step_slab *sl = d->slab;
    int32_t info[6] = {0, 0, 1, -1, -1, 0};
    if (sl && sl->comm && lpa_comm_info(sl->comm, info)) return -1;
    for (int s = 0; s < d->nspecies && s < 64; s++) {
        const lpa_step_species *sp = &d->species[s];
        if (overflow && sp->p.n > 0 && sp->t && sp->n_sorted > 0 && sp->overflow_count) w[n++] = sp->overflow_count;
        if (overflow && sp->pp.leaver_count) w[n++] = sp->pp.leaver_count;
        if (overflow && sl && sl->overlap_cols > 0 && sp->mig.overflow_edge_count) w[n++] = sp->mig.overflow_edge_count;
        if (sl && sl->comm) {
            uint32_t *h[4] = {(uint32_t *)sp->mig.s_lo, (uint32_t *)sp->mig.s_hi,
                              info[3] < 0 ? (uint32_t *)sp->mig.r_lo : nullptr, info[4] < 0 ? (uint32_t *)sp->mig.r_hi : nullptr};
            for (int k = 0; k < 4; k++)
                if (h[k]) { w[n++] = h[k]; w[n++] = h[k] + 1; }      // (the header is a 64-bit count)
        }
    }
    return n;
}

static int step_zero_counters(const lpa_step_desc *d, void *st, bool overflow = true) {
    uint32_t *w[11 * 64];
    const int n = step_counters(d, w, overflow);
    if (n < 0) return LPA_ERR_ARG;
    return n ? lpai_zero_words(w, n, st) : LPA_OK;
}

static int record(void *ev, void *st) {
    if (ev && hipEventRecord((hipEvent_t)ev, (hipStream_t)st) != hipSuccess) {
        lpa_set_error("lpa_step: hipEventRecord failed");
        return LPA_ERR_HIP;
    }
    return LPA_OK;
}

// One PART of the pushes of every species on `st`: LPA_PART_ALL (the in-line step), or -- overlapped slab steps --
// LPA_PART_EDGE: the `cols` tile columns at each x face with the species' edge overflow lists, plus everything that is not
// tile ordered (the arrival area, unsorted stores): all that can deposit into the x guard planes or leave the slab;
// LPA_PART_INTERIOR: the other tiles with the main overflow lists.
// 3-D, fuse_species: the tile-ordered part of every species in one launch (one E / B staging per tile).
enum { LOOSE_WITH_PART = 0, LOOSE_SKIP = 1, LOOSE_ONLY = 2 };

// which species' leavers were packed by their rest launch already (lpa_migrate.hpp; one lpa_step call)
struct StepCtx { bool packed[64] = {false}; };

// the pack arguments of a species' rest launch on a slab rank with leaver lists; `with_list`: this launch also walks the list
// the tiled kernel wrote (false: it only packs what it pushes itself)
static bool pack_args_of(const lpa_step_desc *d, const lpa_step_species *sp, bool with_list, lpa_pack_args *a) {
    const lpa_step_slab *sl = d->slab;
    if (!(sl && sl->comm) || !sp->pp.leavers || !sp->t || (d->flags & LPA_STEP_SEPARATE_PACK)) return false;
    const lpa_step_migrate *mg = &sp->mig;
    if (!(mg->s_lo && mg->s_hi) || sl->migrate_capacity <= 0 || !(sl->xlo < sl->xhi)) return false;
    *a = lpa_pack_args{sp->pp.leavers, sp->pp.leaver_count, sp->pp.leaver_capacity, sp->t, sl->xlo, sl->xhi, mg->s_lo, mg->s_hi,
                       sl->migrate_capacity, mg->fs, mg->surplus, with_list ? 1 : 0};
    return true;
}

static int push_part(const lpa_step_desc *d, StepCtx *ctx, int part, int cols, void *st, int loose_mode = LOOSE_WITH_PART) {
    const lpa_grid *g = &d->grid;
    const bool edge = part == LPA_PART_EDGE, loose = part != LPA_PART_INTERIOR && loose_mode != LOOSE_SKIP;
    if (loose_mode == LOOSE_ONLY) {     // the arrival areas alone (they depend on nothing the tiled kernels produce)
        for (int s = 0; s < d->nspecies; s++) {
            const lpa_step_species *sp = &d->species[s];
            if (sp->p.n == 0 || !sp->t || !sp->mig.cursor || sp->p.n <= sp->n_sorted) continue;
            lpa_push_params pp = species_params(d, sp);
            lpa_pack_args pa;       // (their own leavers are packed on the spot; the tiled kernel's list is another launch's)
            const lpa_pack_args *pk = pack_args_of(d, sp, false, &pa) ? &pa : nullptr;
            const int e = d->dim == 2 ? lpai_push_deposit_rest_2d(g, &sp->p, &pp, nullptr, nullptr, 0, sp->n_sorted, sp->p.n - sp->n_sorted,
                                                                  (const int32_t *)sp->mig.cursor, pk, st)
                                      : lpai_push_deposit_rest_3d(g, &sp->p, &pp, nullptr, nullptr, 0, sp->n_sorted, sp->p.n - sp->n_sorted,
                                                                  (const int32_t *)sp->mig.cursor, pk, st);
            if (e) return e;
        }
        return LPA_OK;
    }
    bool done[64] = {false};
    auto ovf_of = [&](const lpa_step_species *sp) { return edge ? sp->mig.overflow_edge : sp->overflow; };
    auto cnt_of = [&](const lpa_step_species *sp) { return edge ? sp->mig.overflow_edge_count : sp->overflow_count; };
    auto rest = [&](const lpa_step_species *sp, const lpa_push_params *pp, bool with_list) -> int {
        const uint32_t *list = with_list ? ovf_of(sp) : nullptr;
        const int64_t nloose = loose ? sp->p.n - sp->n_sorted : 0;
        // slab ranks: the launch also packs the step's leavers -- those it pushes itself and (the ALL / EDGE part, whose
        // tiled kernel is the one that can list any) the tiled kernel's list
        lpa_pack_args pa;
        const bool lists = part != LPA_PART_INTERIOR;
        const lpa_pack_args *pk = pack_args_of(d, sp, lists, &pa) ? &pa : nullptr;
        if (!list && nloose <= 0 && !(pk && lists)) return LPA_OK;
        const int e = d->dim == 2 ? lpai_push_deposit_rest_2d(g, &sp->p, pp, list, cnt_of(sp), sp->n_sorted, sp->n_sorted, nloose,
                                                              (const int32_t *)sp->mig.cursor, pk, st)
                                  : lpai_push_deposit_rest_3d(g, &sp->p, pp, list, cnt_of(sp), sp->n_sorted, sp->n_sorted, nloose,
                                                              (const int32_t *)sp->mig.cursor, pk, st);
        if (!e && pk && lists) ctx->packed[sp - d->species] = true;
        return e;
    };
    if (d->dim == 3 && d->fuse_species) {
        constexpr int MAXS = 4;
        const lpa_particles *p[MAXS];
        const lpa_push_params *ppp[MAXS];
        const lpa_tiling *t[MAXS];
        uint32_t *ovf[MAXS], *cnt[MAXS];
        lpa_push_params pp[MAXS];
        int idx[MAXS], n = 0;
        for (int s = 0; s < d->nspecies && n < MAXS; s++) {
            const lpa_step_species *sp = &d->species[s];
            if (sp->p.n == 0 || !sp->t || sp->n_sorted <= 0) continue;
            pp[n] = species_params(d, sp);
            p[n] = &sp->p; ppp[n] = &pp[n]; t[n] = sp->t; ovf[n] = ovf_of(sp); cnt[n] = cnt_of(sp); idx[n] = s;
            n++;
        }
        if (n > 0) {
            const lpa_step_species *first = &d->species[idx[0]];
            if (int e = record(edge ? first->mig.ev_edge_start : first->ev_start, st)) return e;
            if (int e = lpai_push_deposit_tiled_multi_part_3d(g, n, p, ppp, t, ovf, cnt, part, cols, st)) return e;
            if (int e = record(edge ? first->mig.ev_edge_stop : first->ev_stop, st)) return e;
            for (int k = 0; k < n; k++) {
                if (int e = rest(&d->species[idx[k]], &pp[k], true)) return e;
                done[idx[k]] = true;
            }
        }
    }
    for (int s = 0; s < d->nspecies; s++) {
        if (done[s]) continue;
        const lpa_step_species *sp = &d->species[s];
        if (sp->p.n == 0) continue;
        lpa_push_params pp = species_params(d, sp);
        int e;
        if (sp->t && sp->n_sorted > 0) {
            if ((e = record(edge ? sp->mig.ev_edge_start : sp->ev_start, st))) return e;
            e = d->dim == 2 ? lpa_push_deposit_tiled_part_2d(g, &sp->p, &pp, sp->t, ovf_of(sp), cnt_of(sp), part, cols, st)
                            : lpa_push_deposit_tiled_part_3d(g, &sp->p, &pp, sp->t, ovf_of(sp), cnt_of(sp), part, cols, st);
            if (e) return e;
            if ((e = record(edge ? sp->mig.ev_edge_stop : sp->ev_stop, st))) return e;
            // the overflow list and the loose particles (appended / arrived since the sort) in one launch
            if ((e = rest(sp, &pp, true))) return e;
        } else if (part == LPA_PART_INTERIOR || (loose_mode == LOOSE_SKIP && sp->t && sp->mig.cursor)) {
            continue;                             // (all of it went with the edge part / with the arrival-area launch)
        } else if (sp->t && sp->mig.cursor) {     // a slab rank's store without a tile-ordered particle: its arrival area
            if ((e = rest(sp, &pp, false))) return e;
        } else {
            e = d->dim == 2 ? lpa_push_deposit_2d(g, &sp->p, &pp, 0, sp->p.n, st)
                            : lpa_push_deposit_3d(g, &sp->p, &pp, 0, sp->p.n, st);
            if (e) return e;
        }
    }
    return LPA_OK;
}

static int step_push(const lpa_step_desc *d, StepCtx *ctx, bool counters_zeroed, void *st) {
    LPA_REQUIRE(d->nspecies <= 64, "lpa_step: more than 64 species");
    if (!counters_zeroed)
        if (int e = step_zero_counters(d, st)) return e;
    // 3-D slab ranks: the arrival areas (tens of thousands of particles one by one through global memory: ~100 us on a C5
    // slab) are pushed on the communicator's second stream BESIDE the tiled kernel instead of behind it (3.37 -> 3.27 ms per
    // step; in 2-D the arrival push is 15 us and the fork / join costs more than it hides: 0.233 -> 0.253 ms)
    bool loose_any = false;
    const bool slab = d->slab && d->slab->comm && d->dim == 3;
    for (int s = 0; slab && s < d->nspecies; s++) {
        const lpa_step_species *sp = &d->species[s];
        loose_any = loose_any || (sp->p.n > sp->n_sorted && sp->t && sp->mig.cursor && sp->n_sorted > 0);
    }
    if (!loose_any) return push_part(d, ctx, LPA_PART_ALL, 0, st);
    void *side, *ev_ready, *ev_done;
    if (int e = lpai_comm_side(d->slab->comm, &side, &ev_ready, &ev_done)) return e;
    if (hipEventRecord((hipEvent_t)ev_ready, (hipStream_t)st) != hipSuccess ||
        hipStreamWaitEvent((hipStream_t)side, (hipEvent_t)ev_ready, 0) != hipSuccess) {
        lpa_set_error("lpa_step: cannot fork the side stream");
        return LPA_ERR_HIP;
    }
    if (int e = push_part(d, ctx, LPA_PART_ALL, 0, side, LOOSE_ONLY)) return e;
    if (hipEventRecord((hipEvent_t)ev_done, (hipStream_t)side) != hipSuccess) {
        lpa_set_error("lpa_step: hipEventRecord failed");
        return LPA_ERR_HIP;
    }
    if (int e = push_part(d, ctx, LPA_PART_ALL, 0, st, LOOSE_SKIP)) return e;
    if (hipStreamWaitEvent((hipStream_t)st, (hipEvent_t)ev_done, 0) != hipSuccess) {
        lpa_set_error("lpa_step: cannot join the side stream");
        return LPA_ERR_HIP;
    }
    return LPA_OK;
}

// slab ranks, first half of the fold: leavers of every species into their face messages, then ONE exchange for the J / rho
// guard planes and all particle messages (sync_currents + sync_particles back to back, simulation.py:1043-1080)
static int slab_pack_exchange(const lpa_step_desc *d, const StepCtx *ctx, bool headers_zeroed, void *st) {
    const lpa_step_slab *sl = d->slab;
    if (!headers_zeroed)      // (LPA_STAGE_PUSH / RESET of this call did it otherwise)
        if (int e = step_zero_counters(d, st, false)) return e;
    const lpa_grid *g = &d->grid;
    LPA_REQUIRE(sl->cur_r_lo && sl->cur_r_hi && sl->migrate_capacity > 0 && sl->xlo < sl->xhi, "lpa_step: bad slab descriptor");
    const long plane = plane_of(g), n = (long)g->ng * plane;
    const long nmig = 1 + (long)LPA_MIG_NATTR * sl->migrate_capacity;
    lpa_face_msg m[5 + 64];
    double *f[4] = {g->jx, g->jy, g->jz, g->rho};
    for (int c = 0; c < 4; c++) {
        m[c].send_lo = f[c];                                   // my low guard planes -> the left neighbour's interior edge
        m[c].send_hi = f[c] + (long)(g->nx + g->ng) * plane;
        m[c].recv_lo = sl->cur_r_lo + (long)c * n;
        m[c].recv_hi = sl->cur_r_hi + (long)c * n;
        m[c].n_send_lo = m[c].n_send_hi = m[c].n_recv_lo = m[c].n_recv_hi = n;
    }
    int nm = 4;
    if (sl->rho_exchange == 2) {    // my own jx deposit on my last node plane -> the right neighbour (see k_fold_all)
        m[nm].send_lo = nullptr; m[nm].recv_hi = nullptr; m[nm].n_send_lo = m[nm].n_recv_hi = 0;
        m[nm].send_hi = g->jx + (long)(g->ng + g->nx - 1) * plane;
        m[nm].recv_lo = sl->jx_left_plane;
        m[nm].n_send_hi = m[nm].n_recv_lo = plane;
        nm++;
    }
    for (int s = 0; s < d->nspecies; s++) {
        const lpa_step_species *sp = &d->species[s];
        const lpa_step_migrate *mg = &sp->mig;
        LPA_REQUIRE(mg->s_lo && mg->s_hi && mg->r_lo && mg->r_hi && mg->cursor, "lpa_step: species without migration buffers");
        LPA_REQUIRE(sp->t && sp->n_sorted >= 0, "lpa_step: a slab rank needs tile-ordered stores (arrival area)");
        if (ctx->packed[s]) {
            // (the species' rest launch packed its leavers: lpa_migrate.hpp)
        } else if (sp->pp.leavers) {       // this step's pushes listed the leavers: no scan
            if (int e = lpai_migrate_pack_list(&sp->p, sp->t, sp->pp.leavers, sp->pp.leaver_count, sp->pp.leaver_capacity, sl->xlo,
                                               sl->xhi, mg->s_lo, mg->s_hi, sl->migrate_capacity, mg->fs, mg->surplus, st)) return e;
        } else if (int e = lpai_migrate_pack(&sp->p, sp->t, mg->edge_cols, sl->xlo, sl->xhi, mg->s_lo, mg->s_hi, sl->migrate_capacity,
                                             mg->edge_cols > 0 ? mg->fs : nullptr, mg->surplus, 0, mg->cursor, st)) return e;
        m[nm].send_lo = mg->s_lo; m[nm].send_hi = mg->s_hi; m[nm].recv_lo = mg->r_lo; m[nm].recv_hi = mg->r_hi;
        m[nm].n_send_lo = m[nm].n_send_hi = m[nm].n_recv_lo = m[nm].n_recv_hi = nmig;
        nm++;
    }
    return lpa_comm_exchange(sl->comm, m, nm, st);
}

// second half: the received planes folded in (+ the periodic fold along the local axes), the arrivals seated
static int slab_fold_unpack(const lpa_step_desc *d, void *st) {
    const lpa_step_slab *sl = d->slab;
    const lpa_grid *g = &d->grid;
    int32_t info[6];
    if (int e = lpa_comm_info(sl->comm, info)) return e;
    const bool has_left = info[3] >= 0, has_right = info[4] >= 0;
    const double *r_lo = has_left ? sl->cur_r_lo : nullptr, *r_hi = has_right ? sl->cur_r_hi : nullptr;
    const double *left_own = has_left && sl->rho_exchange == 2 ? sl->jx_left_plane : nullptr;
    if (d->nspecies >= 1 && d->nspecies <= LPA_FOLD_UNPACK_MAX_SPECIES && !(d->flags & LPA_STEP_SEPARATE_UNPACK)) {
        // the fold and every species' arrivals in one launch (they touch disjoint data)
        lpa_unpack_args u[LPA_FOLD_UNPACK_MAX_SPECIES];
        bool fused = true;
        for (int s = 0; s < d->nspecies; s++) {
            const lpa_step_species *sp = &d->species[s];
            const lpa_step_migrate *mg = &sp->mig;
            const lpa_free_slots *fs = (mg->edge_cols > 0 || sp->pp.leavers) ? mg->fs : nullptr;
            if (mg->area_capacity == 0 && !fs) fused = false;      // (nothing to seat: lpai_migrate_unpack2 returns at once)
            u[s] = lpa_unpack_args{&sp->p, sp->t, fs, sp->n_sorted, mg->area_capacity, mg->cursor, mg->r_lo, mg->r_hi};
        }
        if (fused)
            return lpai_fold_unpack(g, d->local_axes, r_lo, r_hi, left_own, u, d->nspecies, sl->migrate_capacity, sl->shift_lo,
                                    sl->shift_hi, st);
    }
    if (int e = lpai_fold_all(g, d->local_axes, r_lo, r_hi, left_own, st)) return e;
    for (int s = 0; s < d->nspecies; s++) {
        const lpa_step_species *sp = &d->species[s];
        const lpa_step_migrate *mg = &sp->mig;
        if (int e = lpai_migrate_unpack2(&sp->p, g, sp->t, (mg->edge_cols > 0 || sp->pp.leavers) ? mg->fs : nullptr, sp->n_sorted, mg->area_capacity,
                                         mg->cursor, mg->r_lo, mg->r_hi, sl->migrate_capacity, sl->shift_lo, sl->shift_hi, st))
            return e;
    }
    return LPA_OK;
}

// Overlapped slab step (lpa_step_slab.overlap_cols): the edge part, the leaver pack and the whole exchange on the
// communicator's second stream, the interior part on the caller's; joined before the fold.
static int slab_push_overlapped(const lpa_step_desc *d, StepCtx *ctx, bool counters_zeroed, void *st) {
    const lpa_step_slab *sl = d->slab;
    LPA_REQUIRE(d->nspecies <= 64, "lpa_step: more than 64 species");
    if (!counters_zeroed)
        if (int e = step_zero_counters(d, st)) return e;
    void *side, *ev_ready, *ev_done;
    if (int e = lpai_comm_side(sl->comm, &side, &ev_ready, &ev_done)) return e;
    if (hipEventRecord((hipEvent_t)ev_ready, (hipStream_t)st) != hipSuccess ||
        hipStreamWaitEvent((hipStream_t)side, (hipEvent_t)ev_ready, 0) != hipSuccess) {
        lpa_set_error("lpa_step: cannot fork the side stream");
        return LPA_ERR_HIP;
    }
    if (int e = push_part(d, ctx, LPA_PART_EDGE, sl->overlap_cols, side)) return e;
    if (int e = slab_pack_exchange(d, ctx, true, side)) return e;
    if (hipEventRecord((hipEvent_t)ev_done, (hipStream_t)side) != hipSuccess) {
        lpa_set_error("lpa_step: hipEventRecord failed");
        return LPA_ERR_HIP;
    }
    if (int e = push_part(d, ctx, LPA_PART_INTERIOR, sl->overlap_cols, st)) return e;
    if (hipStreamWaitEvent((hipStream_t)st, (hipEvent_t)ev_done, 0) != hipSuccess) {
        lpa_set_error("lpa_step: cannot join the side stream");
        return LPA_ERR_HIP;
    }
    return LPA_OK;
}

// the arguments of the slab's rho continuity update (lpa_rho_continuity)
static int slab_rho_args(const lpa_step_desc *d, int *split, const double **left) {
    const lpa_step_slab *sl = d->slab;
    int32_t info[6];
    if (int e = lpa_comm_info(sl->comm, info)) return e;
    *split = (info[3] >= 0 ? 1 : 0) | (info[4] >= 0 ? 2 : 0);
    // (rho_exchange 2: the fold left the neighbour's folded plane in my jx guard plane at node -1)
    const double *l = sl->rho_exchange == 2 ? d->grid.jx + (long)(d->grid.ng - 1) * plane_of(&d->grid) : sl->jx_left_plane;
    *left = info[3] >= 0 ? l : nullptr;
    return LPA_OK;
}

static int slab_rho(const lpa_step_desc *d, void *st) {
    int split;
    const double *left;
    if (int e = slab_rho_args(d, &split, &left)) return e;
    return lpa_rho_continuity(&d->grid, d->dt, d->local_axes, split, left, st);
}

extern "C" int lpa_step(const lpa_step_desc *d, int first_stage, int last_stage, void *stream) {
    LPA_REQUIRE(d && (d->dim == 2 || d->dim == 3) && d->dt > 0 && d->nspecies >= 0 && (d->nspecies == 0 || d->species),
                "lpa_step: bad descriptor");
    LPA_REQUIRE(first_stage >= LPA_STAGE_E1 && last_stage <= LPA_STAGE_E2 && first_stage <= last_stage,
                "lpa_step: bad stage range");
    LPA_REQUIRE(!d->continuity || !d->absorbed || (d->absorbed_count && d->absorbed_capacity > 0),
                "lpa_step: bad absorbed list");
    const lpa_grid *g = &d->grid;
    // slab rank with a transport of its own: the guard stages move the x faces too
    const bool slab = d->slab && d->slab->comm;
    LPA_REQUIRE(!slab || !(d->local_axes & 1), "lpa_step: x is split over slabs, not periodic inside one");
    LPA_REQUIRE(!slab || !d->slab->rho_exchange || d->slab->jx_left_plane, "lpa_step: jx_left_plane missing");
    LPA_REQUIRE(!local_b(d) || !(d->local_axes & 1), "lpa_step: LPA_STEP_B_EXT_* is for slabs split along x");
    LPA_REQUIRE(!slab || !local_b(d) || d->slab->rho_exchange != 1, "lpa_step: without B messages the jx plane travels with J (rho_exchange 2)");
    // the B guard stages of such a slab wrap the y / z guards of the x guard planes it advanced, too
    const int b_wrap = d->local_axes | (local_b(d) ? 8 : 0);
    bool headers_zeroed = false, counters_zeroed = false, exchanged = false;
    StepCtx ctx;
    // two ~5 us launches ride in the B sweeps' (lpa_tail.hpp): the reset that follows the first B half step and the rho
    // continuity update that precedes the second
    const bool tails = !(d->flags & LPA_STEP_SEPARATE_TAILS);
    bool reset_rode = false, rho_pending = false;
    uint32_t *rw[11 * 64 + 1];     // (the reset's counter words: alive until the B1 launch that carries them was issued)
    lpai_tail rho_tail{};
    // overlapped push + exchange: only when this call runs on through the fold and every store is tile ordered
    bool overlap = slab && d->slab->overlap_cols > 0 && first_stage <= LPA_STAGE_PUSH && last_stage >= LPA_STAGE_FOLD;
    for (int s_ = 0; overlap && s_ < d->nspecies; s_++) {
        const lpa_step_species *sp = &d->species[s_];
        overlap = sp->t && sp->mig.overflow_edge && sp->mig.overflow_edge_count && sp->overflow && sp->overflow_count;
    }
    // Overlapped steps that start at LPA_STAGE_E1 also hide the E round: the E guard planes travel on the communicator's
    // second stream, followed there by the part of the B half step that reads them (node nx - 1 and the guard planes), while
    // the caller's stream runs the rest of the B sweep, the reset and the interior tiles -- none of which touches an E guard
    // plane, B at node nx - 1 or a B guard plane (the interior tiles stage at most LPA_TILE_MARGIN + 3 nodes beyond
    // themselves and lie overlap_cols >= 1 tile columns inside).  The streams join before the fold, as for the J round.
    const bool e_side = overlap && first_stage <= LPA_STAGE_E1 && local_b(d) && !(d->flags & LPA_STEP_E_ROUND_IN_LINE) &&
                        g->nx >= 2 * g->ng + 2;
    void *side = nullptr, *ev_ready = nullptr, *ev_done = nullptr, *ev_early = nullptr;
    if (e_side)
        if (int e = lpai_comm_side(d->slab->comm, &side, &ev_ready, &ev_done, &ev_early)) return e;
    for (int stage = first_stage; stage <= last_stage; stage++) {
        int e = LPA_OK;
        switch (stage) {
        case LPA_STAGE_E1:
        case LPA_STAGE_E2: {    // update_efield(dt / 2) + sync_guard_fields(E): simulation.py:946-952, 1112-1118
            if (stage == LPA_STAGE_E2 && (d->flags & LPA_STEP_DEFER_E2)) break;     // (done by the next call's LPA_STAGE_E1)
            const bool defer = stage == LPA_STAGE_E2 && (d->flags & LPA_STEP_DEFER_E2_GUARDS);
            const bool twice = stage == LPA_STAGE_E1 && (d->flags & LPA_STEP_E1_DOUBLE);
            e = step_fields(d, true, defer ? 0 : d->local_axes, stream, twice);      // (the periodic guard wrap rides in the sweep)
            if (!e && slab && !defer) {
                if (stage == LPA_STAGE_E1 && e_side) {
                    e = stream_fork(stream, side, ev_early);
                    if (!e) e = slab_exchange_guards(d, 1, false, side);
                } else {
                    e = slab_exchange_guards(d, 1, false, stream);
                }
            }
            break;
        }
        case LPA_STAGE_B1: {    // update_bfield(dt / 2) + sync_guard_fields(B): :954-960
            // the reset of LPA_STAGE_RESET rides in this sweep's launch when it follows in the same call
            lpai_tail rt{};
            const lpai_tail *tail = nullptr;
            if (tails && last_stage >= LPA_STAGE_RESET) {
                int ns = 0;
                if (last_stage >= LPA_STAGE_PUSH) {
                    ns = step_counters(d, rw, true);
                    if (ns < 0) return LPA_ERR_ARG;
                    if (ns > 31) ns = 0;
                }
                int nw = ns;
                rt.mode = B_TAIL_RESET;
                rt.with_rho = d->continuity ? 0 : 1;
                rt.also = (!d->continuity && d->absorbed) ? d->absorbed_spill : nullptr;
                if (!d->continuity && d->absorbed_count) rw[nw++] = d->absorbed_count;
                rt.words = rw; rt.nwords = nw;
                if (lpai_tail_rides(g, d->dim, &rt)) {
                    tail = &rt;
                    reset_rode = true;
                    counters_zeroed = ns > 0;
                }
            }
            if (e_side) {       // (behind the E planes on the second stream: what reads them; here: everything else)
                e = step_fields(d, false, d->local_axes, side, false, 2);
                if (!e) e = step_fields(d, false, d->local_axes, stream, false, 1, tail);
                break;
            }
            e = step_fields(d, false, d->local_axes, stream, false, 0, tail);
            if (!e && slab && !local_b(d)) e = slab_exchange_guards(d, 2, false, stream);
            break;
        }
        case LPA_STAGE_RESET: { // current_depositor.reset(): :980-981
            if (reset_rode) {   // (LPA_STAGE_B1's launch zeroed J [, rho] and the counters; what is left: the absorbed charge)
                if (d->continuity && d->absorbed)
                    e = lpa_rho_absorbed_spill(g, d->absorbed, d->absorbed_count, d->absorbed_capacity, d->absorbed_spill, stream);
                break;
            }
            // (+ the per-step counters of the push that follows in the same call: one launch for both)
            uint32_t *w[11 * 64 + 1];
            int ns = 0;
            if (last_stage >= LPA_STAGE_PUSH) {
                ns = step_counters(d, w, true);
                if (ns < 0) return LPA_ERR_ARG;
                if (ns > 31) ns = 0;        // (many species: LPA_STAGE_PUSH zeroes them with launches of its own)
            }
            int nw = ns;
            if (d->continuity) {
                if (d->absorbed)
                    e = lpa_rho_absorbed_spill(g, d->absorbed, d->absorbed_count, d->absorbed_capacity, d->absorbed_spill, stream);
                if (!e) e = lpai_reset_step(g, 0, nullptr, w, nw, stream);
            } else {        // a real deposit contains no absorbed particle: the list and the spill array start empty
                if (d->absorbed_count) w[nw++] = d->absorbed_count;
                e = lpai_reset_step(g, 1, d->absorbed ? d->absorbed_spill : nullptr, w, nw, stream);
            }
            counters_zeroed = !e && ns > 0;
            break;
        }
        case LPA_STAGE_PUSH:    // pusher[ispec](dt, unified=True) for every species: :983-990
            if (overlap) {      // edge part + leaver pack + exchange on the second stream beside the interior part
                e = slab_push_overlapped(d, &ctx, counters_zeroed, stream);
                exchanged = !e;
            } else {
                e = step_push(d, &ctx, counters_zeroed, stream);
            }
            headers_zeroed = true;
            break;
        case LPA_STAGE_FOLD:    // sync_currents (+ sync_particles between slabs): :1043-1080, 1155-1176
            if (slab) {
                if (!exchanged) e = slab_pack_exchange(d, &ctx, headers_zeroed, stream);
                if (!e) e = slab_fold_unpack(d, stream);
                // (rho: with rho_exchange the jx plane rides with the B planes of LPA_STAGE_B2_GUARD and rho follows there)
                if (!e && d->continuity && d->slab->rho_exchange != 1) {
                    int split;
                    const double *left;
                    e = slab_rho_args(d, &split, &left);
                    rho_tail = lpai_tail{B_TAIL_RHO, 0, nullptr, nullptr, 0, d->dt, d->local_axes, split, left};
                    // (rides in the B sweep of LPA_STAGE_B2 when that follows in this call)
                    if (!e && tails && last_stage >= LPA_STAGE_B2 && lpai_tail_rides(g, d->dim, &rho_tail)) rho_pending = true;
                    else if (!e) e = slab_rho(d, stream);
                }
                break;
            }
            e = lpai_fold_all(g, d->local_axes, nullptr, nullptr, nullptr, stream);
            if (!e && d->continuity) {
                rho_tail = lpai_tail{B_TAIL_RHO, 0, nullptr, nullptr, 0, d->dt, d->local_axes, 0, nullptr};
                if (tails && last_stage >= LPA_STAGE_B2 && lpai_tail_rides(g, d->dim, &rho_tail)) rho_pending = true;
                else e = lpa_rho_continuity(g, d->dt, d->local_axes, 0, nullptr, stream);
            }
            break;
        case LPA_STAGE_B2:      // update_bfield(dt / 2): :1098 (the '_laser' stage follows: :1101)
            // a call that runs on through LPA_STAGE_B2_GUARD has no injection in between: the wrap rides in the sweep
            e = step_fields(d, false, last_stage >= LPA_STAGE_B2_GUARD ? d->local_axes : 0, stream, false, 0,
                            rho_pending ? &rho_tail : nullptr);
            rho_pending = false;
            break;
        case LPA_STAGE_B2_GUARD:    // sync_guard_fields(B): :1103-1108
            if (first_stage > LPA_STAGE_B2) e = lpa_guard_wrap(g, 2, b_wrap, stream);
            if (!e && slab && !local_b(d)) {
                e = slab_exchange_guards(d, 2, d->slab->rho_exchange == 1, stream);
                if (!e && d->continuity && d->slab->rho_exchange == 1) e = slab_rho(d, stream);
            }
            break;
        }
        if (e) return e;
    }
    return LPA_OK;
}

// lpa_step.hip -- the no-callback stage sequence of one time step enqueued by ONE host call.
//
// The reference's Simulation.run walks its stages from Python, one facade call each
// (simulation/simulation.py:937-1122); on the device every facade call is a kernel launch of 5-40 us, and a Python
// stage loop issues a 2-D laser-target step (config C3: ~20 launches) no faster than the GPU executes it.  lpa_step
// enqueues the same launches, in the same order, from C: the host cost drops from ~15 us to ~3 us per launch and the
// step becomes GPU bound.  It calls the public entry points of this library -- nothing is re-implemented here.
#include "lpa_common.hpp"

static int step_fields(const lpa_step_desc *d, bool efield, void *st) {
    const lpa_grid *g = &d->grid;
    const double h = 0.5 * d->dt;
    const lpa_cpml_axis *const *ax = efield ? d->e_axes : d->b_axes;
    const bool cpml = ax[0] != nullptr;
    if (d->dim == 2) {
        if (efield) return cpml ? lpa_fdtd_e_cpml_fused_2d(g, h, d->eps0, ax[0], ax[1], st) : lpa_fdtd_e_2d(g, h, d->eps0, st);
        return cpml ? lpa_fdtd_b_cpml_fused_2d(g, h, ax[0], ax[1], st) : lpa_fdtd_b_2d(g, h, st);
    }
    if (efield)
        return cpml ? lpa_fdtd_e_cpml_fused_3d(g, h, d->eps0, ax[0], ax[1], ax[2], st) : lpa_fdtd_e_3d(g, h, d->eps0, st);
    return cpml ? lpa_fdtd_b_cpml_fused_3d(g, h, ax[0], ax[1], ax[2], st) : lpa_fdtd_b_3d(g, h, st);
}

static lpa_push_params species_params(const lpa_step_desc *d, const lpa_step_species *sp) {
    lpa_push_params pp = sp->pp;
    pp.dt = d->dt;
    pp.flags = (sp->pp.flags & ~LPA_PUSH_NO_RHO) | (d->continuity ? LPA_PUSH_NO_RHO : 0);   // (LPA_PUSH_NO_IG: per species)
    pp.absorbed = d->absorbed; pp.absorbed_count = d->absorbed_count; pp.absorbed_capacity = d->absorbed_capacity;
    return pp;
}

// 3-D, fuse_species: the tile-ordered part of every species in one launch (one E / B staging per tile), then each
// species' overflow list and loose particles; unsorted species take the per-species path below
static int step_push_fused_3d(const lpa_step_desc *d, void *st, bool *done) {
    const lpa_grid *g = &d->grid;
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
        p[n] = &sp->p; ppp[n] = &pp[n]; t[n] = sp->t; ovf[n] = sp->overflow; cnt[n] = sp->overflow_count; idx[n] = s;
        n++;
    }
    if (n == 0) return LPA_OK;
    for (int k = 0; k < n; k++)
        if (hipMemsetAsync(cnt[k], 0, sizeof(uint32_t), (hipStream_t)st) != hipSuccess) {
            lpa_set_error("lpa_step: memset of the overflow counter failed");
            return LPA_ERR_HIP;
        }
    const lpa_step_species *first = &d->species[idx[0]];
    if (first->ev_start && hipEventRecord((hipEvent_t)first->ev_start, (hipStream_t)st) != hipSuccess) {
        lpa_set_error("lpa_step: hipEventRecord failed");
        return LPA_ERR_HIP;
    }
    if (int e = lpa_push_deposit_tiled_multi_3d(g, n, p, ppp, t, ovf, cnt, st)) return e;
    if (first->ev_stop && hipEventRecord((hipEvent_t)first->ev_stop, (hipStream_t)st) != hipSuccess) {
        lpa_set_error("lpa_step: hipEventRecord failed");
        return LPA_ERR_HIP;
    }
    for (int k = 0; k < n; k++) {
        const lpa_step_species *sp = &d->species[idx[k]];
        if (int e = lpa_push_deposit_list_3d(g, &sp->p, &pp[k], sp->overflow, sp->overflow_count, sp->n_sorted, st)) return e;
        const int64_t loose = sp->p.n - sp->n_sorted;
        if (loose > 0)
            if (int e = lpa_push_deposit_3d(g, &sp->p, &pp[k], sp->n_sorted, loose, st)) return e;
        done[idx[k]] = true;
    }
    return LPA_OK;
}

static int step_push(const lpa_step_desc *d, void *st) {
    const lpa_grid *g = &d->grid;
    bool done[64] = {false};
    if (d->dim == 3 && d->fuse_species && d->nspecies <= 64)
        if (int e = step_push_fused_3d(d, st, done)) return e;
    for (int s = 0; s < d->nspecies; s++) {
        if (s < 64 && done[s]) continue;
        const lpa_step_species *sp = &d->species[s];
        if (sp->p.n == 0) continue;
        lpa_push_params pp = species_params(d, sp);
        int e;
        if (sp->t && sp->n_sorted > 0) {
            if (hipMemsetAsync(sp->overflow_count, 0, sizeof(uint32_t), (hipStream_t)st) != hipSuccess) {
                lpa_set_error("lpa_step: memset of the overflow counter failed");
                return LPA_ERR_HIP;
            }
            if (sp->ev_start && hipEventRecord((hipEvent_t)sp->ev_start, (hipStream_t)st) != hipSuccess) {
                lpa_set_error("lpa_step: hipEventRecord failed");
                return LPA_ERR_HIP;
            }
            e = d->dim == 2 ? lpa_push_deposit_tiled_2d(g, &sp->p, &pp, sp->t, sp->overflow, sp->overflow_count, st)
                            : lpa_push_deposit_tiled_3d(g, &sp->p, &pp, sp->t, sp->overflow, sp->overflow_count, st);
            if (e) return e;
            if (sp->ev_stop && hipEventRecord((hipEvent_t)sp->ev_stop, (hipStream_t)st) != hipSuccess) {
                lpa_set_error("lpa_step: hipEventRecord failed");
                return LPA_ERR_HIP;
            }
            e = d->dim == 2 ? lpa_push_deposit_list_2d(g, &sp->p, &pp, sp->overflow, sp->overflow_count, sp->n_sorted, st)
                            : lpa_push_deposit_list_3d(g, &sp->p, &pp, sp->overflow, sp->overflow_count, sp->n_sorted, st);
            if (e) return e;
            const int64_t loose = sp->p.n - sp->n_sorted;     // appended / arrived since the sort
            if (loose > 0) {
                e = d->dim == 2 ? lpa_push_deposit_2d(g, &sp->p, &pp, sp->n_sorted, loose, st)
                                : lpa_push_deposit_3d(g, &sp->p, &pp, sp->n_sorted, loose, st);
                if (e) return e;
            }
        } else {
            e = d->dim == 2 ? lpa_push_deposit_2d(g, &sp->p, &pp, 0, sp->p.n, st)
                            : lpa_push_deposit_3d(g, &sp->p, &pp, 0, sp->p.n, st);
            if (e) return e;
        }
    }
    return LPA_OK;
}

extern "C" int lpa_step(const lpa_step_desc *d, int first_stage, int last_stage, void *stream) {
    LPA_REQUIRE(d && (d->dim == 2 || d->dim == 3) && d->dt > 0 && d->nspecies >= 0 && (d->nspecies == 0 || d->species),
                "lpa_step: bad descriptor");
    LPA_REQUIRE(first_stage >= LPA_STAGE_E1 && last_stage <= LPA_STAGE_E2 && first_stage <= last_stage,
                "lpa_step: bad stage range");
    LPA_REQUIRE(!d->continuity || !d->absorbed || (d->absorbed_count && d->absorbed_capacity > 0),
                "lpa_step: bad absorbed list");
    const lpa_grid *g = &d->grid;
    for (int stage = first_stage; stage <= last_stage; stage++) {
        int e = LPA_OK;
        switch (stage) {
        case LPA_STAGE_E1:
        case LPA_STAGE_E2:      // update_efield(dt / 2) + sync_guard_fields(E): simulation.py:946-952, 1112-1118
            e = step_fields(d, true, stream);
            if (!e) e = lpa_guard_wrap(g, 1, d->local_axes, stream);
            break;
        case LPA_STAGE_B1:      // update_bfield(dt / 2) + sync_guard_fields(B): :954-960
            e = step_fields(d, false, stream);
            if (!e) e = lpa_guard_wrap(g, 2, d->local_axes, stream);
            break;
        case LPA_STAGE_RESET:   // current_depositor.reset(): :980-981
            if (d->continuity) {
                if (d->absorbed) e = lpa_rho_absorbed(g, d->absorbed, d->absorbed_count, d->absorbed_capacity, stream);
                if (!e) e = lpa_reset_j(g, stream);
            } else {
                e = lpa_reset_current(g, stream);
                if (!e && d->absorbed_count &&
                    hipMemsetAsync(d->absorbed_count, 0, sizeof(uint32_t), (hipStream_t)stream) != hipSuccess) {
                    lpa_set_error("lpa_step: memset of the absorbed counter failed");
                    e = LPA_ERR_HIP;
                }
            }
            break;
        case LPA_STAGE_PUSH:    // pusher[ispec](dt, unified=True) for every species: :983-990
            e = step_push(d, stream);
            break;
        case LPA_STAGE_FOLD:    // sync_currents (one slab: the periodic fold): :1043, 1155-1176
            e = lpa_current_fold(g, d->local_axes, stream);
            if (!e && d->continuity) e = lpa_rho_continuity(g, d->dt, d->local_axes, 0, nullptr, stream);
            break;
        case LPA_STAGE_B2:      // update_bfield(dt / 2): :1098 (the '_laser' stage follows: :1101)
            e = step_fields(d, false, stream);
            break;
        case LPA_STAGE_B2_GUARD:    // sync_guard_fields(B): :1103-1108
            e = lpa_guard_wrap(g, 2, d->local_axes, stream);
            break;
        }
        if (e) return e;
    }
    return LPA_OK;
}

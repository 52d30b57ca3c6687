// lpa_common.hpp -- shared device helpers of the MI355X PIC kernels (gfx950, wave64, FP64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/lambdapic_amd.h"

#define LPA_C 299792458.0 /* core/utils/cutils.h:17 */

// ---- host side error plumbing -------------------------------------------------------------------
void lpa_set_error(const char *fmt, ...);
#define LPA_REQUIRE(cond, ...)                 \
    do {                                       \
        if (!(cond)) {                         \
            lpa_set_error(__VA_ARGS__);        \
            return LPA_ERR_ARG;                \
        }                                      \
    } while (0)
#define LPA_CHECK_LAUNCH(name)                                                        \
    do {                                                                              \
        hipError_t e_ = hipGetLastError();                                            \
        if (e_ != hipSuccess) {                                                       \
            lpa_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));      \
            return LPA_ERR_HIP;                                                       \
        }                                                                             \
    } while (0)

static inline int lpa_grid_ok(const lpa_grid *g, int dim, int need_j) {
    if (!g || g->nx <= 0 || g->ny <= 0 || g->ng < 1 || !(g->dx > 0) || !(g->dy > 0)) return 0;
    if (dim == 3 && (g->nz <= 0 || !(g->dz > 0))) return 0;
    if (!g->ex || !g->ey || !g->ez || !g->bx || !g->by || !g->bz) return 0;
    if (need_j && (!g->jx || !g->jy || !g->jz || !g->rho)) return 0;
    return 1;
}

// ---- internal helpers shared between the translation units (not part of the C ABI; used by lpa_step) --------------
// one E / B half step with the periodic guard wrap of the axes in `wrap` fused into the sweep (lpa_fields.hip)
// (`twice`: the E sweep applies two half steps in one pass -- see FDTD_TWICE in lpa_fields.hip; `ext_lo` / `ext_hi`: the
// B sweep of a slab split along x also advances that many x guard planes at the low / high face -- FDTD_EXT_* there)
// (`b_part`: 0 the whole sweep; 1 / 2: the B sweep in two launches -- nodes [0, nx - 1) / node nx - 1 + the guard planes:
// the part that reads no E guard plane and the part that does, FDTD_B_* there)
// (`tail`, may be NULL: a small grid-wide job the B sweep's launch carries -- lpa_tail.hpp; lpai_tail_rides says whether it can)
struct lpai_tail;
int lpai_fdtd(const lpa_grid *g, int dim, int efield, double dt, double eps0, const lpa_cpml_axis *const *ax, int wrap,
              int twice, int ext_lo, int ext_hi, int b_part, const lpai_tail *tail, void *stream);
int lpai_tail_rides(const lpa_grid *g, int dim, const lpai_tail *t);
// the global-memory remainder of a tiled push in one launch: overflow list (NULL = none) + the loose range
// [loose_first, loose_first + min(loose_count, *loose_limit)) (loose_limit: device cursor of the arrival area, may be NULL)
// (`pack`, may be NULL: the step's leaver pack rides in the launch -- lpa_migrate.hpp)
struct lpa_pack_args;
int lpai_push_deposit_rest_2d(const lpa_grid *g, const lpa_particles *p, const lpa_push_params *pp, const uint32_t *list,
                              const uint32_t *list_count, int64_t max_count, int64_t loose_first, int64_t loose_count,
                              const int32_t *loose_limit, const lpa_pack_args *pack, void *stream);
int lpai_push_deposit_rest_3d(const lpa_grid *g, const lpa_particles *p, const lpa_push_params *pp, const uint32_t *list,
                              const uint32_t *list_count, int64_t max_count, int64_t loose_first, int64_t loose_count,
                              const int32_t *loose_limit, const lpa_pack_args *pack, void *stream);
// jx jy jz (and rho) including guards = 0 and up to 32 device words = 0, one launch (lpa_step: reset + per-step counters)
// (`also`: one more array shaped like rho to zero, or NULL)
int lpai_reset_step(const lpa_grid *g, int with_rho, double *also, uint32_t *const *words, int nwords, void *stream);
// the communicator's second stream and two events (overlapped steps)
int lpai_comm_side(lpa_comm *c, void **side, void **ev_ready, void **ev_done, void **ev_early = nullptr);
// lpa_push_deposit_tiled_multi_3d restricted to the edge / interior tile columns (LPA_PART_*)
int lpai_push_deposit_tiled_multi_part_3d(const lpa_grid *g, int32_t nspecies, const lpa_particles *const *p,
                                          const lpa_push_params *const *pp, const lpa_tiling *const *t,
                                          uint32_t *const *overflow, uint32_t *const *overflow_count, int part, int edge_cols,
                                          void *stream);
// zero up to 32 device words in one launch (the per-step counters: overflow lists, message headers)
int lpai_zero_words(uint32_t *const *words, int n, void *stream);
// the current fold of a step in one launch: periodic fold along `axes` + (slab ranks) the J / rho guard planes received
// from the neighbours (r_lo / r_hi: [4][ng][plane] each, NULL = no neighbour on that face) added to the interior edge,
// consumed guards and the guard planes this rank sent away zeroed (`left_own`, may be NULL: the left neighbour's own jx
// deposit on its last node plane -- the jx guard plane at node -1 then ends up holding that neighbour's folded jx)
int lpai_fold_all(const lpa_grid *g, int axes, const double *r_lo, const double *r_hi, const double *left_own, void *stream);
// lpa_migrate_pack_edges_x / lpa_migrate_pack_x without their header memsets (zero_headers == 0: the caller zeroed them)
int lpai_migrate_pack(const lpa_particles *p, const lpa_tiling *t, int32_t edge_cols, double xlo, double xhi,
                      double *buf_lo, double *buf_hi, int64_t capacity, const lpa_free_slots *fs, int32_t *surplus,
                      int zero_headers, const int32_t *loose_limit, void *stream);
// lpa_migrate_pack_list without its header memsets
int lpai_migrate_pack_list(const lpa_particles *p, const lpa_tiling *t, const uint64_t *list, const uint32_t *list_count,
                           int64_t list_capacity, double xlo, double xhi, double *buf_lo, double *buf_hi, int64_t capacity,
                           const lpa_free_slots *fs, int32_t *surplus, void *stream);
// lpa_migrate_unpack(_tiled) of BOTH faces in one launch (fs == NULL: arrival area only)
int lpai_migrate_unpack2(const lpa_particles *p, const lpa_grid *g, const lpa_tiling *t, const lpa_free_slots *fs,
                         int64_t first_slot, int64_t area_capacity, int32_t *cursor, const double *buf_lo,
                         const double *buf_hi, int64_t capacity, double shift_lo, double shift_hi, void *stream);

// ---- kernel-side views (passed by value) ---------------------------------------------------------
struct GridV {
    int nx, ny, nz, ng, NX, NY, NZ;
    double dx, dy, dz, x0, y0, z0;
    double *ex, *ey, *ez, *bx, *by, *bz, *jx, *jy, *jz, *rho;
};

static inline GridV make_gridv(const lpa_grid *g, int dim) {
    GridV v;
    v.nx = g->nx; v.ny = g->ny; v.nz = dim == 3 ? g->nz : 1; v.ng = g->ng;
    v.NX = g->nx + 2 * g->ng; v.NY = g->ny + 2 * g->ng; v.NZ = dim == 3 ? g->nz + 2 * g->ng : 1;
    v.dx = g->dx; v.dy = g->dy; v.dz = g->dz; v.x0 = g->x0; v.y0 = g->y0; v.z0 = g->z0;
    v.ex = g->ex; v.ey = g->ey; v.ez = g->ez; v.bx = g->bx; v.by = g->by; v.bz = g->bz;
    v.jx = g->jx; v.jy = g->jy; v.jz = g->jz; v.rho = g->rho;
    return v;
}

#ifndef LPA_ABS_OFFSETS
#define LPA_ABS_OFFSETS 0
#endif

struct PartV {
    long n;
    double *x, *y, *z, *ux, *uy, *uz, *ig, *w;
    double *eb[6];
    unsigned long long *id;
    unsigned char *dead;
};

static inline PartV make_partv(const lpa_particles *p) {
    PartV v;
    v.n = p->n; v.x = p->x; v.y = p->y; v.z = p->z; v.ux = p->ux; v.uy = p->uy; v.uz = p->uz;
    v.ig = p->inv_gamma; v.w = p->w;
    for (int c = 0; c < 6; c++) v.eb[c] = p->part_eb[c];
    v.id = (unsigned long long *)p->id; v.dead = p->is_dead;
    return v;
}

static inline int lpa_part_ok(const lpa_particles *p, int dim) {
    if (!p || p->n < 0) return 0;
    if (p->n == 0) return 1;
    if (!p->x || !p->y || !p->ux || !p->uy || !p->uz || !p->inv_gamma || !p->w) return 0;
    if (dim == 3 && !p->z) return 0;
    int have = 0;
    for (int c = 0; c < 6; c++) have += p->part_eb[c] != nullptr;
    return have == 0 || have == 6;
}

// ---- device math ---------------------------------------------------------------------------------
// torus index into a padded axis of length N (conventional layout: node i -> i + ng).
__device__ __forceinline__ int torus(int c, int N) {
    if ((unsigned)c >= (unsigned)N) {
        c %= N;
        if (c < 0) c += N;
    }
    return c;
}

// floor to int: v_floor_f64 + v_cvt_i32_f64.  The hardware conversion saturates out-of-range values
// and maps NaN to 0, so garbage in gives bounded garbage out; every index derived from it is clamped
// (LDS path) or wrapped on the torus (global path) before use.
__device__ __forceinline__ int ifloor(double v) { return (int)floor(v); }

// TSC gather weights, core/pusher/unified/unified_pusher_2d.c:64-69
__device__ __forceinline__ void tsc3(double d, double g[3]) {
    double d2 = d * d;
    g[0] = 0.5 * (0.25 + d2 + d);
    g[1] = 0.75 - d2;
    g[2] = 0.5 * (0.25 + d2 - d);
}

// 1/sqrt(a) and 1/a for finite a of order 1 (a = 1 + u^2 >= 1 here): hardware seed (v_rsq_f64 /
// v_rcp_f64, ~23 bits) + two Newton steps in FMA form.  <= 1.5 ulp, against the <= 1 ulp of the
// reference's 1.0 / sqrt(a); replaces the IEEE sqrt + divide sequences (~25 VALU instructions each).
__device__ __forceinline__ double rsqrt_nr(double a) {
    double y = __builtin_amdgcn_rsq(a);
    double h = 0.5 * y;
    double e = fma(-a * y, h, 0.5);   // e = (1 - a y^2) / 2
    y = fma(y, e, y);
    h = 0.5 * y;
    e = fma(-a * y, h, 0.5);
    return fma(y, e, y);
}

__device__ __forceinline__ double rcp_nr(double a) {
    double y = __builtin_amdgcn_rcp(a);
    double e = fma(-a, y, 1.0);
    y = fma(y, e, y);
    e = fma(-a, y, 1.0);
    return fma(y, e, y);
}

// the TSC shape of a deposit end point (e0, e1, e2: cells from node 0) times `v`, added to `dst` (an array shaped like rho)
// on the torus of the padded array: what an absorbed particle had deposited into rho (current_deposit.h:7-35, S1)
__device__ __forceinline__ void spread_tsc(const GridV &g, double *dst, double e0, double e1, double e2, double v) {
    const bool d3 = g.NZ > 1;
    const double e[3] = {e0, e1, e2};
    int i1[3];
    double s[3][3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        i1[a] = ifloor(e[a] + 0.5);
        tsc3(i1[a] - e[a], s[a]);
    }
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const long r = (long)torus(i1[0] - 1 + i + g.ng, g.NX) * g.NY;
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const long rc = (r + torus(i1[1] - 1 + j + g.ng, g.NY)) * g.NZ;
            if (!d3) {
                atomicAdd(&dst[rc], v * s[0][i] * s[1][j]);
                continue;
            }
#pragma unroll
            for (int kk = 0; kk < 3; kk++)
                atomicAdd(&dst[rc + torus(i1[2] - 1 + kk + g.ng, g.NZ)], v * s[0][i] * s[1][j] * s[2][kk]);
        }
    }
}

// a particle whose advanced position left the slab along x: its slot goes on the leaver list (lpa_push_params.leavers;
// rare: one atomic each).  K = PushK / PushK3.
template <class K>
__device__ __forceinline__ void report_leaver(const K &k, double xs, long ip, int tile = -1) {
#ifdef LPA_NO_LEAVERS      // A/B build: what the check costs the tiled kernels
    return;
#endif
    if (k.leavers && (xs < k.leave_lo || xs > k.leave_hi)) {      // (NaN -- dead, absorbed -- compares false)
        const uint32_t slot = atomicAdd(k.leaver_count, 1u);
        // (the tile the slot belongs to, when the caller knows it: the pack need not search for it)
        if ((long)slot < k.leaver_cap) k.leavers[slot] = ((unsigned long long)(unsigned)(tile + 1) << 32) | (unsigned long long)(uint32_t)ip;
    }
}

// relativistic Boris rotation, core/pusher/unified/unified_pusher_2d.c:15-51
// 1 / gamma of a momentum (unified_pusher_2d.c:50).  ONE function for the value the Boris rotation leaves behind and
// for its recomputation from the stored momenta (LPA_PUSH_NO_IG): the same bits by construction.
__device__ __forceinline__ double inv_gamma_of(double ux, double uy, double uz) {
    return rsqrt_nr(fma(uz, uz, fma(uy, uy, fma(ux, ux, 1.0))));
}

__device__ __forceinline__ void boris(double &ux, double &uy, double &uz, double &ig, double Ex,
                                      double Ey, double Ez, double Bx, double By, double Bz,
                                      double efactor, double bfactor) {
    double umx = ux + efactor * Ex, umy = uy + efactor * Ey, umz = uz + efactor * Ez;
    double g = rsqrt_nr(1 + umx * umx + umy * umy + umz * umz);
    double Tx = bfactor * Bx * g, Ty = bfactor * By * g, Tz = bfactor * Bz * g;
    double upx = umx + umy * Tz - umz * Ty;
    double upy = umy + umz * Tx - umx * Tz;
    double upz = umz + umx * Ty - umy * Tx;
    double Tf = 2.0 * rcp_nr(1 + Tx * Tx + Ty * Ty + Tz * Tz);
    double Sx = Tf * Tx, Sy = Tf * Ty, Sz = Tf * Tz;
    double px = umx + upy * Sz - upz * Sy;
    double py = umy + upz * Sx - upx * Sz;
    double pz = umz + upx * Sy - upy * Sx;
    ux = px + efactor * Ex;
    uy = py + efactor * Ey;
    uz = pz + efactor * Ez;
    ig = inv_gamma_of(ux, uy, uz);
}

// One axis of the Esirkepov deposit on a 4-cell window (current/current_deposit.h:7-35,206-249).
// The reference works on the 5-cell window i0-2..i0+2 and loops [lo,hi) = [dc<0?0:1, dc>0?5:4);
// that range is always contained in the 4 cells starting at i0-2+shift with shift = (dc<0 ? 0 : 1),
// so the window is re-based there and all 4 cells are visited unconditionally (no divergence).
// Cells the reference does not visit carry exactly-zero shape values here; the two running sums
// whose residual would land on such a cell are masked by the caller through `tail_zero`.
struct AxisW {
    double S0[4], S1[4], DS[4];
    int base;       // node index of window cell 0
    bool tail_zero; // dc == 0: the 4th window cell is outside the reference's loop (all its values are 0)
};

// `inv_d` = 1/d: the reference divides by d (current_deposit.h:201-204); multiplying by the reciprocal
// differs by <= 1 ulp of the cell coordinate (<= 3e-13 cells at cell 1000) -- inside the stated
// tolerances -- and saves four FP64 divisions (~45 VALU instructions) per particle.
__device__ __forceinline__ void axis_window(AxisW &a, double r_old, double r_adv, double inv_d) {
    double o0 = r_old * inv_d, o1 = r_adv * inv_d;
    int i0 = ifloor(o0 + 0.5), i1 = ifloor(o1 + 0.5);
    int dc = i1 - i0;
    double d0 = i0 - o0, d1 = i1 - o1;
    double q0 = d0 * d0, q1 = d1 * d1;
    double lo0 = 0.5 * (q0 + d0 + 0.25), mi0 = 0.75 - q0, hi0 = 0.5 * (q0 - d0 + 0.25);
    double lo1 = 0.5 * (q1 + d1 + 0.25), mi1 = 0.75 - q1, hi1 = 0.5 * (q1 - d1 + 0.25);
    bool sh = dc >= 0;            // shift = 1
    bool up = dc > 0;
    bool valid = dc >= -1 && dc <= 1;  // |dc| > 1: the reference's calculate_S gives S1 == 0
    if (!valid) { lo1 = 0.0; mi1 = 0.0; hi1 = 0.0; }
    a.S0[0] = sh ? lo0 : 0.0; a.S0[1] = sh ? mi0 : lo0; a.S0[2] = sh ? hi0 : mi0; a.S0[3] = sh ? 0.0 : hi0;
    a.S1[0] = up ? 0.0 : lo1; a.S1[1] = up ? lo1 : mi1; a.S1[2] = up ? mi1 : hi1; a.S1[3] = up ? hi1 : 0.0;
#pragma unroll
    for (int k = 0; k < 4; k++) a.DS[k] = a.S1[k] - a.S0[k];
    a.base = i0 - 2 + (sh ? 1 : 0);
    a.tail_zero = dc == 0;
}

// 2-D Esirkepov deposit of one particle on the 4x4 window; `sink(k, l, djx, djy, djz, drho)` adds the
// contributions of window cell (k, l).  Factor grouping of the fused CPU kernel
// (current/current_deposit.h:238-241) when FAST, of the standalone one (:104-108) otherwise.
// The three quotients of the FAST grouping depend on the launch only: the tiled kernel takes them precomputed
// (DepK, IEEE divisions on the host = the same doubles) -- left in the loop they were three FP64 divisions, ~45 of the
// 436 VALU instructions per wave iteration, which the compiler did not hoist.
struct DepK { double c_rho, c_jx, c_jy; };   // q / (dx dy), q / (dy dt), q / (dx dt)

template <bool FAST, class Sink>
__device__ __forceinline__ void esirkepov_2d(const AxisW &ax, const AxisW &ay, double vz, double w,
                                             double q, double dx, double dy, double dt, Sink &&sink,
                                             const DepK *pre = nullptr) {
    double cd, fdx_, fdy_, fvz;
#ifdef LPA_NO_DEPK     // A/B build: the divisions back in the loop
    pre = nullptr;
#endif
    if (FAST && pre) {
        cd = pre->c_rho * w;
        fdx_ = pre->c_jx * w;
        fdy_ = pre->c_jy * w;
        fvz = cd * vz;
    } else if (FAST) {
        cd = (q / (dx * dy)) * w;
        fdx_ = (q / (dy * dt)) * w;
        fdy_ = (q / (dx * dt)) * w;
        fvz = cd * vz;
    } else {
        cd = q * w / (dx * dy);
        double f = cd / dt;
        fdx_ = f * dx;
        fdy_ = f * dy;
        fvz = f * dt * vz;
    }
    const double one_twelfth = 1.0 / 12.0;
    double jx_run[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        double a = ax.S0[k] + 0.5 * ax.DS[k];
        double fdx = fdx_ * ax.DS[k];
        double t12 = one_twelfth * ax.DS[k];
        double jy_run = 0.0;
        bool xz = ax.tail_zero && k == 3;
#pragma unroll
        for (int l = 0; l < 4; l++) {
            double b = ay.S0[l] + 0.5 * ay.DS[l];
            double wy = ay.DS[l] * a;
            double wz = a * b + t12 * ay.DS[l];
            jx_run[l] -= fdx * b;
            jy_run -= fdy_ * wy;
            bool yz = ay.tail_zero && l == 3;
            sink(k, l, xz ? 0.0 : jx_run[l], yz ? 0.0 : jy_run, fvz * wz, cd * ax.S1[k] * ay.S1[l]);
        }
    }
}

// ---- wave-level reduce-scatter of 64 FP64 values across the 64 lanes of a wave -----------------------
// Every lane contributes v[0..63]; afterwards lane L holds sum over lanes of v[L].  Six butterfly
// stages, each halving the number of live values: lane-xor 32 and 16 with the CDNA4 half-wave / row
// swap instructions (v_permlane32_swap / v_permlane16_swap: one instruction exchanges the halves of
// two registers), lane-xor 8, 7, 2, 1 with DPP moves (row_ror:8, row_half_mirror, quad_perm).  Pure
// VALU: nothing goes through the LDS crossbar, which is the unit this reduction is meant to unload.
typedef unsigned lpa_u2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double wr_red32(double a, double b) {  // lanes <32 keep a, lanes >=32 keep b
    lpa_u2 l = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    lpa_u2 h = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    return __hiloint2double((int)h.x, (int)l.x) + __hiloint2double((int)h.y, (int)l.y);
}

__device__ __forceinline__ double wr_red16(double a, double b) {  // even 16-lane rows keep a, odd keep b
    lpa_u2 l = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    lpa_u2 h = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    return __hiloint2double((int)h.x, (int)l.x) + __hiloint2double((int)h.y, (int)l.y);
}

template <int CTRL>
__device__ __forceinline__ double wr_dpp(double v) {
    int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
    int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

// lanes with bit BIT clear keep a, the others keep b; the partner (CTRL) supplies its copy of the kept one
template <int CTRL, int BIT>
__device__ __forceinline__ double wr_redd(double a, double b, int lane) {
    bool up = (lane >> BIT) & 1;
    double keep = up ? b : a, send = up ? a : b;
    return keep + wr_dpp<CTRL>(send);
}

// last four stages: 16 values (index c = bits 3..0) -> this lane's value (c == lane & 15)
__device__ __forceinline__ double wr_finish16(const double s16[16], int lane) {
    double s8[8], s4[4], s2[2];
#pragma unroll
    for (int c = 0; c < 8; c++) s8[c] = wr_redd<0x128, 3>(s16[c], s16[c + 8], lane);  // row_ror:8
#pragma unroll
    for (int c = 0; c < 4; c++) s4[c] = wr_redd<0x141, 2>(s8[c], s8[c + 4], lane);    // row_half_mirror
#pragma unroll
    for (int c = 0; c < 2; c++) s2[c] = wr_redd<0x4E, 1>(s4[c], s4[c + 2], lane);     // quad_perm 2,3,0,1
    return wr_redd<0xB1, 0>(s2[0], s2[1], lane);                                      // quad_perm 1,0,3,2
}

// block-wide sum of `v` into *out with one atomic per block (blockDim.x multiple of 64, <= 1024)
__device__ __forceinline__ void block_atomic_sum(double v, double *out) {
    __shared__ double red[16];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();  // protect `red` against a previous use
    if (lane == 0) red[wv] = v;
    __syncthreads();
    if (wv == 0) {
        int nw = (blockDim.x + 63) >> 6;
        double s = lane < nw ? red[lane] : 0.0;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
        if (lane == 0) atomicAdd(out, s);
    }
}

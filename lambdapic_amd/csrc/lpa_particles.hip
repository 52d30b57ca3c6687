// lpa_particles.hip -- the fused particle kernel (half push, TSC gather, Boris, half push, Esirkepov
// deposit) in two forms, plus the split kernels and the particle diagnostics.
//
//  K1-global  any particle order; gathers E/B from global memory (L1/L2) and deposits with FP64
//             global atomics on the torus.  Correct for every input; used for loose particles, for
//             the overflow list of the tiled kernel and as the drop-in for unsorted patch arrays.
//  K1-tiled   particles binned by 8x32-cell tiles (lpa_sort.hip).  One workgroup per work block of
//             a tile: the tile's E/B (+4-node halo) are staged once in LDS, particles stream through
//             in SoA order (fully coalesced 512-B wave loads), J/rho accumulate in an LDS tile with
//             ds_add_f64 and are flushed with one global atomic per touched cell.  The product build has ONE
//             deposit path: per-lane LDS atomics, conflict free on STRIPED stores (the lanes of a half-wave sit
//             in consecutive y-cells), cell-crossers deposited in a dense second pass.  Three measured-slower
//             alternatives (DESIGN_HISTORY.md) are compiled only with -DLPA_K1_VARIANTS=1 into
//             csrc/build/liblambdapic_amd_variants.so: the wave reduce-scatter deposit of CELL_MAJOR stores, the
//             in-kernel re-seating (slot classes) and the cooperative deposit of PADDED stores.
//
// Restates unified_boris_pusher_cpu_2d (core/pusher/unified/unified_pusher_2d.c:157-365).
#include "lpa_common.hpp"
#include "lpa_migrate.hpp"
#ifndef LPA_NT_PARTICLES_2D
#define LPA_NT_PARTICLES_2D 0
#endif

struct PushK {
    double dt, q, m;
    double efactor, bfactor, cdt_half;
    int wrap;
    double lo[3], hi[3], alo[3], ahi[3];
    DepK dep;      // deposit factors of the fused (FAST) grouping; valid when the grid was given to make_pushk
    // LPA_PUSH_NO_RHO: rho is not deposited (lpa_rho_continuity advances it); `absorbed` (optional, any mode): the
    // particles absorbed at an open face are reported there (lpa_rho_absorbed), see lpa_particles3d.hip
    int flags;
    double *absorbed;
    uint32_t *absorbed_count;
    long absorbed_cap;
    double *absorbed_spill;
    // slab ranks (optional): slots of the particles that now belong to a neighbour slab (lpa_push_params.leavers)
    unsigned long long *leavers;
    uint32_t *leaver_count;
    long leaver_cap;
    double leave_lo, leave_hi;
};

#ifndef LPA_K1_VARIANTS
#define LPA_K1_VARIANTS 0
#endif

__device__ __forceinline__ void report_absorbed_2d(const GridV &g, const PushK &k, double o1x, double o1y, double cd) {
    const uint32_t slot = atomicAdd(k.absorbed_count, 1u);
    if ((long)slot < k.absorbed_cap) {
        double *e = k.absorbed + 4 * (long)slot;
        e[0] = o1x; e[1] = o1y; e[2] = 0.0; e[3] = cd;
    } else if (k.absorbed_spill) {      // the list is full: what the particle had deposited goes to the spill array, node by node
        spread_tsc(g, k.absorbed_spill, o1x, o1y, 0.0, cd);
    }
}

static PushK make_pushk(const lpa_push_params *pp, const lpa_grid *g = nullptr) {
    PushK k;
    k.leavers = (unsigned long long *)pp->leavers; k.leaver_count = pp->leaver_count; k.leaver_cap = (long)pp->leaver_capacity;
    k.leave_lo = pp->leave_lo; k.leave_hi = pp->leave_hi;
    k.dep.c_rho = k.dep.c_jx = k.dep.c_jy = 0.0;
    if (g) {   // current/current_deposit.h:238-241: (q / (dx dy)) w, (q / (dy dt)) w, (q / (dx dt)) w
        k.dep.c_rho = pp->q / (g->dx * g->dy);
        k.dep.c_jx = pp->q / (g->dy * pp->dt);
        k.dep.c_jy = pp->q / (g->dx * pp->dt);
    }
    k.dt = pp->dt; k.q = pp->q; k.m = pp->m;
    k.efactor = pp->q * pp->dt / (2 * pp->m * LPA_C);  // unified_pusher_2d.c:246-248
    k.bfactor = pp->q * pp->dt / (2 * pp->m);
    k.cdt_half = LPA_C * 0.5 * pp->dt;
    k.wrap = pp->wrap;
    k.flags = pp->flags;
    k.absorbed = pp->absorbed; k.absorbed_count = pp->absorbed_count; k.absorbed_cap = (long)pp->absorbed_capacity;
    k.absorbed_spill = pp->absorbed_spill;
    for (int a = 0; a < 3; a++) {
        k.lo[a] = pp->lo[a]; k.hi[a] = pp->hi[a];
        k.alo[a] = pp->alo[a]; k.ahi[a] = pp->ahi[a];
    }
    return k;
}

// periodic fold and / or absorption of the advanced position (what sync_particles does after the
// deposit: core/patch/sync_particles_2d.c:168-202); an absorbed particle becomes a dead slot (NaN)
__device__ __forceinline__ bool finish_position_2d(double &x, double &y, const PushK &k) {
    double L;
    if (k.wrap & 1) { L = k.hi[0] - k.lo[0]; if (x > k.hi[0]) x -= L; if (x < k.lo[0]) x += L; }
    if (k.wrap & 2) { L = k.hi[1] - k.lo[1]; if (y > k.hi[1]) y -= L; if (y < k.lo[1]) y += L; }
    bool dead = ((k.wrap & LPA_ABSORB_X) && (x < k.alo[0] || x > k.ahi[0])) ||
                ((k.wrap & (LPA_ABSORB_X << 1)) && (y < k.alo[1] || y > k.ahi[1]));
    if (dead) { x = __longlong_as_double(0x7ff8000000000000ll); y = x; }
    return dead;
}

// periodic fold of a coordinate into [lo, hi] (sync_particles_2d.c:168-182 with a self neighbour)
__device__ __forceinline__ double fold_coord(double v, double lo, double hi) {
    double L = hi - lo;
    if (v > hi) v -= L;
    if (v < lo) v += L;
    return v;
}

// ---- global-memory gather (torus indices) -----------------------------------------------------------
struct GIdx2 { int r[3]; int c[3]; };  // 3 wrapped row offsets (already * NY) and 3 wrapped columns

__device__ __forceinline__ void gidx(GIdx2 &o, int ix, int iy, const GridV &g) {
#pragma unroll
    for (int a = 0; a < 3; a++) {
        o.r[a] = torus(ix - 1 + a + g.ng, g.NX) * g.NY;
        o.c[a] = torus(iy - 1 + a + g.ng, g.NY);
    }
}

__device__ __forceinline__ double gather9_g(const double *__restrict__ f, const GIdx2 &o,
                                            const double fx[3], const double fy[3]) {
    return fy[0] * (fx[0] * f[o.r[0] + o.c[0]] + fx[1] * f[o.r[1] + o.c[0]] + fx[2] * f[o.r[2] + o.c[0]]) +
           fy[1] * (fx[0] * f[o.r[0] + o.c[1]] + fx[1] * f[o.r[1] + o.c[1]] + fx[2] * f[o.r[2] + o.c[1]]) +
           fy[2] * (fx[0] * f[o.r[0] + o.c[2]] + fx[1] * f[o.r[1] + o.c[2]] + fx[2] * f[o.r[2] + o.c[2]]);
}

__device__ __forceinline__ void gather_global_2d(const GridV &g, double xo, double yo, double eb[6]) {
    // xo, yo: position in cells relative to node 0 (unified_pusher_2d.c:112-135)
    int ix1 = ifloor(xo + 0.5), ix2 = ifloor(xo), iy1 = ifloor(yo + 0.5), iy2 = ifloor(yo);
    double gx[3], hx[3], gy[3], hy[3];
    tsc3(ix1 - xo, gx);
    tsc3(ix2 - xo + 0.5, hx);
    tsc3(iy1 - yo, gy);
    tsc3(iy2 - yo + 0.5, hy);
    GIdx2 a;
    gidx(a, ix2, iy1, g);
    eb[0] = gather9_g(g.ex, a, hx, gy);
    eb[4] = gather9_g(g.by, a, hx, gy);
    gidx(a, ix1, iy2, g);
    eb[1] = gather9_g(g.ey, a, gx, hy);
    eb[3] = gather9_g(g.bx, a, gx, hy);
    gidx(a, ix1, iy1, g);
    eb[2] = gather9_g(g.ez, a, gx, gy);
    gidx(a, ix2, iy2, g);
    eb[5] = gather9_g(g.bz, a, hx, hy);
}

template <bool FAST>
__device__ __forceinline__ void deposit_global_2d(const GridV &g, double x, double y, double ux,
                                                  double uy, double uz, double ig, double w, double q,
                                                  double dt, const DepK *pre = nullptr, bool rho = true) {
    double vx = ux * LPA_C * ig, vy = uy * LPA_C * ig, vz = uz * LPA_C * ig;
    AxisW ax, ay;
    axis_window(ax, x - vx * 0.5 * dt - g.x0, x + vx * 0.5 * dt - g.x0, 1.0 / g.dx);
    axis_window(ay, y - vy * 0.5 * dt - g.y0, y + vy * 0.5 * dt - g.y0, 1.0 / g.dy);
    int rows[4], cols[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        rows[k] = torus(ax.base + k + g.ng, g.NX) * g.NY;
        cols[k] = torus(ay.base + k + g.ng, g.NY);
    }
    esirkepov_2d<FAST>(ax, ay, vz, w, q, g.dx, g.dy, dt,
                       [&](int k, int l, double djx, double djy, double djz, double drho) {
                           int idx = rows[k] + cols[l];
                           if (djx != 0.0) atomicAdd(&g.jx[idx], djx);
                           if (djy != 0.0) atomicAdd(&g.jy[idx], djy);
                           if (djz != 0.0) atomicAdd(&g.jz[idx], djz);
                           if (rho && drho != 0.0) atomicAdd(&g.rho[idx], drho);
                       }, pre);
}

// the whole per-particle update on global memory
// (returns the particle's new x; NaN: a dead slot)
__device__ __forceinline__ double update_global_2d(const GridV &g, const PartV &p, const PushK &k, long ip) {
    double x = p.x[ip], y = p.y[ip];
    if ((p.dead && p.dead[ip]) || isnan(x) || isnan(y)) return __longlong_as_double(0x7ff8000000000000ll);
    double ux = p.ux[ip], uy = p.uy[ip], uz = p.uz[ip], w = p.w[ip];
    const bool noig = k.flags & LPA_PUSH_NO_IG;      // the store's inv_gamma is neither read nor written
    double ig = noig ? inv_gamma_of(ux, uy, uz) : p.ig[ip];
    x += k.cdt_half * ig * ux;
    y += k.cdt_half * ig * uy;
    double eb[6];
    gather_global_2d(g, (x - g.x0) * (1.0 / g.dx), (y - g.y0) * (1.0 / g.dy), eb);
    if (p.eb[0]) {
#pragma unroll
        for (int c = 0; c < 6; c++) p.eb[c][ip] = eb[c];
    }
    boris(ux, uy, uz, ig, eb[0], eb[1], eb[2], eb[3], eb[4], eb[5], k.efactor, k.bfactor);
    x += k.cdt_half * ig * ux;
    y += k.cdt_half * ig * uy;
    deposit_global_2d<true>(g, x, y, ux, uy, uz, ig, w, k.q, k.dt, &k.dep, !(k.flags & LPA_PUSH_NO_RHO));
    const double xe = x, ye = y;
    if (finish_position_2d(x, y, k) && k.absorbed)     // deposit end point = r + v dt / 2, v = u c / gamma
        report_absorbed_2d(g, k, (xe + ux * LPA_C * ig * 0.5 * k.dt - g.x0) * (1.0 / g.dx),
                           (ye + uy * LPA_C * ig * 0.5 * k.dt - g.y0) * (1.0 / g.dy), k.dep.c_rho * w);
    report_leaver(k, x, ip);
    p.x[ip] = x; p.y[ip] = y;
    p.ux[ip] = ux; p.uy[ip] = uy; p.uz[ip] = uz;
    if (!noig) p.ig[ip] = ig;
    return x;
}

__global__ void __launch_bounds__(256) k_push_deposit_global_2d(GridV g, PartV p, PushK k, long first,
                                                                long count) {
    long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    update_global_2d(g, p, k, first + t);
}

__global__ void __launch_bounds__(256) k_push_deposit_list_2d(GridV g, PartV p, PushK k,
                                                              const uint32_t *__restrict__ list,
                                                              const uint32_t *__restrict__ list_count) {
    long n = *list_count;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long)gridDim.x * blockDim.x)
        update_global_2d(g, p, k, list[t]);
}

// what the tiled kernel leaves to the global-memory path, in ONE launch (lpa_step): the overflow list (particles outside
// their tile's staged region) and the loose range behind the tile-ordered part -- on a slab rank the arrival area, of which
// only the first *loose_limit slots have ever been handed out (the unpack kernels' cursor); the rest is NaN from the sort
__global__ void __launch_bounds__(256) k_push_deposit_rest_2d(GridV g, PartV p, PushK k,
                                                              const uint32_t *__restrict__ list,
                                                              const uint32_t *__restrict__ list_count, long loose_first,
                                                              long loose_count, const int32_t *__restrict__ loose_limit) {
    const long stride = (long)gridDim.x * blockDim.x, t0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (list) {
        const long n = *list_count;
        for (long t = t0; t < n; t += stride) update_global_2d(g, p, k, list[t]);
    }
    long m = loose_count;
    if (loose_limit && (long)*loose_limit < m) m = *loose_limit;
    for (long t = t0; t < m; t += stride) update_global_2d(g, p, k, loose_first + t);
}

// the same launch with the step's leaver pack in it (lpa_migrate.hpp: rest_pack_body)
__global__ void __launch_bounds__(256) k_push_deposit_rest_pack_2d(GridV g, PartV p, PushK k,
                                                                   const uint32_t *__restrict__ list,
                                                                   const uint32_t *__restrict__ list_count, long loose_first,
                                                                   long loose_count, const int32_t *__restrict__ loose_limit,
                                                                   PackArgsV pk) {
    PushK kk = k;
    kk.leavers = nullptr;       // (what these particles do is seen right here: no list entry)
    rest_pack_body(p, list, list_count, loose_first, loose_count, loose_limit, pk,
                   [&](long ip) { return update_global_2d(g, p, kk, ip); });
}

// =====================================================================================================
// K1-tiled
// =====================================================================================================
constexpr int TX = LPA_TILE_X, TY = LPA_TILE_Y;  // 8 x 32 cells
constexpr int HALO = LPA_TILE_MARGIN + 3;  // margin + (1 cell of motion + 2 cells of stencil), see DESIGN.md
constexpr int RWX = TX + 2 * HALO;         // 16: staged region, nodes along x
constexpr int RWY = TY + 2 * HALO;         // 40: staged region, nodes along y
// LDS image of E/B: one row of RS doubles per x-node.  ey and bx are gathered with the same stencil
// (gx, hy), ex and by with (hx, gy): each pair is stored interleaved, so one 16-byte ds_read_b128 serves
// both -- measured (tools/ubench/lds_atomic.hip) a ds_read_b128 wave instruction costs 1.16 x a
// ds_read_b64, i.e. 0.58 x per double; the address must be 16-byte aligned (8-byte aligned: 13 x slower).
// Row = [ey bx] x CS | [ex by] x CS | ez x CS | bz x CS.  RS is a multiple of 32 doubles, so the bank of
// a node depends on its column only: the 32 lanes of a half-wave (32 different y-cells, but two or
// three different x-rows because the nearest node depends on the sub-cell position) never collide.
// With one padded array per component (stride 41) 30 % of the LDS cycles of this kernel were bank
// conflicts of the gather.
constexpr int CS = RWY;                    // 42
constexpr int RS = 256;                    // 6 * 42 = 252, padded to 8 * 32
constexpr int EB_PAIR_A = 0;               // [ey, bx]
constexpr int EB_PAIR_B = 2 * CS;          // [ex, by]
constexpr int EB_EZ = 4 * CS, EB_BZ = 5 * CS;
static_assert((EB_PAIR_B * 8) % 16 == 0 && (RS * 8) % 16 == 0 && 6 * CS <= RS, "E/B image layout");
constexpr int RSZ = RWX * RS;              // 4096 doubles = 32 KiB
// LDS row stride of the J/rho accumulators: a multiple of 32 doubles, so that consecutive y-cells map to
// consecutive bank pairs ACROSS a row wrap too -- a half-wave whose lanes run from the end of one grid
// row into the start of the next still touches 32 different bank pairs.
constexpr int RSJ = 64;
constexpr int RSZJ = RWX * RSJ;
constexpr int K1_THREADS = 512;

// gather from the LDS copy; (lx, ly) = local index of the stencil centre, guaranteed inside by the
// margin test (and clamped against non-finite input)
// The loads are volatile to keep them as nine ds_read_b64 (2 LDS cycles each, 64 banks): merged into
// ds_read2_b64 the same bytes cost 8 cycles per pair (MI355X_MICROARCH.md, LDS table).
__device__ __forceinline__ double gather9_l(const double *f, int lx, int ly, const double fx[3],
                                            const double fy[3]) {
    // explicit LDS address space: a volatile access through a generic pointer becomes flat_load
    typedef const volatile __attribute__((address_space(3))) double *lds_ptr;
    lds_ptr c = (lds_ptr)(f + lx * RS + ly);
    double m0 = c[-RS - 1], m1 = c[-1], m2 = c[RS - 1];
    double z0 = c[-RS], z1 = c[0], z2 = c[RS];
    double p0 = c[-RS + 1], p1 = c[1], p2 = c[RS + 1];
    return fy[0] * (fx[0] * m0 + fx[1] * m1 + fx[2] * m2) +
           fy[1] * (fx[0] * z0 + fx[1] * z1 + fx[2] * z2) +
           fy[2] * (fx[0] * p0 + fx[1] * p1 + fx[2] * p2);
}

// the same stencil on an interleaved pair: nine ds_read_b128, two results
typedef double lpa_d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gather9_pair_l(const double *f, int lx, int ly, const double fx[3],
                                               const double fy[3], double &r0, double &r1) {
    typedef const volatile __attribute__((address_space(3))) lpa_d2 *lds_ptr2;
    lds_ptr2 c = (lds_ptr2)(f + lx * RS + 2 * ly);
    constexpr int R2 = RS / 2;
    lpa_d2 m0 = c[-R2 - 1], m1 = c[-1], m2 = c[R2 - 1];
    lpa_d2 z0 = c[-R2], z1 = c[0], z2 = c[R2];
    lpa_d2 p0 = c[-R2 + 1], p1 = c[1], p2 = c[R2 + 1];
    r0 = fy[0] * (fx[0] * m0.x + fx[1] * m1.x + fx[2] * m2.x) +
         fy[1] * (fx[0] * z0.x + fx[1] * z1.x + fx[2] * z2.x) +
         fy[2] * (fx[0] * p0.x + fx[1] * p1.x + fx[2] * p2.x);
    r1 = fy[0] * (fx[0] * m0.y + fx[1] * m1.y + fx[2] * m2.y) +
         fy[1] * (fx[0] * z0.y + fx[1] * z1.y + fx[2] * z2.y) +
         fy[2] * (fx[0] * p0.y + fx[1] * p1.y + fx[2] * p2.y);
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// minimum number of lanes sharing one deposit window for the wave reduction to pay; smaller groups go
// straight to LDS atomics
[[maybe_unused]] constexpr int WR_MIN_GROUP = 12;
[[maybe_unused]] constexpr int WR_MAX_ROUNDS = 2;

// Window row 3 / column 3 only receive something from a particle that changed cell during the step (a few
// per cent of the lanes), yet a wave has to issue those 28 ds_add_f64 whenever ANY of its 64 lanes did.
// With DEFER such lanes park their advanced state (7 doubles) in a scratch store (the idle half of the
// ping-pong sort buffers) and deposit nothing in the main loop, which then handles only particles that
// stay in their cell: 3 x 3 window, old shape = the gather weights, no window re-basing.  The workgroup
// deposits the parked particles afterwards on the general 4 x 4 window with every lane busy.
struct Scratch7 { double *a[7]; };

#ifndef LPA_SKIP_NULL_RUN
#define LPA_SKIP_NULL_RUN 1
#endif
// RELOC -- the in-kernel cell-index sort.  What makes the deposit fast is that the 16 lanes the LDS serves per pass
// touch 16 different bank pairs, i.e. sit in 16 different y-cells mod 16 (a freshly sorted stripe); every lane that
// has drifted to another y-cell since the sort costs its group a second pass on all 30 atomics (K1 is 1.65 ms right
// after a sort, 1.95 ms averaged over 20 steps on C2).  So every slot of the ordered store carries the y-class
// (ly mod 32) it was sorted for (`cls`, one byte per slot, written by the first push after a sort), and every step
// the particles whose NEXT gather cell has another class are re-seated: they are parked like the cell-crossers
// (state + id), their slots form per-class pools, and each takes a slot of its new class from the pool -- a
// permutation among the movers of one work block, ~2.4 % of the particles per step at 1 keV, no holes created.  A
// mover that finds no slot of its class takes any left-over one and tries again next step (it is a mover as long
// as its class differs from its slot's).  Row (x) drift needs no repair: the LDS row stride is a multiple of the
// bank count.
struct Reloc {
    uint16_t *cls;           // [n_sorted] y-class of every slot (16-bit: a byte store would alias every array)
    uint32_t *aux_slot;      // [n_sorted] scratch: slot of a parked particle
    uint32_t *aux_info;      // [n_sorted] scratch: flags | slot class << 8 | new class << 16
    unsigned long long *aux_id;   // [n_sorted] scratch: id of a parked mover
    uint32_t *stats;         // optional [4]: parked, movers, movers that left their slot, movers without a slot of their class
    int init;                // first push after a sort: the classes are (re)written, not read (host side: selects the instantiation)
};
[[maybe_unused]] constexpr int RL_CLASSES = 32, RL_DEPTH = 16;
[[maybe_unused]] constexpr uint32_t RL_DEP = 1u, RL_MOV = 2u;

// RELOC_MODE: 0 = off, 1 = on, 2 = on and this is the first push after a sort (the classes are written, not read:
// its own instantiation, the 16-bit store in the loop costs the steady-state kernel 44 spilled VGPRs otherwise)
// COOP -- cooperative deposit on LPA_ORDER_PADDED stores.  In the full stripes a slot's position tells its cell, so
// lane l of a half-wave OWNS cell (lx, l) of one tile row whether or not its slot holds a particle of that cell.  The
// three columns of a particle's 3 x 3 window are the own columns of lanes l - 1, l, l + 1: every lane hands its outer
// columns to the neighbours (DPP wave_shr:1 / wave_shl:1), adds what it receives to its own middle column and issues
// ONE ds_add_f64 per row and quantity -- 11 per particle instead of 30, all to consecutive addresses (conflict free
// whatever the particles did since the sort).  Particles that are not in their slot's cell any more, and the
// cell-crossers, contribute zeros here and are parked for the second pass; the two lanes at the ends of a 32-cell row
// deposit their outward column themselves (11 partially filled atomics per iteration).
struct Coop {
    const int32_t *tile_off;    // [ntiles + 1]
    const int32_t *pad_ranks;   // [ntiles] full stripes per tile
};

// Template parameters of the product build: WRITE_EB (store the gathered E / B per particle), DEFER (second pass for
// the cell-crossers), RHO (deposit rho; false = LPA_PUSH_NO_RHO).  WAVE_REDUCE / RELOC_MODE / COOP select the variant
// deposit paths and are false / 0 unless LPA_K1_VARIANTS (their code is fenced by `#if LPA_K1_VARIANTS`).
template <bool WRITE_EB, bool WAVE_REDUCE, bool DEFER, int RELOC_MODE, bool COOP, bool RHO, bool NOIG = false>
__global__ void __launch_bounds__(K1_THREADS, (RELOC_MODE || COOP) ? 4 : 1) k_push_deposit_tiled_2d(GridV g, PartV p, PushK k,
                                                              const int32_t *__restrict__ blk_tile,
                                                              const int32_t *__restrict__ blk_begin,
                                                              const int32_t *__restrict__ blk_end,
                                                              const int32_t *__restrict__ n_blocks,
                                                              int tiles_y, uint32_t *overflow,
                                                              uint32_t *overflow_count, int part,
                                                              int tiles_x, int edge_cols, Scratch7 sc, Reloc rl,
                                                              Coop co) {
    constexpr bool RELOC = RELOC_MODE != 0, RL_INIT = RELOC_MODE == 2;
    static_assert(!RELOC || (DEFER && !WAVE_REDUCE && !WRITE_EB), "RELOC rides on the parked-crosser pass");
    static_assert(!COOP || (DEFER && !WAVE_REDUCE), "COOP parks what it cannot deposit");
    // NOIG (lpa_push_params.flags & LPA_PUSH_NO_IG): inv_gamma is a function of the momenta, so the resident store need
    // not stream it -- 1 / gamma is recomputed from (ux, uy, uz) with the very function the Boris rotation ends with
    // (inv_gamma_of: ~10 VALU instructions) instead of being loaded, and is not written back: 16 of the 105 bytes per
    // particle-update.  K1 2-D moves its bytes at 72 % of what a plain streaming kernel with the same accesses reaches
    // (tools/ubench/stream_soa.hip: 6.07 TB/s) and every attribute stream it drops is worth 0.06-0.09 ms (DESIGN.md
    // DESIGN_HISTORY.md, round 3).  The array goes stale; lpa_refresh_inv_gamma rebuilds it for whoever reads it.
    static_assert(!NOIG || (DEFER && !WRITE_EB && !WAVE_REDUCE && !RELOC_MODE && !COOP), "NOIG: product path only");
    static_assert(LPA_K1_VARIANTS || (!WAVE_REDUCE && !RELOC && !COOP), "variant paths need -DLPA_K1_VARIANTS=1");
    static_assert(RHO || (!WAVE_REDUCE && !COOP), "the variant deposits always carry rho");
    constexpr int NJ = RHO ? 4 : 3;      // jx jy jz (rho)
    __shared__ __attribute__((aligned(16))) double s_eb[RSZ];
    __shared__ double s_j[NJ][RSZJ];
    __shared__ int s_ncross;
#if LPA_K1_VARIANTS
    __shared__ int s_stk_cnt[RELOC ? RL_CLASSES : 1], s_stk[RELOC ? RL_CLASSES * RL_DEPTH : 1];
    __shared__ int s_hl_dst[RELOC ? RL_CLASSES * RL_DEPTH : 1], s_nhl;
#endif
    // XCD-aware order: workgroups are dealt round-robin over the 8 XCDs (each with its own L2), work blocks
    // are in tile order: give every XCD a contiguous run of them, so that neighbouring tiles -- which
    // share halo rows of E / B and flush into the same J lines -- meet in one L2 (measured effect on C2:
    // none, the staging is 4 % of the kernel's reads and the kernel is not HBM bound)
    const int nb = *n_blocks, chunk = (nb + 7) >> 3;
    // (an edge / interior part launch keeps the plain order: a run of tile columns per XCD would leave the
    // XCDs that own the other part's columns idle -- measured 2x on the 3-D slab)
    const int wb = part ? (int)blockIdx.x : (int)(blockIdx.x & 7u) * chunk + (int)(blockIdx.x >> 3);
    if ((int)blockIdx.x >= 8 * chunk || wb >= nb) return;  // block-uniform
    const int tile = blk_tile[wb];
    if (part) {  // LPA_PART_EDGE: the edge_cols tile columns at each x face; LPA_PART_INTERIOR: the others
        const int txi = tile / tiles_y;
        const bool edge = txi < edge_cols || txi >= tiles_x - edge_cols;
        if ((part == LPA_PART_EDGE) != edge) return;  // block-uniform
    }
    const int begin = blk_begin[wb], end = blk_end[wb];
    // The attribute and scratch arrays, rebased to the block's first slot (once per workgroup, on the scalar unit): the
    // 32-bit byte offsets of the accesses below are relative to it -- a work block holds a few thousand slots -- so a
    // store is limited by the int32 slot numbers of the sort's tables (2^31), not by 2^32 bytes per array (2^29 slots).
    const int rb = LPA_ABS_OFFSETS ? 0 : begin;     // (LPA_ABS_OFFSETS=1: the old absolute offsets, for A/B timing builds)
    p.x += rb; p.y += rb; p.ux += rb; p.uy += rb; p.uz += rb; p.w += rb;
    if (!NOIG) p.ig += rb;
    if (DEFER) {
#pragma unroll
        for (int c = 0; c < 7; c++) sc.a[c] += rb;
    }
    const int tx0 = (tile / tiles_y) * TX, ty0 = (tile % tiles_y) * TY;  // first node of the tile
    const int rx0 = tx0 - HALO, ry0 = ty0 - HALO;                            // first node of the region
    const int lane = threadIdx.x & 63;
    // COOP: first slot of the tile and one past its last full-stripe slot (both multiples of 64, like `begin`)
    if (DEFER && threadIdx.x == 0) s_ncross = 0;
#if LPA_K1_VARIANTS
    [[maybe_unused]] int tile_first = 0, pad_end = 0;
    if (COOP) {
        tile_first = co.tile_off[tile];
        pad_end = tile_first + co.pad_ranks[tile] * 256;
    }
    if (RELOC) {
        if (threadIdx.x < RL_CLASSES) s_stk_cnt[threadIdx.x] = 0;
        if (threadIdx.x == 0) s_nhl = 0;
    }
#endif

    // ---- stage E/B (along an open axis nodes outside the padded array are never touched by a fast-path
    //      particle)
    {
        const double *src[6] = {g.ex, g.ey, g.ez, g.bx, g.by, g.bz};
        for (int t = threadIdx.x; t < RWX * RWY; t += blockDim.x) {
            int lx = t / RWY, ly = t - lx * RWY;
            // along a locally periodic axis a region node outside [0, n) is its periodic image inside
            // (a tile at a box edge, or the image of a particle that was folded through the face)
            int nxn = rx0 + lx, nyn = ry0 + ly;
            if ((k.wrap & 1) && (unsigned)nxn >= (unsigned)g.nx) { nxn %= g.nx; if (nxn < 0) nxn += g.nx; }
            if ((k.wrap & 2) && (unsigned)nyn >= (unsigned)g.ny) { nyn %= g.ny; if (nyn < 0) nyn += g.ny; }
            int cx = nxn + g.ng, cy = nyn + g.ng;
            bool in = (unsigned)cx < (unsigned)g.NX && (unsigned)cy < (unsigned)g.NY;
            long gi = (long)cx * g.NY + cy;
            double v[6];
#pragma unroll
            for (int c = 0; c < 6; c++) v[c] = in ? src[c][gi] : 0.0;
            double *row = s_eb + lx * RS;
            *(lpa_d2 *)(row + EB_PAIR_A + 2 * ly) = lpa_d2{v[1], v[3]};   // ey, bx
            *(lpa_d2 *)(row + EB_PAIR_B + 2 * ly) = lpa_d2{v[0], v[4]};   // ex, by
            row[EB_EZ + ly] = v[2];
            row[EB_BZ + ly] = v[5];
#pragma unroll
            for (int c = 0; c < NJ; c++) s_j[c][lx * RSJ + ly] = 0.0;
        }
    }
    __syncthreads();

    const double inv_dx = 1.0 / g.dx, inv_dy = 1.0 / g.dy;
    // wave-uniform trip count: every lane of a wave runs the same iterations (the deposit below uses
    // wave-wide DPP / permlane operations)
    // software pipeline: the seven attribute loads of the NEXT iteration are issued before the current
    // particle is processed, so their HBM latency hides under the VALU / LDS work of this one
    // particle attributes are addressed as uniform base + 32-bit byte offset (the sorted range is far
    // relative to the work block's first slot): one VALU instruction per access instead of a 64-bit address each
    auto ld = [](const double *base, uint32_t off) { return *(const double *)((const char *)base + off); };
    auto st = [](double *base, uint32_t off, double v) { *(double *)((char *)base + off) = v; };
    // the particle attributes stream through once per step: non-temporal (see lpa_particles3d.hip)
    // LPA_NT_PARTICLES_2D: bit 0 = loads, bit 1 = stores
    auto ldp = [](const double *base, uint32_t off) {
        const double *q = (const double *)((const char *)base + off);
        return (LPA_NT_PARTICLES_2D & 1) ? __builtin_nontemporal_load(q) : *q; };
    auto stp = [](double *base, uint32_t off, double v) {
        double *q = (double *)((char *)base + off);
        if (LPA_NT_PARTICLES_2D & 2) __builtin_nontemporal_store(v, q); else *q = v; };
    double nx_ = 0.0, ny_ = 0.0, nux = 0.0, nuy = 0.0, nuz = 0.0, nig = 1.0, nw = 0.0;
    {
        const int ip0 = begin + (int)(threadIdx.x & ~63u) + lane;
        if (ip0 < end) {
            const uint32_t o = (uint32_t)(ip0 - rb) * 8u;
            nx_ = ldp(p.x, o); ny_ = ldp(p.y, o); nux = ldp(p.ux, o); nuy = ldp(p.uy, o); nuz = ldp(p.uz, o);
            if (!NOIG) nig = ldp(p.ig, o);
            nw = ldp(p.w, o);
        }
    }
    for (int it = begin + (int)(threadIdx.x & ~63u); it < end; it += blockDim.x) {
        const int ip = it + lane;
        bool valid = ip < end;
        double x = nx_, y = ny_, ux = nux, uy = nuy, uz = nuz, ig = nig, w = nw;
        // the slot's class: issued here, consumed after the Boris rotation (not carried across iterations: the
        // loop has no VGPR to spare at 4 waves per SIMD)
        [[maybe_unused]] uint32_t ccls = 0;
        if (RELOC && !RL_INIT && valid) ccls = rl.cls[ip] & 31u;   // (slots never classified hold whatever: 5 bits)
        {
            const int ipn = ip + (int)blockDim.x;
            if (ipn < end) {
                const uint32_t o = (uint32_t)(ipn - rb) * 8u;
                nx_ = ldp(p.x, o); ny_ = ldp(p.y, o); nux = ldp(p.ux, o); nuy = ldp(p.uy, o); nuz = ldp(p.uz, o);
                if (!NOIG) nig = ldp(p.ig, o);
                nw = ldp(p.w, o);
            }
        }
        if (NOIG) ig = inv_gamma_of(ux, uy, uz);
        valid = valid && !(isnan(x) || isnan(y));  // NaN: killed since the last sort (migration)
        // first half push and the nearest node (ix1, iy1) of the mid-step position.  The LDS path is
        // valid iff that node lies within the tile + margin: the gather then reads nodes ix1-2..ix1+1
        // and the deposit window (mid-step cell and the cell one step of < 1 cell further, +-1 ulp of
        // wobble in either) stays within ix1-3..ix1+3, all inside the HALO = margin + 3 nodes staged
        // around the tile.  Everything else goes to the overflow list untouched (nothing stored yet).
        double xo = 0.0, yo = 0.0;
        int ix1 = 0, iy1 = 0;
        if (valid) {
            x += k.cdt_half * ig * ux;
            y += k.cdt_half * ig * uy;
            xo = (x - g.x0) * inv_dx; yo = (y - g.y0) * inv_dy;
            ix1 = ifloor(xo + 0.5); iy1 = ifloor(yo + 0.5);
            // a particle that was folded through a locally periodic face since the last sort sits a whole box
            // away from its tile: work on its periodic image next to the tile (the staged halo covers the
            // guard cells, whose E / B are periodic images and whose J / rho are folded back afterwards) --
            // ~3 000 particles per step on C2 that would otherwise go through the overflow list
            // (only in boxes at least two padded tiles wide, where "half a box away from the tile centre"
            // cannot be confused with "inside the tile")
            if (k.wrap & 3) {
                const int ddx = ix1 - (tx0 + TX / 2), ddy = iy1 - (ty0 + TY / 2);
                if ((k.wrap & 1) && g.nx >= 2 * (TX + 8) && (ddx > (g.nx >> 1) || ddx < -(g.nx >> 1))) {
                    x += (ddx > 0 ? -1.0 : 1.0) * (k.hi[0] - k.lo[0]);
                    xo = (x - g.x0) * inv_dx; ix1 = ifloor(xo + 0.5);
                }
                if ((k.wrap & 2) && g.ny >= 2 * (TY + 8) && (ddy > (g.ny >> 1) || ddy < -(g.ny >> 1))) {
                    y += (ddy > 0 ? -1.0 : 1.0) * (k.hi[1] - k.lo[1]);
                    yo = (y - g.y0) * inv_dy; iy1 = ifloor(yo + 0.5);
                }
            }
            if ((unsigned)(ix1 - (tx0 - LPA_TILE_MARGIN)) >= (unsigned)(TX + 2 * LPA_TILE_MARGIN) ||
                (unsigned)(iy1 - (ty0 - LPA_TILE_MARGIN)) >= (unsigned)(TY + 2 * LPA_TILE_MARGIN)) {
                uint32_t slot = atomicAdd(overflow_count, 1u);
                overflow[slot] = (uint32_t)ip;
                valid = false;
            }
        }
        AxisW ax, ay;
        double vz = 0.0;
        int b0 = 0;
        bool cross = false;   // DEFER: changed cell during the step, deposited by the second pass
        [[maybe_unused]] bool mover = false;   // RELOC: the next gather cell has another y-class than this slot
        [[maybe_unused]] uint32_t cnow = 0;
        if (valid) {
            double eb[6];
            double gx[3], gy[3];
            {
                int ix2 = ifloor(xo), iy2 = ifloor(yo);
                double hx[3], hy[3];
                tsc3(ix1 - xo, gx);
                tsc3(ix2 - xo + 0.5, hx);
                tsc3(iy1 - yo, gy);
                tsc3(iy2 - yo + 0.5, hy);
                // in range by the margin test (ix2 is ix1 or ix1 - 1)
                int lx1 = ix1 - rx0, lx2 = ix2 - rx0, ly1 = iy1 - ry0, ly2 = iy2 - ry0;
                gather9_pair_l(s_eb + EB_PAIR_B, lx2, ly1, hx, gy, eb[0], eb[4]);   // ex, by
                gather9_pair_l(s_eb + EB_PAIR_A, lx1, ly2, gx, hy, eb[1], eb[3]);   // ey, bx
                eb[2] = gather9_l(s_eb + EB_EZ, lx1, ly1, gx, gy);
                eb[5] = gather9_l(s_eb + EB_BZ, lx2, ly2, hx, hy);
            }
            if (WRITE_EB) {
#pragma unroll
                for (int c = 0; c < 6; c++) p.eb[c][ip] = eb[c];
            }
            boris(ux, uy, uz, ig, eb[0], eb[1], eb[2], eb[3], eb[4], eb[5], k.efactor, k.bfactor);
            x += k.cdt_half * ig * ux;
            y += k.cdt_half * ig * uy;
            double vx = ux * LPA_C * ig, vy = uy * LPA_C * ig;
            vz = uz * LPA_C * ig;
            if (DEFER) {
                // The deposit runs from x - v dt/2 -- the mid-step position the fields were gathered at, up
                // to rounding -- to x + v dt/2: the old shape S0 IS the node-centred gather weights (gx, gy)
                // around (ix1, iy1).  A particle whose advanced position has the same nearest node needs
                // only the new weights on the same three cells; the others (~6 %) are parked and get the
                // general 4 x 4 window in the second pass.  (Taking the old cell from the gather position
                // instead of re-deriving it moves a particle that sits within an ulp of a cell boundary to
                // the neighbouring window, where the shape values agree to that ulp.)
                const double d1x = ix1 - (x + vx * 0.5 * k.dt - g.x0) * inv_dx;
                const double d1y = iy1 - (y + vy * 0.5 * k.dt - g.y0) * inv_dy;
                cross = !(d1x > -0.5 && d1x <= 0.5 && d1y > -0.5 && d1y <= 0.5);
#if LPA_K1_VARIANTS
                if (RELOC) {
                    // nearest node of the advanced deposit end point = the gather cell of the NEXT step
                    const int jy = iy1 + (int)floor(0.5 - d1y);
                    cnow = (uint32_t)(jy - ty0) & 31u;
                    if (RL_INIT) {
                        ccls = (uint32_t)(iy1 - ty0) & 31u;
                        rl.cls[ip] = (uint16_t)ccls;
                    }
                    mover = cnow != ccls;
                }
#endif
                tsc3(d1x, ax.S1);
                tsc3(d1y, ay.S1);
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    ax.S0[c] = gx[c]; ay.S0[c] = gy[c];
                    ax.DS[c] = ax.S1[c] - gx[c]; ay.DS[c] = ay.S1[c] - gy[c];
                }
                ax.S0[3] = ax.S1[3] = ax.DS[3] = 0.0;
                ay.S0[3] = ay.S1[3] = ay.DS[3] = 0.0;
                ax.base = ix1 - 1; ay.base = iy1 - 1;
                ax.tail_zero = ay.tail_zero = true;
            } else {
                axis_window(ax, x - vx * 0.5 * k.dt - g.x0, x + vx * 0.5 * k.dt - g.x0, 1.0 / g.dx);
                axis_window(ay, y - vy * 0.5 * k.dt - g.y0, y + vy * 0.5 * k.dt - g.y0, 1.0 / g.dy);
            }
            int bx = clampi(ax.base - rx0, 0, RWX - 4), by = clampi(ay.base - ry0, 0, RWY - 4);
            b0 = bx * RSJ + by;
            double xs = x, ys = y;
            if (finish_position_2d(xs, ys, k) && k.absorbed)    // rare: a particle reached an open face
                report_absorbed_2d(g, k, (x + vx * 0.5 * k.dt - g.x0) * inv_dx, (y + vy * 0.5 * k.dt - g.y0) * inv_dy,
                                   k.dep.c_rho * w);
            report_leaver(k, xs, ip, tile);
            if (RELOC) mover = mover && !isnan(xs);     // absorbed at an open face: the slot becomes a hole
            const uint32_t o = (uint32_t)(ip - rb) * 8u;
            stp(p.x, o, xs); stp(p.y, o, ys);
            stp(p.ux, o, ux); stp(p.uy, o, uy); stp(p.uz, o, uz);
            if (!NOIG) stp(p.ig, o, ig);
        } else {
#pragma unroll
            for (int c = 0; c < 4; c++) {
                ax.S0[c] = ax.S1[c] = ax.DS[c] = 0.0;
                ay.S0[c] = ay.S1[c] = ay.DS[c] = 0.0;
            }
            ax.base = ay.base = 0;
            ax.tail_zero = ay.tail_zero = false;
        }
#if LPA_K1_VARIANTS
        if (COOP && it + 64 <= pad_end) {      // wave-uniform: this wave walks 64 slots of a full stripe
            const int cn = (ip - tile_first) & 255;                 // the cell this lane owns
            const int lxn = cn >> 5, lyn = cn & 31;
            const bool fast = valid && !cross && ix1 - tx0 == lxn && iy1 - ty0 == lyn;
            if (valid && !fast) {              // changed cell, or not (any more) in its slot's cell: second pass
                const int slot = atomicAdd(&s_ncross, 1);
                const uint32_t o = (uint32_t)(begin - rb + slot) * 8u;
                st(sc.a[0], o, x); st(sc.a[1], o, y); st(sc.a[2], o, ux); st(sc.a[3], o, uy);
                st(sc.a[4], o, uz); st(sc.a[5], o, ig); st(sc.a[6], o, w);
            }
            // every lane from here on (the exchange reads all 64 lanes): lanes without a fast particle carry zeros
            const double wq = fast ? w : 0.0;
            const double cd = k.dep.c_rho * wq, fdx_ = k.dep.c_jx * wq, fdy_ = k.dep.c_jy * wq, fvz = cd * vz;
            const double mp = lyn != 0 ? 1.0 : 0.0, mn = lyn != 31 ? 1.0 : 0.0;   // row ends: nothing comes across
            const bool edge = fast && (lyn == 0 || lyn == 31);
            const int bn = (lxn + HALO - 1) * RSJ + lyn + HALO;    // own cell, window row 0
            const int be = bn + (lyn == 0 ? -1 : 1);               // the outward column of a row-end lane
            double bb[3], jxr[3] = {0.0, 0.0, 0.0};
#pragma unroll
            for (int l = 0; l < 3; l++) bb[l] = ay.S0[l] + 0.5 * ay.DS[l];
#pragma unroll
            for (int kk = 0; kk < 3; kk++) {
                const double a = ax.S0[kk] + 0.5 * ax.DS[kk];
                const double fdx = fdx_ * ax.DS[kk], t12 = (1.0 / 12.0) * ax.DS[kk];
                double vjx[3], vjy[3], vjz[3], vrh[3], jyr = 0.0;
#pragma unroll
                for (int l = 0; l < 3; l++) {
                    jxr[l] -= fdx * bb[l];
                    jyr -= fdy_ * (ay.DS[l] * a);
                    vjx[l] = jxr[l]; vjy[l] = jyr;
                    vjz[l] = fvz * (a * bb[l] + t12 * ay.DS[l]);
                    vrh[l] = cd * ax.S1[kk] * ay.S1[l];
                }
                const int o = bn + kk * RSJ;
                // own middle column + the left neighbour's upper column + the right neighbour's lower column
                atomicAdd(&s_j[NJ - 1][o], fma(mn, wr_dpp<0x130>(vrh[0]), fma(mp, wr_dpp<0x138>(vrh[2]), vrh[1])));
                atomicAdd(&s_j[2][o], fma(mn, wr_dpp<0x130>(vjz[0]), fma(mp, wr_dpp<0x138>(vjz[2]), vjz[1])));
                atomicAdd(&s_j[1][o], fma(mn, wr_dpp<0x130>(vjy[0]), vjy[1]));           // (column 2 of jy is the null run)
                if (kk < 2) atomicAdd(&s_j[0][o], fma(mn, wr_dpp<0x130>(vjx[0]), fma(mp, wr_dpp<0x138>(vjx[2]), vjx[1])));
                if (edge) {                    // nobody owns the column beyond the row end
                    const int oe = be + kk * RSJ;
                    const bool lo = lyn == 0;
                    atomicAdd(&s_j[NJ - 1][oe], lo ? vrh[0] : vrh[2]);
                    atomicAdd(&s_j[2][oe], lo ? vjz[0] : vjz[2]);
                    if (lo) atomicAdd(&s_j[1][oe], vjy[0]);
                    if (kk < 2) atomicAdd(&s_j[0][oe], lo ? vjx[0] : vjx[2]);
                }
            }
            continue;
        }
#endif

        if (!WAVE_REDUCE) {
            // ---- deposit, STRIPED order: the lanes of a half-wave sit in consecutive y-cells, so each
            // ds_add_f64 below hits 32 different bank pairs.
            if (valid) {
                if (DEFER && (cross || (RELOC && mover))) {
                    const int slot = atomicAdd(&s_ncross, 1);
                    const uint32_t o = (uint32_t)(begin - rb + slot) * 8u;
                    st(sc.a[0], o, x); st(sc.a[1], o, y); st(sc.a[2], o, ux); st(sc.a[3], o, uy);
                    st(sc.a[4], o, uz); if (!NOIG) st(sc.a[5], o, ig); st(sc.a[6], o, w);
                    if (RELOC) {
                        rl.aux_slot[begin + slot] = (uint32_t)ip;
                        rl.aux_info[begin + slot] = (ccls << 8) | (cnow << 16) | (cross ? RL_DEP : 0u) | (mover ? RL_MOV : 0u);
                    }
                }
                esirkepov_2d<true>(ax, ay, vz, w, k.q, g.dx, g.dy, k.dt,
                                   [&](int kk, int ll, double djx, double djy, double djz, double drho) {
                                       int o = b0 + kk * RSJ + ll;
                                       // window row 3 / column 3 carry exact zeros unless the particle
                                       // changed cell along that axis: predicate on the crossing flags
                                       // (one exec-mask region per run of cells) instead of testing 64
                                       // values -- crossers are ~3 % of the lanes
                                       bool on = (kk < 3 || !ax.tail_zero) && (ll < 3 || !ay.tail_zero);
                                       if (DEFER) on = kk < 3 && ll < 3 && !cross;   // cell-crossers come later
                                       // the running sum of jx over the 3 window rows of a particle that
                                       // stayed in its x-cell is (sum of DS) * b = 0 up to rounding (the
                                       // reference adds that 1e-16-relative residue): row 2 of jx and column
                                       // 2 of jy carry nothing unless the particle crossed along that axis,
                                       // and the crossers' values go with their tails
                                       bool on_x = on && (!LPA_SKIP_NULL_RUN || kk < 2 || (!DEFER && !ax.tail_zero));
                                       bool on_y = on && (!LPA_SKIP_NULL_RUN || ll < 2 || (!DEFER && !ay.tail_zero));
                                       if (on_x) atomicAdd(&s_j[0][o], djx);
                                       if (on_y) atomicAdd(&s_j[1][o], djy);
                                       if (on) {
                                           atomicAdd(&s_j[2][o], djz);
                                           if (RHO) atomicAdd(&s_j[NJ - 1][o], drho);
                                       }
                                   }, &k.dep);
            }
            continue;
        }
#if LPA_K1_VARIANTS
        // ---- deposit.  Particles are cell sorted, so most lanes of the wave share one 4x4 window:
        // those are summed across the wave in registers (reduce-scatter: lane L ends with the total
        // of window value L = quantity*16 + kx*4 + ly) and leave ONE ds_add_f64 per lane, all to
        // different addresses.  Lanes with another window (cell-crossers, wave straddling two cells)
        // form a second group or fall back to per-lane LDS atomics.
        unsigned long long todo = __ballot(valid);
        int round = 0;
        while (todo) {  // wave-uniform
            const int leader = __ffsll((long long)todo) - 1;
            const int lb = __builtin_amdgcn_readlane(b0, leader);
            const bool mine = ((todo >> lane) & 1ull) != 0;
            const unsigned long long grp = __ballot(mine && b0 == lb);
            if (__popcll(grp) >= WR_MIN_GROUP && round < WR_MAX_ROUNDS) {
                const bool ing = (grp >> lane) & 1ull;
                double s16[16];
                esirkepov_2d<true>(ax, ay, vz, ing ? w : 0.0, k.q, g.dx, g.dy, k.dt,
                                   [&](int kk, int ll, double djx, double djy, double djz, double drho) {
                                       double r0 = wr_red32(djx, djz);   // lanes <32: jx, >=32: jz
                                       double r1 = wr_red32(djy, drho);  // lanes <32: jy, >=32: rho
                                       s16[kk * 4 + ll] = wr_red16(r0, r1);  // rows: jx jy jz rho
                                       __builtin_amdgcn_sched_barrier(0);    // bound live registers
                                   });
                double tot = wr_finish16(s16, lane);
                if (tot != 0.0) atomicAdd(&s_j[lane >> 4][lb + ((lane >> 2) & 3) * RSJ + (lane & 3)], tot);
                todo &= ~grp;
                round++;
            } else {
                if (mine) {
                    // opaque copies: keeps the compiler from sharing sub-expressions between this
                    // cold path and the reduction above (that sharing cost 120 extra VGPRs)
                    AxisW cx = ax, cy = ay;
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        asm volatile("" : "+v"(cx.S0[c]), "+v"(cx.S1[c]), "+v"(cx.DS[c]));
                        asm volatile("" : "+v"(cy.S0[c]), "+v"(cy.S1[c]), "+v"(cy.DS[c]));
                    }
                    esirkepov_2d<true>(cx, cy, vz, w, k.q, g.dx, g.dy, k.dt,
                                       [&](int kk, int ll, double djx, double djy, double djz, double drho) {
                                           int o = b0 + kk * RSJ + ll;
                                           if (djx != 0.0) atomicAdd(&s_j[0][o], djx);
                                           if (djy != 0.0) atomicAdd(&s_j[1][o], djy);
                                           if (djz != 0.0) atomicAdd(&s_j[2][o], djz);
                                           if (drho != 0.0) atomicAdd(&s_j[NJ - 1][o], drho);
                                       });
                }
                todo = 0;
            }
        }
#endif
    }
    if (DEFER) {
        // ---- the particles that changed cell: the general 4 x 4 window, every lane busy
        __syncthreads();
        const int ncross = s_ncross;
        // drift gauge for the caller's sort policy: particles that changed cell this step (one atomic per block)
        if (rl.stats && threadIdx.x == 0 && ncross) atomicAdd(&rl.stats[0], (uint32_t)ncross);
        // RELOC: a thread re-seats (at most) the first parked particle it handles and keeps its state in registers
        // across the three barriers below -- no global-memory round trip between the phases (a version that
        // re-read the parked state in every phase cost more than the conflicts it removed)
        [[maybe_unused]] double m[7];
        [[maybe_unused]] unsigned long long mid = 0;
        [[maybe_unused]] bool mvac = false;
        [[maybe_unused]] int mcls = 0, mh = -1;
        for (int i = threadIdx.x; i < ncross; i += blockDim.x) {
            uint32_t info = RL_DEP;
            if (RELOC) info = rl.aux_info[begin + i];
            const bool first = RELOC && i == (int)threadIdx.x;
            if (!(info & RL_DEP) && !(first && (info & RL_MOV))) continue;
            const uint32_t o = (uint32_t)(begin - rb + i) * 8u;
            const double x = ld(sc.a[0], o), y = ld(sc.a[1], o), ux = ld(sc.a[2], o), uy = ld(sc.a[3], o),
                         uz = ld(sc.a[4], o), ig = NOIG ? inv_gamma_of(ux, uy, uz) : ld(sc.a[5], o), w = ld(sc.a[6], o);
#if LPA_K1_VARIANTS
            if (first && (info & RL_MOV)) {
                // phase A: the mover's slot joins the pool of its class (it only leaves if the pool has room)
                const int c = (int)((info >> 8) & 31u);
                const int pos = atomicAdd(&s_stk_cnt[c], 1);
                if (rl.stats) atomicAdd(&rl.stats[1], 1u);
                if (pos < RL_DEPTH) {
                    const uint32_t slot = rl.aux_slot[begin + i];
                    s_stk[c * RL_DEPTH + pos] = (int)slot;
                    if (p.id) mid = p.id[slot];      // read before anybody re-occupies the slot (barrier below)
                    mvac = true;
                    mcls = (int)((info >> 16) & 31u);
                    m[0] = x; m[1] = y; m[2] = ux; m[3] = uy; m[4] = uz; m[5] = ig; m[6] = w;
                } else {
                    atomicSub(&s_stk_cnt[c], 1);
                }
            }
#endif
            if (info & RL_DEP) {
                const double vx = ux * LPA_C * ig, vy = uy * LPA_C * ig, vz = uz * LPA_C * ig;
                AxisW ax, ay;
                axis_window(ax, x - vx * 0.5 * k.dt - g.x0, x + vx * 0.5 * k.dt - g.x0, 1.0 / g.dx);
                axis_window(ay, y - vy * 0.5 * k.dt - g.y0, y + vy * 0.5 * k.dt - g.y0, 1.0 / g.dy);
                const int bx = clampi(ax.base - rx0, 0, RWX - 4), by = clampi(ay.base - ry0, 0, RWY - 4);
                const int b0 = bx * RSJ + by;
                esirkepov_2d<true>(ax, ay, vz, w, k.q, g.dx, g.dy, k.dt,
                                   [&](int kk, int ll, double djx, double djy, double djz, double drho) {
                                       bool on = (kk < 3 || !ax.tail_zero) && (ll < 3 || !ay.tail_zero);
                                       if (on) {
                                           int oo = b0 + kk * RSJ + ll;
                                           atomicAdd(&s_j[0][oo], djx);
                                           atomicAdd(&s_j[1][oo], djy);
                                           atomicAdd(&s_j[2][oo], djz);
                                           if (RHO) atomicAdd(&s_j[NJ - 1][oo], drho);
                                       }
                                   }, &k.dep);
            }
        }
#if LPA_K1_VARIANTS
        if (RELOC) {
            // the mover's state (as parked: before the periodic fold) goes to slot `dst`
            auto seat = [&](int dst) {
                const uint32_t od = (uint32_t)(dst - rb) * 8u;
                double x = m[0], y = m[1];
                finish_position_2d(x, y, k);
                report_leaver(k, x, dst);
                st(p.x, od, x); st(p.y, od, y);
                st(p.ux, od, m[2]); st(p.uy, od, m[3]); st(p.uz, od, m[4]); st(p.ig, od, m[5]); st(p.w, od, m[6]);
                if (p.id) p.id[dst] = mid;
            };
            __syncthreads();
            // phase B: every mover that left takes a slot of its new class
            if (mvac) {
                const int pos = atomicSub(&s_stk_cnt[mcls], 1) - 1;
                if (pos >= 0) {
                    seat(s_stk[mcls * RL_DEPTH + pos]);
                } else {
                    atomicAdd(&s_stk_cnt[mcls], 1);
                    mh = atomicAdd(&s_nhl, 1);       // at most RL_CLASSES * RL_DEPTH movers left their slots
                }
            }
            __syncthreads();
            // phase C: the others share what is left (exactly as many slots as movers without one): the pools are
            // flattened into one list and the h-th of them takes the h-th slot; it is a mover again next step
            if (rl.stats && threadIdx.x == 0) {
                int left = 0;
                for (int c = 0; c < RL_CLASSES; c++) left += max(s_stk_cnt[c], 0);
                atomicAdd(&rl.stats[3], (uint32_t)s_nhl);
                atomicAdd(&rl.stats[2], (uint32_t)left);      // == s_nhl: slots left in the pools
            }
            if (threadIdx.x < 64) {         // one wave: exclusive scan of the 32 pool sizes
                const int c = (int)threadIdx.x;
                const int n = c < RL_CLASSES ? max(s_stk_cnt[c], 0) : 0;
                int inc = n;
#pragma unroll
                for (int o = 1; o < RL_CLASSES; o <<= 1) {
                    const int t = __shfl_up(inc, o, 64);
                    if (c >= o) inc += t;
                }
                for (int e = 0; e < n; e++) s_hl_dst[inc - n + e] = s_stk[c * RL_DEPTH + e];
            }
            __syncthreads();
            if (mh >= 0) seat(s_hl_dst[mh]);
        }
#endif
    }
    __syncthreads();

    // ---- flush the J tile: one FP64 global atomic per touched cell and component.  Consecutive
    //      threads walk consecutive y -> each wave instruction covers contiguous 8-B segments of a row
    {
        double *dst[4] = {g.jx, g.jy, g.jz, g.rho};
        for (int t = threadIdx.x; t < RWX * RWY; t += blockDim.x) {
            int lx = t / RWY, ly = t - lx * RWY;
            int nxn = rx0 + lx, nyn = ry0 + ly;     // periodic images as in the staging above
            if ((k.wrap & 1) && (unsigned)nxn >= (unsigned)g.nx) { nxn %= g.nx; if (nxn < 0) nxn += g.nx; }
            if ((k.wrap & 2) && (unsigned)nyn >= (unsigned)g.ny) { nyn %= g.ny; if (nyn < 0) nyn += g.ny; }
            int cx = nxn + g.ng, cy = nyn + g.ng;
            if ((unsigned)cx >= (unsigned)g.NX || (unsigned)cy >= (unsigned)g.NY) continue;
            long gi = (long)cx * g.NY + cy;
#pragma unroll
            for (int c = 0; c < NJ; c++) {
                double v = s_j[c][lx * RSJ + ly];
                if (v != 0.0) atomicAdd(&dst[c][gi], v);
            }
        }
    }
}

// self-test of the wave reduce-scatter: in[64][64] (value, lane) -> out[lane] = sum over lanes of in[lane][.]
__global__ void __launch_bounds__(64) k_selftest_wave_reduce(const double *in, double *out) {
    int lane = threadIdx.x;
    double s16[16];
#pragma unroll
    for (int c = 0; c < 16; c++) {
        double r0 = wr_red32(in[(0 * 16 + c) * 64 + lane], in[(2 * 16 + c) * 64 + lane]);
        double r1 = wr_red32(in[(1 * 16 + c) * 64 + lane], in[(3 * 16 + c) * 64 + lane]);
        s16[c] = wr_red16(r0, r1);
    }
    out[lane] = wr_finish16(s16, lane);
}

// self-test of the neighbour exchange of the cooperative deposit: out[lane] = value of lane - 1 (wave_shr:1, 0 for
// lane 0), out[64 + lane] = value of lane + 1 (wave_shl:1, 0 for lane 63)
__global__ void __launch_bounds__(64) k_selftest_wave_shift(const double *in, double *out) {
    const int lane = threadIdx.x;
    const double v = in[lane];
    out[lane] = wr_dpp<0x138>(v);
    out[64 + lane] = wr_dpp<0x130>(v);
}

extern "C" int lpa_selftest_wave_shift(const double *in, double *out, void *stream) {
    LPA_REQUIRE(in && out, "lpa_selftest_wave_shift: bad args");
    hipLaunchKernelGGL(k_selftest_wave_shift, dim3(1), dim3(64), 0, (hipStream_t)stream, in, out);
    LPA_CHECK_LAUNCH("lpa_selftest_wave_shift");
    return LPA_OK;
}

extern "C" int lpa_selftest_wave_reduce(const double *in, double *out, void *stream) {
    LPA_REQUIRE(in && out, "lpa_selftest_wave_reduce: bad args");
    hipLaunchKernelGGL(k_selftest_wave_reduce, dim3(1), dim3(64), 0, (hipStream_t)stream, in, out);
    LPA_CHECK_LAUNCH("lpa_selftest_wave_reduce");
    return LPA_OK;
}

// =====================================================================================================
// host entry points
// =====================================================================================================
static int check_push(const lpa_grid *g, const lpa_particles *p, const lpa_push_params *pp,
                      const char *name) {
    LPA_REQUIRE(lpa_grid_ok(g, 2, 1), "%s: bad grid", name);
    LPA_REQUIRE(lpa_part_ok(p, 2), "%s: bad particle store", name);
    LPA_REQUIRE(pp && pp->dt > 0 && pp->m > 0, "%s: dt and m must be > 0", name);
    LPA_REQUIRE(!(pp->flags & LPA_PUSH_NO_RHO) || !(pp->wrap & (3 * LPA_ABSORB_X)) || pp->absorbed,
                "%s: LPA_PUSH_NO_RHO with absorbing faces needs the absorbed list", name);
    LPA_REQUIRE(!pp->absorbed || (pp->absorbed_count && pp->absorbed_capacity > 0), "%s: bad absorbed list", name);
    LPA_REQUIRE(!pp->leavers || (pp->leaver_count && pp->leaver_capacity > 0 && pp->leave_lo < pp->leave_hi),
                "%s: bad leaver list", name);
    return LPA_OK;
}

extern "C" int lpa_push_deposit_2d(const lpa_grid *g, const lpa_particles *p, const lpa_push_params *pp,
                                   int64_t first, int64_t count, void *stream) {
    if (int e = check_push(g, p, pp, "lpa_push_deposit_2d")) return e;
    LPA_REQUIRE(first >= 0 && count >= 0 && first + count <= p->n, "lpa_push_deposit_2d: bad range");
    if (count == 0) return LPA_OK;
    long nb = (count + 255) / 256;
    hipLaunchKernelGGL(k_push_deposit_global_2d, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream,
                       make_gridv(g, 2), make_partv(p), make_pushk(pp, g), (long)first, (long)count);
    LPA_CHECK_LAUNCH("lpa_push_deposit_2d");
    return LPA_OK;
}

extern "C" int lpa_push_deposit_list_2d(const lpa_grid *g, const lpa_particles *p,
                                        const lpa_push_params *pp, const uint32_t *list,
                                        const uint32_t *list_count, int64_t max_count, void *stream) {
    if (int e = check_push(g, p, pp, "lpa_push_deposit_list_2d")) return e;
    LPA_REQUIRE(list && list_count && max_count >= 0, "lpa_push_deposit_list_2d: bad list");
    if (max_count == 0 || p->n == 0) return LPA_OK;
    long nb = (max_count + 255) / 256;
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(k_push_deposit_list_2d, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream,
                       make_gridv(g, 2), make_partv(p), make_pushk(pp, g), list, list_count);
    LPA_CHECK_LAUNCH("lpa_push_deposit_list_2d");
    return LPA_OK;
}

int lpai_push_deposit_rest_2d(const lpa_grid *g, const lpa_particles *p, const lpa_push_params *pp, const uint32_t *list,
                              const uint32_t *list_count, int64_t max_count, int64_t loose_first, int64_t loose_count,
                              const int32_t *loose_limit, const lpa_pack_args *pack, void *stream) {
    if (int e = check_push(g, p, pp, "lpai_push_deposit_rest_2d")) return e;
    LPA_REQUIRE((!list || list_count) && max_count >= 0 && loose_first >= 0 && loose_count >= 0 &&
                    loose_first + loose_count <= p->n, "lpai_push_deposit_rest_2d: bad list / range");
    LPA_REQUIRE(!pack || pack_args_ok(pack, p), "lpai_push_deposit_rest_2d: bad pack arguments");
    const long work = (list ? max_count : 0) > loose_count ? (long)max_count : (long)loose_count;
    if ((work == 0 && !(pack && pack->with_list)) || p->n == 0) return LPA_OK;
    long nb = (work + 255) / 256;
    if (nb > 2048) nb = 2048;
    if (pack) {     // the step's leaver pack rides in this launch
        const PackArgsV pk = make_pack_args_v(pack);
        hipLaunchKernelGGL(k_push_deposit_rest_pack_2d, dim3((unsigned)(nb + pk.pack_blocks)), dim3(256), 0, (hipStream_t)stream,
                           make_gridv(g, 2), make_partv(p), make_pushk(pp, g), list, list_count, (long)loose_first,
                           (long)loose_count, loose_limit, pk);
        LPA_CHECK_LAUNCH("lpai_push_deposit_rest_2d (with the leaver pack)");
        return LPA_OK;
    }
    hipLaunchKernelGGL(k_push_deposit_rest_2d, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, make_gridv(g, 2),
                       make_partv(p), make_pushk(pp, g), list, list_count, (long)loose_first, (long)loose_count, loose_limit);
    LPA_CHECK_LAUNCH("lpai_push_deposit_rest_2d");
    return LPA_OK;
}

extern "C" int lpa_push_deposit_tiled_2d(const lpa_grid *g, const lpa_particles *p,
                                         const lpa_push_params *pp, const lpa_tiling *t,
                                         uint32_t *overflow, uint32_t *overflow_count, void *stream) {
    return lpa_push_deposit_tiled_part_2d(g, p, pp, t, overflow, overflow_count, LPA_PART_ALL, 0, stream);
}

extern "C" int lpa_push_deposit_tiled_part_2d(const lpa_grid *g, const lpa_particles *p,
                                              const lpa_push_params *pp, const lpa_tiling *t,
                                              uint32_t *overflow, uint32_t *overflow_count, int part,
                                              int edge_cols, void *stream) {
    if (int e = check_push(g, p, pp, "lpa_push_deposit_tiled_2d")) return e;
    LPA_REQUIRE(part == LPA_PART_ALL || part == LPA_PART_EDGE || part == LPA_PART_INTERIOR,
                "lpa_push_deposit_tiled_part_2d: bad part");
    LPA_REQUIRE(part == LPA_PART_ALL || edge_cols >= 1, "lpa_push_deposit_tiled_part_2d: edge_cols must be >= 1");
    LPA_REQUIRE(t && t->blk_tile && t->blk_begin && t->blk_end && t->n_blocks && t->max_blocks > 0 &&
                    overflow && overflow_count,
                "lpa_push_deposit_tiled_2d: bad tiling");
    LPA_REQUIRE(t->tiles_x == (g->nx + TX - 1) / TX && t->tiles_y == (g->ny + TY - 1) / TY,
                "lpa_push_deposit_tiled_2d: tiling does not match the grid");
    LPA_REQUIRE(p->is_dead == nullptr,
                "lpa_push_deposit_tiled_2d: tile-binned stores carry no is_dead array (dead = NaN x)");
    // the LDS region covers margin + 3 nodes around the tile: one cell of motion per step at most
    LPA_REQUIRE(LPA_C * pp->dt <= g->dx && LPA_C * pp->dt <= g->dy,
                "lpa_push_deposit_tiled_2d: c*dt exceeds a cell (CFL); use lpa_push_deposit_2d");
    if (t->n_sorted == 0) return LPA_OK;
    // particle attributes are addressed with 32-bit byte offsets inside the kernel
    LPA_REQUIRE(p->n < (1ll << 31) - 1, "lpa_push_deposit_tiled_2d: more than 2^31 - 2 slots in one store");
    GridV gv = make_gridv(g, 2);
    PartV pv = make_partv(p);
    PushK k = make_pushk(pp, g);
    const bool eb = p->part_eb[0] != nullptr, rho = !(pp->flags & LPA_PUSH_NO_RHO);
    Scratch7 sc;
    bool defer = true;
    for (int c = 0; c < 7; c++) {
        sc.a[c] = t->scratch[c];
        defer = defer && sc.a[c] != nullptr;
    }
    Reloc rl{t->slot_class, t->aux_slot, t->aux_info, (unsigned long long *)t->scratch[7], t->reloc_stats,
             t->class_init};
    Coop co{t->tile_off, t->pad_ranks};
#define LPA_LAUNCH_TILED(E, W, D, R, CO, RH, ...)                                                             \
    hipLaunchKernelGGL((k_push_deposit_tiled_2d<E, W, D, R, CO, RH, ##__VA_ARGS__>), dim3(t->max_blocks), dim3(K1_THREADS), 0, \
                       (hipStream_t)stream, gv, pv, k, t->blk_tile, t->blk_begin, t->blk_end, t->n_blocks,   \
                       t->tiles_y, overflow, overflow_count, part, t->tiles_x, edge_cols, sc, rl, co)
    const bool noig = pp->flags & LPA_PUSH_NO_IG;
    LPA_REQUIRE(!noig || (defer && !eb && t->order == LPA_ORDER_STRIPED && !t->slot_class),
                "lpa_push_deposit_tiled_2d: LPA_PUSH_NO_IG needs a STRIPED store without E / B write-back and slot "
                "classes, and the scratch arrays of the second pass");
#if LPA_K1_VARIANTS
    // CELL_MAJOR stores use the wave reduce-scatter deposit, PADDED stores the cooperative one; with slot classes the
    // striped path re-seats its movers.  The variant deposits always carry rho.
    const bool wr = t->order == LPA_ORDER_CELL_MAJOR;
    defer = defer && !wr;
    // the in-kernel re-seating needs the slot classes, two uint32 scratch arrays and (when ids are carried) an
    // 8-byte one
    const bool reloc = defer && !eb && rho && rl.cls && rl.aux_slot && rl.aux_info && (rl.aux_id || !pv.id);
    const bool coop = t->order == LPA_ORDER_PADDED && defer && !eb && rho && t->pad_ranks && t->tile_off;
    LPA_REQUIRE(t->order != LPA_ORDER_PADDED || !reloc, "lpa_push_deposit_tiled_2d: slot classes and the padded order exclude each other");
    LPA_REQUIRE(rho || !wr, "lpa_push_deposit_tiled_2d: LPA_PUSH_NO_RHO is not available for CELL_MAJOR stores");
    if (coop) LPA_LAUNCH_TILED(false, false, true, 0, true, true);
    else if (eb && wr) LPA_LAUNCH_TILED(true, true, false, 0, false, true);
    else if (wr) LPA_LAUNCH_TILED(false, true, false, 0, false, true);
    else if (reloc && rl.init) LPA_LAUNCH_TILED(false, false, true, 2, false, true);
    else if (reloc) LPA_LAUNCH_TILED(false, false, true, 1, false, true);
    else
#else
    // the product build has one deposit path (per-lane LDS atomics + the dense second pass); it is correct for every
    // order the sort can produce (holes of a PADDED store are dead slots), only the STRIPED one makes it conflict free
    LPA_REQUIRE(!t->slot_class, "lpa_push_deposit_tiled_2d: the in-kernel re-seating (slot classes) is compiled only "
                                "into the variants library (-DLPA_K1_VARIANTS=1)");
#endif
    if (noig && rho) LPA_LAUNCH_TILED(false, false, true, 0, false, true, true);
    else if (noig) LPA_LAUNCH_TILED(false, false, true, 0, false, false, true);
    else if (eb && defer && rho) LPA_LAUNCH_TILED(true, false, true, 0, false, true);
    else if (eb && defer) LPA_LAUNCH_TILED(true, false, true, 0, false, false);
    else if (eb && rho) LPA_LAUNCH_TILED(true, false, false, 0, false, true);
    else if (eb) LPA_LAUNCH_TILED(true, false, false, 0, false, false);
    else if (defer && rho) LPA_LAUNCH_TILED(false, false, true, 0, false, true);
    else if (defer) LPA_LAUNCH_TILED(false, false, true, 0, false, false);
    else if (rho) LPA_LAUNCH_TILED(false, false, false, 0, false, true);
    else LPA_LAUNCH_TILED(false, false, false, 0, false, false);
#undef LPA_LAUNCH_TILED
    LPA_CHECK_LAUNCH("lpa_push_deposit_tiled_2d");
    return LPA_OK;
}

// =====================================================================================================
// split kernels (callback-in-pusher-stage path)
// =====================================================================================================
__global__ void __launch_bounds__(256) k_interpolate_2d(GridV g, PartV p) {
    long ip = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (ip >= p.n) return;
    if (p.dead && p.dead[ip]) return;  // interpolation/cpu2d.c:127 skips is_dead only
    double eb[6];
    gather_global_2d(g, (p.x[ip] - g.x0) / g.dx, (p.y[ip] - g.y0) / g.dy, eb);
#pragma unroll
    for (int c = 0; c < 6; c++) p.eb[c][ip] = eb[c];
}

__global__ void __launch_bounds__(256) k_boris(PartV p, double efactor, double bfactor) {
    long ip = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (ip >= p.n) return;
    if (p.dead && p.dead[ip]) return;
    double ux = p.ux[ip], uy = p.uy[ip], uz = p.uz[ip], ig;
    boris(ux, uy, uz, ig, p.eb[0][ip], p.eb[1][ip], p.eb[2][ip], p.eb[3][ip], p.eb[4][ip], p.eb[5][ip],
          efactor, bfactor);
    p.ux[ip] = ux; p.uy[ip] = uy; p.uz[ip] = uz; p.ig[ip] = ig;
}

__global__ void __launch_bounds__(256) k_push_position_2d(PartV p, double cdt) {
    long ip = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (ip >= p.n) return;
    if (p.dead && p.dead[ip]) return;
    double ig = p.ig[ip];
    p.x[ip] += cdt * ig * p.ux[ip];
    p.y[ip] += cdt * ig * p.uy[ip];
}

__global__ void __launch_bounds__(256) k_deposit_2d(GridV g, PartV p, double dt, double q) {
    long ip = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (ip >= p.n) return;
    double x = p.x[ip], y = p.y[ip];
    if ((p.dead && p.dead[ip]) || isnan(x) || isnan(y)) return;
    deposit_global_2d<false>(g, x, y, p.ux[ip], p.uy[ip], p.uz[ip], p.ig[ip], p.w[ip], q, dt);
}

__global__ void __launch_bounds__(256) k_wrap_positions_2d(PartV p, PushK k) {
    long ip = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (ip >= p.n) return;
    if (p.dead && p.dead[ip]) return;
    double x = p.x[ip], y = p.y[ip];
    if (isnan(x) || isnan(y)) return;
    finish_position_2d(x, y, k);
    p.x[ip] = x; p.y[ip] = y;
}

extern "C" int lpa_wrap_positions_2d(const lpa_particles *p, const lpa_push_params *pp, void *stream) {
    LPA_REQUIRE(lpa_part_ok(p, 2) && pp, "lpa_wrap_positions_2d: bad args");
    if (p->n == 0 || !pp->wrap) return LPA_OK;
    lpa_push_params q = *pp;
    if (!(q.dt > 0)) q.dt = 1.0;
    if (!(q.m > 0)) q.m = 1.0;
    hipLaunchKernelGGL(k_wrap_positions_2d, dim3((unsigned)((p->n + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, make_partv(p), make_pushk(&q));
    LPA_CHECK_LAUNCH("lpa_wrap_positions_2d");
    return LPA_OK;
}

extern "C" int lpa_interpolate_2d(const lpa_grid *g, const lpa_particles *p, void *stream) {
    LPA_REQUIRE(lpa_grid_ok(g, 2, 0) && lpa_part_ok(p, 2) && (p->n == 0 || p->part_eb[0]),
                "lpa_interpolate_2d: bad args (part_eb required)");
    if (p->n == 0) return LPA_OK;
    hipLaunchKernelGGL(k_interpolate_2d, dim3((unsigned)((p->n + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, make_gridv(g, 2), make_partv(p));
    LPA_CHECK_LAUNCH("lpa_interpolate_2d");
    return LPA_OK;
}

extern "C" int lpa_boris(const lpa_particles *p, double dt, double q, double m, void *stream) {
    LPA_REQUIRE(p && p->n >= 0 && (p->n == 0 || (p->ux && p->uy && p->uz && p->inv_gamma && p->part_eb[0])) &&
                    m > 0,
                "lpa_boris: bad args (part_eb required)");
    if (p->n == 0) return LPA_OK;
    hipLaunchKernelGGL(k_boris, dim3((unsigned)((p->n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       make_partv(p), q * dt / (2 * m * LPA_C), q * dt / (2 * m));
    LPA_CHECK_LAUNCH("lpa_boris");
    return LPA_OK;
}

extern "C" int lpa_push_position_2d(const lpa_particles *p, double dt, void *stream) {
    LPA_REQUIRE(lpa_part_ok(p, 2), "lpa_push_position_2d: bad particle store");
    if (p->n == 0) return LPA_OK;
    hipLaunchKernelGGL(k_push_position_2d, dim3((unsigned)((p->n + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, make_partv(p), LPA_C * dt);
    LPA_CHECK_LAUNCH("lpa_push_position_2d");
    return LPA_OK;
}

extern "C" int lpa_deposit_2d(const lpa_grid *g, const lpa_particles *p, double dt, double q,
                              void *stream) {
    LPA_REQUIRE(lpa_grid_ok(g, 2, 1) && lpa_part_ok(p, 2) && dt > 0, "lpa_deposit_2d: bad args");
    if (p->n == 0) return LPA_OK;
    hipLaunchKernelGGL(k_deposit_2d, dim3((unsigned)((p->n + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, make_gridv(g, 2), make_partv(p), dt, q);
    LPA_CHECK_LAUNCH("lpa_deposit_2d");
    return LPA_OK;
}

// =====================================================================================================
// particle diagnostics: kinetic energy sum w (gamma - 1) m c^2 and live count
// (reference tests/test_numerical_heating.py:40-50)
// =====================================================================================================
__global__ void __launch_bounds__(256) k_diag_particles(PartV p, double mc2, double *out) {
    double e = 0.0, n = 0.0;
    for (long ip = (long)blockIdx.x * blockDim.x + threadIdx.x; ip < p.n;
         ip += (long)gridDim.x * blockDim.x) {
        if ((p.dead && p.dead[ip]) || isnan(p.x[ip])) continue;
        e += p.w[ip] * (1.0 / p.ig[ip] - 1.0);
        n += 1.0;
    }
    block_atomic_sum(e * mc2, out + 0);
    block_atomic_sum(n, out + 1);
}

__global__ void __launch_bounds__(256) k_refresh_inv_gamma(PartV p, long first, long count) {
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < count; t += (long)gridDim.x * blockDim.x) {
        const long ip = first + t;
        p.ig[ip] = inv_gamma_of(p.ux[ip], p.uy[ip], p.uz[ip]);
    }
}

extern "C" int lpa_refresh_inv_gamma(const lpa_particles *p, int64_t first, int64_t count, void *stream) {
    LPA_REQUIRE(p && first >= 0 && count >= 0 && first + count <= p->n &&
                    (count == 0 || (p->ux && p->uy && p->uz && p->inv_gamma)),
                "lpa_refresh_inv_gamma: bad args");
    if (count == 0) return LPA_OK;
    const long nb = (count + 255) / 256;
    hipLaunchKernelGGL(k_refresh_inv_gamma, dim3((unsigned)(nb < 65536 ? nb : 65536)), dim3(256), 0, (hipStream_t)stream,
                       make_partv(p), (long)first, (long)count);
    LPA_CHECK_LAUNCH("lpa_refresh_inv_gamma");
    return LPA_OK;
}

extern "C" int lpa_diag_particles(const lpa_particles *p, double m, double *out, void *stream) {
    LPA_REQUIRE(p && out && p->n >= 0 && (p->n == 0 || (p->x && p->w && p->inv_gamma)),
                "lpa_diag_particles: bad args");
    if (p->n == 0) return LPA_OK;
    hipLaunchKernelGGL(k_diag_particles, dim3(1024), dim3(256), 0, (hipStream_t)stream, make_partv(p),
                       m * LPA_C * LPA_C, out);
    LPA_CHECK_LAUNCH("lpa_diag_particles");
    return LPA_OK;
}

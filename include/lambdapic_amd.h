/*
 * lambdapic_amd.h -- C ABI of the MI355X (gfx950) PIC inner loop.
 *
 * Drop-in boundary for lambdaPIC's per-step hot path: every entry point is
 * what a lambdaPIC facade would bind in place of one of its compiled CPU kernels; the reference
 * interface each one replaces is cited as file:line relative to /root/reference/src/lambdapic.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes, no C++/torch types.  All array pointers are DEVICE
 *    pointers (hipMalloc / torch CUDA tensors); the library never allocates or frees memory and
 *    never synchronises: every call enqueues work on `stream` (a hipStream_t passed as void*,
 *    NULL = the default stream) and returns.  Scratch comes from caller-provided workspaces.
 *  - every function returns 0 on success, a negative lpa_status on error; lpa_last_error()
 *    returns a message for the calling thread.  Nothing throws across the ABI.
 *  - FP64 throughout (the reference computes in float64 everywhere).
 *  - Field arrays: row-major double[NX][NY]([NZ]), NX = nx + 2*ng, in the CONVENTIONAL guard
 *    layout [ng | interior | ng]: interior node i lives at index i + ng.  This is lambdaPIC's
 *    wrapped layout (core/fields.py:24-27) rolled by +ng along every axis; all kernels treat the
 *    padded array as a torus exactly like the reference's INDEX2/INDEX3 macros and
 *    PRECOMPUTE_WRAP_INDICES (core/utils/cutils.h:19-26, core/current/current_deposit.h:41-49).
 *  - Particle arrays: SoA double[n]; a particle is skipped when is_dead[i] != 0 (if is_dead is
 *    given) or when x (or y, z) is NaN -- the reference's rule
 *    (core/pusher/unified/unified_pusher_2d.c:265-268,316-317).
 */
#ifndef LAMBDAPIC_AMD_H
#define LAMBDAPIC_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    LPA_OK = 0,
    LPA_ERR_ARG = -1,      /* bad argument (null pointer, non-positive size, unsupported shape) */
    LPA_ERR_HIP = -2,      /* a HIP runtime call / launch failed; see lpa_last_error()            */
    LPA_ERR_WORKSPACE = -3 /* workspace too small                                                 */
} lpa_status;

/* One rank's field slab.  Mirrors the attribute bag Fields2D/Fields3D (core/fields.py:56,78-170):
 * ex ey ez bx by bz jx jy jz rho, nx ny nz, n_guard, dx dy dz, x0 y0 z0.  nz = 1, dz = 0 in 2-D. */
typedef struct {
    int32_t nx, ny, nz; /* interior cells */
    int32_t ng;         /* guard cells per side (reference default 3, simulation.py:156) */
    double dx, dy, dz;
    double x0, y0, z0;  /* position of interior node 0 */
    double *ex, *ey, *ez, *bx, *by, *bz, *jx, *jy, *jz, *rho;
} lpa_grid;

/* One species' particle store.  Mirrors ParticlesBase (core/particles.py:63-67): x y z w ux uy uz
 * inv_gamma, ex_part..bz_part, _id, is_dead.  Optional pointers may be NULL:
 *   z (2-D), part_eb[0..5] (per-particle E/B write-back for callbacks), id, is_dead. */
typedef struct {
    int64_t n; /* slots in use (alive + dead) */
    double *x, *y, *z, *ux, *uy, *uz, *inv_gamma, *w;
    double *part_eb[6]; /* ex_part ey_part ez_part bx_part by_part bz_part, or all NULL */
    uint64_t *id;       /* bit pattern of ParticlesBase._id */
    uint8_t *is_dead;
} lpa_particles;

/* Tile binning of a particle store (product of lpa_sort_tiles_2d, consumed by
 * lpa_push_deposit_tiled_2d).  All pointers are device memory inside the sort workspace. */
typedef struct {
    int32_t tiles_x, tiles_y;  /* tile grid; tile = LPA_TILE_X x LPA_TILE_Y cells                */
    int64_t n_sorted;          /* particles [0, n_sorted) are tile ordered; the rest are "loose" */
    int32_t max_blocks;        /* launch bound for the tiled kernel                              */
    int32_t order;             /* LPA_ORDER_* the store was sorted into                          */
    const int32_t *tile_off;   /* [ntiles+1] first particle of each tile                         */
    const int32_t *blk_tile;   /* [max_blocks] tile of each work block                           */
    const int32_t *blk_begin;  /* [max_blocks] first particle of each work block                 */
    const int32_t *blk_end;    /* [max_blocks] one past the last particle                        */
    const int32_t *n_blocks;   /* [1] number of valid work blocks                                */
    int32_t tiles_z;           /* 3-D tilings (lpa_sort_tiles_3d): tiles along z; 0 for 2-D      */
    int32_t prefix_hint;       /* IN, lpa_sort_tiles_*: the caller knows that the first prefix_hint slots of `src` are
                                  what the previous sort through this workspace produced (its live count, read back
                                  with lpa_sort_live_count; dead slots among them are fine) -- the per-particle kernels
                                  of the re-sort then start behind them instead of launching a thread per slot that
                                  returns at once.  0 = unknown.  A hint the workspace header does not confirm makes
                                  the sort refuse (lpa_sort_overflow bit 2).  Reset to 0 on return.               */
    double *scratch[8];        /* optional (set by the caller, all or none): device arrays of at least n_sorted
                                  doubles each, e.g. the idle half of the ping-pong sort stores -- seven for a
                                  2-D tiling, eight for a 3-D one.  The tiled push kernels park the particles that
                                  changed cell there and deposit them on the general window in a dense second
                                  pass.  scratch[7] of a 2-D tiling (optional): one more 8-byte array, holds the ids
                                  of the particles the in-kernel re-seating moves                          */
    /* in-kernel cell-index sort of lpa_push_deposit_tiled_2d (optional; all three pointers or none):
     * slot_class[n_sorted] (caller's memory, two bytes per slot) holds the y-class (cell row mod 32 inside the
     * tile) every slot was sorted for; with class_init != 0 the push (re)writes it -- set it for the first push
     * after every sort.  Each step the kernel re-seats the particles whose next gather cell has another class
     * than their slot (a permutation inside each work block), so the LDS atomics of a 16-lane group keep
     * hitting 16 different bank pairs between two sorts.  aux_slot / aux_info: uint32 scratch of n_sorted
     * entries each -- lpa_sort_tiles_* points them at two arrays of its workspace that are idle between sorts. */
    const int32_t *pad_ranks;  /* [ntiles] LPA_ORDER_PADDED: full stripes per tile (0 for the other orders) */
    uint16_t *slot_class;
    uint32_t *aux_slot, *aux_info;
    uint32_t *reloc_stats;     /* optional, 4 counters the tiled 2-D kernel adds to.  [0]: particles parked for the second
                                  pass = particles that changed cell during the step (every build with the second
                                  pass; a drift gauge for the caller's sort policy).  With the re-seating: [1] movers,
                                  [2] / [3] slots left in the class pools after the class-matched round, counted from
                                  the pools and from the movers without a seat -- they must agree */
    int32_t class_init;
    int32_t stripe_ranks;      /* IN, lpa_sort_tiles_*: ranks per cell the striped orders keep in stripes -- 0 = the default rule
                                  (lpa_sort_stripe_ranks), else the number the workspace was sized for with
                                  lpa_sort_workspace_bytes_ranks (rounded up to a power of two in [32, 16384], as long as the table -- 32 bytes per
                                  tile and rank -- stays below 1 GiB).  OUT: the number
                                  used.  A cell's particles beyond it follow cell by cell behind the stripes of their tile,
                                  where the lanes of a wave share a cell and the LDS atomics of the tiled kernels serialise:
                                  a store whose deepest cell (lpa_sort_deepest_cell) exceeds it wants a larger one */
} lpa_tiling;

#define LPA_TILE_X 8       /* cells per tile along x                                            */
#define LPA_TILE_Y 32      /* cells per tile along y (the fastest axis): one half-wave = one row */
#ifndef LPA_TILE_MARGIN
#define LPA_TILE_MARGIN 2  /* cells a particle may sit outside its tile and stay on the LDS path */
#endif
/* 3-D tiles: 4 x 4 x 16 cells (z is the fastest axis): 256 cells like the 2-D tile; a 16-lane group
 * of a wave = 16 consecutive z-cells of one (x, y) column */
#define LPA_TILE3_X 4
#define LPA_TILE3_Y 4
#define LPA_TILE3_Z 16
#define LPA_TILE3_MARGIN 1
/* order of the particles inside a tile:
 *   CELL_MAJOR : all particles of cell 0, then of cell 1, ... -- the lanes of a wave share a cell;
 *                the tiled kernel sums their deposit windows across the wave in registers
 *                (permlane-swap / DPP reduce-scatter) and issues one LDS atomic per lane.
 *   STRIPED    : the r-th particle of every cell, for r = 0, 1, ... (cells y-fastest) -- the lanes
 *                of a half-wave sit in 32 consecutive y-cells, so their LDS gather reads and LDS
 *                atomics fall on 32 different bank pairs (conflict free). */
#define LPA_ORDER_CELL_MAJOR 0
#define LPA_ORDER_STRIPED 1
/* LPA_ORDER_PADDED (2-D and 3-D tilings): the striped order with its leading ranks -- those that at least LPA_PAD_MIN_CELLS of the
 * tile's 256 cells have -- stored as FULL stripes: slot = tile start + rank * 256 + cell, the missing cells are holes
 * (x = y = NaN); the other ranks follow compacted as in LPA_ORDER_STRIPED, and every tile's slot count is rounded up to
 * 64.  A slot's position then tells the cell it belongs to (lpa_tiling.pad_ranks[tile] = number of full stripes):
 * what the cooperative deposit of lpa_push_deposit_tiled_2d needs.  The store grows by the holes (a few per cent);
 * the live count is NOT the slot count (lpa_sort_live_count returns the slots). */
#define LPA_ORDER_PADDED 2
#define LPA_PAD_MIN_CELLS 192
/* LPA_ORDER_COLUMN: stripes over ONE column of a tile at a time -- the 16 z-cells of an (x, y) column of a 3-D tile, the
 * 32 y-cells of a row of a 2-D tile: the r-th particle of each of its cells for r = 0, 1, ..., then the next column.  A
 * 16-lane group then stays inside one column even where the stripes are partial (few particles per cell), so its LDS
 * gather reads fall on consecutive banks or on the same address. */
#define LPA_ORDER_COLUMN 3

const char *lpa_last_error(void);
int lpa_version(void);

/* ---- Yee FDTD (replaces update_efield_patches_2d / update_bfield_patches_2d,
 *      core/maxwell/cpu.py:38-79, called through MaxwellSolver2D.update_efield/bfield,
 *      core/maxwell/solver/solver.py:143-190).  Interior cells only; dt is the caller's (half)
 *      step; eps0 is scipy.constants.epsilon_0 of the host environment. */
int lpa_fdtd_e_2d(const lpa_grid *g, double dt, double eps0, void *stream);
int lpa_fdtd_b_2d(const lpa_grid *g, double dt, void *stream);
/* 3-D twins (core/maxwell/cpu.py:115-158) */
int lpa_fdtd_e_3d(const lpa_grid *g, double dt, double eps0, void *stream);
int lpa_fdtd_b_3d(const lpa_grid *g, double dt, void *stream);

/* ---- CPML absorbing layers (core/boundary/cpml.py), slab form.
 *      lpa_fdtd_*_cpml_2d : kappa-scaled Yee half steps (update_efield_cpml_2d / update_bfield_cpml_2d,
 *                           cpml.py:343-377, driven by MaxwellSolver2D.update_efield/bfield,
 *                           core/maxwell/solver/solver.py:154-190); kappa_*x [nx], kappa_*y [ny] are 1
 *                           outside the layers, so one sweep covers the reference's mix of plain and
 *                           PML patches.
 *      lpa_cpml_psi_2d    : psi recursion + field correction of ONE layer (update_psi_{x,y}_and_{e,b}_2d,
 *                           cpml.py:531-606): efield != 0 for the E form, axis 0/1 = layer normal,
 *                           [start, stop) the layer's cells along it, bcoeff / ccoeff_d [n along axis]
 *                           the coefficients of cpml.py:537-538 for this dt (host computed),
 *                           psi_a / psi_b compact arrays [stop-start][ny] (axis 0) or [nx][stop-start]
 *                           (axis 1): (psi_ey_x, psi_ez_x), (psi_ex_y, psi_ez_y), (psi_by_x, psi_bz_x),
 *                           (psi_bx_y, psi_bz_y). */
int lpa_fdtd_e_cpml_2d(const lpa_grid *g, double dt, double eps0, const double *kappa_ex,
                       const double *kappa_ey, void *stream);
int lpa_fdtd_b_cpml_2d(const lpa_grid *g, double dt, const double *kappa_bx, const double *kappa_by,
                       void *stream);
int lpa_cpml_psi_2d(const lpa_grid *g, int efield, int axis, int start, int stop, double dt,
                    const double *bcoeff, const double *ccoeff_d, double *psi_a, double *psi_b,
                    void *stream);

/* ---- laser injection through the x-min boundary (replaces _update_laser_bfields_2d,
 *      callback/laser.py:17-46, called by Laser.__call__ at stage '_laser', :109-137):
 *      ey_source / ez_source [ny] are the source fields on the boundary at the current time,
 *      rows iy in [iy_start, iy_end) are driven. */
int lpa_laser_inject_2d(const lpa_grid *g, int laserpos, double dt, double eps0, int iy_start,
                        int iy_end, const double *ey_source, const double *ez_source, void *stream);
/* the same with the sources factorised: the profiles of callback/laser.py:351-386,504-555 are amp(y) T(t) times
 * sin / cos of theta(t) + phi(y) (plane phase fronts at normal incidence, Gaussian / Laguerre-Gaussian beams), i.e.
 *      ey_source = k4[0] pc + k4[1] ps,   ez_source = k4[2] pc + k4[3] ps,   pc = amp cos(phi), ps = amp sin(phi)
 * with pc, ps [ny] fixed for the run and four numbers per step (lambdapic_amd/laser.py: Laser2D.separable): the
 * kernel evaluates the sources itself and the '_laser' stage is one launch. */
int lpa_laser_inject_sep_2d(const lpa_grid *g, int laserpos, double dt, double eps0, int iy_start, int iy_end,
                            const double *pc, const double *ps, const double *k4, void *stream);
/* fused form: the kappa-scaled update and the psi recursions of all the layers a cell lies in, in one launch
 * per field update (same operations, same order; the psi recursion of E reads only B and vice versa).
 * One descriptor per axis: kappa [n]; the low / high layer's cell range ([x0, x1) empty = none), bcoeff /
 * ccoeff_d [n] and the compact psi arrays of each layer (layouts as for lpa_cpml_psi_2d). */
typedef struct {
    const double *kappa, *bcoeff, *ccoeff_d;
    int32_t lo0, lo1, hi0, hi1;
    double *psi_a_lo, *psi_b_lo, *psi_a_hi, *psi_b_hi;
} lpa_cpml_axis;
int lpa_fdtd_e_cpml_fused_2d(const lpa_grid *g, double dt, double eps0, const lpa_cpml_axis *ax,
                             const lpa_cpml_axis *ay, void *stream);
int lpa_fdtd_b_cpml_fused_2d(const lpa_grid *g, double dt, const lpa_cpml_axis *ax, const lpa_cpml_axis *ay,
                             void *stream);
int lpa_fdtd_e_cpml_fused_3d(const lpa_grid *g, double dt, double eps0, const lpa_cpml_axis *ax,
                             const lpa_cpml_axis *ay, const lpa_cpml_axis *az, void *stream);
int lpa_fdtd_b_cpml_fused_3d(const lpa_grid *g, double dt, const lpa_cpml_axis *ax, const lpa_cpml_axis *ay,
                             const lpa_cpml_axis *az, void *stream);
/* 3-D twins: update_efield/bfield_cpml_3d (core/boundary/cpml.py:431-475), update_psi_{x,y,z}_and_{e,b}_3d
 * (:609-729; psi arrays compact: axis 0 [layer][ny][nz], axis 1 [nx][layer][nz], axis 2 [nx][ny][layer]),
 * _update_laser_bfields_3d (callback/laser.py:63-92; sources [ny][nz] over the interior nodes) */
int lpa_fdtd_e_cpml_3d(const lpa_grid *g, double dt, double eps0, const double *kappa_ex,
                       const double *kappa_ey, const double *kappa_ez, void *stream);
int lpa_fdtd_b_cpml_3d(const lpa_grid *g, double dt, const double *kappa_bx, const double *kappa_by,
                       const double *kappa_bz, void *stream);
int lpa_cpml_psi_3d(const lpa_grid *g, int efield, int axis, int start, int stop, double dt,
                    const double *bcoeff, const double *ccoeff_d, double *psi_a, double *psi_b, void *stream);
int lpa_laser_inject_3d(const lpa_grid *g, int laserpos, double dt, double eps0, int iy_start, int iy_end,
                        int iz_start, int iz_end, const double *ey_source, const double *ez_source,
                        void *stream);
/* factorised sources (see lpa_laser_inject_sep_2d); pc, ps [ny][nz] */
int lpa_laser_inject_sep_3d(const lpa_grid *g, int laserpos, double dt, double eps0, int iy_start, int iy_end,
                            int iz_start, int iz_end, const double *pc, const double *ps, const double *k4,
                            void *stream);

/* ---- zero jx jy jz rho including guards (replaces reset_current_cpu_2d/3d,
 *      core/current/cpu2d.c:19-72, cpu3d.c:185-240) */
int lpa_reset_current(const lpa_grid *g, void *stream);
/* the same for jx jy jz only: rho persists between steps when it is advanced by lpa_rho_continuity */
int lpa_reset_j(const lpa_grid *g, void *stream);

/* ---- rho from the discrete continuity equation (companion of LPA_PUSH_NO_RHO; 2-D when g->nz <= 1).
 *      lpa_rho_continuity: rho -= dt * (D-x jx + D-y jy [+ D-z jz]) with backward differences, AFTER the currents
 *        were folded (lpa_current_fold / halo exchange).  Per axis: bit set in `periodic_axes` = folded inside this
 *        slab: interior nodes only, node 0 takes node n-1 as its lower neighbour; split_x != 0 = x is cut into
 *        slabs, bit 0 / bit 1 = this slab has a left / right neighbour: the guard planes of such a face were sent
 *        away (interior nodes only on that side) and node 0 takes jx_left_plane[NY(*NZ)], the left neighbour's
 *        folded jx at its node nx-1; a face without a neighbour -- every face of an axis that is neither periodic nor
 *        split, and the outer face of a chain's end slab -- is open: its guard nodes are updated too, on the torus of
 *        the padded array, where the deposit itself lands (core/utils/cutils.h:19-26).
 *      lpa_rho_absorbed: subtract from rho what the particles listed by a LPA_PUSH_NO_RHO kernel (absorbed at an
 *        open face, mark_out_of_bound_as_dead, core/patch/sync_particles_2d.c:185-202) had deposited there:
 *        entry = {x1, y1, z1 (deposit end point in cells from node 0), q w / cell volume}; runs before the fold of
 *        the step after the absorption (the reference's rho of the absorbing step still contains the particle);
 *        count[0] is consumed (reset to 0), entries beyond `capacity` are added to count[1]. */
int lpa_rho_continuity(const lpa_grid *g, double dt, int periodic_axes, int split_x, const double *jx_left_plane,
                       void *stream);
int lpa_rho_absorbed(const lpa_grid *g, const double *list, uint32_t *count, int64_t capacity, void *stream);
/* the same with the spill array of lpa_push_params.absorbed_spill: when count[0] exceeded `capacity` the array is
 * subtracted from rho (and zeroed) as well -- nothing is lost, count[1] stays 0 */
int lpa_rho_absorbed_spill(const lpa_grid *g, const double *list, uint32_t *count, int64_t capacity, double *spill,
                           void *stream);

/* ---- periodic guard handling inside one slab (replaces sync_guard_fields_2d and
 *      sync_currents_2d with a self-neighbour table, core/patch/sync_fields2d.c:150-255,43-148).
 *      `which`: bit 0 = ex ey ez, bit 1 = bx by bz.  `axes`: bit 0 = x, bit 1 = y (, bit 2 = z):
 *      the axes that are periodic INSIDE this slab (x is excluded when x is split over ranks).
 *      lpa_guard_wrap, bit 3 of `axes` (x split over ranks): the x guard planes wrap their own y / z guards too (slabs that
 *      advance B on those planes themselves, LPA_STEP_B_EXT_*). */
int lpa_guard_wrap(const lpa_grid *g, int which, int axes, void *stream);
int lpa_current_fold(const lpa_grid *g, int axes, void *stream);

/* ---- x-face halo buffers for the slab decomposition (replaces the MPI subarray / packed-buffer
 *      exchange of core/mpi/sync_fields2d.c:365-640).  Buffers are [ncomp][ng][NY(*NZ)] doubles.
 *      side: 0 = low-x face, 1 = high-x face.
 *      pack_guard_src  : interior edge planes that become the neighbour's guard (E/B, `which` as
 *                        above)
 *      unpack_guard    : write received planes into my guard on `side`
 *      pack_current    : my guard planes of jx jy jz rho on `side`, zeroing them
 *                        (fill_currents_buf, core/mpi/sync_fields2d.c:44-74)
 *      unpack_current  : add received planes into my interior edge on `side` (:76-102) */
int lpa_halo_pack_guard_src(const lpa_grid *g, int which, int side, double *buf, void *stream);
int lpa_halo_unpack_guard(const lpa_grid *g, int which, int side, const double *buf, void *stream);
int lpa_halo_pack_current(const lpa_grid *g, int side, double *buf, void *stream);
int lpa_halo_unpack_current(const lpa_grid *g, int side, const double *buf, void *stream);
/* the same four operations on BOTH x faces in one launch (half the launches of a halo step):
 * op = LPA_HALO_PACK_GUARD_SRC / _UNPACK_GUARD (E and / or B, `which` as above) or _PACK_CURRENT /
 * _UNPACK_CURRENT (`which` ignored); a NULL buffer skips its face (open end of a slab chain). */
#define LPA_HALO_PACK_GUARD_SRC 0
#define LPA_HALO_UNPACK_GUARD 1
#define LPA_HALO_PACK_CURRENT 2
#define LPA_HALO_UNPACK_CURRENT 3
int lpa_halo_faces(const lpa_grid *g, int op, int which, double *buf_lo, double *buf_hi, void *stream);

/* ---- fused particle kernel (replaces unified_boris_pusher_cpu_2d(particles_list, fields_list,
 *      npatches, dt, q, m), core/pusher/unified/unified_pusher_2d.c:157-365): half push, TSC
 *      gather, Boris, half push, Esirkepov deposit of rho jx jy jz.
 *      lpa_push_deposit_2d      : any particle order, global-memory gather + FP64 atomics;
 *                                 processes particles [first, first+count).
 *      lpa_push_deposit_tiled_2d: tile-binned particles, E/B and J staged in LDS per tile;
 *                                 particles outside their tile's margin are appended to
 *                                 `overflow` (uint32 indices, count in overflow_count[0]) and must
 *                                 then be finished with lpa_push_deposit_list_2d.
 *      wrap flags: bit 0/1 = after the deposit, fold x/y back into the periodic box
 *      [lo, hi] of size L (Patches.sync_particles with a self neighbour,
 *      core/patch/sync_particles_2d.c:168-182) -- the fused form of the periodic migration. */
typedef struct {
    double dt, q, m;
    int32_t wrap;              /* bits 0..2: periodic fold of x, y, z into [lo, hi];
                                  bits 4..6 (LPA_ABSORB_X << axis): kill (x = y = NaN) a particle that is
                                  outside [alo, ahi] on that axis after the deposit -- the open / PML
                                  edge rule of mark_out_of_bound_as_dead
                                  (core/patch/sync_particles_2d.c:185-202, bounds core/patch/patch.py:105-148) */
    int32_t flags;             /* LPA_PUSH_* (0 = the reference's kernel: rho deposited with the currents) */
    double lo[3], hi[3];       /* global particle box: lo = -d/2, hi = L - d/2 */
    double alo[3], ahi[3];     /* absorption bounds (only read where the absorb bit is set) */
    /* optional (NULL = off; required by LPA_PUSH_NO_RHO when a face absorbs): the particles the kernel absorbs are
     * appended here (see lpa_rho_absorbed) -- in real-deposit steps too, since the NEXT step may carry rho over;
     * entries are four doubles each, absorbed_count = uint32[2] {entries, entries that did not fit}, device memory */
    double *absorbed;
    uint32_t *absorbed_count;
    int64_t absorbed_capacity;
    /* optional (NULL = off; slab ranks): the slot of every particle whose advanced x lies outside [leave_lo, leave_hi]
     * -- it now belongs to a neighbour slab -- is appended to `leavers` (uint64 entries: low 32 bits = the slot, high 32 bits
     * = 1 + the tile whose range holds the slot when the reporting kernel knew it -- the tiled kernels do --, else 0;
     * leaver_count = device uint32
     * the caller zeroes before the step's pushes; entries beyond leaver_capacity are dropped and only counted: such a
     * particle is reported again by the next step's push).  The migration pack then visits the listed slots instead of
     * scanning the edge tile columns for them (lpa_migrate_pack_list). */
    uint64_t *leavers;
    uint32_t *leaver_count;
    int64_t leaver_capacity;
    double leave_lo, leave_hi;
    /* optional companion of `absorbed` (NULL = entries beyond absorbed_capacity are only counted): a device array shaped
     * like rho into which the kernel adds, node by node, what an absorbed particle that found the list full had
     * deposited -- lpa_rho_absorbed_spill takes it out of rho together with the listed entries, so the list cannot lose
     * a particle however many are absorbed in one step */
    double *absorbed_spill;
} lpa_push_params;
#define LPA_ABSORB_X 16
/* LPA_PUSH_NO_RHO: the fused kernels deposit jx jy jz only.  Esirkepov's deposit satisfies the discrete continuity
 * equation per particle and node, (rho1 - rho0) / dt + D-x jx + D-y jy + D-z jz = 0 with backward differences
 * (current/current_deposit.h:185-268,333-440 by construction of the scheme; asserted per node in the tests), so the
 * caller advances rho with lpa_rho_continuity from the folded currents instead of paying a third (2-D) / a quarter
 * (3-D) of the LDS atomics for it.  rho then equals the deposited one up to rounding as long as every particle that
 * contributed to the old rho also contributes its current: the caller re-deposits rho for real (flags = 0, after
 * lpa_reset_current) whenever particles appear or vanish outside the kernel, and the kernel reports the particles it
 * absorbs at open faces itself (`absorbed`), whose charge lpa_rho_absorbed takes out of rho before the next update. */
#define LPA_PUSH_NO_RHO 1
/* LPA_PUSH_NO_IG (2-D resident stores): inv_gamma is a function of the momenta, 1 / sqrt(1 + u^2) -- the value the
 * Boris rotation of the reference leaves in the array (unified_pusher_2d.c:50) and reads back for the next half push
 * (:59-60).  With this flag the fused kernels recompute it from (ux, uy, uz) where the reference loads it and do not
 * write it back: two of the thirteen attribute streams of a particle-update (16 of 105 bytes) go.  The array is then
 * STALE until lpa_refresh_inv_gamma rebuilds it (same function, same bits as the fused kernel would have stored);
 * whoever reads inv_gamma -- mirrors, diagnostics, the split kernels, a checkpoint -- refreshes first.  Tiled kernel:
 * STRIPED stores with the second-pass scratch arrays and without E / B write-back; global / list kernels: always. */
#define LPA_PUSH_NO_IG 2

int lpa_push_deposit_2d(const lpa_grid *g, const lpa_particles *p, const lpa_push_params *pp,
                        int64_t first, int64_t count, void *stream);
int lpa_push_deposit_tiled_2d(const lpa_grid *g, const lpa_particles *p,
                              const lpa_push_params *pp, const lpa_tiling *t, uint32_t *overflow,
                              uint32_t *overflow_count, void *stream);
/* the same launch restricted to the `edge_cols` tile columns at each x face (LPA_PART_EDGE) or to all the
 * others (LPA_PART_INTERIOR): only edge tiles (and overflow / loose particles) deposit into the x guard planes
 * -- provided edge_cols * LPA_TILE_X - 2 cells exceed what a particle can drift between two sorts --,
 * so the slab engine pushes them first and sends the J / rho guard planes to the ring neighbours on a
 * second stream while the interior tiles are pushed (the overlap the reference gets from
 * sync_currents_start / _wait around its intra-rank work, simulation.py:1155-1188) */
#define LPA_PART_ALL 0
#define LPA_PART_EDGE 1
#define LPA_PART_INTERIOR 2
int lpa_push_deposit_tiled_part_2d(const lpa_grid *g, const lpa_particles *p,
                                   const lpa_push_params *pp, const lpa_tiling *t, uint32_t *overflow,
                                   uint32_t *overflow_count, int part, int edge_cols, void *stream);
int lpa_push_deposit_list_2d(const lpa_grid *g, const lpa_particles *p, const lpa_push_params *pp,
                             const uint32_t *list, const uint32_t *list_count, int64_t max_count,
                             void *stream);
int lpa_push_deposit_3d(const lpa_grid *g, const lpa_particles *p, const lpa_push_params *pp,
                        int64_t first, int64_t count, void *stream);
/* 3-D twins of the tiled / list forms (replace unified_boris_pusher_cpu_3d,
 * core/pusher/unified/unified_pusher_3d.c:219-436): J / rho of a 4 x 4 x 16-cell tile and its halo are
 * accumulated in LDS, E / B are gathered from global memory; particles whose deposit window leaves the
 * staged region go to `overflow` */
int lpa_push_deposit_tiled_3d(const lpa_grid *g, const lpa_particles *p,
                              const lpa_push_params *pp, const lpa_tiling *t, uint32_t *overflow,
                              uint32_t *overflow_count, void *stream);
int lpa_push_deposit_tiled_part_3d(const lpa_grid *g, const lpa_particles *p,
                                   const lpa_push_params *pp, const lpa_tiling *t, uint32_t *overflow,
                                   uint32_t *overflow_count, int part, int edge_cols, void *stream);
int lpa_push_deposit_list_3d(const lpa_grid *g, const lpa_particles *p, const lpa_push_params *pp,
                             const uint32_t *list, const uint32_t *list_count, int64_t max_count,
                             void *stream);
/* the tiled 3-D kernel for up to four species in ONE launch (the reference pushes its species one after the other,
 * simulation.py:983-990; with no callback between them the order is immaterial: they add into the same J).  A
 * workgroup per tile stages the tile's E / B image once, runs every species' particles of that tile and flushes J
 * once -- the staging is 82 KB per tile, as much as 700 particles.  p / pp / t / overflow / overflow_count are HOST
 * arrays of `nspecies` pointers; the species share the grid, dt, wrap and flags; each species' overflow list is
 * finished with lpa_push_deposit_list_3d as after the single-species form. */
int lpa_push_deposit_tiled_multi_3d(const lpa_grid *g, int32_t nspecies, const lpa_particles *const *p,
                                    const lpa_push_params *const *pp, const lpa_tiling *const *t,
                                    uint32_t *const *overflow, uint32_t *const *overflow_count, void *stream);

/* ---- split kernels of the callback-in-pusher-stage path
 *      interpolation_patches_2d (core/interpolation/cpu2d.c:71-136), boris_push_patches
 *      (core/pusher/cpu.py:11-35), push_position_patches_2d (core/pusher/cpu.py:73-91),
 *      current_deposition_cpu_2d (core/current/cpu2d.c:74-184) */
int lpa_interpolate_2d(const lpa_grid *g, const lpa_particles *p, void *stream);
int lpa_boris(const lpa_particles *p, double dt, double q, double m, void *stream);
int lpa_push_position_2d(const lpa_particles *p, double dt, void *stream);
int lpa_deposit_2d(const lpa_grid *g, const lpa_particles *p, double dt, double q, void *stream);
/* 3-D standalone twins: interpolation_patches_3d (core/interpolation/cpu3d.c:99-169) and
 * current_deposition_cpu_3d (core/current/cpu3d.c:118-183).  There is no lpa_push_position_3d: the
 * reference's PusherBase.push_position (core/pusher/pusher.py:103-110) moves particles in 2-D only,
 * so it has no working 3-D split step to mirror. */
int lpa_interpolate_3d(const lpa_grid *g, const lpa_particles *p, void *stream);
int lpa_deposit_3d(const lpa_grid *g, const lpa_particles *p, double dt, double q, void *stream);
/* periodic fold of the positions into the global box (pp->wrap, lo, hi): what Patches.sync_particles
 * does for a patch that is its own neighbour (core/patch/sync_particles_2d.c:168-182); the fused
 * kernels apply it themselves, the split path calls this after the deposit */
int lpa_wrap_positions_2d(const lpa_particles *p, const lpa_push_params *pp, void *stream);
/* 3-D twin (core/patch/sync_particles_3d.c with a self neighbour; the fused 3-D kernels apply it themselves) */
int lpa_wrap_positions_3d(const lpa_particles *p, const lpa_push_params *pp, void *stream);

/* ---- cell-index sort (replaces sort_particles_patches_2d, core/sort/cpu2d.c:220-303, as driven
 *      by ParticleSort2D.__call__, core/sort/particle_sort.py:196-211).  Out of place: `src` is
 *      binned by LPA_TILE_X x LPA_TILE_Y cell tiles into `dst` (same capacity), dead / NaN
 *      particles are dropped (they sort behind every live particle in the reference and are
 *      recycled by sync_particles); the order inside a tile is `order` (LPA_ORDER_*).  The number
 *      of live particles is written to the workspace header (lpa_sort_live_count).  `inv_gamma` may be NULL in
 *      BOTH stores (see LPA_PUSH_NO_IG): it is then not moved. */
int64_t lpa_sort_workspace_bytes(const lpa_grid *g, int64_t capacity);
/* ranks per cell the striped orders of a workspace of this capacity keep in stripes (twice the mean occupancy at full
 * capacity, a power of two in [32, 1024]); a cell's deeper particles follow cell by cell behind the stripes of its tile */
int32_t lpa_sort_stripe_ranks(const lpa_grid *g, int64_t capacity);
/* workspace size for an explicit number of striped ranks (lpa_tiling.stripe_ranks; 0 = the default rule): a store that
 * fills only part of the grid -- a solid target in an empty box -- is deeper where it is occupied than its mean */
int64_t lpa_sort_workspace_bytes_ranks(const lpa_grid *g, int64_t capacity, int32_t stripe_ranks);
/* three numbers of the last sort (device pointer inside the workspace, next to lpa_sort_live_count: one read-back serves
 * all): [0] the largest number of particles one cell held and [1] the particles that lay beyond the striped ranks (striped
 * orders), [2] the tiles that hold at least one particle */
const int32_t *lpa_sort_deepest_cell(void *workspace);
int lpa_sort_tiles_2d(const lpa_grid *g, const lpa_particles *src, const lpa_particles *dst,
                      void *workspace, int64_t workspace_bytes, int32_t block_particles,
                      int32_t order, lpa_tiling *out, void *stream);
/* 3-D twin (replaces sort_particles_patches_3d, core/sort/cpu3d.c): tiles of LPA_TILE3_X x LPA_TILE3_Y x
 * LPA_TILE3_Z cells; the workspace size comes from lpa_sort_workspace_bytes with the 3-D grid */
int lpa_sort_tiles_3d(const lpa_grid *g, const lpa_particles *src, const lpa_particles *dst,
                      void *workspace, int64_t workspace_bytes, int32_t block_particles,
                      int32_t order, lpa_tiling *out, void *stream);
/* the same sorts with every particle binned where it will be `ahead` seconds down its straight path (x + v * ahead; 0 =
 * the plain sorts).  For stores the caller re-sorts every T steps because their particles outrun the tile margin: binned
 * for the middle of the interval (ahead = T dt / 2) the order stays usable about twice as long.  Any binning gives the
 * same results from the tiled kernels (what does not fit a tile's staged region takes their overflow list) */
int lpa_sort_tiles_ahead_2d(const lpa_grid *g, const lpa_particles *src, const lpa_particles *dst,
                            void *workspace, int64_t workspace_bytes, int32_t block_particles,
                            int32_t order, lpa_tiling *out, double ahead, void *stream);
int lpa_sort_tiles_ahead_3d(const lpa_grid *g, const lpa_particles *src, const lpa_particles *dst,
                            void *workspace, int64_t workspace_bytes, int32_t block_particles,
                            int32_t order, lpa_tiling *out, double ahead, void *stream);
/* number of live particles after the last sort (device pointer inside the workspace); for LPA_ORDER_PADDED: the slots
 * of the order (live + holes + the 64-slot rounding of every tile -- up to 4/3 n + 63 per tile) */
const int32_t *lpa_sort_live_count(void *workspace);
/* device flag of the last sort, non-zero = NOTHING was moved: bit 0 = the sorted order needs more slots than dst->n
 * (only a padded order can: size its stores for 4/3 n + 64 per tile), bit 1 = more work blocks than the table holds.
 * The caller reads it together with the live count and treats it as an error. */
const int32_t *lpa_sort_overflow(void *workspace);

/* ---- particle ownership along x for the slab decomposition (replaces get_npart_to_extend_2d +
 *      fill_particles_from_boundary_2d, core/patch/sync_particles_2d.c:204-518 and the MPI twins
 *      core/mpi/sync_particles_2d.c:274-770: count message + AoS payload message per boundary).
 *      One fixed-size message per face carries its own count, so no host round trip is needed:
 *      buffer = 1 + LPA_MIG_NATTR*capacity doubles; [0] = count (int64 bit pattern), then SoA
 *      [attr][capacity].
 *      pack  : live particles with x < xlo / x > xhi are copied to buf_lo / buf_hi and killed
 *              (x = y = NaN, sync_particles_2d.c:185-202).  A particle that does not fit stays
 *              where it is and leaves one step later; `surplus` (device int32, may be NULL) is
 *              incremented by the number of such particles -- the caller zeroes and reads it.
 *      unpack: append the received particles to the arrival area [first_slot, first_slot +
 *              area_capacity) at the device-side cursor, adding shift_x to x (periodic wrap at
 *              the global edge, sync_particles_2d.c:168-182).  cursor > area_capacity = overflow. */
/* Free-slot stacks of a tile-ordered store (optional): lpa_migrate_pack_edges_x records the slots its
 * leavers free, per edge tile; lpa_migrate_unpack_tiled hands them to the arrivals of the same tile.  `count`
 * has 2 * edge_cols * tiles_y (* tiles_z) entries (the tiles of the edge_cols columns at the low face, then
 * those at the high face), `slot` `depth` entries per tile.  The caller zeroes `count` after every sort;
 * edge_cols must cover every column a leaver can come from until the next sort. */
typedef struct {
    int32_t *count;
    int32_t *slot;
    int32_t edge_cols, depth;
} lpa_free_slots;

#define LPA_MIG_NATTR 9 /* x y z ux uy uz inv_gamma w id */
int lpa_migrate_pack_x(const lpa_particles *p, double xlo, double xhi, double *buf_lo,
                       double *buf_hi, int64_t capacity, int32_t *surplus, void *stream);
/* the same scan restricted to the particles that can have left a tile-ordered store since its sort: the
 * `edge_cols` tile columns next to each x face (tile index is x-slowest) and the loose particles behind
 * t->n_sorted.  The caller chooses edge_cols from the age of the order (c*dt*age / tile width, rounded
 * up); the ranges are read from t->tile_off on the device.  2-D and 3-D tilings.  `fs` (may be NULL): record
 * the freed slots, see lpa_free_slots. */
int lpa_migrate_pack_edges_x(const lpa_particles *p, const lpa_tiling *t, int32_t edge_cols, double xlo,
                             double xhi, double *buf_lo, double *buf_hi, int64_t capacity,
                             const lpa_free_slots *fs, int32_t *surplus, void *stream);
/* the same pack over a list of slots (lpa_push_params.leavers, written by the push kernels of this step): visits
 * min(*list_count, list_capacity) slots; a listed slot whose particle is not (or no longer) outside [xlo, xhi] is left
 * alone.  `t` / `fs` as for lpa_migrate_pack_edges_x (both may be NULL: no free-slot bookkeeping). */
int lpa_migrate_pack_list(const lpa_particles *p, const lpa_tiling *t, const uint64_t *list, const uint32_t *list_count,
                          int64_t list_capacity, double xlo, double xhi, double *buf_lo, double *buf_hi,
                          int64_t capacity, const lpa_free_slots *fs, int32_t *surplus, void *stream);
/* unpack for a tile-ordered store with free-slot stacks: an arrival whose tile (from its position on grid
 * `g`) has a recorded free slot takes it and is pushed by the tiled kernel from the next step on; the others are
 * appended to the arrival area exactly like lpa_migrate_unpack. */
int lpa_migrate_unpack_tiled(const lpa_particles *p, const lpa_grid *g, const lpa_tiling *t,
                             const lpa_free_slots *fs, int64_t first_slot, int64_t area_capacity,
                             int32_t *cursor, const double *buf, int64_t capacity, double shift_x, void *stream);
int lpa_migrate_unpack(const lpa_particles *p, int64_t first_slot, int64_t area_capacity,
                       int32_t *cursor, const double *buf, int64_t capacity, double shift_x,
                       void *stream);

/* ---- patch-list drop-ins: the reference's own data model (lists of per-patch arrays in the WRAPPED guard
 *      layout of core/fields.py:24-27, neighbour tables Patch.neighbor_ipatch[8] in Boundary2D order,
 *      core/patch/patch.py:24-35) kept on the device.  `arrays` is a DEVICE table of device pointers,
 *      [npatches][ncomp] (ncomp = 4 for the currents: jx jy jz rho); `neighbor_ipatch` a device int64
 *      [npatches][8], -1 = no neighbour.
 *      lpa_sync_guard_fields_2d replaces sync_guard_fields_2d(fields_list, patches_list, attrs, npatches, nx,
 *        ny, ng) (core/patch/sync_fields2d.c:150-255): guard <- the neighbour's interior edge, 8 neighbours.
 *      lpa_sync_currents_2d replaces sync_currents_2d(fields_list, patches_list, npatches, nx, ny, ng)
 *        (core/patch/sync_fields2d.c:43-148): interior edge += the neighbour's guard (added in the
 *        reference's boundary order: bit-identical sums), consumed guards zeroed. */
int lpa_sync_guard_fields_2d(double *const *arrays, int32_t ncomp, const int64_t *neighbor_ipatch,
                             int32_t npatches, int32_t nx, int32_t ny, int32_t ng, void *stream);
int lpa_sync_currents_2d(double *const *arrays, const int64_t *neighbor_ipatch, int32_t npatches, int32_t nx,
                         int32_t ny, int32_t ng, void *stream);

/* ---- particle ownership between the patches of a list: the two halves of Patches.sync_particles
 *      (core/patch/patch.py:705-742).  Tables are DEVICE arrays of per-patch device pointers; `bounds` =
 *      [npatches][4] xmin xmax ymin ymax WITH the half cell the reference adds (sync_particles_2d.c:236-241).
 *      lpa_sync_particles_count_2d: the counting loop of get_npart_to_extend_2d (core/patch/sync_particles_2d.c:37-84,
 *        204-283).  `xy` = [npatches][2] x and y pointers; npart_outgoing[npatches][8] (Boundary2D order) and
 *        ndead[npatches] (device int64) are zeroed and filled; the caller derives npart_incoming / npart_to_extend /
 *        npart_alive from them exactly like :285-318 and grows its arrays.
 *      lpa_sync_particles_fill_2d: fill_particles_from_boundary_2d (:322-518).  `attrs` = [npatches][nattrs] pointers
 *        (x at iattr_x, y at iattr_y).  The leavers of the 8 neighbours, in (boundary, index) order, go into the
 *        receiver's dead slots in ascending order -- the reference's own placement, slot for slot --, with +- L on
 *        x / y like handle_periodic (:168-182); then every live particle outside its patch's bounds dies (x = y = NaN,
 *        :185-202).  Workspace: lpa_sync_particles_workspace_bytes. */
int lpa_sync_particles_count_2d(const double *const *xy, const uint8_t *const *is_dead, const int64_t *npart,
                                const double *bounds, int32_t npatches, int64_t max_npart,
                                int64_t *npart_outgoing, int64_t *ndead, void *stream);
int64_t lpa_sync_particles_workspace_bytes(int32_t npatches, int64_t max_npart);
int lpa_sync_particles_fill_2d(double *const *attrs, int32_t nattrs, int32_t iattr_x, int32_t iattr_y,
                               uint8_t *const *is_dead, const int64_t *npart, const double *bounds,
                               const int64_t *neighbor_ipatch, const int64_t *npart_incoming,
                               const int64_t *npart_outgoing, int32_t npatches, int64_t max_npart,
                               double xmin_global, double xmax_global, double ymin_global, double ymax_global,
                               double dx, double dy, void *workspace, int64_t workspace_bytes, void *stream);

/* ---- 3-D twins of the patch-list drop-ins: neighbor_ipatch is [npatches][26] in Boundary3D order
 *      (core/patch/patch.py:37-69: 6 faces, 12 edges xy / xz / yz, 8 vertices), arrays double[NX][NY][NZ] in the
 *      wrapped guard layout.
 *      lpa_sync_guard_fields_3d replaces sync_guard_fields_3d(fields_list, patches_list, attrs, npatches, nx, ny, nz, ng)
 *        (core/patch/sync_fields3d.c:350-612); lpa_sync_currents_3d replaces sync_currents_3d(fields_list,
 *        patches_list, npatches, nx, ny, nz, ng) (:84-348; the neighbours' guards are added in Boundary3D order like the
 *        reference's sweep, consumed guards zeroed).
 *      lpa_sync_particles_count_3d / _fill_3d replace get_npart_to_extend_3d + fill_particles_from_boundary_3d
 *        (core/patch/sync_particles_3d.c:365-482,484-700): `xyz` = [npatches][3] pointers, `bounds` = [npatches][6]
 *        xmin xmax ymin ymax zmin zmax WITH the half cell (:392-399), npart_outgoing [npatches][26]; global_min /
 *        global_max / cell = host double[3]; slot placement and +- L as in the 2-D twins; mark_out_of_bound_as_dead
 *        of the 3-D file (:324-345) also blanks the positions of slots that are dead already.  Workspace:
 *        lpa_sync_particles_workspace_bytes. */
int lpa_sync_guard_fields_3d(double *const *arrays, int32_t ncomp, const int64_t *neighbor_ipatch,
                             int32_t npatches, int32_t nx, int32_t ny, int32_t nz, int32_t ng, void *stream);
int lpa_sync_currents_3d(double *const *arrays, const int64_t *neighbor_ipatch, int32_t npatches, int32_t nx,
                         int32_t ny, int32_t nz, int32_t ng, void *stream);
int lpa_sync_particles_count_3d(const double *const *xyz, const uint8_t *const *is_dead, const int64_t *npart,
                                const double *bounds, int32_t npatches, int64_t max_npart,
                                int64_t *npart_outgoing, int64_t *ndead, void *stream);
int lpa_sync_particles_fill_3d(double *const *attrs, int32_t nattrs, int32_t iattr_x, int32_t iattr_y,
                               int32_t iattr_z, uint8_t *const *is_dead, const int64_t *npart, const double *bounds,
                               const int64_t *neighbor_ipatch, const int64_t *npart_incoming,
                               const int64_t *npart_outgoing, int32_t npatches, int64_t max_npart,
                               const double *global_min, const double *global_max, const double *cell,
                               void *workspace, int64_t workspace_bytes, void *stream);

/* ---- bucket sort with the reference's bookkeeping, one patch per call: replaces the body of
 *      sort_particles_patches_2d / _3d (core/sort/cpu2d.c:220-303, cpu3d.c) = calculate_bucket_index (:9-54:
 *      bucket = floor((r - r0) / d) per axis, out of range -> last bucket or clamped when reverse_x, a dead
 *      slot inherits the bucket of the slot before it) + calculate_bucket_bound (:78-91) + bucket_sort
 *      (:108-189: in-place permutation of all attributes and is_dead restricted to the misplaced slots; an
 *      already sorted store moves nothing).  Outputs: bucket_count / bucket_bound_min / bucket_bound_max
 *      (device int64[nx*ny*nz], what the collision module reads) and *nbuf (device int64) = slots moved.
 *      Which misplaced particle of a bucket lands in which of its free slots is implementation defined
 *      (here: atomic order), as in the reference.  `attrs` is a HOST array of `nattrs` device pointers
 *      (x, y, z are moved whether or not they are listed); nz = 1 selects 2-D (z, dz, z0 unused). */
int64_t lpa_bucket_sort_workspace_bytes(int64_t npart, int64_t nbuckets);
int lpa_bucket_sort(double *x, double *y, double *z, uint8_t *is_dead, double *const *attrs, int32_t nattrs,
                    int64_t npart, int64_t nx, int64_t ny, int64_t nz, double dx, double dy, double dz,
                    double x0, double y0, double z0, int32_t reverse_x, int64_t *bucket_count,
                    int64_t *bucket_bound_min, int64_t *bucket_bound_max, void *workspace,
                    int64_t workspace_bytes, int64_t *nbuf, void *stream);

/* ---- slab-to-slab transport (replaces the MPI point-to-point traffic of MPIManager2D/3D, core/mpi/mpi_manager.py:96-298
 *      and core/mpi/sync_fields2d.c:365-640: Isend / Irecv per (patch, boundary, attribute) on duplicated communicators).
 *      A communicator connects this rank with its two x neighbours in a ring (periodic x) or chain (open x).  One
 *      EXCHANGE is a grouped set of face messages enqueued on a stream, no host synchronisation:
 *        send_lo -> the left neighbour (arrives there as its recv_hi), send_hi -> the right neighbour (its recv_lo).
 *      Four counts in doubles, one per buffer; what one rank sends through a face must be what its neighbour receives
 *      through the opposite one (n_send_lo here == n_recv_hi of the left neighbour, ...).  A zero count skips that
 *      buffer, a face without a neighbour (chain end) is skipped.  Every rank must pass the same number of messages in
 *      the same order -- messages between one pair of ranks match in posting order (RCCL has no tags): per message the
 *      sends are posted (hi, lo), the receives (lo, hi).
 *      Kinds:
 *        LPA_COMM_RCCL      ncclSend / ncclRecv inside one ncclGroupStart / End per exchange, on the caller's stream,
 *                           device-to-device over xGMI.  The RCCL library is dlopen'ed (`librccl_path`, NULL =
 *                           "librccl.so": the one the process has loaded already, e.g. PyTorch's) -- the library has no
 *                           link-time dependency on it.  rank 0 makes the 128-byte id (lpa_comm_unique_id), the caller
 *                           distributes it (any side channel) and every rank calls lpa_comm_create_rccl.
 *        LPA_COMM_LOOPBACK  one process plays EVERY rank of a ring of identical slabs: what leaves through the high face
 *                           comes back through the low one and vice versa, one copy kernel per exchange (size-1 periodic
 *                           ring; with `size` = 2 the engines treat the slab as rank 0 of a 2-slab ring whose other slab is
 *                           its translated copy -- every kernel of the N > 1 path runs, no wire). */
typedef struct lpa_comm lpa_comm;
#define LPA_COMM_RCCL 1
#define LPA_COMM_LOOPBACK 2
typedef struct {
    const double *send_lo, *send_hi;
    double *recv_lo, *recv_hi;
    int64_t n_send_lo, n_send_hi, n_recv_lo, n_recv_hi;
} lpa_face_msg;
int lpa_comm_unique_id(void *id128, const char *librccl_path);
int lpa_comm_create_rccl(lpa_comm **out, const void *id128, int32_t rank, int32_t size, int32_t periodic,
                         const char *librccl_path);
int lpa_comm_create_loopback(lpa_comm **out, int32_t size, int32_t periodic);
int lpa_comm_destroy(lpa_comm *c);
/* kind, rank, size, left, right (-1 = none), version of the transport library (NCCL_VERSION_CODE form; 0 = loopback) */
int lpa_comm_info(const lpa_comm *c, int32_t info[6]);
int lpa_comm_exchange(lpa_comm *c, const lpa_face_msg *msgs, int32_t nmsgs, void *stream);

/* ---- one time step of ONE slab in one host call: the no-callback stage sequence of Simulation.run
 *      (simulation/simulation.py:937-1122) -- E half step + E guards, B half step + B guards, current reset, fused push +
 *      deposit of every species (tiled kernel + overflow list + loose particles, or the global kernel for an unsorted
 *      store), current fold, B half step, ['_laser' stage: between LPA_STAGE_B2 and LPA_STAGE_B2_GUARD the caller may
 *      inject], B guards, E half step + E guards.  Enqueues exactly the launches the per-stage entry points above
 *      would, in that order; `first_stage .. last_stage` (inclusive) selects a sub-range, so a caller with a callback
 *      at some stage splits the step there.  Not included: the sort (it needs the host for the live count -- the
 *      caller sorts before the step when due) and anything a callback does.
 *      `continuity` != 0: this is a step between two real deposits, see LPA_PUSH_NO_RHO (the kernels skip rho,
 *      LPA_STAGE_RESET keeps it and takes out the absorbed particles' charge, LPA_STAGE_FOLD advances it).
 *      ev_start / ev_stop (optional): hipEvent_t recorded on `stream` around the species' tiled launch.
 *
 *      SLAB RANKS (`slab` != NULL: one of several x slabs, mpi.sync_* of simulation.py:948-960,1043-1080,1104-1118): the
 *      guard stages also exchange the x faces through slab->comm -- E / B planes straight from and into the field
 *      arrays (x planes are contiguous: no pack / unpack launch) --, LPA_STAGE_FOLD sends the J / rho guard planes and
 *      every species' leavers in ONE exchange (guard planes straight from the arrays into slab->cur_r_*; they are added to the
 *      interior edge and the sent planes zeroed by one launch; arrivals are unpacked by one launch per species), and
 *      on continuity steps the left neighbour's folded jx plane rides with the B planes of LPA_STAGE_B2_GUARD, after
 *      which rho is advanced (instead of at LPA_STAGE_FOLD; see rho_exchange and LPA_STEP_B_EXT_* for the form without B
 *      messages: two rounds per step).  Stores must be tile ordered (arrival area behind
 *      n_sorted).  With slab->comm == NULL the caller moves the faces itself (another transport): the stages then do
 *      the work of a single slab with `local_axes` and the caller packs / exchanges / unpacks between sub-ranges. */
typedef struct {
    double *s_lo, *s_hi, *r_lo, *r_hi;  /* face messages of this species: 1 + LPA_MIG_NATTR * capacity doubles each */
    int32_t *cursor;                    /* arrival-area cursor (device) */
    int32_t *surplus;                   /* leavers that did not fit a message (device, may be NULL) */
    const lpa_free_slots *fs;           /* NULL: arrivals go to the arrival area only */
    int64_t area_capacity;
    int32_t edge_cols;                  /* leaver scan: tile columns at each face (lpa_migrate_pack_edges_x), 0 = all slots */
    int32_t reserved_;
    /* overlapped steps (lpa_step_slab.overlap_cols > 0): the edge part of the tiled push runs on a second stream beside the
     * interior part and needs an overflow list and counter of its own; optional event pair around its launch */
    uint32_t *overflow_edge, *overflow_edge_count;
    void *ev_edge_start, *ev_edge_stop;
} lpa_step_migrate;

typedef struct {
    lpa_particles p;            /* the store as of this step (p.n = slots in use) */
    const lpa_tiling *t;        /* NULL: unsorted store, everything through the global kernel */
    int64_t n_sorted;           /* slots [0, n_sorted) are tile ordered, [n_sorted, p.n) loose */
    lpa_push_params pp;         /* q, m, wrap, lo / hi, alo / ahi; dt, flags and the absorbed list are set by lpa_step */
    uint32_t *overflow, *overflow_count;
    void *ev_start, *ev_stop;
    lpa_step_migrate mig;       /* slab ranks only */
} lpa_step_species;

typedef struct {
    lpa_comm *comm;             /* NULL: the caller exchanges between sub-ranges */
    double xlo, xhi;            /* a particle with x < xlo / x > xhi leaves through the low / high face */
    double shift_lo, shift_hi;  /* added to x of what arrives through the low / high face (periodic wrap at the box ends) */
    int64_t migrate_capacity;   /* slots per particle face message = its SoA stride: 1 + LPA_MIG_NATTR * migrate_capacity doubles
                                   travel per species and face.  May be less than the buffers hold (the engines send a
                                   window sized from the counts they saw: dist.MigrateWindowMixin); both neighbours must use
                                   the same value in the same step */
    double *cur_r_lo, *cur_r_hi;    /* 4 * ng * plane doubles each: the neighbours' J / rho guard planes */
    double *jx_left_plane;          /* plane doubles (continuity steps: the left neighbour's folded jx at its node nx-1) */
    int32_t rho_exchange;           /* every step, all ranks alike -- 1: the left neighbour's folded jx plane travels with the
                                       B planes of LPA_STAGE_B2_GUARD, after which rho is advanced; 2: each rank's OWN jx deposit
                                       on its last node plane travels to the right with the J / rho guard planes of
                                       LPA_STAGE_FOLD into jx_left_plane, the fold forms the neighbour's folded plane from it in
                                       the jx guard plane at node -1 (the same sums, bit for bit) and rho is advanced at once --
                                       for steps whose B guard planes do not travel (LPA_STEP_B_EXT_*) */
    int32_t overlap_cols;           /* > 0 (and LPA_STAGE_PUSH .. LPA_STAGE_FOLD in one call, every store tile ordered): the
                                       `overlap_cols` tile columns at each x face (+ overflow list + arrival area: everything
                                       that can deposit into the x guard planes or leave the slab) are pushed first, on the
                                       communicator's second stream (high priority), followed there by the leaver pack and the
                                       J / rho + particle exchange -- while the main stream pushes the interior tiles (the
                                       reference's sync_currents_start .. intra-rank work .. _wait bracket,
                                       simulation.py:1155-1188); the streams join before the fold.  The caller chooses
                                       overlap_cols * tile width >= the drift since the sort + 4 cells (engines: edge_columns) */
} lpa_step_slab;

typedef struct {
    lpa_grid grid;
    int32_t dim;                /* 2 or 3 */
    int32_t local_axes;         /* bit a: axis a is periodic inside this slab (guard wrap, current fold) */
    double dt, eps0;
    const lpa_cpml_axis *e_axes[3], *b_axes[3];   /* all NULL: plain Yee update; else the fused CPML descriptors */
    int32_t nspecies, continuity;
    int32_t fuse_species;       /* 3-D: push all tile-ordered species with ONE lpa_push_deposit_tiled_multi_3d launch */
    int32_t flags;              /* LPA_STEP_* */
    const lpa_step_species *species;
    double *absorbed;           /* see lpa_push_params.absorbed (NULL: no face absorbs) */
    uint32_t *absorbed_count;
    int64_t absorbed_capacity;
    const lpa_step_slab *slab;  /* NULL: single slab */
    double *absorbed_spill;     /* see lpa_push_params.absorbed_spill (may be NULL) */
} lpa_step_desc;

/* LPA_STEP_DEFER_E2_GUARDS: LPA_STAGE_E2 leaves the E guard cells stale (no wrap, no exchange).  For a caller that runs
 * several steps back to back with nothing reading E in between: the next step starts with another E half step followed by
 * its own guard stage (simulation.py:1112-1118, then :946-952 of the following step) and the E update itself never reads
 * E guards -- one launch and, between slabs, one message round less per step.  The LAST step of such a run must not set it. */
#define LPA_STEP_DEFER_E2_GUARDS 1
/* LPA_STEP_DEFER_E2: LPA_STAGE_E2 is left out altogether -- the caller's NEXT lpa_step starts with LPA_STEP_E1_DOUBLE, whose
 * LPA_STAGE_E1 applies both half steps in one sweep (B and J are the same for both; a cell's E update reads no other cell's
 * E, so two sequential updates in registers are the two sweeps bit for bit, for a third of the pair's E traffic) before its
 * guard stage.  Between the two calls E is half a step behind: nothing may read it (engines: run_steps). */
#define LPA_STEP_DEFER_E2 2
#define LPA_STEP_E1_DOUBLE 4
/* LPA_STEP_B_EXT_LO / _HI: this slab has a neighbour at its low / high x face and the B half steps advance the x guard planes
 * there themselves (ng planes low, ng - 1 high) instead of receiving them: a B update reads E at its node and one node up, the
 * E guard planes are current after every E guard stage, so the guard values are the ones the neighbour computes for its
 * interior, bit for bit.  With either flag set the B guard stages exchange nothing (slab ranks: two message rounds per step
 * instead of four -- E, J + rho + particles); use lpa_step_slab.rho_exchange = 2 with it.  The psi arrays of the y / z CPML
 * layers (lpa_cpml_axis) must then hold ng extra x rows in front of and behind the nx the pointers address, and whoever
 * changes B outside lpa_step (initial fields, an injection within ng + 1 nodes of such a face, a window shift) leaves all ng
 * guard planes current, as lpa_halo_faces / a shift do. */
#define LPA_STEP_B_EXT_LO 8
#define LPA_STEP_B_EXT_HI 16
/* LPA_STEP_E_ROUND_IN_LINE: an overlapped slab step (lpa_step_slab.overlap_cols > 0, LPA_STEP_B_EXT_*) that starts at
 * LPA_STAGE_E1 sends its E guard planes on the communicator's second stream too, followed there by the rows of the B half
 * step that read them, beside the rest of that sweep, the reset and the interior tiles on the caller's stream: BOTH message
 * rounds of the step are hidden behind the interior push.  This flag keeps the E round on the caller's stream (A/B). */
#define LPA_STEP_E_ROUND_IN_LINE 32
/* LPA_STEP_SEPARATE_UNPACK: slab ranks fold the received J / rho planes and seat every species' arrivals in ONE launch (up to
 * four species); this flag keeps the launches apart (A/B). */
#define LPA_STEP_SEPARATE_UNPACK 64
/* LPA_STEP_SEPARATE_PACK: on slab ranks with leaver lists the launch that pushes what the tiled kernel left to global memory
 * (overflow list + arrival area) also packs the step's leavers -- the tiled kernel's list and, on the spot, those of the
 * particles it pushes itself; this flag keeps the pack a launch of its own (A/B). */
#define LPA_STEP_SEPARATE_PACK 128
/* LPA_STEP_SEPARATE_TAILS: the current reset of LPA_STAGE_RESET rides in the launch of LPA_STAGE_B1's sweep and the rho
 * continuity update of LPA_STAGE_FOLD in that of LPA_STAGE_B2's when the stages run in one call (each is a ~5 us launch that
 * touches nothing the sweep reads or writes); this flag keeps them launches of their own (A/B). */
#define LPA_STEP_SEPARATE_TAILS 256
#define LPA_STAGE_E1 0
#define LPA_STAGE_B1 1
#define LPA_STAGE_RESET 2
#define LPA_STAGE_PUSH 3
#define LPA_STAGE_FOLD 4
#define LPA_STAGE_B2 5
#define LPA_STAGE_B2_GUARD 6
#define LPA_STAGE_E2 7
int lpa_step(const lpa_step_desc *d, int first_stage, int last_stage, void *stream);

/* ---- diagnostics the parity contract is stated on (field energy, kinetic energy, total
 *      charge; reference tests/test_numerical_heating.py:19-50).  out[] is device memory and is
 *      accumulated into (zero it first).
 *      field: out[0] += sum_interior eps0/2 E^2 dV, out[1] += sum_interior B^2/(2 mu0) dV,
 *             out[2] += sum_interior rho dV, out[3..5] += sum_interior jx jy jz
 *      particles: out[0] += sum w (1/inv_gamma - 1) m c^2, out[1] += number alive */
int lpa_diag_fields(const lpa_grid *g, double eps0, double mu0, double *out, void *stream);
int lpa_diag_particles(const lpa_particles *p, double m, double *out, void *stream);
/* inv_gamma[i] = 1 / sqrt(1 + ux[i]^2 + uy[i]^2 + uz[i]^2) for the slots [first, first + count): what the fused kernels
 * would have stored had they not been run with LPA_PUSH_NO_IG (dead slots: whatever their momenta give) */
int lpa_refresh_inv_gamma(const lpa_particles *p, int64_t first, int64_t count, void *stream);

/* ---- self test of the wave-level reduce-scatter used by the tiled deposit: in[64][64] doubles
 *      (value index, lane) -> out[lane] = sum over lanes of in[lane][.]; one wave. */
int lpa_selftest_wave_reduce(const double *in, double *out, void *stream);
/* self test of the neighbour exchange of the cooperative deposit (DPP wave_shr:1 / wave_shl:1): in[64] ->
 * out[0..63] = value of lane - 1 (0 for lane 0), out[64..127] = value of lane + 1 (0 for lane 63); one wave. */
int lpa_selftest_wave_shift(const double *in, double *out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* LAMBDAPIC_AMD_H */

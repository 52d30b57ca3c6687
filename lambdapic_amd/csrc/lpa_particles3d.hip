// lpa_particles3d.hip -- 3-D fused particle kernels: half push, 27-point TSC gather on the staggered Yee
// grid, Boris, half push, 3-D Esirkepov deposit.  Two forms: global memory (any particle order, FP64
// global atomics on the torus; also the overflow-list form) and LDS-tiled (tile-sorted particles, E / B and
// J / rho of a 4 x 4 x 16-cell tile staged in LDS).
// Restates unified_boris_pusher_cpu_3d (core/pusher/unified/unified_pusher_3d.c:219-436) and
// current_deposit_3d_fast (core/current/current_deposit.h:275-440).
#include "lpa_common.hpp"
#include "lpa_migrate.hpp"

struct PushK3 {
    double dt, q, efactor, bfactor, cdt_half;
    int wrap;
    double lo[3], hi[3], alo[3], ahi[3];
    // loop invariants of the tiled kernel, computed on the host with the same IEEE operations the kernel used to
    // do: as kernel arguments they sit in SGPRs -- computed in the kernel they are VALU results that occupied
    // 14 VGPRs for the whole particle loop, spilled to scratch and reloaded 13 times per particle
    double inv_d[3];        // 1 / dx, 1 / dy, 1 / dz
    double c_rho, c_j[3];   // q / (dx dy dz),  q / (dy dz dt), q / (dx dz dt), q / (dx dy dt)
    // LPA_PUSH_NO_RHO: rho is not deposited (the caller advances it with lpa_rho_continuity).  `absorbed` (optional,
    // any mode): particles absorbed at an open face are reported there, so that lpa_rho_absorbed can take their
    // charge out of a rho that is carried over to the next step
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

// the deposit end point (cells from node 0) and charge density factor of a particle that was just absorbed
__device__ __forceinline__ void report_absorbed(const GridV &g, const PushK3 &k, double o1x, double o1y, double o1z, double cd) {
    const uint32_t slot = atomicAdd(k.absorbed_count, 1u);
    if ((long)slot < k.absorbed_cap) {
        double *e = k.absorbed + 4 * (long)slot;
        e[0] = o1x; e[1] = o1y; e[2] = o1z; e[3] = cd;
    } else if (k.absorbed_spill) {      // the list is full: see report_absorbed_2d
        spread_tsc(g, k.absorbed_spill, o1x, o1y, o1z, cd);
    }
}

struct GIdx3 { long r[3]; int c[3]; int d[3]; };

__device__ __forceinline__ void gidx3(GIdx3 &o, int ix, int iy, int iz, const GridV &g) {
#pragma unroll
    for (int a = 0; a < 3; a++) {
        o.r[a] = (long)torus(ix - 1 + a + g.ng, g.NX) * g.NY * g.NZ;
        o.c[a] = torus(iy - 1 + a + g.ng, g.NY) * g.NZ;
        o.d[a] = torus(iz - 1 + a + g.ng, g.NZ);
    }
}

// evaluation order of interp_field_fast_3d (unified_pusher_3d.c:111-143): z outermost, x innermost
__device__ __forceinline__ double gather27_g(const double *__restrict__ f, const GIdx3 &o,
                                             const double fx[3], const double fy[3],
                                             const double fz[3]) {
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        double pl = 0.0;
#pragma unroll
        for (int j = 0; j < 3; j++) {
            long b = o.c[j] + o.d[k];
            pl += fy[j] * (fx[0] * f[o.r[0] + b] + fx[1] * f[o.r[1] + b] + fx[2] * f[o.r[2] + b]);
        }
        acc += fz[k] * pl;
    }
    return acc;
}

__device__ __forceinline__ void gather_global_3d(const GridV &g, double xo, double yo, double zo,
                                                 double eb[6]) {
    int ix1 = ifloor(xo + 0.5), ix2 = ifloor(xo);
    int iy1 = ifloor(yo + 0.5), iy2 = ifloor(yo);
    int iz1 = ifloor(zo + 0.5), iz2 = ifloor(zo);
    double gx[3], hx[3], gy[3], hy[3], gz[3], hz[3];
    tsc3(ix1 - xo, gx); tsc3(ix2 - xo + 0.5, hx);
    tsc3(iy1 - yo, gy); tsc3(iy2 - yo + 0.5, hy);
    tsc3(iz1 - zo, gz); tsc3(iz2 - zo + 0.5, hz);
    GIdx3 a;  // stagger table: unified_pusher_3d.c:190-195
    gidx3(a, ix2, iy1, iz1, g); eb[0] = gather27_g(g.ex, a, hx, gy, gz);
    gidx3(a, ix1, iy2, iz1, g); eb[1] = gather27_g(g.ey, a, gx, hy, gz);
    gidx3(a, ix1, iy1, iz2, g); eb[2] = gather27_g(g.ez, a, gx, gy, hz);
    gidx3(a, ix1, iy2, iz2, g); eb[3] = gather27_g(g.bx, a, gx, hy, hz);
    gidx3(a, ix2, iy1, iz2, g); eb[4] = gather27_g(g.by, a, hx, gy, hz);
    gidx3(a, ix2, iy2, iz1, g); eb[5] = gather27_g(g.bz, a, hx, hy, gz);
}

// 3-D Esirkepov deposit of one particle on the 4x4x4 window (factored form of
// current_deposit_3d_fast, current/current_deposit.h:293-321); `sink(i, j, k, djx, djy, djz, drho)`
// adds the contributions of window cell (i, j, k).
template <class Sink>
__device__ __forceinline__ void esirkepov_3d(const AxisW &ax, const AxisW &ay, const AxisW &az, double w,
                                             double q, double dx, double dy, double dz, double dt,
                                             Sink &&sink) {
    const double one_third = 0.3333333333333333;  // core/utils/cutils.h:18
    double cd = (q / (dx * dy * dz)) * w;
    double fdx_ = (q / (dy * dz * dt)) * w;
    double fdy_ = (q / (dx * dz * dt)) * w;
    double fdz_ = (q / (dx * dy * dt)) * w;
    double jx_run[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) jx_run[a][b] = 0.0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        double a_x = ax.S0[i] + 0.5 * ax.DS[i];
        double c_x = 0.5 * ax.S0[i] + one_third * ax.DS[i];
        double fdx = fdx_ * ax.DS[i];
        bool xz = ax.tail_zero && i == 3;
        double jy_run[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            double a_y = ay.S0[j] + 0.5 * ay.DS[j];
            double c_y = 0.5 * ay.S0[j] + one_third * ay.DS[j];
            double fdy = fdy_ * ay.DS[j];
            double tz_ij = a_x * ay.S0[j] + c_x * ay.DS[j];
            bool yz = ay.tail_zero && j == 3;
            double jz_run = 0.0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                double tjx = a_y * az.S0[k] + c_y * az.DS[k];
                double tjy = a_x * az.S0[k] + c_x * az.DS[k];
                jx_run[k][j] -= fdx * tjx;
                jy_run[k] -= fdy * tjy;
                jz_run -= fdz_ * az.DS[k] * tz_ij;
                bool zz = az.tail_zero && k == 3;
                sink(i, j, k, xz ? 0.0 : jx_run[k][j], yz ? 0.0 : jy_run[k], zz ? 0.0 : jz_run,
                     cd * ax.S1[i] * ay.S1[j] * az.S1[k]);
            }
        }
    }
}

// The same deposit in two sweeps over the window, for the LDS kernel where registers are the scarce
// resource: sweep 1 walks x innermost and needs ONE running sum for jx; sweep 2 walks x outermost for jy
// (4 running sums), jz (1) and rho.  The single-sweep form above keeps 16 + 4 + 1 running sums alive.
// Every product is formed exactly as above (same operands, same order), so the results are bit-identical;
// DS is re-derived as S1 - S0 (its definition in axis_window) instead of being kept in registers.
template <class SinkX, class SinkYZR>
__device__ __forceinline__ void esirkepov_3d_lean(const AxisW &ax, const AxisW &ay, const AxisW &az, double w,
                                                  double c_rho, double c_jx, double c_jy, double c_jz,
                                                  SinkX &&sink_x, SinkYZR &&sink_yzr) {
    // c_rho = q / (dx dy dz), c_jx = q / (dy dz dt), c_jy = q / (dx dz dt), c_jz = q / (dx dy dt)
    const double one_third = 0.3333333333333333;
    {
        double fdx_ = c_jx * w;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            double dsy = ay.S1[j] - ay.S0[j];
            double a_y = ay.S0[j] + 0.5 * dsy;
            double c_y = 0.5 * ay.S0[j] + one_third * dsy;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                double tjx = a_y * az.S0[k] + c_y * (az.S1[k] - az.S0[k]);
                double run = 0.0;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    double fdx = fdx_ * (ax.S1[i] - ax.S0[i]);
                    run -= fdx * tjx;
                    sink_x(i, j, k, (ax.tail_zero && i == 3) ? 0.0 : run);
                }
            }
        }
    }
    double cd = c_rho * w;
    double fdy_ = c_jy * w;
    double fdz_ = c_jz * w;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        double dsx = ax.S1[i] - ax.S0[i];
        double a_x = ax.S0[i] + 0.5 * dsx;
        double c_x = 0.5 * ax.S0[i] + one_third * dsx;
        double jy_run[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            double dsy = ay.S1[j] - ay.S0[j];
            double fdy = fdy_ * dsy;
            double tz_ij = a_x * ay.S0[j] + c_x * dsy;
            bool yz = ay.tail_zero && j == 3;
            double jz_run = 0.0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                double dsz = az.S1[k] - az.S0[k];
                double tjy = a_x * az.S0[k] + c_x * dsz;
                jy_run[k] -= fdy * tjy;
                jz_run -= fdz_ * dsz * tz_ij;
                bool zz = az.tail_zero && k == 3;
                sink_yzr(i, j, k, yz ? 0.0 : jy_run[k], zz ? 0.0 : jz_run, cd * ax.S1[i] * ay.S1[j] * az.S1[k]);
            }
        }
    }
}

__device__ __forceinline__ void deposit_global_3d(const GridV &g, double x, double y, double z,
                                                  double ux, double uy, double uz, double ig, double w,
                                                  double q, double dt, bool rho = true) {
    double vx = ux * LPA_C * ig, vy = uy * LPA_C * ig, vz = uz * LPA_C * ig;
    AxisW ax, ay, az;
    axis_window(ax, x - vx * 0.5 * dt - g.x0, x + vx * 0.5 * dt - g.x0, 1.0 / g.dx);
    axis_window(ay, y - vy * 0.5 * dt - g.y0, y + vy * 0.5 * dt - g.y0, 1.0 / g.dy);
    axis_window(az, z - vz * 0.5 * dt - g.z0, z + vz * 0.5 * dt - g.z0, 1.0 / g.dz);
    long rows[4];
    int cols[4], deps[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        rows[k] = (long)torus(ax.base + k + g.ng, g.NX) * g.NY * g.NZ;
        cols[k] = torus(ay.base + k + g.ng, g.NY) * g.NZ;
        deps[k] = torus(az.base + k + g.ng, g.NZ);
    }
    // the register-lean two-sweep form (k_deposit_3d: 204 -> 144 VGPRs; the global push kernels stay at 349,
    // held there by the 162-load gather -- capping them at three waves per SIMD spilled 170-200 dwords)
    esirkepov_3d_lean(
        ax, ay, az, w, q / (g.dx * g.dy * g.dz), q / (g.dy * g.dz * dt), q / (g.dx * g.dz * dt), q / (g.dx * g.dy * dt),
        [&](int i, int j, int k, double djx) {
            if (djx != 0.0) atomicAdd(&g.jx[rows[i] + cols[j] + deps[k]], djx);
        },
        [&](int i, int j, int k, double djy, double djz, double dr) {
            long idx = rows[i] + cols[j] + deps[k];
            if (djy != 0.0) atomicAdd(&g.jy[idx], djy);
            if (djz != 0.0) atomicAdd(&g.jz[idx], djz);
            if (rho && dr != 0.0) atomicAdd(&g.rho[idx], dr);
        });
}

__device__ __forceinline__ double fold3(double v, double lo, double hi) {
    double L = hi - lo;
    if (v > hi) v -= L;
    if (v < lo) v += L;
    return v;
}

// periodic fold and / or absorption of the advanced position (sync_particles_3d with a self neighbour)
__device__ __forceinline__ bool finish_position_3d(double &x, double &y, double &z, const PushK3 &k) {
    if (k.wrap & 1) x = fold3(x, k.lo[0], k.hi[0]);
    if (k.wrap & 2) y = fold3(y, k.lo[1], k.hi[1]);
    if (k.wrap & 4) z = fold3(z, k.lo[2], k.hi[2]);
    bool dead = false;
    double c3[3] = {x, y, z};
#pragma unroll
    for (int a = 0; a < 3; a++)
        dead = dead || ((k.wrap & (LPA_ABSORB_X << a)) && (c3[a] < k.alo[a] || c3[a] > k.ahi[a]));
    if (dead) { x = __longlong_as_double(0x7ff8000000000000ll); y = x; z = x; }
    return dead;
}

// the whole per-particle update on global memory
// (returns the particle's new x; NaN: a dead slot)
__device__ __forceinline__ double update_global_3d(const GridV &g, const PartV &p, const PushK3 &k, long ip) {
    double x = p.x[ip], y = p.y[ip], z = p.z[ip];
    if ((p.dead && p.dead[ip]) || isnan(x) || isnan(y) || isnan(z)) return __longlong_as_double(0x7ff8000000000000ll);
    double ux = p.ux[ip], uy = p.uy[ip], uz = p.uz[ip], ig = p.ig[ip], w = p.w[ip];
    x += k.cdt_half * ig * ux;
    y += k.cdt_half * ig * uy;
    z += k.cdt_half * ig * uz;
    double eb[6];
    gather_global_3d(g, (x - g.x0) * (1.0 / g.dx), (y - g.y0) * (1.0 / g.dy), (z - g.z0) * (1.0 / g.dz),
                     eb);
    if (p.eb[0]) {
#pragma unroll
        for (int c = 0; c < 6; c++) p.eb[c][ip] = eb[c];
    }
    boris(ux, uy, uz, ig, eb[0], eb[1], eb[2], eb[3], eb[4], eb[5], k.efactor, k.bfactor);
    x += k.cdt_half * ig * ux;
    y += k.cdt_half * ig * uy;
    z += k.cdt_half * ig * uz;
    const bool no_rho = k.flags & LPA_PUSH_NO_RHO;
    deposit_global_3d(g, x, y, z, ux, uy, uz, ig, w, k.q, k.dt, !no_rho);
    const double xe = x, ye = y, ze = z;
    if (finish_position_3d(x, y, z, k) && k.absorbed) {
        // deposit_global_3d: end point = r + v dt / 2 with v = u c / gamma
        report_absorbed(g, k, (xe + ux * LPA_C * ig * 0.5 * k.dt - g.x0) * (1.0 / g.dx),
                        (ye + uy * LPA_C * ig * 0.5 * k.dt - g.y0) * (1.0 / g.dy),
                        (ze + uz * LPA_C * ig * 0.5 * k.dt - g.z0) * (1.0 / g.dz), (k.q / (g.dx * g.dy * g.dz)) * w);
    }
    report_leaver(k, x, ip);
    p.x[ip] = x; p.y[ip] = y; p.z[ip] = z;
    p.ux[ip] = ux; p.uy[ip] = uy; p.uz[ip] = uz; p.ig[ip] = ig;
    return x;
}

__global__ void __launch_bounds__(256) k_push_deposit_global_3d(GridV g, PartV p, PushK3 k, long first,
                                                                long count) {
    long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    update_global_3d(g, p, k, first + t);
}

__global__ void __launch_bounds__(256) k_push_deposit_list_3d(GridV g, PartV p, PushK3 k,
                                                              const uint32_t *__restrict__ list,
                                                              const uint32_t *__restrict__ list_count) {
    long n = *list_count;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long)gridDim.x * blockDim.x)
        update_global_3d(g, p, k, list[t]);
}

// overflow list + loose range in one launch (see k_push_deposit_rest_2d)
__global__ void __launch_bounds__(256) k_push_deposit_rest_3d(GridV g, PartV p, PushK3 k,
                                                              const uint32_t *__restrict__ list,
                                                              const uint32_t *__restrict__ list_count, long loose_first,
                                                              long loose_count, const int32_t *__restrict__ loose_limit) {
    const long stride = (long)gridDim.x * blockDim.x, t0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (list) {
        const long n = *list_count;
        for (long t = t0; t < n; t += stride) update_global_3d(g, p, k, list[t]);
    }
    long m = loose_count;
    if (loose_limit && (long)*loose_limit < m) m = *loose_limit;
    for (long t = t0; t < m; t += stride) update_global_3d(g, p, k, loose_first + t);
}

// the same launch with the step's leaver pack in it (lpa_migrate.hpp: rest_pack_body)
__global__ void __launch_bounds__(256) k_push_deposit_rest_pack_3d(GridV g, PartV p, PushK3 k,
                                                                   const uint32_t *__restrict__ list,
                                                                   const uint32_t *__restrict__ list_count, long loose_first,
                                                                   long loose_count, const int32_t *__restrict__ loose_limit,
                                                                   PackArgsV pk) {
    PushK3 kk = k;
    kk.leavers = nullptr;       // (what these particles do is seen right here: no list entry)
    rest_pack_body(p, list, list_count, loose_first, loose_count, loose_limit, pk,
                   [&](long ip) { return update_global_3d(g, p, kk, ip); });
}

// =====================================================================================================
// K1-tiled, 3-D.  Tiles of 4 x 4 x 16 cells (256 cells, z fastest), STRIPED order: a 16-lane group of a
// wave sits in 16 consecutive z-cells of one (x, y) column, and the LDS services a 64-bit atomic in
// 16-lane groups -- conflict free whatever the row strides.  One 512-thread workgroup per CU owns
// 152 KB of LDS: J / rho of the tile + 3 nodes on every side (margin 1 + the 2 nodes a deposit window
// reaches beyond the mid-step node): 4 x 10 x 10 x 22 f64 = 70 KB, and E / B on the nodes the gather can
// touch (mid-step node - 2 ... + 1 with the node within tile +- 1): 6 x 9 x 9 x 21 f64 = 82 KB.
// (Gathering E / B from global memory instead -- 162 cached 8-byte loads per particle -- was bound by
// the address rate of the texture path: 11.8 ms per step on C5's slab against 13.2 ms total.)
// =====================================================================================================
#ifndef LPA_SKIP_NULL_RUN
#define LPA_SKIP_NULL_RUN 1
#endif
#ifndef LPA_NT_PARTICLES_3D
#define LPA_NT_PARTICLES_3D 3
#endif
constexpr int T3X = LPA_TILE3_X, T3Y = LPA_TILE3_Y, T3Z = LPA_TILE3_Z;
constexpr int H3 = LPA_TILE3_MARGIN + 2;
constexpr int R3X = T3X + 2 * H3, R3Y = T3Y + 2 * H3, R3Z = T3Z + 2 * H3;  // 10 x 10 x 22
// E / B image layout experiment (LPA_EB_PAIRED=1, off): components stored in pairs, two 24-double column slots per
// 48-double row, so that the y stride (48) and the x stride (432) are 16 mod 32 doubles and the two y-adjacent
// 16-cell columns of a half-wave read the two halves of the bank array; the 11.7 KB this costs come out of the J
// image's z stride (24 -> 22).  Measured (profiles/r02_pmc3d_layouts.txt, same box): SQ_LDS_BANK_CONFLICT 4.45e8 ->
// 4.18e8 (-6 %), K1-3D 3.25 -> 3.29 ms: the conflicts of this kernel are NOT the overlap of the two columns.  They
// are there right after a sort as much as ten steps later (4.06e8 / 4.45e8, profiles/r02_pmc3d_age.txt): at 8
// particles per cell most stripes of the striped order are partial (a rank-8 stripe holds 41 % of the cells), so
// the 16 lanes of a group come from several columns and repeat z values -- the order cannot be conflict free at
// this density, whatever the strides.
#ifndef LPA_EB_PAIRED
#define LPA_EB_PAIRED 0
#endif
#ifndef LPA_R3ZS
#define LPA_R3ZS (LPA_EB_PAIRED ? 22 : 24)
#endif
#ifndef LPA_R3ZS_NORHO
#define LPA_R3ZS_NORHO 32
#endif
// (LPA_R3ZS: z stride of the J image (>= R3Z) of the kernel that deposits rho: 24 makes the x stride (240) a multiple
// of 16 doubles, so a lane that drifted along x keeps its bank (-3 % against 22))
constexpr int G3L = LPA_TILE3_MARGIN + 2, G3H = LPA_TILE3_MARGIN + 1;      // gather reach below / above the tile
constexpr int E3X = T3X + G3L + G3H, E3Y = T3Y + G3L + G3H, E3Z = T3Z + G3L + G3H;  // 9 x 9 x 21
constexpr int E3N = E3X * E3Y * E3Z;                                       // 1701
#if LPA_EB_PAIRED
constexpr int EBZ = 24, EBSY = 2 * EBZ, EBSX = E3Y * EBSY;                 // column slot, y stride, x stride
constexpr int EBN = E3X * EBSX;                                            // doubles per component PAIR
static_assert(EBSY % 32 == 16 && EBSX % 32 == 16 && EBZ >= E3Z, "E/B image strides");
__device__ __forceinline__ int eb_base(int c) { return (c >> 1) * EBN + (c & 1) * EBZ; }
#else
constexpr int EBSY = E3Z, EBSX = E3Y * E3Z, EBN = 2 * E3N;
__device__ __forceinline__ int eb_base(int c) { return c * E3N; }
#endif
#ifndef LPA_K13_THREADS
#define LPA_K13_THREADS 768
#endif
constexpr int K13_THREADS = LPA_K13_THREADS;

typedef const volatile __attribute__((address_space(3))) double *lds_ptr3;

// padded index of region node `node` along an axis: its periodic image inside [0, n) when the axis is
// locally periodic (a tile at a box edge, or the image of a particle folded through the face), else the
// padded-array torus of the global kernels
__device__ __forceinline__ int node_index(int node, int n, int ng, int N, bool periodic) {
    if (periodic && (unsigned)node >= (unsigned)n) { node %= n; if (node < 0) node += n; }
    return torus(node + ng, N);
}

// 27-point gather from the LDS image of one component; (lx, ly, lz) = local index of the stencil
// centre; evaluation order of interp_field_fast_3d (unified_pusher_3d.c:111-143): z outermost
// The TSC weights are rebuilt from the three offsets here (5 flops per axis) instead of keeping six
// weight triples alive across the whole gather: registers, not flops, bound this kernel.
#ifndef LPA_GATHER3_BATCH
#define LPA_GATHER3_BATCH 9
#endif
__device__ __forceinline__ double gather27_l(const double *f, int lx, int ly, int lz, double ddx, double ddy,
                                             double ddz) {
    double fx[3], fy[3], fz[3];
    tsc3(ddx, fx); tsc3(ddy, fy); tsc3(ddz, fz);
    lds_ptr3 c = (lds_ptr3)(f + lx * EBSX + ly * EBSY + lz);
    // the 3-D loop runs at 3 waves per SIMD: the reads of a batch are issued back to back and the arithmetic waits on
    // them with counted lgkmcnt (LDS returns in order).  Left to itself the scheduler, short of registers, issued 1-2
    // reads, waited for lgkmcnt(0), used them ... 108 exposed LDS latencies per particle (see DESIGN_HISTORY.md, round 2).
    double acc = 0.0;
#if LPA_GATHER3_BATCH == 27
    double v[3][3][3];
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            int o = (j - 1) * EBSY + (k - 1);
            v[k][j][0] = c[o - EBSX]; v[k][j][1] = c[o]; v[k][j][2] = c[o + EBSX];
        }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 3; k++) {
        double pl = 0.0;
#pragma unroll
        for (int j = 0; j < 3; j++) pl += fy[j] * (fx[0] * v[k][j][0] + fx[1] * v[k][j][1] + fx[2] * v[k][j][2]);
        acc += fz[k] * pl;
    }
#else
#pragma unroll
    for (int k = 0; k < 3; k++) {
        double v[3][3];
#pragma unroll
        for (int j = 0; j < 3; j++) {
            int o = (j - 1) * EBSY + (k - 1);
            v[j][0] = c[o - EBSX]; v[j][1] = c[o]; v[j][2] = c[o + EBSX];
        }
        __builtin_amdgcn_sched_barrier(0);
        double pl = 0.0;
#pragma unroll
        for (int j = 0; j < 3; j++) pl += fy[j] * (fx[0] * v[j][0] + fx[1] * v[j][1] + fx[2] * v[j][2]);
        acc += fz[k] * pl;
        __builtin_amdgcn_sched_barrier(0);
    }
#endif
    return acc;
}

// DEFER (as in the 2-D kernel): particles that change cell during the step park their advanced state (8
// doubles) in a scratch store -- the idle half of the ping-pong sort buffers -- and are deposited on the general
// 4 x 4 x 4 window by a dense second pass of the workgroup; the main loop then only sees particles that stay in
// their cell: old shape = the gather weights, 3 x 3 x 3 window, no window re-basing, no predicated atomics
// (static instruction count of the loop: 1275 -> 856 VALU, 627 -> 363 SALU, 256 -> 81 ds_add_f64).
struct Scratch8 { double *a[8]; };
__device__ __forceinline__ int clamp3(int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); }

// ---- the three phases of one workgroup's visit of a tile: images (zero J, stage E / B), the particles of one species,
// flush.  The single-species kernel runs them once each; the multi-species kernel runs the middle one per species
// between ONE staging and ONE flush (the E / B image is 82 KB per tile: for two species of 2048 particles per tile it is
// a quarter of the launch's HBM traffic and most of its per-tile fixed cost).
template <bool RHO> struct K13Geom {
    static constexpr int NJ = RHO ? 4 : 3;      // jx jy jz (rho)
    // z stride of the J image.  Without rho three components fit at stride LPA_R3ZS_NORHO = 32 (77 KB): y stride 32
    // and x stride 320 are both 0 mod 16 doubles, so the bank of a ds_add_f64 depends on the lane's z cell alone -- the
    // 16 lanes of a group conflict only where two of them sit in the same z cell, whatever columns a compacted
    // (partial) stripe draws them from.  With rho (four components) only stride 24 fits beside the E / B image.
    static constexpr int R3ZS = RHO ? LPA_R3ZS : LPA_R3ZS_NORHO;
    static constexpr int R3N = R3X * R3Y * R3ZS;
    static_assert(R3ZS >= R3Z, "J image z stride");
};

struct TileCtx { int t0[3], r0[3], e0[3]; int tile; };   // first node of the tile, of the J region, of the E / B image

__device__ __forceinline__ TileCtx tile_ctx(int tile, int tiles_y, int tiles_z, int &tx_) {
    const int tz_ = tile % tiles_z, ty_ = (tile / tiles_z) % tiles_y;
    tx_ = tile / (tiles_z * tiles_y);
    TileCtx c;
    c.tile = tile;
    c.t0[0] = tx_ * T3X; c.t0[1] = ty_ * T3Y; c.t0[2] = tz_ * T3Z;
#pragma unroll
    for (int a = 0; a < 3; a++) { c.r0[a] = c.t0[a] - H3; c.e0[a] = c.t0[a] - G3L; }
    return c;
}

template <bool RHO>
__device__ __forceinline__ void tile_images(const GridV &g, int wrap, const TileCtx &tc,
                                            double (*s_j)[K13Geom<RHO>::R3N], double *s_eb) {
    constexpr int NJ = K13Geom<RHO>::NJ, R3ZS = K13Geom<RHO>::R3ZS, R3N = K13Geom<RHO>::R3N;
    for (int t = threadIdx.x; t < R3N; t += blockDim.x) {
        if (R3ZS > R3Z + 2 && t % R3ZS >= R3Z) continue;       // (stride padding is never read)
#pragma unroll
        for (int c = 0; c < NJ; c++) s_j[c][t] = 0.0;
    }
    const double *src[6] = {g.ex, g.ey, g.ez, g.bx, g.by, g.bz};
    for (int t = threadIdx.x; t < E3N; t += blockDim.x) {
        int lz = t % E3Z, ly = (t / E3Z) % E3Y, lx = t / (E3Z * E3Y);
        long gi = ((long)node_index(tc.e0[0] + lx, g.nx, g.ng, g.NX, wrap & 1) * g.NY +
                   node_index(tc.e0[1] + ly, g.ny, g.ng, g.NY, wrap & 2)) * g.NZ +
                  node_index(tc.e0[2] + lz, g.nz, g.ng, g.NZ, wrap & 4);
#pragma unroll
        for (int c = 0; c < 6; c++) s_eb[eb_base(c) + lx * EBSX + ly * EBSY + lz] = src[c][gi];
    }
}

// flush: one FP64 global atomic per touched node and component, on the torus
template <bool RHO>
__device__ __forceinline__ void tile_flush(const GridV &g, int wrap, const TileCtx &tc,
                                           double (*s_j)[K13Geom<RHO>::R3N]) {
    constexpr int NJ = K13Geom<RHO>::NJ, R3ZS = K13Geom<RHO>::R3ZS, R3N = K13Geom<RHO>::R3N;
    double *dst[4] = {g.jx, g.jy, g.jz, g.rho};
    for (int t = threadIdx.x; t < R3N; t += blockDim.x) {
        int lz = t % R3ZS, ly = (t / R3ZS) % R3Y, lx = t / (R3ZS * R3Y);
        if (lz >= R3Z) continue;   // stride padding
        long gi = ((long)node_index(tc.r0[0] + lx, g.nx, g.ng, g.NX, wrap & 1) * g.NY +
                   node_index(tc.r0[1] + ly, g.ny, g.ng, g.NY, wrap & 2)) * g.NZ +
                  node_index(tc.r0[2] + lz, g.nz, g.ng, g.NZ, wrap & 4);
#pragma unroll
        for (int c = 0; c < NJ; c++) {
            double v = s_j[c][t];
            if (v != 0.0) atomicAdd(&dst[c][gi], v);
        }
    }
}

// the particles [begin, end) of one species in this tile: main loop (+ the dense second pass of the cell-crossers).
// On entry the images are ready and *s_ncross == 0 (both behind a barrier); on exit every lane has issued its LDS atomics
// (the caller puts a barrier before it reads s_j or starts the next species).
template <bool DEFER, bool RHO>
__device__ __forceinline__ void species_pass(const GridV &g, const PartV &p_, const PushK3 &k, const int begin,
                                             const int end, const Scratch8 &sc_, uint32_t *overflow,
                                             uint32_t *overflow_count, const TileCtx &tc,
                                             double (*s_j)[K13Geom<RHO>::R3N], double *s_eb, int *s_ncross_p) {
    constexpr int NJ = K13Geom<RHO>::NJ, R3ZS = K13Geom<RHO>::R3ZS;
    // the attribute and scratch arrays rebased to the pass' first slot (uniform: scalar adds): the 32-bit byte offsets
    // below are relative to it, so a store is limited by the sort's int32 slot numbers (2^31), not by 2^32 bytes per array
    PartV p = p_;
    Scratch8 sc = sc_;
    const int rb = LPA_ABS_OFFSETS ? 0 : begin;     // (LPA_ABS_OFFSETS=1: the old absolute offsets, for A/B timing builds)
    p.x += rb; p.y += rb; p.z += rb; p.ux += rb; p.uy += rb; p.uz += rb; p.ig += rb; p.w += rb;
    if (DEFER) {
#pragma unroll
        for (int c = 0; c < 8; c++) sc.a[c] += rb;
    }
    const int *t0 = tc.t0, *r0 = tc.r0, *e0 = tc.e0;
    const int lane = threadIdx.x & 63;
    int &s_ncross = *s_ncross_p;
    const double inv_dx = k.inv_d[0], inv_dy = k.inv_d[1], inv_dz = k.inv_d[2];
    auto ld = [](const double *base, uint32_t off) { return *(const double *)((const char *)base + off); };
    auto st = [](double *base, uint32_t off, double v) { *(double *)((char *)base + off) = v; };
    // the particle attributes stream through once per step (0.5 MB per tile visit against a 4 MB L2 per XCD): with the
    // non-temporal hint they do not push the E / B and J lines the neighbouring tiles are about to reuse out of the L2
    // LPA_NT_PARTICLES_3D: bit 0 = loads, bit 1 = stores
    auto ldp = [](const double *base, uint32_t off) {
        const double *q = (const double *)((const char *)base + off);
        return (LPA_NT_PARTICLES_3D & 1) ? __builtin_nontemporal_load(q) : *q; };
    auto stp = [](double *base, uint32_t off, double v) {
        double *q = (double *)((char *)base + off);
        if (LPA_NT_PARTICLES_3D & 2) __builtin_nontemporal_store(v, q); else *q = v; };
    // software pipeline: the eight attribute loads of the next iteration are in flight during this one
    double nx_ = 0.0, ny_ = 0.0, nz_ = 0.0, nux = 0.0, nuy = 0.0, nuz = 0.0, nig = 1.0, nw = 0.0;
    {
        const int ip0 = begin + (int)(threadIdx.x & ~63u) + lane;
        if (ip0 < end) {
            const uint32_t o = (uint32_t)(ip0 - rb) * 8u;
            nx_ = ldp(p.x, o); ny_ = ldp(p.y, o); nz_ = ldp(p.z, o); nux = ldp(p.ux, o); nuy = ldp(p.uy, o);
            nuz = ldp(p.uz, o); nig = ldp(p.ig, o); nw = ldp(p.w, o);
        }
    }
    for (int it = begin + (int)(threadIdx.x & ~63u); it < end; it += blockDim.x) {
        const int ip = it + lane;
        bool valid = ip < end;
        double x = nx_, y = ny_, z = nz_, ux = nux, uy = nuy, uz = nuz, ig = nig, w = nw;
        {
            const int ipn = ip + (int)blockDim.x;
            if (ipn < end) {
                const uint32_t o = (uint32_t)(ipn - rb) * 8u;
                nx_ = ldp(p.x, o); ny_ = ldp(p.y, o); nz_ = ldp(p.z, o); nux = ldp(p.ux, o); nuy = ldp(p.uy, o);
                nuz = ldp(p.uz, o); nig = ldp(p.ig, o); nw = ldp(p.w, o);
            }
        }
        valid = valid && !(isnan(x) || isnan(y) || isnan(z));
        if (!valid) continue;
        x += k.cdt_half * ig * ux;
        y += k.cdt_half * ig * uy;
        z += k.cdt_half * ig * uz;
        double eb[6];
        [[maybe_unused]] int mid[3] = {0, 0, 0};          // mid-step nearest node
        [[maybe_unused]] double gd[3] = {0.0, 0.0, 0.0};  // node - position in cells: the TSC offsets of the gather
        {
            // a particle folded through a locally periodic face since the last sort: work on its periodic
            // image next to the tile (see the 2-D kernel)
            if (k.wrap & 7) {
                const int n3[3] = {g.nx, g.ny, g.nz}, tt[3] = {T3X, T3Y, T3Z};
                double *pos[3] = {&x, &y, &z};
                const double org[3] = {g.x0, g.y0, g.z0}, inv[3] = {inv_dx, inv_dy, inv_dz};
#pragma unroll
                for (int a = 0; a < 3; a++) {
                    if (!((k.wrap >> a) & 1) || n3[a] < 2 * (tt[a] + 8)) continue;
                    int dd = ifloor((*pos[a] - org[a]) * inv[a] + 0.5) - (t0[a] + tt[a] / 2);
                    if (dd > (n3[a] >> 1)) *pos[a] -= k.hi[a] - k.lo[a];
                    else if (dd < -(n3[a] >> 1)) *pos[a] += k.hi[a] - k.lo[a];
                }
            }
            double xo = (x - g.x0) * inv_dx, yo = (y - g.y0) * inv_dy, zo = (z - g.z0) * inv_dz;
            int ix1 = ifloor(xo + 0.5), ix2 = ifloor(xo);
            int iy1 = ifloor(yo + 0.5), iy2 = ifloor(yo);
            int iz1 = ifloor(zo + 0.5), iz2 = ifloor(zo);
            // LDS gather iff the mid-step node lies within tile +- margin (stencil: node - 2 ... + 1)
            if ((unsigned)(ix1 - (t0[0] - LPA_TILE3_MARGIN)) >= (unsigned)(T3X + 2 * LPA_TILE3_MARGIN) ||
                (unsigned)(iy1 - (t0[1] - LPA_TILE3_MARGIN)) >= (unsigned)(T3Y + 2 * LPA_TILE3_MARGIN) ||
                (unsigned)(iz1 - (t0[2] - LPA_TILE3_MARGIN)) >= (unsigned)(T3Z + 2 * LPA_TILE3_MARGIN)) {
                uint32_t slot = atomicAdd(overflow_count, 1u);
                overflow[slot] = (uint32_t)ip;
                continue;
            }
            // offsets of the node-centred (g) and half-cell-centred (h) stencils
            const double gxd = ix1 - xo, hxd = ix2 - xo + 0.5, gyd = iy1 - yo, hyd = iy2 - yo + 0.5,
                         gzd = iz1 - zo, hzd = iz2 - zo + 0.5;
            const int lx1 = ix1 - e0[0], lx2 = ix2 - e0[0], ly1 = iy1 - e0[1], ly2 = iy2 - e0[1],
                      lz1 = iz1 - e0[2], lz2 = iz2 - e0[2];
            mid[0] = ix1; mid[1] = iy1; mid[2] = iz1;
            gd[0] = gxd; gd[1] = gyd; gd[2] = gzd;
            // stagger table: unified_pusher_3d.c:190-195
            eb[0] = gather27_l(s_eb + eb_base(0), lx2, ly1, lz1, hxd, gyd, gzd);
            __builtin_amdgcn_sched_barrier(0);
            eb[1] = gather27_l(s_eb + eb_base(1), lx1, ly2, lz1, gxd, hyd, gzd);
            __builtin_amdgcn_sched_barrier(0);
            eb[2] = gather27_l(s_eb + eb_base(2), lx1, ly1, lz2, gxd, gyd, hzd);
            __builtin_amdgcn_sched_barrier(0);
            eb[3] = gather27_l(s_eb + eb_base(3), lx1, ly2, lz2, gxd, hyd, hzd);
            __builtin_amdgcn_sched_barrier(0);
            eb[4] = gather27_l(s_eb + eb_base(4), lx2, ly1, lz2, hxd, gyd, hzd);
            __builtin_amdgcn_sched_barrier(0);
            eb[5] = gather27_l(s_eb + eb_base(5), lx2, ly2, lz1, hxd, hyd, gzd);
            __builtin_amdgcn_sched_barrier(0);
        }
        boris(ux, uy, uz, ig, eb[0], eb[1], eb[2], eb[3], eb[4], eb[5], k.efactor, k.bfactor);
        x += k.cdt_half * ig * ux;
        y += k.cdt_half * ig * uy;
        z += k.cdt_half * ig * uz;
        double vx = ux * LPA_C * ig, vy = uy * LPA_C * ig, vz = uz * LPA_C * ig;
        AxisW ax, ay, az;
        bool cross = false;
        int bx, by, bz;
        if (DEFER) {
            // old shape = gather weights around the mid-step node; the particle stays in its cell iff its
            // advanced position has the same nearest node on every axis
            const double d1[3] = {mid[0] - (x + vx * 0.5 * k.dt - g.x0) * inv_dx,
                                  mid[1] - (y + vy * 0.5 * k.dt - g.y0) * inv_dy,
                                  mid[2] - (z + vz * 0.5 * k.dt - g.z0) * inv_dz};
            cross = !(d1[0] > -0.5 && d1[0] <= 0.5 && d1[1] > -0.5 && d1[1] <= 0.5 && d1[2] > -0.5 && d1[2] <= 0.5);
            if (cross) {
                // the general window covers the cells of the old and the new nearest node +- 1: inside the staged
                // region?  (otherwise: overflow list, untouched)
                bool inside = true;
#pragma unroll
                for (int a = 0; a < 3; a++) {
                    const int i1 = mid[a] - ifloor(d1[a] + 0.5);      // nearest node of the advanced position
                    const int lo = (i1 < mid[a] ? i1 : mid[a]) - 1 - r0[a], hi = (i1 > mid[a] ? i1 : mid[a]) + 1 - r0[a];
                    const int rr = a == 0 ? R3X : (a == 1 ? R3Y : R3Z);
                    inside = inside && lo >= 0 && hi < rr;
                }
                if (!inside) {
                    uint32_t slot = atomicAdd(overflow_count, 1u);
                    overflow[slot] = (uint32_t)ip;
                    continue;
                }
            }
            AxisW *aw[3] = {&ax, &ay, &az};
#pragma unroll
            for (int a = 0; a < 3; a++) {
                tsc3(gd[a], aw[a]->S0);
                tsc3(d1[a], aw[a]->S1);
                aw[a]->S0[3] = aw[a]->S1[3] = aw[a]->DS[3] = 0.0;
#pragma unroll
                for (int c = 0; c < 3; c++) aw[a]->DS[c] = aw[a]->S1[c] - aw[a]->S0[c];
                aw[a]->base = mid[a] - 1;
                aw[a]->tail_zero = true;
            }
            // (mid within tile +- 1 => the 3-cell window starts 1 ... T + 3 nodes into the region: inside)
            bx = ax.base - r0[0]; by = ay.base - r0[1]; bz = az.base - r0[2];
        } else {
            axis_window(ax, x - vx * 0.5 * k.dt - g.x0, x + vx * 0.5 * k.dt - g.x0, inv_dx);
            axis_window(ay, y - vy * 0.5 * k.dt - g.y0, y + vy * 0.5 * k.dt - g.y0, inv_dy);
            axis_window(az, z - vz * 0.5 * k.dt - g.z0, z + vz * 0.5 * k.dt - g.z0, inv_dz);
            // the LDS path is valid iff the whole 4 x 4 x 4 window lies inside the staged region; anything
            // else (drifted further than the margin since the sort, non-finite input) goes to the overflow
            // list untouched -- nothing has been stored or deposited yet
            bx = ax.base - r0[0]; by = ay.base - r0[1]; bz = az.base - r0[2];
            if ((unsigned)bx > (unsigned)(R3X - 4) || (unsigned)by > (unsigned)(R3Y - 4) ||
                (unsigned)bz > (unsigned)(R3Z - 4)) {
                uint32_t slot = atomicAdd(overflow_count, 1u);
                overflow[slot] = (uint32_t)ip;
                continue;
            }
        }
        if (p.eb[0]) {
#pragma unroll
            for (int c = 0; c < 6; c++) p.eb[c][ip] = eb[c];
        }
        {
            double xs = x, ys = y, zs = z;
            if (finish_position_3d(xs, ys, zs, k) && k.absorbed)    // rare: a particle reached an open face
                report_absorbed(g, k, (x + vx * 0.5 * k.dt - g.x0) * inv_dx, (y + vy * 0.5 * k.dt - g.y0) * inv_dy,
                                (z + vz * 0.5 * k.dt - g.z0) * inv_dz, k.c_rho * w);
            report_leaver(k, xs, ip, tc.tile);
            const uint32_t o = (uint32_t)(ip - rb) * 8u;
            stp(p.x, o, xs); stp(p.y, o, ys); stp(p.z, o, zs);
            stp(p.ux, o, ux); stp(p.uy, o, uy); stp(p.uz, o, uz); stp(p.ig, o, ig);
        }
        if (DEFER && cross) {   // parked: deposited by the second pass (unfolded position: the window is local)
            const int slot = atomicAdd(&s_ncross, 1);
            const uint32_t o = (uint32_t)(begin - rb + slot) * 8u;
            st(sc.a[0], o, x); st(sc.a[1], o, y); st(sc.a[2], o, z); st(sc.a[3], o, ux); st(sc.a[4], o, uy);
            st(sc.a[5], o, uz); st(sc.a[6], o, ig); st(sc.a[7], o, w);
            continue;
        }
        const int b0 = (bx * R3Y + by) * R3ZS + bz;
        // window plane 3 of an axis carries exact zeros unless the particle changed cell along that axis
        // (see the 2-D kernel)
        esirkepov_3d_lean(
            ax, ay, az, w, k.c_rho, k.c_j[0], k.c_j[1], k.c_j[2],
            [&](int i, int j, int kk, double djx) {
                // a running sum over the 3 window planes of a particle that stayed in its cell along
                // that axis ends at (sum of DS) * (...) = 0 up to rounding (the reference adds that
                // 1e-16-relative residue): plane 2 of jx / jy / jz carries nothing unless the particle
                // crossed along x / y / z
                bool on = (i < (LPA_SKIP_NULL_RUN ? 2 : 3) || !ax.tail_zero) && (j < 3 || !ay.tail_zero) &&
                          (kk < 3 || !az.tail_zero);
                if (on) atomicAdd(&s_j[0][b0 + (i * R3Y + j) * R3ZS + kk], djx);
            },
            [&](int i, int j, int kk, double djy, double djz, double dr) {
                bool on = (i < 3 || !ax.tail_zero) && (j < 3 || !ay.tail_zero) && (kk < 3 || !az.tail_zero);
                if (on) {
                    int o = b0 + (i * R3Y + j) * R3ZS + kk;
                    if (!LPA_SKIP_NULL_RUN || j < 2 || !ay.tail_zero) atomicAdd(&s_j[1][o], djy);
                    if (!LPA_SKIP_NULL_RUN || kk < 2 || !az.tail_zero) atomicAdd(&s_j[2][o], djz);
                    if (RHO) atomicAdd(&s_j[NJ - 1][o], dr);
                }
            });
    }
    if (DEFER) {
        // ---- the particles that changed cell: the general 4 x 4 x 4 window, every lane busy
        __syncthreads();
        const int ncross = s_ncross;
        for (int i = threadIdx.x; i < ncross; i += blockDim.x) {
            const uint32_t o = (uint32_t)(begin - rb + i) * 8u;
            const double x = ld(sc.a[0], o), y = ld(sc.a[1], o), z = ld(sc.a[2], o), ux = ld(sc.a[3], o),
                         uy = ld(sc.a[4], o), uz = ld(sc.a[5], o), ig = ld(sc.a[6], o), w = ld(sc.a[7], o);
            const double vx = ux * LPA_C * ig, vy = uy * LPA_C * ig, vz = uz * LPA_C * ig;
            AxisW ax, ay, az;
            axis_window(ax, x - vx * 0.5 * k.dt - g.x0, x + vx * 0.5 * k.dt - g.x0, inv_dx);
            axis_window(ay, y - vy * 0.5 * k.dt - g.y0, y + vy * 0.5 * k.dt - g.y0, inv_dy);
            axis_window(az, z - vz * 0.5 * k.dt - g.z0, z + vz * 0.5 * k.dt - g.z0, inv_dz);
            // (clamped: the old cell is re-derived here and may differ by one from the main loop's for a particle
            // within an ulp of a cell boundary, where the shape value of the extra cell is an ulp as well)
            const int bx = clamp3(ax.base - r0[0], R3X - 4), by = clamp3(ay.base - r0[1], R3Y - 4),
                      bz = clamp3(az.base - r0[2], R3Z - 4);
            const int b0 = (bx * R3Y + by) * R3ZS + bz;
            esirkepov_3d_lean(
                ax, ay, az, w, k.c_rho, k.c_j[0], k.c_j[1], k.c_j[2],
                [&](int ii, int j, int kk, double djx) {
                    bool on = (ii < (LPA_SKIP_NULL_RUN ? 2 : 3) || !ax.tail_zero) && (j < 3 || !ay.tail_zero) &&
                              (kk < 3 || !az.tail_zero);
                    if (on) atomicAdd(&s_j[0][b0 + (ii * R3Y + j) * R3ZS + kk], djx);
                },
                [&](int ii, int j, int kk, double djy, double djz, double dr) {
                    bool on = (ii < 3 || !ax.tail_zero) && (j < 3 || !ay.tail_zero) && (kk < 3 || !az.tail_zero);
                    if (on) {
                        int oo = b0 + (ii * R3Y + j) * R3ZS + kk;
                        if (!LPA_SKIP_NULL_RUN || j < 2 || !ay.tail_zero) atomicAdd(&s_j[1][oo], djy);
                        if (!LPA_SKIP_NULL_RUN || kk < 2 || !az.tail_zero) atomicAdd(&s_j[2][oo], djz);
                        if (RHO) atomicAdd(&s_j[NJ - 1][oo], dr);
                    }
                });
        }
    }
}

// One kernel, two work partitions.  Work blocks (blk_tile != nullptr; ONE species): a workgroup per (tile, particle
// range) of the sort's block table -- tiles with more particles than a block get several workgroups, and an edge /
// interior part launch skips the other part's tiles.  Whole tiles (blk_tile == nullptr; up to four species): a
// workgroup per tile stages the E / B image once, runs every species' particles of that tile (ranges from the species'
// own tile_off tables) and flushes J once -- the staging is 82 KB per tile, as much as 700 particles.
// The per-species arguments live in an array indexed at run time even for one species: taken as plain kernel arguments
// the compiler keeps all ~60 of them in SGPRs for the whole kernel and spills 90 more (2.89 against 2.62 ms).
constexpr int K13_MAX_SPECIES = 4;
struct MultiSp {
    PartV p;
    PushK3 k;
    Scratch8 sc;
    const int32_t *tile_off;
    uint32_t *overflow, *overflow_count;
};
struct MultiArgs { int ns; MultiSp s[K13_MAX_SPECIES]; };

template <bool DEFER, bool RHO>
__global__ void __launch_bounds__(K13_THREADS) k_push_deposit_tiled_3d(
    GridV g, MultiArgs m, const int32_t *__restrict__ blk_tile, const int32_t *__restrict__ blk_begin,
    const int32_t *__restrict__ blk_end, const int32_t *__restrict__ n_blocks, int ntiles, int tiles_y, int tiles_z,
    int part, int tiles_x, int edge_cols) {
    __shared__ double s_j[K13Geom<RHO>::NJ][K13Geom<RHO>::R3N];
    __shared__ double s_eb[3 * EBN];     // see eb_base()
    __shared__ int s_ncross;
    // plain order: consecutive workgroups (dealt round-robin over the 8 XCDs) take consecutive tiles.  Giving
    // every XCD a contiguous run of tiles, as the 2-D kernel does, measured 1.2 % SLOWER here (4.57 against
    // 4.51 ms per step) and idles XCDs in an edge / interior part launch.  (Persistent workgroups walking several
    // work blocks spilled 163 SGPRs + 35 VGPRs: 3.85 against 3.25 ms, profiles/r02_k13d_ablations.txt.)
    const int wb = (int)blockIdx.x;
    int tile = wb, blk_b = 0, blk_e = 0;
    if (blk_tile) {
        if (wb >= *n_blocks) return;  // block-uniform
        tile = blk_tile[wb]; blk_b = blk_begin[wb]; blk_e = blk_end[wb];
    } else {
#ifdef LPA_XCD_TILES_3D   // experiment: every XCD (workgroups are dealt round-robin over 8) walks a contiguous run of tiles.
                          // C5 leg (profiles/r03_ab_xcd3.txt): HBM fetch -3 % only, K1 1.85 -> 3.5 ms (the vacuum tiles of
                          // the slab all fall to two XCDs): off
        const int chunk = (ntiles + 7) >> 3;
        tile = (wb & 7) * chunk + (wb >> 3);
        if (wb >= 8 * chunk || tile >= ntiles) return;
#else
        if (wb >= ntiles) return;
#endif
        bool any = false;                  // an empty tile (vacuum) costs nothing: no staging, no flush
        for (int s = 0; s < m.ns; s++) any = any || m.s[s].tile_off[tile] < m.s[s].tile_off[tile + 1];
        if (!any) return;                  // block-uniform
    }
    int tx_;
    const TileCtx tc = tile_ctx(tile, tiles_y, tiles_z, tx_);
    if (part) {  // LPA_PART_EDGE / LPA_PART_INTERIOR: see lpa_push_deposit_tiled_part_2d
        const bool edge = tx_ < edge_cols || tx_ >= tiles_x - edge_cols;
        if ((part == LPA_PART_EDGE) != edge) return;  // block-uniform, before any barrier
    }
    const int wrap = m.s[0].k.wrap;        // the periodic axes are the slab's: the same for every species
    tile_images<RHO>(g, wrap, tc, s_j, s_eb);
    for (int s = 0; s < m.ns; s++) {
        if (threadIdx.x == 0) s_ncross = 0;
        __syncthreads();                   // images ready / the previous species' second pass has read s_ncross
        const int begin = blk_tile ? blk_b : m.s[s].tile_off[tile], end = blk_tile ? blk_e : m.s[s].tile_off[tile + 1];
        if (begin < end)                   // block-uniform
            species_pass<DEFER, RHO>(g, m.s[s].p, m.s[s].k, begin, end, m.s[s].sc, m.s[s].overflow,
                                     m.s[s].overflow_count, tc, s_j, s_eb, &s_ncross);
    }
    __syncthreads();
    tile_flush<RHO>(g, wrap, tc, s_j);
}

static PushK3 make_pushk3(const lpa_push_params *pp, const lpa_grid *g = nullptr) {
    PushK3 k;
    k.inv_d[0] = k.inv_d[1] = k.inv_d[2] = 0.0;
    k.c_rho = k.c_j[0] = k.c_j[1] = k.c_j[2] = 0.0;
    if (g) {
        k.inv_d[0] = 1.0 / g->dx; k.inv_d[1] = 1.0 / g->dy; k.inv_d[2] = 1.0 / g->dz;
        k.c_rho = pp->q / (g->dx * g->dy * g->dz);
        k.c_j[0] = pp->q / (g->dy * g->dz * pp->dt);
        k.c_j[1] = pp->q / (g->dx * g->dz * pp->dt);
        k.c_j[2] = pp->q / (g->dx * g->dy * pp->dt);
    }
    k.dt = pp->dt; k.q = pp->q;
    k.efactor = pp->q * pp->dt / (2 * pp->m * LPA_C);
    k.bfactor = pp->q * pp->dt / (2 * pp->m);
    k.cdt_half = LPA_C * 0.5 * pp->dt;
    k.wrap = pp->wrap;
    k.flags = pp->flags;
    k.absorbed = pp->absorbed; k.absorbed_count = pp->absorbed_count; k.absorbed_cap = (long)pp->absorbed_capacity;
    k.absorbed_spill = pp->absorbed_spill;
    k.leavers = (unsigned long long *)pp->leavers; k.leaver_count = pp->leaver_count; k.leaver_cap = (long)pp->leaver_capacity;
    k.leave_lo = pp->leave_lo; k.leave_hi = pp->leave_hi;
    for (int a = 0; a < 3; a++) {
        k.lo[a] = pp->lo[a]; k.hi[a] = pp->hi[a];
        k.alo[a] = pp->alo[a]; k.ahi[a] = pp->ahi[a];
    }
    return k;
}

__global__ void __launch_bounds__(256) k_wrap_positions_3d(PartV p, PushK3 k) {
    long ip = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (ip >= p.n) return;
    if (p.dead && p.dead[ip]) return;
    double x = p.x[ip], y = p.y[ip], z = p.z[ip];
    if (isnan(x) || isnan(y) || isnan(z)) return;
    finish_position_3d(x, y, z, k);
    p.x[ip] = x; p.y[ip] = y; p.z[ip] = z;
}

extern "C" int lpa_wrap_positions_3d(const lpa_particles *p, const lpa_push_params *pp, void *stream) {
    LPA_REQUIRE(lpa_part_ok(p, 3) && pp, "lpa_wrap_positions_3d: bad args");
    if (p->n == 0 || !pp->wrap) return LPA_OK;
    lpa_push_params q = *pp;
    if (!(q.dt > 0)) q.dt = 1.0;
    if (!(q.m > 0)) q.m = 1.0;
    hipLaunchKernelGGL(k_wrap_positions_3d, dim3((unsigned)((p->n + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, make_partv(p), make_pushk3(&q));
    LPA_CHECK_LAUNCH("lpa_wrap_positions_3d");
    return LPA_OK;
}

static int check_push3(const lpa_grid *g, const lpa_particles *p, const lpa_push_params *pp, const char *name) {
    LPA_REQUIRE(lpa_grid_ok(g, 3, 1), "%s: bad grid", name);
    LPA_REQUIRE(lpa_part_ok(p, 3), "%s: bad particle store", name);
    LPA_REQUIRE(pp && pp->dt > 0 && pp->m > 0, "%s: dt and m must be > 0", name);
    LPA_REQUIRE(!(pp->flags & LPA_PUSH_NO_RHO) || !(pp->wrap & (7 * LPA_ABSORB_X)) || pp->absorbed,
                "%s: LPA_PUSH_NO_RHO with absorbing faces needs the absorbed list", name);
    LPA_REQUIRE(!pp->absorbed || (pp->absorbed_count && pp->absorbed_capacity > 0), "%s: bad absorbed list", name);
    LPA_REQUIRE(!pp->leavers || (pp->leaver_count && pp->leaver_capacity > 0 && pp->leave_lo < pp->leave_hi),
                "%s: bad leaver list", name);
    return LPA_OK;
}

extern "C" int lpa_push_deposit_list_3d(const lpa_grid *g, const lpa_particles *p,
                                        const lpa_push_params *pp, const uint32_t *list,
                                        const uint32_t *list_count, int64_t max_count, void *stream) {
    if (int e = check_push3(g, p, pp, "lpa_push_deposit_list_3d")) return e;
    LPA_REQUIRE(list && list_count && max_count >= 0, "lpa_push_deposit_list_3d: bad list");
    if (max_count == 0 || p->n == 0) return LPA_OK;
    long nb = (max_count + 255) / 256;
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(k_push_deposit_list_3d, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream,
                       make_gridv(g, 3), make_partv(p), make_pushk3(pp), list, list_count);
    LPA_CHECK_LAUNCH("lpa_push_deposit_list_3d");
    return LPA_OK;
}

int lpai_push_deposit_rest_3d(const lpa_grid *g, const lpa_particles *p, const lpa_push_params *pp, const uint32_t *list,
                              const uint32_t *list_count, int64_t max_count, int64_t loose_first, int64_t loose_count,
                              const int32_t *loose_limit, const lpa_pack_args *pack, void *stream) {
    if (int e = check_push3(g, p, pp, "lpai_push_deposit_rest_3d")) return e;
    LPA_REQUIRE((!list || list_count) && max_count >= 0 && loose_first >= 0 && loose_count >= 0 &&
                    loose_first + loose_count <= p->n, "lpai_push_deposit_rest_3d: bad list / range");
    LPA_REQUIRE(!pack || pack_args_ok(pack, p), "lpai_push_deposit_rest_3d: bad pack arguments");
    const long work = (list ? max_count : 0) > loose_count ? (long)max_count : (long)loose_count;
    if ((work == 0 && !(pack && pack->with_list)) || p->n == 0) return LPA_OK;
    long nb = (work + 255) / 256;
    if (nb > 2048) nb = 2048;
    if (pack) {     // the step's leaver pack rides in this launch
        const PackArgsV pk = make_pack_args_v(pack);
        hipLaunchKernelGGL(k_push_deposit_rest_pack_3d, dim3((unsigned)(nb + pk.pack_blocks)), dim3(256), 0, (hipStream_t)stream,
                           make_gridv(g, 3), make_partv(p), make_pushk3(pp), list, list_count, (long)loose_first,
                           (long)loose_count, loose_limit, pk);
        LPA_CHECK_LAUNCH("lpai_push_deposit_rest_3d (with the leaver pack)");
        return LPA_OK;
    }
    hipLaunchKernelGGL(k_push_deposit_rest_3d, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, make_gridv(g, 3),
                       make_partv(p), make_pushk3(pp), list, list_count, (long)loose_first, (long)loose_count, loose_limit);
    LPA_CHECK_LAUNCH("lpai_push_deposit_rest_3d");
    return LPA_OK;
}

// ---- standalone 3-D kernels (the reference's non-unified twins: interpolation/cpu3d.c:99-169,
//      current/cpu3d.c:118-183).  The reference has no 3-D push_position (PusherBase.push_position,
//      core/pusher/pusher.py:103-110, moves particles in 2-D only), so there is no 3-D split step.
__global__ void __launch_bounds__(256) k_interpolate_3d(GridV g, PartV p) {
    long ip = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (ip >= p.n) return;
    if (p.dead && p.dead[ip]) return;  // interpolation/cpu3d.c skips is_dead only
    double eb[6];
    gather_global_3d(g, (p.x[ip] - g.x0) / g.dx, (p.y[ip] - g.y0) / g.dy, (p.z[ip] - g.z0) / g.dz, eb);
#pragma unroll
    for (int c = 0; c < 6; c++) p.eb[c][ip] = eb[c];
}

__global__ void __launch_bounds__(256) k_deposit_3d(GridV g, PartV p, double dt, double q) {
    long ip = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (ip >= p.n) return;
    double x = p.x[ip], y = p.y[ip], z = p.z[ip];
    if ((p.dead && p.dead[ip]) || isnan(x) || isnan(y) || isnan(z)) return;
    deposit_global_3d(g, x, y, z, p.ux[ip], p.uy[ip], p.uz[ip], p.ig[ip], p.w[ip], q, dt);
}

extern "C" int lpa_interpolate_3d(const lpa_grid *g, const lpa_particles *p, void *stream) {
    LPA_REQUIRE(lpa_grid_ok(g, 3, 0) && lpa_part_ok(p, 3) && (p->n == 0 || p->part_eb[0]),
                "lpa_interpolate_3d: bad args (part_eb required)");
    if (p->n == 0) return LPA_OK;
    hipLaunchKernelGGL(k_interpolate_3d, dim3((unsigned)((p->n + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, make_gridv(g, 3), make_partv(p));
    LPA_CHECK_LAUNCH("lpa_interpolate_3d");
    return LPA_OK;
}

extern "C" int lpa_deposit_3d(const lpa_grid *g, const lpa_particles *p, double dt, double q,
                              void *stream) {
    LPA_REQUIRE(lpa_grid_ok(g, 3, 1) && lpa_part_ok(p, 3) && dt > 0, "lpa_deposit_3d: bad args");
    if (p->n == 0) return LPA_OK;
    hipLaunchKernelGGL(k_deposit_3d, dim3((unsigned)((p->n + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, make_gridv(g, 3), make_partv(p), dt, q);
    LPA_CHECK_LAUNCH("lpa_deposit_3d");
    return LPA_OK;
}

extern "C" int lpa_push_deposit_tiled_3d(const lpa_grid *g, const lpa_particles *p,
                                         const lpa_push_params *pp, const lpa_tiling *t,
                                         uint32_t *overflow, uint32_t *overflow_count, void *stream) {
    return lpa_push_deposit_tiled_part_3d(g, p, pp, t, overflow, overflow_count, LPA_PART_ALL, 0, stream);
}

static int launch_tiled_3d(const lpa_grid *g, const MultiArgs &m, const lpa_tiling *t0, bool blocks, bool defer,
                           bool rho, int part, int edge_cols, void *stream) {
    const int ntiles = t0->tiles_x * t0->tiles_y * t0->tiles_z;
    const unsigned grid = blocks ? (unsigned)t0->max_blocks : (unsigned)(8 * ((ntiles + 7) / 8));
#define LPA_LAUNCH_TILED3(D, R)                                                                                     \
    hipLaunchKernelGGL((k_push_deposit_tiled_3d<D, R>), dim3(grid), dim3(K13_THREADS), 0, (hipStream_t)stream,       \
                       make_gridv(g, 3), m, blocks ? t0->blk_tile : nullptr, blocks ? t0->blk_begin : nullptr,       \
                       blocks ? t0->blk_end : nullptr, blocks ? t0->n_blocks : nullptr, ntiles, t0->tiles_y,         \
                       t0->tiles_z, part, t0->tiles_x, edge_cols)
    if (defer && rho) LPA_LAUNCH_TILED3(true, true);
    else if (defer) LPA_LAUNCH_TILED3(true, false);
    else if (rho) LPA_LAUNCH_TILED3(false, true);
    else LPA_LAUNCH_TILED3(false, false);
#undef LPA_LAUNCH_TILED3
    return LPA_OK;
}

extern "C" int lpa_push_deposit_tiled_part_3d(const lpa_grid *g, const lpa_particles *p,
                                              const lpa_push_params *pp, const lpa_tiling *t,
                                              uint32_t *overflow, uint32_t *overflow_count, int part,
                                              int edge_cols, void *stream) {
    if (int e = check_push3(g, p, pp, "lpa_push_deposit_tiled_3d")) return e;
    LPA_REQUIRE(part == LPA_PART_ALL || ((part == LPA_PART_EDGE || part == LPA_PART_INTERIOR) && edge_cols >= 1),
                "lpa_push_deposit_tiled_part_3d: bad part / edge_cols");
    LPA_REQUIRE(t && t->blk_tile && t->blk_begin && t->blk_end && t->n_blocks && t->max_blocks > 0 &&
                    overflow && overflow_count,
                "lpa_push_deposit_tiled_3d: bad tiling");
    LPA_REQUIRE(t->tiles_x == (g->nx + T3X - 1) / T3X && t->tiles_y == (g->ny + T3Y - 1) / T3Y &&
                    t->tiles_z == (g->nz + T3Z - 1) / T3Z,
                "lpa_push_deposit_tiled_3d: tiling does not match the grid");
    LPA_REQUIRE(p->is_dead == nullptr,
                "lpa_push_deposit_tiled_3d: tile-binned stores carry no is_dead array (dead = NaN x)");
    if (t->n_sorted == 0) return LPA_OK;
    LPA_REQUIRE(p->n < (1ll << 31) - 1, "lpa_push_deposit_tiled_3d: more than 2^31 - 2 slots in one store");
    MultiArgs m;
    m.ns = 1;
    MultiSp &d = m.s[0];
    d.p = make_partv(p);
    d.k = make_pushk3(pp, g);
    bool defer = true;
    for (int c = 0; c < 8; c++) {
        d.sc.a[c] = t->scratch[c];
        defer = defer && d.sc.a[c] != nullptr;
    }
    d.tile_off = t->tile_off;
    d.overflow = overflow;
    d.overflow_count = overflow_count;
    launch_tiled_3d(g, m, t, true, defer, !(pp->flags & LPA_PUSH_NO_RHO), part, edge_cols, stream);
    LPA_CHECK_LAUNCH("lpa_push_deposit_tiled_3d");
    return LPA_OK;
}

extern "C" int lpa_push_deposit_tiled_multi_3d(const lpa_grid *g, int32_t nspecies, const lpa_particles *const *p,
                                               const lpa_push_params *const *pp, const lpa_tiling *const *t,
                                               uint32_t *const *overflow, uint32_t *const *overflow_count,
                                               void *stream) {
    return lpai_push_deposit_tiled_multi_part_3d(g, nspecies, p, pp, t, overflow, overflow_count, LPA_PART_ALL, 0, stream);
}

int lpai_push_deposit_tiled_multi_part_3d(const lpa_grid *g, int32_t nspecies, const lpa_particles *const *p,
                                          const lpa_push_params *const *pp, const lpa_tiling *const *t,
                                          uint32_t *const *overflow, uint32_t *const *overflow_count, int part, int edge_cols,
                                          void *stream) {
    LPA_REQUIRE(part == LPA_PART_ALL || ((part == LPA_PART_EDGE || part == LPA_PART_INTERIOR) && edge_cols >= 1),
                "lpa_push_deposit_tiled_multi_3d: bad part / edge_cols");
    LPA_REQUIRE(nspecies >= 1 && nspecies <= K13_MAX_SPECIES && p && pp && t && overflow && overflow_count,
                "lpa_push_deposit_tiled_multi_3d: 1 .. %d species", K13_MAX_SPECIES);
    MultiArgs m;
    m.ns = 0;
    bool defer = true;
    for (int s = 0; s < nspecies; s++) {
        if (int e = check_push3(g, p[s], pp[s], "lpa_push_deposit_tiled_multi_3d")) return e;
        LPA_REQUIRE(t[s] && t[s]->tile_off && overflow[s] && overflow_count[s], "lpa_push_deposit_tiled_multi_3d: bad tiling");
        LPA_REQUIRE(t[s]->tiles_x == (g->nx + T3X - 1) / T3X && t[s]->tiles_y == (g->ny + T3Y - 1) / T3Y &&
                        t[s]->tiles_z == (g->nz + T3Z - 1) / T3Z,
                    "lpa_push_deposit_tiled_multi_3d: tiling does not match the grid");
        LPA_REQUIRE(p[s]->is_dead == nullptr && p[s]->n < (1ll << 31) - 1,
                    "lpa_push_deposit_tiled_multi_3d: tile-binned stores carry no is_dead array and hold < 2^31 - 1 slots");
        LPA_REQUIRE(pp[s]->wrap == pp[0]->wrap && pp[s]->flags == pp[0]->flags && pp[s]->dt == pp[0]->dt,
                    "lpa_push_deposit_tiled_multi_3d: the species of one launch share wrap, flags and dt");
        if (t[s]->n_sorted == 0) continue;
        MultiSp &d = m.s[m.ns++];
        d.p = make_partv(p[s]);
        d.k = make_pushk3(pp[s], g);
        for (int c = 0; c < 8; c++) {
            d.sc.a[c] = t[s]->scratch[c];
            defer = defer && d.sc.a[c] != nullptr;
        }
        d.tile_off = t[s]->tile_off;
        d.overflow = overflow[s];
        d.overflow_count = overflow_count[s];
    }
    if (m.ns == 0) return LPA_OK;
    launch_tiled_3d(g, m, t[0], false, defer, !(pp[0]->flags & LPA_PUSH_NO_RHO), part, edge_cols, stream);
    LPA_CHECK_LAUNCH("lpa_push_deposit_tiled_multi_3d");
    return LPA_OK;
}

extern "C" int lpa_push_deposit_3d(const lpa_grid *g, const lpa_particles *p, const lpa_push_params *pp,
                                   int64_t first, int64_t count, void *stream) {
    if (int e = check_push3(g, p, pp, "lpa_push_deposit_3d")) return e;
    LPA_REQUIRE(first >= 0 && count >= 0 && first + count <= p->n, "lpa_push_deposit_3d: bad range");
    if (count == 0) return LPA_OK;
    long nb = (count + 255) / 256;
    hipLaunchKernelGGL(k_push_deposit_global_3d, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream,
                       make_gridv(g, 3), make_partv(p), make_pushk3(pp), (long)first, (long)count);
    LPA_CHECK_LAUNCH("lpa_push_deposit_3d");
    return LPA_OK;
}

// lpa_particles3d.hip -- 3-D fused particle kernel, global-memory form (any particle order):
// half push, 27-point TSC gather on the staggered Yee grid, Boris, half push, 3-D Esirkepov deposit
// with FP64 global atomics on the torus.
// Restates unified_boris_pusher_cpu_3d (core/pusher/unified/unified_pusher_3d.c:219-436) and
// current_deposit_3d_fast (core/current/current_deposit.h:275-440).
#include "lpa_common.hpp"

struct PushK3 {
    double dt, q, efactor, bfactor, cdt_half;
    int wrap;
    double lo[3], hi[3], alo[3], ahi[3];
};

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

__device__ __forceinline__ void deposit_global_3d(const GridV &g, double x, double y, double z,
                                                  double ux, double uy, double uz, double ig, double w,
                                                  double q, double dt) {
    double vx = ux * LPA_C * ig, vy = uy * LPA_C * ig, vz = uz * LPA_C * ig;
    AxisW ax, ay, az;
    axis_window(ax, x - vx * 0.5 * dt - g.x0, x + vx * 0.5 * dt - g.x0, 1.0 / g.dx);
    axis_window(ay, y - vy * 0.5 * dt - g.y0, y + vy * 0.5 * dt - g.y0, 1.0 / g.dy);
    axis_window(az, z - vz * 0.5 * dt - g.z0, z + vz * 0.5 * dt - g.z0, 1.0 / g.dz);
    const double one_third = 0.3333333333333333;  // core/utils/cutils.h:18
    double cd = (q / (g.dx * g.dy * g.dz)) * w;
    double fdx_ = (q / (g.dy * g.dz * dt)) * w;
    double fdy_ = (q / (g.dx * g.dz * dt)) * w;
    double fdz_ = (q / (g.dx * g.dy * dt)) * w;
    long rows[4];
    int cols[4], deps[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        rows[k] = (long)torus(ax.base + k + g.ng, g.NX) * g.NY * g.NZ;
        cols[k] = torus(ay.base + k + g.ng, g.NY) * g.NZ;
        deps[k] = torus(az.base + k + g.ng, g.NZ);
    }
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
                long idx = rows[i] + cols[j] + deps[k];
                double djx = xz ? 0.0 : jx_run[k][j];
                double djy = yz ? 0.0 : jy_run[k];
                double djz = zz ? 0.0 : jz_run;
                double dr = cd * ax.S1[i] * ay.S1[j] * az.S1[k];
                if (djx != 0.0) atomicAdd(&g.jx[idx], djx);
                if (djy != 0.0) atomicAdd(&g.jy[idx], djy);
                if (djz != 0.0) atomicAdd(&g.jz[idx], djz);
                if (dr != 0.0) atomicAdd(&g.rho[idx], dr);
            }
        }
    }
}

__device__ __forceinline__ double fold3(double v, double lo, double hi) {
    double L = hi - lo;
    if (v > hi) v -= L;
    if (v < lo) v += L;
    return v;
}

__global__ void __launch_bounds__(256) k_push_deposit_global_3d(GridV g, PartV p, PushK3 k, long first,
                                                                long count) {
    long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    long ip = first + t;
    double x = p.x[ip], y = p.y[ip], z = p.z[ip];
    if ((p.dead && p.dead[ip]) || isnan(x) || isnan(y) || isnan(z)) return;
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
    deposit_global_3d(g, x, y, z, ux, uy, uz, ig, w, k.q, k.dt);
    if (k.wrap & 1) x = fold3(x, k.lo[0], k.hi[0]);
    if (k.wrap & 2) y = fold3(y, k.lo[1], k.hi[1]);
    if (k.wrap & 4) z = fold3(z, k.lo[2], k.hi[2]);
    {
        bool dead = false;
        double c3[3] = {x, y, z};
#pragma unroll
        for (int a = 0; a < 3; a++)
            dead = dead || ((k.wrap & (LPA_ABSORB_X << a)) && (c3[a] < k.alo[a] || c3[a] > k.ahi[a]));
        if (dead) { x = __longlong_as_double(0x7ff8000000000000ll); y = x; z = x; }
    }
    p.x[ip] = x; p.y[ip] = y; p.z[ip] = z;
    p.ux[ip] = ux; p.uy[ip] = uy; p.uz[ip] = uz; p.ig[ip] = ig;
}

extern "C" int lpa_push_deposit_3d(const lpa_grid *g, const lpa_particles *p, const lpa_push_params *pp,
                                   int64_t first, int64_t count, void *stream) {
    LPA_REQUIRE(lpa_grid_ok(g, 3, 1), "lpa_push_deposit_3d: bad grid");
    LPA_REQUIRE(lpa_part_ok(p, 3), "lpa_push_deposit_3d: bad particle store");
    LPA_REQUIRE(pp && pp->dt > 0 && pp->m > 0, "lpa_push_deposit_3d: dt and m must be > 0");
    LPA_REQUIRE(first >= 0 && count >= 0 && first + count <= p->n, "lpa_push_deposit_3d: bad range");
    if (count == 0) return LPA_OK;
    PushK3 k;
    k.dt = pp->dt; k.q = pp->q;
    k.efactor = pp->q * pp->dt / (2 * pp->m * LPA_C);
    k.bfactor = pp->q * pp->dt / (2 * pp->m);
    k.cdt_half = LPA_C * 0.5 * pp->dt;
    k.wrap = pp->wrap;
    for (int a = 0; a < 3; a++) {
        k.lo[a] = pp->lo[a]; k.hi[a] = pp->hi[a];
        k.alo[a] = pp->alo[a]; k.ahi[a] = pp->ahi[a];
    }
    long nb = (count + 255) / 256;
    hipLaunchKernelGGL(k_push_deposit_global_3d, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream,
                       make_gridv(g, 3), make_partv(p), k, (long)first, (long)count);
    LPA_CHECK_LAUNCH("lpa_push_deposit_3d");
    return LPA_OK;
}

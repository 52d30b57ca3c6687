/*
 * picoracle.c -- TEST INFRASTRUCTURE ONLY (oracle).  Not part of the product path.
 *
 * A plain-C (IEEE-754 double, -ffp-contract=off) restatement of the CPU algorithms on
 * lambdaPIC's per-step hot path, written from the reference's behaviour, each function citing the
 * reference file:line it follows (paths relative to /root/reference/src/lambdapic/core).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library,
 * and only as the checker / the reported CPU baseline -- never as the thing shipped.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every entry point below against
 * tests/golden/*.npz, which were produced in the build container by running the reference's own
 * compiled C extensions (oracle/_ref, built by oracle/Makefile from the read-only tree) and its
 * numba-free Python FDTD (maxwell/cpu.py) on seeded inputs (tests/golden/gen_golden.py).
 *
 * Array conventions are the reference's (fields.py:24-27, utils/cutils.h:19-26): row-major
 * double[NX][NY]([NZ]) with NX = nx + 2*ng; index k in [0,n) interior, [n,n+ng) upper guard,
 * [n+ng,n+2ng) == [-ng,0) lower guard ("wrapped guard layout"); x0,y0,z0 = position of node 0.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define C_LIGHT 299792458.0          /* utils/cutils.h:17 */
/* epsilon_0 is an argument: the reference takes scipy.constants.epsilon_0 (maxwell/cpu.py:3), whose
 * value depends on the installed scipy (CODATA 2018: 8.8541878128e-12, CODATA 2022 as shipped by
 * scipy >= 1.15: 8.8541878188e-12).  Callers pass the value of the scipy the goldens were made with. */
#define ONE_THIRD 0.3333333333333333 /* utils/cutils.h:18 */

typedef unsigned char u8;

/* torus index: the reference wraps negative indices (cutils.h:19-26) and, in the deposit, both
 * directions (current/current_deposit.h:41-49) */
static inline long wrap(long i, long n) {
    while (i < 0) i += n;
    while (i >= n) i -= n;
    return i;
}

/* ---------------------------------------------------------------------------------------------
 * a1  Boris rotation, momenta in units of mc.  pusher/unified/unified_pusher_2d.c:15-51
 * ------------------------------------------------------------------------------------------- */
static inline void boris_kick(double *ux, double *uy, double *uz, double *inv_gamma,
                              double Ex, double Ey, double Ez, double Bx, double By, double Bz,
                              double efactor, double bfactor) {
    double umx = *ux + efactor * Ex;
    double umy = *uy + efactor * Ey;
    double umz = *uz + efactor * Ez;
    double ig = 1.0 / sqrt(1 + umx * umx + umy * umy + umz * umz);
    double Tx = bfactor * Bx * ig, Ty = bfactor * By * ig, Tz = bfactor * Bz * ig;
    double upx = umx + umy * Tz - umz * Ty;
    double upy = umy + umz * Tx - umx * Tz;
    double upz = umz + umx * Ty - umy * Tx;
    double Tf = 2.0 / (1 + Tx * Tx + Ty * Ty + Tz * Tz);
    double Sx = Tf * Tx, Sy = Tf * Ty, Sz = Tf * Tz;
    double uplx = umx + upy * Sz - upz * Sy;
    double uply = umy + upz * Sx - upx * Sz;
    double uplz = umz + upx * Sy - upy * Sx;
    *ux = uplx + efactor * Ex;
    *uy = uply + efactor * Ey;
    *uz = uplz + efactor * Ez;
    *inv_gamma = 1.0 / sqrt(1 + (*ux) * (*ux) + (*uy) * (*uy) + (*uz) * (*uz));
}

/* ---------------------------------------------------------------------------------------------
 * a3  TSC gather weights and the staggered 9/27-point gather.
 *     unified_pusher_2d.c:64-154, unified_pusher_3d.c:65-217
 * ------------------------------------------------------------------------------------------- */
static inline void tsc3(double d, double g[3]) {
    double d2 = d * d;
    g[0] = 0.5 * (0.25 + d2 + d);
    g[1] = 0.75 - d2;
    g[2] = 0.5 * (0.25 + d2 - d);
}

static inline double gather9(const double *f, const double fx[3], const double fy[3],
                             long ix, long iy, long NX, long NY) {
    long r0 = wrap(ix - 1, NX) * NY, r1 = wrap(ix, NX) * NY, r2 = wrap(ix + 1, NX) * NY;
    long c0 = wrap(iy - 1, NY), c1 = wrap(iy, NY), c2 = wrap(iy + 1, NY);
    return fy[0] * (fx[0] * f[r0 + c0] + fx[1] * f[r1 + c0] + fx[2] * f[r2 + c0])
         + fy[1] * (fx[0] * f[r0 + c1] + fx[1] * f[r1 + c1] + fx[2] * f[r2 + c1])
         + fy[2] * (fx[0] * f[r0 + c2] + fx[1] * f[r1 + c2] + fx[2] * f[r2 + c2]);
}

static inline double gather27(const double *f, const double fx[3], const double fy[3],
                              const double fz[3], long ix, long iy, long iz,
                              long NX, long NY, long NZ) {
    long r[3] = {wrap(ix - 1, NX) * NY * NZ, wrap(ix, NX) * NY * NZ, wrap(ix + 1, NX) * NY * NZ};
    long c[3] = {wrap(iy - 1, NY) * NZ, wrap(iy, NY) * NZ, wrap(iy + 1, NY) * NZ};
    long d[3] = {wrap(iz - 1, NZ), wrap(iz, NZ), wrap(iz + 1, NZ)};
    double acc = 0.0;
    /* evaluation order of the reference: fz[0]*(...) + fz[1]*(...) + fz[2]*(...) */
    double plane[3];
    for (int k = 0; k < 3; k++) {
        double row[3];
        for (int j = 0; j < 3; j++)
            row[j] = fx[0] * f[r[0] + c[j] + d[k]] + fx[1] * f[r[1] + c[j] + d[k]] +
                     fx[2] * f[r[2] + c[j] + d[k]];
        plane[k] = fz[k] * (fy[0] * row[0] + fy[1] * row[1] + fy[2] * row[2]);
    }
    acc = plane[0] + plane[1] + plane[2];
    return acc;
}

static inline void gather_2d(double x, double y, double eb[6], const double *const f[6],
                             double xo, double yo, long NX, long NY) {
    /* xo, yo: (x - x0) scaled to cells by the caller (inv_dx multiply in the fused kernel,
     * division in the standalone interpolator: interpolation/cpu2d.c:46-47) */
    (void)x; (void)y;
    double gx[3], gy[3], hx[3], hy[3];
    long ix1 = (long)floor(xo + 0.5), ix2 = (long)floor(xo);
    long iy1 = (long)floor(yo + 0.5), iy2 = (long)floor(yo);
    tsc3(ix1 - xo, gx);
    tsc3(ix2 - xo + 0.5, hx);
    tsc3(iy1 - yo, gy);
    tsc3(iy2 - yo + 0.5, hy);
    eb[0] = gather9(f[0], hx, gy, ix2, iy1, NX, NY); /* ex (i+1/2, j)   */
    eb[1] = gather9(f[1], gx, hy, ix1, iy2, NX, NY); /* ey (i, j+1/2)   */
    eb[2] = gather9(f[2], gx, gy, ix1, iy1, NX, NY); /* ez (i, j)       */
    eb[3] = gather9(f[3], gx, hy, ix1, iy2, NX, NY); /* bx (i, j+1/2)   */
    eb[4] = gather9(f[4], hx, gy, ix2, iy1, NX, NY); /* by (i+1/2, j)   */
    eb[5] = gather9(f[5], hx, hy, ix2, iy2, NX, NY); /* bz (i+1/2,j+1/2)*/
}

static inline void gather_3d(double eb[6], const double *const f[6], double xo, double yo,
                             double zo, long NX, long NY, long NZ) {
    double gx[3], gy[3], gz[3], hx[3], hy[3], hz[3];
    long ix1 = (long)floor(xo + 0.5), ix2 = (long)floor(xo);
    long iy1 = (long)floor(yo + 0.5), iy2 = (long)floor(yo);
    long iz1 = (long)floor(zo + 0.5), iz2 = (long)floor(zo);
    tsc3(ix1 - xo, gx); tsc3(ix2 - xo + 0.5, hx);
    tsc3(iy1 - yo, gy); tsc3(iy2 - yo + 0.5, hy);
    tsc3(iz1 - zo, gz); tsc3(iz2 - zo + 0.5, hz);
    /* stagger table: unified_pusher_3d.c:190-195 */
    eb[0] = gather27(f[0], hx, gy, gz, ix2, iy1, iz1, NX, NY, NZ);
    eb[1] = gather27(f[1], gx, hy, gz, ix1, iy2, iz1, NX, NY, NZ);
    eb[2] = gather27(f[2], gx, gy, hz, ix1, iy1, iz2, NX, NY, NZ);
    eb[3] = gather27(f[3], gx, hy, hz, ix1, iy2, iz2, NX, NY, NZ);
    eb[4] = gather27(f[4], hx, gy, hz, ix2, iy1, iz2, NX, NY, NZ);
    eb[5] = gather27(f[5], hx, hy, gz, ix2, iy2, iz1, NX, NY, NZ);
}

/* ---------------------------------------------------------------------------------------------
 * a4/a5  Esirkepov charge-conserving deposition, 5-point window, |dcell| <= 1.
 *        current/current_deposit.h:7-35 (shape), :185-268 (2-D fused "fast" grouping),
 *        :51-145 (2-D standalone grouping), :275-440 (3-D fused), current/cpu3d.c:34-116
 *        (3-D standalone).
 * ------------------------------------------------------------------------------------------- */
static inline void shape5(double d, int shift, double S[5]) {
    double d2 = d * d;
    double lo = 0.5 * (d2 + d + 0.25), mid = 0.75 - d2, hi = 0.5 * (d2 - d + 0.25);
    S[0] = S[1] = S[2] = S[3] = S[4] = 0.0;
    S[1 + shift] = lo;
    S[2 + shift] = mid;
    S[3 + shift] = hi;
}

typedef struct {
    double S0[5], S1[5], DS[5];
    long i0;   /* nearest node of the old position */
    int lo, hi; /* loop bounds [lo, hi) */
} axis_shape;

static inline void axis_setup(axis_shape *a, double r_old, double r_adv, double d) {
    double o0 = r_old / d, o1 = r_adv / d;
    long i0 = (long)floor(o0 + 0.5), i1 = (long)floor(o1 + 0.5);
    int dc = (int)(i1 - i0);
    shape5(i0 - o0, 0, a->S0);
    /* for |dc| > 1 (CFL violation) the reference's calculate_S yields an all-zero S1
     * (current_deposit.h:14-22); reproduce that rather than index out of the window */
    if (dc >= -1 && dc <= 1) shape5(i1 - o1, dc, a->S1);
    else memset(a->S1, 0, sizeof a->S1);
    for (int k = 0; k < 5; k++) a->DS[k] = a->S1[k] - a->S0[k];
    a->i0 = i0;
    a->lo = dc < 0 ? 0 : 1;
    a->hi = dc > 0 ? 5 : 4;
}

/* fused-kernel grouping of the scale factors (current_deposit.h:238-241) when fast != 0,
 * standalone grouping (current_deposit.h:104-108) otherwise */
static inline void deposit_2d(double *rho, double *jx, double *jy, double *jz,
                              double x, double y, double ux, double uy, double uz, double ig,
                              long NX, long NY, double dx, double dy, double x0, double y0,
                              double dt, double w, double q, int fast) {
    double vx = ux * C_LIGHT * ig, vy = uy * C_LIGHT * ig, vz = uz * C_LIGHT * ig;
    axis_shape ax, ay;
    axis_setup(&ax, x - vx * 0.5 * dt - x0, x + vx * 0.5 * dt - x0, dx);
    axis_setup(&ay, y - vy * 0.5 * dt - y0, y + vy * 0.5 * dt - y0, dy);

    double charge_density, factor_dx, factor_dy, factor_dt_vz;
    if (fast) {
        charge_density = (q / (dx * dy)) * w;
        factor_dx = (q / (dy * dt)) * w;
        factor_dy = (q / (dx * dt)) * w;
        factor_dt_vz = charge_density * vz;
    } else {
        charge_density = q * w / (dx * dy);
        double factor = charge_density / dt;
        factor_dx = factor * dx;
        factor_dy = factor * dy;
        factor_dt_vz = factor * dt * vz;
    }
    const double one_twelfth = 1.0 / 12.0;
    double jx_run[5] = {0, 0, 0, 0, 0};
    for (int i = ax.lo; i < ax.hi; i++) {
        double jy_run = 0.0;
        long row = wrap(ax.i0 + (i - 2), NX) * NY;
        double a = ax.S0[i] + 0.5 * ax.DS[i];
        double fdx = factor_dx * ax.DS[i];
        double t12 = one_twelfth * ax.DS[i];
        for (int j = ay.lo; j < ay.hi; j++) {
            double b = ay.S0[j] + 0.5 * ay.DS[j];
            double wy = ay.DS[j] * a;
            double wz = a * b + t12 * ay.DS[j];
            jx_run[j] -= fdx * b;
            jy_run -= factor_dy * wy;
            long idx = wrap(ay.i0 + (j - 2), NY) + row;
            jx[idx] += jx_run[j];
            jy[idx] += jy_run;
            jz[idx] += factor_dt_vz * wz;
            rho[idx] += charge_density * ax.S1[i] * ay.S1[j];
        }
    }
}

static inline void deposit_3d(double *rho, double *jx, double *jy, double *jz,
                              double x, double y, double z, double ux, double uy, double uz,
                              double ig, long NX, long NY, long NZ, double dx, double dy,
                              double dz, double x0, double y0, double z0, double dt, double w,
                              double q, int fast) {
    double vx = ux * C_LIGHT * ig, vy = uy * C_LIGHT * ig, vz = uz * C_LIGHT * ig;
    axis_shape ax, ay, az;
    axis_setup(&ax, x - vx * 0.5 * dt - x0, x + vx * 0.5 * dt - x0, dx);
    axis_setup(&ay, y - vy * 0.5 * dt - y0, y + vy * 0.5 * dt - y0, dy);
    axis_setup(&az, z - vz * 0.5 * dt - z0, z + vz * 0.5 * dt - z0, dz);
    double jx_run[5][5];
    memset(jx_run, 0, sizeof jx_run);
    if (fast) { /* current_deposit.h:275-331,398-401 */
        double charge_density = (q / (dx * dy * dz)) * w;
        double factor_dx = (q / (dy * dz * dt)) * w;
        double factor_dy = (q / (dx * dz * dt)) * w;
        double factor_dz = (q / (dx * dy * dt)) * w;
        for (int i = ax.lo; i < ax.hi; i++) {
            long pi = wrap(ax.i0 + (i - 2), NX) * NY * NZ;
            double a_x = ax.S0[i] + 0.5 * ax.DS[i];
            double c_x = 0.5 * ax.S0[i] + ONE_THIRD * ax.DS[i];
            double fdx = factor_dx * ax.DS[i];
            double jy_run[5] = {0, 0, 0, 0, 0};
            for (int j = ay.lo; j < ay.hi; j++) {
                long pj = wrap(ay.i0 + (j - 2), NY) * NZ;
                double a_y = ay.S0[j] + 0.5 * ay.DS[j];
                double c_y = 0.5 * ay.S0[j] + ONE_THIRD * ay.DS[j];
                double fdy = factor_dy * ay.DS[j];
                double tz_ij = a_x * ay.S0[j] + c_x * ay.DS[j];
                double jz_run = 0;
                for (int k = az.lo; k < az.hi; k++) {
                    long idx = wrap(az.i0 + (k - 2), NZ) + pj + pi;
                    double tjx = a_y * az.S0[k] + c_y * az.DS[k];
                    double tjy = a_x * az.S0[k] + c_x * az.DS[k];
                    jx_run[k][j] -= fdx * tjx;
                    jy_run[k] -= fdy * tjy;
                    jz_run -= factor_dz * az.DS[k] * tz_ij;
                    jx[idx] += jx_run[k][j];
                    jy[idx] += jy_run[k];
                    jz[idx] += jz_run;
                    rho[idx] += charge_density * ax.S1[i] * ay.S1[j] * az.S1[k];
                }
            }
        }
    } else { /* current/cpu3d.c:93-116 */
        double charge_density = q * w / (dx * dy * dz);
        double factor = charge_density / dt;
        for (int i = ax.lo; i < ax.hi; i++) {
            long pi = wrap(ax.i0 + (i - 2), NX) * NY * NZ;
            double jy_run[5] = {0, 0, 0, 0, 0};
            for (int j = ay.lo; j < ay.hi; j++) {
                long pj = wrap(ay.i0 + (j - 2), NY) * NZ;
                double jz_run = 0;
                for (int k = az.lo; k < az.hi; k++) {
                    long idx = wrap(az.i0 + (k - 2), NZ) + pj + pi;
                    const double *S0x = ax.S0, *S0y = ay.S0, *S0z = az.S0;
                    const double *DSx = ax.DS, *DSy = ay.DS, *DSz = az.DS;
                    double wx = DSx[i] * (S0y[j] * S0z[k] + 0.5 * DSy[j] * S0z[k] +
                                          0.5 * S0y[j] * DSz[k] + ONE_THIRD * DSy[j] * DSz[k]);
                    double wy = DSy[j] * (S0x[i] * S0z[k] + 0.5 * DSx[i] * S0z[k] +
                                          0.5 * S0x[i] * DSz[k] + ONE_THIRD * DSx[i] * DSz[k]);
                    double wz = DSz[k] * (S0x[i] * S0y[j] + 0.5 * DSx[i] * S0y[j] +
                                          0.5 * S0x[i] * DSy[j] + ONE_THIRD * DSx[i] * DSy[j]);
                    jx_run[k][j] -= factor * dx * wx;
                    jy_run[k] -= factor * dy * wy;
                    jz_run -= factor * dz * wz;
                    jx[idx] += jx_run[k][j];
                    jy[idx] += jy_run[k];
                    jz[idx] += jz_run;
                    rho[idx] += charge_density * ax.S1[i] * ay.S1[j] * az.S1[k];
                }
            }
        }
    }
}

/* ---------------------------------------------------------------------------------------------
 * a6  fused driver for one patch: half push, gather, Boris, half push, deposit; dead / NaN
 *     particles skipped.  unified_pusher_2d.c:157-365, unified_pusher_3d.c:219-436
 * ------------------------------------------------------------------------------------------- */
void orc_unified_2d(long npart, double *x, double *y, double *ux, double *uy, double *uz,
                    double *inv_gamma, const double *w, const u8 *is_dead, double *const part_eb[6],
                    const double *const eb[6], double *rho, double *jx, double *jy, double *jz,
                    long nx, long ny, long ng, double dx, double dy, double x0, double y0,
                    double dt, double q, double m) {
    const long NX = nx + 2 * ng, NY = ny + 2 * ng;
    const double efactor = q * dt / (2 * m * C_LIGHT);
    const double bfactor = q * dt / (2 * m);
    const double cdt_half = C_LIGHT * 0.5 * dt;
    const double inv_dx = 1.0 / dx, inv_dy = 1.0 / dy;
    for (long ip = 0; ip < npart; ip++) {
        if (is_dead[ip] || isnan(x[ip]) || isnan(y[ip])) continue;
        x[ip] += cdt_half * inv_gamma[ip] * ux[ip];
        y[ip] += cdt_half * inv_gamma[ip] * uy[ip];
        double f[6];
        gather_2d(x[ip], y[ip], f, eb, (x[ip] - x0) * inv_dx, (y[ip] - y0) * inv_dy, NX, NY);
        for (int c = 0; c < 6; c++) part_eb[c][ip] = f[c];
        boris_kick(&ux[ip], &uy[ip], &uz[ip], &inv_gamma[ip], f[0], f[1], f[2], f[3], f[4], f[5],
                   efactor, bfactor);
        x[ip] += cdt_half * inv_gamma[ip] * ux[ip];
        y[ip] += cdt_half * inv_gamma[ip] * uy[ip];
    }
    for (long ip = 0; ip < npart; ip++) {
        if (is_dead[ip] || isnan(x[ip]) || isnan(y[ip])) continue;
        deposit_2d(rho, jx, jy, jz, x[ip], y[ip], ux[ip], uy[ip], uz[ip], inv_gamma[ip], NX, NY,
                   dx, dy, x0, y0, dt, w[ip], q, 1);
    }
}

void orc_unified_3d(long npart, double *x, double *y, double *z, double *ux, double *uy,
                    double *uz, double *inv_gamma, const double *w, const u8 *is_dead,
                    double *const part_eb[6], const double *const eb[6], double *rho, double *jx,
                    double *jy, double *jz, long nx, long ny, long nz, long ng, double dx,
                    double dy, double dz, double x0, double y0, double z0, double dt, double q,
                    double m) {
    const long NX = nx + 2 * ng, NY = ny + 2 * ng, NZ = nz + 2 * ng;
    const double efactor = q * dt / (2 * m * C_LIGHT);
    const double bfactor = q * dt / (2 * m);
    const double cdt_half = C_LIGHT * 0.5 * dt;
    const double inv_dx = 1.0 / dx, inv_dy = 1.0 / dy, inv_dz = 1.0 / dz;
    for (long ip = 0; ip < npart; ip++) {
        if (is_dead[ip] || isnan(x[ip]) || isnan(y[ip]) || isnan(z[ip])) continue;
        x[ip] += cdt_half * inv_gamma[ip] * ux[ip];
        y[ip] += cdt_half * inv_gamma[ip] * uy[ip];
        z[ip] += cdt_half * inv_gamma[ip] * uz[ip];
        double f[6];
        gather_3d(f, eb, (x[ip] - x0) * inv_dx, (y[ip] - y0) * inv_dy, (z[ip] - z0) * inv_dz, NX,
                  NY, NZ);
        for (int c = 0; c < 6; c++) part_eb[c][ip] = f[c];
        boris_kick(&ux[ip], &uy[ip], &uz[ip], &inv_gamma[ip], f[0], f[1], f[2], f[3], f[4], f[5],
                   efactor, bfactor);
        x[ip] += cdt_half * inv_gamma[ip] * ux[ip];
        y[ip] += cdt_half * inv_gamma[ip] * uy[ip];
        z[ip] += cdt_half * inv_gamma[ip] * uz[ip];
        deposit_3d(rho, jx, jy, jz, x[ip], y[ip], z[ip], ux[ip], uy[ip], uz[ip], inv_gamma[ip],
                   NX, NY, NZ, dx, dy, dz, x0, y0, z0, dt, w[ip], q, 1);
    }
}

/* multi-patch OpenMP drivers (one patch per iteration, static schedule, as the reference:
 * unified_pusher_2d.c:213-214).  ptr tables: [npatches] of the per-patch arrays. */
void orc_unified_2d_patches(long npatches, const long *npart, double **x, double **y, double **ux,
                            double **uy, double **uz, double **inv_gamma, double **w, u8 **is_dead,
                            double **part_eb /* [npatches*6] */, double **fields /* [npatches*10] */,
                            const double *x0, const double *y0, long nx, long ny, long ng, double dx,
                            double dy, double dt, double q, double m) {
#pragma omp parallel for schedule(static)
    for (long p = 0; p < npatches; p++) {
        double **f = fields + 10 * p;
        const double *eb[6] = {f[0], f[1], f[2], f[3], f[4], f[5]};
        orc_unified_2d(npart[p], x[p], y[p], ux[p], uy[p], uz[p], inv_gamma[p], w[p], is_dead[p],
                       part_eb + 6 * p, eb, f[9], f[6], f[7], f[8], nx, ny, ng, dx, dy, x0[p],
                       y0[p], dt, q, m);
    }
}

void orc_unified_3d_patches(long npatches, const long *npart, double **x, double **y, double **z,
                            double **ux, double **uy, double **uz, double **inv_gamma, double **w,
                            u8 **is_dead, double **part_eb, double **fields, const double *x0,
                            const double *y0, const double *z0, long nx, long ny, long nz, long ng,
                            double dx, double dy, double dz, double dt, double q, double m) {
#pragma omp parallel for schedule(static)
    for (long p = 0; p < npatches; p++) {
        double **f = fields + 10 * p;
        const double *eb[6] = {f[0], f[1], f[2], f[3], f[4], f[5]};
        orc_unified_3d(npart[p], x[p], y[p], z[p], ux[p], uy[p], uz[p], inv_gamma[p], w[p],
                       is_dead[p], part_eb + 6 * p, eb, f[9], f[6], f[7], f[8], nx, ny, nz, ng, dx,
                       dy, dz, x0[p], y0[p], z0[p], dt, q, m);
    }
}

/* ---------------------------------------------------------------------------------------------
 * split (non-fused) kernels of the callback-in-pusher-stage path
 * ------------------------------------------------------------------------------------------- */
/* interpolation/cpu2d.c:32-136 : divides by dx (not inv_dx), skips only is_dead */
void orc_interpolate_2d(long npart, const double *x, const double *y, const u8 *is_dead,
                        double *const part_eb[6], const double *const eb[6], long nx, long ny,
                        long ng, double dx, double dy, double x0, double y0) {
    const long NX = nx + 2 * ng, NY = ny + 2 * ng;
    for (long ip = 0; ip < npart; ip++) {
        if (is_dead[ip]) continue;
        double f[6];
        gather_2d(x[ip], y[ip], f, eb, (x[ip] - x0) / dx, (y[ip] - y0) / dy, NX, NY);
        for (int c = 0; c < 6; c++) part_eb[c][ip] = f[c];
    }
}

void orc_interpolate_3d(long npart, const double *x, const double *y, const double *z,
                        const u8 *is_dead, double *const part_eb[6], const double *const eb[6],
                        long nx, long ny, long nz, long ng, double dx, double dy, double dz,
                        double x0, double y0, double z0) {
    const long NX = nx + 2 * ng, NY = ny + 2 * ng, NZ = nz + 2 * ng;
    for (long ip = 0; ip < npart; ip++) {
        if (is_dead[ip]) continue;
        double f[6];
        gather_3d(f, eb, (x[ip] - x0) / dx, (y[ip] - y0) / dy, (z[ip] - z0) / dz, NX, NY, NZ);
        for (int c = 0; c < 6; c++) part_eb[c][ip] = f[c];
    }
}

/* pusher/boris.py:6-48 + pusher/cpu.py:11-35 (Boris on stored *_part), skips is_dead */
void orc_boris(long npart, double *ux, double *uy, double *uz, double *inv_gamma,
               const double *const part_eb[6], const u8 *is_dead, double q, double m, double dt) {
    const double efactor = q * dt / (2 * m * C_LIGHT);
    const double bfactor = q * dt / (2 * m);
    for (long ip = 0; ip < npart; ip++) {
        if (is_dead[ip]) continue;
        boris_kick(&ux[ip], &uy[ip], &uz[ip], &inv_gamma[ip], part_eb[0][ip], part_eb[1][ip],
                   part_eb[2][ip], part_eb[3][ip], part_eb[4][ip], part_eb[5][ip], efactor, bfactor);
    }
}

/* pusher/cpu.py:58-91 : x += c*dt*inv_gamma*u  (dt = half step given by the caller) */
void orc_push_position_2d(long npart, double *x, double *y, const double *ux, const double *uy,
                          const double *inv_gamma, const u8 *is_dead, double dt) {
    const double cdt = C_LIGHT * dt;
    for (long ip = 0; ip < npart; ip++) {
        if (is_dead[ip]) continue;
        x[ip] += cdt * inv_gamma[ip] * ux[ip];
        y[ip] += cdt * inv_gamma[ip] * uy[ip];
    }
}

/* current/cpu2d.c:74-184 (standalone Esirkepov, non-fast grouping), skips is_dead / NaN */
void orc_deposit_2d(long npart, const double *x, const double *y, const double *ux,
                    const double *uy, const double *uz, const double *inv_gamma, const double *w,
                    const u8 *is_dead, double *rho, double *jx, double *jy, double *jz, long nx,
                    long ny, long ng, double dx, double dy, double x0, double y0, double dt,
                    double q) {
    const long NX = nx + 2 * ng, NY = ny + 2 * ng;
    for (long ip = 0; ip < npart; ip++) {
        if (is_dead[ip] || isnan(x[ip]) || isnan(y[ip])) continue;
        deposit_2d(rho, jx, jy, jz, x[ip], y[ip], ux[ip], uy[ip], uz[ip], inv_gamma[ip], NX, NY,
                   dx, dy, x0, y0, dt, w[ip], q, 0);
    }
}

void orc_deposit_3d(long npart, const double *x, const double *y, const double *z,
                    const double *ux, const double *uy, const double *uz, const double *inv_gamma,
                    const double *w, const u8 *is_dead, double *rho, double *jx, double *jy,
                    double *jz, long nx, long ny, long nz, long ng, double dx, double dy, double dz,
                    double x0, double y0, double z0, double dt, double q) {
    const long NX = nx + 2 * ng, NY = ny + 2 * ng, NZ = nz + 2 * ng;
    for (long ip = 0; ip < npart; ip++) {
        if (is_dead[ip] || isnan(x[ip]) || isnan(y[ip]) || isnan(z[ip])) continue;
        deposit_3d(rho, jx, jy, jz, x[ip], y[ip], z[ip], ux[ip], uy[ip], uz[ip], inv_gamma[ip],
                   NX, NY, NZ, dx, dy, dz, x0, y0, z0, dt, w[ip], q, 0);
    }
}

/* ---------------------------------------------------------------------------------------------
 * a7  Yee FDTD half-step updates over the patch interior.  maxwell/cpu.py:9-35 (2-D),
 *     :83-112 (3-D).  i-1 / j-1 at 0 reads the lower guard through the negative-index wrap,
 *     i+1 at n-1 reads the upper guard at index n.
 * ------------------------------------------------------------------------------------------- */
#define I2(i, j) (wrap((i), NX) * NY + wrap((j), NY))
void orc_fdtd_e_2d(double *ex, double *ey, double *ez, const double *bx, const double *by,
                   const double *bz, const double *jx, const double *jy, const double *jz, long nx,
                   long ny, long ng, double dx, double dy, double dt, double eps0) {
    const long NX = nx + 2 * ng, NY = ny + 2 * ng;
    const double bfactor = dt * (C_LIGHT * C_LIGHT);
    const double jfactor = dt / eps0;
    for (long i = 0; i < nx; i++)
        for (long j = 0; j < ny; j++) {
            long c = I2(i, j), xm = I2(i - 1, j), ym = I2(i, j - 1);
            ex[c] += bfactor * ((bz[c] - bz[ym]) / dy) - jfactor * jx[c];
            ey[c] += bfactor * (-(bz[c] - bz[xm]) / dx) - jfactor * jy[c];
            ez[c] += bfactor * ((by[c] - by[xm]) / dx - (bx[c] - bx[ym]) / dy) - jfactor * jz[c];
        }
}

void orc_fdtd_b_2d(const double *ex, const double *ey, const double *ez, double *bx, double *by,
                   double *bz, long nx, long ny, long ng, double dx, double dy, double dt) {
    const long NX = nx + 2 * ng, NY = ny + 2 * ng;
    for (long i = 0; i < nx; i++)
        for (long j = 0; j < ny; j++) {
            long c = I2(i, j), xp = I2(i + 1, j), yp = I2(i, j + 1);
            bx[c] -= dt * ((ez[yp] - ez[c]) / dy);
            by[c] -= dt * (-(ez[xp] - ez[c]) / dx);
            bz[c] -= dt * ((ey[xp] - ey[c]) / dx - (ex[yp] - ex[c]) / dy);
        }
}

#define I3(i, j, k) ((wrap((i), NX) * NY + wrap((j), NY)) * NZ + wrap((k), NZ))
void orc_fdtd_e_3d(double *ex, double *ey, double *ez, const double *bx, const double *by,
                   const double *bz, const double *jx, const double *jy, const double *jz, long nx,
                   long ny, long nz, long ng, double dx, double dy, double dz, double dt, double eps0) {
    const long NX = nx + 2 * ng, NY = ny + 2 * ng, NZ = nz + 2 * ng;
    const double bfactor = dt * (C_LIGHT * C_LIGHT);
    const double jfactor = dt / eps0;
#pragma omp parallel for schedule(static)
    for (long i = 0; i < nx; i++)
        for (long j = 0; j < ny; j++)
            for (long k = 0; k < nz; k++) {
                long c = I3(i, j, k), xm = I3(i - 1, j, k), ym = I3(i, j - 1, k),
                     zm = I3(i, j, k - 1);
                ex[c] += bfactor * ((bz[c] - bz[ym]) / dy - (by[c] - by[zm]) / dz) - jfactor * jx[c];
                ey[c] += bfactor * ((bx[c] - bx[zm]) / dz - (bz[c] - bz[xm]) / dx) - jfactor * jy[c];
                ez[c] += bfactor * ((by[c] - by[xm]) / dx - (bx[c] - bx[ym]) / dy) - jfactor * jz[c];
            }
}

void orc_fdtd_b_3d(const double *ex, const double *ey, const double *ez, double *bx, double *by,
                   double *bz, long nx, long ny, long nz, long ng, double dx, double dy, double dz,
                   double dt) {
    const long NX = nx + 2 * ng, NY = ny + 2 * ng, NZ = nz + 2 * ng;
#pragma omp parallel for schedule(static)
    for (long i = 0; i < nx; i++)
        for (long j = 0; j < ny; j++)
            for (long k = 0; k < nz; k++) {
                long c = I3(i, j, k), xp = I3(i + 1, j, k), yp = I3(i, j + 1, k),
                     zp = I3(i, j, k + 1);
                bx[c] -= dt * ((ez[yp] - ez[c]) / dy - (ey[zp] - ey[c]) / dz);
                by[c] -= dt * ((ex[zp] - ex[c]) / dz - (ez[xp] - ez[c]) / dx);
                bz[c] -= dt * ((ey[xp] - ey[c]) / dx - (ex[yp] - ex[c]) / dy);
            }
}

/* multi-patch FDTD drivers for the CPU baseline (maxwell/cpu.py:38-79: prange over patches) */
void orc_fdtd_e_2d_patches(long npatches, double **fields, long nx, long ny, long ng, double dx,
                           double dy, double dt, double eps0) {
#pragma omp parallel for schedule(static)
    for (long p = 0; p < npatches; p++) {
        double **f = fields + 10 * p;
        orc_fdtd_e_2d(f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7], f[8], nx, ny, ng, dx, dy, dt, eps0);
    }
}
void orc_fdtd_b_2d_patches(long npatches, double **fields, long nx, long ny, long ng, double dx,
                           double dy, double dt) {
#pragma omp parallel for schedule(static)
    for (long p = 0; p < npatches; p++) {
        double **f = fields + 10 * p;
        orc_fdtd_b_2d(f[0], f[1], f[2], f[3], f[4], f[5], nx, ny, ng, dx, dy, dt);
    }
}

/* ---------------------------------------------------------------------------------------------
 * a9  bucket index of the cell sort.  sort/cpu2d.c:9-54 : bucket = floor((x - x0)/dx) with x0
 *     already shifted by -dx/2 by the caller (sort/particle_sort.py:193); out-of-range -> last
 *     bucket; dead particles inherit the previous particle's bucket; optional mirrored x.
 * ------------------------------------------------------------------------------------------- */
void orc_bucket_index_2d(long npart, const double *x, const double *y, const u8 *is_dead, long nx,
                         long ny, double dx, double dy, double x0, double y0, int64_t *index,
                         int64_t *count, int reverse_x) {
    long nbin = nx * ny, cur = 0;
    memset(count, 0, sizeof(int64_t) * (size_t)nbin);
    for (long ip = 0; ip < npart; ip++) {
        if (!is_dead[ip]) {
            long ix = (long)floor((x[ip] - x0) / dx), iy = (long)floor((y[ip] - y0) / dy);
            if (reverse_x) {
                if (ix < 0) ix = 0; else if (ix >= nx) ix = nx - 1;
                if (iy < 0) iy = 0; else if (iy >= ny) iy = ny - 1;
                cur = iy + (nx - 1 - ix) * ny;
            } else if (0 <= ix && ix < nx && 0 <= iy && iy < ny) {
                cur = iy + ix * ny;
            } else {
                cur = nbin - 1;
            }
        }
        index[ip] = cur;
        count[cur] += 1;
    }
}

/* ---------------------------------------------------------------------------------------------
 * a10 / a11  intra-rank guard copy and current fold over a patch set (OpenMP over patches, like
 *            patch/sync_fields2d.c:62-63,172-173).  Same semantics as oracle/sync.py (which the tests
 *            pin to the reference); this C form exists so that the CPU baseline is not timed through
 *            numpy.  `arr[p]` = one attribute's array of patch p (wrapped guard layout), `nb[8*p+b]`
 *            = neighbour patch across boundary b (Boundary2D order) or -1.
 * ------------------------------------------------------------------------------------------- */
static const int SIDE_X[8] = {-1, 1, 0, 0, -1, 1, -1, 1};
static const int SIDE_Y[8] = {0, 0, -1, 1, -1, -1, 1, 1};

static inline void range_guard(int side, long n, long ng, long *dst0, long *src0, long *len) {
    if (side == 0) { *dst0 = 0; *src0 = 0; *len = n; }
    else if (side < 0) { *dst0 = -ng; *src0 = n - ng; *len = ng; }  /* my low guard <- nb's top edge */
    else { *dst0 = n; *src0 = 0; *len = ng; }                       /* my high guard <- nb's bottom edge */
}
static inline void range_fold(int side, long n, long ng, long *dst0, long *src0, long *len) {
    if (side == 0) { *dst0 = 0; *src0 = 0; *len = n; }
    else if (side < 0) { *dst0 = 0; *src0 = n; *len = ng; }         /* my low edge += nb's high guard */
    else { *dst0 = n - ng; *src0 = -ng; *len = ng; }                /* my high edge += nb's low guard */
}

void orc_sync_guard_2d_patches(long npatches, double **arr, const int64_t *nb, long nx, long ny,
                               long ng) {
    const long NX = nx + 2 * ng, NY = ny + 2 * ng;
#pragma omp parallel for schedule(static)
    for (long p = 0; p < npatches; p++)
        for (int b = 0; b < 8; b++) {
            long q = nb[8 * p + b];
            if (q < 0) continue;
            long dx0, sx0, lx, dy0, sy0, ly;
            range_guard(SIDE_X[b], nx, ng, &dx0, &sx0, &lx);
            range_guard(SIDE_Y[b], ny, ng, &dy0, &sy0, &ly);
            for (long i = 0; i < lx; i++)
                for (long j = 0; j < ly; j++)
                    arr[p][I2(dx0 + i, dy0 + j)] = arr[q][I2(sx0 + i, sy0 + j)];
        }
}

void orc_sync_currents_2d_patches(long npatches, double **arr, const int64_t *nb, long nx, long ny,
                                  long ng) {
    const long NX = nx + 2 * ng, NY = ny + 2 * ng;
#pragma omp parallel for schedule(static)
    for (long p = 0; p < npatches; p++)
        for (int b = 0; b < 8; b++) {
            long q = nb[8 * p + b];
            if (q < 0) continue;
            long dx0, sx0, lx, dy0, sy0, ly;
            range_fold(SIDE_X[b], nx, ng, &dx0, &sx0, &lx);
            range_fold(SIDE_Y[b], ny, ng, &dy0, &sy0, &ly);
            for (long i = 0; i < lx; i++)
                for (long j = 0; j < ly; j++) {
                    arr[p][I2(dx0 + i, dy0 + j)] += arr[q][I2(sx0 + i, sy0 + j)];
                    arr[q][I2(sx0 + i, sy0 + j)] = 0.0;
                }
        }
}

void orc_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

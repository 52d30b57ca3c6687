// lpa_fields.hip -- grid-side kernels: Yee FDTD half steps, current reset, periodic guard wrap,
// current fold, x-face halo pack/unpack, field diagnostics.  All are HBM-bound streaming kernels:
// one thread per cell, threads consecutive along the fastest (last) axis so every wave reads and
// writes whole 512-B rows.
#include <stdarg.h>

#include "lpa_common.hpp"
#include "lpa_fold.hpp"
#include "lpa_tail.hpp"

// ---- error string ----------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void lpa_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
extern "C" const char *lpa_last_error(void) { return g_err; }
extern "C" int lpa_version(void) { return 100; }

// ---- guard wrap fused into the field updates (lpa_step): an interior cell within ng of a face of an axis in `axes`
// (the axes that are periodic INSIDE this slab) also stores its new value in the guard cell(s) it is the periodic
// image source of -- what lpa_guard_wrap would copy in a launch of its own right after the update.  The E sweep reads
// no E neighbour and the B sweep no B neighbour, so writing the guards inside the sweep races with nothing.
__device__ __forceinline__ void store3_wrapped(const GridV &g, double *fa, double *fb, double *fc, long c, int i, int j,
                                               int k, int axes, double va, double vb, double vc) {
    fa[c] = va; fb[c] = vb; fc[c] = vc;
    axes &= 7;                       // (bit 8 of the sweeps' `wrap` argument: FDTD_TWICE)
    if (!axes) return;
    const int ng = g.ng;
    const bool d3 = g.NZ > 1;
    int ox[3] = {0, ((axes & 1) && i < ng) ? g.nx : 0, ((axes & 1) && i >= g.nx - ng) ? -g.nx : 0};
    int oy[3] = {0, ((axes & 2) && j < ng) ? g.ny : 0, ((axes & 2) && j >= g.ny - ng) ? -g.ny : 0};
    int oz[3] = {0, (d3 && (axes & 4) && k < ng) ? g.nz : 0, (d3 && (axes & 4) && k >= g.nz - ng) ? -g.nz : 0};
    if (!(ox[1] | ox[2] | oy[1] | oy[2] | oz[1] | oz[2])) return;
    const long sY = g.NZ, sX = (long)g.NY * g.NZ;
    for (int a = 0; a < 3; a++) {
        if (a && !ox[a]) continue;
        for (int b = 0; b < 3; b++) {
            if (b && !oy[b]) continue;
            for (int e = 0; e < 3; e++) {
                if ((e && !oz[e]) || !(a | b | e)) continue;
                const long t = c + ox[a] * sX + oy[b] * sY + oz[e];
                fa[t] = va; fb[t] = vb; fc[t] = vc;
            }
        }
    }
}

// bit 8 of the E sweeps' `wrap` argument: TWO half steps in one sweep (lpa_step, LPA_STEP_E1_DOUBLE: the E half step
// that ended the previous step was left out -- no reader in between -- and is done here together with this step's
// first one: B and J are the same for both, every cell's update reads only its own E, so the two sequential updates in
// registers are the two sweeps, bit for bit, for a third of the E traffic of the pair)
constexpr int FDTD_TWICE = 256;

// bits 3-4 / 5-6 of the B sweeps' `wrap` argument: the sweep also advances that many x GUARD planes below node 0 / above
// node nx - 1 (lpa_step, LPA_STEP_B_EXT_*).  A B update reads E at the node and one node up, so with all ng E guard planes
// current (they are exchanged after every E sweep) B is exact on the ng low and the ng - 1 high guard planes -- the same
// arithmetic on the same E values the neighbour slab performs on its interior -- and never has to travel: two of a slab
// step's four message rounds are gone.  (psi arrays of the y / z CPML layers carry x guard rows for this, see the engines.)
constexpr int FDTD_EXT_LO_SHIFT = 3, FDTD_EXT_HI_SHIFT = 5;
// bits 9 / 10: the B sweep in two parts (lpa_step, overlapped slab steps) -- FDTD_B_INTERIOR: nodes [0, nx - 1), the rows
// that read no E guard plane; FDTD_B_EDGE: node nx - 1 (it reads E at node nx) and the x guard planes of FDTD_EXT_*: what has
// to wait for the E guard planes, launched behind their exchange on the communicator's second stream while the first part,
// the current reset and the interior tiles of the push run on the caller's
constexpr int FDTD_B_INTERIOR = 512, FDTD_B_EDGE = 1024;
// x node of a B sweep's block row `b`
__device__ __forceinline__ int fdtd_b_row(int b, int wrap, int nx) {
    const int lo = (wrap >> FDTD_EXT_LO_SHIFT) & 3;
    if (wrap & FDTD_B_EDGE) return b < lo ? b - lo : nx - 1 + (b - lo);
    return b - lo;
}
static int fdtd_b_rows(int wrap, int nx) {      // block rows of the launch
    const int ext = ((wrap >> FDTD_EXT_LO_SHIFT) & 3) + ((wrap >> FDTD_EXT_HI_SHIFT) & 3);
    if (wrap & FDTD_B_EDGE) return ext + 1;
    if (wrap & FDTD_B_INTERIOR) return nx - 1;
    return nx + ext;
}

// =====================================================================================================
// FDTD.  Restates update_efield_2d / update_bfield_2d (core/maxwell/cpu.py:9-35) on the conventional
// layout: interior node (i,j) is at [i+ng][j+ng]; i-1 at i=0 is the low guard, i+1 at nx-1 the high one.
// AI: E sweep 13 loads + 3 stores per cell (104+24 B, 17 flop) -> HBM bound.
// =====================================================================================================
__global__ void __launch_bounds__(256) k_fdtd_e_2d(GridV g, double bfac, double jfac, int wrap) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    int i = blockIdx.y;
    if (j >= g.ny) return;
    long c = (long)(i + g.ng) * g.NY + (j + g.ng);
    long xm = c - g.NY, ym = c - 1;
    double bzc = g.bz[c];
    const double kx = bfac * ((bzc - g.bz[ym]) / g.dy) - jfac * g.jx[c];
    const double ky = bfac * (-(bzc - g.bz[xm]) / g.dx) - jfac * g.jy[c];
    const double kz = bfac * ((g.by[c] - g.by[xm]) / g.dx - (g.bx[c] - g.bx[ym]) / g.dy) - jfac * g.jz[c];
    double ex = g.ex[c] + kx, ey = g.ey[c] + ky, ez = g.ez[c] + kz;
    if (wrap & FDTD_TWICE) { ex += kx; ey += ky; ez += kz; }
    store3_wrapped(g, g.ex, g.ey, g.ez, c, i, j, 0, wrap, ex, ey, ez);
}

__global__ void __launch_bounds__(256) k_fdtd_b_2d(GridV g, double dt, int wrap, BTail tail) {
    b_tail(g, tail);
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    int i = fdtd_b_row((int)blockIdx.y, wrap, g.nx);
    if (j >= g.ny) return;
    long c = (long)(i + g.ng) * g.NY + (j + g.ng);
    long xp = c + g.NY, yp = c + 1;
    double ezc = g.ez[c];
    double bx = g.bx[c] - dt * ((g.ez[yp] - ezc) / g.dy);
    double by = g.by[c] - dt * (-(g.ez[xp] - ezc) / g.dx);
    double bz = g.bz[c] - dt * ((g.ey[xp] - g.ey[c]) / g.dx - (g.ex[yp] - g.ex[c]) / g.dy);
    store3_wrapped(g, g.bx, g.by, g.bz, c, i, j, 0, wrap, bx, by, bz);
}

// 3-D (core/maxwell/cpu.py:83-112): grid (ceil(nz/256), ny, nx)
__global__ void __launch_bounds__(256) k_fdtd_e_3d(GridV g, double bfac, double jfac, int wrap) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    int j = blockIdx.y, i = blockIdx.z;
    if (k >= g.nz) return;
    long sy = g.NZ, sx = (long)g.NY * g.NZ;
    long c = (long)(i + g.ng) * sx + (long)(j + g.ng) * sy + (k + g.ng);
    long xm = c - sx, ym = c - sy, zm = c - 1;
    double bxc = g.bx[c], byc = g.by[c], bzc = g.bz[c];
    const double kx = bfac * ((bzc - g.bz[ym]) / g.dy - (byc - g.by[zm]) / g.dz) - jfac * g.jx[c];
    const double ky = bfac * ((bxc - g.bx[zm]) / g.dz - (bzc - g.bz[xm]) / g.dx) - jfac * g.jy[c];
    const double kz = bfac * ((byc - g.by[xm]) / g.dx - (bxc - g.bx[ym]) / g.dy) - jfac * g.jz[c];
    double ex = g.ex[c] + kx, ey = g.ey[c] + ky, ez = g.ez[c] + kz;
    if (wrap & FDTD_TWICE) { ex += kx; ey += ky; ez += kz; }
    store3_wrapped(g, g.ex, g.ey, g.ez, c, i, j, k, wrap, ex, ey, ez);
}

__global__ void __launch_bounds__(256) k_fdtd_b_3d(GridV g, double dt, int wrap, BTail tail) {
    b_tail(g, tail);
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    int j = blockIdx.y, i = fdtd_b_row((int)blockIdx.z, wrap, g.nx);
    if (k >= g.nz) return;
    long sy = g.NZ, sx = (long)g.NY * g.NZ;
    long c = (long)(i + g.ng) * sx + (long)(j + g.ng) * sy + (k + g.ng);
    long xp = c + sx, yp = c + sy, zp = c + 1;
    double exc = g.ex[c], eyc = g.ey[c], ezc = g.ez[c];
    double bx = g.bx[c] - dt * ((g.ez[yp] - ezc) / g.dy - (g.ey[zp] - eyc) / g.dz);
    double by = g.by[c] - dt * ((g.ex[zp] - exc) / g.dz - (g.ez[xp] - ezc) / g.dx);
    double bz = g.bz[c] - dt * ((g.ey[xp] - eyc) / g.dx - (g.ex[yp] - exc) / g.dy);
    store3_wrapped(g, g.bx, g.by, g.bz, c, i, j, k, wrap, bx, by, bz);
}

static int fdtd_e_2d(const lpa_grid *g, double dt, double eps0, int wrap, void *stream) {
    LPA_REQUIRE(lpa_grid_ok(g, 2, 1), "lpa_fdtd_e_2d: bad grid");
    LPA_REQUIRE(eps0 > 0, "lpa_fdtd_e_2d: eps0 must be > 0");
    GridV v = make_gridv(g, 2);
    dim3 grid((g->ny + 255) / 256, g->nx);
    hipLaunchKernelGGL(k_fdtd_e_2d, grid, dim3(256), 0, (hipStream_t)stream, v,
                       dt * (LPA_C * LPA_C), dt / eps0, wrap);
    LPA_CHECK_LAUNCH("lpa_fdtd_e_2d");
    return LPA_OK;
}
extern "C" int lpa_fdtd_e_2d(const lpa_grid *g, double dt, double eps0, void *stream) { return fdtd_e_2d(g, dt, eps0, 0, stream); }

static int fdtd_b_2d(const lpa_grid *g, double dt, int wrap, void *stream, const BTail &tail = BTail{}) {
    LPA_REQUIRE(lpa_grid_ok(g, 2, 0), "lpa_fdtd_b_2d: bad grid");
    GridV v = make_gridv(g, 2);
    dim3 grid((g->ny + 255) / 256, fdtd_b_rows(wrap, g->nx));
    hipLaunchKernelGGL(k_fdtd_b_2d, grid, dim3(256), 0, (hipStream_t)stream, v, dt, wrap, tail);
    LPA_CHECK_LAUNCH("lpa_fdtd_b_2d");
    return LPA_OK;
}
extern "C" int lpa_fdtd_b_2d(const lpa_grid *g, double dt, void *stream) { return fdtd_b_2d(g, dt, 0, stream); }

static int fdtd_e_3d(const lpa_grid *g, double dt, double eps0, int wrap, void *stream) {
    LPA_REQUIRE(lpa_grid_ok(g, 3, 1), "lpa_fdtd_e_3d: bad grid");
    LPA_REQUIRE(eps0 > 0, "lpa_fdtd_e_3d: eps0 must be > 0");
    LPA_REQUIRE(g->ny <= 65535 && g->nx <= 65535, "lpa_fdtd_e_3d: nx, ny must be <= 65535");
    GridV v = make_gridv(g, 3);
    dim3 grid((g->nz + 255) / 256, g->ny, g->nx);
    hipLaunchKernelGGL(k_fdtd_e_3d, grid, dim3(256), 0, (hipStream_t)stream, v,
                       dt * (LPA_C * LPA_C), dt / eps0, wrap);
    LPA_CHECK_LAUNCH("lpa_fdtd_e_3d");
    return LPA_OK;
}
extern "C" int lpa_fdtd_e_3d(const lpa_grid *g, double dt, double eps0, void *stream) { return fdtd_e_3d(g, dt, eps0, 0, stream); }

static int fdtd_b_3d(const lpa_grid *g, double dt, int wrap, void *stream, const BTail &tail = BTail{}) {
    LPA_REQUIRE(lpa_grid_ok(g, 3, 0), "lpa_fdtd_b_3d: bad grid");
    LPA_REQUIRE(g->ny <= 65535 && g->nx <= 65535, "lpa_fdtd_b_3d: nx, ny must be <= 65535");
    GridV v = make_gridv(g, 3);
    dim3 grid((g->nz + 255) / 256, g->ny, fdtd_b_rows(wrap, g->nx));
    hipLaunchKernelGGL(k_fdtd_b_3d, grid, dim3(256), 0, (hipStream_t)stream, v, dt, wrap, tail);
    LPA_CHECK_LAUNCH("lpa_fdtd_b_3d");
    return LPA_OK;
}
extern "C" int lpa_fdtd_b_3d(const lpa_grid *g, double dt, void *stream) { return fdtd_b_3d(g, dt, 0, stream); }

// =====================================================================================================
// CPML absorbing layers (core/boundary/cpml.py).  The reference attaches PML objects to edge patches and
// runs (a) a kappa-scaled Yee update on every patch that has one (cpml.py:343-377; kappa == 1 outside
// the layer, so it equals the plain update there) and (b) the psi recursions on the layer's cells
// (cpml.py:531-606).  On one slab per rank this is one kappa-scaled sweep with per-axis kappa arrays
// over the whole slab + one small kernel per layer.  bcoeff / ccoeff_d (cpml.py:537-538) depend only on
// the cell index along the layer normal and on dt: the host passes them as arrays.
// =====================================================================================================
__global__ void __launch_bounds__(256) k_fdtd_e_cpml_2d(GridV g, double bfac, double jfac,
                                                        const double *__restrict__ kx,
                                                        const double *__restrict__ ky) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    int i = blockIdx.y;
    if (j >= g.ny) return;
    long c = (long)(i + g.ng) * g.NY + (j + g.ng);
    long xm = c - g.NY, ym = c - 1;
    double bfx = bfac / kx[i], bfy = bfac / ky[j];
    double bzc = g.bz[c];
    g.ex[c] += bfy * ((bzc - g.bz[ym]) / g.dy) - jfac * g.jx[c];
    g.ey[c] += bfx * (-(bzc - g.bz[xm]) / g.dx) - jfac * g.jy[c];
    g.ez[c] += bfx * ((g.by[c] - g.by[xm]) / g.dx) - bfy * ((g.bx[c] - g.bx[ym]) / g.dy) - jfac * g.jz[c];
}

__global__ void __launch_bounds__(256) k_fdtd_b_cpml_2d(GridV g, double dt, const double *__restrict__ kx,
                                                        const double *__restrict__ ky) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    int i = blockIdx.y;
    if (j >= g.ny) return;
    long c = (long)(i + g.ng) * g.NY + (j + g.ng);
    long xp = c + g.NY, yp = c + 1;
    double efx = dt / kx[i], efy = dt / ky[j];
    double ezc = g.ez[c];
    g.bx[c] -= efy * ((g.ez[yp] - ezc) / g.dy);
    g.by[c] -= efx * (-(g.ez[xp] - ezc) / g.dx);
    g.bz[c] -= efx * ((g.ey[xp] - g.ey[c]) / g.dx) - efy * ((g.ex[yp] - g.ex[c]) / g.dy);
}

// psi recursion of one layer.  AXIS 0: cells (ipos, t) with ipos in [start, stop), t over ny;
// AXIS 1: cells (t, ipos) with t over nx.  psi is compact: [stop-start][ny] resp. [nx][stop-start].
// EFIELD: psi_a/psi_b = (psi_ey_x, psi_ez_x) or (psi_ex_y, psi_ez_y); else the B twins.
template <int AXIS, bool EFIELD>
__global__ void __launch_bounds__(256) k_cpml_psi_2d(GridV g, int start, int stop, double fac,
                                                     const double *__restrict__ bco,
                                                     const double *__restrict__ cco, double *psi_a,
                                                     double *psi_b) {
    int nt = AXIS == 0 ? g.ny : g.nx, nl = stop - start;
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    int l = blockIdx.y;
    if (t >= nt || l >= nl) return;
    int ipos = start + l;
    int i = AXIS == 0 ? ipos : t, j = AXIS == 0 ? t : ipos;
    long c = (long)(i + g.ng) * g.NY + (j + g.ng);
    long step = AXIS == 0 ? g.NY : 1;
    long ps = AXIS == 0 ? (long)l * g.ny + t : (long)t * nl + l;
    double b = bco[ipos], cc = cco[ipos];
    if (EFIELD) {
        // x layer (cpml.py:531-548): psi_ey_x <- bz, psi_ez_x <- by; ey -= fac psi, ez += fac psi
        // y layer (cpml.py:569-586): psi_ex_y <- bz, psi_ez_y <- bx; ex += fac psi, ez -= fac psi
        const double *f1 = g.bz, *f2 = AXIS == 0 ? g.by : g.bx;
        double pa = b * psi_a[ps] + cc * (f1[c] - f1[c - step]);
        double pb = b * psi_b[ps] + cc * (f2[c] - f2[c - step]);
        psi_a[ps] = pa;
        psi_b[ps] = pb;
        if (AXIS == 0) { g.ey[c] -= fac * pa; g.ez[c] += fac * pb; }
        else { g.ex[c] += fac * pa; g.ez[c] -= fac * pb; }
    } else {
        // x layer (cpml.py:550-567): psi_by_x <- ez, psi_bz_x <- ey; by += fac psi, bz -= fac psi
        // y layer (cpml.py:588-606): psi_bx_y <- ez, psi_bz_y <- ex; bx -= fac psi, bz += fac psi
        const double *f1 = g.ez, *f2 = AXIS == 0 ? g.ey : g.ex;
        double pa = b * psi_a[ps] + cc * (f1[c + step] - f1[c]);
        double pb = b * psi_b[ps] + cc * (f2[c + step] - f2[c]);
        psi_a[ps] = pa;
        psi_b[ps] = pb;
        if (AXIS == 0) { g.by[c] += fac * pa; g.bz[c] -= fac * pb; }
        else { g.bx[c] -= fac * pa; g.bz[c] += fac * pb; }
    }
}

extern "C" int lpa_fdtd_e_cpml_2d(const lpa_grid *g, double dt, double eps0, const double *kappa_ex,
                                  const double *kappa_ey, void *stream) {
    LPA_REQUIRE(lpa_grid_ok(g, 2, 1) && eps0 > 0 && kappa_ex && kappa_ey, "lpa_fdtd_e_cpml_2d: bad args");
    GridV v = make_gridv(g, 2);
    dim3 grid((g->ny + 255) / 256, g->nx);
    hipLaunchKernelGGL(k_fdtd_e_cpml_2d, grid, dim3(256), 0, (hipStream_t)stream, v, dt * (LPA_C * LPA_C),
                       dt / eps0, kappa_ex, kappa_ey);
    LPA_CHECK_LAUNCH("lpa_fdtd_e_cpml_2d");
    return LPA_OK;
}

extern "C" int lpa_fdtd_b_cpml_2d(const lpa_grid *g, double dt, const double *kappa_bx,
                                  const double *kappa_by, void *stream) {
    LPA_REQUIRE(lpa_grid_ok(g, 2, 0) && kappa_bx && kappa_by, "lpa_fdtd_b_cpml_2d: bad args");
    GridV v = make_gridv(g, 2);
    dim3 grid((g->ny + 255) / 256, g->nx);
    hipLaunchKernelGGL(k_fdtd_b_cpml_2d, grid, dim3(256), 0, (hipStream_t)stream, v, dt, kappa_bx, kappa_by);
    LPA_CHECK_LAUNCH("lpa_fdtd_b_cpml_2d");
    return LPA_OK;
}

extern "C" int lpa_cpml_psi_2d(const lpa_grid *g, int efield, int axis, int start, int stop, double dt,
                               const double *bcoeff, const double *ccoeff_d, double *psi_a, double *psi_b,
                               void *stream) {
    LPA_REQUIRE(lpa_grid_ok(g, 2, 0) && (axis == 0 || axis == 1) && bcoeff && ccoeff_d && psi_a && psi_b,
                "lpa_cpml_psi_2d: bad args");
    int n = axis == 0 ? g->nx : g->ny, nt = axis == 0 ? g->ny : g->nx;
    // the E recursion reads node start-1 (the low guard when start == 0), the B one node stop
    LPA_REQUIRE(start >= 0 && stop > start && stop <= n, "lpa_cpml_psi_2d: layer [%d,%d) outside [0,%d)",
                start, stop, n);
    GridV v = make_gridv(g, 2);
    dim3 grid((nt + 255) / 256, stop - start);
    hipStream_t st = (hipStream_t)stream;
    double fac = efield ? dt * (LPA_C * LPA_C) : dt;
    if (efield && axis == 0) hipLaunchKernelGGL((k_cpml_psi_2d<0, true>), grid, dim3(256), 0, st, v, start, stop, fac, bcoeff, ccoeff_d, psi_a, psi_b);
    else if (efield) hipLaunchKernelGGL((k_cpml_psi_2d<1, true>), grid, dim3(256), 0, st, v, start, stop, fac, bcoeff, ccoeff_d, psi_a, psi_b);
    else if (axis == 0) hipLaunchKernelGGL((k_cpml_psi_2d<0, false>), grid, dim3(256), 0, st, v, start, stop, fac, bcoeff, ccoeff_d, psi_a, psi_b);
    else hipLaunchKernelGGL((k_cpml_psi_2d<1, false>), grid, dim3(256), 0, st, v, start, stop, fac, bcoeff, ccoeff_d, psi_a, psi_b);
    LPA_CHECK_LAUNCH("lpa_cpml_psi_2d");
    return LPA_OK;
}

// ---- fused form: kappa-scaled update + the psi recursions of every layer the cell lies in, one launch per
// field update instead of 1 + (number of layers).  Exact: a psi recursion of the E update reads only B (and
// vice versa), so doing it right after the cell's own kappa update, x layer before y (before z), performs
// the same operations in the same order as the separate launches.
struct CpmlAxisV {
    const double *kappa, *bco, *cco;
    int lo0, lo1, hi0, hi1;          // the layers' cell ranges along this axis ([x0, x1) empty = no layer)
    double *pa_lo, *pb_lo, *pa_hi, *pb_hi;
};

static CpmlAxisV make_axisv(const lpa_cpml_axis *a) {
    CpmlAxisV v;
    v.kappa = a->kappa; v.bco = a->bcoeff; v.cco = a->ccoeff_d;
    v.lo0 = a->lo0; v.lo1 = a->lo1; v.hi0 = a->hi0; v.hi1 = a->hi1;
    v.pa_lo = a->psi_a_lo; v.pb_lo = a->psi_b_lo; v.pa_hi = a->psi_a_hi; v.pb_hi = a->psi_b_hi;
    return v;
}

static int cpml_axis_ok(const lpa_cpml_axis *a, int n) {
    if (!a || !a->kappa) return 0;
    if (a->lo1 > a->lo0 && !(a->lo0 >= 0 && a->lo1 <= n && a->bcoeff && a->ccoeff_d && a->psi_a_lo && a->psi_b_lo)) return 0;
    if (a->hi1 > a->hi0 && !(a->hi0 >= 0 && a->hi1 <= n && a->bcoeff && a->ccoeff_d && a->psi_a_hi && a->psi_b_hi)) return 0;
    return 1;
}

// which layer of the axis holds position `pos` (0 none, 1 low, 2 high) and the layer-local index
__device__ __forceinline__ int cpml_layer(const CpmlAxisV &a, int pos, int &l, int &nl) {
    if (pos >= a.lo0 && pos < a.lo1) { l = pos - a.lo0; nl = a.lo1 - a.lo0; return 1; }
    if (pos >= a.hi0 && pos < a.hi1) { l = pos - a.hi0; nl = a.hi1 - a.hi0; return 2; }
    return 0;
}

__global__ void __launch_bounds__(256) k_fdtd_e_cpml_fused_2d(GridV g, double bfac, double jfac, double fac,
                                                              CpmlAxisV ax, CpmlAxisV ay, int wrap) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    int i = blockIdx.y;
    if (j >= g.ny) return;
    long c = (long)(i + g.ng) * g.NY + (j + g.ng);
    long xm = c - g.NY, ym = c - 1;
    double bfx = bfac / ax.kappa[i], bfy = bfac / ay.kappa[j];
    double bxc = g.bx[c], byc = g.by[c], bzc = g.bz[c];
    double bz_xm = g.bz[xm], bz_ym = g.bz[ym], by_xm = g.by[xm], bx_ym = g.bx[ym];
    double ex = g.ex[c], ey = g.ey[c], ez = g.ez[c];
    const double kx = bfy * ((bzc - bz_ym) / g.dy) - jfac * g.jx[c];
    const double ky = bfx * (-(bzc - bz_xm) / g.dx) - jfac * g.jy[c];
    const double kz = bfx * ((byc - by_xm) / g.dx) - bfy * ((bxc - bx_ym) / g.dy) - jfac * g.jz[c];
    int lx_, nlx, ly_, nly;
    const int wx = cpml_layer(ax, i, lx_, nlx), wy = cpml_layer(ay, j, ly_, nly);
    double *pax = nullptr, *pbx = nullptr, *pay = nullptr, *pby = nullptr;
    double pa_x = 0.0, pb_x = 0.0, pa_y = 0.0, pb_y = 0.0, bx_ = 0.0, cx_ = 0.0, by_ = 0.0, cy_ = 0.0;
    if (wx) {                                        // cpml.py:531-548
        const long ps = (long)lx_ * g.ny + j;
        pax = (wx == 1 ? ax.pa_lo : ax.pa_hi) + ps; pbx = (wx == 1 ? ax.pb_lo : ax.pb_hi) + ps;
        pa_x = *pax; pb_x = *pbx; bx_ = ax.bco[i]; cx_ = ax.cco[i];
    }
    if (wy) {                                        // cpml.py:569-586
        const long ps = (long)i * nly + ly_;
        pay = (wy == 1 ? ay.pa_lo : ay.pa_hi) + ps; pby = (wy == 1 ? ay.pb_lo : ay.pb_hi) + ps;
        pa_y = *pay; pb_y = *pby; by_ = ay.bco[j]; cy_ = ay.cco[j];
    }
    for (int rep = (wrap & FDTD_TWICE) ? 2 : 1; rep > 0; rep--) {     // (FDTD_TWICE: two half steps, see above)
        ex += kx; ey += ky; ez += kz;
        if (wx) {
            pa_x = bx_ * pa_x + cx_ * (bzc - bz_xm);
            pb_x = bx_ * pb_x + cx_ * (byc - by_xm);
            ey -= fac * pa_x; ez += fac * pb_x;
        }
        if (wy) {
            pa_y = by_ * pa_y + cy_ * (bzc - bz_ym);
            pb_y = by_ * pb_y + cy_ * (bxc - bx_ym);
            ex += fac * pa_y; ez -= fac * pb_y;
        }
    }
    if (wx) { *pax = pa_x; *pbx = pb_x; }
    if (wy) { *pay = pa_y; *pby = pb_y; }
    store3_wrapped(g, g.ex, g.ey, g.ez, c, i, j, 0, wrap, ex, ey, ez);
}

__global__ void __launch_bounds__(256) k_fdtd_b_cpml_fused_2d(GridV g, double dt, CpmlAxisV ax, CpmlAxisV ay, int wrap,
                                                              BTail tail) {
    b_tail(g, tail);
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    int i = fdtd_b_row((int)blockIdx.y, wrap, g.nx);
    if (j >= g.ny) return;
    long c = (long)(i + g.ng) * g.NY + (j + g.ng);
    long xp = c + g.NY, yp = c + 1;
    // (an x guard plane lies outside every x layer -- a face with a neighbour slab has none: kappa_x = 1 there; the y
    // layers' psi arrays have rows for it)
    double efx = dt / ax.kappa[min(max(i, 0), g.nx - 1)], efy = dt / ay.kappa[j];
    double exc = g.ex[c], eyc = g.ey[c], ezc = g.ez[c];
    double ez_xp = g.ez[xp], ez_yp = g.ez[yp], ey_xp = g.ey[xp], ex_yp = g.ex[yp];
    double bx = g.bx[c], by = g.by[c], bz = g.bz[c];
    bx -= efy * ((ez_yp - ezc) / g.dy);
    by -= efx * (-(ez_xp - ezc) / g.dx);
    bz -= efx * ((ey_xp - eyc) / g.dx) - efy * ((ex_yp - exc) / g.dy);
    int l, nl;
    if (int w = cpml_layer(ax, i, l, nl)) {          // cpml.py:550-567
        double *pa_ = w == 1 ? ax.pa_lo : ax.pa_hi, *pb_ = w == 1 ? ax.pb_lo : ax.pb_hi;
        long ps = (long)l * g.ny + j;
        double b = ax.bco[i], cc = ax.cco[i];
        double pa = b * pa_[ps] + cc * (ez_xp - ezc);
        double pb = b * pb_[ps] + cc * (ey_xp - eyc);
        pa_[ps] = pa; pb_[ps] = pb;
        by += dt * pa; bz -= dt * pb;
    }
    if (int w = cpml_layer(ay, j, l, nl)) {          // cpml.py:588-606
        double *pa_ = w == 1 ? ay.pa_lo : ay.pa_hi, *pb_ = w == 1 ? ay.pb_lo : ay.pb_hi;
        long ps = (long)i * nl + l;
        double b = ay.bco[j], cc = ay.cco[j];
        double pa = b * pa_[ps] + cc * (ez_yp - ezc);
        double pb = b * pb_[ps] + cc * (ex_yp - exc);
        pa_[ps] = pa; pb_[ps] = pb;
        bx -= dt * pa; bz += dt * pb;
    }
    store3_wrapped(g, g.bx, g.by, g.bz, c, i, j, 0, wrap, bx, by, bz);
}

static int fdtd_e_cpml_fused_2d(const lpa_grid *g, double dt, double eps0, const lpa_cpml_axis *ax,
                                const lpa_cpml_axis *ay, int wrap, void *stream) {
    LPA_REQUIRE(lpa_grid_ok(g, 2, 1) && eps0 > 0 && cpml_axis_ok(ax, g->nx) && cpml_axis_ok(ay, g->ny),
                "lpa_fdtd_e_cpml_fused_2d: bad args");
    GridV v = make_gridv(g, 2);
    dim3 grid((g->ny + 255) / 256, g->nx);
    hipLaunchKernelGGL(k_fdtd_e_cpml_fused_2d, grid, dim3(256), 0, (hipStream_t)stream, v, dt * (LPA_C * LPA_C),
                       dt / eps0, dt * (LPA_C * LPA_C), make_axisv(ax), make_axisv(ay), wrap);
    LPA_CHECK_LAUNCH("lpa_fdtd_e_cpml_fused_2d");
    return LPA_OK;
}
extern "C" int lpa_fdtd_e_cpml_fused_2d(const lpa_grid *g, double dt, double eps0, const lpa_cpml_axis *ax,
                                        const lpa_cpml_axis *ay, void *stream) {
    return fdtd_e_cpml_fused_2d(g, dt, eps0, ax, ay, 0, stream);
}

static int fdtd_b_cpml_fused_2d(const lpa_grid *g, double dt, const lpa_cpml_axis *ax, const lpa_cpml_axis *ay, int wrap,
                                void *stream, const BTail &tail = BTail{}) {
    LPA_REQUIRE(lpa_grid_ok(g, 2, 0) && cpml_axis_ok(ax, g->nx) && cpml_axis_ok(ay, g->ny),
                "lpa_fdtd_b_cpml_fused_2d: bad args");
    GridV v = make_gridv(g, 2);
    dim3 grid((g->ny + 255) / 256, fdtd_b_rows(wrap, g->nx));
    hipLaunchKernelGGL(k_fdtd_b_cpml_fused_2d, grid, dim3(256), 0, (hipStream_t)stream, v, dt, make_axisv(ax),
                       make_axisv(ay), wrap, tail);
    LPA_CHECK_LAUNCH("lpa_fdtd_b_cpml_fused_2d");
    return LPA_OK;
}
extern "C" int lpa_fdtd_b_cpml_fused_2d(const lpa_grid *g, double dt, const lpa_cpml_axis *ax,
                                        const lpa_cpml_axis *ay, void *stream) {
    return fdtd_b_cpml_fused_2d(g, dt, ax, ay, 0, stream);
}

// 3-D fused twins.  psi layouts: axis 0 [layer][ny][nz], axis 1 [nx][layer][nz], axis 2 [nx][ny][layer].
__device__ __forceinline__ long psi_index_3d(int axis, int i, int j, int k, int l, int nl, int ny, int nz) {
    if (axis == 0) return ((long)l * ny + j) * nz + k;
    if (axis == 1) return ((long)i * nl + l) * nz + k;
    return ((long)i * ny + j) * nl + l;
}

// one layer's recursion for the cell: psi_{a,b} <- b psi + cc d{a,b}; returns the two psi values
__device__ __forceinline__ void psi_step(const CpmlAxisV &a, int w, long ps, int pos, double da, double db,
                                         double &pa, double &pb) {
    double *pa_ = w == 1 ? a.pa_lo : a.pa_hi, *pb_ = w == 1 ? a.pb_lo : a.pb_hi;
    double b = a.bco[pos], cc = a.cco[pos];
    pa = b * pa_[ps] + cc * da;
    pb = b * pb_[ps] + cc * db;
    pa_[ps] = pa; pb_[ps] = pb;
}

__global__ void __launch_bounds__(256) k_fdtd_e_cpml_fused_3d(GridV g, double bfac, double jfac, double fac,
                                                              CpmlAxisV ax, CpmlAxisV ay, CpmlAxisV az, int wrap) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    int j = blockIdx.y, i = blockIdx.z;
    if (k >= g.nz) return;
    long sy = g.NZ, sx = (long)g.NY * g.NZ;
    long c = (long)(i + g.ng) * sx + (long)(j + g.ng) * sy + (k + g.ng);
    double bfx = bfac / ax.kappa[i], bfy = bfac / ay.kappa[j], bfz = bfac / az.kappa[k];
    double bxc = g.bx[c], byc = g.by[c], bzc = g.bz[c];
    double dbz_x = bzc - g.bz[c - sx], dby_x = byc - g.by[c - sx];
    double dbz_y = bzc - g.bz[c - sy], dbx_y = bxc - g.bx[c - sy];
    double dby_z = byc - g.by[c - 1], dbx_z = bxc - g.bx[c - 1];
    double ex = g.ex[c], ey = g.ey[c], ez = g.ez[c];
    const double kx = (bfy * dbz_y / g.dy - bfz * dby_z / g.dz) - jfac * g.jx[c];
    const double ky = (bfz * dbx_z / g.dz - bfx * dbz_x / g.dx) - jfac * g.jy[c];
    const double kz = (bfx * dby_x / g.dx - bfy * dbx_y / g.dy) - jfac * g.jz[c];
    int l, nl;
    double pa, pb;
    for (int rep = (wrap & FDTD_TWICE) ? 2 : 1; rep > 0; rep--) {     // (FDTD_TWICE: two half steps; psi_step re-reads what it wrote)
        ex += kx; ey += ky; ez += kz;
        if (int w = cpml_layer(ax, i, l, nl)) {   // psi(ey,ez) <- (bz,by); ey -=, ez +=   (cpml.py:609-628)
            psi_step(ax, w, psi_index_3d(0, i, j, k, l, nl, g.ny, g.nz), i, dbz_x, dby_x, pa, pb);
            ey -= fac * pa; ez += fac * pb;
        }
        if (int w = cpml_layer(ay, j, l, nl)) {   // psi(ex,ez) <- (bz,bx); ex +=, ez -=   (:651-669)
            psi_step(ay, w, psi_index_3d(1, i, j, k, l, nl, g.ny, g.nz), j, dbz_y, dbx_y, pa, pb);
            ex += fac * pa; ez -= fac * pb;
        }
        if (int w = cpml_layer(az, k, l, nl)) {   // psi(ex,ey) <- (by,bx); ex -=, ey +=   (:691-710)
            psi_step(az, w, psi_index_3d(2, i, j, k, l, nl, g.ny, g.nz), k, dby_z, dbx_z, pa, pb);
            ex -= fac * pa; ey += fac * pb;
        }
    }
    store3_wrapped(g, g.ex, g.ey, g.ez, c, i, j, k, wrap, ex, ey, ez);
}

__global__ void __launch_bounds__(256) k_fdtd_b_cpml_fused_3d(GridV g, double dt, CpmlAxisV ax, CpmlAxisV ay,
                                                              CpmlAxisV az, int wrap, BTail tail) {
    b_tail(g, tail);
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    int j = blockIdx.y, i = fdtd_b_row((int)blockIdx.z, wrap, g.nx);
    if (k >= g.nz) return;
    long sy = g.NZ, sx = (long)g.NY * g.NZ;
    long c = (long)(i + g.ng) * sx + (long)(j + g.ng) * sy + (k + g.ng);
    double efx = dt / ax.kappa[min(max(i, 0), g.nx - 1)], efy = dt / ay.kappa[j], efz = dt / az.kappa[k];
    double exc = g.ex[c], eyc = g.ey[c], ezc = g.ez[c];
    double dez_x = g.ez[c + sx] - ezc, dey_x = g.ey[c + sx] - eyc;
    double dez_y = g.ez[c + sy] - ezc, dex_y = g.ex[c + sy] - exc;
    double dey_z = g.ey[c + 1] - eyc, dex_z = g.ex[c + 1] - exc;
    double bx = g.bx[c], by = g.by[c], bz = g.bz[c];
    bx -= (efy * dez_y / g.dy - efz * dey_z / g.dz);
    by -= (efz * dex_z / g.dz - efx * dez_x / g.dx);
    bz -= (efx * dey_x / g.dx - efy * dex_y / g.dy);
    int l, nl;
    double pa, pb;
    if (int w = cpml_layer(ax, i, l, nl)) {   // psi(by,bz) <- (ez,ey); by +=, bz -=   (cpml.py:630-649)
        psi_step(ax, w, psi_index_3d(0, i, j, k, l, nl, g.ny, g.nz), i, dez_x, dey_x, pa, pb);
        by += dt * pa; bz -= dt * pb;
    }
    if (int w = cpml_layer(ay, j, l, nl)) {   // psi(bx,bz) <- (ez,ex); bx -=, bz +=   (:671-689)
        psi_step(ay, w, psi_index_3d(1, i, j, k, l, nl, g.ny, g.nz), j, dez_y, dex_y, pa, pb);
        bx -= dt * pa; bz += dt * pb;
    }
    if (int w = cpml_layer(az, k, l, nl)) {   // psi(bx,by) <- (ey,ex); bx +=, by -=   (:712-729)
        psi_step(az, w, psi_index_3d(2, i, j, k, l, nl, g.ny, g.nz), k, dey_z, dex_z, pa, pb);
        bx += dt * pa; by -= dt * pb;
    }
    store3_wrapped(g, g.bx, g.by, g.bz, c, i, j, k, wrap, bx, by, bz);
}

static int fdtd_e_cpml_fused_3d(const lpa_grid *g, double dt, double eps0, const lpa_cpml_axis *ax,
                                const lpa_cpml_axis *ay, const lpa_cpml_axis *az, int wrap, void *stream) {
    LPA_REQUIRE(lpa_grid_ok(g, 3, 1) && eps0 > 0 && cpml_axis_ok(ax, g->nx) && cpml_axis_ok(ay, g->ny) &&
                    cpml_axis_ok(az, g->nz) && g->ny <= 65535 && g->nx <= 65535,
                "lpa_fdtd_e_cpml_fused_3d: bad args");
    GridV v = make_gridv(g, 3);
    dim3 grid((g->nz + 255) / 256, g->ny, g->nx);
    hipLaunchKernelGGL(k_fdtd_e_cpml_fused_3d, grid, dim3(256), 0, (hipStream_t)stream, v, dt * (LPA_C * LPA_C),
                       dt / eps0, dt * (LPA_C * LPA_C), make_axisv(ax), make_axisv(ay), make_axisv(az), wrap);
    LPA_CHECK_LAUNCH("lpa_fdtd_e_cpml_fused_3d");
    return LPA_OK;
}
extern "C" int lpa_fdtd_e_cpml_fused_3d(const lpa_grid *g, double dt, double eps0, const lpa_cpml_axis *ax,
                                        const lpa_cpml_axis *ay, const lpa_cpml_axis *az, void *stream) {
    return fdtd_e_cpml_fused_3d(g, dt, eps0, ax, ay, az, 0, stream);
}

static int fdtd_b_cpml_fused_3d(const lpa_grid *g, double dt, const lpa_cpml_axis *ax, const lpa_cpml_axis *ay,
                                const lpa_cpml_axis *az, int wrap, void *stream, const BTail &tail = BTail{}) {
    LPA_REQUIRE(lpa_grid_ok(g, 3, 0) && cpml_axis_ok(ax, g->nx) && cpml_axis_ok(ay, g->ny) && cpml_axis_ok(az, g->nz) &&
                    g->ny <= 65535 && g->nx <= 65535,
                "lpa_fdtd_b_cpml_fused_3d: bad args");
    GridV v = make_gridv(g, 3);
    dim3 grid((g->nz + 255) / 256, g->ny, fdtd_b_rows(wrap, g->nx));
    hipLaunchKernelGGL(k_fdtd_b_cpml_fused_3d, grid, dim3(256), 0, (hipStream_t)stream, v, dt, make_axisv(ax),
                       make_axisv(ay), make_axisv(az), wrap, tail);
    LPA_CHECK_LAUNCH("lpa_fdtd_b_cpml_fused_3d");
    return LPA_OK;
}
extern "C" int lpa_fdtd_b_cpml_fused_3d(const lpa_grid *g, double dt, const lpa_cpml_axis *ax,
                                        const lpa_cpml_axis *ay, const lpa_cpml_axis *az, void *stream) {
    return fdtd_b_cpml_fused_3d(g, dt, ax, ay, az, 0, stream);
}

// one half step of E or B over the slab with the guard wrap of the axes in `wrap` fused in (lpa_step); ax[]: the fused
// CPML descriptors, all NULL = the plain Yee update
// can the B sweep of this grid carry `t` (lpa_tail.hpp)?  Fills the device form.  The reset needs jx jy jz [rho] in one
// allocation (it is one memset-like range), like lpai_reset_step's single launch
static bool make_btail(const lpa_grid *g, int dim, const lpai_tail *t, BTail *out) {
    *out = BTail{};
    if (!t || t->mode == B_TAIL_NONE || !g->jx || !g->jy || !g->jz || !g->rho) return false;
    const long cnt = (long)(g->nx + 2 * g->ng) * (g->ny + 2 * g->ng) * (dim == 3 ? (long)(g->nz + 2 * g->ng) : 1);
    if (t->mode == B_TAIL_RESET) {
        const bool contiguous = g->jy == g->jx + cnt && g->jz == g->jy + cnt && (!t->with_rho || g->rho == g->jz + cnt);
        if (!contiguous || t->nwords < 0 || t->nwords > 32) return false;
        out->mode = B_TAIL_RESET;
        out->a = g->jx; out->n = (t->with_rho ? 4 : 3) * cnt;
        out->b = t->also; out->nb = t->also ? cnt : 0;
        out->nw = t->nwords;
        for (int i = 0; i < t->nwords; i++) out->w.w[i] = t->words[i];
        return true;
    }
    if (t->mode == B_TAIL_RHO) {
        if (!(t->dt > 0) || (t->split_x && (t->periodic_axes & 1)) || t->split_x < 0 || t->split_x > 3 || ((t->split_x & 1) && !t->left))
            return false;
        out->mode = B_TAIL_RHO;
        out->dtdx = t->dt / g->dx; out->dtdy = t->dt / g->dy; out->dtdz = dim == 3 ? t->dt / g->dz : 0.0;
        out->mx = t->split_x ? ((t->split_x & 1 ? RHO_NB_LO : 0) | (t->split_x & 2 ? RHO_NB_HI : 0)) : (t->periodic_axes & 1);
        out->my = (t->periodic_axes >> 1) & 1; out->mz = (t->periodic_axes >> 2) & 1;
        out->left = t->left;
        return true;
    }
    return false;
}

int lpai_tail_rides(const lpa_grid *g, int dim, const lpai_tail *t) {
    BTail b;
    return g && make_btail(g, dim, t, &b) ? 1 : 0;
}

int lpai_fdtd(const lpa_grid *g, int dim, int efield, double dt, double eps0, const lpa_cpml_axis *const *ax, int wrap,
              int twice, int ext_lo, int ext_hi, int b_part, const lpai_tail *tail, void *stream) {
    LPA_REQUIRE(g && g->nx >= g->ng && g->ny >= g->ng && (dim == 2 || g->nz >= g->ng), "lpai_fdtd: slab thinner than the guard");
    LPA_REQUIRE(!twice || efield, "lpai_fdtd: only the E sweep does two half steps at once");
    LPA_REQUIRE(ext_lo >= 0 && ext_hi >= 0 && ext_lo <= g->ng && ext_hi < g->ng && ext_lo <= 3 && ext_hi <= 3 &&
                    (!(ext_lo | ext_hi) || (!efield && !(wrap & 1))),
                "lpai_fdtd: only the B sweep of a slab split along x advances x guard planes (ng low, ng - 1 high at most)");
    LPA_REQUIRE(b_part >= 0 && b_part <= 2 && (!b_part || (!efield && !(wrap & 1) && g->nx >= 2)),
                "lpai_fdtd: only the B sweep of a slab split along x runs in two parts");
    if (b_part == 1) ext_lo = ext_hi = 0;       // (the guard planes belong to the edge part)
    wrap = (wrap & 7) | (twice ? FDTD_TWICE : 0) | (ext_lo << FDTD_EXT_LO_SHIFT) | (ext_hi << FDTD_EXT_HI_SHIFT) |
           (b_part == 1 ? FDTD_B_INTERIOR : (b_part == 2 ? FDTD_B_EDGE : 0));
    const bool cpml = ax && ax[0];
    BTail bt{};
    LPA_REQUIRE(!tail || (!efield && b_part != 2 && make_btail(g, dim, tail, &bt)),
                "lpai_fdtd: this sweep cannot carry the tail (ask lpai_tail_rides first)");
    if (dim == 2) {
        if (efield) return cpml ? fdtd_e_cpml_fused_2d(g, dt, eps0, ax[0], ax[1], wrap, stream) : fdtd_e_2d(g, dt, eps0, wrap, stream);
        return cpml ? fdtd_b_cpml_fused_2d(g, dt, ax[0], ax[1], wrap, stream, bt) : fdtd_b_2d(g, dt, wrap, stream, bt);
    }
    if (efield)
        return cpml ? fdtd_e_cpml_fused_3d(g, dt, eps0, ax[0], ax[1], ax[2], wrap, stream) : fdtd_e_3d(g, dt, eps0, wrap, stream);
    return cpml ? fdtd_b_cpml_fused_3d(g, dt, ax[0], ax[1], ax[2], wrap, stream, bt) : fdtd_b_3d(g, dt, wrap, stream, bt);
}

// ---- 3-D CPML (cpml.py:431-475 kappa-scaled update, :609-729 psi recursions) ---------------------------
__global__ void __launch_bounds__(256) k_fdtd_e_cpml_3d(GridV g, double bfac, double jfac,
                                                        const double *__restrict__ kx,
                                                        const double *__restrict__ ky,
                                                        const double *__restrict__ kz) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    int j = blockIdx.y, i = blockIdx.z;
    if (k >= g.nz) return;
    long sy = g.NZ, sx = (long)g.NY * g.NZ;
    long c = (long)(i + g.ng) * sx + (long)(j + g.ng) * sy + (k + g.ng);
    long xm = c - sx, ym = c - sy, zm = c - 1;
    double bfx = bfac / kx[i], bfy = bfac / ky[j], bfz = bfac / kz[k];
    double bxc = g.bx[c], byc = g.by[c], bzc = g.bz[c];
    g.ex[c] += (bfy * (bzc - g.bz[ym]) / g.dy - bfz * (byc - g.by[zm]) / g.dz) - jfac * g.jx[c];
    g.ey[c] += (bfz * (bxc - g.bx[zm]) / g.dz - bfx * (bzc - g.bz[xm]) / g.dx) - jfac * g.jy[c];
    g.ez[c] += (bfx * (byc - g.by[xm]) / g.dx - bfy * (bxc - g.bx[ym]) / g.dy) - jfac * g.jz[c];
}

__global__ void __launch_bounds__(256) k_fdtd_b_cpml_3d(GridV g, double dt, const double *__restrict__ kx,
                                                        const double *__restrict__ ky,
                                                        const double *__restrict__ kz) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    int j = blockIdx.y, i = blockIdx.z;
    if (k >= g.nz) return;
    long sy = g.NZ, sx = (long)g.NY * g.NZ;
    long c = (long)(i + g.ng) * sx + (long)(j + g.ng) * sy + (k + g.ng);
    long xp = c + sx, yp = c + sy, zp = c + 1;
    double efx = dt / kx[i], efy = dt / ky[j], efz = dt / kz[k];
    double exc = g.ex[c], eyc = g.ey[c], ezc = g.ez[c];
    g.bx[c] -= (efy * (g.ez[yp] - ezc) / g.dy - efz * (g.ey[zp] - eyc) / g.dz);
    g.by[c] -= (efz * (g.ex[zp] - exc) / g.dz - efx * (g.ez[xp] - ezc) / g.dx);
    g.bz[c] -= (efx * (g.ey[xp] - eyc) / g.dx - efy * (g.ex[yp] - exc) / g.dy);
}

// psi recursion + field correction of one layer, normal to AXIS; psi arrays are compact:
// AXIS 0: [layer][ny][nz], AXIS 1: [nx][layer][nz], AXIS 2: [nx][ny][layer].
//   E: x: psi(ey,ez) <- (bz,by), ey -= , ez +=   y: psi(ex,ez) <- (bz,bx), ex +=, ez -=
//      z: psi(ex,ey) <- (by,bx), ex -= , ey +=                       (cpml.py:609-628,651-669,691-710)
//   B: x: psi(by,bz) <- (ez,ey), by += , bz -=   y: psi(bx,bz) <- (ez,ex), bx -=, bz +=
//      z: psi(bx,by) <- (ey,ex), bx += , by -=                       (cpml.py:630-649,671-689,712-729)
template <int AXIS, bool EFIELD>
__global__ void __launch_bounds__(256) k_cpml_psi_3d(GridV g, int start, int stop, double fac,
                                                     const double *__restrict__ bco,
                                                     const double *__restrict__ cco, double *psi_a,
                                                     double *psi_b) {
    const int nl = stop - start;
    // thread grid: x = fastest transverse (or layer, for AXIS 2) index
    int n0 = AXIS == 0 ? nl : g.nx, n1 = AXIS == 1 ? nl : g.ny, n2 = AXIS == 2 ? nl : g.nz;
    int t2 = blockIdx.x * blockDim.x + threadIdx.x, t1 = blockIdx.y, t0 = blockIdx.z;
    if (t2 >= n2 || t1 >= n1 || t0 >= n0) return;
    int i = AXIS == 0 ? start + t0 : t0, j = AXIS == 1 ? start + t1 : t1, k = AXIS == 2 ? start + t2 : t2;
    int ipos = AXIS == 0 ? i : (AXIS == 1 ? j : k);
    long sy = g.NZ, sx = (long)g.NY * g.NZ;
    long c = (long)(i + g.ng) * sx + (long)(j + g.ng) * sy + (k + g.ng);
    long step = AXIS == 0 ? sx : (AXIS == 1 ? sy : 1);
    long ps = ((long)t0 * n1 + t1) * n2 + t2;
    double b = bco[ipos], cc = cco[ipos];
    const double *f1, *f2;
    double *ta, *tb;
    double sa, sb;
    if (EFIELD) {
        if (AXIS == 0) { f1 = g.bz; f2 = g.by; ta = g.ey; tb = g.ez; sa = -1; sb = 1; }
        else if (AXIS == 1) { f1 = g.bz; f2 = g.bx; ta = g.ex; tb = g.ez; sa = 1; sb = -1; }
        else { f1 = g.by; f2 = g.bx; ta = g.ex; tb = g.ey; sa = -1; sb = 1; }
        double pa = b * psi_a[ps] + cc * (f1[c] - f1[c - step]);
        double pb = b * psi_b[ps] + cc * (f2[c] - f2[c - step]);
        psi_a[ps] = pa; psi_b[ps] = pb;
        ta[c] += sa * (fac * pa);
        tb[c] += sb * (fac * pb);
    } else {
        if (AXIS == 0) { f1 = g.ez; f2 = g.ey; ta = g.by; tb = g.bz; sa = 1; sb = -1; }
        else if (AXIS == 1) { f1 = g.ez; f2 = g.ex; ta = g.bx; tb = g.bz; sa = -1; sb = 1; }
        else { f1 = g.ey; f2 = g.ex; ta = g.bx; tb = g.by; sa = 1; sb = -1; }
        double pa = b * psi_a[ps] + cc * (f1[c + step] - f1[c]);
        double pb = b * psi_b[ps] + cc * (f2[c + step] - f2[c]);
        psi_a[ps] = pa; psi_b[ps] = pb;
        ta[c] += sa * (fac * pa);
        tb[c] += sb * (fac * pb);
    }
}

extern "C" int lpa_fdtd_e_cpml_3d(const lpa_grid *g, double dt, double eps0, const double *kappa_ex,
                                  const double *kappa_ey, const double *kappa_ez, void *stream) {
    LPA_REQUIRE(lpa_grid_ok(g, 3, 1) && kappa_ex && kappa_ey && kappa_ez && eps0 > 0,
                "lpa_fdtd_e_cpml_3d: bad args");
    LPA_REQUIRE(g->ny <= 65535 && g->nx <= 65535, "lpa_fdtd_e_cpml_3d: nx, ny must be <= 65535");
    GridV v = make_gridv(g, 3);
    dim3 grid((g->nz + 255) / 256, g->ny, g->nx);
    hipLaunchKernelGGL(k_fdtd_e_cpml_3d, grid, dim3(256), 0, (hipStream_t)stream, v, dt * (LPA_C * LPA_C),
                       dt / eps0, kappa_ex, kappa_ey, kappa_ez);
    LPA_CHECK_LAUNCH("lpa_fdtd_e_cpml_3d");
    return LPA_OK;
}

extern "C" int lpa_fdtd_b_cpml_3d(const lpa_grid *g, double dt, const double *kappa_bx, const double *kappa_by,
                                  const double *kappa_bz, void *stream) {
    LPA_REQUIRE(lpa_grid_ok(g, 3, 0) && kappa_bx && kappa_by && kappa_bz, "lpa_fdtd_b_cpml_3d: bad args");
    LPA_REQUIRE(g->ny <= 65535 && g->nx <= 65535, "lpa_fdtd_b_cpml_3d: nx, ny must be <= 65535");
    GridV v = make_gridv(g, 3);
    dim3 grid((g->nz + 255) / 256, g->ny, g->nx);
    hipLaunchKernelGGL(k_fdtd_b_cpml_3d, grid, dim3(256), 0, (hipStream_t)stream, v, dt, kappa_bx, kappa_by,
                       kappa_bz);
    LPA_CHECK_LAUNCH("lpa_fdtd_b_cpml_3d");
    return LPA_OK;
}

extern "C" int lpa_cpml_psi_3d(const lpa_grid *g, int efield, int axis, int start, int stop, double dt,
                               const double *bcoeff, const double *ccoeff_d, double *psi_a, double *psi_b,
                               void *stream) {
    LPA_REQUIRE(lpa_grid_ok(g, 3, 0) && axis >= 0 && axis <= 2 && bcoeff && ccoeff_d && psi_a && psi_b,
                "lpa_cpml_psi_3d: bad args");
    int n = axis == 0 ? g->nx : (axis == 1 ? g->ny : g->nz);
    LPA_REQUIRE(start >= 0 && stop > start && stop <= n, "lpa_cpml_psi_3d: layer [%d,%d) outside [0,%d)",
                start, stop, n);
    int nl = stop - start;
    int n0 = axis == 0 ? nl : g->nx, n1 = axis == 1 ? nl : g->ny, n2 = axis == 2 ? nl : g->nz;
    LPA_REQUIRE(n0 <= 65535 && n1 <= 65535, "lpa_cpml_psi_3d: nx, ny must be <= 65535");
    GridV v = make_gridv(g, 3);
    dim3 grid((n2 + 255) / 256, n1, n0), blk(256);
    hipStream_t st = (hipStream_t)stream;
    double fac = efield ? dt * (LPA_C * LPA_C) : dt;
#define LPA_PSI3(A, E) hipLaunchKernelGGL((k_cpml_psi_3d<A, E>), grid, blk, 0, st, v, start, stop, fac, bcoeff, ccoeff_d, psi_a, psi_b)
    if (efield) { if (axis == 0) LPA_PSI3(0, true); else if (axis == 1) LPA_PSI3(1, true); else LPA_PSI3(2, true); }
    else { if (axis == 0) LPA_PSI3(0, false); else if (axis == 1) LPA_PSI3(1, false); else LPA_PSI3(2, false); }
#undef LPA_PSI3
    LPA_CHECK_LAUNCH("lpa_cpml_psi_3d");
    return LPA_OK;
}

// =====================================================================================================
// laser injection at the x-min boundary (callback/laser.py:17-46): Mur-type condition on bz, by (and a
// copy of bx) on the node row laserpos-1, driven by the source fields ey_source, ez_source [ny].
// All right-hand sides are read from rows 0, -1 (low guard) and laserpos, never from row laserpos-1.
// =====================================================================================================
// SEP: the sources arrive factorised, ey = k[0] pc + k[1] ps, ez = k[2] pc + k[3] ps (pc, ps: time-independent arrays of
// the profile, k: four numbers per step; see lpa_laser_inject_sep_2d) and are evaluated here -- the profile then costs no
// launch of its own (as torch expressions: ~17 launches of 4 us, 0.14 ms per step, a third of a C3 step)
struct LaserK { double k[4]; };
template <bool SEP>
__global__ void __launch_bounds__(256) k_laser_inject_2d(GridV g, int lp, double dt, double eps0,
                                                         int iy0, int iy1,
                                                         const double *__restrict__ eys,
                                                         const double *__restrict__ ezs, LaserK lk) {
    int j = iy0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= iy1) return;
    const double c = LPA_C;
    const double ey_s = SEP ? fma(lk.k[1], ezs[j], lk.k[0] * eys[j]) : eys[j];
    const double ez_s = SEP ? fma(lk.k[3], ezs[j], lk.k[2] * eys[j]) : ezs[j];
    long r0 = (long)g.ng * g.NY + (j + g.ng);          // row 0
    long rm = r0 - g.NY;                                // row -1 (low guard)
    long rl = (long)(lp + g.ng) * g.NY + (j + g.ng);    // row laserpos
    long rt = rl - g.NY;                                // row laserpos - 1
    double k = 1 / ((c * dt / g.dx + 1) * c);
    double bxv = g.bx[r0];
    double bzv = k * (+4 * ey_s + 2 * (g.ey[r0] + c * 0.5 * (g.bz[r0] + g.bz[rm])) - 2 * g.ey[rl] +
                      dt / eps0 * g.jy[rl] + (c * dt / g.dx - 1) * c * g.bz[rl]);
    double byv = k * (-4 * ez_s - 2 * (g.ez[r0] - c * 0.5 * (g.by[r0] + g.by[rm])) + 2 * g.ez[rl] -
                      (dt * (c * c)) * (g.bx[rl] - g.bx[rl - 1]) / g.dy - dt / eps0 * g.jz[rl] +
                      (c * dt / g.dx - 1) * c * g.by[rl]);
    g.bx[rt] = bxv;
    g.bz[rt] = bzv;
    g.by[rt] = byv;
}

static int laser_inject_2d(const char *name, const lpa_grid *g, int laserpos, double dt, double eps0, int iy_start,
                           int iy_end, const double *a, const double *b, const double *k4, void *stream) {
    LPA_REQUIRE(lpa_grid_ok(g, 2, 1) && a && b && eps0 > 0, "%s: bad args", name);
    LPA_REQUIRE(laserpos >= 2 && laserpos < g->nx && iy_start >= 0 && iy_end <= g->ny,
                "%s: laserpos / iy range outside the slab", name);
    if (iy_end <= iy_start) return LPA_OK;
    GridV v = make_gridv(g, 2);
    LaserK lk{{0, 0, 0, 0}};
    const dim3 grid((iy_end - iy_start + 255) / 256);
    if (k4) {
        for (int c = 0; c < 4; c++) lk.k[c] = k4[c];
        hipLaunchKernelGGL(k_laser_inject_2d<true>, grid, dim3(256), 0, (hipStream_t)stream, v, laserpos, dt, eps0,
                           iy_start, iy_end, a, b, lk);
    } else {
        hipLaunchKernelGGL(k_laser_inject_2d<false>, grid, dim3(256), 0, (hipStream_t)stream, v, laserpos, dt, eps0,
                           iy_start, iy_end, a, b, lk);
    }
    LPA_CHECK_LAUNCH(name);
    return LPA_OK;
}

extern "C" int lpa_laser_inject_2d(const lpa_grid *g, int laserpos, double dt, double eps0, int iy_start,
                                   int iy_end, const double *ey_source, const double *ez_source,
                                   void *stream) {
    return laser_inject_2d("lpa_laser_inject_2d", g, laserpos, dt, eps0, iy_start, iy_end, ey_source, ez_source, nullptr,
                           stream);
}

extern "C" int lpa_laser_inject_sep_2d(const lpa_grid *g, int laserpos, double dt, double eps0, int iy_start,
                                       int iy_end, const double *pc, const double *ps, const double *k4,
                                       void *stream) {
    LPA_REQUIRE(k4, "lpa_laser_inject_sep_2d: bad args");
    return laser_inject_2d("lpa_laser_inject_sep_2d", g, laserpos, dt, eps0, iy_start, iy_end, pc, ps, k4, stream);
}

// 3-D twin (callback/laser.py:63-92): the bx copy runs over the whole z row (guards included), bz gets
// the extra d(bx)/dz term; sources are [ny][nz] over the interior nodes
template <bool SEP>
__global__ void __launch_bounds__(256) k_laser_inject_3d(GridV g, int lp, double dt, double eps0, int iy0,
                                                         int iy1, int iz0, int iz1,
                                                         const double *__restrict__ eys,
                                                         const double *__restrict__ ezs, LaserK lk) {
    int kz = blockIdx.x * blockDim.x + threadIdx.x;   // padded z index
    int j = iy0 + blockIdx.y;
    if (kz >= g.NZ || j >= iy1) return;
    const double c = LPA_C;
    long sy = g.NZ, sx = (long)g.NY * g.NZ;
    long r0 = (long)g.ng * sx + (long)(j + g.ng) * sy + kz;   // row 0
    long rm = r0 - sx;                                         // row -1 (low guard)
    long rl = (long)(lp + g.ng) * sx + (long)(j + g.ng) * sy + kz;
    long rt = rl - sx;
    g.bx[rt] = g.bx[r0];
    int k = kz - g.ng;
    if (k < iz0 || k >= iz1) return;
    long si = (long)j * g.nz + k;
    const double ey_s = SEP ? fma(lk.k[1], ezs[si], lk.k[0] * eys[si]) : eys[si];
    const double ez_s = SEP ? fma(lk.k[3], ezs[si], lk.k[2] * eys[si]) : ezs[si];
    double f = 1 / ((c * dt / g.dx + 1) * c);
    double bzv = f * (+4 * ey_s + 2 * (g.ey[r0] + c * 0.5 * (g.bz[r0] + g.bz[rm])) - 2 * g.ey[rl] -
                      (dt * (c * c)) * (g.bx[rl] - g.bx[rl - 1]) / g.dz + dt / eps0 * g.jy[rl] +
                      (c * dt / g.dx - 1) * c * g.bz[rl]);
    double byv = f * (-4 * ez_s - 2 * (g.ez[r0] - c * 0.5 * (g.by[r0] + g.by[rm])) + 2 * g.ez[rl] -
                      (dt * (c * c)) * (g.bx[rl] - g.bx[rl - sy]) / g.dy - dt / eps0 * g.jz[rl] +
                      (c * dt / g.dx - 1) * c * g.by[rl]);
    g.bz[rt] = bzv;
    g.by[rt] = byv;
}

static int laser_inject_3d(const char *name, const lpa_grid *g, int laserpos, double dt, double eps0, int iy_start,
                           int iy_end, int iz_start, int iz_end, const double *a, const double *b, const double *k4,
                           void *stream) {
    LPA_REQUIRE(lpa_grid_ok(g, 3, 1) && a && b && eps0 > 0, "%s: bad args", name);
    LPA_REQUIRE(laserpos >= 2 && laserpos < g->nx && iy_start >= 0 && iy_end <= g->ny && iz_start >= 0 &&
                    iz_end <= g->nz,
                "%s: laserpos / iy / iz range outside the slab", name);
    if (iy_end <= iy_start) return LPA_OK;
    GridV v = make_gridv(g, 3);
    dim3 grid((v.NZ + 255) / 256, iy_end - iy_start);
    LaserK lk{{0, 0, 0, 0}};
    if (k4) {
        for (int c = 0; c < 4; c++) lk.k[c] = k4[c];
        hipLaunchKernelGGL(k_laser_inject_3d<true>, grid, dim3(256), 0, (hipStream_t)stream, v, laserpos, dt, eps0,
                           iy_start, iy_end, iz_start, iz_end, a, b, lk);
    } else {
        hipLaunchKernelGGL(k_laser_inject_3d<false>, grid, dim3(256), 0, (hipStream_t)stream, v, laserpos, dt, eps0,
                           iy_start, iy_end, iz_start, iz_end, a, b, lk);
    }
    LPA_CHECK_LAUNCH(name);
    return LPA_OK;
}

extern "C" int lpa_laser_inject_3d(const lpa_grid *g, int laserpos, double dt, double eps0, int iy_start,
                                   int iy_end, int iz_start, int iz_end, const double *ey_source,
                                   const double *ez_source, void *stream) {
    return laser_inject_3d("lpa_laser_inject_3d", g, laserpos, dt, eps0, iy_start, iy_end, iz_start, iz_end, ey_source,
                           ez_source, nullptr, stream);
}

extern "C" int lpa_laser_inject_sep_3d(const lpa_grid *g, int laserpos, double dt, double eps0, int iy_start,
                                       int iy_end, int iz_start, int iz_end, const double *pc, const double *ps,
                                       const double *k4, void *stream) {
    LPA_REQUIRE(k4, "lpa_laser_inject_sep_3d: bad args");
    return laser_inject_3d("lpa_laser_inject_sep_3d", g, laserpos, dt, eps0, iy_start, iy_end, iz_start, iz_end, pc, ps,
                           k4, stream);
}

// =====================================================================================================
// current reset (core/current/cpu2d.c:19-72): memset of the four arrays including guards
// =====================================================================================================
extern "C" int lpa_reset_current(const lpa_grid *g, void *stream) {
    LPA_REQUIRE(g && g->jx && g->jy && g->jz && g->rho, "lpa_reset_current: bad grid");
    size_t n = (size_t)(g->nx + 2 * g->ng) * (g->ny + 2 * g->ng) *
               (g->nz > 1 ? (size_t)(g->nz + 2 * g->ng) : 1) * sizeof(double);
    double *a[4] = {g->jx, g->jy, g->jz, g->rho};
    const size_t cnt = n / sizeof(double);
    if (a[1] == a[0] + cnt && a[2] == a[1] + cnt && a[3] == a[2] + cnt) {  // one allocation: one memset
        if (hipMemsetAsync(a[0], 0, 4 * n, (hipStream_t)stream) != hipSuccess) {
            lpa_set_error("lpa_reset_current: hipMemsetAsync failed");
            return LPA_ERR_HIP;
        }
        return LPA_OK;
    }
    for (int c = 0; c < 4; c++)
        if (hipMemsetAsync(a[c], 0, n, (hipStream_t)stream) != hipSuccess) {
            lpa_set_error("lpa_reset_current: hipMemsetAsync failed");
            return LPA_ERR_HIP;
        }
    return LPA_OK;
}

// =====================================================================================================
// periodic guard wrap (sync_guard_fields_2d with the patch as its own neighbour,
// core/patch/sync_fields2d.c:150-255): every guard cell takes the value of the interior cell it is
// the periodic image of along the axes in `axes`; one thread per padded cell, interior cells exit.
// =====================================================================================================
struct Ptr6 { double *p[6]; int n; };

__device__ __forceinline__ int image_of(int c, int n, int ng, bool periodic, bool &guard) {
    // padded index c -> padded index of the interior cell it mirrors (or itself)
    int i = c - ng;
    if (i < 0) { guard = true; return periodic ? c + n : -1; }
    if (i >= n) { guard = true; return periodic ? c - n : -1; }
    return c;
}

__global__ void __launch_bounds__(256) k_guard_wrap(GridV g, Ptr6 f, int axes) {
    int z = blockIdx.x * blockDim.x + threadIdx.x;  // fastest axis
    int NF = g.NZ > 1 ? g.NZ : g.NY;
    if (z >= NF) return;
    int cx = g.NZ > 1 ? blockIdx.z : blockIdx.y;
    int cy = g.NZ > 1 ? blockIdx.y : z;
    int cz = g.NZ > 1 ? z : 0;
    bool guard = false;
    // (axes & 8: x is split over slabs and the x guard planes are advanced in place -- they wrap their own y / z guards)
    int sx = (axes & 8) ? cx : image_of(cx, g.nx, g.ng, axes & 1, guard);
    int sy = image_of(cy, g.ny, g.ng, axes & 2, guard);
    int sz = g.NZ > 1 ? image_of(cz, g.nz, g.ng, axes & 4, guard) : 0;
    if (!guard || sx < 0 || sy < 0 || sz < 0) return;
    long dst = ((long)cx * g.NY + cy) * g.NZ + cz;
    long src = ((long)sx * g.NY + sy) * g.NZ + sz;
    for (int c = 0; c < f.n; c++) f.p[c][dst] = f.p[c][src];
}

extern "C" int lpa_guard_wrap(const lpa_grid *g, int which, int axes, void *stream) {
    LPA_REQUIRE(lpa_grid_ok(g, g && g->nz > 1 ? 3 : 2, 0), "lpa_guard_wrap: bad grid");
    LPA_REQUIRE(g->nx >= g->ng && g->ny >= g->ng, "lpa_guard_wrap: slab thinner than the guard");
    int dim = g->nz > 1 ? 3 : 2;
    GridV v = make_gridv(g, dim);
    Ptr6 f;
    f.n = 0;
    if (which & 1) { f.p[f.n++] = g->ex; f.p[f.n++] = g->ey; f.p[f.n++] = g->ez; }
    if (which & 2) { f.p[f.n++] = g->bx; f.p[f.n++] = g->by; f.p[f.n++] = g->bz; }
    if (f.n == 0 || axes == 0) return LPA_OK;
    dim3 grid = dim == 3 ? dim3((v.NZ + 255) / 256, v.NY, v.NX) : dim3((v.NY + 255) / 256, v.NX);
    hipLaunchKernelGGL(k_guard_wrap, grid, dim3(256), 0, (hipStream_t)stream, v, f, axes);
    LPA_CHECK_LAUNCH("lpa_guard_wrap");
    return LPA_OK;
}

// =====================================================================================================
// current fold (sync_currents_2d with a self neighbour, core/patch/sync_fields2d.c:43-148): an
// interior cell within ng of a periodic face accumulates its images in the guards (x face first,
// then y, then the corner -- the reference's order), then the guards that were consumed are zeroed.
// Two launches: the fold only reads guards and writes interior cells, the second pass zeroes.
// =====================================================================================================
__global__ void __launch_bounds__(256) k_current_fold(GridV g, int axes) {
    int z = blockIdx.x * blockDim.x + threadIdx.x;
    bool d3 = g.NZ > 1;
    int nf = d3 ? g.nz : g.ny;
    if (z >= nf) return;
    int i = d3 ? blockIdx.z : blockIdx.y, j = d3 ? blockIdx.y : z, k = d3 ? z : 0;
    int ng = g.ng;
    // image offsets along each axis: 0 = none, else +-n
    int ox = (axes & 1) ? (i < ng ? g.nx : (i >= g.nx - ng ? -g.nx : 0)) : 0;
    int oy = (axes & 2) ? (j < ng ? g.ny : (j >= g.ny - ng ? -g.ny : 0)) : 0;
    int oz = (d3 && (axes & 4)) ? (k < ng ? g.nz : (k >= g.nz - ng ? -g.nz : 0)) : 0;
    // thin slabs (n < 2 ng) would need both images of an axis; rejected on the host
    if (!(ox | oy | oz)) return;
    long sY = g.NZ, sX = (long)g.NY * g.NZ;
    long c = (long)(i + ng) * sX + (long)(j + ng) * sY + (d3 ? k + ng : 0);
    double *arr[4] = {g.jx, g.jy, g.jz, g.rho};
    for (int a = 0; a < 4; a++) {
        double *f = arr[a];
        double v = f[c];
        // faces, edges, vertex in the reference's order (x, y, (z), then mixed)
        if (ox) v += f[c + ox * sX];
        if (oy) v += f[c + oy * sY];
        if (oz) v += f[c + oz];
        if (ox && oy) v += f[c + ox * sX + oy * sY];
        if (ox && oz) v += f[c + ox * sX + oz];
        if (oy && oz) v += f[c + oy * sY + oz];
        if (ox && oy && oz) v += f[c + ox * sX + oy * sY + oz];
        f[c] = v;
    }
}

__global__ void __launch_bounds__(256) k_current_zero_guard(GridV g, int axes) {
    int z = blockIdx.x * blockDim.x + threadIdx.x;
    bool d3 = g.NZ > 1;
    int NF = d3 ? g.NZ : g.NY;
    if (z >= NF) return;
    int cx = d3 ? blockIdx.z : blockIdx.y, cy = d3 ? blockIdx.y : z, cz = d3 ? z : 0;
    bool gx = cx < g.ng || cx >= g.nx + g.ng, gy = cy < g.ng || cy >= g.ny + g.ng;
    bool gz = d3 && (cz < g.ng || cz >= g.nz + g.ng);
    // a guard cell was consumed iff every axis on which it is a guard is periodic here
    if (!(gx || gy || gz)) return;
    if ((gx && !(axes & 1)) || (gy && !(axes & 2)) || (gz && !(axes & 4))) return;
    long c = ((long)cx * g.NY + cy) * g.NZ + cz;
    g.jx[c] = 0.0; g.jy[c] = 0.0; g.jz[c] = 0.0; g.rho[c] = 0.0;
}

extern "C" int lpa_current_fold(const lpa_grid *g, int axes, void *stream) {
    LPA_REQUIRE(g && g->jx && g->jy && g->jz && g->rho && g->nx > 0 && g->ny > 0 && g->ng > 0,
                "lpa_current_fold: bad grid");
    int dim = g->nz > 1 ? 3 : 2;
    LPA_REQUIRE(g->nx >= 2 * g->ng && g->ny >= 2 * g->ng && (dim == 2 || g->nz >= 2 * g->ng),
                "lpa_current_fold: slab thinner than 2*ng");
    if (axes == 0) return LPA_OK;
    GridV v;
    memset(&v, 0, sizeof v);
    v.nx = g->nx; v.ny = g->ny; v.nz = dim == 3 ? g->nz : 1; v.ng = g->ng;
    v.NX = g->nx + 2 * g->ng; v.NY = g->ny + 2 * g->ng; v.NZ = dim == 3 ? g->nz + 2 * g->ng : 1;
    v.jx = g->jx; v.jy = g->jy; v.jz = g->jz; v.rho = g->rho;
    dim3 gi = dim == 3 ? dim3((v.nz + 255) / 256, v.ny, v.nx) : dim3((v.ny + 255) / 256, v.nx);
    hipLaunchKernelGGL(k_current_fold, gi, dim3(256), 0, (hipStream_t)stream, v, axes);
    LPA_CHECK_LAUNCH("lpa_current_fold");
    dim3 gp = dim == 3 ? dim3((v.NZ + 255) / 256, v.NY, v.NX) : dim3((v.NY + 255) / 256, v.NX);
    hipLaunchKernelGGL(k_current_zero_guard, gp, dim3(256), 0, (hipStream_t)stream, v, axes);
    LPA_CHECK_LAUNCH("lpa_current_zero_guard");
    return LPA_OK;
}

// =====================================================================================================
// x-face halo buffers of the slab decomposition.  Planes along x are contiguous (x is the slowest
// index), so a face is ng * NY(*NZ) consecutive doubles per component: these kernels are plain
// copies; buffers are [ncomp][ng * plane].
// =====================================================================================================
enum { HALO_PACK_SRC = 0, HALO_UNPACK_GUARD = 1, HALO_PACK_CUR = 2, HALO_UNPACK_CUR = 3 };

__global__ void __launch_bounds__(256) k_halo(Ptr6 f, double *buf, long plane, long first_row, int ng,
                                              int mode) {
    long n = (long)ng * plane;
    long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    int c = blockIdx.y;
    double *a = f.p[c] + first_row * plane + t;
    double *b = buf + (long)c * n + t;
    if (mode == HALO_PACK_SRC) *b = *a;
    else if (mode == HALO_UNPACK_GUARD) *a = *b;
    else if (mode == HALO_PACK_CUR) { *b = *a; *a = 0.0; }
    else *a += *b;
}

static int halo_launch(const lpa_grid *g, Ptr6 f, long first_row, double *buf, int mode, void *stream,
                       const char *name) {
    long plane = (long)(g->ny + 2 * g->ng) * (g->nz > 1 ? g->nz + 2 * g->ng : 1);
    long n = (long)g->ng * plane;
    dim3 grid((unsigned)((n + 255) / 256), f.n);
    hipLaunchKernelGGL(k_halo, grid, dim3(256), 0, (hipStream_t)stream, f, buf, plane, first_row,
                       g->ng, mode);
    LPA_CHECK_LAUNCH(name);
    return LPA_OK;
}

// both faces in one launch (blockIdx.z = face); a null buffer skips its face
__global__ void __launch_bounds__(256) k_halo2(Ptr6 f, double *buf_lo, double *buf_hi, long plane, long first_lo,
                                               long first_hi, int ng, int mode) {
    long n = (long)ng * plane;
    long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    double *buf = blockIdx.z == 0 ? buf_lo : buf_hi;
    if (!buf) return;
    int c = blockIdx.y;
    double *a = f.p[c] + (blockIdx.z == 0 ? first_lo : first_hi) * plane + t;
    double *b = buf + (long)c * n + t;
    if (mode == HALO_PACK_SRC) *b = *a;
    else if (mode == HALO_UNPACK_GUARD) *a = *b;
    else if (mode == HALO_PACK_CUR) { *b = *a; *a = 0.0; }
    else *a += *b;
}

static Ptr6 eb_ptrs(const lpa_grid *g, int which) {
    Ptr6 f;
    f.n = 0;
    if (which & 1) { f.p[f.n++] = g->ex; f.p[f.n++] = g->ey; f.p[f.n++] = g->ez; }
    if (which & 2) { f.p[f.n++] = g->bx; f.p[f.n++] = g->by; f.p[f.n++] = g->bz; }
    return f;
}

extern "C" int lpa_halo_pack_guard_src(const lpa_grid *g, int which, int side, double *buf,
                                       void *stream) {
    LPA_REQUIRE(g && buf && (side == 0 || side == 1) && g->nx >= g->ng, "lpa_halo_pack_guard_src: bad args");
    Ptr6 f = eb_ptrs(g, which);
    if (!f.n) return LPA_OK;
    // low face: interior rows [0, ng) -> padded rows [ng, 2ng); high face: [nx-ng, nx) -> [nx, nx+ng)
    long first = side == 0 ? g->ng : g->nx;
    return halo_launch(g, f, first, buf, HALO_PACK_SRC, stream, "lpa_halo_pack_guard_src");
}

extern "C" int lpa_halo_unpack_guard(const lpa_grid *g, int which, int side, const double *buf,
                                     void *stream) {
    LPA_REQUIRE(g && buf && (side == 0 || side == 1), "lpa_halo_unpack_guard: bad args");
    Ptr6 f = eb_ptrs(g, which);
    if (!f.n) return LPA_OK;
    long first = side == 0 ? 0 : g->nx + g->ng;  // my low guard / my high guard
    return halo_launch(g, f, first, (double *)buf, HALO_UNPACK_GUARD, stream, "lpa_halo_unpack_guard");
}

static Ptr6 cur_ptrs(const lpa_grid *g) {
    Ptr6 f;
    f.n = 4;
    f.p[0] = g->jx; f.p[1] = g->jy; f.p[2] = g->jz; f.p[3] = g->rho;
    return f;
}

extern "C" int lpa_halo_pack_current(const lpa_grid *g, int side, double *buf, void *stream) {
    LPA_REQUIRE(g && buf && g->jx && g->jy && g->jz && g->rho && (side == 0 || side == 1),
                "lpa_halo_pack_current: bad args");
    long first = side == 0 ? 0 : g->nx + g->ng;  // guard planes
    return halo_launch(g, cur_ptrs(g), first, buf, HALO_PACK_CUR, stream, "lpa_halo_pack_current");
}

extern "C" int lpa_halo_unpack_current(const lpa_grid *g, int side, const double *buf, void *stream) {
    LPA_REQUIRE(g && buf && g->jx && g->jy && g->jz && g->rho && (side == 0 || side == 1) &&
                    g->nx >= g->ng,
                "lpa_halo_unpack_current: bad args");
    long first = side == 0 ? g->ng : g->nx;  // interior edge planes
    return halo_launch(g, cur_ptrs(g), first, (double *)buf, HALO_UNPACK_CUR, stream,
                       "lpa_halo_unpack_current");
}

extern "C" int lpa_halo_faces(const lpa_grid *g, int op, int which, double *buf_lo, double *buf_hi,
                              void *stream) {
    LPA_REQUIRE(g && op >= HALO_PACK_SRC && op <= HALO_UNPACK_CUR && g->nx >= g->ng, "lpa_halo_faces: bad args");
    if (!buf_lo && !buf_hi) return LPA_OK;
    Ptr6 f;
    long first_lo, first_hi;
    if (op == HALO_PACK_CUR || op == HALO_UNPACK_CUR) {
        LPA_REQUIRE(g->jx && g->jy && g->jz && g->rho, "lpa_halo_faces: current arrays missing");
        f = cur_ptrs(g);
        // pack: my guard planes; unpack: added into my interior edge
        first_lo = op == HALO_PACK_CUR ? 0 : g->ng;
        first_hi = op == HALO_PACK_CUR ? g->nx + g->ng : g->nx;
    } else {
        f = eb_ptrs(g, which);
        if (!f.n) return LPA_OK;
        // pack: my interior edge rows; unpack: my guard rows
        first_lo = op == HALO_PACK_SRC ? g->ng : 0;
        first_hi = op == HALO_PACK_SRC ? g->nx : g->nx + g->ng;
    }
    long plane = (long)(g->ny + 2 * g->ng) * (g->nz > 1 ? g->nz + 2 * g->ng : 1);
    long n = (long)g->ng * plane;
    dim3 grid((unsigned)((n + 255) / 256), f.n, 2);
    hipLaunchKernelGGL(k_halo2, grid, dim3(256), 0, (hipStream_t)stream, f, buf_lo, buf_hi, plane, first_lo,
                       first_hi, g->ng, op);
    LPA_CHECK_LAUNCH("lpa_halo_faces");
    return LPA_OK;
}

// =====================================================================================================
// helpers of lpa_step (internal): per-step counter reset, and the slab form of the J / rho face fold
// =====================================================================================================
__global__ void k_zero_words(Words32 a, int n) {
    if ((int)threadIdx.x < n) *a.w[threadIdx.x] = 0u;
}

// current reset of a step (lpa_reset_current / lpa_reset_j) with the per-step counters zeroed by the same launch
__global__ void __launch_bounds__(256) k_reset_step(double *a, long n, double *b, long nb, Words32 w, int nw) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride) a[t] = 0.0;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < nb; t += stride) b[t] = 0.0;
    if (blockIdx.x == 0 && (int)threadIdx.x < nw) *w.w[threadIdx.x] = 0u;
}

int lpai_reset_step(const lpa_grid *g, int with_rho, double *also, uint32_t *const *words, int nwords, void *stream) {
    LPA_REQUIRE(g && g->jx && g->jy && g->jz && g->rho && nwords >= 0 && nwords <= 32, "lpai_reset_step: bad args");
    const long cnt = (long)(g->nx + 2 * g->ng) * (g->ny + 2 * g->ng) * (g->nz > 1 ? (long)(g->nz + 2 * g->ng) : 1);
    Words32 w;
    for (int i = 0; i < nwords; i++) w.w[i] = words[i];
    const bool contiguous = g->jy == g->jx + cnt && g->jz == g->jy + cnt && (!with_rho || g->rho == g->jz + cnt);
    if (!contiguous) {      // separate allocations: the plain entry points, then the counters
        if (int e = with_rho ? lpa_reset_current(g, stream) : lpa_reset_j(g, stream)) return e;
        if (also && hipMemsetAsync(also, 0, cnt * sizeof(double), (hipStream_t)stream) != hipSuccess) {
            lpa_set_error("lpai_reset_step: memset failed");
            return LPA_ERR_HIP;
        }
        return nwords ? lpai_zero_words(words, nwords, stream) : LPA_OK;
    }
    const long n = (with_rho ? 4 : 3) * cnt;
    long nb = (n + 1023) / 1024;
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(k_reset_step, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, g->jx, n, also, also ? cnt : 0L, w,
                       nwords);
    LPA_CHECK_LAUNCH("lpai_reset_step");
    return LPA_OK;
}

int lpai_zero_words(uint32_t *const *words, int n, void *stream) {
    for (int done = 0; done < n; done += 32) {
        Words32 a;
        const int k = n - done < 32 ? n - done : 32;
        for (int i = 0; i < k; i++) a.w[i] = words[done + i];
        hipLaunchKernelGGL(k_zero_words, dim3(1), dim3(64), 0, (hipStream_t)stream, a, k);
        LPA_CHECK_LAUNCH("lpai_zero_words");
    }
    return LPA_OK;
}

// The whole current fold of a step in ONE launch (lpa_step): what k_fold_faces-style face addition, k_current_fold and
// k_current_zero_guard do in three; the per-cell body lives in lpa_fold.hpp.
__global__ void __launch_bounds__(256) k_fold_all(GridV g, int axes, const double *__restrict__ r_lo,
                                                  const double *__restrict__ r_hi, const double *__restrict__ left_own) {
    fold_all_body(g, axes, r_lo, r_hi, left_own, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z);
}

int lpai_fold_all(const lpa_grid *g, int axes, const double *r_lo, const double *r_hi, const double *left_own, void *stream) {
    LPA_REQUIRE(g && g->jx && g->jy && g->jz && g->rho && g->nx > 0 && g->ny > 0 && g->ng > 0, "lpai_fold_all: bad grid");
    const int dim = g->nz > 1 ? 3 : 2;
    LPA_REQUIRE(g->nx >= 2 * g->ng && g->ny >= 2 * g->ng && (dim == 2 || g->nz >= 2 * g->ng),
                "lpai_fold_all: slab thinner than 2*ng");
    LPA_REQUIRE(!((r_lo || r_hi) && (axes & 1)), "lpai_fold_all: x is either folded locally or split over slabs");
    LPA_REQUIRE(!left_own || r_lo, "lpai_fold_all: the left neighbour's own jx plane comes with its guard planes");
    if (axes == 0 && !r_lo && !r_hi) return LPA_OK;
    GridV v;
    memset(&v, 0, sizeof v);
    v.nx = g->nx; v.ny = g->ny; v.nz = dim == 3 ? g->nz : 1; v.ng = g->ng;
    v.NX = g->nx + 2 * g->ng; v.NY = g->ny + 2 * g->ng; v.NZ = dim == 3 ? g->nz + 2 * g->ng : 1;
    v.jx = g->jx; v.jy = g->jy; v.jz = g->jz; v.rho = g->rho;
    dim3 gp = dim == 3 ? dim3((v.NZ + 255) / 256, v.NY, v.NX) : dim3((v.NY + 255) / 256, v.NX);
    hipLaunchKernelGGL(k_fold_all, gp, dim3(256), 0, (hipStream_t)stream, v, axes, r_lo, r_hi, left_own);
    LPA_CHECK_LAUNCH("lpai_fold_all");
    return LPA_OK;
}

// =====================================================================================================
// field diagnostics over the interior (reference tests/test_numerical_heating.py:19-37)
// =====================================================================================================
__global__ void __launch_bounds__(256) k_diag_fields(GridV g, double ce, double cb, double dv,
                                                     double *out) {
    long ncell = (long)g.nx * g.ny * g.nz;
    double e = 0, b = 0, r = 0, sx = 0, sy = 0, sz = 0;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < ncell;
         t += (long)gridDim.x * blockDim.x) {
        int k = (int)(t % g.nz);
        long r2 = t / g.nz;
        int j = (int)(r2 % g.ny), i = (int)(r2 / g.ny);
        long c = ((long)(i + g.ng) * g.NY + (j + g.ng)) * g.NZ + (g.NZ > 1 ? k + g.ng : 0);
        double a0 = g.ex[c], a1 = g.ey[c], a2 = g.ez[c];
        double b0 = g.bx[c], b1 = g.by[c], b2 = g.bz[c];
        e += a0 * a0 + a1 * a1 + a2 * a2;
        b += b0 * b0 + b1 * b1 + b2 * b2;
        if (g.rho) { r += g.rho[c]; sx += g.jx[c]; sy += g.jy[c]; sz += g.jz[c]; }
    }
    block_atomic_sum(e * ce * dv, out + 0);
    block_atomic_sum(b * cb * dv, out + 1);
    block_atomic_sum(r * dv, out + 2);
    block_atomic_sum(sx, out + 3);
    block_atomic_sum(sy, out + 4);
    block_atomic_sum(sz, out + 5);
}

extern "C" int lpa_diag_fields(const lpa_grid *g, double eps0, double mu0, double *out, void *stream) {
    int dim = g && g->nz > 1 ? 3 : 2;
    LPA_REQUIRE(lpa_grid_ok(g, dim, 0) && out && mu0 > 0, "lpa_diag_fields: bad args");
    GridV v = make_gridv(g, dim);
    if (!g->jx || !g->jy || !g->jz) v.rho = nullptr;
    double dv = g->dx * g->dy * (dim == 3 ? g->dz : 1.0);
    hipLaunchKernelGGL(k_diag_fields, dim3(512), dim3(256), 0, (hipStream_t)stream, v, 0.5 * eps0,
                       0.5 / mu0, dv, out);
    LPA_CHECK_LAUNCH("lpa_diag_fields");
    return LPA_OK;
}

// lpa_rho.hip -- rho from the discrete continuity equation: the grid-side companions of the LPA_PUSH_NO_RHO form of the
// fused particle kernels (lpa_particles.hip, lpa_particles3d.hip).
//
// The reference deposits rho together with the currents (current/current_deposit.h:180,  :436-439: `rho += q w /
// (dx dy [dz]) S1x S1y [S1z]`).  Esirkepov's scheme is built so that this rho and the deposited currents satisfy
//     (rho1 - rho0) / dt + (jx[i] - jx[i-1]) / dx + (jy[j] - jy[j-1]) / dy [+ (jz[k] - jz[k-1]) / dz] = 0
// per particle and per node (rho0 = the same particle's shape at the start of the step).  On the device the rho
// atomics are 9 of the 30 (2-D) / 27 of the 81 (3-D) LDS atomics of the kernel that decides the step time, so between
// two real deposits rho is advanced from the folded currents here -- streaming kernels over the grid, 5-6 doubles per
// node, against 27 LDS atomics per particle.
#include "lpa_common.hpp"
#include "lpa_tail.hpp"

// ---- jx jy jz = 0 (rho persists) ---------------------------------------------------------------------------------
extern "C" int lpa_reset_j(const lpa_grid *g, void *stream) {
    LPA_REQUIRE(g && g->jx && g->jy && g->jz, "lpa_reset_j: bad grid");
    const size_t cnt = (size_t)(g->nx + 2 * g->ng) * (g->ny + 2 * g->ng) * (g->nz > 1 ? (size_t)(g->nz + 2 * g->ng) : 1);
    const size_t n = cnt * sizeof(double);
    double *a[3] = {g->jx, g->jy, g->jz};
    if (a[1] == a[0] + cnt && a[2] == a[1] + cnt) {  // one allocation: one memset
        if (hipMemsetAsync(a[0], 0, 3 * n, (hipStream_t)stream) != hipSuccess) {
            lpa_set_error("lpa_reset_j: hipMemsetAsync failed");
            return LPA_ERR_HIP;
        }
        return LPA_OK;
    }
    for (int c = 0; c < 3; c++)
        if (hipMemsetAsync(a[c], 0, n, (hipStream_t)stream) != hipSuccess) {
            lpa_set_error("lpa_reset_j: hipMemsetAsync failed");
            return LPA_ERR_HIP;
        }
    return LPA_OK;
}

// ---- rho -= dt div J (the per-cell body: lpa_tail.hpp -- a B sweep's launch can carry it too) --------------------------------
__global__ void __launch_bounds__(256) k_rho_continuity(GridV g, double dtdx, double dtdy, double dtdz, int mx, int my,
                                                        int mz, const double *__restrict__ left) {
    // one x plane per blockIdx.y, its (y, z) nodes flattened over blockIdx.x (a 262-node z row per block would leave
    // the second of its two 256-thread blocks with six lanes)
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= g.NY * g.NZ) return;
    const int cx = blockIdx.y, cy = f / g.NZ, cz = f - cy * g.NZ;
    rho_continuity_cell(g, cx, cy, cz, dtdx, dtdy, dtdz, mx, my, mz, left);
}

extern "C" int lpa_rho_continuity(const lpa_grid *g, double dt, int periodic_axes, int split_x,
                                  const double *jx_left_plane, void *stream) {
    const int dim = g && g->nz > 1 ? 3 : 2;
    LPA_REQUIRE(lpa_grid_ok(g, dim, 1) && dt > 0, "lpa_rho_continuity: bad args");
    LPA_REQUIRE(!(split_x && (periodic_axes & 1)), "lpa_rho_continuity: x is either folded locally or split");
    LPA_REQUIRE(split_x >= 0 && split_x <= 3 && (!(split_x & 1) || jx_left_plane),
                "lpa_rho_continuity: a slab with a left neighbour needs its jx plane");
    GridV v = make_gridv(g, dim);
    const int mx = split_x ? ((split_x & 1 ? RHO_NB_LO : 0) | (split_x & 2 ? RHO_NB_HI : 0)) : (periodic_axes & 1),
              my = (periodic_axes >> 1) & 1, mz = (periodic_axes >> 2) & 1;
    dim3 grid((v.NY * v.NZ + 255) / 256, v.NX);
    hipLaunchKernelGGL(k_rho_continuity, grid, dim3(256), 0, (hipStream_t)stream, v, dt / g->dx, dt / g->dy,
                       dim == 3 ? dt / g->dz : 0.0, mx, my, mz, jx_left_plane);
    LPA_CHECK_LAUNCH("lpa_rho_continuity");
    return LPA_OK;
}

// ---- the charge of absorbed particles leaves rho -----------------------------------------------------------------
// listed entries: minus their TSC shape; entries that found the list full were spread into `spill` by the push kernels
// (lpa_push_params.absorbed_spill): subtracted node by node and zeroed here
__global__ void __launch_bounds__(256) k_rho_absorbed(GridV g, const double *__restrict__ list,
                                                      const uint32_t *__restrict__ count, long capacity, double *spill) {
    const long total = count[0];
    const long n = min(total, capacity);
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long)gridDim.x * blockDim.x) {
        const double *e = list + 4 * t;
        spread_tsc(g, g.rho, e[0], e[1], e[2], -e[3]);
    }
    if (spill && total > capacity) {        // (uniform; rare: more absorptions in one step than the list holds)
        const long cells = (long)g.NX * g.NY * g.NZ;
        for (long c = (long)blockIdx.x * blockDim.x + threadIdx.x; c < cells; c += (long)gridDim.x * blockDim.x) {
            const double v = spill[c];
            if (v != 0.0) { atomicAdd(&g.rho[c], -v); spill[c] = 0.0; }
        }
    }
}

__global__ void k_rho_absorbed_done(uint32_t *count, long capacity, int spilled) {
    const uint32_t n = count[0];
    if (!spilled && (long)n > capacity) count[1] += (uint32_t)((long)n - capacity);
    count[0] = 0;
}

extern "C" int lpa_rho_absorbed_spill(const lpa_grid *g, const double *list, uint32_t *count, int64_t capacity,
                                      double *spill, void *stream) {
    const int dim = g && g->nz > 1 ? 3 : 2;
    LPA_REQUIRE(lpa_grid_ok(g, dim, 1) && list && count && capacity > 0, "lpa_rho_absorbed: bad args");
    // absorptions are rare (particles reaching an open face): a small fixed grid walks whatever the list holds
    hipLaunchKernelGGL(k_rho_absorbed, dim3(64), dim3(256), 0, (hipStream_t)stream, make_gridv(g, dim), list, count,
                       (long)capacity, spill);
    LPA_CHECK_LAUNCH("lpa_rho_absorbed");
    hipLaunchKernelGGL(k_rho_absorbed_done, dim3(1), dim3(1), 0, (hipStream_t)stream, count, (long)capacity, spill ? 1 : 0);
    LPA_CHECK_LAUNCH("lpa_rho_absorbed_done");
    return LPA_OK;
}

extern "C" int lpa_rho_absorbed(const lpa_grid *g, const double *list, uint32_t *count, int64_t capacity,
                                void *stream) {
    return lpa_rho_absorbed_spill(g, list, count, capacity, nullptr, stream);
}

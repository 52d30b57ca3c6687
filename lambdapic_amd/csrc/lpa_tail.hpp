// lpa_tail.hpp -- small grid-wide jobs that ride in a B sweep's launch (lpa_step): the current reset of the step that follows
// the first B half step, and the rho continuity update that precedes the second.  Each is a launch of ~5 us on its own -- the
// floor a kernel occupies the stream however little it does -- and touches nothing the sweep reads or writes (the sweep reads
// E and writes B; the reset zeroes J [and rho], the continuity update reads the folded J and advances rho).
#pragma once
#include "lpa_common.hpp"

struct Words32 { uint32_t *w[32]; };

// ---- rho -= dt div J: one cell (lpa_rho.hip: k_rho_continuity) --------------------------------------------------------------
// mode per axis: bit 0 = folded inside the slab (periodic: interior nodes, node 0 -> node n - 1); else bit 1 / bit 2 =
// the low / high face has a neighbour slab (its guard planes were sent away: interior nodes on that side, node 0 ->
// the left neighbour's plane); a face with neither is open: its guard nodes are updated too, on the padded torus --
// where the deposit itself lands
constexpr int RHO_PERIODIC = 1, RHO_NB_LO = 2, RHO_NB_HI = 4;
__device__ __forceinline__ bool rho_axis(int c, int n, int ng, int N, int mode, int &prev) {
    if (mode & RHO_PERIODIC) {
        if (c < ng || c >= ng + n) return false;
        prev = c == ng ? ng + n - 1 : c - 1;
        return true;
    }
    if (((mode & RHO_NB_LO) && c < ng) || ((mode & RHO_NB_HI) && c >= ng + n)) return false;
    prev = c == 0 ? N - 1 : c - 1;
    return true;
}

__device__ __forceinline__ void rho_continuity_cell(const GridV &g, int cx, int cy, int cz, double dtdx, double dtdy, double dtdz,
                                                    int mx, int my, int mz, const double *__restrict__ left) {
    const bool d3 = g.NZ > 1;
    int px, py, pz = 0;
    if (!rho_axis(cx, g.nx, g.ng, g.NX, mx, px) || !rho_axis(cy, g.ny, g.ng, g.NY, my, py)) return;
    if (d3 && !rho_axis(cz, g.nz, g.ng, g.NZ, mz, pz)) return;
    const long sX = (long)g.NY * g.NZ, sY = g.NZ;
    const long c = cx * sX + cy * sY + cz;
    const double jxp = ((mx & RHO_NB_LO) && cx == g.ng) ? left[cy * sY + cz] : g.jx[px * sX + cy * sY + cz];
    double div = (g.jx[c] - jxp) * dtdx + (g.jy[c] - g.jy[cx * sX + py * sY + cz]) * dtdy;
    if (d3) div += (g.jz[c] - g.jz[cx * sX + cy * sY + pz]) * dtdz;
    g.rho[c] -= div;
}

// ---- what a B sweep's launch does beside the sweep ------------------------------------------------------------------------------
constexpr int B_TAIL_NONE = 0, B_TAIL_RESET = 1, B_TAIL_RHO = 2;
struct BTail {
    int mode;
    // B_TAIL_RESET: a[0, n) = 0, b[0, nb) = 0, *w[0 .. nw) = 0  (k_reset_step)
    double *a;
    long n;
    double *b;
    long nb;
    int nw;
    // B_TAIL_RHO  (k_rho_continuity)
    double dtdx, dtdy, dtdz;
    int mx, my, mz;
    const double *left;
    Words32 w;
};

// called by every thread of the launch BEFORE it looks at its own cell (threads beyond the sweep's range take part)
__device__ __forceinline__ void b_tail(const GridV &g, const BTail &t) {
    if (t.mode == B_TAIL_NONE) return;
    const long nthreads = (long)gridDim.x * gridDim.y * gridDim.z * blockDim.x;
    const long tid = (((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x;
    if (t.mode == B_TAIL_RESET) {
        for (long i = tid; i < t.n; i += nthreads) t.a[i] = 0.0;
        for (long i = tid; i < t.nb; i += nthreads) t.b[i] = 0.0;
        if (tid < t.nw) *t.w.w[tid] = 0u;
        return;
    }
    const long plane = (long)g.NY * g.NZ, cells = (long)g.NX * plane;
    for (long f = tid; f < cells; f += nthreads) {
        const int cx = (int)(f / plane);
        const long r = f - (long)cx * plane;
        const int cy = (int)(r / g.NZ), cz = (int)(r - (long)cy * g.NZ);
        rho_continuity_cell(g, cx, cy, cz, t.dtdx, t.dtdy, t.dtdz, t.mx, t.my, t.mz, t.left);
    }
}

// host side: what lpa_step asks a B sweep to carry
struct lpai_tail {
    int mode;                       // B_TAIL_RESET / B_TAIL_RHO
    int with_rho;                   // reset: zero rho too (a real-deposit step)
    double *also;                   // reset: one more array shaped like rho (may be NULL)
    uint32_t *const *words;         // reset: per-step counters
    int nwords;
    double dt;                      // rho: the arguments of lpa_rho_continuity
    int periodic_axes, split_x;
    const double *left;
};

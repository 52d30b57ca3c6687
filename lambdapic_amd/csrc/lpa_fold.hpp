// lpa_fold.hpp -- the per-cell body of the slab form of the J / rho fold (k_fold_all, lpa_fields.hip), shared with the
// launch that folds AND seats the arrivals of every species (k_fold_unpack, lpa_sort.hip): both follow the J round of a slab
// step and touch disjoint data, so lpa_step issues them as one launch.
#pragma once
#include "lpa_common.hpp"

// One thread per padded cell (bx, by, bz = the block indices of k_fold_all's grid):
//   * an interior cell adds up its periodic images along the local axes in the reference's order (x, y, z, xy, xz, yz,
//     xyz: core/patch/sync_fields2d.c:43-148) and zeroes each image after reading it -- every consumed guard cell is the
//     image of exactly one interior cell (n >= 2 ng), so nobody else reads or writes it;
//   * on a slab rank the planes received from a neighbour (r_lo / r_hi: [4][ng][plane], NULL = no neighbour) are added
//     on the fly, to the cell and to its y / z images alike: (f + r) per cell first, then the fold -- the sums the
//     separate launches form (fill of sync_currents, core/mpi/sync_fields2d.c:76-102);
//   * a cell of an x guard plane that was sent to a neighbour is zeroed (:44-74) -- except, with `left_own` (the LEFT
//     neighbour's own jx deposit on its last node plane, [plane] doubles, which travelled with the guard planes), the jx
//     plane at node -1: it becomes the neighbour's FOLDED jx there -- (mine + its own) per cell, then the y / z images, the
//     very sums the neighbour forms for that plane, bit for bit -- which the backward difference of the rho continuity
//     update at node 0 reads (no message of its own for that plane).
__device__ __forceinline__ void fold_all_body(const GridV &g, int axes, const double *__restrict__ r_lo,
                                              const double *__restrict__ r_hi, const double *__restrict__ left_own,
                                              int bx, int by, int bz) {
    const int z = bx * (int)blockDim.x + (int)threadIdx.x;
    const bool d3 = g.NZ > 1;
    const int NF = d3 ? g.NZ : g.NY;
    if (z >= NF) return;
    const int cx = d3 ? bz : by, cy = d3 ? by : z, cz = d3 ? z : 0;
    const int ng = g.ng;
    const long sY = g.NZ, sX = (long)g.NY * g.NZ;
    const long c = (long)cx * sX + (long)cy * sY + cz;
    double *arr[4] = {g.jx, g.jy, g.jz, g.rho};
    const int i = cx - ng, j = cy - ng, k = d3 ? cz - ng : 0;
    if (i < 0 || i >= g.nx) {                       // x guard plane: sent to a neighbour -> zero
        if (!((i < 0 && r_lo) || (i >= g.nx && r_hi))) return;
        const bool mirror = i == -1 && left_own;
#pragma unroll
        for (int a = mirror ? 1 : 0; a < 4; a++) arr[a][c] = 0.0;
        if (!mirror) return;
        double *f = g.jx;
        const long pc = (long)cy * sY + cz;
        const int j = cy - ng, k = d3 ? cz - ng : 0;
        const bool gy = j < 0 || j >= g.ny, gz = d3 && (k < 0 || k >= g.nz);
        if (gy || gz) {     // (a consumed y / z guard cell is read by its owner below and rewritten by the next reset)
            const bool consumed = !((gy && !(axes & 2)) || (gz && !(axes & 4)));
            if (!consumed) f[c] += left_own[pc];
            return;
        }
        const int oy = (axes & 2) ? (j < ng ? g.ny : (j >= g.ny - ng ? -g.ny : 0)) : 0;
        const int oz = (d3 && (axes & 4)) ? (k < ng ? g.nz : (k >= g.nz - ng ? -g.nz : 0)) : 0;
        double v = f[c] + left_own[pc];
        if (oy) v += f[c + oy * sY] + left_own[pc + oy * sY];
        if (oz) v += f[c + oz] + left_own[pc + oz];
        if (oy && oz) v += f[c + oy * sY + oz] + left_own[pc + oy * sY + oz];
        f[c] = v;
        return;
    }
    // received planes cover the interior edge rows i < ng (low face) / i >= nx - ng (high face), whole planes
    const double *r = (r_lo && i < ng) ? r_lo : ((r_hi && i >= g.nx - ng) ? r_hi : nullptr);
    const long n = (long)ng * sX;                                        // doubles per component in a face message
    const long rbase = r ? (long)(r == r_lo ? i : i - (g.nx - ng)) * sX + (long)cy * sY + cz : 0;
    const bool gy = j < 0 || j >= g.ny, gz = d3 && (k < 0 || k >= g.nz);
    if (gy || gz) {
        // a y / z guard cell that the fold consumes is handled by its interior owner (read with its received share,
        // then zeroed); one at an open face keeps what was deposited there and takes the neighbour's share here
        const bool consumed = !((gy && !(axes & 2)) || (gz && !(axes & 4)));
        if (!consumed && r) {
#pragma unroll
            for (int a = 0; a < 4; a++) arr[a][c] += r[(long)a * n + rbase];
        }
        return;
    }
    const int ox = (axes & 1) ? (i < ng ? g.nx : (i >= g.nx - ng ? -g.nx : 0)) : 0;
    const int oy = (axes & 2) ? (j < ng ? g.ny : (j >= g.ny - ng ? -g.ny : 0)) : 0;
    const int oz = (d3 && (axes & 4)) ? (k < ng ? g.nz : (k >= g.nz - ng ? -g.nz : 0)) : 0;
    if (!(ox | oy | oz) && !r) return;
#pragma unroll
    for (int a = 0; a < 4; a++) {
        double *f = arr[a];
        const double *ra = r ? r + (long)a * n + rbase : nullptr;
        auto take = [&](long off) {                 // value of the image at c + off (with its received share), then zero it
            double v = f[c + off];
            if (ra) v += ra[off];
            f[c + off] = 0.0;
            return v;
        };
        double v = f[c];
        if (ra) v += ra[0];
        if (ox) v += take(ox * sX);
        if (oy) v += take(oy * sY);
        if (oz) v += take(oz);
        if (ox && oy) v += take(ox * sX + oy * sY);
        if (ox && oz) v += take(ox * sX + oz);
        if (oy && oz) v += take(oy * sY + oz);
        if (ox && oy && oz) v += take(ox * sX + oy * sY + oz);
        f[c] = v;
    }
}

// species whose arrivals one k_fold_unpack launch seats
constexpr int LPA_FOLD_UNPACK_MAX_SPECIES = 4;
struct lpa_unpack_args {
    const lpa_particles *p;
    const lpa_tiling *t;
    const lpa_free_slots *fs;       // NULL: arrivals go to the arrival area only
    int64_t first_slot, area_capacity;
    int32_t *cursor;
    const double *buf_lo, *buf_hi;
};
// lpai_fold_all + lpai_migrate_unpack2 of up to LPA_FOLD_UNPACK_MAX_SPECIES species in ONE launch
int lpai_fold_unpack(const lpa_grid *g, int axes, const double *r_lo, const double *r_hi, const double *left_own,
                     const lpa_unpack_args *u, int nspecies, int64_t capacity, double shift_lo, double shift_hi, void *stream);

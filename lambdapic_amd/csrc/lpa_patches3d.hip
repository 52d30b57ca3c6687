// lpa_patches3d.hip -- the 3-D twins of the patch-list drop-ins in lpa_patches.hip: the reference's PATCH-LIST data
// model (per-patch arrays in the WRAPPED guard layout, core/fields.py:24-27; neighbour tables neighbor_ipatch[26] in
// Boundary3D order, core/patch/patch.py:37-69) kept on the device.
//
//   lpa_sync_guard_fields_3d     core/patch/sync_fields3d.c:350-612   guard <- neighbour's interior edge, 26 neighbours
//   lpa_sync_currents_3d         core/patch/sync_fields3d.c:84-348    interior edge += neighbour's guard; guard = 0
//   lpa_sync_particles_count_3d  core/patch/sync_particles_3d.c:78-192,365-437   leavers per boundary, dead slots
//   lpa_sync_particles_fill_3d   core/patch/sync_particles_3d.c:194-361,484-700  incoming -> dead slots, +- L, kill
//
// One slab per GPU has no intra-GPU patches, so the resident engines do not use these: they are what a lambdaPIC
// facade binds when it keeps its own patch lists.
#include "lpa_common.hpp"

constexpr int NB3 = 26;
__device__ __forceinline__ int widx3(int i, int N) { return i < 0 ? i + N : i; }

// Boundary3D (sync_fields3d.c:19-50): faces, edges (xy, xz, yz), vertices; (0, 0, 0) -> -1
__host__ __device__ __forceinline__ int boundary_of3(int sx, int sy, int sz) {
    const int nz = (sx != 0) + (sy != 0) + (sz != 0);
    if (nz == 0) return -1;
    if (nz == 1) return sx ? (sx > 0) : (sy ? 2 + (sy > 0) : 4 + (sz > 0));
    if (nz == 3) return 18 + 4 * (sx > 0) + 2 * (sy > 0) + (sz > 0);
    if (sz == 0) return 6 + 4 * (sx > 0) + (sy > 0);      // XMINYMIN XMINYMAX . . XMAXYMIN XMAXYMAX
    if (sy == 0) return 8 + 4 * (sx > 0) + (sz > 0);      // XMINZMIN XMINZMAX . . XMAXZMIN XMAXZMAX
    return 14 + 2 * (sy > 0) + (sz > 0);                  // YMINZMIN YMINZMAX YMAXZMIN YMAXZMAX
}

struct Side3 { signed char s[NB3][3]; signed char opp[NB3]; };
static Side3 make_sides() {
    Side3 t;
    for (int sx = -1; sx <= 1; sx++)
        for (int sy = -1; sy <= 1; sy++)
            for (int sz = -1; sz <= 1; sz++) {
                const int b = boundary_of3(sx, sy, sz);
                if (b < 0) continue;
                t.s[b][0] = (signed char)sx; t.s[b][1] = (signed char)sy; t.s[b][2] = (signed char)sz;
                t.opp[b] = (signed char)boundary_of3(-sx, -sy, -sz);      // OPPOSITE_BOUNDARY (:52-82)
            }
    return t;
}

// ---- guard copy: one thread per (array, padded node) ---------------------------------------------------------
__global__ void __launch_bounds__(256) k_sync_guard_patches_3d(double *const *__restrict__ arrays, int ncomp,
                                                               const int64_t *__restrict__ neighbor, int nx, int ny,
                                                               int nz, int ng) {
    const int NX = nx + 2 * ng, NY = ny + 2 * ng, NZ = nz + 2 * ng;
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)NX * NY * NZ) return;
    const int p = blockIdx.y / ncomp, c = blockIdx.y % ncomp;
    const int li = (int)(t / ((long)NY * NZ)) - ng, lj = (int)((t / NZ) % NY) - ng, lk = (int)(t % NZ) - ng;
    const int sx = li < 0 ? -1 : (li >= nx ? 1 : 0), sy = lj < 0 ? -1 : (lj >= ny ? 1 : 0),
              sz = lk < 0 ? -1 : (lk >= nz ? 1 : 0);
    const int b = boundary_of3(sx, sy, sz);
    if (b < 0) return;
    const long q = neighbor[(long)p * NB3 + b];
    if (q < 0) return;
    // a guard node on side s of patch p is the neighbour's interior node one patch width away (:380-612)
    const double *src = arrays[q * ncomp + c];
    double *dst = arrays[(long)p * ncomp + c];
    dst[((long)widx3(li, NX) * NY + widx3(lj, NY)) * NZ + widx3(lk, NZ)] =
        src[((long)(li - sx * nx) * NY + (lj - sy * ny)) * NZ + (lk - sz * nz)];
}

// ---- current fold: one thread per (array, interior node within ng of a face); the neighbours' guard nodes are added in
// Boundary3D order -- the order of the reference's sweep (:131-343) -- so the sum is bit-identical to the CPU's
__global__ void __launch_bounds__(256) k_sync_currents_patches_3d(double *const *__restrict__ arrays,
                                                                  const int64_t *__restrict__ neighbor, int nx, int ny,
                                                                  int nz, int ng, Side3 sd) {
    const int NX = nx + 2 * ng, NY = ny + 2 * ng, NZ = nz + 2 * ng;
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)nx * ny * nz) return;
    const int p = blockIdx.y / 4, c = blockIdx.y % 4;
    const int i = (int)(t / ((long)ny * nz)), j = (int)((t / nz) % ny), k = (int)(t % nz);
    const bool lo[3] = {i < ng, j < ng, k < ng}, hi[3] = {i >= nx - ng, j >= ny - ng, k >= nz - ng};
    if (!(lo[0] || hi[0] || lo[1] || hi[1] || lo[2] || hi[2])) return;
    double *dst = arrays[(long)p * 4 + c];
    const long o = ((long)i * NY + j) * NZ + k;
    double v = dst[o];
    for (int b = 0; b < NB3; b++) {
        bool in = true;
#pragma unroll
        for (int a = 0; a < 3; a++) in = in && (sd.s[b][a] == 0 || (sd.s[b][a] < 0 ? lo[a] : hi[a]));
        if (!in) continue;
        const long q = neighbor[(long)p * NB3 + b];
        if (q < 0) continue;
        // the same node in the neighbour's frame: its upper guard [n, n + ng) / lower guard [-ng, 0)
        const int si = i - sd.s[b][0] * nx, sj = j - sd.s[b][1] * ny, sk = k - sd.s[b][2] * nz;
        v += arrays[q * 4 + c][((long)widx3(si, NX) * NY + widx3(sj, NY)) * NZ + widx3(sk, NZ)];
    }
    dst[o] = v;
}

// a guard region on side s was consumed (by the neighbour on that side) iff the patch has a neighbour there
__global__ void __launch_bounds__(256) k_zero_consumed_guards_3d(double *const *__restrict__ arrays,
                                                                 const int64_t *__restrict__ neighbor, int nx, int ny,
                                                                 int nz, int ng) {
    const int NX = nx + 2 * ng, NY = ny + 2 * ng, NZ = nz + 2 * ng;
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)NX * NY * NZ) return;
    const int p = blockIdx.y / 4, c = blockIdx.y % 4;
    const int li = (int)(t / ((long)NY * NZ)) - ng, lj = (int)((t / NZ) % NY) - ng, lk = (int)(t % NZ) - ng;
    const int b = boundary_of3(li < 0 ? -1 : (li >= nx ? 1 : 0), lj < 0 ? -1 : (lj >= ny ? 1 : 0),
                               lk < 0 ? -1 : (lk >= nz ? 1 : 0));
    if (b < 0 || neighbor[(long)p * NB3 + b] < 0) return;
    arrays[(long)p * 4 + c][((long)widx3(li, NX) * NY + widx3(lj, NY)) * NZ + widx3(lk, NZ)] = 0.0;
}

extern "C" int lpa_sync_guard_fields_3d(double *const *arrays, int32_t ncomp, const int64_t *neighbor_ipatch,
                                        int32_t npatches, int32_t nx, int32_t ny, int32_t nz, int32_t ng,
                                        void *stream) {
    LPA_REQUIRE(arrays && neighbor_ipatch && ncomp >= 1 && npatches >= 0 && nx >= ng && ny >= ng && nz >= ng && ng >= 1,
                "lpa_sync_guard_fields_3d: bad args (patches must be at least n_guard cells wide)");
    if (npatches == 0) return LPA_OK;
    const long cells = (long)(nx + 2 * ng) * (ny + 2 * ng) * (nz + 2 * ng);
    hipLaunchKernelGGL(k_sync_guard_patches_3d, dim3((unsigned)((cells + 255) / 256), npatches * ncomp), dim3(256), 0,
                       (hipStream_t)stream, arrays, ncomp, neighbor_ipatch, nx, ny, nz, ng);
    LPA_CHECK_LAUNCH("lpa_sync_guard_fields_3d");
    return LPA_OK;
}

extern "C" int lpa_sync_currents_3d(double *const *arrays, const int64_t *neighbor_ipatch, int32_t npatches,
                                    int32_t nx, int32_t ny, int32_t nz, int32_t ng, void *stream) {
    LPA_REQUIRE(arrays && neighbor_ipatch && npatches >= 0 && nx >= 2 * ng && ny >= 2 * ng && nz >= 2 * ng && ng >= 1,
                "lpa_sync_currents_3d: bad args (patches must be at least 2 n_guard cells wide)");
    if (npatches == 0) return LPA_OK;
    hipLaunchKernelGGL(k_sync_currents_patches_3d, dim3((unsigned)(((long)nx * ny * nz + 255) / 256), npatches * 4),
                       dim3(256), 0, (hipStream_t)stream, arrays, neighbor_ipatch, nx, ny, nz, ng, make_sides());
    LPA_CHECK_LAUNCH("lpa_sync_currents_3d (fold)");
    const long cells = (long)(nx + 2 * ng) * (ny + 2 * ng) * (nz + 2 * ng);
    hipLaunchKernelGGL(k_zero_consumed_guards_3d, dim3((unsigned)((cells + 255) / 256), npatches * 4), dim3(256), 0,
                       (hipStream_t)stream, arrays, neighbor_ipatch, nx, ny, nz, ng);
    LPA_CHECK_LAUNCH("lpa_sync_currents_3d (zero)");
    return LPA_OK;
}

// =====================================================================================================
// particle ownership between the patches of a list, 3-D (Patches.sync_particles, core/patch/patch.py:739-763)
// =====================================================================================================
constexpr int SP3_DEAD = 26, SP3_STAY = 27, SP3_NCLS = 27;     // classes 0..25 = Boundary3D, 26 = dead slot

struct XYZTab {
    const double *const *tab;
    int stride, ix, iy, iz;
    __device__ __forceinline__ const double *get(int p, int a) const {
        return tab[(long)p * stride + (a == 0 ? ix : (a == 1 ? iy : iz))];
    }
};

// count_outgoing_particles (sync_particles_3d.c:78-192): per axis below min / above max / inside
__device__ __forceinline__ int sp3_class(double x, double y, double z, bool dead, const double *b) {
    if (dead) return SP3_DEAD;
    const int sx = x < b[0] ? -1 : (x > b[1] ? 1 : 0), sy = y < b[2] ? -1 : (y > b[3] ? 1 : 0),
              sz = z < b[4] ? -1 : (z > b[5] ? 1 : 0);
    const int c = boundary_of3(sx, sy, sz);
    return c < 0 ? SP3_STAY : c;
}

__global__ void __launch_bounds__(256) k_sync_particles_count_3d(XYZTab xyz, const uint8_t *const *is_dead,
                                                                 const int64_t *npart, const double *bounds,
                                                                 unsigned long long *nout, unsigned long long *ndead) {
    const int p = blockIdx.y;
    const long n = npart[p];
    const double *x = xyz.get(p, 0), *y = xyz.get(p, 1), *z = xyz.get(p, 2);
    const uint8_t *dead = is_dead[p];
    __shared__ unsigned int s_cnt[SP3_NCLS];
    if (threadIdx.x < SP3_NCLS) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    for (long ip = (long)blockIdx.x * blockDim.x + threadIdx.x; ip < n; ip += (long)gridDim.x * blockDim.x) {
        const int c = sp3_class(x[ip], y[ip], z[ip], dead[ip] != 0, bounds + 6 * p);
        if (c != SP3_STAY) atomicAdd(&s_cnt[c], 1u);
    }
    __syncthreads();
    if (threadIdx.x < NB3 && s_cnt[threadIdx.x])
        atomicAdd(&nout[(long)p * NB3 + threadIdx.x], (unsigned long long)s_cnt[threadIdx.x]);
    if (threadIdx.x == SP3_DEAD && s_cnt[SP3_DEAD]) atomicAdd(&ndead[p], (unsigned long long)s_cnt[SP3_DEAD]);
}

// one workgroup per patch: stable rank of every leaver inside its class and of every dead slot among the dead
// (ascending index, like the reference's sequential loops: get_incoming_index :209-300); leavers listed class by class
__global__ void __launch_bounds__(256) k_sync_particles_rank_3d(XYZTab xyz, const uint8_t *const *is_dead,
                                                                const int64_t *npart, const double *bounds,
                                                                const int64_t *nout, int32_t *list, int32_t *drank,
                                                                long stride) {
    const int p = blockIdx.x;
    const long n = npart[p];
    const double *x = xyz.get(p, 0), *y = xyz.get(p, 1), *z = xyz.get(p, 2);
    const uint8_t *dead = is_dead[p];
    __shared__ int s_base[SP3_NCLS], s_w[4][SP3_NCLS], s_cbase[NB3];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x < SP3_NCLS) s_base[threadIdx.x] = 0;
    if (threadIdx.x == 0) {
        int run = 0;
        for (int c = 0; c < NB3; c++) { s_cbase[c] = run; run += (int)nout[(long)p * NB3 + c]; }
    }
    __syncthreads();
    for (long c0 = 0; c0 < n; c0 += 256) {
        const long ip = c0 + threadIdx.x;
        const int cls = ip < n ? sp3_class(x[ip], y[ip], z[ip], dead[ip] != 0, bounds + 6 * p) : SP3_STAY;
        int pre = 0;
        for (int c = 0; c < SP3_NCLS; c++) {          // wave-uniform loop: rank inside the wave, per class
            const unsigned long long m = __ballot(cls == c);
            if (cls == c) pre = __popcll(m & ((1ull << lane) - 1ull));
            if (lane == 0) s_w[wv][c] = __popcll(m);
        }
        __syncthreads();
        if (cls != SP3_STAY) {
            int r = s_base[cls] + pre;
            for (int w = 0; w < wv; w++) r += s_w[w][cls];
            if (cls == SP3_DEAD) drank[(long)p * stride + ip] = r;
            else list[(long)p * stride + s_cbase[cls] + r] = (int32_t)ip;
        }
        __syncthreads();
        if (threadIdx.x < SP3_NCLS)
            s_base[threadIdx.x] += s_w[0][threadIdx.x] + s_w[1][threadIdx.x] + s_w[2][threadIdx.x] + s_w[3][threadIdx.x];
        __syncthreads();
    }
}

__device__ __forceinline__ double sp3_periodic(double v, double gmin, double gmax, double pmin, double pmax,
                                               double cell) {   // handle_periodic, sync_particles_3d.c:347-361
    const double L = gmax - gmin;
    double out = v;
    if (v > gmax && fabs(pmin - gmin) < cell) out -= L;
    if (v < gmin && fabs(pmax - gmax) < cell) out += L;
    return out;
}

struct Glob3 { double lo[3], hi[3], d[3]; };

__global__ void __launch_bounds__(256) k_sync_particles_fill_3d(double *const *attrs, int nattrs, int iax, int iay,
                                                                int iaz, uint8_t *const *is_dead, const int64_t *npart,
                                                                const double *bounds, const int64_t *neighbor,
                                                                const int64_t *nin, const int64_t *nout,
                                                                const int32_t *list, const int32_t *drank, long stride,
                                                                Glob3 gl, Side3 sd) {
    const int p = blockIdx.y;
    const long n = npart[p], new_n = nin[p];
    uint8_t *dead = is_dead[p];
    for (long ip = (long)blockIdx.x * blockDim.x + threadIdx.x; ip < n; ip += (long)gridDim.x * blockDim.x) {
        if (!dead[ip]) continue;
        long r = drank[(long)p * stride + ip];         // the r-th dead slot takes incoming particle r (:643-684)
        if (r >= new_n) continue;
        long q = -1, src = -1;
        for (int b = 0; b < NB3 && q < 0; b++) {       // incoming order: boundary by boundary (:302-322)
            const long nb = neighbor[(long)p * NB3 + b];
            if (nb < 0) continue;
            const int ob = sd.opp[b];
            const long cnt = nout[nb * NB3 + ob];
            if (r < cnt) {
                long base = 0;
                for (int c = 0; c < ob; c++) base += nout[nb * NB3 + c];
                q = nb;
                src = list[nb * stride + base + r];
            } else {
                r -= cnt;
            }
        }
        if (q < 0) continue;                            // (counts inconsistent with the arrays: nothing to take)
        const double *bp = bounds + 6 * p;
        for (int a = 0; a < nattrs; a++) {
            double v = attrs[q * nattrs + a][src];
            if (a == iax) v = sp3_periodic(v, gl.lo[0], gl.hi[0], bp[0], bp[1], gl.d[0]);
            if (a == iay) v = sp3_periodic(v, gl.lo[1], gl.hi[1], bp[2], bp[3], gl.d[1]);
            if (a == iaz) v = sp3_periodic(v, gl.lo[2], gl.hi[2], bp[4], bp[5], gl.d[2]);
            attrs[(long)p * nattrs + a][ip] = v;
        }
        dead[ip] = 0;
    }
}

// mark_out_of_bound_as_dead (sync_particles_3d.c:324-345): unlike the 2-D twin it also blanks the position of every
// slot that is dead already
__global__ void __launch_bounds__(256) k_sync_particles_mark_3d(double *const *attrs, int nattrs, int iax, int iay,
                                                                int iaz, uint8_t *const *is_dead, const int64_t *npart,
                                                                const double *bounds) {
    const int p = blockIdx.y;
    const long n = npart[p];
    const double *b = bounds + 6 * p;
    double *x = attrs[(long)p * nattrs + iax], *y = attrs[(long)p * nattrs + iay], *z = attrs[(long)p * nattrs + iaz];
    uint8_t *dead = is_dead[p];
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    for (long ip = (long)blockIdx.x * blockDim.x + threadIdx.x; ip < n; ip += (long)gridDim.x * blockDim.x) {
        const bool out = x[ip] < b[0] || x[ip] > b[1] || y[ip] < b[2] || y[ip] > b[3] || z[ip] < b[4] || z[ip] > b[5];
        if (dead[ip] || out) {
            dead[ip] = 1;
            x[ip] = nan; y[ip] = nan; z[ip] = nan;
        }
    }
}

static unsigned sp3_blocks(int64_t max_npart) {
    long nb = (max_npart + 255) / 256;
    return (unsigned)(nb < 1 ? 1 : (nb > 1024 ? 1024 : nb));
}

extern "C" int lpa_sync_particles_count_3d(const double *const *xyz, const uint8_t *const *is_dead,
                                           const int64_t *npart, const double *bounds, int32_t npatches,
                                           int64_t max_npart, int64_t *npart_outgoing, int64_t *ndead, void *stream) {
    LPA_REQUIRE(xyz && is_dead && npart && bounds && npatches >= 0 && max_npart >= 0 && npart_outgoing && ndead,
                "lpa_sync_particles_count_3d: bad args");
    if (npatches == 0) return LPA_OK;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(npart_outgoing, 0, 8 * (size_t)npatches * NB3, st) != hipSuccess ||
        hipMemsetAsync(ndead, 0, 8 * (size_t)npatches, st) != hipSuccess) {
        lpa_set_error("lpa_sync_particles_count_3d: memset failed");
        return LPA_ERR_HIP;
    }
    if (max_npart == 0) return LPA_OK;
    hipLaunchKernelGGL(k_sync_particles_count_3d, dim3(sp3_blocks(max_npart), npatches), dim3(256), 0, st,
                       XYZTab{xyz, 3, 0, 1, 2}, is_dead, npart, bounds, (unsigned long long *)npart_outgoing,
                       (unsigned long long *)ndead);
    LPA_CHECK_LAUNCH("lpa_sync_particles_count_3d");
    return LPA_OK;
}

extern "C" int lpa_sync_particles_fill_3d(double *const *attrs, int32_t nattrs, int32_t iattr_x, int32_t iattr_y,
                                          int32_t iattr_z, uint8_t *const *is_dead, const int64_t *npart,
                                          const double *bounds, const int64_t *neighbor_ipatch,
                                          const int64_t *npart_incoming, const int64_t *npart_outgoing,
                                          int32_t npatches, int64_t max_npart, const double *global_min,
                                          const double *global_max, const double *cell, void *workspace,
                                          int64_t workspace_bytes, void *stream) {
    LPA_REQUIRE(attrs && nattrs >= 3 && iattr_x >= 0 && iattr_x < nattrs && iattr_y >= 0 && iattr_y < nattrs &&
                    iattr_z >= 0 && iattr_z < nattrs && iattr_x != iattr_y && iattr_x != iattr_z && iattr_y != iattr_z &&
                    is_dead && npart && bounds && neighbor_ipatch && npart_incoming && npart_outgoing && npatches >= 0 &&
                    max_npart >= 0 && global_min && global_max && cell && workspace,
                "lpa_sync_particles_fill_3d: bad args (attrs must contain x, y and z)");
    LPA_REQUIRE(cell[0] > 0 && cell[1] > 0 && cell[2] > 0, "lpa_sync_particles_fill_3d: cell sizes must be > 0");
    if (npatches == 0 || max_npart == 0) return LPA_OK;
    if (workspace_bytes < lpa_sync_particles_workspace_bytes(npatches, max_npart)) {
        lpa_set_error("lpa_sync_particles_fill_3d: workspace too small");
        return LPA_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    int32_t *list = (int32_t *)workspace, *drank = list + (size_t)npatches * max_npart;
    const XYZTab xyz{(const double *const *)attrs, nattrs, iattr_x, iattr_y, iattr_z};
    Glob3 gl;
    for (int a = 0; a < 3; a++) { gl.lo[a] = global_min[a]; gl.hi[a] = global_max[a]; gl.d[a] = cell[a]; }
    hipLaunchKernelGGL(k_sync_particles_rank_3d, dim3(npatches), dim3(256), 0, st, xyz, (const uint8_t *const *)is_dead,
                       npart, bounds, npart_outgoing, list, drank, (long)max_npart);
    LPA_CHECK_LAUNCH("lpa_sync_particles_fill_3d (rank)");
    hipLaunchKernelGGL(k_sync_particles_fill_3d, dim3(sp3_blocks(max_npart), npatches), dim3(256), 0, st, attrs, nattrs,
                       iattr_x, iattr_y, iattr_z, is_dead, npart, bounds, neighbor_ipatch, npart_incoming,
                       npart_outgoing, list, drank, (long)max_npart, gl, make_sides());
    LPA_CHECK_LAUNCH("lpa_sync_particles_fill_3d (fill)");
    hipLaunchKernelGGL(k_sync_particles_mark_3d, dim3(sp3_blocks(max_npart), npatches), dim3(256), 0, st, attrs, nattrs,
                       iattr_x, iattr_y, iattr_z, is_dead, npart, bounds);
    LPA_CHECK_LAUNCH("lpa_sync_particles_fill_3d (mark)");
    return LPA_OK;
}

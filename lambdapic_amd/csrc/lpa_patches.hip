// lpa_patches.hip -- kernel-level drop-ins that keep the reference's PATCH-LIST data model on the device:
//
//   lpa_sync_guard_fields_2d   core/patch/sync_fields2d.c:150-255   guard <- neighbour's interior edge
//   lpa_sync_currents_2d       core/patch/sync_fields2d.c:43-148    interior edge += neighbour's guard; guard = 0
//   lpa_bucket_sort            core/sort/cpu2d.c:9-54,78-189,220-303 (and cpu3d.c): bucket index with dead
//                              slots inheriting the previous particle's bucket, bucket bounds, in-place
//                              permutation restricted to the misplaced slots
//
// The resident engines do not use these (one slab per GPU has no intra-GPU patches, and the tiled kernels
// want the tile order of lpa_sort_tiles_*): they are what a lambdaPIC facade binds when it keeps its own
// patch lists and bucket bookkeeping (collisions read bucket_bound_min / max).  Arrays are in lambdaPIC's
// WRAPPED guard layout here (core/fields.py:24-27): index i in [0, n) interior, [n, n + ng) upper guard,
// [n + ng, n + 2 ng) = [-ng, 0) lower guard -- exactly what the reference's extensions take.
#include "lpa_common.hpp"

// wrapped index of logical node i in [-ng, n + ng)
__device__ __forceinline__ int widx(int i, int N) { return i < 0 ? i + N : i; }

// Boundary2D order (core/patch/sync_fields2d.c:19-29): side (x, y) of each of the 8 neighbours
__constant__ int c_side[8][2] = {{-1, 0}, {1, 0}, {0, -1}, {0, 1}, {-1, -1}, {1, -1}, {-1, 1}, {1, 1}};

__device__ __forceinline__ int boundary_of(int sx, int sy) {
    // inverse of c_side; (0, 0) -> -1
    if (sy == 0) return sx < 0 ? 0 : (sx > 0 ? 1 : -1);
    if (sx == 0) return sy < 0 ? 2 : 3;
    return 4 + (sx > 0 ? 1 : 0) + (sy > 0 ? 2 : 0);
}

// ---- guard copy: one thread per (array, padded cell); a guard cell on side (sx, sy) of patch p takes the
// neighbour's interior cell one patch width away (sync_fields2d.c:191-247).  Sources are interior cells,
// destinations guard cells: no ordering between threads is needed.
__global__ void __launch_bounds__(256) k_sync_guard_patches_2d(double *const *__restrict__ arrays, int ncomp,
                                                               const int64_t *__restrict__ neighbor, int nx,
                                                               int ny, int ng) {
    const int NX = nx + 2 * ng, NY = ny + 2 * ng;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= NX * NY) return;
    const int p = blockIdx.y / ncomp, c = blockIdx.y % ncomp;
    const int li = t / NY - ng, lj = t % NY - ng;               // logical node in [-ng, n + ng)
    const int sx = li < 0 ? -1 : (li >= nx ? 1 : 0), sy = lj < 0 ? -1 : (lj >= ny ? 1 : 0);
    const int b = boundary_of(sx, sy);
    if (b < 0) return;
    const long q = neighbor[(long)p * 8 + b];
    if (q < 0) return;
    const double *src = arrays[q * ncomp + c];
    double *dst = arrays[(long)p * ncomp + c];
    dst[widx(li, NX) * NY + widx(lj, NY)] = src[(li - sx * nx) * NY + (lj - sy * ny)];
}

// ---- current fold: one thread per (array, interior cell within ng of an edge); adds the neighbours' guard
// cells in the reference's boundary order (faces, then corners: sync_fields2d.c:84-144), so the sum is
// bit-identical to the CPU sweep.
__global__ void __launch_bounds__(256) k_sync_currents_patches_2d(double *const *__restrict__ arrays,
                                                                  const int64_t *__restrict__ neighbor, int nx,
                                                                  int ny, int ng) {
    const int NX = nx + 2 * ng, NY = ny + 2 * ng;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nx * ny) return;
    const int p = blockIdx.y / 4, c = blockIdx.y % 4;
    const int i = t / ny, j = t % ny;
    // the edge strips this cell lies in: low strip [0, ng) receives the low neighbour's UPPER guard [n, n + ng),
    // high strip [n - ng, n) the high neighbour's LOWER guard [-ng, 0)
    const bool xl = i < ng, xh = i >= nx - ng, yl = j < ng, yh = j >= ny - ng;
    if (!(xl || xh || yl || yh)) return;
    double *dst = arrays[(long)p * 4 + c];
    double v = dst[i * NY + j];
    for (int b = 0; b < 8; b++) {
        const int sx = c_side[b][0], sy = c_side[b][1];
        if ((sx < 0 && !xl) || (sx > 0 && !xh) || (sy < 0 && !yl) || (sy > 0 && !yh)) continue;
        const long q = neighbor[(long)p * 8 + b];
        if (q < 0) continue;
        const int si = i - sx * nx, sj = j - sy * ny;          // the same node in the neighbour's frame
        v += arrays[q * 4 + c][widx(si, NX) * NY + widx(sj, NY)];
    }
    dst[i * NY + j] = v;
}

// a guard region on side (sx, sy) was consumed iff the patch has a neighbour there
__global__ void __launch_bounds__(256) k_zero_consumed_guards_2d(double *const *__restrict__ arrays,
                                                                 const int64_t *__restrict__ neighbor, int nx,
                                                                 int ny, int ng) {
    const int NX = nx + 2 * ng, NY = ny + 2 * ng;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= NX * NY) return;
    const int p = blockIdx.y / 4, c = blockIdx.y % 4;
    const int li = t / NY - ng, lj = t % NY - ng;
    const int b = boundary_of(li < 0 ? -1 : (li >= nx ? 1 : 0), lj < 0 ? -1 : (lj >= ny ? 1 : 0));
    if (b < 0 || neighbor[(long)p * 8 + b] < 0) return;
    arrays[(long)p * 4 + c][widx(li, NX) * NY + widx(lj, NY)] = 0.0;
}

extern "C" int lpa_sync_guard_fields_2d(double *const *arrays, int32_t ncomp, const int64_t *neighbor_ipatch,
                                        int32_t npatches, int32_t nx, int32_t ny, int32_t ng, void *stream) {
    LPA_REQUIRE(arrays && neighbor_ipatch && ncomp >= 1 && npatches >= 0 && nx >= ng && ny >= ng && ng >= 1,
                "lpa_sync_guard_fields_2d: bad args (patches must be at least n_guard cells wide)");
    if (npatches == 0) return LPA_OK;
    const int cells = (nx + 2 * ng) * (ny + 2 * ng);
    hipLaunchKernelGGL(k_sync_guard_patches_2d, dim3((cells + 255) / 256, npatches * ncomp), dim3(256), 0,
                       (hipStream_t)stream, arrays, ncomp, neighbor_ipatch, nx, ny, ng);
    LPA_CHECK_LAUNCH("lpa_sync_guard_fields_2d");
    return LPA_OK;
}

extern "C" int lpa_sync_currents_2d(double *const *arrays, const int64_t *neighbor_ipatch, int32_t npatches,
                                    int32_t nx, int32_t ny, int32_t ng, void *stream) {
    LPA_REQUIRE(arrays && neighbor_ipatch && npatches >= 0 && nx >= 2 * ng && ny >= 2 * ng && ng >= 1,
                "lpa_sync_currents_2d: bad args (patches must be at least 2 n_guard cells wide)");
    if (npatches == 0) return LPA_OK;
    hipLaunchKernelGGL(k_sync_currents_patches_2d, dim3((nx * ny + 255) / 256, npatches * 4), dim3(256), 0,
                       (hipStream_t)stream, arrays, neighbor_ipatch, nx, ny, ng);
    LPA_CHECK_LAUNCH("lpa_sync_currents_2d (fold)");
    const int cells = (nx + 2 * ng) * (ny + 2 * ng);
    hipLaunchKernelGGL(k_zero_consumed_guards_2d, dim3((cells + 255) / 256, npatches * 4), dim3(256), 0,
                       (hipStream_t)stream, arrays, neighbor_ipatch, nx, ny, ng);
    LPA_CHECK_LAUNCH("lpa_sync_currents_2d (zero)");
    return LPA_OK;
}

// =====================================================================================================
// bucket sort with the reference's bookkeeping (one patch per call)
// =====================================================================================================
struct BucketGeom {
    long nx, ny, nz;      // buckets per axis (nz = 1 in 2-D)
    double dx, dy, dz, x0, y0, z0;
    int reverse_x;
};

constexpr int BS_CHUNK = 4096;   // slots per workgroup of the inheritance pass (256 threads x 16)

// bucket of a LIVE particle (core/sort/cpu2d.c:20-43, cpu3d.c:22-47): floor((r - r0) / d) per axis; out of
// range -> last bucket (or clamped when the x order is mirrored)
__device__ __forceinline__ long bucket_of(double x, double y, double z, const BucketGeom &g) {
    // a live slot with a non-finite position: the reference's cast of NaN / inf to a C integer yields the most
    // negative value on x86-64 (cvttsd2si), i.e. out of range -> last bucket (cpu2d.c:38-42), or index 0 of that axis
    // after the clamp of the mirrored order (:25-32); the device conversion would give 0 / saturate -- made explicit
    auto cell = [](double r, double r0, double d) {
        const double c = floor((r - r0) / d);
        return isfinite(c) ? (long)c : (long)0x8000000000000000ull;
    };
    long ix = cell(x, g.x0, g.dx), iy = cell(y, g.y0, g.dy);
    long iz = g.nz > 1 ? cell(z, g.z0, g.dz) : 0;
    if (g.reverse_x) {
        ix = ix < 0 ? 0 : (ix >= g.nx ? g.nx - 1 : ix);
        iy = iy < 0 ? 0 : (iy >= g.ny ? g.ny - 1 : iy);
        iz = iz < 0 ? 0 : (iz >= g.nz ? g.nz - 1 : iz);
        return iz + iy * g.nz + (g.nx - 1 - ix) * g.ny * g.nz;
    }
    if (ix < 0 || ix >= g.nx || iy < 0 || iy >= g.ny || iz < 0 || iz >= g.nz) return g.nx * g.ny * g.nz - 1;
    return iz + iy * g.nz + ix * g.ny * g.nz;
}

// pass 1: key of every live slot (-1 for dead ones) and, per chunk, the key of its LAST live slot
__global__ void __launch_bounds__(256) k_bucket_key(const double *__restrict__ x, const double *__restrict__ y,
                                                    const double *__restrict__ z,
                                                    const uint8_t *__restrict__ dead, long n, BucketGeom g,
                                                    int64_t *key, int64_t *chunk_last) {
    __shared__ long s_last[256];
    const long base = (long)blockIdx.x * BS_CHUNK;
    long last = -1;   // slots of a thread are consecutive: thread t owns [base + 16 t, base + 16 t + 16)
    for (int k = 0; k < BS_CHUNK / 256; k++) {
        const long ip = base + (long)threadIdx.x * (BS_CHUNK / 256) + k;
        if (ip >= n) break;
        long b = -1;
        if (!dead[ip]) { b = bucket_of(x[ip], y[ip], z ? z[ip] : 0.0, g); last = b; }
        key[ip] = b;
    }
    s_last[threadIdx.x] = last;
    __syncthreads();
    if (threadIdx.x == 0) {
        long l = -1;
        for (int t = 255; t >= 0; t--) if (s_last[t] >= 0) { l = s_last[t]; break; }
        chunk_last[blockIdx.x] = l;
    }
}

// pass 2 (one thread): what a dead slot at the START of every chunk inherits -- the reference's running
// `icell`, which starts at 0 (cpu2d.c:18)
__global__ void k_bucket_carry(const int64_t *chunk_last, int64_t *chunk_carry, long nchunks) {
    long run = 0;
    for (long c = 0; c < nchunks; c++) {
        chunk_carry[c] = run;
        if (chunk_last[c] >= 0) run = chunk_last[c];
    }
}

// pass 3: dead slots inherit (cpu2d.c:44-52), histogram
__global__ void __launch_bounds__(256) k_bucket_inherit_count(int64_t *key, const int64_t *chunk_carry, long n,
                                                              unsigned long long *bucket_count) {
    __shared__ long s_last[256];
    const long base = (long)blockIdx.x * BS_CHUNK;
    constexpr int PT = BS_CHUNK / 256;
    const long first = base + (long)threadIdx.x * PT;
    long last = -1;
    for (int k = 0; k < PT && first + k < n; k++) if (key[first + k] >= 0) last = key[first + k];
    s_last[threadIdx.x] = last;
    __syncthreads();
    long run = -1;    // key of the last live slot before this thread's range, inside the chunk
    for (int t = (int)threadIdx.x - 1; t >= 0; t--) if (s_last[t] >= 0) { run = s_last[t]; break; }
    if (run < 0) run = chunk_carry[blockIdx.x];
    for (int k = 0; k < PT && first + k < n; k++) {
        long b = key[first + k];
        if (b < 0) { b = run; key[first + k] = b; } else run = b;
        atomicAdd(&bucket_count[b], 1ull);
    }
}

// bounds (cpu2d.c:78-91): one workgroup, chunked scan
__global__ void __launch_bounds__(1024) k_bucket_bounds(const int64_t *count, int64_t *bmin, int64_t *bmax, long nbin) {
    __shared__ long s_part[1024];
    const int tid = threadIdx.x;
    const long per = (nbin + 1023) / 1024, lo = tid * per, hi = lo + per < nbin ? lo + per : nbin;
    long sum = 0;
    for (long b = lo; b < hi; b++) sum += count[b];
    s_part[tid] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        long a = tid >= o ? s_part[tid - o] : 0;
        __syncthreads();
        s_part[tid] += a;
        __syncthreads();
    }
    long off = s_part[tid] - sum;
    for (long b = lo; b < hi; b++) {
        bmin[b] = off;
        off += count[b];
        bmax[b] = off;
    }
}

// bucket whose slot range [bmin, bmax) holds slot ip: last b with bmin[b] <= ip among the non-empty ones --
// particle_index_ref of the reference (cpu2d.c:118-127)
__device__ __forceinline__ long ref_bucket(const int64_t *bmax, long nbin, long ip) {
    long lo = 0, hi = nbin - 1;       // first b with bmax[b] > ip
    while (lo < hi) {
        long mid = (lo + hi) >> 1;
        if (bmax[mid] > ip) hi = mid; else lo = mid + 1;
    }
    return lo;
}

// misplaced slots: the slot is a HOLE of the bucket that owns it, its particle a MOVER of its own bucket; both
// lists are kept inside the owning bucket's own slot range (a bucket has as many holes as movers, and no more
// of either than it has slots)
__global__ void __launch_bounds__(256) k_bucket_mismatch(const int64_t *__restrict__ key,
                                                         const int64_t *__restrict__ bmin,
                                                         const int64_t *__restrict__ bmax, long nbin, long n,
                                                         unsigned long long *hole_cnt, unsigned long long *mov_cnt,
                                                         int64_t *holes, int64_t *movers, unsigned long long *nbuf) {
    const long ip = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (ip >= n) return;
    const long b = key[ip], r = ref_bucket(bmax, nbin, ip);
    if (b == r) return;
    holes[bmin[r] + (long)atomicAdd(&hole_cnt[r], 1ull)] = ip;
    movers[bmin[b] + (long)atomicAdd(&mov_cnt[b], 1ull)] = ip;
    atomicAdd(nbuf, 1ull);
}

// pair the k-th mover of a bucket with its k-th hole: list position j = bmin[b] + k
template <class T>
__global__ void __launch_bounds__(256) k_bucket_gather(const T *__restrict__ attr, T *__restrict__ buf,
                                                       const int64_t *__restrict__ movers,
                                                       const int64_t *__restrict__ bmin,
                                                       const int64_t *__restrict__ bmax, long nbin,
                                                       const unsigned long long *__restrict__ hole_cnt, long n) {
    const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const long b = ref_bucket(bmax, nbin, j);
    if (j - bmin[b] < (long)hole_cnt[b]) buf[j] = attr[movers[j]];
}

template <class T>
__global__ void __launch_bounds__(256) k_bucket_fill(T *__restrict__ attr, const T *__restrict__ buf,
                                                     const int64_t *__restrict__ holes,
                                                     const int64_t *__restrict__ bmin,
                                                     const int64_t *__restrict__ bmax, long nbin,
                                                     const unsigned long long *__restrict__ hole_cnt, long n) {
    const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const long b = ref_bucket(bmax, nbin, j);
    if (j - bmin[b] < (long)hole_cnt[b]) attr[holes[j]] = buf[j];
}

struct BucketWs {
    int64_t *key, *holes, *movers, *chunk_last, *chunk_carry;
    unsigned long long *hole_cnt, *mov_cnt, *nbuf;
    double *buf;
};

static int64_t bucket_ws_layout(int64_t npart, int64_t nbin, char *base, BucketWs *w) {
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = (off + bytes + 255) & ~(size_t)255; return base ? base + o : nullptr; };
    const int64_t nchunks = (npart + BS_CHUNK - 1) / BS_CHUNK + 1;
    char *p;
    p = take(8 * (size_t)npart); if (w) w->key = (int64_t *)p;
    p = take(8 * (size_t)npart); if (w) w->holes = (int64_t *)p;
    p = take(8 * (size_t)npart); if (w) w->movers = (int64_t *)p;
    p = take(8 * (size_t)npart); if (w) w->buf = (double *)p;
    p = take(8 * (size_t)nchunks); if (w) w->chunk_last = (int64_t *)p;
    p = take(8 * (size_t)nchunks); if (w) w->chunk_carry = (int64_t *)p;
    p = take(8 * (size_t)nbin); if (w) w->hole_cnt = (unsigned long long *)p;
    p = take(8 * (size_t)nbin); if (w) w->mov_cnt = (unsigned long long *)p;
    p = take(8); if (w) w->nbuf = (unsigned long long *)p;
    return (int64_t)off;
}

extern "C" int64_t lpa_bucket_sort_workspace_bytes(int64_t npart, int64_t nbuckets) {
    if (npart < 0 || nbuckets <= 0) return -1;
    return bucket_ws_layout(npart, nbuckets, nullptr, nullptr);
}

extern "C" int lpa_bucket_sort(double *x, double *y, double *z, uint8_t *is_dead, double *const *attrs,
                               int32_t nattrs, int64_t npart, int64_t nx, int64_t ny, int64_t nz, double dx,
                               double dy, double dz, double x0, double y0, double z0, int32_t reverse_x,
                               int64_t *bucket_count, int64_t *bucket_bound_min, int64_t *bucket_bound_max,
                               void *workspace, int64_t workspace_bytes, int64_t *nbuf, void *stream) {
    LPA_REQUIRE(npart >= 0 && nx > 0 && ny > 0 && nz > 0 && dx > 0 && dy > 0 && (nz == 1 || (z && dz > 0)) &&
                    nattrs >= 0 && nattrs <= 32 && (nattrs == 0 || attrs) && bucket_count && bucket_bound_min &&
                    bucket_bound_max && workspace && nbuf && (npart == 0 || (x && y && is_dead)),
                "lpa_bucket_sort: bad args");
    // validated before anything is launched: a null entry found half way would leave the attributes it follows
    // permuted and the rest (and is_dead) not
    LPA_REQUIRE(nattrs == 0 || attrs, "lpa_bucket_sort: null attribute list");
    for (int a = 0; a < nattrs; a++) LPA_REQUIRE(attrs[a], "lpa_bucket_sort: null attribute %d", a);
    const int64_t nbin = nx * ny * nz;
    BucketWs w;
    const int64_t need = bucket_ws_layout(npart, nbin, (char *)workspace, &w);
    if (need > workspace_bytes) {
        lpa_set_error("lpa_bucket_sort: workspace %lld B < %lld B", (long long)workspace_bytes, (long long)need);
        return LPA_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(bucket_count, 0, 8 * (size_t)nbin, st) != hipSuccess ||
        hipMemsetAsync(w.hole_cnt, 0, 8 * (size_t)nbin, st) != hipSuccess ||
        hipMemsetAsync(w.mov_cnt, 0, 8 * (size_t)nbin, st) != hipSuccess ||
        hipMemsetAsync(w.nbuf, 0, 8, st) != hipSuccess) {
        lpa_set_error("lpa_bucket_sort: memset failed");
        return LPA_ERR_HIP;
    }
    BucketGeom g{(long)nx, (long)ny, (long)nz, dx, dy, nz > 1 ? dz : 1.0, x0, y0, nz > 1 ? z0 : 0.0, reverse_x};
    if (npart > 0) {
        const unsigned nchunks = (unsigned)((npart + BS_CHUNK - 1) / BS_CHUNK), nb = (unsigned)((npart + 255) / 256);
        hipLaunchKernelGGL(k_bucket_key, dim3(nchunks), dim3(256), 0, st, x, y, nz > 1 ? z : nullptr, is_dead,
                           (long)npart, g, w.key, w.chunk_last);
        LPA_CHECK_LAUNCH("k_bucket_key");
        hipLaunchKernelGGL(k_bucket_carry, dim3(1), dim3(1), 0, st, w.chunk_last, w.chunk_carry, (long)nchunks);
        LPA_CHECK_LAUNCH("k_bucket_carry");
        hipLaunchKernelGGL(k_bucket_inherit_count, dim3(nchunks), dim3(256), 0, st, w.key, w.chunk_carry, (long)npart,
                           (unsigned long long *)bucket_count);
        LPA_CHECK_LAUNCH("k_bucket_inherit_count");
        hipLaunchKernelGGL(k_bucket_bounds, dim3(1), dim3(1024), 0, st, bucket_count, bucket_bound_min,
                           bucket_bound_max, (long)nbin);
        LPA_CHECK_LAUNCH("k_bucket_bounds");
        hipLaunchKernelGGL(k_bucket_mismatch, dim3(nb), dim3(256), 0, st, w.key, bucket_bound_min, bucket_bound_max,
                           (long)nbin, (long)npart, w.hole_cnt, w.mov_cnt, w.holes, w.movers, w.nbuf);
        LPA_CHECK_LAUNCH("k_bucket_mismatch");
        // every attribute (x, y, z included by the caller's list or not: they move too), then is_dead
        auto move = [&](double *a) {
            hipLaunchKernelGGL(k_bucket_gather<double>, dim3(nb), dim3(256), 0, st, a, w.buf, w.movers,
                               bucket_bound_min, bucket_bound_max, (long)nbin, w.hole_cnt, (long)npart);
            hipLaunchKernelGGL(k_bucket_fill<double>, dim3(nb), dim3(256), 0, st, a, w.buf, w.holes,
                               bucket_bound_min, bucket_bound_max, (long)nbin, w.hole_cnt, (long)npart);
        };
        bool has_x = false, has_y = false, has_z = false;
        // `attrs` is a HOST array of device pointers (like the reference's attrs list)
        for (int a = 0; a < nattrs; a++) {
            has_x = has_x || attrs[a] == x; has_y = has_y || attrs[a] == y; has_z = has_z || attrs[a] == z;
            move(attrs[a]);
        }
        if (!has_x) move(x);
        if (!has_y) move(y);
        if (nz > 1 && !has_z) move(z);
        hipLaunchKernelGGL(k_bucket_gather<uint8_t>, dim3(nb), dim3(256), 0, st, is_dead, (uint8_t *)w.buf, w.movers,
                           bucket_bound_min, bucket_bound_max, (long)nbin, w.hole_cnt, (long)npart);
        hipLaunchKernelGGL(k_bucket_fill<uint8_t>, dim3(nb), dim3(256), 0, st, is_dead, (const uint8_t *)w.buf,
                           w.holes, bucket_bound_min, bucket_bound_max, (long)nbin, w.hole_cnt, (long)npart);
        LPA_CHECK_LAUNCH("k_bucket_gather / k_bucket_fill");
    } else {
        hipLaunchKernelGGL(k_bucket_bounds, dim3(1), dim3(1024), 0, st, bucket_count, bucket_bound_min,
                           bucket_bound_max, (long)nbin);
        LPA_CHECK_LAUNCH("k_bucket_bounds");
    }
    if (hipMemcpyAsync(nbuf, w.nbuf, 8, hipMemcpyDeviceToDevice, st) != hipSuccess) {
        lpa_set_error("lpa_bucket_sort: copy of nbuf failed");
        return LPA_ERR_HIP;
    }
    return LPA_OK;
}

// =====================================================================================================
// particle ownership between the patches of a list (core/patch/sync_particles_2d.c:37-518):
//   lpa_sync_particles_count_2d   count_outgoing_particles of get_npart_to_extend_2d (:37-84,258-283): the leavers of
//                                 every patch per boundary (Boundary2D order) + its dead slots
//   lpa_sync_particles_fill_2d    fill_particles_from_boundary_2d (:322-518): the leavers of the 8 neighbours, in
//                                 (boundary, index) order, go into the receiver's dead slots in ascending order
//                                 (periodic +- L on x / y like handle_periodic), then everything outside a patch's
//                                 bounds dies (x = y = NaN).  The placement is the reference's own, slot for slot.
// Patch arrays: device tables of per-patch device pointers; `bounds` = [npatches][4] xmin xmax ymin ymax WITH the half
// cell the reference adds (:236-241).
// =====================================================================================================
constexpr int SP_DEAD = 8, SP_STAY = 9;

// x / y of patch p inside a table of per-patch array pointers: tab[p * stride + ix], tab[p * stride + iy]
struct XYTab {
    const double *const *tab;
    int stride, ix, iy;
    __device__ __forceinline__ const double *x(int p) const { return tab[(long)p * stride + ix]; }
    __device__ __forceinline__ const double *y(int p) const { return tab[(long)p * stride + iy]; }
};

// the reference's classification (count_outgoing_particles, sync_particles_2d.c:37-84)
__device__ __forceinline__ int sp_class(double x, double y, bool dead, const double *b) {
    if (dead) return SP_DEAD;
    const double xmin = b[0], xmax = b[1], ymin = b[2], ymax = b[3];
    if (y < ymin) return x < xmin ? 4 : (x > xmax ? 5 : 2);
    if (y > ymax) return x < xmin ? 6 : (x > xmax ? 7 : 3);
    return x < xmin ? 0 : (x > xmax ? 1 : SP_STAY);
}

__global__ void __launch_bounds__(256) k_sync_particles_count(XYTab xy, const uint8_t *const *is_dead,
                                                              const int64_t *npart, const double *bounds,
                                                              unsigned long long *nout, unsigned long long *ndead) {
    const int p = blockIdx.y;
    const long n = npart[p];
    const double *x = xy.x(p), *y = xy.y(p);
    const uint8_t *dead = is_dead[p];
    __shared__ unsigned int s_cnt[9];
    if (threadIdx.x < 9) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    for (long ip = (long)blockIdx.x * blockDim.x + threadIdx.x; ip < n; ip += (long)gridDim.x * blockDim.x) {
        const int c = sp_class(x[ip], y[ip], dead[ip] != 0, bounds + 4 * p);
        if (c != SP_STAY) atomicAdd(&s_cnt[c], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 8 && s_cnt[threadIdx.x])
        atomicAdd(&nout[(long)p * 8 + threadIdx.x], (unsigned long long)s_cnt[threadIdx.x]);
    if (threadIdx.x == 8 && s_cnt[8]) atomicAdd(&ndead[p], (unsigned long long)s_cnt[8]);
}

// one workgroup per patch: stable rank of every leaver inside its class and of every dead slot among the dead
// (ascending index, like the reference's sequential loops); leavers are listed class by class
__global__ void __launch_bounds__(256) k_sync_particles_rank(XYTab xy, const uint8_t *const *is_dead,
                                                             const int64_t *npart, const double *bounds,
                                                             const int64_t *nout, int32_t *list, int32_t *drank,
                                                             long stride) {
    const int p = blockIdx.x;
    const long n = npart[p];
    const double *x = xy.x(p), *y = xy.y(p);
    const uint8_t *dead = is_dead[p];
    __shared__ int s_base[9], s_w[4][9], s_cbase[8];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x < 9) s_base[threadIdx.x] = 0;
    if (threadIdx.x == 0) {
        int run = 0;
        for (int c = 0; c < 8; c++) { s_cbase[c] = run; run += (int)nout[(long)p * 8 + c]; }
    }
    __syncthreads();
    for (long c0 = 0; c0 < n; c0 += 256) {
        const long ip = c0 + threadIdx.x;
        const int cls = ip < n ? sp_class(x[ip], y[ip], dead[ip] != 0, bounds + 4 * p) : SP_STAY;
        int pre = 0;
        for (int c = 0; c < 9; c++) {          // wave-uniform loop: rank inside the wave, per class
            const unsigned long long m = __ballot(cls == c);
            if (cls == c) pre = __popcll(m & ((1ull << lane) - 1ull));
            if (lane == 0) s_w[wv][c] = __popcll(m);
        }
        __syncthreads();
        if (cls != SP_STAY) {
            int r = s_base[cls] + pre;
            for (int w = 0; w < wv; w++) r += s_w[w][cls];
            if (cls == SP_DEAD) drank[(long)p * stride + ip] = r;
            else list[(long)p * stride + s_cbase[cls] + r] = (int32_t)ip;
        }
        __syncthreads();
        if (threadIdx.x < 9)
            s_base[threadIdx.x] += s_w[0][threadIdx.x] + s_w[1][threadIdx.x] + s_w[2][threadIdx.x] + s_w[3][threadIdx.x];
        __syncthreads();
    }
}

__device__ __forceinline__ double sp_periodic(double v, double gmin, double gmax, double L, double pmin, double pmax,
                                              double cell) {   // handle_periodic, sync_particles_2d.c:168-182
    double out = v;
    if (v > gmax && fabs(pmin - gmin) < cell) out -= L;
    if (v < gmin && fabs(pmax - gmax) < cell) out += L;
    return out;
}

__global__ void __launch_bounds__(256) k_sync_particles_fill(double *const *attrs, int nattrs, int iax, int iay,
                                                             uint8_t *const *is_dead, const int64_t *npart,
                                                             const double *bounds, const int64_t *neighbor,
                                                             const int64_t *nin, const int64_t *nout,
                                                             const int32_t *list, const int32_t *drank, long stride,
                                                             double gx0, double gx1, double gy0, double gy1,
                                                             double dx, double dy) {
    const int p = blockIdx.y;
    const long n = npart[p], new_n = nin[p];
    const int opp[8] = {1, 0, 3, 2, 7, 6, 5, 4};      // OPPOSITE_BOUNDARY, sync_particles_2d.c:26-35
    uint8_t *dead = is_dead[p];
    for (long ip = (long)blockIdx.x * blockDim.x + threadIdx.x; ip < n; ip += (long)gridDim.x * blockDim.x) {
        if (!dead[ip]) continue;
        long r = drank[(long)p * stride + ip];         // this is the r-th dead slot: it takes incoming particle r
        if (r >= new_n) continue;
        long q = -1, src = -1;
        for (int b = 0; b < 8 && q < 0; b++) {         // incoming order: boundary by boundary (:131-146)
            const long nb = neighbor[(long)p * 8 + b];
            if (nb < 0) continue;
            const long cnt = nout[nb * 8 + opp[b]];
            if (r < cnt) {
                long base = 0;
                for (int c = 0; c < opp[b]; c++) base += nout[nb * 8 + c];
                q = nb;
                src = list[nb * stride + base + r];
            } else {
                r -= cnt;
            }
        }
        if (q < 0) continue;                            // (counts inconsistent with the arrays: nothing to take)
        const double *bp = bounds + 4 * p;
        for (int a = 0; a < nattrs; a++) {
            double v = attrs[q * nattrs + a][src];
            if (a == iax) v = sp_periodic(v, gx0, gx1, gx1 - gx0, bp[0], bp[1], dx);
            if (a == iay) v = sp_periodic(v, gy0, gy1, gy1 - gy0, bp[2], bp[3], dy);
            attrs[(long)p * nattrs + a][ip] = v;
        }
        dead[ip] = 0;
    }
}

// mark_out_of_bound_as_dead (:185-202)
__global__ void __launch_bounds__(256) k_sync_particles_mark(double *const *attrs, int nattrs, int iax, int iay,
                                                             uint8_t *const *is_dead, const int64_t *npart,
                                                             const double *bounds) {
    const int p = blockIdx.y;
    const long n = npart[p];
    const double *b = bounds + 4 * p;
    double *x = attrs[(long)p * nattrs + iax], *y = attrs[(long)p * nattrs + iay];
    uint8_t *dead = is_dead[p];
    for (long ip = (long)blockIdx.x * blockDim.x + threadIdx.x; ip < n; ip += (long)gridDim.x * blockDim.x) {
        if (dead[ip]) continue;
        if (x[ip] < b[0] || x[ip] > b[1] || y[ip] < b[2] || y[ip] > b[3]) {
            dead[ip] = 1;
            x[ip] = __longlong_as_double(0x7ff8000000000000ll);
            y[ip] = x[ip];
        }
    }
}

static unsigned sp_blocks(int64_t max_npart) {
    long nb = (max_npart + 255) / 256;
    return (unsigned)(nb < 1 ? 1 : (nb > 1024 ? 1024 : nb));
}

extern "C" int lpa_sync_particles_count_2d(const double *const *xy, const uint8_t *const *is_dead,
                                           const int64_t *npart, const double *bounds, int32_t npatches,
                                           int64_t max_npart, int64_t *npart_outgoing, int64_t *ndead, void *stream) {
    LPA_REQUIRE(xy && is_dead && npart && bounds && npatches >= 0 && max_npart >= 0 && npart_outgoing && ndead,
                "lpa_sync_particles_count_2d: bad args");
    if (npatches == 0) return LPA_OK;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(npart_outgoing, 0, 8 * (size_t)npatches * 8, st) != hipSuccess ||
        hipMemsetAsync(ndead, 0, 8 * (size_t)npatches, st) != hipSuccess) {
        lpa_set_error("lpa_sync_particles_count_2d: memset failed");
        return LPA_ERR_HIP;
    }
    if (max_npart == 0) return LPA_OK;
    hipLaunchKernelGGL(k_sync_particles_count, dim3(sp_blocks(max_npart), npatches), dim3(256), 0, st,
                       XYTab{xy, 2, 0, 1}, is_dead, npart, bounds, (unsigned long long *)npart_outgoing,
                       (unsigned long long *)ndead);
    LPA_CHECK_LAUNCH("lpa_sync_particles_count_2d");
    return LPA_OK;
}

extern "C" int64_t lpa_sync_particles_workspace_bytes(int32_t npatches, int64_t max_npart) {
    if (npatches < 0 || max_npart < 0) return -1;
    return 2 * (int64_t)sizeof(int32_t) * (npatches > 0 ? npatches : 1) * (max_npart > 0 ? max_npart : 1);
}

extern "C" int lpa_sync_particles_fill_2d(double *const *attrs, int32_t nattrs, int32_t iattr_x, int32_t iattr_y,
                                          uint8_t *const *is_dead, const int64_t *npart, const double *bounds,
                                          const int64_t *neighbor_ipatch, const int64_t *npart_incoming,
                                          const int64_t *npart_outgoing, int32_t npatches, int64_t max_npart,
                                          double xmin_global, double xmax_global, double ymin_global,
                                          double ymax_global, double dx, double dy, void *workspace,
                                          int64_t workspace_bytes, void *stream) {
    LPA_REQUIRE(attrs && nattrs >= 2 && iattr_x >= 0 && iattr_x < nattrs && iattr_y >= 0 && iattr_y < nattrs &&
                    iattr_x != iattr_y && is_dead && npart && bounds && neighbor_ipatch && npart_incoming &&
                    npart_outgoing && npatches >= 0 && max_npart >= 0 && dx > 0 && dy > 0 && workspace,
                "lpa_sync_particles_fill_2d: bad args (attrs must contain x and y)");
    if (npatches == 0 || max_npart == 0) return LPA_OK;
    if (workspace_bytes < lpa_sync_particles_workspace_bytes(npatches, max_npart)) {
        lpa_set_error("lpa_sync_particles_fill_2d: workspace too small");
        return LPA_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    int32_t *list = (int32_t *)workspace, *drank = list + (size_t)npatches * max_npart;
    const XYTab xy{(const double *const *)attrs, nattrs, iattr_x, iattr_y};
    hipLaunchKernelGGL(k_sync_particles_rank, dim3(npatches), dim3(256), 0, st, xy, (const uint8_t *const *)is_dead,
                       npart, bounds, npart_outgoing, list, drank, (long)max_npart);
    LPA_CHECK_LAUNCH("lpa_sync_particles_fill_2d (rank)");
    hipLaunchKernelGGL(k_sync_particles_fill, dim3(sp_blocks(max_npart), npatches), dim3(256), 0, st, attrs, nattrs,
                       iattr_x, iattr_y, is_dead, npart, bounds, neighbor_ipatch, npart_incoming, npart_outgoing, list,
                       drank, (long)max_npart, xmin_global, xmax_global, ymin_global, ymax_global, dx, dy);
    LPA_CHECK_LAUNCH("lpa_sync_particles_fill_2d (fill)");
    hipLaunchKernelGGL(k_sync_particles_mark, dim3(sp_blocks(max_npart), npatches), dim3(256), 0, st, attrs, nattrs,
                       iattr_x, iattr_y, is_dead, npart, bounds);
    LPA_CHECK_LAUNCH("lpa_sync_particles_fill_2d (mark)");
    return LPA_OK;
}

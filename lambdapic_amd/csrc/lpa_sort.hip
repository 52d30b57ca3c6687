// lpa_sort.hip -- cell-index sort and slab migration of particles.
//
// The reference sorts every species every step by x-cell buckets in place and keeps dead particles
// inside the arrays (core/sort/cpu2d.c:9-54,108-189; policy core/sort/particle_sort.py:196-211).  The
// permutation inside a bucket is implementation defined there; what the rest of the step relies on is
// only locality.  On the GPU the sort is what makes the LDS-tiled kernel possible and fast:
//   key = (tile_x * tiles_y + tile_y) * 256 + (cell_x_in_tile * 32 + cell_y_in_tile)
//   with 8 x 32-cell tiles (y is the fastest grid axis) and nearest-node cells, so that a tile's
//   particles are one contiguous range (one LDS staging of E/B/J per work block).
// Inside a tile two orders are produced (LPA_ORDER_*):
//   CELL_MAJOR: cell by cell.  The 64 lanes of a wave share a cell: identical deposit windows, summed
//               across the wave in registers before touching LDS.
//   STRIPED   : rank by rank -- first the 0-th particle of every cell (cells in index order, y
//               fastest), then the 1-st of every cell that has one, ...  A half-wave (32 lanes) then
//               sits in 32 consecutive y-cells of one grid row: its LDS gather reads and its LDS
//               atomics go to 32 consecutive doubles = 32 different bank pairs, conflict free.
// Counting sort, out of place:
//   1. k_cell_count   : key + rank-in-cell of every live particle (atomics on the cell counters,
//                       wave-aggregated when lanes share a cell);
//   2. k_tile_sum / k_tile_scan : tile totals, their exclusive scan, the work-block table;
//   3a. CELL_MAJOR: k_cell_scan (per-tile scan of the 256 counters) + k_cell_scatter;
//   3b. STRIPED   : k_stripe_table (per tile and rank r < 128: 256-bit mask of the cells that have an
//                   r-th particle + running total) + k_stripe_scatter
//                   (slot = tile_off + total[r] + popcount(mask[r] below the cell)).
// Dead / NaN particles are dropped (compaction) -- they are the reference's recycled "dead slots".
#include "lpa_common.hpp"
#include "lpa_fold.hpp"

constexpr int TX = LPA_TILE_X, TY = LPA_TILE_Y;
constexpr int TCELLS = TX * TY;  // 256
// ranks that are striped (one 256-bit cell mask per tile and rank); deeper particles follow cell by cell, where all the
// lanes of a wave sit in ONE cell and every LDS atomic of the tiled kernels is a 64-way conflict (K1 2-D at 256 particles
// per cell with 128 striped ranks: 3.5 ms instead of 1.6).  Sized per workspace from the mean occupancy the store can
// hold: twice the particles per cell at full capacity, as a power of two in [32, 1024] -- C2 (64 per cell): 256 ranks,
// 34 MB of masks; a 3-D slab at 8 per cell: 32 ranks instead of 128.
// `requested` > 0 (lpa_tiling.stripe_ranks / lpa_sort_workspace_bytes_ranks): the caller knows better -- a store whose
// particles sit in a fraction of the grid (a solid target in an empty box) is far deeper where it is occupied than
// its mean over all tiles; rounded up to a power of two in the same range.
static int stripe_ranks(const lpa_grid *g, int64_t cap, int ntiles, int requested = 0) {
    (void)g;
    const int64_t per_cell = (cap + (int64_t)ntiles * 256 - 1) / ((int64_t)ntiles * 256);
    const int64_t want = requested > 0 ? requested : 2 * per_cell;
    // the default rule stops at 1024; a request may go on to 16384 as long as the mask table (32 bytes per tile and rank)
    // stays below 1 GiB
    const int64_t rcap = requested > 0 ? 16384 : 1024;
    int r = 32;
    while (r < want && r < rcap && (requested <= 0 || r < 1024 || (int64_t)ntiles * (2 * r) * 32 <= (1ll << 30))) r *= 2;
    return r;
}
static_assert(TCELLS == 256, "one workgroup thread per tile cell");

struct SortHdr {      // first 64 bytes of the workspace
    int32_t n_live;   // live particles after the sort
    int32_t n_blocks; // valid work blocks
    // the tiling the SOURCE of the running sort was left in by the previous sort through this
    // workspace (the engines ping-pong two stores): a work partition for the tile-staged scatter
    int32_t magic, prev_ntiles, prev_n, prev_valid;
    // set by k_tile_scan, read by the scatter kernels (which then move nothing) and by the caller after the sort:
    // bit 0 = the slots of the sorted order (live + the holes and 64-slot rounding of LPA_ORDER_PADDED) exceed the
    // destination's capacity; bit 1 = more work blocks than the table holds; bit 2 = the caller's prefix_hint exceeds
    // the tile-ordered prefix this header vouches for
    int32_t overflow;
    int32_t deepest;  // striped orders: the largest number of particles one cell holds (k_stripe_table) ...
    int32_t tail;     // ... and the particles that lie beyond the striped ranks (cell by cell behind their tile's stripes)
    int32_t tiles_in_use;   // tiles that hold at least one particle (k_tile_scan)
    int32_t pad[6];
};
constexpr int32_t SORT_OVF_SLOTS = 1, SORT_OVF_BLOCKS = 2, SORT_BAD_HINT = 4;
constexpr int32_t SORT_MAGIC = 0x4c504131;

struct SortWs {
    SortHdr *hdr;
    int32_t *cell_cnt, *cell_off, *cell_base, *tile_cnt, *tile_off, *tile_off_prev, *blk_tile, *blk_begin, *blk_end, *apre;
    int32_t *run_off;   // [ntiles + 1] k_scatter_tiled's work list: first run (ST_RUN chunks of the old tile) of each tile
    int32_t *pad_ranks;
    unsigned long long *masks;
    uint32_t *key, *rank;
    int rmax, ntiles, max_blocks;
};

static size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

constexpr int T3X = LPA_TILE3_X, T3Y = LPA_TILE3_Y, T3Z = LPA_TILE3_Z;
static_assert(T3X * T3Y * T3Z == TCELLS, "3-D tiles hold 256 cells like the 2-D ones");

// tiles of the grid: 2-D (nz <= 1) or 3-D
static int tile_count(const lpa_grid *g) {
    if (g->nz > 1)
        return ((g->nx + T3X - 1) / T3X) * ((g->ny + T3Y - 1) / T3Y) * ((g->nz + T3Z - 1) / T3Z);
    return ((g->nx + TX - 1) / TX) * ((g->ny + TY - 1) / TY);
}

static int64_t ws_layout(const lpa_grid *g, int64_t cap, int32_t block_particles, char *base, SortWs *w,
                         int ranks = 0) {
    int nt = tile_count(g);
    const int RMAX = stripe_ranks(g, cap, nt, ranks);
    // + 8: the tiled kernels launch max_blocks workgroups and deal work blocks to them in XCD order
    int64_t maxb = nt + cap / (block_particles > 0 ? block_particles : 4096) + 1 + 8;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes); return base ? base + o : nullptr; };
    char *p;
    p = take(sizeof(SortHdr)); if (w) w->hdr = (SortHdr *)p;
    p = take(sizeof(int32_t) * (size_t)nt * TCELLS); if (w) w->cell_cnt = (int32_t *)p;
    p = take(sizeof(int32_t) * (size_t)nt * TCELLS); if (w) w->cell_off = (int32_t *)p;
    p = take(sizeof(int32_t) * (size_t)nt * TCELLS); if (w) w->cell_base = (int32_t *)p;
    p = take(sizeof(int32_t) * nt); if (w) w->tile_cnt = (int32_t *)p;
    p = take(sizeof(int32_t) * nt); if (w) w->pad_ranks = (int32_t *)p;
    p = take(sizeof(int32_t) * (nt + 1)); if (w) w->tile_off = (int32_t *)p;
    p = take(sizeof(int32_t) * (nt + 1)); if (w) w->tile_off_prev = (int32_t *)p;
    p = take(sizeof(int32_t) * (nt + 1)); if (w) w->run_off = (int32_t *)p;
    p = take(sizeof(int32_t) * maxb); if (w) w->blk_tile = (int32_t *)p;
    p = take(sizeof(int32_t) * maxb); if (w) w->blk_begin = (int32_t *)p;
    p = take(sizeof(int32_t) * maxb); if (w) w->blk_end = (int32_t *)p;
    p = take(sizeof(unsigned long long) * (size_t)nt * RMAX * 4); if (w) w->masks = (unsigned long long *)p;
    p = take(sizeof(int32_t) * (size_t)nt * (RMAX + 1)); if (w) w->apre = (int32_t *)p;
    p = take(sizeof(uint32_t) * cap); if (w) w->key = (uint32_t *)p;
    p = take(sizeof(uint32_t) * cap); if (w) w->rank = (uint32_t *)p;
    if (w) { w->ntiles = nt; w->max_blocks = (int)maxb; w->rmax = RMAX; }
    return (int64_t)off;
}

extern "C" int64_t lpa_sort_workspace_bytes(const lpa_grid *g, int64_t capacity) {
    if (!g || g->nx <= 0 || g->ny <= 0 || capacity < 0) return -1;  // nz > 1 selects the 3-D tiles
    // sized for the smallest block size accepted by lpa_sort_tiles_2d
    return ws_layout(g, capacity, 1024, nullptr, nullptr);
}

extern "C" int64_t lpa_sort_workspace_bytes_ranks(const lpa_grid *g, int64_t capacity, int32_t stripe_ranks_) {
    if (!g || g->nx <= 0 || g->ny <= 0 || capacity < 0 || stripe_ranks_ < 0) return -1;
    return ws_layout(g, capacity, 1024, nullptr, nullptr, stripe_ranks_);
}

extern "C" const int32_t *lpa_sort_deepest_cell(void *workspace) {
    return workspace ? &((SortHdr *)workspace)->deepest : nullptr;
}

extern "C" int32_t lpa_sort_stripe_ranks(const lpa_grid *g, int64_t capacity) {
    if (!g || g->nx <= 0 || g->ny <= 0 || capacity < 0) return -1;
    return stripe_ranks(g, capacity, tile_count(g));
}

extern "C" const int32_t *lpa_sort_live_count(void *workspace) {
    return workspace ? &((SortHdr *)workspace)->n_live : nullptr;
}

extern "C" const int32_t *lpa_sort_overflow(void *workspace) {
    return workspace ? &((SortHdr *)workspace)->overflow : nullptr;
}

constexpr uint32_t KEY_DEAD = 0xFFFFFFFFu;

// rank inside the cell: lanes that share a cell elect a leader that reserves the whole group with one
// atomic; after 4 rounds the remaining lanes reserve their slots individually
__device__ __forceinline__ uint32_t cell_rank(bool live, uint32_t ck, int32_t *cell_cnt) {
    uint32_t r = 0;
    unsigned long long todo = __ballot(live);
    const int lane = threadIdx.x & 63;
    for (int round = 0; round < 4 && todo; round++) {
        int leader = __ffsll((long long)todo) - 1;
        uint32_t lk = __shfl(ck, leader, 64);
        unsigned long long grp = __ballot(live && ck == lk) & todo;
        uint32_t base = 0;
        if (lane == leader) base = (uint32_t)atomicAdd(&cell_cnt[lk], (int32_t)__popcll(grp));
        base = __shfl(base, leader, 64);
        if ((grp >> lane) & 1ull) r = base + (uint32_t)__popcll(grp & ((1ull << lane) - 1ull));
        todo &= ~grp;
    }
    if ((todo >> lane) & 1ull) r = (uint32_t)atomicAdd(&cell_cnt[ck], 1);
    return r;
}

// geometry of the key: nearest-node cell, clamped into the slab (a particle may sit up to one cell
// outside between the push and the migration); 2-D key = tile * 256 + lx * 32 + ly, 3-D key =
// tile * 256 + (lx * 4 + ly) * 16 + lz (z fastest, like the grid)
struct KeyGeom {
    int dim, nx, ny, nz, tiles_y, tiles_z;
    double x0, y0, z0, inv_dx, inv_dy, inv_dz;
    double ahead = 0.0;   // c x look-ahead time: bin by where the particle will be then (lpa_sort_tiles_ahead_*)
};

__device__ __forceinline__ uint32_t cell_key(const PartV &p, long ip, const KeyGeom &k) {
    double x = p.x[ip], y = p.y[ip];
    if ((p.dead && p.dead[ip]) || isnan(x) || isnan(y)) return KEY_DEAD;
    [[maybe_unused]] double zz = 0.0;
    if (k.ahead != 0.0) {   // (uniform) ballistic look-ahead: x + v t; 1 / gamma from the momenta (the store's may be stale)
        const double ux = p.ux[ip], uy = p.uy[ip], uz = p.uz[ip];
        const double s = k.ahead * rsqrt_nr(fma(uz, uz, fma(uy, uy, fma(ux, ux, 1.0))));
        x = fma(ux, s, x); y = fma(uy, s, y);
        zz = uz * s;
    }
    int is = ifloor((x - k.x0) * k.inv_dx + 0.5), js = ifloor((y - k.y0) * k.inv_dy + 0.5);
    is = is < 0 ? 0 : (is >= k.nx ? k.nx - 1 : is);
    js = js < 0 ? 0 : (js >= k.ny ? k.ny - 1 : js);
    if (k.dim == 2) {
        int tile = (is / TX) * k.tiles_y + js / TY;
        return (uint32_t)(tile * TCELLS + (is % TX) * TY + (js % TY));
    }
    double z = p.z[ip];
    if (isnan(z)) return KEY_DEAD;
    z += zz;
    int ks = ifloor((z - k.z0) * k.inv_dz + 0.5);
    ks = ks < 0 ? 0 : (ks >= k.nz ? k.nz - 1 : ks);
    int tile = ((is / T3X) * k.tiles_y + js / T3Y) * k.tiles_z + ks / T3Z;
    return (uint32_t)(tile * TCELLS + ((is % T3X) * T3Y + (js % T3Y)) * T3Z + (ks % T3Z));
}

// any particle order: key + rank by atomics on the global cell counters
__global__ void __launch_bounds__(256) k_cell_count(PartV p, KeyGeom k, const SortHdr *hdr, int32_t *cell_cnt,
                                                    uint32_t *key, uint32_t *rank, long first_block) {
    const long blk = first_block + blockIdx.x;     // (blocks below the caller's prefix_hint are not launched)
    long ip = blk * blockDim.x + threadIdx.x;
    if (hdr->prev_valid && (blk + 1) * blockDim.x <= hdr->prev_n) return;  // block-uniform
    bool mine = ip < p.n && !(hdr->prev_valid && ip < hdr->prev_n);   // the rest: k_cell_count_tiled
    uint32_t ck = mine ? cell_key(p, ip, k) : KEY_DEAD;
    uint32_t r = cell_rank(ck != KEY_DEAD, ck, cell_cnt);
    if (mine) {
        key[ip] = ck;
        rank[ip] = r;
    }
}

// Re-sorts: the source is tile ordered, so one workgroup walks one OLD tile and ranks the particles
// that stay in it with LDS counters (rank = RANK_LOCAL | local rank); only particles that changed tile
// use a global atomic.  At the end the tile's counts are added to the global counters in one atomic
// per cell, whose return value is the offset of the local ranks (cell_base) -- the scatter adds it.
constexpr uint32_t RANK_LOCAL = 0x80000000u;

__global__ void __launch_bounds__(512) k_cell_count_tiled(PartV p, KeyGeom k, const SortHdr *hdr,
                                                          const int32_t *__restrict__ tile_off_prev,
                                                          int32_t *cell_cnt, int32_t *cell_base,
                                                          uint32_t *key, uint32_t *rank) {
    __shared__ int32_t s_cnt[TCELLS];
    if (!hdr->prev_valid) return;
    const int t = blockIdx.x;
    if (threadIdx.x < TCELLS) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const int sb = tile_off_prev[t], se = tile_off_prev[t + 1];
    for (int ip = sb + (int)threadIdx.x; ip < se; ip += blockDim.x) {
        uint32_t ck = cell_key(p, ip, k);
        uint32_t r = 0;
        if (ck != KEY_DEAD) {
            if ((int)(ck >> 8) == t) r = RANK_LOCAL | (uint32_t)atomicAdd(&s_cnt[ck & 255], 1);
            else r = (uint32_t)atomicAdd(&cell_cnt[ck], 1);
        }
        key[ip] = ck;
        rank[ip] = r;
    }
    __syncthreads();
    if (threadIdx.x < TCELLS) {
        int n = s_cnt[threadIdx.x];
        long c = (long)t * TCELLS + threadIdx.x;
        cell_base[c] = n ? atomicAdd(&cell_cnt[c], n) : 0;
    }
}

// one workgroup per tile: total of its 256 cell counters
__global__ void __launch_bounds__(256) k_tile_sum(const int32_t *__restrict__ cell_cnt, int32_t *tile_cnt) {
    __shared__ int32_t red[4];
    int v = cell_cnt[(long)blockIdx.x * TCELLS + threadIdx.x];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) tile_cnt[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// one workgroup: exclusive scan of the tile totals + the work-block table
__global__ void __launch_bounds__(1024) k_tile_scan(int ntiles, const int32_t *tile_cnt, int32_t *tile_off,
                                                    int32_t *blk_tile, int32_t *blk_begin, int32_t *blk_end,
                                                    SortHdr *hdr, int block_particles, int max_blocks,
                                                    long dst_capacity) {
    __shared__ int32_t s_part[1024], s_blk[1024];
    int tid = threadIdx.x;
    int per = (ntiles + 1023) / 1024;
    int lo = tid * per, hi = min(lo + per, ntiles);
    int32_t sum = 0, sb = 0;
    for (int t = lo; t < hi; t++) {
        int32_t c = tile_cnt[t];
        sum += c;
        sb += (c + block_particles - 1) / block_particles;
    }
    s_part[tid] = sum;
    s_blk[tid] = sb;
    __syncthreads();
    // Hillis-Steele inclusive scan over the 1024 partials
    for (int o = 1; o < 1024; o <<= 1) {
        int32_t a = tid >= o ? s_part[tid - o] : 0, b = tid >= o ? s_blk[tid - o] : 0;
        __syncthreads();
        s_part[tid] += a;
        s_blk[tid] += b;
        __syncthreads();
    }
    int32_t off = s_part[tid] - sum, boff = s_blk[tid] - sb;
    int used = 0;
    for (int t = lo; t < hi; t++) {
        int32_t c = tile_cnt[t];
        used += c > 0;
        tile_off[t] = off;
        int nb = (c + block_particles - 1) / block_particles;
        for (int b = 0; b < nb && boff + b < max_blocks; b++) {
            // equal split of the tile's particles over its blocks, on 64-particle boundaries
            long s0 = ((long)c * b / nb) & ~63l, s1 = b + 1 == nb ? c : (((long)c * (b + 1) / nb) & ~63l);
            blk_tile[boff + b] = t;
            blk_begin[boff + b] = off + (int32_t)s0;
            blk_end[boff + b] = off + (int32_t)s1;
        }
        off += c;
        boff += nb;
    }
    if (used) atomicAdd(&hdr->tiles_in_use, used);
    if (tid == 1023) {
        tile_off[ntiles] = s_part[1023];
        hdr->n_live = s_part[1023];
        hdr->n_blocks = min(s_blk[1023], max_blocks);
        // a PADDED order stores up to 4/3 n + 63 slots per tile: it may not fit a destination sized for the live
        // count; nothing is scattered then (every slot beyond the capacity would be an out-of-bounds store)
        hdr->overflow = (hdr->overflow & SORT_BAD_HINT) | ((long)s_part[1023] > dst_capacity ? SORT_OVF_SLOTS : 0) |
                        (s_blk[1023] > max_blocks ? SORT_OVF_BLOCKS : 0);
    }
}

// exclusive scan of 256 per-thread values inside one workgroup; returns the prefix, *total gets the sum
__device__ __forceinline__ int block_excl_scan256(int c, int *total) {
    __shared__ int32_t wsum[4];
    int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int inc = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    __syncthreads();
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wv; w++) base += wsum[w];
    if (total) *total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    return base + inc - c;
}

// CELL_MAJOR: one workgroup per tile: exclusive scan of its 256 cell counters, offset by the tile's start
__global__ void __launch_bounds__(256) k_cell_scan(const int32_t *__restrict__ cell_cnt,
                                                   const int32_t *__restrict__ tile_off, int32_t *cell_off) {
    int c = cell_cnt[(long)blockIdx.x * TCELLS + threadIdx.x];
    int ex = block_excl_scan256(c, nullptr);
    cell_off[(long)blockIdx.x * TCELLS + threadIdx.x] = tile_off[blockIdx.x] + ex;
}

// STRIPED: one workgroup per tile, thread c = cell c.  For every rank r < RMAX: the 256-bit mask of the
// cells with more than r particles and the number of striped slots before rank r; cells deeper than
// RMAX keep their surplus cell by cell behind the striped part (cell_off = offset of that surplus
// relative to the tile start).
__global__ void __launch_bounds__(256) k_stripe_table(const int32_t *__restrict__ cell_cnt,
                                                      unsigned long long *masks, int32_t *apre,
                                                      int32_t *cell_off, int pad_min, int32_t *tile_cnt_out,
                                                      int32_t *pad_ranks, const int RMAX, SortHdr *hdr) {
    __shared__ unsigned long long s_mask[4];
    const int c = threadIdx.x, lane = c & 63, wv = c >> 6;
    const long t = blockIdx.x;
    const int n = cell_cnt[t * TCELLS + c];
    {   // the deepest cell of the store (for the caller: cells deeper than RMAX leave the stripes)
        int m = n;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o, 64));
        // (65 000 tiles x 4 waves on one address would cost more than the table: most waves see a maximum that is
        // already there -- a stale read only lets a redundant atomic through)
        if (lane == 0 && m > *(volatile int32_t *)&hdr->deepest) atomicMax(&hdr->deepest, m);
    }
    int run = 0, ra = 0;
    for (int r = 0; r < RMAX; r++) {
        unsigned long long m = __ballot(n > r);
        if (lane == 0) s_mask[wv] = m;
        __syncthreads();
        unsigned long long m0 = s_mask[0], m1 = s_mask[1], m2 = s_mask[2], m3 = s_mask[3];
        if (c < 4) masks[(t * RMAX + r) * 4 + c] = s_mask[c];
        if (c == 0) apre[t * (RMAX + 1) + r] = run;
        const int pop = __popcll(m0) + __popcll(m1) + __popcll(m2) + __popcll(m3);
        // LPA_ORDER_PADDED: a rank that most cells have is stored as a FULL stripe of 256 slots (slot = cell; the
        // missing cells are holes), so that a slot's position tells its cell; pop is non-increasing in r, so the
        // padded ranks are the leading ones
        const bool pad = pad_min > 0 && pop >= pad_min;     // block-uniform
        run += pad ? TCELLS : pop;
        if (pad) ra = r + 1;
        __syncthreads();
        if (!pop) break;  // block-uniform: no cell is this deep; later rows are never read
    }
    if (c == 0) apre[t * (RMAX + 1) + RMAX] = run;
    int extra = n > RMAX ? n - RMAX : 0, total_extra = 0;
    int ex = block_excl_scan256(extra, &total_extra);
    cell_off[t * TCELLS + c] = run + ex;
    if (c == 0) {
        if (total_extra > 0) atomicAdd(&hdr->tail, total_extra);
        // (the padded tile total is rounded up to a whole wave: tile starts stay multiples of 64 slots)
        if (tile_cnt_out) tile_cnt_out[t] = (run + total_extra + 63) & ~63;
        if (pad_ranks) pad_ranks[t] = ra;
    }
}

// slot of (cell c, rank r < RMAX) inside its tile for the striped orders: a stripe that holds all 256 slots (full, or
// padded: next offset - this offset == 256) is indexed by the cell itself
__device__ __forceinline__ int stripe_slot(const unsigned long long *__restrict__ m, const int32_t *__restrict__ ap,
                                           uint32_t r, int c) {
    const int a0 = ap[r], a1 = ap[r + 1];
    if (a1 - a0 == TCELLS) return a0 + c;
    int w = c >> 6, b = c & 63, below = 0;
    for (int q = 0; q < w; q++) below += __popcll(m[q]);
    below += __popcll(m[w] & ((1ull << b) - 1ull));
    return a0 + below;
}

// `striped`: 0 = CELL_MAJOR, 1 = STRIPED / PADDED (stripe tables), 2 + log2(L) = COLUMN with columns of L cells:
// stripes over the L cells of ONE column (the run of cells along the fastest axis) at a time, rank after rank, the
// columns one after the other.  slot = start of the column (cell-major scan) + what the column's ranks below r hold
// + the cells before c that reach rank r -- from the column's L counters (one or two cache lines, shared by all its
// particles).
__device__ __forceinline__ long dest_slot(uint32_t ck, uint32_t r, int striped,
                                          const int32_t *__restrict__ cell_base,
                                          const int32_t *__restrict__ tile_off,
                                          const int32_t *__restrict__ cell_off,
                                          const unsigned long long *__restrict__ masks,
                                          const int32_t *__restrict__ apre, const int RMAX,
                                          const int32_t *__restrict__ cell_cnt = nullptr) {
    if (r & RANK_LOCAL) r = (r & ~RANK_LOCAL) + (uint32_t)cell_base[ck];
    if (!striped) return (long)cell_off[ck] + r;
    if (striped >= 2) {
        const int L = 1 << (striped - 2);
        const uint32_t c0 = ck & ~(uint32_t)(L - 1);
        const int32_t *cnt = cell_cnt + c0;
        long o = cell_off[c0];
        const int cc = (int)(ck - c0);
        for (int q = 0; q < L; q++) {
            const int n = cnt[q];
            o += min(n, (int)r) + (q < cc && n > (int)r ? 1 : 0);
        }
        return o;
    }
    long t = ck >> 8;
    int c = ck & 255;
    if (r < RMAX) return (long)tile_off[t] + stripe_slot(masks + (t * RMAX + r) * 4, apre + t * (RMAX + 1), r, c);
    return (long)tile_off[t] + cell_off[ck] + (r - RMAX);
}

__global__ void __launch_bounds__(256) k_scatter(PartV s, PartV d, const uint32_t *__restrict__ key,
                                                 const uint32_t *__restrict__ rank,
                                                 const int32_t *__restrict__ tile_off,
                                                 const int32_t *__restrict__ cell_off,
                                                 const unsigned long long *__restrict__ masks,
                                                 const int32_t *__restrict__ apre, int striped,
                                                 const SortHdr *hdr, const int32_t *__restrict__ cell_cnt,
                                                 long first_block, int rmax) {
    long ip = (first_block + blockIdx.x) * blockDim.x + threadIdx.x;
    if (ip >= s.n) return;
    if (hdr->overflow) return;                         // the destination cannot hold the order: see SortHdr
    if (hdr->prev_valid && ip < hdr->prev_n) return;   // moved by k_scatter_tiled
    uint32_t ck = key[ip];
    if (ck == KEY_DEAD) return;
    const long o = dest_slot(ck, rank[ip], striped, nullptr, tile_off, cell_off, masks, apre, rmax, cell_cnt);
    d.x[o] = s.x[ip]; d.y[o] = s.y[ip];
    if (s.z && d.z) d.z[o] = s.z[ip];
    d.ux[o] = s.ux[ip]; d.uy[o] = s.uy[ip]; d.uz[o] = s.uz[ip];
    if (s.ig && d.ig) d.ig[o] = s.ig[ip];
    d.w[o] = s.w[ip];
    if (s.id && d.id) d.id[o] = s.id[ip];
    if (s.eb[0] && d.eb[0]) {
#pragma unroll
        for (int c = 0; c < 6; c++) d.eb[c][o] = s.eb[c][ip];
    }
}

// ---- tile-staged scatter ------------------------------------------------------------------------------
// k_scatter above writes 8-byte values to 64 unrelated cache lines per wave instruction: 6-7 ms for 67 M
// particles, bound by store transactions, not bytes.  When the source is itself tile ordered (every
// re-sort) a particle almost always lands in the destination range of its old tile, so one workgroup per
// old tile stages one attribute of a destination window in LDS (random LDS writes are cheap) and writes
// the window out as contiguous lines; only particles that changed tile take the scattered store.
// The old tiling is only a work partition: the result is right whatever order the source is in.
struct AttrList {
    int n;
    const double *src[16];
    double *dst[16];
};

#ifndef LPA_ST_NBUF
#define LPA_ST_NBUF 2
#endif
#ifndef LPA_ST_PREFETCH
#define LPA_ST_PREFETCH 1
#endif
#ifndef LPA_ST_THREADS
#define LPA_ST_THREADS 1024
#define LPA_ST_PT 8
#define LPA_ST_W 8192
#endif
constexpr int ST_THREADS = LPA_ST_THREADS, ST_PT = LPA_ST_PT, ST_W = LPA_ST_W;   // threads, particles per thread and chunk, window slots
constexpr int ST_BITS = 65536;                               // destination slots covered by one bitmap pass
constexpr int ST_RUN = 4;                                    // chunks of an old tile per workgroup (see the kernel)

__global__ void __launch_bounds__(1024) k_save_prev(SortHdr *hdr, const int32_t *tile_off, int32_t *tile_off_prev,
                                                    int32_t *run_off, int ntiles, long src_n, long prefix_hint) {
    for (int t = threadIdx.x; t <= ntiles; t += blockDim.x) tile_off_prev[t] = tile_off[t];
    {   // the tile-staged scatter's work list: a tile of c chunks is ceil(c / ST_RUN) runs, an empty tile none
        __shared__ int32_t s_runs[1024];
        const int tid = threadIdx.x, per = (ntiles + 1023) / 1024;
        const int lo = min(tid * per, ntiles), hi = min(lo + per, ntiles);
        constexpr int RUN_SLOTS = ST_RUN * ST_THREADS * ST_PT;
        int32_t sum = 0;
        for (int t = lo; t < hi; t++) sum += (tile_off[t + 1] - tile_off[t] + RUN_SLOTS - 1) / RUN_SLOTS;
        s_runs[tid] = sum;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            int32_t a = tid >= o ? s_runs[tid - o] : 0;
            __syncthreads();
            s_runs[tid] += a;
            __syncthreads();
        }
        int32_t off = s_runs[tid] - sum;
        for (int t = lo; t < hi; t++) {
            run_off[t] = off;
            off += (tile_off[t + 1] - tile_off[t] + RUN_SLOTS - 1) / RUN_SLOTS;
        }
        if (tid == 1023) run_off[ntiles] = s_runs[1023];
    }
    if (threadIdx.x == 0) {
        bool ok = hdr->magic == SORT_MAGIC && hdr->prev_ntiles == ntiles && hdr->n_live <= src_n;
        hdr->prev_valid = ok ? 1 : 0;
        hdr->prev_n = ok ? hdr->n_live : 0;
        // the per-particle kernels were launched from slot prefix_hint on: everything below must be covered by the
        // tile-ordered prefix, or those particles would be lost -- refuse instead (k_tile_scan keeps the bit)
        hdr->overflow = prefix_hint > (ok ? (long)hdr->n_live : 0l) ? SORT_BAD_HINT : 0;
        hdr->deepest = hdr->tail = hdr->tiles_in_use = 0;
        hdr->magic = SORT_MAGIC;
        hdr->prev_ntiles = ntiles;
    }
}

__global__ void __launch_bounds__(ST_THREADS) k_scatter_tiled(
    AttrList al, const SortHdr *hdr, const int32_t *__restrict__ tile_off_prev,
    const uint32_t *__restrict__ key, const uint32_t *__restrict__ rank, const int32_t *__restrict__ cell_base,
    const int32_t *__restrict__ tile_off, const int32_t *__restrict__ cell_off,
    const unsigned long long *__restrict__ masks, const int32_t *__restrict__ apre, int striped,
    const int32_t *__restrict__ cell_cnt, int rmax, const int32_t *__restrict__ run_off, int ntiles) {
    __shared__ double s_val[LPA_ST_NBUF][ST_W]; // double buffered: one barrier per (attribute, window)
    __shared__ uint32_t s_bits[ST_BITS / 32];   // slots of the tile's destination range this chunk fills
    __shared__ int s_band[2];                   // lowest / highest destination of the chunk inside its own tile
    if (!hdr->prev_valid || hdr->overflow) return;
    // One workgroup per RUN of ST_RUN chunks of an old tile (k_save_prev's list: a tile of up to ST_RUN chunks -- C2's hold
    // two -- is one workgroup as before, a deep tile is shared by as many as it has runs, an empty tile costs none; the
    // chunks are independent of each other).  The launch covers an upper bound of the runs.
    const int w = blockIdx.x;
    if (w >= run_off[ntiles]) return;
    int t;
    {   // the tile this run belongs to: last t with run_off[t] <= w (uniform binary search)
        int lo_ = 0, hi_ = ntiles;
        while (hi_ - lo_ > 1) {
            const int mid = (lo_ + hi_) >> 1;
            if (run_off[mid] <= w) lo_ = mid; else hi_ = mid;
        }
        t = lo_;
    }
    const int run = w - run_off[t];
    const int sb = tile_off_prev[t] + run * (ST_RUN * ST_THREADS * ST_PT);
    const int se = min(tile_off_prev[t + 1], sb + ST_RUN * ST_THREADS * ST_PT);
    const int db0 = tile_off[t], de0 = tile_off[t + 1];   // destination range of the same tile
    for (int c0 = sb; c0 < se; c0 += ST_THREADS * ST_PT) {
        int dest[ST_PT];                                   // -1: dead, dropped
#pragma unroll
        for (int j = 0; j < ST_PT; j++) {
            int ip = c0 + j * ST_THREADS + (int)threadIdx.x;
            dest[j] = -1;
            if (ip < se) {
                uint32_t ck = key[ip];
                if (ck != KEY_DEAD) dest[j] = (int)dest_slot(ck, rank[ip], striped, cell_base, tile_off, cell_off, masks, apre, rmax, cell_cnt);
            }
        }
        // The band of the tile's destination range this chunk writes to.  Ranks follow the order of the source (the
        // count pass walks a tile chunk by chunk), so a chunk of a deep tile lands in a narrow band of it: walking the
        // whole range for every chunk is quadratic in the tile's population (16 tiles of 10^6 particles: 160 ms a sort).
        if (threadIdx.x == 0) { s_band[0] = de0; s_band[1] = db0 - 1; }
        __syncthreads();
        {
            int lo = de0, hi = db0 - 1;
#pragma unroll
            for (int j = 0; j < ST_PT; j++)
                if (dest[j] >= db0 && dest[j] < de0) { lo = min(lo, dest[j]); hi = max(hi, dest[j]); }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { lo = min(lo, __shfl_xor(lo, o, 64)); hi = max(hi, __shfl_xor(hi, o, 64)); }
            if ((threadIdx.x & 63) == 0 && hi >= lo) { atomicMin(&s_band[0], lo); atomicMax(&s_band[1], hi); }
        }
        __syncthreads();
        // (window starts stay on the tile's own ST_W grid: whole lines whatever the band)
        const int bb = s_band[1] >= s_band[0] ? db0 + (s_band[0] - db0) / ST_W * ST_W : db0;
        const int be = s_band[1] >= s_band[0] ? s_band[1] + 1 : db0;
        bool first_pass = true;
        // (a band of more than ST_BITS slots is covered in several passes)
        for (int db = bb; db < be || first_pass; db += ST_BITS) {
            const int de = min(be, db + ST_BITS);
            for (int i = threadIdx.x; i < ST_BITS / 32; i += ST_THREADS) s_bits[i] = 0u;
            __syncthreads();
#pragma unroll
            for (int j = 0; j < ST_PT; j++)
                if (dest[j] >= db && dest[j] < de) atomicOr(&s_bits[(dest[j] - db) >> 5], 1u << ((dest[j] - db) & 31));
            __syncthreads();
            int phase = 0;
            // the loads of attribute a + 1 are issued before the window phases of attribute a: with one
            // workgroup per CU nothing else hides their latency behind the barriers
            double vn[ST_PT];
#pragma unroll
            for (int j = 0; j < ST_PT; j++) {
                int ip = c0 + j * ST_THREADS + (int)threadIdx.x;
                vn[j] = (LPA_ST_PREFETCH && dest[j] >= 0) ? al.src[0][ip] : 0.0;
            }
            for (int a = 0; a < al.n; a++) {
                const double *__restrict__ src = al.src[a];
                double *__restrict__ dst = al.dst[a];
                double v[ST_PT];
#pragma unroll
                for (int j = 0; j < ST_PT; j++) {
                    int ip = c0 + j * ST_THREADS + (int)threadIdx.x;
                    if (LPA_ST_PREFETCH) {
                        v[j] = vn[j];
                        if (a + 1 < al.n && dest[j] >= 0) vn[j] = al.src[a + 1][ip];
                    } else {
                        v[j] = dest[j] >= 0 ? src[ip] : 0.0;
                    }
                    // changed tile: the scattered store (a few per cent of the particles), first pass only
                    if (first_pass && dest[j] >= 0 && (dest[j] < db0 || dest[j] >= de0)) dst[dest[j]] = v[j];
                }
                for (int wb = db; wb < de; wb += ST_W, phase++) {
                    double *buf = s_val[LPA_ST_NBUF == 2 ? (phase & 1) : 0];
#pragma unroll
                    for (int j = 0; j < ST_PT; j++) {
                        unsigned o = (unsigned)(dest[j] - wb);
                        if (dest[j] >= 0 && o < (unsigned)ST_W && dest[j] < de) buf[o] = v[j];
                    }
                    // one barrier per phase: the buffer written now was last read two phases ago, and
                    // every thread has passed the barrier of the phase in between since
                    __syncthreads();
                    int wn = min(ST_W, de - wb);
                    for (int i = threadIdx.x; i < wn; i += ST_THREADS) {
                        int bi = wb - db + i;
                        if ((s_bits[bi >> 5] >> (bi & 31)) & 1u) dst[wb + i] = buf[i];
                    }
                    if (LPA_ST_NBUF == 1) __syncthreads();   // single buffer: drained before it is refilled
                }
            }
            __syncthreads();   // before the bitmap / buffers are reused
            first_pass = false;
        }
    }
}

static int sort_tiles(int dim, const char *name, const lpa_grid *g, const lpa_particles *src,
                      const lpa_particles *dst, void *workspace, int64_t workspace_bytes,
                      int32_t block_particles, int32_t order, lpa_tiling *out, void *stream, double ahead = 0.0) {
    LPA_REQUIRE(ahead >= 0.0 && ahead == ahead, "%s: bad look-ahead time", name);
    LPA_REQUIRE(g && g->nx > 0 && g->ny > 0 && g->dx > 0 && g->dy > 0 && (dim == 2 || (g->nz > 1 && g->dz > 0)),
                "%s: bad grid", name);
    // inv_gamma may be left out of BOTH stores: a store whose fused kernels run with LPA_PUSH_NO_IG holds a stale array
    // that nobody wants moved (the destination's is rebuilt by lpa_refresh_inv_gamma before anything reads it)
    lpa_particles s_chk = *src, d_chk = *dst;
    if (!src->inv_gamma && !dst->inv_gamma) s_chk.inv_gamma = s_chk.ux, d_chk.inv_gamma = d_chk.ux;
    LPA_REQUIRE(lpa_part_ok(&s_chk, dim) && lpa_part_ok(&d_chk, dim) && workspace && out,
                "%s: bad particle stores / workspace", name);
    LPA_REQUIRE(order == LPA_ORDER_CELL_MAJOR || order == LPA_ORDER_STRIPED || order == LPA_ORDER_PADDED ||
                    order == LPA_ORDER_COLUMN,
                "%s: bad order", name);
    LPA_REQUIRE(src->n < (1ll << 31) - 1, "%s: more than 2^31 particles in one store", name);
    LPA_REQUIRE(dst->n >= src->n, "%s: dst capacity (dst->n) smaller than src->n", name);
    LPA_REQUIRE(block_particles >= 1024, "%s: block_particles must be >= 1024", name);
    LPA_REQUIRE(dst->x != src->x, "%s: the sort is out of place", name);
    lpa_grid gg = *g;
    if (dim == 2) gg.nz = 1;
    SortWs w;
    // the work-block table is sized for the slots the DESTINATION can hold (a padded order has more slots than src->n)
    LPA_REQUIRE(out->stripe_ranks >= 0, "%s: lpa_tiling.stripe_ranks < 0", name);
    int64_t need = ws_layout(&gg, dst->n, block_particles, (char *)workspace, &w, out->stripe_ranks);
    if (need > workspace_bytes) {
        lpa_set_error("%s: workspace %lld B < %lld B", name, (long long)workspace_bytes, (long long)need);
        return LPA_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    int tiles_x, tiles_y, tiles_z = 0;
    if (dim == 2) {
        tiles_x = (g->nx + TX - 1) / TX; tiles_y = (g->ny + TY - 1) / TY;
    } else {
        tiles_x = (g->nx + T3X - 1) / T3X; tiles_y = (g->ny + T3Y - 1) / T3Y; tiles_z = (g->nz + T3Z - 1) / T3Z;
    }
    if (hipError_t me = hipMemsetAsync(w.cell_cnt, 0, sizeof(int32_t) * (size_t)w.ntiles * TCELLS, st); me != hipSuccess) {
        lpa_set_error("%s: memset of the cell counters failed: %s", name, hipGetErrorString(me));
        return LPA_ERR_HIP;
    }
    PartV sv = make_partv(src), dv = make_partv(dst);
    const long hint = out->prefix_hint;
    LPA_REQUIRE(hint >= 0 && hint <= src->n, "%s: prefix_hint outside the source", name);
    const long first_block = hint / 256;       // per-particle kernels start here (see lpa_tiling.prefix_hint)
    hipLaunchKernelGGL(k_save_prev, dim3(1), dim3(1024), 0, st, w.hdr, w.tile_off, w.tile_off_prev, w.run_off, w.ntiles,
                       (long)src->n, hint);
    LPA_CHECK_LAUNCH("k_save_prev");
    if (src->n > 0) {
        KeyGeom kg;
        kg.dim = dim; kg.nx = g->nx; kg.ny = g->ny; kg.nz = dim == 3 ? g->nz : 1;
        kg.tiles_y = tiles_y; kg.tiles_z = tiles_z;
        kg.x0 = g->x0; kg.y0 = g->y0; kg.z0 = dim == 3 ? g->z0 : 0.0;
        kg.inv_dx = 1.0 / g->dx; kg.inv_dy = 1.0 / g->dy; kg.inv_dz = dim == 3 ? 1.0 / g->dz : 0.0;
        kg.ahead = LPA_C * ahead;
        // tile-ordered prefix (re-sorts): LDS counters per old tile; does nothing on a first sort
        hipLaunchKernelGGL(k_cell_count_tiled, dim3(w.ntiles), dim3(512), 0, st, sv, kg, w.hdr, w.tile_off_prev,
                           w.cell_cnt, w.cell_base, w.key, w.rank);
        LPA_CHECK_LAUNCH("k_cell_count_tiled");
        const long nb = (src->n + 255) / 256 - first_block;
        if (nb > 0) {
            hipLaunchKernelGGL(k_cell_count, dim3((unsigned)nb), dim3(256), 0, st, sv, kg, w.hdr, w.cell_cnt, w.key, w.rank,
                               first_block);
            LPA_CHECK_LAUNCH("k_cell_count");
        }
    }
    const bool padded = order == LPA_ORDER_PADDED;
    if (padded) {
        // the padded tile totals come out of the stripe tables: tables first, then the scan; the destination's
        // positions start as NaN, so every slot the scatter does not fill is a hole
        hipLaunchKernelGGL(k_stripe_table, dim3(w.ntiles), dim3(256), 0, st, w.cell_cnt, w.masks, w.apre, w.cell_off,
                           LPA_PAD_MIN_CELLS, w.tile_cnt, w.pad_ranks, w.rmax, w.hdr);
        LPA_CHECK_LAUNCH("k_stripe_table (padded)");
        if (hipMemsetAsync(dv.x, 0xFF, sizeof(double) * (size_t)dst->n, st) != hipSuccess ||
            hipMemsetAsync(dv.y, 0xFF, sizeof(double) * (size_t)dst->n, st) != hipSuccess ||
            (dim == 3 && hipMemsetAsync(dv.z, 0xFF, sizeof(double) * (size_t)dst->n, st) != hipSuccess)) {
            lpa_set_error("%s: memset failed", name);
            return LPA_ERR_HIP;
        }
    } else {
        hipLaunchKernelGGL(k_tile_sum, dim3(w.ntiles), dim3(256), 0, st, w.cell_cnt, w.tile_cnt);
        LPA_CHECK_LAUNCH("k_tile_sum");
    }
    hipLaunchKernelGGL(k_tile_scan, dim3(1), dim3(1024), 0, st, w.ntiles, w.tile_cnt, w.tile_off, w.blk_tile,
                       w.blk_begin, w.blk_end, w.hdr, (int)block_particles, w.max_blocks, (long)dst->n);
    LPA_CHECK_LAUNCH("k_tile_scan");
    if (order == LPA_ORDER_STRIPED)
        hipLaunchKernelGGL(k_stripe_table, dim3(w.ntiles), dim3(256), 0, st, w.cell_cnt, w.masks, w.apre,
                           w.cell_off, 0, (int32_t *)nullptr, w.pad_ranks, w.rmax, w.hdr);
    else if (!padded)
        hipLaunchKernelGGL(k_cell_scan, dim3(w.ntiles), dim3(256), 0, st, w.cell_cnt, w.tile_off, w.cell_off);
    LPA_CHECK_LAUNCH("k_cell_scan / k_stripe_table");
    if (src->n > 0) {
        AttrList al;
        al.n = 0;
        auto add = [&](const double *a, double *b) { if (a && b) { al.src[al.n] = a; al.dst[al.n] = b; al.n++; } };
        add(sv.x, dv.x); add(sv.y, dv.y); add(sv.z, dv.z); add(sv.ux, dv.ux); add(sv.uy, dv.uy);
        add(sv.uz, dv.uz); add(sv.ig, dv.ig); add(sv.w, dv.w);
        add((const double *)sv.id, (double *)dv.id);       // 8-byte payload, moved as is
        if (sv.eb[0] && dv.eb[0])
            for (int c = 0; c < 6; c++) add(sv.eb[c], dv.eb[c]);
        // columns: the 16 z-cells of an (x, y) column of a 3-D tile, the 32 y-cells of a row of a 2-D tile
        const int striped = order == LPA_ORDER_COLUMN ? 2 + (dim == 3 ? 4 : 5)
                                                      : (int)(order == LPA_ORDER_STRIPED || order == LPA_ORDER_PADDED);
        // tile-ordered prefix of the source (re-sorts): staged per tile; does nothing on a first sort
        // (runs <= tiles in use + slots / run size: the bound below; workgroups beyond the list return at once)
        const long runs_max = (long)w.ntiles + src->n / ((long)ST_RUN * ST_THREADS * ST_PT) + 1;
        hipLaunchKernelGGL(k_scatter_tiled, dim3((unsigned)runs_max), dim3(ST_THREADS), 0, st, al, w.hdr, w.tile_off_prev,
                           w.key, w.rank, w.cell_base, w.tile_off, w.cell_off, w.masks, w.apre, striped, w.cell_cnt, w.rmax,
                           w.run_off, w.ntiles);
        LPA_CHECK_LAUNCH("k_scatter_tiled");
        const long nb = (src->n + 255) / 256 - first_block;
        if (nb > 0) {
            hipLaunchKernelGGL(k_scatter, dim3((unsigned)nb), dim3(256), 0, st, sv, dv, w.key, w.rank, w.tile_off,
                               w.cell_off, w.masks, w.apre, striped, w.hdr, w.cell_cnt, first_block, w.rmax);
            LPA_CHECK_LAUNCH("k_scatter");
        }
    }
    out->tiles_x = tiles_x;
    out->tiles_y = tiles_y;
    out->tiles_z = tiles_z;
    out->prefix_hint = 0;
    for (int c = 0; c < 8; c++) out->scratch[c] = nullptr;   // the caller may attach the idle store
    out->n_sorted = src->n;  // upper bound known on the host; the exact count is hdr->n_live
    out->max_blocks = w.max_blocks;
    out->order = order;
    out->tile_off = w.tile_off;
    out->blk_tile = w.blk_tile;
    out->blk_begin = w.blk_begin;
    out->blk_end = w.blk_end;
    out->n_blocks = &w.hdr->n_blocks;
    // idle until the next sort: scratch of the tiled push kernel's in-kernel re-seating
    out->pad_ranks = w.pad_ranks;
    out->aux_slot = w.key;
    out->aux_info = w.rank;
    out->slot_class = nullptr;   // the caller may attach a class array (and must then set class_init)
    out->class_init = 0;
    out->reloc_stats = nullptr;
    out->stripe_ranks = w.rmax;   // what the layout used: pass it again (or 0 for the default rule) with the same workspace
    return LPA_OK;
}

extern "C" int lpa_sort_tiles_2d(const lpa_grid *g, const lpa_particles *src, const lpa_particles *dst,
                                 void *workspace, int64_t workspace_bytes, int32_t block_particles,
                                 int32_t order, lpa_tiling *out, void *stream) {
    return sort_tiles(2, "lpa_sort_tiles_2d", g, src, dst, workspace, workspace_bytes, block_particles, order,
                      out, stream);
}

extern "C" int lpa_sort_tiles_3d(const lpa_grid *g, const lpa_particles *src, const lpa_particles *dst,
                                 void *workspace, int64_t workspace_bytes, int32_t block_particles,
                                 int32_t order, lpa_tiling *out, void *stream) {
    return sort_tiles(3, "lpa_sort_tiles_3d", g, src, dst, workspace, workspace_bytes, block_particles, order,
                      out, stream);
}

// The same sorts with the cells taken `ahead` seconds down every particle's straight path (x + v * ahead): a store the
// caller re-sorts every T steps because its particles outrun the tile margin stays valid about twice as long when it is
// binned for the MIDDLE of the interval (ahead = T dt / 2) -- the kernels accept any binning (what does not fit a tile's
// staged region takes the overflow list), so this is a work partition, not a change of results.
extern "C" int lpa_sort_tiles_ahead_2d(const lpa_grid *g, const lpa_particles *src, const lpa_particles *dst,
                                       void *workspace, int64_t workspace_bytes, int32_t block_particles,
                                       int32_t order, lpa_tiling *out, double ahead, void *stream) {
    return sort_tiles(2, "lpa_sort_tiles_ahead_2d", g, src, dst, workspace, workspace_bytes, block_particles, order,
                      out, stream, ahead);
}

extern "C" int lpa_sort_tiles_ahead_3d(const lpa_grid *g, const lpa_particles *src, const lpa_particles *dst,
                                       void *workspace, int64_t workspace_bytes, int32_t block_particles,
                                       int32_t order, lpa_tiling *out, double ahead, void *stream) {
    return sort_tiles(3, "lpa_sort_tiles_ahead_3d", g, src, dst, workspace, workspace_bytes, block_particles, order,
                      out, stream, ahead);
}

// =====================================================================================================
// slab migration along x (core/patch/sync_particles_2d.c:204-518, core/mpi/sync_particles_2d.c:274-770):
// leavers are copied into the low / high send buffer and killed; arrivals are appended to the arrival
// area behind the tile-ordered particles.  No host round trip: the message has a fixed size and carries
// its own count.  Buffer layout (doubles): [0] = count (int64 bit pattern), then
// [LPA_MIG_NATTR][capacity] SoA, attribute order x y z ux uy uz inv_gamma w id.
// =====================================================================================================
#include "lpa_migrate.hpp"   // FreeSlots, migrate_pack_one (shared with the push kernels' rest launch)


__global__ void __launch_bounds__(256) k_migrate_pack_x(PartV p, double xlo, double xhi, double *buf_lo,
                                                        double *buf_hi, long cap, int32_t *surplus) {
    long ip = (long)blockIdx.x * blockDim.x + threadIdx.x;
    migrate_pack_one(p, ip, xlo, xhi, buf_lo, buf_hi, cap, FreeSlots{nullptr, nullptr, 0, 0}, nullptr, 0, 0,
                     ip < p.n, surplus);
}

// the same over the only particles that can have left a tile-ordered store: the first / last `edge_tiles`
// tiles (x is the slowest tile index, so these are the tile columns next to the two x faces) and the
// loose particles behind the ordered range.  The ranges come from the device-side tile offsets: no host
// round trip, fixed grid.
__global__ void __launch_bounds__(256) k_migrate_pack_edges_x(PartV p, const int32_t *__restrict__ tile_off,
                                                              int ntiles, int edge_tiles, long n_sorted,
                                                              double xlo, double xhi, double *buf_lo,
                                                              double *buf_hi, long cap, FreeSlots fs,
                                                              int32_t *surplus, const int32_t *loose_limit = nullptr) {
    const long a1 = tile_off[edge_tiles], b0 = tile_off[ntiles - edge_tiles], b1 = tile_off[ntiles];
    long nl = p.n > n_sorted ? p.n - n_sorted : 0;
    // (arrival area: only the slots the unpack kernels have handed out since the sort can hold a particle)
    if (loose_limit && (long)*loose_limit < nl) nl = *loose_limit;
    const long nb = b1 - b0;
    const long total = a1 + nb + nl;
    // wave-uniform trip count (the slot allocation is a wave-wide operation); four positions per thread are loaded before
    // any is looked at: one load per wave and trip left the scan latency bound (27 us for 5 M positions)
    const long lane = threadIdx.x & 63u, stride = (long)gridDim.x * blockDim.x;
    for (long t0 = (long)blockIdx.x * blockDim.x + threadIdx.x - lane; t0 < total; t0 += 4 * stride) {
        long ip[4];
        double xv[4];
        bool act[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const long t = t0 + u * stride + lane;
            act[u] = t < total;
            ip[u] = !act[u] ? 0 : (t < a1 ? t : (t < a1 + nb ? b0 + (t - a1) : n_sorted + (t - a1 - nb)));
            xv[u] = act[u] ? p.x[ip[u]] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (t0 + u * stride >= total) break;      // (wave-uniform)
            migrate_pack_one(p, ip[u], xlo, xhi, buf_lo, buf_hi, cap, fs, tile_off, ntiles, n_sorted, act[u], surplus, true, xv[u]);
        }
    }
}

// the same over the slots the push kernels of this step reported (lpa_push_params.leavers): no scan at all
__global__ void __launch_bounds__(256) k_migrate_pack_list(PartV p, const unsigned long long *__restrict__ list,
                                                           const uint32_t *__restrict__ list_count, long list_cap,
                                                           const int32_t *__restrict__ tile_off, int ntiles, long n_sorted,
                                                           double xlo, double xhi, double *buf_lo, double *buf_hi,
                                                           long cap, FreeSlots fs, int32_t *surplus) {
    long total = *list_count;
    if (total > list_cap) total = list_cap;
    const long lane = threadIdx.x & 63u;
    for (long t0 = (long)blockIdx.x * blockDim.x + threadIdx.x - lane; t0 < total; t0 += (long)gridDim.x * blockDim.x) {
        const long t = t0 + lane;
        const bool active = t < total;
        const unsigned long long ent = active ? list[t] : 0ull;
        const long ip = (long)(ent & 0xffffffffull);
        const int tile = (int)(ent >> 32) - 1;
        migrate_pack_one(p, ip, xlo, xhi, buf_lo, buf_hi, cap, fs, tile_off, ntiles, n_sorted, active && ip < p.n, surplus, false,
                         0.0, tile < ntiles ? tile : -1);
    }
}

// one atomic per wave for the lanes that need an arrival-area slot; -1 for the others
__device__ __forceinline__ long area_slot_wave(bool need, int32_t *cursor) {
    const unsigned long long m = __ballot(need);
    if (!m) return -1;
    const int lane = (int)(threadIdx.x & 63u), leader = __ffsll((long long)m) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(cursor, __popcll(m));
    base = __shfl(base, leader);
    return need ? (long)base + __popcll(m & ((1ull << lane) - 1ull)) : -1;
}

__global__ void __launch_bounds__(256) k_migrate_unpack(PartV p, long first_slot, long area_cap,
                                                        int32_t *cursor, const double *buf, long cap,
                                                        double shift_x) {
    long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long n = (long)*(const unsigned long long *)buf;
    if (n > cap) n = cap;
    long slot = area_slot_wave(t < n, cursor);
    if (t >= n || slot >= area_cap) return;  // cursor > area_cap tells the host the area overflowed
    long o = first_slot + slot;
    const double *d = buf + 1;
    p.x[o] = d[0 * cap + t] + shift_x;
    p.y[o] = d[1 * cap + t];
    if (p.z) p.z[o] = d[2 * cap + t];
    p.ux[o] = d[3 * cap + t];
    p.uy[o] = d[4 * cap + t];
    p.uz[o] = d[5 * cap + t];
    p.ig[o] = d[6 * cap + t];
    p.w[o] = d[7 * cap + t];
    if (p.id) p.id[o] = (unsigned long long)__double_as_longlong(d[8 * cap + t]);
    if (p.dead) p.dead[o] = 0;
}

// unpack with free slots: an arrival whose tile has a recorded free slot takes it (and is back on the tiled
// path at once); the others go to the arrival area as before
__global__ void __launch_bounds__(256) k_migrate_unpack_tiled(PartV p, KeyGeom kg, int ntiles, FreeSlots fs,
                                                              long first_slot, long area_cap, int32_t *cursor,
                                                              const double *buf, long cap, double shift_x) {
    long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long n = (long)*(const unsigned long long *)buf;
    if (n > cap) n = cap;
    const bool active = t < n;
    const double *d = buf + 1;
    const double x = active ? d[0 * cap + t] + shift_x : 0.0, y = active ? d[1 * cap + t] : 0.0,
                 z = (active && kg.dim == 3) ? d[2 * cap + t] : 0.0;
    long o = -1;
    if (active && !(isnan(x) || isnan(y) || isnan(z))) {
        int is = ifloor((x - kg.x0) * kg.inv_dx + 0.5), js = ifloor((y - kg.y0) * kg.inv_dy + 0.5);
        is = is < 0 ? 0 : (is >= kg.nx ? kg.nx - 1 : is);
        js = js < 0 ? 0 : (js >= kg.ny ? kg.ny - 1 : js);
        int tile;
        if (kg.dim == 2) {
            tile = (is / TX) * kg.tiles_y + js / TY;
        } else {
            int ks = ifloor((z - kg.z0) * kg.inv_dz + 0.5);
            ks = ks < 0 ? 0 : (ks >= kg.nz ? kg.nz - 1 : ks);
            tile = ((is / T3X) * kg.tiles_y + js / T3Y) * kg.tiles_z + ks / T3Z;
        }
        int e = edge_index(tile, ntiles, fs.edge_tiles);
        if (e >= 0) {
            int k = atomicSub(&fs.count[e], 1) - 1;
            if (k >= 0) o = fs.slot[(long)e * fs.depth + k];
            else atomicAdd(&fs.count[e], 1);
        }
    }
    {
        long slot = area_slot_wave(active && o < 0, cursor);
        if (!active) return;
        if (o < 0) {
            if (slot >= area_cap) return;  // cursor > area_cap tells the host the area overflowed
            o = first_slot + slot;
        }
    }
    p.x[o] = x;
    p.y[o] = y;
    if (p.z) p.z[o] = d[2 * cap + t];
    p.ux[o] = d[3 * cap + t];
    p.uy[o] = d[4 * cap + t];
    p.uz[o] = d[5 * cap + t];
    p.ig[o] = d[6 * cap + t];
    p.w[o] = d[7 * cap + t];
    if (p.id) p.id[o] = (unsigned long long)__double_as_longlong(d[8 * cap + t]);
    if (p.dead) p.dead[o] = 0;
}

extern "C" int lpa_migrate_pack_x(const lpa_particles *p, double xlo, double xhi, double *buf_lo,
                                  double *buf_hi, int64_t capacity, int32_t *surplus, void *stream) {
    LPA_REQUIRE(lpa_part_ok(p, 2) && buf_lo && buf_hi && capacity > 0 && xlo < xhi,
                "lpa_migrate_pack_x: bad args");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(buf_lo, 0, sizeof(double), st) != hipSuccess ||
        hipMemsetAsync(buf_hi, 0, sizeof(double), st) != hipSuccess) {
        lpa_set_error("lpa_migrate_pack_x: memset failed");
        return LPA_ERR_HIP;
    }
    if (p->n == 0) return LPA_OK;
    hipLaunchKernelGGL(k_migrate_pack_x, dim3((unsigned)((p->n + 255) / 256)), dim3(256), 0, st,
                       make_partv(p), xlo, xhi, buf_lo, buf_hi, (long)capacity, surplus);
    LPA_CHECK_LAUNCH("lpa_migrate_pack_x");
    return LPA_OK;
}

extern "C" int lpa_migrate_pack_edges_x(const lpa_particles *p, const lpa_tiling *t, int32_t edge_cols,
                                        double xlo, double xhi, double *buf_lo, double *buf_hi,
                                        int64_t capacity, const lpa_free_slots *fs, int32_t *surplus,
                                        void *stream) {
    LPA_REQUIRE(!fs || (free_slots_ok(fs, t) && fs->edge_cols >= edge_cols),
                "lpa_migrate_pack_edges_x: bad free-slot stacks (edge_cols must cover the scanned columns)");
    LPA_REQUIRE(lpa_part_ok(p, 2) && buf_lo && buf_hi && capacity > 0 && xlo < xhi,
                "lpa_migrate_pack_edges_x: bad args");
    LPA_REQUIRE(t && t->tile_off && t->tiles_x > 0 && t->tiles_y > 0 && t->n_sorted >= 0 && t->n_sorted <= p->n,
                "lpa_migrate_pack_edges_x: bad tiling");
    LPA_REQUIRE(edge_cols >= 1 && 2 * edge_cols <= t->tiles_x, "lpa_migrate_pack_edges_x: bad edge_cols");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(buf_lo, 0, sizeof(double), st) != hipSuccess ||
        hipMemsetAsync(buf_hi, 0, sizeof(double), st) != hipSuccess) {
        lpa_set_error("lpa_migrate_pack_edges_x: memset failed");
        return LPA_ERR_HIP;
    }
    if (p->n == 0) return LPA_OK;
    const int per_col = t->tiles_y * (t->tiles_z > 0 ? t->tiles_z : 1);
    const int ntiles = t->tiles_x * per_col;
    // grid-stride over a range only known on the device: enough workgroups to keep the loads of a large
    // edge region in flight (1024 of them made the 3-D scan latency bound: 0.17 ms for 12 M positions)
    long nblk = (p->n + 1023) / 1024;      // (four positions per thread)
    if (nblk > 8192) nblk = 8192;
    hipLaunchKernelGGL(k_migrate_pack_edges_x, dim3((unsigned)nblk), dim3(256), 0, st, make_partv(p), t->tile_off, ntiles,
                       edge_cols * per_col, (long)t->n_sorted, xlo, xhi, buf_lo, buf_hi, (long)capacity,
                       make_free_slots(fs, t), surplus);
    LPA_CHECK_LAUNCH("lpa_migrate_pack_edges_x");
    return LPA_OK;
}

static int migrate_pack_list(const lpa_particles *p, const lpa_tiling *t, const uint64_t *list, const uint32_t *list_count,
                             int64_t list_capacity, double xlo, double xhi, double *buf_lo, double *buf_hi, int64_t capacity,
                             const lpa_free_slots *fs, int32_t *surplus, int zero_headers, void *stream) {
    LPA_REQUIRE(lpa_part_ok(p, 2) && list && list_count && list_capacity > 0 && buf_lo && buf_hi && capacity > 0 && xlo < xhi,
                "lpa_migrate_pack_list: bad args");
    LPA_REQUIRE(!fs || (t && t->tile_off && free_slots_ok(fs, t) && t->n_sorted >= 0 && t->n_sorted <= p->n),
                "lpa_migrate_pack_list: bad tiling / free-slot stacks");
    hipStream_t st = (hipStream_t)stream;
    if (zero_headers && (hipMemsetAsync(buf_lo, 0, sizeof(double), st) != hipSuccess ||
                         hipMemsetAsync(buf_hi, 0, sizeof(double), st) != hipSuccess)) {
        lpa_set_error("lpa_migrate_pack_list: memset failed");
        return LPA_ERR_HIP;
    }
    if (p->n == 0) return LPA_OK;
    int ntiles = 0;
    if (fs) ntiles = t->tiles_x * t->tiles_y * (t->tiles_z > 0 ? t->tiles_z : 1);
    // the list is short (what crosses a face in one step): a fixed small grid, grid-stride over the device-side count
    long nblk = (list_capacity + 255) / 256;
    if (nblk > 64) nblk = 64;
    hipLaunchKernelGGL(k_migrate_pack_list, dim3((unsigned)nblk), dim3(256), 0, st, make_partv(p), (const unsigned long long *)list, list_count,
                       (long)list_capacity, fs ? t->tile_off : nullptr, ntiles, fs ? (long)t->n_sorted : 0L, xlo, xhi, buf_lo,
                       buf_hi, (long)capacity, make_free_slots(fs, t), surplus);
    LPA_CHECK_LAUNCH("lpa_migrate_pack_list");
    return LPA_OK;
}

extern "C" int lpa_migrate_pack_list(const lpa_particles *p, const lpa_tiling *t, const uint64_t *list,
                                     const uint32_t *list_count, int64_t list_capacity, double xlo, double xhi,
                                     double *buf_lo, double *buf_hi, int64_t capacity, const lpa_free_slots *fs,
                                     int32_t *surplus, void *stream) {
    return migrate_pack_list(p, t, list, list_count, list_capacity, xlo, xhi, buf_lo, buf_hi, capacity, fs, surplus, 1, stream);
}

int lpai_migrate_pack_list(const lpa_particles *p, const lpa_tiling *t, const uint64_t *list, const uint32_t *list_count,
                           int64_t list_capacity, double xlo, double xhi, double *buf_lo, double *buf_hi, int64_t capacity,
                           const lpa_free_slots *fs, int32_t *surplus, void *stream) {
    return migrate_pack_list(p, t, list, list_count, list_capacity, xlo, xhi, buf_lo, buf_hi, capacity, fs, surplus, 0, stream);
}

extern "C" int lpa_migrate_unpack(const lpa_particles *p, int64_t first_slot, int64_t area_capacity,
                                  int32_t *cursor, const double *buf, int64_t capacity, double shift_x,
                                  void *stream) {
    LPA_REQUIRE(p && p->x && p->y && p->ux && p->uy && p->uz && p->inv_gamma && p->w && buf && cursor &&
                    capacity > 0 && first_slot >= 0 && area_capacity >= 0,
                "lpa_migrate_unpack: bad args");
    if (area_capacity == 0) return LPA_OK;
    hipLaunchKernelGGL(k_migrate_unpack, dim3((unsigned)((capacity + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, make_partv(p), (long)first_slot, (long)area_capacity, cursor,
                       buf, (long)capacity, shift_x);
    LPA_CHECK_LAUNCH("lpa_migrate_unpack");
    return LPA_OK;
}

extern "C" int lpa_migrate_unpack_tiled(const lpa_particles *p, const lpa_grid *g, const lpa_tiling *t,
                                        const lpa_free_slots *fs, int64_t first_slot, int64_t area_capacity,
                                        int32_t *cursor, const double *buf, int64_t capacity, double shift_x,
                                        void *stream) {
    LPA_REQUIRE(p && p->x && p->y && p->ux && p->uy && p->uz && p->inv_gamma && p->w && buf && cursor &&
                    capacity > 0 && first_slot >= 0 && area_capacity >= 0,
                "lpa_migrate_unpack_tiled: bad args");
    LPA_REQUIRE(g && g->nx > 0 && g->ny > 0 && g->dx > 0 && g->dy > 0 && t && t->tiles_x > 0 && t->tiles_y > 0 &&
                    free_slots_ok(fs, t),
                "lpa_migrate_unpack_tiled: bad grid / tiling / free-slot stacks");
    const int dim = t->tiles_z > 0 ? 3 : 2;
    LPA_REQUIRE(dim == 2 || (p->z && g->nz > 1 && g->dz > 0), "lpa_migrate_unpack_tiled: 3-D tiling needs z");
    KeyGeom kg;
    kg.dim = dim; kg.nx = g->nx; kg.ny = g->ny; kg.nz = dim == 3 ? g->nz : 1;
    kg.tiles_y = t->tiles_y; kg.tiles_z = dim == 3 ? t->tiles_z : 1;
    kg.x0 = g->x0; kg.y0 = g->y0; kg.z0 = g->z0;
    kg.inv_dx = 1.0 / g->dx; kg.inv_dy = 1.0 / g->dy; kg.inv_dz = dim == 3 ? 1.0 / g->dz : 0.0;
    const int ntiles = t->tiles_x * t->tiles_y * (dim == 3 ? t->tiles_z : 1);
    hipLaunchKernelGGL(k_migrate_unpack_tiled, dim3((unsigned)((capacity + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, make_partv(p), kg, ntiles, make_free_slots(fs, t), (long)first_slot,
                       (long)area_capacity, cursor, buf, (long)capacity, shift_x);
    LPA_CHECK_LAUNCH("lpa_migrate_unpack_tiled");
    return LPA_OK;
}

// =====================================================================================================
// internal forms used by lpa_step on slab ranks: pack without the header memsets (lpa_step zeroes every per-step
// counter with one launch), unpack of both faces in one launch
// =====================================================================================================
int lpai_migrate_pack(const lpa_particles *p, const lpa_tiling *t, int32_t edge_cols, double xlo, double xhi,
                      double *buf_lo, double *buf_hi, int64_t capacity, const lpa_free_slots *fs, int32_t *surplus,
                      int zero_headers, const int32_t *loose_limit, void *stream) {
    if (zero_headers)
        return edge_cols > 0 ? lpa_migrate_pack_edges_x(p, t, edge_cols, xlo, xhi, buf_lo, buf_hi, capacity, fs, surplus, stream)
                             : lpa_migrate_pack_x(p, xlo, xhi, buf_lo, buf_hi, capacity, surplus, stream);
    LPA_REQUIRE(lpa_part_ok(p, 2) && buf_lo && buf_hi && capacity > 0 && xlo < xhi, "lpai_migrate_pack: bad args");
    if (p->n == 0) return LPA_OK;
    hipStream_t st = (hipStream_t)stream;
    if (edge_cols <= 0) {
        hipLaunchKernelGGL(k_migrate_pack_x, dim3((unsigned)((p->n + 255) / 256)), dim3(256), 0, st, make_partv(p), xlo, xhi,
                           buf_lo, buf_hi, (long)capacity, surplus);
        LPA_CHECK_LAUNCH("lpai_migrate_pack");
        return LPA_OK;
    }
    LPA_REQUIRE(!fs || (free_slots_ok(fs, t) && fs->edge_cols >= edge_cols), "lpai_migrate_pack: bad free-slot stacks");
    LPA_REQUIRE(t && t->tile_off && t->tiles_x > 0 && t->tiles_y > 0 && t->n_sorted >= 0 && t->n_sorted <= p->n &&
                    2 * edge_cols <= t->tiles_x, "lpai_migrate_pack: bad tiling / edge_cols");
    const int per_col = t->tiles_y * (t->tiles_z > 0 ? t->tiles_z : 1);
    const int ntiles = t->tiles_x * per_col;
    long nblk = (p->n + 1023) / 1024;
    if (nblk > 8192) nblk = 8192;
    hipLaunchKernelGGL(k_migrate_pack_edges_x, dim3((unsigned)nblk), dim3(256), 0, st, make_partv(p), t->tile_off, ntiles,
                       edge_cols * per_col, (long)t->n_sorted, xlo, xhi, buf_lo, buf_hi, (long)capacity,
                       make_free_slots(fs, t), surplus, loose_limit);
    LPA_CHECK_LAUNCH("lpai_migrate_pack");
    return LPA_OK;
}

// arrivals of one face message seated: chunk `bx` (256 of its slots) -- every thread of the block takes part
__device__ __forceinline__ void migrate_unpack_body(const PartV &p, const KeyGeom &kg, int ntiles, const FreeSlots &fs,
                                                    long first_slot, long area_cap, int32_t *cursor, const double *buf,
                                                    long cap, double shift_x, long bx) {
    long t = bx * (long)blockDim.x + threadIdx.x;
    long n = (long)*(const unsigned long long *)buf;
    if (n > cap) n = cap;
    if (bx * (long)blockDim.x >= n) return;      // (block-uniform: nothing of this block's range arrived)
    const bool active = t < n;
    const double *d = buf + 1;
    const double x = active ? d[0 * cap + t] + shift_x : 0.0, y = active ? d[1 * cap + t] : 0.0,
                 z = (active && kg.dim == 3) ? d[2 * cap + t] : 0.0;
    long o = -1;
    if (fs.count && active && !(isnan(x) || isnan(y) || isnan(z))) {
        int is = ifloor((x - kg.x0) * kg.inv_dx + 0.5), js = ifloor((y - kg.y0) * kg.inv_dy + 0.5);
        is = is < 0 ? 0 : (is >= kg.nx ? kg.nx - 1 : is);
        js = js < 0 ? 0 : (js >= kg.ny ? kg.ny - 1 : js);
        int tile;
        if (kg.dim == 2) {
            tile = (is / TX) * kg.tiles_y + js / TY;
        } else {
            int ks = ifloor((z - kg.z0) * kg.inv_dz + 0.5);
            ks = ks < 0 ? 0 : (ks >= kg.nz ? kg.nz - 1 : ks);
            tile = ((is / T3X) * kg.tiles_y + js / T3Y) * kg.tiles_z + ks / T3Z;
        }
        int e = edge_index(tile, ntiles, fs.edge_tiles);
        if (e >= 0) {
            int k = atomicSub(&fs.count[e], 1) - 1;
            if (k >= 0) o = fs.slot[(long)e * fs.depth + k];
            else atomicAdd(&fs.count[e], 1);
        }
    }
    {
        long slot = area_slot_wave(active && o < 0, cursor);
        if (!active) return;
        if (o < 0) {
            if (slot >= area_cap) return;  // cursor > area_cap tells the host the area overflowed
            o = first_slot + slot;
        }
    }
    p.x[o] = x;
    p.y[o] = y;
    if (p.z) p.z[o] = d[2 * cap + t];
    p.ux[o] = d[3 * cap + t];
    p.uy[o] = d[4 * cap + t];
    p.uz[o] = d[5 * cap + t];
    p.ig[o] = d[6 * cap + t];
    p.w[o] = d[7 * cap + t];
    if (p.id) p.id[o] = (unsigned long long)__double_as_longlong(d[8 * cap + t]);
    if (p.dead) p.dead[o] = 0;
}

__global__ void __launch_bounds__(256) k_migrate_unpack2(PartV p, KeyGeom kg, int ntiles, FreeSlots fs, long first_slot,
                                                         long area_cap, int32_t *cursor, const double *buf_lo,
                                                         const double *buf_hi, long cap, double shift_lo,
                                                         double shift_hi) {
    migrate_unpack_body(p, kg, ntiles, fs, first_slot, area_cap, cursor, blockIdx.y == 0 ? buf_lo : buf_hi, cap,
                        blockIdx.y == 0 ? shift_lo : shift_hi, (long)blockIdx.x);
}

// The J / rho fold of a slab step AND the arrivals of every species in one launch (lpa_step; both follow the J round and
// touch disjoint data): the first fold_blocks blocks are k_fold_all's grid (fx x fy x fz, flattened), then per species and
// face cap / 256 blocks of k_migrate_unpack2's.
struct UnpackOne {
    PartV p;
    KeyGeom kg;
    int ntiles;
    FreeSlots fs;
    long first_slot, area_cap;
    int32_t *cursor;
    const double *buf_lo, *buf_hi;
};
struct UnpackSet { UnpackOne u[LPA_FOLD_UNPACK_MAX_SPECIES]; };

__global__ void __launch_bounds__(256) k_fold_unpack(GridV g, int axes, const double *__restrict__ r_lo,
                                                     const double *__restrict__ r_hi, const double *__restrict__ left_own,
                                                     int fx, int fy, int fz, UnpackSet us, int nspecies, long cap,
                                                     int chunks, double shift_lo, double shift_hi) {
    long b = blockIdx.x;
    const long fold_blocks = (long)fx * fy * fz;
    if (b < fold_blocks) {
        const int bx = (int)(b % fx), by = (int)((b / fx) % fy), bz = (int)(b / ((long)fx * fy));
        fold_all_body(g, axes, r_lo, r_hi, left_own, bx, by, bz);
        return;
    }
    b -= fold_blocks;
    const int s = (int)(b / (2L * chunks));
    if (s >= nspecies) return;
    const long r = b - (long)s * 2 * chunks;
    const int face = (int)(r / chunks);
    const UnpackOne &u = us.u[s];
    migrate_unpack_body(u.p, u.kg, u.ntiles, u.fs, u.first_slot, u.area_cap, u.cursor, face == 0 ? u.buf_lo : u.buf_hi, cap,
                        face == 0 ? shift_lo : shift_hi, r - (long)face * chunks);
}

// geometry of the free-slot lookup of an unpack (fs == NULL: arrival area only)
static int unpack_geom(const lpa_particles *p, const lpa_grid *g, const lpa_tiling *t, const lpa_free_slots *fs, KeyGeom *kg,
                       int *ntiles) {
    *kg = KeyGeom{};
    *ntiles = 0;
    if (fs) {
        LPA_REQUIRE(g && g->nx > 0 && g->ny > 0 && g->dx > 0 && g->dy > 0 && t && t->tiles_x > 0 && t->tiles_y > 0 &&
                        free_slots_ok(fs, t), "lpai_migrate_unpack2: bad grid / tiling / free-slot stacks");
        const int dim = t->tiles_z > 0 ? 3 : 2;
        LPA_REQUIRE(dim == 2 || (p->z && g->nz > 1 && g->dz > 0), "lpai_migrate_unpack2: 3-D tiling needs z");
        kg->dim = dim; kg->nx = g->nx; kg->ny = g->ny; kg->nz = dim == 3 ? g->nz : 1;
        kg->tiles_y = t->tiles_y; kg->tiles_z = dim == 3 ? t->tiles_z : 1;
        kg->x0 = g->x0; kg->y0 = g->y0; kg->z0 = g->z0;
        kg->inv_dx = 1.0 / g->dx; kg->inv_dy = 1.0 / g->dy; kg->inv_dz = dim == 3 ? 1.0 / g->dz : 0.0;
        *ntiles = t->tiles_x * t->tiles_y * (dim == 3 ? t->tiles_z : 1);
    } else {
        kg->dim = p->z ? 3 : 2;
    }
    return LPA_OK;
}

int lpai_fold_unpack(const lpa_grid *g, int axes, const double *r_lo, const double *r_hi, const double *left_own,
                     const lpa_unpack_args *u, int nspecies, int64_t capacity, double shift_lo, double shift_hi, void *stream) {
    LPA_REQUIRE(g && g->jx && g->jy && g->jz && g->rho && g->nx > 0 && g->ny > 0 && g->ng > 0, "lpai_fold_unpack: bad grid");
    const int dim = g->nz > 1 ? 3 : 2;
    LPA_REQUIRE(g->nx >= 2 * g->ng && g->ny >= 2 * g->ng && (dim == 2 || g->nz >= 2 * g->ng), "lpai_fold_unpack: slab thinner than 2*ng");
    LPA_REQUIRE(!((r_lo || r_hi) && (axes & 1)) && (!left_own || r_lo), "lpai_fold_unpack: bad face planes");
    LPA_REQUIRE(u && nspecies >= 1 && nspecies <= LPA_FOLD_UNPACK_MAX_SPECIES && capacity > 0, "lpai_fold_unpack: bad species");
    GridV v;
    memset(&v, 0, sizeof v);
    v.nx = g->nx; v.ny = g->ny; v.nz = dim == 3 ? g->nz : 1; v.ng = g->ng;
    v.NX = g->nx + 2 * g->ng; v.NY = g->ny + 2 * g->ng; v.NZ = dim == 3 ? g->nz + 2 * g->ng : 1;
    v.jx = g->jx; v.jy = g->jy; v.jz = g->jz; v.rho = g->rho;
    const int fx = dim == 3 ? (v.NZ + 255) / 256 : (v.NY + 255) / 256, fy = dim == 3 ? v.NY : v.NX, fz = dim == 3 ? v.NX : 1;
    UnpackSet us;
    memset(&us, 0, sizeof us);
    for (int s = 0; s < nspecies; s++) {
        const lpa_unpack_args &a = u[s];
        const lpa_particles *p = a.p;
        LPA_REQUIRE(p && p->x && p->y && p->ux && p->uy && p->uz && p->inv_gamma && p->w && a.buf_lo && a.buf_hi && a.cursor &&
                        a.first_slot >= 0 && a.area_capacity >= 0, "lpai_fold_unpack: bad species arguments");
        UnpackOne &o = us.u[s];
        if (int e = unpack_geom(p, g, a.t, a.fs, &o.kg, &o.ntiles)) return e;
        o.p = make_partv(p);
        o.fs = make_free_slots(a.fs, a.t);
        o.first_slot = (long)a.first_slot; o.area_cap = (long)a.area_capacity; o.cursor = a.cursor;
        o.buf_lo = a.buf_lo; o.buf_hi = a.buf_hi;
    }
    const int chunks = (int)((capacity + 255) / 256);
    const long blocks = (long)fx * fy * fz + 2L * chunks * nspecies;
    LPA_REQUIRE(blocks < (1L << 31), "lpai_fold_unpack: grid too large");
    hipLaunchKernelGGL(k_fold_unpack, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, v, axes, r_lo, r_hi, left_own,
                       fx, fy, fz, us, nspecies, (long)capacity, chunks, shift_lo, shift_hi);
    LPA_CHECK_LAUNCH("lpai_fold_unpack");
    return LPA_OK;
}

int lpai_migrate_unpack2(const lpa_particles *p, const lpa_grid *g, const lpa_tiling *t, const lpa_free_slots *fs,
                         int64_t first_slot, int64_t area_capacity, int32_t *cursor, const double *buf_lo,
                         const double *buf_hi, int64_t capacity, double shift_lo, double shift_hi, void *stream) {
    LPA_REQUIRE(p && p->x && p->y && p->ux && p->uy && p->uz && p->inv_gamma && p->w && buf_lo && buf_hi && cursor &&
                    capacity > 0 && first_slot >= 0 && area_capacity >= 0,
                "lpai_migrate_unpack2: bad args");
    if (area_capacity == 0 && !fs) return LPA_OK;
    KeyGeom kg{};
    int ntiles = 0;
    if (int e = unpack_geom(p, g, t, fs, &kg, &ntiles)) return e;
    hipLaunchKernelGGL(k_migrate_unpack2, dim3((unsigned)((capacity + 255) / 256), 2), dim3(256), 0, (hipStream_t)stream,
                       make_partv(p), kg, ntiles, make_free_slots(fs, t), (long)first_slot, (long)area_capacity, cursor,
                       buf_lo, buf_hi, (long)capacity, shift_lo, shift_hi);
    LPA_CHECK_LAUNCH("lpai_migrate_unpack2");
    return LPA_OK;
}

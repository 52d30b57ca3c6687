// lpa_migrate.hpp -- the device side of the leaver pack (slab migration along x), shared by the pack kernels of
// lpa_sort.hip and the rest launch of the push kernels (lpa_particles.hip / lpa_particles3d.hip), which packs the leavers
// of a step in the same launch that pushes the overflow list and the arrival area.
#pragma once
#include "lpa_common.hpp"

// Free-slot stacks: a particle that leaves a tile-ordered store through an x face frees a slot of an edge
// tile, and about as many particles ARRIVE in that tile through the same face.  The pack kernel records the
// freed slots per edge tile, the unpack kernel hands them to the arrivals of that tile: those sit inside their
// tile's range again and take the LDS-tiled push, instead of waiting in the arrival area -- pushed one by one
// through global memory -- for the next sort (0.34 ms per step on the 3-D slab).
struct FreeSlots {
    int32_t *count;   // [2 * edge_tiles]
    int32_t *slot;    // [2 * edge_tiles][depth]
    int edge_tiles;   // tiles in the edge columns of ONE face
    int depth;
};

// edge index of a tile: low-face columns first, then the high-face columns; -1 for interior tiles
__device__ __forceinline__ int edge_index(int tile, int ntiles, int edge_tiles) {
    if (tile < edge_tiles) return tile;
    if (tile >= ntiles - edge_tiles) return tile - (ntiles - 2 * edge_tiles);
    return -1;
}

__device__ __forceinline__ void migrate_pack_one(const PartV &p, long ip, double xlo, double xhi,
                                                 double *buf_lo, double *buf_hi, long cap,
                                                 const FreeSlots &fs = FreeSlots{nullptr, nullptr, 0, 0},
                                                 const int32_t *tile_off = nullptr, int ntiles = 0,
                                                 long n_sorted = 0, bool active = true,
                                                 int32_t *surplus = nullptr, bool have_x = false, double x_in = 0.0,
                                                 int known_tile = -1) {
    // called by every lane of the wave (`active` = this lane has a particle): the message slots are taken
    // with ONE atomic per wave and face -- tens of thousands of leavers bumping a single counter one by one
    // took 0.14 ms of the 3-D scan
    double x = have_x ? x_in : (active ? p.x[ip] : 0.0);      // (have_x: the caller loaded a batch of positions up front)
    const bool live = active && !((p.dead && p.dead[ip]) || isnan(x));
    const int side = !live ? -1 : (x < xlo ? 0 : (x > xhi ? 1 : -1));
    const int lane = (int)(threadIdx.x & 63u);
    long slot = -1;
#pragma unroll
    for (int s = 0; s < 2; s++) {
        const unsigned long long m = __ballot(side == s);
        if (!m) continue;   // wave-uniform
        const int leader = __ffsll((long long)m) - 1;
        unsigned long long base = 0;
        if (lane == leader) {
            const unsigned long long k = (unsigned long long)__popcll(m);
            base = atomicAdd((unsigned long long *)(s == 0 ? buf_lo : buf_hi), k);
            // leavers that do not fit into the message: counted for the host (checked at the next sort)
            if (surplus && base + k > (unsigned long long)cap)
                atomicAdd(surplus, (int32_t)(base >= (unsigned long long)cap ? k : base + k - (unsigned long long)cap));
        }
        base = __shfl(base, leader);
        if (side == s) slot = (long)base + __popcll(m & ((1ull << lane) - 1ull));
    }
    if (side < 0) return;
    double *b = side == 0 ? buf_lo : buf_hi;
    if (slot < cap) {
        double *d = b + 1;
        d[0 * cap + slot] = x;
        d[1 * cap + slot] = p.y[ip];
        d[2 * cap + slot] = p.z ? p.z[ip] : 0.0;
        d[3 * cap + slot] = p.ux[ip];
        d[4 * cap + slot] = p.uy[ip];
        d[5 * cap + slot] = p.uz[ip];
        d[6 * cap + slot] = p.ig[ip];
        d[7 * cap + slot] = p.w[ip];
        d[8 * cap + slot] = p.id ? __longlong_as_double((long long)p.id[ip]) : 0.0;
        // the particle now belongs to the neighbour (sync_particles_2d.c:185-202)
        p.x[ip] = __longlong_as_double(0x7ff8000000000000ll);
        p.y[ip] = __longlong_as_double(0x7ff8000000000000ll);
        if (p.dead) p.dead[ip] = 1;
        if (fs.count && ip < n_sorted) {   // the freed slot belongs to the tile whose range holds it
            int lo = 0, hi = ntiles;       // last tile with tile_off[tile] <= ip
            if (known_tile >= 0) lo = known_tile, hi = known_tile + 1;      // (the tiled push kernels say which)
            while (hi - lo > 1) {
                int mid = (lo + hi) >> 1;
                if ((long)tile_off[mid] <= ip) lo = mid; else hi = mid;
            }
            int e = edge_index(lo, ntiles, fs.edge_tiles);
            if (e >= 0) {
                int k = atomicAdd(&fs.count[e], 1);
                if (k < fs.depth) fs.slot[(long)e * fs.depth + k] = (int32_t)ip;
                else atomicSub(&fs.count[e], 1);
            }
        }
    }
    // slot >= cap: the particle is NOT lost -- it stays where it is (outside the slab, handled by the
    // torus path) and leaves at the next step; the event is counted in *surplus, which the engines read
    // at their next sort and turn into an error (the torus path deposits such a particle on the wrong side
    // of the slab).
}

// ---- the leaver pack inside the push kernels' rest launch -------------------------------------------------------------------
// The launch that pushes what the tiled kernel left to global memory (overflow list + arrival area) also packs the step's
// leavers: its last `pack_blocks` blocks walk the list the TILED kernel wrote (complete when this launch starts), the
// others push their particles and pack those of them that left on the spot (their positions are in registers) -- one
// launch instead of two behind the tiled kernel.
struct PackArgsV {
    const unsigned long long *list;     // the tiled kernels' leaver list (lpa_push_params.leavers); NULL: no pack role
    const uint32_t *list_count;
    long list_cap;
    const int32_t *tile_off;
    int ntiles;
    long n_sorted;
    double xlo, xhi;
    double *buf_lo, *buf_hi;
    long cap;
    FreeSlots fs;
    int32_t *surplus;
    int pack_blocks;
};

// UPDATE: double(long ip) -- pushes particle ip and returns its new x (NaN: nothing there)
template <class UPDATE>
__device__ __forceinline__ void rest_pack_body(const PartV &p, const uint32_t *__restrict__ list,
                                               const uint32_t *__restrict__ list_count, long loose_first, long loose_count,
                                               const int32_t *__restrict__ loose_limit, const PackArgsV &pk, UPDATE update) {
    const long lane = threadIdx.x & 63u;
    const long rest_blocks = (long)gridDim.x - pk.pack_blocks;
    if ((long)blockIdx.x >= rest_blocks) {      // pack role
        long total = *pk.list_count;
        if (total > pk.list_cap) total = pk.list_cap;
        const long b = (long)blockIdx.x - rest_blocks;
        for (long t0 = b * blockDim.x + threadIdx.x - lane; t0 < total; t0 += (long)pk.pack_blocks * blockDim.x) {
            const long t = t0 + lane;
            const bool active = t < total;
            const unsigned long long ent = active ? pk.list[t] : 0ull;
            const long ip = (long)(ent & 0xffffffffull);
            const int tile = (int)(ent >> 32) - 1;
            migrate_pack_one(p, ip, pk.xlo, pk.xhi, pk.buf_lo, pk.buf_hi, pk.cap, pk.fs, pk.tile_off, pk.ntiles, pk.n_sorted,
                             active && ip < p.n, pk.surplus, false, 0.0, tile < pk.ntiles ? tile : -1);
        }
        return;
    }
    // rest role (wave-uniform trip counts: the slot allocation of the pack is a wave-wide operation)
    const long stride = rest_blocks * blockDim.x, t0 = (long)blockIdx.x * blockDim.x + threadIdx.x - lane;
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    if (list) {
        const long n = *list_count;
        for (long tw = t0; tw < n; tw += stride) {
            const bool active = tw + lane < n;
            const long ip = active ? (long)list[tw + lane] : 0;
            const double x = active ? update(ip) : nan;
            migrate_pack_one(p, ip, pk.xlo, pk.xhi, pk.buf_lo, pk.buf_hi, pk.cap, pk.fs, pk.tile_off, pk.ntiles, pk.n_sorted,
                             active, pk.surplus, true, x);
        }
    }
    long m = loose_count;
    if (loose_limit && (long)*loose_limit < m) m = *loose_limit;
    for (long tw = t0; tw < m; tw += stride) {
        const bool active = tw + lane < m;
        const long ip = loose_first + (active ? tw + lane : 0);
        const double x = active ? update(ip) : nan;
        migrate_pack_one(p, ip, pk.xlo, pk.xhi, pk.buf_lo, pk.buf_hi, pk.cap, pk.fs, pk.tile_off, pk.ntiles, pk.n_sorted, active,
                         pk.surplus, true, x);
    }
}

// ---- host side ----------------------------------------------------------------------------------------------------------
static int free_slots_ok(const lpa_free_slots *fs, const lpa_tiling *t) {
    return fs && fs->count && fs->slot && fs->depth > 0 && fs->edge_cols >= 1 && t &&
           2 * fs->edge_cols <= t->tiles_x;
}

static FreeSlots make_free_slots(const lpa_free_slots *fs, const lpa_tiling *t) {
    FreeSlots f{nullptr, nullptr, 0, 0};
    if (fs) {
        f.count = fs->count; f.slot = fs->slot; f.depth = fs->depth;
        f.edge_tiles = fs->edge_cols * t->tiles_y * (t->tiles_z > 0 ? t->tiles_z : 1);
    }
    return f;
}


// the pack that rides in a rest launch (lpai_push_deposit_rest_*): `with_list` != 0 -> the launch also walks the leaver list
// the tiled kernel of the same store wrote (at most ONE launch per store and step may do that)
struct lpa_pack_args {
    const uint64_t *list;
    const uint32_t *list_count;
    int64_t list_capacity;
    const lpa_tiling *t;
    double xlo, xhi;
    double *buf_lo, *buf_hi;
    int64_t capacity;
    const lpa_free_slots *fs;
    int32_t *surplus;
    int with_list;
};

static int pack_args_ok(const lpa_pack_args *a, const lpa_particles *p) {
    return a && a->buf_lo && a->buf_hi && a->capacity > 0 && a->xlo < a->xhi &&
           (!a->with_list || (a->list && a->list_count && a->list_capacity > 0)) &&
           (!a->fs || (a->t && a->t->tile_off && free_slots_ok(a->fs, a->t) && a->t->n_sorted >= 0 && a->t->n_sorted <= p->n));
}

static PackArgsV make_pack_args_v(const lpa_pack_args *a) {
    PackArgsV v;
    const lpa_tiling *t = a->t;
    v.list = a->with_list ? (const unsigned long long *)a->list : nullptr;
    v.list_count = a->list_count; v.list_cap = (long)a->list_capacity;
    // (the tile of a freed slot is only looked up for the free-slot stacks)
    v.tile_off = a->fs ? t->tile_off : nullptr;
    v.ntiles = a->fs ? t->tiles_x * t->tiles_y * (t->tiles_z > 0 ? t->tiles_z : 1) : 0;
    v.n_sorted = a->fs ? (long)t->n_sorted : 0L;
    v.xlo = a->xlo; v.xhi = a->xhi; v.buf_lo = a->buf_lo; v.buf_hi = a->buf_hi; v.cap = (long)a->capacity;
    v.fs = make_free_slots(a->fs, t);
    v.surplus = a->surplus;
    v.pack_blocks = a->with_list ? 32 : 0;
    return v;
}

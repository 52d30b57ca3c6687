#!/usr/bin/env python3
"""CPU model of K1-3D's LDS conflicts: what the particle order and the image strides cost, per wave instruction.

Rules taken from the hardware (profiles/r02_ubench_lds.txt, MI355X_MICROARCH.md section LDS):
  * ds_add_f64: four groups of 16 lanes; 16 double-wide banks (address in doubles mod 16); the lanes of a group that
    fall on one bank are served one after the other, same address or not.
  * ds_read_b64: two groups of 32 lanes; 32 double-wide banks; equal addresses broadcast, DISTINCT addresses on one
    bank are served one after the other.
The model draws a thermal plasma (tools/bench3d.py's numbers), sorts it the way lpa_sort_tiles_3d does (striped or
padded stripes of a 4 x 4 x 16 tile), lets it drift `age` steps and reports the mean cost of an atomic of the
3 x 3 x 3 deposit and of a gather read, relative to the conflict-free instruction.

    python tools/lds_model.py [--ppc 8] [--ages 0 1 5 9] [--r3zs 24] [--ebsy 21] [--ebsx 189]
"""
import argparse
import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument("--ppc", type=int, default=8)
ap.add_argument("--ages", type=int, nargs="*", default=[0, 1, 3, 5, 9])
ap.add_argument("--r3zs", type=int, default=24)      # J image z stride (R3Y = 10 rows per x plane)
ap.add_argument("--ebsy", type=int, default=21)      # E/B image strides
ap.add_argument("--ebsx", type=int, default=189)
ap.add_argument("--tiles", type=int, nargs=3, default=[2, 4, 2])
ap.add_argument("--pad-min", type=int, default=192)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--orders", nargs="*", default=["striped", "padded"],
                help="striped / padded (the sort's orders) / column (model only: stripes over the 16 z-cells of one "
                     "(x, y) column at a time, rank after rank, columns one after the other)")
a = ap.parse_args()

T = np.array([4, 4, 16])
ncell = T * np.array(a.tiles)
rng = np.random.default_rng(a.seed)
n = int(np.prod(ncell)) * a.ppc
cell0 = np.arange(n) // a.ppc
pos = np.stack([cell0 // (ncell[1] * ncell[2]), (cell0 // ncell[2]) % ncell[1], cell0 % ncell[2]], 1).astype(float)
pos += rng.random((n, 3)) - 0.5
# per-step displacement in cells: beta c dt / d, uth = 0.0442, c dt = 0.95 / sqrt(sum d^-2), d = lambda / (20, 10, 10)
cdt = 0.95 / np.sqrt(20.0 ** 2 + 10.0 ** 2 + 10.0 ** 2)
vel = rng.standard_normal((n, 3)) * 0.0442 * cdt * np.array([20.0, 10.0, 10.0])
# let the plasma mix first so that the per-cell counts are Poisson like in a running simulation
pos = (pos + 40 * vel + 0.5) % ncell - 0.5


def order_of(sort_pos, padded, column=False):
    """returns, per tile, the list of particle indices in slot order (-1 = hole)"""
    node = np.floor(sort_pos + 0.5).astype(int) % ncell
    tile = node // T
    tid = (tile[:, 0] * a.tiles[1] + tile[:, 1]) * a.tiles[2] + tile[:, 2]
    loc = node - tile * T
    cid = (loc[:, 0] * 4 + loc[:, 1]) * 16 + loc[:, 2]          # z fastest
    out = []
    for t in range(int(np.prod(a.tiles))):
        idx = np.nonzero(tid == t)[0]
        c = cid[idx]
        o = np.argsort(c, kind="stable")
        idx, c = idx[o], c[o]
        first = np.searchsorted(c, c, side="left")
        rank = np.arange(len(c)) - first
        if column:
            o2 = np.lexsort((c % 16, rank, c // 16))          # column, then rank, then z
            out.append((t, idx[o2]))
            continue
        slots = []
        for r in range(rank.max() + 1):
            m = rank == r
            if padded and m.sum() >= a.pad_min:
                s = np.full(256, -1)
                s[c[m]] = idx[m]
                slots.append(s)
            else:
                slots.append(idx[m])
        out.append((t, np.concatenate(slots)))
    return out


def group_cost(bank, addr, active, width, same_addr_broadcast):
    """bank/addr/active: [nwaves, 64]; mean over groups with at least one active lane of max lanes per bank"""
    nb = {16: 16, 32: 32}[width]
    costs = []
    b = bank.reshape(-1, width); ad = addr.reshape(-1, width); ac = active.reshape(-1, width)
    for g in range(b.shape[0]):
        if not ac[g].any():
            continue
        if same_addr_broadcast:
            pairs = np.unique(np.stack([b[g][ac[g]], ad[g][ac[g]]], 1), axis=0)
            cnt = np.bincount(pairs[:, 0], minlength=nb)
        else:
            cnt = np.bincount(b[g][ac[g]], minlength=nb)
        costs.append(cnt.max())
    return float(np.mean(costs)), len(costs)


def run(kind):
    padded = kind == "padded"
    orders = order_of(pos, padded, kind == "column")
    print(f"--- {kind} order, {a.ppc} ppc, J z stride {a.r3zs}, E/B strides {a.ebsx}/{a.ebsy}")
    for age in a.ages:
        p_mid = pos + (age + 0.5) * vel              # position the gather sees (half step after `age` full steps)
        p_new = pos + (age + 1.0) * vel
        tot = {"atomic": [0.0, 0], "g": [0.0, 0], "hx": [0.0, 0], "hy": [0.0, 0], "hz": [0.0, 0], "hxy": [0.0, 0]}
        stay = 0; alive = 0; slots_n = 0
        for t, sl in orders:
            t3 = np.array([t // (a.tiles[1] * a.tiles[2]), (t // a.tiles[2]) % a.tiles[1], t % a.tiles[2]]) * T
            k = len(sl); pad = (-k) % 64
            sl = np.concatenate([sl, np.full(pad, -1)])
            hole = sl < 0
            i = np.where(hole, 0, sl)
            # tile-relative coordinates (unwrap through the periodic box)
            rel = (p_mid[i] - t3 + ncell / 2) % ncell - ncell / 2
            reln = (p_new[i] - t3 + ncell / 2) % ncell - ncell / 2
            n1 = np.floor(rel + 0.5).astype(int); n2 = np.floor(rel).astype(int)
            nn = np.floor(reln + 0.5).astype(int)
            inside = np.all((n1 >= -1) & (n1 <= T), 1) & ~hole       # LPA_TILE3_MARGIN = 1
            stayer = inside & np.all(nn == n1, 1)
            alive += int((~hole).sum()); stay += int(stayer.sum()); slots_n += len(sl)
            # deposit: base node = n1 - 1, region origin = tile - 3
            b = n1 - 1 + 3
            addr = (b[:, 0] * 10 + b[:, 1]) * a.r3zs + b[:, 2]
            c, m = group_cost((addr % 16).reshape(-1, 64), addr.reshape(-1, 64), stayer.reshape(-1, 64), 16, False)
            tot["atomic"][0] += c * m; tot["atomic"][1] += m
            # gather: image origin = tile - 3; component bases use n1 or n2 per axis
            for name, sx, sy, sz in (("g", 0, 0, 0), ("hx", 1, 0, 0), ("hy", 0, 1, 0), ("hz", 0, 0, 1), ("hxy", 1, 1, 0)):
                lx = (n2 if sx else n1)[:, 0] + 3; ly = (n2 if sy else n1)[:, 1] + 3; lz = (n2 if sz else n1)[:, 2] + 3
                ad = lx * a.ebsx + ly * a.ebsy + lz
                c, m = group_cost((ad % 32).reshape(-1, 64), ad.reshape(-1, 64), inside.reshape(-1, 64), 32, True)
                tot[name][0] += c * m; tot[name][1] += m
        s = " ".join(f"{k_}={v[0] / max(v[1], 1):.2f}" for k_, v in tot.items())
        print(f"age {age}: stayers {stay / alive:.3f} of alive, lanes used {alive / slots_n:.3f}; cost x ideal: {s}")


for kind in a.orders:
    run(kind)

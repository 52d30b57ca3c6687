"""oracle/cpu_ratio.py -- TEST INFRASTRUCTURE ONLY, build container only (needs oracle/_ref).

Times the reference's own compiled fused kernel (unified_boris_pusher_cpu_2d, built in place from
/root/reference by `make -C oracle ref`) and the oracle's restatement on the same inputs and threads, so that
the `cpu_baseline` of bench.py (kind "port", timed on the GPU box where the reference cannot travel) can be
read as a reference-equivalent number (SURVEY.md 8d ii).

    python -m oracle.cpu_ratio [cells] [ppc] [steps]
"""
import os
import sys
import time

import numpy as np

import oracle
from oracle import driver
from lambdapic_amd.patch import make_patches_2d

C = 299792458.0


def main():
    cells = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    ppc = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    assert oracle.ref_available(), "run `make -C oracle ref` first (build container only)"
    ref = oracle.ref_module("pusher", "unified_pusher_2d")
    lam = 0.8e-6
    dx = dy = lam / 20
    dt = 0.95 / (C * np.sqrt(dx ** -2 + dy ** -2))
    q, m = -oracle.E_CHARGE, oracle.M_E
    n_c = oracle.EPSILON_0 * m * (2 * np.pi * C / lam) ** 2 / q ** 2
    out = {}
    for name in ("reference", "port"):
        P = make_patches_2d(cells, cells, dx, dy, cells // 32, cells // 32)
        driver.load_uniform_plasma(P, 0, ppc, n_c, 0.0442, np.random.default_rng(1))
        fl = [p.fields for p in P]
        parts = [p.particles[0] for p in P]
        for f in fl:                      # some field so that the push does work
            f.ez[...] = 1e9
            f.bz[...] = 10.0
        n = sum(p.npart for p in parts)
        L = oracle.lib(native=True)
        threads = int(L.orc_num_threads())
        fn = (lambda: ref.unified_boris_pusher_cpu_2d(parts, fl, P.npatches, dt, q, m)) if name == "reference" \
            else (lambda: oracle.unified_boris_pusher_cpu_2d(parts, fl, P.npatches, dt, q, m, native=True))
        fn()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        el = time.perf_counter() - t0
        out[name] = n * steps / el
        print(f"{name:9s}: {out[name]:.3e} particle-updates/s (fused push+deposit only, {n} particles, "
              f"{os.cpu_count()} cpus, {threads} OpenMP threads)")
    import hashlib
    from pathlib import Path
    sha = hashlib.sha256((Path(__file__).resolve().parent / "picoracle.c").read_bytes()).hexdigest()[:16]
    print(f"oracle/picoracle.c sha256[:16] = {sha}")
    print(f"port / reference = {out['port'] / out['reference']:.3f}")


if __name__ == "__main__":
    main()

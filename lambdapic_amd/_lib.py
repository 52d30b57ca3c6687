"""ctypes binding of the C ABI declared in include/lambdapic_amd.h.

The HIP library is the product: there is no CPU fallback.  ``lib()`` raises if
``liblambdapic_amd.so`` is missing or a symbol the header declares is not exported.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

HERE = Path(__file__).resolve().parent
import os as _os
# LPA_LIB_PATH selects another build of the SAME library (profiling / ablation builds)
LIB_PATH = Path(_os.environ.get("LPA_LIB_PATH", HERE / "liblambdapic_amd.so"))

LPA_TILE_X = 8
LPA_MAX_STRIPE_RANKS = 16384      # what lpa_tiling.stripe_ranks may ask for (the default rule stops at 1024)
LPA_TILE_Y = 32
LPA_ORDER_CELL_MAJOR = 0
LPA_ORDER_STRIPED = 1
LPA_ORDER_PADDED = 2
LPA_ORDER_COLUMN = 3
LPA_TILE_MARGIN = 2
LPA_TILE3_X, LPA_TILE3_Y, LPA_TILE3_Z, LPA_TILE3_MARGIN = 4, 4, 16, 1
LPA_MIG_NATTR = 9
LPA_HALO_PACK_GUARD_SRC, LPA_HALO_UNPACK_GUARD, LPA_HALO_PACK_CURRENT, LPA_HALO_UNPACK_CURRENT = 0, 1, 2, 3
LPA_PART_ALL, LPA_PART_EDGE, LPA_PART_INTERIOR = 0, 1, 2
LPA_ABSORB_X = 16
LPA_PUSH_NO_RHO = 1
LPA_PUSH_NO_IG = 2


class LpaError(RuntimeError):
    pass


class lpa_grid(C.Structure):
    _fields_ = [("nx", C.c_int32), ("ny", C.c_int32), ("nz", C.c_int32), ("ng", C.c_int32),
                ("dx", C.c_double), ("dy", C.c_double), ("dz", C.c_double),
                ("x0", C.c_double), ("y0", C.c_double), ("z0", C.c_double)] + \
               [(n, C.c_void_p) for n in ("ex", "ey", "ez", "bx", "by", "bz", "jx", "jy", "jz", "rho")]


class lpa_particles(C.Structure):
    _fields_ = [("n", C.c_int64)] + \
               [(n, C.c_void_p) for n in ("x", "y", "z", "ux", "uy", "uz", "inv_gamma", "w")] + \
               [("part_eb", C.c_void_p * 6), ("id", C.c_void_p), ("is_dead", C.c_void_p)]


class lpa_tiling(C.Structure):
    _fields_ = [("tiles_x", C.c_int32), ("tiles_y", C.c_int32), ("n_sorted", C.c_int64),
                ("max_blocks", C.c_int32), ("order", C.c_int32),
                ("tile_off", C.c_void_p), ("blk_tile", C.c_void_p), ("blk_begin", C.c_void_p),
                ("blk_end", C.c_void_p), ("n_blocks", C.c_void_p),
                ("tiles_z", C.c_int32), ("prefix_hint", C.c_int32), ("scratch", C.c_void_p * 8),
                ("pad_ranks", C.c_void_p), ("slot_class", C.c_void_p), ("aux_slot", C.c_void_p), ("aux_info", C.c_void_p),
                ("reloc_stats", C.c_void_p), ("class_init", C.c_int32), ("stripe_ranks", C.c_int32)]


class lpa_cpml_axis(C.Structure):
    _fields_ = [("kappa", C.c_void_p), ("bcoeff", C.c_void_p), ("ccoeff_d", C.c_void_p),
                ("lo0", C.c_int32), ("lo1", C.c_int32), ("hi0", C.c_int32), ("hi1", C.c_int32),
                ("psi_a_lo", C.c_void_p), ("psi_b_lo", C.c_void_p), ("psi_a_hi", C.c_void_p),
                ("psi_b_hi", C.c_void_p)]


class lpa_push_params(C.Structure):
    _fields_ = [("dt", C.c_double), ("q", C.c_double), ("m", C.c_double), ("wrap", C.c_int32), ("flags", C.c_int32),
                ("lo", C.c_double * 3), ("hi", C.c_double * 3),
                ("alo", C.c_double * 3), ("ahi", C.c_double * 3),
                ("absorbed", C.c_void_p), ("absorbed_count", C.c_void_p), ("absorbed_capacity", C.c_int64),
                ("leavers", C.c_void_p), ("leaver_count", C.c_void_p), ("leaver_capacity", C.c_int64),
                ("leave_lo", C.c_double), ("leave_hi", C.c_double), ("absorbed_spill", C.c_void_p)]



class lpa_free_slots(C.Structure):
    _fields_ = [("count", C.c_void_p), ("slot", C.c_void_p), ("edge_cols", C.c_int32), ("depth", C.c_int32)]


class lpa_face_msg(C.Structure):
    _fields_ = [("send_lo", C.c_void_p), ("send_hi", C.c_void_p), ("recv_lo", C.c_void_p), ("recv_hi", C.c_void_p),
                ("n_send_lo", C.c_int64), ("n_send_hi", C.c_int64), ("n_recv_lo", C.c_int64), ("n_recv_hi", C.c_int64)]


class lpa_step_migrate(C.Structure):
    _fields_ = [("s_lo", C.c_void_p), ("s_hi", C.c_void_p), ("r_lo", C.c_void_p), ("r_hi", C.c_void_p),
                ("cursor", C.c_void_p), ("surplus", C.c_void_p), ("fs", C.POINTER(lpa_free_slots)),
                ("area_capacity", C.c_int64), ("edge_cols", C.c_int32), ("reserved_", C.c_int32),
                ("overflow_edge", C.c_void_p), ("overflow_edge_count", C.c_void_p), ("ev_edge_start", C.c_void_p),
                ("ev_edge_stop", C.c_void_p)]


class lpa_step_species(C.Structure):
    _fields_ = [("p", lpa_particles), ("t", C.POINTER(lpa_tiling)), ("n_sorted", C.c_int64), ("pp", lpa_push_params),
                ("overflow", C.c_void_p), ("overflow_count", C.c_void_p), ("ev_start", C.c_void_p), ("ev_stop", C.c_void_p),
                ("mig", lpa_step_migrate)]


class lpa_step_slab(C.Structure):
    _fields_ = [("comm", C.c_void_p), ("xlo", C.c_double), ("xhi", C.c_double), ("shift_lo", C.c_double),
                ("shift_hi", C.c_double), ("migrate_capacity", C.c_int64), ("cur_r_lo", C.c_void_p),
                ("cur_r_hi", C.c_void_p), ("jx_left_plane", C.c_void_p), ("rho_exchange", C.c_int32),
                ("overlap_cols", C.c_int32)]


class lpa_step_desc(C.Structure):
    _fields_ = [("grid", lpa_grid), ("dim", C.c_int32), ("local_axes", C.c_int32), ("dt", C.c_double), ("eps0", C.c_double),
                ("e_axes", C.POINTER(lpa_cpml_axis) * 3), ("b_axes", C.POINTER(lpa_cpml_axis) * 3),
                ("nspecies", C.c_int32), ("continuity", C.c_int32), ("fuse_species", C.c_int32), ("flags", C.c_int32),
                ("species", C.POINTER(lpa_step_species)),
                ("absorbed", C.c_void_p), ("absorbed_count", C.c_void_p), ("absorbed_capacity", C.c_int64),
                ("slab", C.POINTER(lpa_step_slab)), ("absorbed_spill", C.c_void_p)]


LPA_COMM_RCCL, LPA_COMM_LOOPBACK = 1, 2
LPA_STEP_DEFER_E2_GUARDS, LPA_STEP_DEFER_E2, LPA_STEP_E1_DOUBLE = 1, 2, 4
LPA_STEP_B_EXT_LO, LPA_STEP_B_EXT_HI, LPA_STEP_E_ROUND_IN_LINE, LPA_STEP_SEPARATE_UNPACK, LPA_STEP_SEPARATE_PACK, LPA_STEP_SEPARATE_TAILS = 8, 16, 32, 64, 128, 256
LPA_STAGE_E1, LPA_STAGE_B1, LPA_STAGE_RESET, LPA_STAGE_PUSH, LPA_STAGE_FOLD, LPA_STAGE_B2, LPA_STAGE_B2_GUARD, \
    LPA_STAGE_E2 = range(8)

_G, _P, _T, _PP = C.POINTER(lpa_grid), C.POINTER(lpa_particles), C.POINTER(lpa_tiling), C.POINTER(lpa_push_params)
_FS = C.POINTER(lpa_free_slots)
_vp, _d, _i, _i64 = C.c_void_p, C.c_double, C.c_int, C.c_int64

# name -> (restype, argtypes); every symbol include/lambdapic_amd.h declares
SIGNATURES = {
    "lpa_last_error": (C.c_char_p, []),
    "lpa_version": (_i, []),
    "lpa_fdtd_e_2d": (_i, [_G, _d, _d, _vp]),
    "lpa_fdtd_b_2d": (_i, [_G, _d, _vp]),
    "lpa_fdtd_e_3d": (_i, [_G, _d, _d, _vp]),
    "lpa_fdtd_b_3d": (_i, [_G, _d, _vp]),
    "lpa_fdtd_e_cpml_2d": (_i, [_G, _d, _d, _vp, _vp, _vp]),
    "lpa_fdtd_b_cpml_2d": (_i, [_G, _d, _vp, _vp, _vp]),
    "lpa_cpml_psi_2d": (_i, [_G, _i, _i, _i, _i, _d, _vp, _vp, _vp, _vp, _vp]),
    "lpa_laser_inject_2d": (_i, [_G, _i, _d, _d, _i, _i, _vp, _vp, _vp]),
    "lpa_laser_inject_sep_2d": (_i, [_G, _i, _d, _d, _i, _i, _vp, _vp, C.POINTER(C.c_double), _vp]),
    "lpa_fdtd_e_cpml_fused_2d": (_i, [_G, _d, _d, C.POINTER(lpa_cpml_axis), C.POINTER(lpa_cpml_axis), _vp]),
    "lpa_fdtd_b_cpml_fused_2d": (_i, [_G, _d, C.POINTER(lpa_cpml_axis), C.POINTER(lpa_cpml_axis), _vp]),
    "lpa_fdtd_e_cpml_fused_3d": (_i, [_G, _d, _d] + [C.POINTER(lpa_cpml_axis)] * 3 + [_vp]),
    "lpa_fdtd_b_cpml_fused_3d": (_i, [_G, _d] + [C.POINTER(lpa_cpml_axis)] * 3 + [_vp]),
    "lpa_fdtd_e_cpml_3d": (_i, [_G, _d, _d, _vp, _vp, _vp, _vp]),
    "lpa_fdtd_b_cpml_3d": (_i, [_G, _d, _vp, _vp, _vp, _vp]),
    "lpa_cpml_psi_3d": (_i, [_G, _i, _i, _i, _i, _d, _vp, _vp, _vp, _vp, _vp]),
    "lpa_laser_inject_3d": (_i, [_G, _i, _d, _d, _i, _i, _i, _i, _vp, _vp, _vp]),
    "lpa_laser_inject_sep_3d": (_i, [_G, _i, _d, _d, _i, _i, _i, _i, _vp, _vp, C.POINTER(C.c_double), _vp]),
    "lpa_reset_current": (_i, [_G, _vp]),
    "lpa_reset_j": (_i, [_G, _vp]),
    "lpa_rho_continuity": (_i, [_G, _d, _i, _i, _vp, _vp]),
    "lpa_rho_absorbed": (_i, [_G, _vp, _vp, _i64, _vp]),
    "lpa_rho_absorbed_spill": (_i, [_G, _vp, _vp, _i64, _vp, _vp]),
    "lpa_guard_wrap": (_i, [_G, _i, _i, _vp]),
    "lpa_current_fold": (_i, [_G, _i, _vp]),
    "lpa_halo_pack_guard_src": (_i, [_G, _i, _i, _vp, _vp]),
    "lpa_halo_unpack_guard": (_i, [_G, _i, _i, _vp, _vp]),
    "lpa_halo_pack_current": (_i, [_G, _i, _vp, _vp]),
    "lpa_halo_unpack_current": (_i, [_G, _i, _vp, _vp]),
    "lpa_halo_faces": (_i, [_G, _i, _i, _vp, _vp, _vp]),
    "lpa_push_deposit_2d": (_i, [_G, _P, _PP, _i64, _i64, _vp]),
    "lpa_push_deposit_tiled_2d": (_i, [_G, _P, _PP, _T, _vp, _vp, _vp]),
    "lpa_push_deposit_tiled_part_2d": (_i, [_G, _P, _PP, _T, _vp, _vp, _i, _i, _vp]),
    "lpa_push_deposit_list_2d": (_i, [_G, _P, _PP, _vp, _vp, _i64, _vp]),
    "lpa_push_deposit_3d": (_i, [_G, _P, _PP, _i64, _i64, _vp]),
    "lpa_push_deposit_tiled_3d": (_i, [_G, _P, _PP, _T, _vp, _vp, _vp]),
    "lpa_push_deposit_tiled_part_3d": (_i, [_G, _P, _PP, _T, _vp, _vp, _i, _i, _vp]),
    "lpa_push_deposit_list_3d": (_i, [_G, _P, _PP, _vp, _vp, _i64, _vp]),
    "lpa_push_deposit_tiled_multi_3d": (_i, [_G, C.c_int32, C.POINTER(_P), C.POINTER(_PP), C.POINTER(_T),
                                            C.POINTER(_vp), C.POINTER(_vp), _vp]),
    "lpa_interpolate_2d": (_i, [_G, _P, _vp]),
    "lpa_boris": (_i, [_P, _d, _d, _d, _vp]),
    "lpa_push_position_2d": (_i, [_P, _d, _vp]),
    "lpa_deposit_2d": (_i, [_G, _P, _d, _d, _vp]),
    "lpa_wrap_positions_2d": (_i, [_P, _PP, _vp]),
    "lpa_wrap_positions_3d": (_i, [_P, _PP, _vp]),
    "lpa_interpolate_3d": (_i, [_G, _P, _vp]),
    "lpa_deposit_3d": (_i, [_G, _P, _d, _d, _vp]),
    "lpa_sort_workspace_bytes": (_i64, [_G, _i64]),
    "lpa_sort_stripe_ranks": (C.c_int32, [_G, _i64]),
    "lpa_sort_workspace_bytes_ranks": (_i64, [_G, _i64, C.c_int32]),
    "lpa_sort_deepest_cell": (_vp, [_vp]),
    "lpa_sort_tiles_2d": (_i, [_G, _P, _P, _vp, _i64, C.c_int32, C.c_int32, _T, _vp]),
    "lpa_sort_tiles_3d": (_i, [_G, _P, _P, _vp, _i64, C.c_int32, C.c_int32, _T, _vp]),
    "lpa_sort_tiles_ahead_2d": (_i, [_G, _P, _P, _vp, _i64, C.c_int32, C.c_int32, _T, _d, _vp]),
    "lpa_sort_tiles_ahead_3d": (_i, [_G, _P, _P, _vp, _i64, C.c_int32, C.c_int32, _T, _d, _vp]),
    "lpa_sort_live_count": (_vp, [_vp]),
    "lpa_sort_overflow": (_vp, [_vp]),
    "lpa_migrate_pack_x": (_i, [_P, _d, _d, _vp, _vp, _i64, _vp, _vp]),
    "lpa_migrate_pack_edges_x": (_i, [_P, _T, C.c_int32, _d, _d, _vp, _vp, _i64, _FS, _vp, _vp]),
    "lpa_migrate_pack_list": (_i, [_P, _T, _vp, _vp, _i64, _d, _d, _vp, _vp, _i64, _FS, _vp, _vp]),
    "lpa_migrate_unpack_tiled": (_i, [_P, _G, _T, _FS, _i64, _i64, _vp, _vp, _i64, _d, _vp]),
    "lpa_migrate_unpack": (_i, [_P, _i64, _i64, _vp, _vp, _i64, _d, _vp]),
    "lpa_sync_guard_fields_2d": (_i, [_vp, C.c_int32, _vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _vp]),
    "lpa_sync_currents_2d": (_i, [_vp, _vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _vp]),
    "lpa_sync_particles_count_2d": (_i, [_vp, _vp, _vp, _vp, C.c_int32, _i64, _vp, _vp, _vp]),
    "lpa_sync_particles_workspace_bytes": (_i64, [C.c_int32, _i64]),
    "lpa_sync_particles_fill_2d": (_i, [_vp, C.c_int32, C.c_int32, C.c_int32, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int32,
                                        _i64, _d, _d, _d, _d, _d, _d, _vp, _i64, _vp]),
    "lpa_sync_guard_fields_3d": (_i, [_vp, C.c_int32, _vp] + [C.c_int32] * 5 + [_vp]),
    "lpa_sync_currents_3d": (_i, [_vp, _vp] + [C.c_int32] * 5 + [_vp]),
    "lpa_sync_particles_count_3d": (_i, [_vp, _vp, _vp, _vp, C.c_int32, _i64, _vp, _vp, _vp]),
    "lpa_sync_particles_fill_3d": (_i, [_vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _vp, _vp, _vp, _vp, _vp, _vp,
                                        C.c_int32, _i64, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                        C.POINTER(C.c_double), _vp, _i64, _vp]),
    "lpa_bucket_sort_workspace_bytes": (_i64, [_i64, _i64]),
    "lpa_bucket_sort": (_i, [_vp, _vp, _vp, _vp, C.POINTER(_vp), C.c_int32, _i64, _i64, _i64, _i64, _d, _d, _d,
                             _d, _d, _d, C.c_int32, _vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    "lpa_comm_unique_id": (_i, [_vp, C.c_char_p]),
    "lpa_comm_create_rccl": (_i, [C.POINTER(_vp), _vp, C.c_int32, C.c_int32, C.c_int32, C.c_char_p]),
    "lpa_comm_create_loopback": (_i, [C.POINTER(_vp), C.c_int32, C.c_int32]),
    "lpa_comm_destroy": (_i, [_vp]),
    "lpa_comm_info": (_i, [_vp, C.POINTER(C.c_int32)]),
    "lpa_comm_exchange": (_i, [_vp, C.POINTER(lpa_face_msg), C.c_int32, _vp]),
    "lpa_step": (_i, [C.POINTER(lpa_step_desc), _i, _i, _vp]),
    "lpa_diag_fields": (_i, [_G, _d, _d, _vp, _vp]),
    "lpa_diag_particles": (_i, [_P, _d, _vp, _vp]),
    "lpa_refresh_inv_gamma": (_i, [_P, _i64, _i64, _vp]),
    "lpa_selftest_wave_reduce": (_i, [_vp, _vp, _vp]),
    "lpa_selftest_wave_shift": (_i, [_vp, _vp, _vp]),
}

_LIB = None
_VARIANTS = None
VARIANTS_PATH = HERE / "csrc" / "build" / "liblambdapic_amd_variants.so"


def _bind(path):
    # torch first: it brings its own HIP runtime (libamdhip64), and the library must resolve its HIP symbols against THAT
    # one -- loaded before torch, the library binds the system runtime and the process ends up with two, of which ours
    # sees no device ("no ROCm-capable device is detected" from the first hipMemsetAsync)
    import torch  # noqa: F401
    L = C.CDLL(str(path))
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(L, name)
        except AttributeError as e:
            raise LpaError(f"{path} does not export {name}") from e
        fn.restype, fn.argtypes = res, args
    return L


class use_variants:
    """context manager: inside it ``lib()`` returns the VARIANTS build of the library (same ABI, plus the three
    measured-slower deposit paths of the 2-D tiled kernel: wave reduce-scatter for CELL_MAJOR stores, in-kernel
    re-seating, cooperative deposit on PADDED stores).  Engines created inside keep that handle.  Test / A-B use only."""

    def __enter__(self):
        global _LIB, _VARIANTS
        if _VARIANTS is None:
            if not VARIANTS_PATH.exists():
                raise LpaError(f"{VARIANTS_PATH} not built: run `python -m lambdapic_amd.build`")
            _VARIANTS = _bind(VARIANTS_PATH)
        self._saved, _LIB = _LIB, _VARIANTS
        return _VARIANTS

    def __exit__(self, *exc):
        global _LIB
        _LIB = self._saved
        return False


def lib():
    """Load liblambdapic_amd.so and bind every declared symbol; fail loudly otherwise."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not LIB_PATH.exists():
        raise LpaError(f"{LIB_PATH} not built: run `python -m lambdapic_amd.build` "
                       "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    _LIB = _bind(LIB_PATH)
    return _LIB


def sort_result(L, wsbuf, with_deepest=False):
    """live count (slots, for a padded order) of the tile sort that just ran through workspace ``wsbuf`` (a device
    uint8 tensor); raises when the device refused the sort (``lpa_sort_overflow``).  One host sync.
    ``with_deepest``: (live count, particles in the deepest cell, particles beyond the striped ranks, tiles in use,
    work blocks) from the same read-back."""
    ptr = wsbuf.data_ptr()
    hdr = wsbuf[:64].view(_torch().int32).tolist()
    ovf = hdr[(L.lpa_sort_overflow(ptr) - ptr) // 4]
    if ovf:
        raise LpaError("tile sort refused (nothing was moved): " +
                       ("the sorted order needs more slots than the destination holds -- a padded order stores up to "
                        "4/3 n + 64 slots per tile" if ovf & 1 else
                        "work-block table too small" if ovf & 2 else
                        "lpa_tiling.prefix_hint is not confirmed by the workspace header (the source is not the previous "
                        "sort's result)"))
    live = hdr[(L.lpa_sort_live_count(ptr) - ptr) // 4]
    k = (L.lpa_sort_deepest_cell(ptr) - ptr) // 4
    return (live, hdr[k], hdr[k + 1], hdr[k + 2], hdr[1]) if with_deepest else live


def _torch():
    import torch
    return torch


def check(status: int, what: str = ""):
    if status != 0:
        msg = lib().lpa_last_error()
        raise LpaError(f"{what} failed ({status}): {msg.decode() if msg else ''}")

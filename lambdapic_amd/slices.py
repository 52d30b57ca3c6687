"""``np.s_``-style sub-domain selections of the diagnostics callbacks (`callback/hdf5.py:14-160` in the reference: same
accepted forms, same errors, same text form), applied to an x-slab of the box instead of to a patch list.

A selection is normalised to one ``slice(start, stop, step)`` per axis with 0 <= start < stop <= n and step >= 1; a
rank then takes the part of it that falls into its slab ON THE DEVICE (a strided view of the resident array), so only
the selected values cross to the host.
"""
from __future__ import annotations

import numpy as np


def normalize_slice(ndim, user, dims):
    """``None`` -> ``None``; ints become one-element slices (negative ints count from the end), open ends are filled
    in, slices are clamped to the axis.  ``ValueError``: Ellipsis / ``None`` entries, wrong number of axes, an index
    outside the axis, a step <= 0, an empty range, any other entry type."""
    if user is None:
        return None
    if isinstance(user, (slice, int, np.integer)):
        user = (user,)
    user = tuple(user)
    if any(u is Ellipsis for u in user):
        raise ValueError("Ellipsis (...) is not supported in a slice specification")
    if any(u is None for u in user):
        raise ValueError("None / np.newaxis is not supported in a slice specification")
    if len(user) != ndim:
        raise ValueError(f"the slice has {len(user)} axes, the simulation {ndim}")
    out = []
    for axis, (u, n) in enumerate(zip(user, dims)):
        if isinstance(u, (int, np.integer)):
            i = int(u) + (n if u < 0 else 0)
            if not 0 <= i < n:
                raise ValueError(f"index {int(u)} is outside axis {axis} of size {n}")
            out.append(slice(i, i + 1, 1))
        elif isinstance(u, slice):
            step = 1 if u.step is None else int(u.step)
            if step <= 0:
                raise ValueError(f"the step of a slice must be positive, not {step}")
            lo = 0 if u.start is None else int(u.start) + (n if u.start < 0 else 0)
            hi = n if u.stop is None else int(u.stop) + (n if u.stop < 0 else 0)
            lo, hi = min(max(lo, 0), n), min(max(hi, 0), n)
            if lo >= hi:
                raise ValueError(f"slice {u} selects nothing on axis {axis} of size {n}")
            out.append(slice(lo, hi, step))
        else:
            raise ValueError(f"a slice entry must be an int or a slice, not {type(u).__name__}")
    return tuple(out)


def slice_text(norm, dims):
    """the ``slice`` attribute of the output files: ``[:, 5]``, ``[::2, ::3]``, ``[16:, :, :]``"""
    parts = []
    for s, n in zip(norm, dims):
        if (s.start, s.stop, s.step) == (0, n, 1):
            parts.append(":")
        elif s.step == 1 and s.stop == s.start + 1:
            parts.append(str(s.start))
        else:
            lo = str(s.start) if s.start else ""
            hi = str(s.stop) if s.stop != n else ""
            parts.append(f"{lo}:{hi}" + (f":{s.step}" if s.step != 1 else ""))
    return "[" + ", ".join(parts) + "]"


def selected_shape(norm):
    return tuple(len(range(s.start, s.stop, s.step)) for s in norm)


def part_in_range(s, offset, size):
    """the members of ``range(s.start, s.stop, s.step)`` inside [offset, offset + size):
    (slice local to the range's origin, index of the first one in the output, how many) or ``None``"""
    k0 = max(0, -(-(offset - s.start) // s.step))
    first = s.start + k0 * s.step
    end = min(s.stop, offset + size)
    if first >= end:
        return None
    count = (end - 1 - first) // s.step + 1
    lo = first - offset
    return slice(lo, lo + (count - 1) * s.step + 1, s.step), k0, count


def slab_selection(sim, norm):
    """this rank's share of a normalised selection (``None`` = the whole box): (index into the slab's interior
    array, index into the output array, output shape) -- the first two ``None`` when the slab holds nothing of it"""
    dims = (sim.nx, sim.ny) + ((sim.nz,) if sim.dimension == 3 else ())
    if norm is None:
        norm = tuple(slice(0, n, 1) for n in dims)
    nx_loc = sim.nx // sim.comm.size
    shape = selected_shape(norm)
    px = part_in_range(norm[0], sim.comm.rank * nx_loc, nx_loc)
    if px is None:
        return None, None, shape
    local = (px[0],) + tuple(norm[1:])
    out = (slice(px[1], px[1] + px[2]),) + tuple(slice(0, n) for n in shape[1:])
    return local, out, shape

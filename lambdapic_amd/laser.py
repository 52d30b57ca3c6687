"""Laser injection callbacks (stage ``_laser``), host side.

Mirrors of the reference's ``SimpleLaser2D`` (`callback/laser.py:272-391`), ``GaussianLaser2D``
(`callback/laser.py:397-555`, Laguerre-Gaussian modes included) and of ``laser1 + laser2``
(`callback/laser.py:139-151,240-270`): same constructor parameters, same source-field formulas; the boundary update itself runs on the device
(``PicEngine2D.laser_inject`` -> ``lpa_laser_inject_2d``, the GPU twin of
``_update_laser_bfields_2d``, `callback/laser.py:17-46`).  Being device native, the callback does not
trigger the host-mirror refresh that ordinary callbacks get.
"""
from __future__ import annotations

import numpy as np
import torch

from . import constants

C = constants.C_LIGHT


def _xp(a):
    """numpy for host arrays (tests, golden rows), torch for device tensors: the callbacks evaluate the
    source rows ON THE DEVICE -- a host array would cost a blocking host-to-device copy per step, which
    serialises the host with everything queued on the GPU"""
    return torch if isinstance(a, torch.Tensor) else np


def _polyval(coeffs, x):
    """Horner evaluation of a polynomial (highest power first) for numpy arrays or torch tensors"""
    acc = x * 0 + float(coeffs[0])
    for c in coeffs[1:]:
        acc = acc * x + float(c)
    return acc


# ---- factorised sources: one launch per '_laser' stage ---------------------------------------------------------------
# Evaluated as torch expressions a profile is ~17 launches of 4 us per step -- 0.14 ms, a third of a C3 step.  The
# profiles of the reference are amp(y[, z]) T(t) times sin / cos of theta(t) + phi(y[, z]) whenever the temporal envelope is
# the same on the whole boundary (Gaussian / Laguerre-Gaussian beams; the plane-front pulse at normal incidence), so
#     ey = k0 pc + k1 ps,  ez = k2 pc + k3 ps,   pc = amp cos(phi), ps = amp sin(phi)  (fixed arrays),
# with four numbers per step; lpa_laser_inject_sep_2d/3d evaluates that inside the injection kernel.  pc, ps are not
# re-derived here: they are solved from the class' own source_fields (the formulas pinned by tests/test_laser_profiles.py)
# at the envelope's peak with (pol_angle, ellipticity) = (0, 1), where ey = amp sin(phase) / sqrt 2 and
# ez = amp cos(phase) / sqrt 2; the factorisation is then checked against source_fields at another time and the laser
# falls back to the general path if it does not reproduce it (oblique incidence: the envelope varies along y).
class _SimAt:
    """what source_fields reads of a simulation, at another time"""

    def __init__(self, sim, time):
        self.time, self.cpml_thickness, self.dx = time, sim.cpml_thickness, sim.dx


def _sep_coefficients(T, theta, pol_angle, ellipticity):
    """k0 .. k3 of the factorised sources (the algebra of _polarise on sin / cos of theta + phi)"""
    norm = float(np.sqrt(1 + ellipticity ** 2))
    major, minor = 1.0 / norm, ellipticity / norm
    cp, sp = float(np.cos(pol_angle)), float(np.sin(pol_angle))
    a, b, c, d = major * cp, minor * sp, major * sp, minor * cp
    st, ct = float(np.sin(theta)), float(np.cos(theta))
    return (T * (a * st - b * ct), T * (a * ct + b * st), T * (c * st + d * ct), T * (c * ct - d * st))


def _factorise(laser, sim, coords):
    """(pc, ps) device tensors of ``laser`` on the boundary ``coords``, or None when its sources do not factorise"""
    import copy
    tf = getattr(laser, "_time_factors", None)
    if tf is None or not getattr(laser, "factorise", True):
        return None
    t_pk = laser._t_peak()
    T, theta = tf(sim, t_pk)
    if not T > 0.5:
        return None
    probe = copy.copy(laser)
    probe.pol_angle, probe.ellipticity = 0.0, 1.0
    e_s, e_c = probe.source_fields(_SimAt(sim, t_pk), *coords)
    if e_s is None:
        return None
    a_s, a_c = e_s * float(np.sqrt(2.0) / T), e_c * float(np.sqrt(2.0) / T)      # amp sin / cos (theta + phi)
    st, ct = float(np.sin(theta)), float(np.cos(theta))
    pc, ps = (a_c * ct + a_s * st).contiguous(), (a_s * ct - a_c * st).contiguous()
    # does it reproduce the profile at another moment of the pulse?
    t2 = t_pk * 0.83 + 0.37 * 2 * np.pi / laser.omega0
    ey, ez = laser.source_fields(_SimAt(sim, t2), *coords)
    if ey is None:
        return None
    k = _sep_coefficients(*tf(sim, t2), laser.pol_angle, laser.ellipticity)
    scale = float(max(ey.abs().max(), ez.abs().max(), pc.abs().max()))
    err = float(max((k[0] * pc + k[1] * ps - ey).abs().max(), (k[2] * pc + k[3] * ps - ez).abs().max()))
    return (pc, ps) if err <= 1e-10 * scale else None


def _inject(laser, sim, coords, key):
    """one '_laser' stage: the factorised form when the profile allows it, else the general one"""
    eng = sim.engine
    # (the factorisation depends on every parameter of the beam: a laser mutated between two runs must not keep it)
    key = key + tuple(sorted((k, v) for k, v in vars(laser).items() if isinstance(v, (int, float, bool, str, type(None)))
                             and not k.startswith("_")))
    if getattr(laser, "_sep_key", None) != key:
        laser._sep, laser._sep_key = _factorise(laser, sim, coords), key
    if laser._sep is not None:
        k = _sep_coefficients(*laser._time_factors(sim, sim.time), laser.pol_angle, laser.ellipticity)
        eng.laser_inject_sep(laser._sep[0], laser._sep[1], k, sim.dt)
        return
    ey, ez = laser.source_fields(sim, *coords)
    if ey is not None:
        eng.laser_inject(ey, ez, sim.dt)


def _simple_time_factors(self, sim, time):
    """SimpleLaser (callback/laser.py:351-386) at normal incidence: sin^2 envelope, plane phase"""
    if self.angle_y != 0:
        return 0.0, 0.0
    ct = C * time
    return (float(np.sin(ct / (2 * self.ctau) * np.pi) ** 2) if ct < 2 * self.ctau else 0.0), self.omega0 * time + self.cep


def _gaussian_time_factors(self, sim, time):
    """GaussianLaser (callback/laser.py:504-555): Gaussian envelope; the phase carries the propagation to the injection
    plane and the Gouy shift"""
    x_rel = sim.cpml_thickness * sim.dx
    psi = self.beam_params(x_rel)[2]
    return (float(np.exp(-(C * time - self.x0) ** 2 / self.ctau ** 2)),
            self.omega0 * time + self.cep - self.k0 * x_rel - (2 * self.p + abs(self.l) + 1) * psi)


class _LaserBase:
    """shared state of the 2-D and 3-D callbacks"""
    stage = "_laser"
    interval = 1
    device_native = True      # touches device state through the engine; no mirror download/upload
    side = "xmin"
    y0 = None
    z0 = None
    tstop = np.inf
    disabled = False


class Laser2D:
    """common part (`callback/laser.py:79-137,154-192`): stage, stop time, the x-min layer check and
    the hand-over of the source rows to the device kernel"""
    stage = "_laser"
    interval = 1
    device_native = True      # touches device state through the engine; no mirror download/upload
    side = "xmin"
    y0 = None
    tstop = np.inf
    disabled = False

    def boundary_y(self, sim):
        """f.yaxis - dy/2 - y0 on the interior nodes (`callback/laser.py:166-172`; ``y0 or Ly/2``
        as there: y0 = 0 means the box centre)"""
        eng = sim.engine
        return eng.y0 + np.arange(eng.ny) * eng.dy - eng.dy / 2 - (self.y0 or sim.Ly / 2)

    def source_fields(self, sim, y):
        """(ey_source, ez_source) on the boundary nodes ``y`` (centred on the beam axis) at
        ``sim.time``, or (None, None) once the pulse is over"""
        raise NotImplementedError

    def __call__(self, sim):
        if self.disabled:
            return
        if C * sim.time >= self.tstop:          # callback/laser.py:113-117
            self.disabled = True
            return
        eng = sim.engine
        # the reference disables a laser whose x-min layer is missing (never there, or removed
        # when a MovingWindow started): callback/laser.py:119-128
        if eng.bc["xmin"] != "pml" or (eng.comm.rank == 0 and (eng.pml is None or "xmin" not in eng.pml.sides)):
            self.disabled = True
            return
        y = self.boundary_y_device(sim)
        if y is None:                        # a sum of lasers: each evaluates on its own coordinates
            ey, ez = self.source_fields(sim, None)
            if ey is not None:
                eng.laser_inject(ey, ez, sim.dt)
            return
        _inject(self, sim, (y,), (id(eng), sim.Ly, sim.cpml_thickness, sim.dx, self.pol_angle, self.ellipticity))

    def boundary_y_device(self, sim):
        key = (id(sim.engine), sim.Ly)
        if getattr(self, "_ydev_key", None) != key:
            y = self.boundary_y(sim)
            self._ydev = None if y is None else torch.from_numpy(np.ascontiguousarray(y)).to(sim.engine.device)
            self._ydev_key = key
        return self._ydev

    def __add__(self, other):
        if not isinstance(other, Laser2D):
            raise TypeError(f"Cannot add Laser with {type(other)}")
        if self.side != other.side:
            raise TypeError(f"Cannot add lasers from different sides: {self.side} and {other.side}")
        return CombinedLaser2D(self, other)


def _polarise(amp, phase, pol_angle, ellipticity):
    """major/minor axis decomposition shared by both profiles (`callback/laser.py:372-383`)"""
    xp = _xp(phase)
    norm = float(np.sqrt(1 + ellipticity ** 2))
    major, minor = 1.0 / norm, ellipticity / norm
    cp, sp = float(np.cos(pol_angle)), float(np.sin(pol_angle))
    return (amp * (major * cp * xp.sin(phase) - minor * sp * xp.cos(phase)),
            amp * (major * sp * xp.sin(phase) + minor * cp * xp.cos(phase)))


class CombinedLaser2D(Laser2D):
    """sum of two sources (`callback/laser.py:240-270`)"""

    def __init__(self, laser1, laser2):
        self.laser1, self.laser2 = laser1, laser2
        self.side = laser1.side
        self.tstop = max(laser1.tstop, laser2.tstop)

    def boundary_y(self, sim):
        return None

    def boundary_y_device(self, sim):
        return None

    def source_fields(self, sim, y):
        dev = hasattr(sim, "engine")
        a = self.laser1.source_fields(sim, self.laser1.boundary_y_device(sim) if dev else self.laser1.boundary_y(sim))
        b = self.laser2.source_fields(sim, self.laser2.boundary_y_device(sim) if dev else self.laser2.boundary_y(sim))
        if a[0] is None:
            return b
        if b[0] is None:
            return a
        return a[0] + b[0], a[1] + b[1]


class SimpleLaser2D(Laser2D):
    def __init__(self, a0, w0, ctau, y0=None, angle_y=0.0, tstop=None, pol_angle=0.0, ellipticity=0.0,
                 cep=0.0, l0=0.8e-6, side="xmin"):
        if any(p <= 0 for p in (a0, l0, w0, ctau)):
            raise ValueError("All parameters (a0, l0, w0, ctau) must be positive")
        if side != "xmin":
            raise NotImplementedError("Invalid side: only 'xmin' is supported.")
        if abs(angle_y) >= np.pi / 2:
            raise ValueError("Angle_y must be in range (-pi/2, pi/2)")
        if abs(ellipticity) > 1:
            raise ValueError("Ellipticity must be in range [-1, 1]")
        self.a0, self.l0, self.w0, self.ctau, self.y0 = a0, l0, w0, ctau, y0
        self.omega0 = 2 * np.pi * C / l0
        self.angle_y, self.pol_angle, self.ellipticity, self.cep = angle_y, pol_angle, ellipticity, cep
        self.tstop = 2 * ctau if tstop is None else C * tstop
        self.E0 = a0 * constants.M_E * C * self.omega0 / constants.E_CHARGE
        self.k0 = self.omega0 / C
        self.ky = self.k0 * np.sin(angle_y)

    def source_fields(self, sim, y):
        """`callback/laser.py:351-386`"""
        time = sim.time
        if C * time >= self.tstop:
            return None, None
        xp = _xp(y)
        r_rot = xp.sqrt((y / float(np.cos(self.angle_y))) ** 2)
        transverse_phase = -(float(self.ky) * y)
        t_rot = C * time - y * float(np.sin(self.angle_y))
        tprof = xp.sin(t_rot / (2 * self.ctau) * np.pi) ** 2 * (t_rot < 2 * self.ctau)
        amp = float(self.E0) * xp.exp(-r_rot ** 2 / float(self.w0 ** 2)) * tprof
        phase = float(self.omega0 * time + self.cep) + transverse_phase
        ey, ez = _polarise(amp, phase, self.pol_angle, self.ellipticity)
        return ey * float(np.cos(self.angle_y)), ez


class GaussianLaser2D(Laser2D):
    """Gaussian beam evaluated at the injection plane: waist evolution, wavefront curvature, Gouy
    phase, Gaussian envelope centred ``x0`` behind the boundary, optional Laguerre-Gaussian (l, p)
    mode (`callback/laser.py:397-555`)"""

    def __init__(self, a0, l0, w0, ctau, x0=None, y0=None, z0=None, tstop=None, pol_angle=0.0, ellipticity=0.0,
                 cep=0.0, focus_position=0.0, side="xmin", l=0, p=0):
        if any(par <= 0 for par in (a0, l0, w0, ctau)):
            raise ValueError("All parameters (a0, l0, w0, ctau) must be positive")
        if side != "xmin":
            raise ValueError("Invalid side: only 'xmin' is implemented.")
        if abs(ellipticity) > 1:
            raise ValueError("Ellipticity must be in range [-1, 1]")
        if not isinstance(p, int) or p < 0:
            raise ValueError("Number of radial nodes p must be a non-negative integer")
        if not isinstance(l, int):
            raise ValueError("Azimuthal index l must be an integer")
        self.a0, self.l0, self.w0, self.ctau, self.y0, self.z0 = a0, l0, w0, ctau, y0, z0
        self.omega0 = 2 * np.pi * C / l0
        self.k0 = self.omega0 / C
        self.x0 = 3 * ctau if x0 is None else x0
        self.tstop = 6 * ctau if tstop is None else C * tstop
        self.E0 = a0 * constants.M_E * C * self.omega0 / constants.E_CHARGE
        self.pol_angle, self.ellipticity, self.cep = pol_angle, ellipticity, cep
        self.focus_position = focus_position
        self.zR = np.pi * w0 ** 2 / l0
        self.l, self.p = l, p
        self._is_lg = l != 0 or p > 0
        if self._is_lg:
            from scipy.special import factorial, genlaguerre
            # normalised so that the fundamental mode has unit norm
            self.lg_norm = np.sqrt(2 * factorial(p) / (np.pi * factorial(p + abs(l)))) / np.sqrt(2 / np.pi)
            self.laguerre = genlaguerre(p, abs(l))

    def beam_params(self, z):
        z = z - self.focus_position
        w = self.w0 * np.sqrt(1 + (z / self.zR) ** 2)
        R = z * (1 + (self.zR / z) ** 2) if abs(z) > 1e-10 else np.inf
        return w, R, np.arctan(z / self.zR)

    def source_fields(self, sim, y):
        time = sim.time
        if C * time >= self.tstop:
            return None, None
        tprof = float(np.exp(-(C * time - self.x0) ** 2 / self.ctau ** 2))
        x_rel = sim.cpml_thickness * sim.dx
        w, R, psi = self.beam_params(x_rel)
        xp = _xp(y)
        r = xp.abs(y)
        if self._is_lg:
            phi = xp.arctan2(y * 0, y)           # 2-D: the azimuth is 0 or pi
            u = float(np.sqrt(2)) * r / float(w)
            amp_lg = float(self.lg_norm) * u ** abs(self.l) * _polyval(self.laguerre.coeffs, u ** 2)
            phase_lg = self.l * phi
        else:
            amp_lg, phase_lg = 1.0, 0.0
        amp = float(self.E0 * (self.w0 / w)) * xp.exp(-r ** 2 / float(w ** 2)) * amp_lg * tprof
        phase = (float(self.omega0 * time + self.cep - self.k0 * x_rel - (2 * self.p + abs(self.l) + 1) * psi)
                 - float(self.k0 / (2 * R)) * r ** 2 - phase_lg)
        return _polarise(amp, phase, self.pol_angle, self.ellipticity)


# ---- 3-D (callback/laser.py:194-238: r and phi from the y-z plane of the boundary) ----------------------
class Laser3D(_LaserBase):
    """callable ``laser(engine3d, dt)`` for ``PicEngine3D.step(dt, laser=...)`` and, as a callback object,
    ``laser(sim)`` for a driver exposing ``sim.engine`` (a PicEngine3D), ``sim.time``, ``sim.dt``,
    ``sim.Ly``, ``sim.Lz``, ``sim.dx``, ``sim.cpml_thickness``"""

    def boundary_yz(self, sim):
        """(y, z, r): f.yaxis - dy/2 - y0, f.zaxis - dz/2 - z0 on the interior nodes, [ny][nz]"""
        eng = sim.engine
        y = (np.arange(eng.n[1]) * eng.d[1] - eng.d[1] / 2 - (self.y0 or sim.Ly / 2))[:, None]
        z = (np.arange(eng.n[2]) * eng.d[2] - eng.d[2] / 2 - (self.z0 or sim.Lz / 2))[None, :]
        return y + 0 * z, z + 0 * y, np.sqrt(y ** 2 + z ** 2)

    def source_fields(self, sim, y, z, r):
        raise NotImplementedError

    def __call__(self, sim):
        if self.disabled:
            return
        if C * sim.time >= self.tstop:
            self.disabled = True
            return
        eng = sim.engine
        if eng.bc["xmin"] != "pml" or (eng.comm.rank == 0 and (eng.pml is None or "xmin" not in eng.pml.sides)):
            self.disabled = True
            return
        key = (id(eng), sim.Ly, sim.Lz)
        if getattr(self, "_yz_key", None) != key:      # boundary coordinates live on the device
            self._yz = tuple(torch.from_numpy(np.ascontiguousarray(a)).to(eng.device) for a in self.boundary_yz(sim))
            self._yz_key = key
        _inject(self, sim, self._yz, key + (sim.cpml_thickness, sim.dx, self.pol_angle, self.ellipticity))


class SimpleLaser3D(Laser3D):
    """`callback/laser.py:272-391` with the 3-D boundary coordinates (angle_z is not implemented there)"""

    def __init__(self, a0, w0, ctau, y0=None, z0=None, angle_y=0.0, angle_z=0.0, tstop=None, pol_angle=0.0,
                 ellipticity=0.0, cep=0.0, l0=0.8e-6, side="xmin"):
        if angle_z != 0:
            raise NotImplementedError("Angle_z is not implemented")
        SimpleLaser2D.__init__(self, a0, w0, ctau, y0=y0, angle_y=angle_y, tstop=tstop, pol_angle=pol_angle,
                               ellipticity=ellipticity, cep=cep, l0=l0, side=side)
        self.z0 = z0

    def source_fields(self, sim, y, z, r):
        time = sim.time
        if C * time >= self.tstop:
            return None, None
        xp = _xp(y)
        r_rot = xp.sqrt((y / float(np.cos(self.angle_y))) ** 2 + z ** 2)
        transverse_phase = -(float(self.ky) * y)
        t_rot = C * time - y * float(np.sin(self.angle_y))
        tprof = xp.sin(t_rot / (2 * self.ctau) * np.pi) ** 2 * (t_rot < 2 * self.ctau)
        amp = float(self.E0) * xp.exp(-r_rot ** 2 / float(self.w0 ** 2)) * tprof
        phase = float(self.omega0 * time + self.cep) + transverse_phase
        ey, ez = _polarise(amp, phase, self.pol_angle, self.ellipticity)
        return ey * float(np.cos(self.angle_y)), ez


class GaussianLaser3D(Laser3D):
    """`callback/laser.py:397-555` with r, phi taken in the y-z plane (Laguerre-Gaussian vortex phase
    l * atan2(z, y))"""

    def __init__(self, a0, l0, w0, ctau, x0=None, y0=None, z0=None, tstop=None, pol_angle=0.0, ellipticity=0.0,
                 cep=0.0, focus_position=0.0, side="xmin", l=0, p=0):
        GaussianLaser2D.__init__(self, a0, l0, w0, ctau, x0=x0, y0=y0, z0=z0, tstop=tstop, pol_angle=pol_angle,
                                 ellipticity=ellipticity, cep=cep, focus_position=focus_position, side=side,
                                 l=l, p=p)

    beam_params = GaussianLaser2D.beam_params

    def source_fields(self, sim, y, z, r):
        time = sim.time
        if C * time >= self.tstop:
            return None, None
        tprof = float(np.exp(-(C * time - self.x0) ** 2 / self.ctau ** 2))
        x_rel = sim.cpml_thickness * sim.dx
        w, R, psi = self.beam_params(x_rel)
        xp = _xp(y)
        if self._is_lg:
            phi = xp.arctan2(z, y)
            u = float(np.sqrt(2)) * r / float(w)
            amp_lg = float(self.lg_norm) * u ** abs(self.l) * _polyval(self.laguerre.coeffs, u ** 2)
            phase_lg = self.l * phi
        else:
            amp_lg, phase_lg = 1.0, 0.0
        amp = float(self.E0 * (self.w0 / w)) * xp.exp(-r ** 2 / float(w ** 2)) * amp_lg * tprof
        phase = (float(self.omega0 * time + self.cep - self.k0 * x_rel - (2 * self.p + abs(self.l) + 1) * psi)
                 - float(self.k0 / (2 * R)) * r ** 2 - phase_lg)
        return _polarise(amp, phase, self.pol_angle, self.ellipticity)


# the time factors / envelope peaks of the factorised form (see _factorise)
for _cls in (SimpleLaser2D, SimpleLaser3D):
    _cls._time_factors = _simple_time_factors
    _cls._t_peak = lambda self: self.ctau / C
for _cls in (GaussianLaser2D, GaussianLaser3D):
    _cls._time_factors = _gaussian_time_factors
    _cls._t_peak = lambda self: self.x0 / C
del _cls

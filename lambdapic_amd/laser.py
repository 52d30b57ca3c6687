"""Laser injection callbacks (stage ``_laser``), host side.

Mirror of the reference's ``SimpleLaser2D`` (`callback/laser.py:272-391`): same constructor
parameters, same source-field formulas; the boundary update itself runs on the device
(``PicEngine2D.laser_inject`` -> ``lpa_laser_inject_2d``, the GPU twin of
``_update_laser_bfields_2d``, `callback/laser.py:17-46`).  Being device native, the callback does not
trigger the host-mirror refresh that ordinary callbacks get.
"""
from __future__ import annotations

import numpy as np

from . import constants

C = constants.C_LIGHT


class SimpleLaser2D:
    stage = "_laser"
    interval = 1
    device_native = True      # touches device state through the engine; no mirror download/upload

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
        self.disabled = False

    def source_fields(self, time, y):
        """ey_source, ez_source on the boundary nodes ``y`` (already centred on the beam axis) at
        ``time`` -- `callback/laser.py:351-386`"""
        r_rot = np.sqrt((y / np.cos(self.angle_y)) ** 2)
        transverse_phase = -(self.ky * y)
        t_rot = C * time - y * np.sin(self.angle_y)
        tprof = np.sin(t_rot / (2 * self.ctau) * np.pi) ** 2 * (t_rot < 2 * self.ctau)
        amp = self.E0 * np.exp(-r_rot ** 2 / self.w0 ** 2) * tprof
        phase = self.omega0 * time + self.cep + transverse_phase
        norm = np.sqrt(1 + self.ellipticity ** 2)
        major, minor = 1.0 / norm, self.ellipticity / norm
        cp, sp = np.cos(self.pol_angle), np.sin(self.pol_angle)
        ey = amp * (major * cp * np.sin(phase) - minor * sp * np.cos(phase)) * np.cos(self.angle_y)
        ez = amp * (major * sp * np.sin(phase) + minor * cp * np.cos(phase))
        return ey, ez

    def __call__(self, sim):
        if self.disabled:
            return
        if C * sim.time >= self.tstop:          # callback/laser.py:113-117
            self.disabled = True
            return
        eng = sim.engine
        if eng.bc["xmin"] != "pml":             # the reference disables a laser without an x-min PML
            self.disabled = True
            return
        y0 = self.y0 if self.y0 is not None else sim.Ly / 2
        # boundary coordinates: f.yaxis - dy/2 - y0 (callback/laser.py:166-172), interior nodes
        y = eng.y0 + np.arange(eng.ny) * eng.dy - eng.dy / 2 - y0
        ey, ez = self.source_fields(sim.time, y)
        eng.laser_inject(ey, ez, sim.dt)

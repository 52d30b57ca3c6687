"""Physical constants as the reference sees them.

The reference's C kernels hard-code ``LIGHT_SPEED = 299792458.0`` (`core/utils/cutils.h:17`); its
numba kernels and species factories take ``c, epsilon_0, mu_0, e, m_e`` from ``scipy.constants``
(`core/maxwell/cpu.py:3`, `core/species.py:102-103`), so their values follow the installed scipy.
When scipy is importable its values are used (what a λPIC process on this host would see);
otherwise CODATA 2022 as shipped by scipy >= 1.15 -- the values the golden vectors were made with.
"""
C_LIGHT = 299792458.0
try:  # pragma: no cover - depends on the host
    from scipy.constants import e as E_CHARGE, epsilon_0 as EPSILON_0, m_e as M_E, m_p as M_P, mu_0 as MU_0
except Exception:  # scipy absent
    EPSILON_0 = 8.8541878188e-12
    MU_0 = 1.25663706127e-06
    M_E = 9.1093837139e-31
    E_CHARGE = 1.602176634e-19
    M_P = 1.67262192595e-27

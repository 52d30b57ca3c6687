"""Maxwell-Juettner sampler of the SetTemperature callback (`callback/utils.py:988-1049`): the three regimes
against the analytic mean <gamma> = 3 theta + K1(1/theta) / K2(1/theta) and isotropy.  Other random streams than
the reference's numpy generator: statistical check (parity unpinned bit-wise)."""
import numpy as np
import pytest
import torch
from scipy.special import kn

from lambdapic_amd.callbacks import sample_maxwell_juttner


@pytest.mark.parametrize("theta", [0.004, 0.05, 0.4, 2.0])
def test_maxwell_juttner_moments(theta):
    g = torch.Generator().manual_seed(11)
    n = 300_000
    ux, uy, uz = sample_maxwell_juttner(n, theta, g, "cpu")
    assert ux.shape == (n,) and ux.dtype == torch.float64
    gam = torch.sqrt(1 + ux * ux + uy * uy + uz * uz)
    want = 3 * theta + kn(1, 1 / theta) / kn(2, 1 / theta)
    if theta <= 0.01:                       # the reference's non-relativistic branch: gamma - 1 ~ Gamma(3/2, theta)
        want = 1 + 1.5 * theta
    assert float(gam.mean()) - 1 == pytest.approx(want - 1, rel=0.02)
    for u in (ux, uy, uz):                  # isotropic, centred
        assert abs(float(u.mean())) < 5 * float(u.std()) / np.sqrt(n)
    s2 = [float((u * u).mean()) for u in (ux, uy, uz)]
    assert max(s2) / min(s2) < 1.03

"""lambdapic_amd -- MI355X-native PIC inner loop (Boris push, Esirkepov deposition, Yee FDTD,
guard-cell halos, cell sort) behind λPIC's facade/callback interface.

The compute path is the HIP library ``liblambdapic_amd.so`` (C ABI in ``include/lambdapic_amd.h``);
this package is the host-side mirror of the reference interface for that path.
"""
from .fields import Fields2D, Fields3D  # noqa: F401
from .particles import ParticlesBase  # noqa: F401
from .patch import Patch2D, Patches, make_patches_2d  # noqa: F401

__all__ = ["Fields2D", "Fields3D", "ParticlesBase", "Patch2D", "Patches", "make_patches_2d"]

"""Shim so that the unmodified reference files can `import roma` in the build container.
Delegates to oracle/roma_ref.py (restated subset; PARITY UNPINNED)."""
from oracle.roma_ref import (RigidUnitQuat, rigid_points_registration, rotmat_to_unitquat,  # noqa: F401
                             special_procrustes, unitquat_to_rotmat)

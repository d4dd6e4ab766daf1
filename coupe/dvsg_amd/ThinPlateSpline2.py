"""Module-name mirror of the reference's ThinPlateSpline2.py (see ThinPlateSpline.py)."""
from .ThinPlateSpline import ThinPlateSpline2  # noqa: F401

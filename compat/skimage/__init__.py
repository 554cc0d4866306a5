"""stand-in for scikit-image: only `skimage.measure` is imported by the reference (sdf_utils.py:11, mesh extraction);
`measure.marching_cubes` is implemented here (marching tetrahedra) so that the reference's mesh export runs without it"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _defer import become_real  # noqa: E402
_REAL = become_real(__name__, globals())

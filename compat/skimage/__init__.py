"""stand-in for scikit-image: only `skimage.measure` is imported by the reference (sdf_utils.py:11, mesh extraction)"""

"""Constants of nemoflux/geo.py (the array math itself runs in nf_geom.hip)."""
import math

EARTH_RADIUS = 1.0          # geo.py:3  (arc lengths are on the unit sphere)
DEG2RAD = math.pi / 180.    # geo.py:4
EARTH_RADIUS_METRES = 6371000.0  # field.py:12 (Sverdrup scaling only)

"""Station-table reader, same behaviour as nemoflux/latlonreader.py:5-20 (host-side text parsing)."""
import re

import numpy

# station number, distance, LAT, LONG (latitude is column 3, longitude column 4: latlonreader.py:5,16-17)
PAT = re.compile(r'^\s*\d+\s+\d+\.\d+\s+(\-?\d+\.?\d*)\s+(\-?\d+\.?\d*)')


class LatLonReader(object):

    def __init__(self, filename):
        self.lonLatTargets = []
        with open(filename) as f:
            for line in f.readlines():
                m = re.match(PAT, line)
                if m:
                    lat, lon = float(m.group(1)), float(m.group(2))
                    self.lonLatTargets.append((lon, lat))

    def getLonLats(self):
        return numpy.array(self.lonLatTargets)

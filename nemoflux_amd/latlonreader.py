"""Station tables (WOCE-style text files under data/): the lon/lat of every station row, in file order.

Same class surface as the reference's reader (nemoflux/latlonreader.py:5-20: LatLonReader(fileName).getLonLats() ->
(n, 2) array of lon, lat) so that fluxplot's -i option works unchanged.  A station row is four or more blank-separated
fields: station number (digits), distance along the section (unsigned decimal with a point), LATITUDE, LONGITUDE, ...
-- note the file order is lat, lon and the result order lon, lat.  Header, unit and separator lines fail the field test
and are skipped.  tests/test_oracle_golden.py checks the parse of the reference's 12 data files against the
reference's own (tests/golden/stations.json).
"""
import numpy


def _station_row(fields):
    """(lon, lat) of a row of blank-separated fields, or None when the row is not a station row."""
    if len(fields) < 4 or not fields[0].isdigit():
        return None
    whole, point, fraction = fields[1].partition('.')
    if not (point and whole.isdigit() and fraction.isdigit()):
        return None
    try:
        latitude, longitude = float(fields[2]), float(fields[3])
    except ValueError:
        return None
    return longitude, latitude


class LatLonReader(object):

    def __init__(self, filename):
        with open(filename) as table:
            rows = (_station_row(text.split()) for text in table)
            self.lonLatTargets = [row for row in rows if row is not None]

    def getLonLats(self):
        return numpy.array(self.lonLatTargets)

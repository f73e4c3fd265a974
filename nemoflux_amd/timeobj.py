"""Time axis labels: same surface as nemoflux/timeobj.py:4-34 (host-side; SURVEY.md marks it out of the GPU scope).

The reference lets xarray/cftime decode the variable whose `standard_name == 'time'` or `long_name == 'Time axis'`
(timeobj.py:9-13) and formats year-month-day (timeobj.py:24-34).  Here the raw values and the CF `units` / `calendar`
attributes (as read by nemoflux_amd.io) are decoded with plain calendar arithmetic for the calendars NEMO writes:
gregorian / standard / proleptic_gregorian, noleap (365_day), all_leap (366_day) and 360_day.  Files without a time
variable (datagen output) get index labels instead of raising."""
import re
from datetime import date, datetime, timedelta

_UNITS = re.compile(r'^\s*(seconds|second|secs|sec|s|minutes|minute|min|hours|hour|hrs|hr|h|days|day|d)\s+since\s+'
                    r'(-?\d+)-(\d+)-(\d+)(?:[ T](\d+):(\d+):(\d+(?:\.\d*)?))?', re.I)
_SECONDS = {'s': 1., 'm': 60., 'h': 3600., 'd': 86400.}
_NOLEAP = [31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31]


def _as_text(v):
    if isinstance(v, bytes):
        return v.decode('utf-8', 'replace')
    try:
        return v.tobytes().decode('utf-8', 'replace') if hasattr(v, 'tobytes') and v.dtype.kind == 'S' else str(v)
    except Exception:
        return str(v)


class TimeObj(object):

    def __init__(self, values=None, units='', calendar='gregorian', name=''):
        self.timeVarName = name
        self.timeVar = [] if values is None else list(values)
        self._dates = None
        if values is not None and units:
            self._dates = [self._decode(float(x), _as_text(units), _as_text(calendar).lower()) for x in self.timeVar]

    @classmethod
    def fromVariables(cls, variables):
        """variables: {name: array} plus '_attrs_<name>' dicts (nemoflux_amd.io); picks the time variable like
        timeobj.py:9-13."""
        for k, v in variables.items():
            if k.startswith('_'):
                continue
            a = variables.get('_attrs_' + k, {})
            if _as_text(a.get('standard_name', '')) == 'time' or _as_text(a.get('long_name', '')) == 'Time axis':
                return cls(v, a.get('units', ''), a.get('calendar', 'gregorian'), k)
        return cls()

    @staticmethod
    def _decode(x, units, calendar):
        m = _UNITS.match(units)
        if not m:
            return None
        step = _SECONDS[m.group(1).lower()[0]]   # seconds / minutes / hours / days
        y0, mo0, d0 = int(m.group(2)), int(m.group(3)), int(m.group(4))
        sec0 = int(m.group(5) or 0) * 3600 + int(m.group(6) or 0) * 60 + float(m.group(7) or 0)
        secs = x * step + sec0
        if calendar in ('gregorian', 'standard', 'proleptic_gregorian', ''):
            d = datetime(y0, mo0, d0) + timedelta(seconds=secs)
            return d.year, d.month, d.day
        days = int(secs // 86400)
        if calendar in ('360_day',):
            n = (y0 * 12 + (mo0 - 1)) * 30 + (d0 - 1) + days
            return n // 360, (n % 360) // 30 + 1, n % 30 + 1
        months = list(_NOLEAP)
        if calendar in ('all_leap', '366_day'):
            months[1] = 29
        elif calendar not in ('noleap', '365_day'):
            return None
        ylen = sum(months)
        n = y0 * ylen + sum(months[:mo0 - 1]) + (d0 - 1) + days
        y, r = n // ylen, n % ylen
        mo = 0
        while r >= months[mo]:
            r -= months[mo]
            mo += 1
        return y, mo + 1, r + 1

    def getValues(self):
        return self.timeVar[:]

    def getSize(self):
        return len(self.timeVar)

    def getTimeAsDate(self, timeIndex):
        """timeobj.py:24-28 (a datetime.date); the bare index when the file has no decodable time axis."""
        if self._dates and self._dates[timeIndex]:
            y, m, d = self._dates[timeIndex]
            try:
                return date(y, m, d)
            except ValueError:      # e.g. 30 February of a 360-day calendar: no datetime.date exists
                return timeIndex
        return timeIndex

    def getTimeAsString(self, timeIndex):
        """timeobj.py:31-34: f'{year}-{month}-{day}'."""
        if self._dates and self._dates[timeIndex]:
            y, m, d = self._dates[timeIndex]
            return f'{y}-{m}-{d}'
        return f'{timeIndex}'

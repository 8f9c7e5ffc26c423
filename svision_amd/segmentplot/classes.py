"""Segment geometry (mirror of the reference's src/segmentplot/classes.py:42-107)."""


class Segment:
    """A +-45 degree diagonal: the read extent is derived from the reference span
    (classes.py:50-54), so ``xEnd = xStart +- (length - 1)``, ``yEnd = yStart + length - 1``."""
    __slots__ = ("_xStart", "_yStart", "_length", "_forward", "_segId", "_xEnd", "_yEnd")

    def __init__(self, x_start, y_start, length, forward, seg_id=0):
        self._xStart, self._yStart, self._length = x_start, y_start, length
        self._forward, self._segId = forward, seg_id
        self._xEnd = x_start + (length - 1) if forward else x_start - (length - 1)
        self._yEnd = y_start + (length - 1)

    def setxEnd(self, v): self._xEnd = v
    def setyEnd(self, v): self._yEnd = v
    def setLength(self, v): self._length = v

    def xStart(self): return self._xStart
    def yStart(self): return self._yStart
    def xEnd(self): return self._xEnd
    def yEnd(self): return self._yEnd
    def forward(self): return self._forward
    def length(self): return abs(self._xEnd - self._xStart)

    def toString(self):
        return "%s\t%s\t%s\t%s\t%s" % (self._xStart, self._xEnd, self._yStart, self._yEnd, self._forward)

    def fields(self):
        """The five TSV integers of this segment (forward as 0/1)."""
        return (self._xStart, self._xEnd, self._yStart, self._yEnd, 1 if self._forward else 0)


def cord_to_segments(cords):
    """[[q0,q1],[r0,r1],rev] triples -> Segments (run_hash_lineplot.py:35-49)."""
    return [Segment(c[0][0], c[1][0], int(c[1][1]) - int(c[1][0]) + 1, c[2] == 0, 0) for c in cords]

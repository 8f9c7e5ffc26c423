# Static types for segmentplot/classes.py (Cython "augmenting .pxd"; the .py source runs unchanged when interpreted).
cdef class Segment:
    cdef public long _xStart, _yStart, _length, _segId, _xEnd, _yEnd
    cdef public object _forward
    cpdef setxEnd(self, long v)
    cpdef setyEnd(self, long v)
    cpdef setLength(self, long v)
    cpdef long xStart(self)
    cpdef long yStart(self)
    cpdef long xEnd(self)
    cpdef long yEnd(self)
    cpdef forward(self)
    cpdef long length(self)
    cpdef str toString(self)
    cpdef tuple fields(self)

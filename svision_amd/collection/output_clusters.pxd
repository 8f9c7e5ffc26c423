# Static types for output_clusters.py (Cython "augmenting .pxd"; the .py source runs unchanged when interpreted).
cimport cython
from ..segmentplot.classes cimport Segment

@cython.locals(d_ref=long, d_read=long, ratio=double)
cpdef bint linearOrNot(Segment a, Segment b)

@cython.locals(s=Segment, total=double, span=long, ys=list)
cpdef cal_non_linear(list segs)

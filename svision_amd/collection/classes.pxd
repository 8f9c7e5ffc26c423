# Static types for classes.py (Cython "augmenting .pxd": the .py source stays plain Python and runs unchanged when interpreted).
cdef class Seg:
    cdef public long q_start, q_end, ref_start, ref_end, ref_id, qual, aln
    cdef public bint is_reverse, is_supplementary, derived
    cdef public object type, read_seq
    cpdef Seg copy(self)
    cpdef bint same_value(self, Seg o)

cpdef tuple by_read_pos(Seg seg)

cimport cython

@cython.locals(first=Seg, s=Seg, q0=long, r0=long, last=long, i=long, main=list, other=list)
cpdef tuple _segs_cords(list segs)

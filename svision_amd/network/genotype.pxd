# Static types for genotype.py (Cython "augmenting .pxd"; the .py source runs unchanged when interpreted).
cimport cython

@cython.locals(n_alt=Py_ssize_t, usable=long, k=long, r=Py_ssize_t, a=Py_ssize_t, nm=int, is_alt=bint, hit=bint, rs=long, re=long)
cpdef long _ref_names(const int[:] pos, const int[:] span, const long[:] ref_end, const unsigned short[:] flag, const unsigned char[:] mapq,
                      const int[:] name_id, Py_ssize_t first, Py_ssize_t stop, long s0, const long[:] alt_ids, long min_mapq, int mode,
                      long start, long end, double ov, int[:] out)

# Static types for cluster_signatures.py (Cython "augmenting .pxd"; the .py source runs unchanged when interpreted).
cimport cython

@cython.locals(n=Py_ssize_t, i=Py_ssize_t, j=Py_ssize_t, k=Py_ssize_t, si=double, ei=double, sj=double, ej=double, spi=double,
               spj=double, ci=double, cj=double, pos=double, d=double, num=double, den=double, spd=double)
cpdef _condensed_loops(double[::1] s, double[::1] e, double normalizer, double[::1] out)

# Static types for collect_signatures.py (Cython "augmenting .pxd"; the .py source runs unchanged when interpreted).
cimport cython
from .classes cimport Seg
from .analyze_reads cimport analyze_between_aligns, analyze_gap, analyze_inside_align

cpdef _emit(tuple ctx, Seg cur, Seg nxt, helpers, next_is_last)

@cython.locals(rid=long, lo=long, hi=long, primary=long, a=long, n=long, p=long, i=long, j=long, seg=Seg, s=Seg, first=Seg, second=Seg,
               supp=list, segs=list, majors=list, minors=list, main_idx=list, row_list=list, cols=dict, ctx=tuple, signatures=list)
cpdef list analyze_alignments(rows, sample, options, part_num=*)

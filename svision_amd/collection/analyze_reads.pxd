# Static types for analyze_reads.py (Cython "augmenting .pxd": the .py source stays plain Python and runs unchanged when
# interpreted).  Coordinates are C longs (a genome coordinate or a read offset: < 2^63), segments the extension type of
# classes.pxd; everything else stays a Python object.
cimport cython
from .classes cimport Seg

@cython.locals(rel_s=long, rel_e=long, n=long, k=long)
cpdef tuple shift_left(ref_seq, long ref_start, long target_start, long target_end)

@cython.locals(span=long)
cpdef cal_overlap_ratio(Seg base, Seg target, long left_most, long right_most)

@cython.locals(gap=long, left_most=long, right_most=long, grow=long, length=long, seg=Seg, other=Seg)
cpdef trim_segs(list segs, Seg first, Seg last)

@cython.locals(h=Seg)
cpdef bint _among(Seg seg, others)

@cython.locals(left=long, right=long, seg=Seg)
cpdef _signature(chrom, qname, sig_type, list first_bkp, list segs, list helpers, Seg trim_first, Seg trim_last, mechanism=*, long extend_end=*)

cpdef list _gap_bkp(long anchor_end, long next_start, long length_if_open, long length_if_closed=*)

@cython.locals(lo=long, hi=long, d_read=long, d_ref=long, diff=long, dup_len=long, blen=long, new_len=long, fixed=long, shift=long,
               s=long, e=long, seg=Seg, dup=Seg, rest=Seg, added=Seg, helpers=list, covered=list, segs=list)
cpdef analyze_gap(Seg cur, Seg nxt, chrom_of, fetch_ref, options, qname, help_segs=*)

@cython.locals(new=Seg)
cpdef _piece(list out, Seg seg, long q0, long q1, long r0, long r1)

@cython.locals(vrp=long, first_ref=long, m=long, ref_pos=long, length=long, kind=long, rows=list, out=list)
cpdef tuple analyze_inside_align(Seg seg, gaps, options=*, sample=*)

@cython.locals(p_rev=bint, a_rev=bint, qlen=long, lead=long, trail=long, r0=long, span=long, tid=long, mapq=long, flag=long,
               q_start=long, q_end=long, seg=Seg, base=Seg, target=Seg, left_most=long, right_most=long, last=long, i=long,
               majors=list, minors=list, same_strand=list, ordered=list)
cpdef tuple analyze_between_aligns(primary, supplementary, table, options, sample=*, cols=*)

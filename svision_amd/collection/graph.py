"""``--graph``: breakpoint graphs (rGFA-like text) of the junctions behind every signature, the most frequent
graph per called complex SV, and the graph-annotated VCF with its two isomorphism summaries.

Host-side mirror of the reference's ``src/collection/graph.py``: ``generate_graph`` :303-491,
``write_graph_to_file`` :141-179, ``parse_gfa_file`` :107-139, ``graph_is_same_as`` :182-275, ``classify_graphs`` :73-104,
``parse_graph_features`` :494-526 and ``collect_csv_same_format`` :518-676, called from
``collect_signatures.py:234-306`` (one graph per junction, built BEFORE analyze_gap touches the segments),
``output_clusters.py:57-67`` (one ``.gfa`` per read in ``graphs/{contig}-{cstart}-{cend}/``) and ``SVision:341-359``
(step 3).  Pure bookkeeping on a few nodes per read: it stays on the host.

Reference behaviour kept on purpose:
  * the ``DP`` tag of a duplicated node names the anchor by its id BEFORE the final renumbering (:397-437 vs :466-488);
  * a link's short form in ``GFA_L`` drops the second orientation (``'{0}{1}{2}'.format(a, sa, b, sb)``, :173);
  * a re-read graph carries the ``SO`` value in both coordinate fields, so the per-record file is the per-read file
    with the same ``S`` / ``L`` lines (:121-126);
  * a complex record whose region directory does not exist is dropped from the graph VCF (:576-577), the loop counter
    ``cnt`` doubles as the id of the matched graph (:604-621).
"""
import logging
import os


class Node:
    __slots__ = ("chr", "ref_start", "ref_end", "read_start", "read_end", "seq", "is_reverse", "id", "host",
                 "is_dup", "dup_from", "dup_from_cord")

    def __init__(self, chrom, ref_start, ref_end, read_start, read_end, seq, is_reverse, node_id, host):
        self.chr, self.ref_start, self.ref_end = chrom, ref_start, ref_end
        self.read_start, self.read_end = read_start, read_end
        self.seq, self.is_reverse, self.id, self.host = seq, is_reverse, node_id, host
        self.is_dup, self.dup_from, self.dup_from_cord = False, -1, -1

    def mark_dup(self, anchor_id, cord):
        self.is_dup, self.dup_from, self.dup_from_cord = True, anchor_id, cord


class Edge:
    __slots__ = ("node1", "rev1", "node2", "rev2")

    def __init__(self, node1, rev1, node2, rev2):
        self.node1, self.rev1, self.node2, self.rev2 = node1, rev1, node2, rev2


class Graph:
    __slots__ = ("nodes", "edges", "qname", "appear_time")

    def __init__(self, nodes, edges, qname=""):
        self.nodes, self.edges, self.qname, self.appear_time = nodes, edges, qname, 1


def _sign(rev):
    return "-" if rev else "+"


def _covered(base, target, left_most, right_most):
    """Share of ``base``'s reference interval under ``target`` (1.0 outside [left_most, right_most]); :278-300."""
    if target is None:
        return 0
    if base.ref_start < left_most or base.ref_end > right_most:
        return 1.0
    if base.ref_start >= target.ref_start and base.ref_end <= target.ref_end:
        return 1.0
    span = base.ref_end - base.ref_start
    if base.ref_end >= target.ref_end > base.ref_start and target.ref_start < base.ref_start:
        return (target.ref_end - base.ref_start) / span
    if base.ref_end < target.ref_start < base.ref_start and target.ref_end > base.ref_end:
        return (base.ref_end - target.ref_start) / span
    return 0


def build_graph(cur, nxt, helpers, min_sv, whole_read_seq, chrom_of, fetch_ref_str, qname, next_is_last=True):
    """Graph of the junction ``cur`` -> ``nxt`` of read ``qname`` (Seg objects carrying ``read_seq``; ``helpers``: the
    segments between them on the read).  Nothing passed in is modified."""
    def seg_node(seg, node_id, host):
        return Node(chrom_of(seg.ref_id), seg.ref_start, seg.ref_end, seg.q_start, seg.q_end, seg.read_seq, seg.is_reverse, node_id, host)

    skeleton, inserts = [], []
    cur_chr, nxt_chr = chrom_of(cur.ref_id), chrom_of(nxt.ref_id)
    cur_node = seg_node(cur, "S0", cur_chr)
    skeleton.append(cur_node)
    helper_nodes = [seg_node(h, "None", qname) for h in helpers]
    d_ref = nxt.ref_start - cur.ref_end
    if d_ref <= -min_sv:                                      # reference overlap: the overlapping head of nxt is a duplicated helper
        dup_len = -d_ref
        helper_nodes.append(Node(nxt_chr, nxt.ref_start, nxt.ref_start + dup_len, nxt.q_start, nxt.q_start + dup_len,
                                 nxt.read_seq[0:dup_len], cur.is_reverse, "None", qname))
        r0, q0, tail = nxt.ref_start + dup_len + 1, nxt.q_start + dup_len + 1, nxt.read_seq[dup_len:]
        if r0 < nxt.ref_end:
            nxt_node = Node(nxt_chr, r0, nxt.ref_end, q0, nxt.q_end, tail, cur.is_reverse, "S1", nxt_chr)
        elif next_is_last:
            nxt_node = None
        else:
            nxt_node = Node(nxt_chr, r0, r0 + 500, q0, q0 + 500, tail, cur.is_reverse, "S1", nxt_chr)
    else:
        nxt_node = seg_node(nxt, "S1", nxt_chr)
    if nxt_node is not None:
        skeleton.append(nxt_node)

    left_most, right_most = cur.ref_start, nxt.ref_end
    for node in helper_nodes:
        over_cur = _covered(node, cur_node, left_most, right_most)
        over_nxt = _covered(node, nxt_node, left_most, right_most)
        anchor = cur_node if over_cur > 0.8 else (nxt_node if over_nxt > 0.8 else None)
        if node.is_reverse and anchor is None:                # an inverted piece of its own: part of the skeleton
            node.id, node.host = "S%d" % len(skeleton), node.chr
            skeleton.append(node)
            continue
        if anchor is not None:
            node.mark_dup(anchor.id, node.ref_start)
        node.id = "I%d" % len(inserts)
        inserts.append(node)

    by_read = sorted(skeleton + inserts, key=lambda n: n.read_start)
    edges = []
    for prev, node in zip(by_read, by_read[1:]):
        if node.read_start - prev.read_end > min_sv:          # unaligned read bases between two nodes: an inserted node
            gap = Node(node.chr, node.ref_start, node.ref_start, prev.read_end + 1, node.read_start - 1,
                       whole_read_seq[prev.read_end + 1:node.read_start - 1], False, "I%d" % len(inserts), qname)
            inserts.append(gap)
            edges.append(Edge(prev.id, prev.is_reverse, gap.id, False))
            edges.append(Edge(gap.id, False, node.id, node.is_reverse))
        else:
            edges.append(Edge(prev.id, prev.is_reverse, node.id, node.is_reverse))

    by_ref = sorted(skeleton, key=lambda n: n.ref_start)
    for prev, node in zip(by_ref, by_ref[1:]):                # skipped reference between two skeleton nodes: a node of its own
        if node.ref_start - prev.ref_end > min_sv:
            s, e = prev.ref_end + 1, node.ref_start - 1
            skeleton.append(Node(node.chr, s, e, -1, -1, fetch_ref_str(node.chr, s, e), False, "S%d" % len(skeleton), node.host))

    renamed = {}
    skeleton = sorted(skeleton, key=lambda n: n.ref_start)
    for i, node in enumerate(skeleton):
        renamed[node.id] = node.id = "S%d" % i
    inserts = sorted(inserts, key=lambda n: n.read_start)
    for i, node in enumerate(inserts):
        renamed[node.id] = node.id = "I%d" % i
    for e in edges:
        e.node1, e.node2 = renamed[e.node1], renamed[e.node2]
    return Graph(skeleton + inserts, edges, qname)


def gfa_text(graph):
    """-> (file text, breakpoint positions, node ids, short link strings)."""
    out, cords, node_ids, links = [], set(), [], []
    for n in graph.nodes:
        seq = n.seq if n.seq != "" else "N"
        if "I" in n.id:
            line = "S\t%s\t%s\tSN:Z:%s\tSO:i:%s\tSR:i:0\tLN:i:%d" % (n.id, seq, n.host, n.read_start, len(seq))
            if n.is_dup:
                line += "\tDP:S:%s:%s" % (n.dup_from, n.dup_from_cord)
                cords.add(n.dup_from_cord)
        else:
            line = "S\t%s\t%s\tSN:Z:%s\tSO:i:%s\tSR:i:0\tLN:i:%d" % (n.id, seq, n.host, n.ref_start, len(seq))
            cords.add(n.ref_start)
        out.append(line + "\n")
        node_ids.append(n.id)
    for e in graph.edges:
        out.append("L\t%s\t%s\t%s\t%s\t0M\tSR:i:0\n" % (e.node1, _sign(e.rev1), e.node2, _sign(e.rev2)))
        links.append("%s%s%s" % (e.node1, _sign(e.rev1), e.node2))
    return "".join(out), list(cords), node_ids, links


def write_gfa(graph, path):
    text, cords, node_ids, links = gfa_text(graph)
    with open(path, "w") as f:
        f.write(text)
    return cords, node_ids, links


def read_gfa(path):
    nodes, edges = [], []
    with open(path) as f:
        for line in f:
            c = line.strip().split("\t")
            if c[0] == "S":
                start = c[4].split(":")[-1]
                node = Node(-1, start, -1, start, -1, c[2], False, c[1], c[3].split(":")[-1])
                if len(c) == 8:
                    dp = c[7].split(":")
                    node.mark_dup(dp[2], int(dp[3]))
                nodes.append(node)
            elif c[0] == "L":
                edges.append(Edge(c[1], c[2] == "-", c[3], c[4] == "-"))
    return Graph(nodes, edges)


def _kind_counts(graph):
    counts = {}
    for n in graph.nodes:
        counts[n.id[0]] = counts.get(n.id[0], 0) + 1
        if n.is_dup:
            counts["D"] = counts.get("D", 0) + 1
    return counts


def _path(graph):
    return "".join("%s%s%s%s" % (e.node1, _sign(e.rev1), e.node2, _sign(e.rev2)) for e in graph.edges)


def same_graph(a, b, strict=False, symmetry=False):
    """Same node / edge numbers and node-kind counts (every kind of ``a`` present in ``b`` with the same count), and --
    ``strict`` -- the same walk; ``symmetry``: ``a``'s walk equals ``b``'s read backwards with mirrored node numbers."""
    if len(a.nodes) != len(b.nodes) or len(a.edges) != len(b.edges):
        return False
    ca, cb = _kind_counts(a), _kind_counts(b)
    for kind, n in ca.items():
        if cb.get(kind) != n:
            return False
    if symmetry:
        mirror = {n.id: "%s%d" % (n.id[0], cb[n.id[0]] - int(n.id[1:]) - 1) for n in b.nodes}
        back = "".join("%s%s%s%s" % (mirror[e.node2], _sign(e.rev2), mirror[e.node1], _sign(e.rev1)) for e in reversed(b.edges))
        if _path(a) != back:
            return False
    if strict and _path(a) != _path(b):
        return False
    return True


def most_common_graphs(graphs):
    """Distinct graphs (strict comparison) by number of appearances, most frequent first (stable)."""
    distinct = [graphs[0]]
    for g in graphs[1:]:
        hits = [d for d in distinct if same_graph(g, d, strict=True)]
        if not hits:
            distinct.append(g)
        for d in hits:
            d.appear_time += 1
    return sorted(distinct, key=lambda g: g.appear_time, reverse=True)


def graph_features(graph):
    counts = _kind_counts(graph)
    return ",".join("%s:%d" % kv for kv in counts.items()), len(graph.edges), _path(graph)


def write_cluster_graphs(cluster, options):
    """One ``.gfa`` per signature of a reported cluster (output_clusters.py:57-67)."""
    path = os.path.join(options.out_path, "graphs", "%s-%d-%d" % (cluster.contig, int(cluster.cstart), int(cluster.cend)))
    os.makedirs(path, exist_ok=True)                          # two windows (two helper processes) can report the same boundary cluster
    for sig in cluster.get_signatures():
        write_gfa(sig.graph, os.path.join(path, "%s.gfa" % sig.graph.qname.replace("/", "_")))


def _vcf_records(vcf_path):
    """-> (header text as htslib prints it back, [tab-split body lines]).  The reference reads the merged VCF through
    pysam.VariantFile and writes ``str(header)`` / ``str(record)``: htslib re-serialises what it parsed, which for this
    writer's own VCF is the text itself plus the ``PASS`` filter line it always defines right after ``##fileformat``."""
    header, body = [], []
    with open(vcf_path) as f:
        for line in f:
            if line.startswith("#"):
                header.append(line)
            elif line.strip():
                body.append(line.rstrip("\n").split("\t"))
    pass_line = '##FILTER=<ID=PASS,Description="All filters passed">\n'
    if pass_line not in header:
        header.insert(1 if header and header[0].startswith("##fileformat") else 0, pass_line)
    return "".join(header), body


def annotate_vcf_with_graphs(gfa_dir, vcf_path, options):
    """Step 3 of the driver (collect_csv_same_format, :518-676): ``{sample}.svision.s{N}.graph.vcf`` = the merged VCF
    with GraphID / GFA_ID / GFA_S / GFA_L appended to INFO, one ``{chr}-{start}-{end}-{id}-{type}.gfa`` per complex
    record (its reads' most frequent graph), ``{sample}.graph_exactly_match.txt`` and ``{sample}.graph_symmetry_match.txt``."""
    out_path, sample = options.out_path, options.sample
    name = "%s.svision.s%s.graph.vcf" % (sample, options.min_support)
    logging.info("Adding GraphID, GFA INFO fields to VCF, output %s", name)
    header, records = _vcf_records(vcf_path)
    exact = {}                                                # representative record graph -> the records sharing it
    with open(os.path.join(out_path, name), "w") as out:
        out.write(header)
        for c in records:
            main, tail = c[:-2], "\t".join(c[-2:])
            if "CSV" not in "\t".join(c):
                main[-1] += ";GraphID=-1;GFA_ID=.;GFA_S=.;GFA_L=."
                out.write("\t".join(main) + "\t" + tail + "\n")
                continue
            info = dict(kv.split("=", 1) for kv in c[7].split(";") if "=" in kv)
            chrom, start, end, rec_id, sv_type = c[0], int(c[1]), int(info["END"]), c[2], info["SVTYPE"]
            region_dir = os.path.join(gfa_dir, "%s-%s-%s" % (chrom, start, end))
            if not os.path.exists(region_dir):
                continue
            read_graphs = [read_gfa(os.path.join(region_dir, "%s.gfa" % r.replace("/", "_"))) for r in info["READS"].split(",")]
            target = "%s-%s-%s-%s-%s" % (chrom, start, end, rec_id, sv_type)
            _cords, node_ids, links = write_gfa(most_common_graphs(read_graphs)[0], os.path.join(gfa_dir, target + ".gfa"))
            mine = read_gfa(os.path.join(gfa_dir, target + ".gfa"))
            graph_id = -1
            for i, base in enumerate(exact):
                if same_graph(mine, read_gfa(os.path.join(gfa_dir, base + ".gfa")), strict=True):
                    exact[base].append(target)
                    graph_id = i
                    break
            if graph_id < 0:
                exact[target] = [target]
                graph_id = len(exact) - 1
            main[-1] += ";GraphID=%d;GFA_ID=%s;GFA_S=%s;GFA_L=%s" % (graph_id, target, ",".join(node_ids), ",".join(links))
            out.write("\t".join(main) + "\t" + tail + "\n")

    logging.info("Find symmetric graphs")
    bases = list(exact)
    loaded = {b: read_gfa(os.path.join(gfa_dir, b + ".gfa")) for b in bases}
    symmetric, pair = {}, {}
    for i, base in enumerate(bases):
        for j in range(i + 1, len(bases)):
            other = bases[j]
            if same_graph(loaded[other], loaded[base], strict=True):
                continue
            if same_graph(loaded[other], loaded[base], strict=False, symmetry=True):
                symmetric[base] = exact[base] + exact[other]
                pair[base] = ["%d,%d" % (i, j), "%d,%d" % (len(exact[base]), len(exact[other]))]
    with open(os.path.join(out_path, "%s.graph_exactly_match.txt" % sample), "w") as f:
        for i, base in enumerate(bases):
            nodes, n_edges, path = graph_features(loaded[base])
            f.write("> GraphId=%d\tNumber=%d\tNodes=%s\tEdges=%s\tPath=%s\n" % (i, len(exact[base]), nodes, n_edges, path))
            f.write("\t".join(exact[base]) + "\n")
    with open(os.path.join(out_path, "%s.graph_symmetry_match.txt" % sample), "w") as f:
        for base, members in symmetric.items():
            nodes, n_edges, path = graph_features(loaded[base])
            f.write("> GraphId=%s\tNumber=%s\tNodes=%s\tEdges=%s\tPath=%s" % (pair[base][0], pair[base][1], nodes, n_edges, path))
            f.write(",%s\n" % graph_features(read_gfa(os.path.join(gfa_dir, members[-1] + ".gfa")))[2])
            f.write("\t".join(members) + "\n")
    return exact, symmetric

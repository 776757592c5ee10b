#!/usr/bin/env python3
"""Writes the base arrays of tests/golden/paper_example.json -- graph, nodes, C, IN, OUT, B_S, B_V, V_S, find, locate -- from
the two figures of the paper's worked GCSA example, read MECHANICALLY by the coordinates of their objects:

  paper/gcsa2_graph_dbg.ipe     Figure 2, left half: the input graph.  Twelve square boxes (closed four-point paths), a label
                                "id:base" at the centre of each, thirteen two-point arrows; an arrow starts on the boundary of
                                its source box and ends on the boundary of its target box.
  paper/gcsa2_pruned_index.ipe  Figure 3: the pruned path graph (left: a key label with the node's values eight units below it)
                                and the index as a table (right): columns key | OUT | BWT | IN | key | B_S | B_V | V_S under
                                their headers, one row every eight units; OUT / BWT / IN have a row per EDGE, the key columns
                                and B_S a row per path node, B_V / V_S a row per sample.

Nothing here is computed by the oracle, the builder or tests/naive.py.  What is derived is derived by the definitions:

  nodes[i].bwt        the BWT rows of node i: IN marks the last incoming edge of every node (paper.tex:534-540)
  nodes[i].outdegree  OUT in unary: zeros, then the one of the node's last outgoing edge
  nodes[i].values     the numbers printed under the node's key in the drawing (checked against B_S / B_V / V_S)
  C[c]                number of BWT rows with a character smaller than c in the order $ACGTN# (support.cpp:69-92)
  find(X)             |X| <= order: the keys that are a prefix of X or that X is a prefix of, and from one of whose values a
                      path of the INPUT graph spells X (Figure 2, walked here edge by edge); longer X: the same with the
                      walk -- for the caption's patterns, which are paths of the graph (no false positives to argue about);
                      a pattern without occurrence: the edge-space pair that backward search over the printed BWT rows ends
                      with (paper.tex:541-557: sp' = C[c] + |{rows before node sp with character c}|, ep' likewise behind
                      node ep, minus one; gcsa.h:160 returns it as it is)
  locate(range)       the union of the values of the range's nodes, sorted; count = its size

Values: the figure numbers the input-graph nodes 0..11 and pads the source with the abstract positions 0:1 and 0:2; the JSON
stores figure value + 2, 0:1 -> 1, 0:2 -> 0, so that a predecessor's value is the value minus one as integers (src/gcsa.cpp:893).

The "suffix_tree" section is added by make_paper_lcp.py from the keys written here (this script calls it).  Runs only where
/root/reference exists; the JSON travels.

    python tests/golden/make_paper_example.py [--check]      (--check: compare with the committed file, write nothing)
"""
import importlib.util
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/paper"
OUT = os.path.join(HERE, "paper_example.json")
ORDER = "$ACGTN#"
QUERIES = ["T", "AT", "CAT", "CATG", "GCA", "TCATA", "GCATCATA$", "GCTTGTA", "TT", "ATT"]       # the patterns asked (inputs)
LOCATE_RANGES = [(9, 12), (2, 4), (5, 5), (6, 6)]

TEXT = re.compile(r'<text[^>]*matrix="1 0 0 1 (-?[\d.]+) (-?[\d.]+)"[^>]*pos="(-?[\d.]+) (-?[\d.]+)"[^>]*>(.*?)</text>')
PATH = re.compile(r'<path([^>]*)>\s*(.*?)</path>', re.S)


def clean(label):
    """'{\\scriptsize $\\dnaseq{A}\\$$}' -> 'A$', '{\\scriptsize 3:$\\baseA$}' -> '3:A', '$\\bvOUT$' -> 'bvOUT', '$B_{S}$' -> 'B_S'."""
    label = re.sub(r"^\{\\scriptsize (.*)\}$", r"\1", label.strip())
    label = label.replace("\\$", "\0").replace("$", "").replace("\0", "$")       # math delimiters out, escaped dollars stay
    label = re.sub(r"\\dnaseq\{([ACGTN]*)\}", r"\1", label)
    label = re.sub(r"\\base([ACGTN])", r"\1", label)
    label = label.replace("\\#", "#").replace("{", "").replace("}", "")
    return label.lstrip("\\")


def labels_of(raw):
    return [(float(px) + float(tx), float(py) + float(ty), clean(body)) for tx, ty, px, py, body in TEXT.findall(raw)]


def paths_of(raw):
    """(attributes, absolute points, closed) of every <path> whose body is plain 'x y m / x y l / h' lines."""
    out = []
    for attrs, body in PATH.findall(raw):
        m = re.search(r'matrix="1 0 0 1 (-?[\d.]+) (-?[\d.]+)"', attrs)
        tx, ty = (float(m.group(1)), float(m.group(2))) if m else (0.0, 0.0)
        points, closed, plain = [], False, True
        for line in body.strip().splitlines():
            parts = line.split()
            if parts == ["h"]:
                closed = True
            elif len(parts) == 3 and parts[2] in ("m", "l"):
                points.append((float(parts[0]) + tx, float(parts[1]) + ty))
            else:
                plain = False
        if plain and points:
            out.append((attrs, points, closed))
    return out


def read_input_graph():
    """Figure 2, left half (x < 340): boxes, their "id:base" labels, arrows -> labels string and sorted edge list."""
    with open(os.path.join(REF, "gcsa2_graph_dbg.ipe")) as f:
        raw = f.read()
    raw = raw[raw.index("<page"):]                              # (the style sheet in front holds symbol definitions: paths too)
    boxes = []
    for attrs, pts, closed in paths_of(raw):
        if closed and len(pts) == 4 and "arrow" not in attrs and max(x for x, y in pts) < 340:
            xs, ys = sorted({x for x, y in pts}), sorted({y for x, y in pts})
            if len(xs) == 2 and len(ys) == 2:
                boxes.append((xs[0], xs[1], ys[0], ys[1]))
    names = {}
    for x, y, label in labels_of(raw):
        m = re.fullmatch(r"(\d+):([ACGTN$#])", label)
        if m and x < 340:
            inside = [b for b in boxes if b[0] < x < b[1] and b[2] < y < b[3]]
            assert len(inside) == 1, (label, inside)
            assert inside[0] not in names
            names[inside[0]] = (int(m.group(1)), m.group(2))
    assert len(names) == len(boxes) == 12, (len(names), len(boxes))

    def box_at(px, py):
        hits = [b for b in boxes if b[0] <= px <= b[1] and b[2] <= py <= b[3] and (px in (b[0], b[1]) or py in (b[2], b[3]))]
        assert len(hits) == 1, ((px, py), hits)
        return names[hits[0]][0]

    edges = []
    for attrs, pts, closed in paths_of(raw):
        if "arrow=" in attrs and not closed and len(pts) == 2 and max(x for x, y in pts) < 340:
            edges.append([box_at(*pts[0]), box_at(*pts[1])])
    by_id = dict(names.values())
    assert sorted(by_id) == list(range(12))
    return "".join(by_id[i] for i in range(12)), sorted(edges)


def read_pruned_index():
    """Figure 3: the table's columns and the values printed under the keys of the drawn path graph."""
    with open(os.path.join(REF, "gcsa2_pruned_index.ipe")) as f:
        raw = f.read()
    labels = labels_of(raw[raw.index("<page"):])
    header_names = {"gkey": "key", "bvOUT": "OUT", "BWT": "BWT", "bvIN": "IN", "B_S": "B_S", "B_V": "B_V", "V_S": "V_S"}
    headers = [(x, y, header_names[l]) for x, y, l in labels if l in header_names]
    assert len({y for x, y, l in headers}) == 1 and len(headers) == 8, headers
    header_y = headers[0][1]
    table_left = min(x for x, y, l in headers) - 16

    def column(x0):
        cells = sorted(((y, l) for x, y, l in labels if x == x0 and y < header_y), reverse=True)
        ys = [y for y, l in cells]
        assert all(abs((ys[i] - ys[i + 1]) - 8) < 1e-6 for i in range(len(ys) - 1)), (x0, ys)      # one row every eight units
        return ys, [l for y, l in cells]

    cols = {}
    for x, y, name in sorted(headers):
        ys, cells = column(x)
        cols.setdefault(name, []).append((ys, cells))
    (key_ys, keys), (key_ys2, keys2) = cols["key"]
    assert keys == keys2 and key_ys == key_ys2 and len(keys) == 16
    assert keys == sorted(keys, key=lambda k: [ORDER.index(c) for c in k]), "the figure lists the keys in lexicographic order"
    (out_ys, out_bits), = cols["OUT"]
    (bwt_ys, bwt), = cols["BWT"]
    (in_ys, in_bits), = cols["IN"]
    assert out_ys == bwt_ys == in_ys and len(bwt) == 20                    # a row per edge
    (bs_ys, bs), = cols["B_S"]
    assert bs_ys == key_ys                                                 # a row per path node
    (bv_ys, bv), = cols["B_V"]
    (vs_ys, vs), = cols["V_S"]
    assert bv_ys == vs_ys and len(vs) == 11                                # a row per sample
    assert all(b in "01" for b in out_bits + in_bits + bs + bv) and all(len(c) == 1 and c in ORDER for c in bwt)
    # the drawing: a key label with the node's values eight units below it, left of the table
    drawn = {}
    for x, y, label in labels:
        if x < table_left and label in keys:
            below = [l for xx, yy, l in labels if xx == x and yy == y - 8]
            assert len(below) == 1 and label not in drawn, (label, below)
            drawn[label] = below[0].split(",")
    assert sorted(drawn) == sorted(keys)
    return keys, "".join(out_bits), bwt, "".join(in_bits), "".join(bs), "".join(bv), vs, drawn


def shift(value):
    """figure value -> stored value: 0:2 -> 0, 0:1 -> 1, v -> v + 2 (module docstring)."""
    return {"0:2": 0, "0:1": 1}.get(value, None) if ":" in value else int(value) + 2


def build():
    labels, edges = read_input_graph()
    keys, out_bits, bwt, in_bits, bs, bv, vs, drawn = read_pruned_index()
    n = len(keys)
    # BWT rows per node (IN ends a node's rows), outdegrees (OUT in unary)
    rows, current = [], ""
    for c, bit in zip(bwt, in_bits):
        current += c
        if bit == "1":
            rows.append(current)
            current = ""
    assert current == "" and len(rows) == n
    outdeg, run = [], 0
    for bit in out_bits:
        run += 1
        if bit == "1":
            outdeg.append(run)
            run = 0
    assert run == 0 and len(outdeg) == n
    values = [sorted(shift(v) for v in drawn[k]) for k in keys]
    # the sampled nodes, their sample counts and the samples, as the three arrays print them, agree with the drawing
    stored, at = [shift(v) for v in vs], 0
    for i in range(n):
        if bs[i] == "1":
            end = at
            while bv[end] == "0":
                end += 1
            assert sorted(stored[at:end + 1]) == values[i], (keys[i], stored[at:end + 1], values[i])
            at = end + 1
    assert at == len(stored)
    C = [sum(1 for c in bwt if ORDER.index(c) < comp) for comp in range(len(ORDER))] + [len(bwt)]
    nodes = [{"key": keys[i], "values": values[i], "bwt": rows[i], "outdegree": outdeg[i]} for i in range(n)]

    # occurrences of a pattern in the input graph: start nodes of the paths that spell it (the sink's $ may repeat: the figure
    # pads keys with $ behind the sink as it pads with # in front of the source)
    succ = {i: [b for a, b in edges if a == i] for i in range(len(labels))}

    def spells(node, pattern):
        if labels[node] != pattern[0]:
            return False
        if len(pattern) == 1:
            return True
        if labels[node] == "$" and not succ[node]:
            return all(c == "$" for c in pattern)
        return any(spells(nxt, pattern[1:]) for nxt in succ[node])

    out_before = [0]                                           # out_before[i] = outgoing edges in front of node i (OUT in unary)
    for d in outdeg:
        out_before.append(out_before[-1] + d)

    def node_of_edge(e):
        return max(i for i in range(n) if out_before[i] <= e)

    def backward_search(pattern):
        """(sp, ep) over the printed arrays by the formulas of paper.tex:541-557: the rows with character c in front of node sp
        / up to node ep, shifted by C[c], are positions among the outgoing edges (OUT); the nodes that own them are the new
        range.  A step that leaves no row returns its pair of edge positions as it is (gcsa.h:160)."""
        sp, ep = 0, n - 1
        for c in reversed(pattern):
            comp = ORDER.index(c)
            lo = C[comp] + sum(r.count(c) for r in rows[:sp])
            hi = C[comp] + sum(r.count(c) for r in rows[:ep + 1]) - 1
            if hi < lo:
                return [lo, hi]
            sp, ep = node_of_edge(lo), node_of_edge(hi)
        return [sp, ep]

    find = []
    for x in QUERIES:
        starts = {shift(str(v)) for v in range(len(labels)) if spells(v, x)}
        hit = [i for i in range(n) if (keys[i].startswith(x) or x.startswith(keys[i])) and starts & set(values[i])]
        searched = backward_search(x)
        if hit:
            assert hit == list(range(hit[0], hit[-1] + 1)) and searched == [hit[0], hit[-1]], (x, hit, searched)
            find.append({"pattern": x, "range": [hit[0], hit[-1]]})
        else:
            assert searched[1] + 1 == searched[0], (x, searched)
            find.append({"pattern": x, "range": searched, "note": "empty, returned in edge space (include/gcsa/gcsa.h:160)"})
    locate = []
    for sp, ep in LOCATE_RANGES:
        vals = sorted({v for i in range(sp, ep + 1) for v in values[i]})
        locate.append({"range": [sp, ep], "values": vals, "count": len(vals)})
    return {
        "_source": "Worked example of the GCSA2 paper, Figures 2-3: reference paper/gcsa2_graph_dbg.ipe and paper/gcsa2_pruned_index.ipe (text objects read by coordinates; transcription in SURVEY.md section 4.3). Data only: the figure's arrays and the answers its captions state or that follow from the arrays by the Appendix-A formulas (paper.tex:534-557).",
        "_values_note": "The figure numbers input-graph nodes 0..11 and pads the source with abstract positions 0:1 and 0:2. Here every value is the figure's value + 2 and 0:1 -> 1, 0:2 -> 0, so that value = predecessor value + 1 holds as integers (src/gcsa.cpp:893 adds steps as integers).",
        "order": max(len(k) for k in keys), "comp_order": ORDER,
        "graph": {"labels": labels, "edges": edges},
        "nodes": nodes, "C": C, "IN": in_bits, "OUT": out_bits, "B_S": bs, "B_V": bv, "V_S": stored,
        "find": find, "locate": locate,
    }


def with_suffix_tree(gold):
    """The "suffix_tree" section, as make_paper_lcp.py derives it from the keys (its main() on an object instead of the file)."""
    spec = importlib.util.spec_from_file_location("make_paper_lcp", os.path.join(HERE, "make_paper_lcp.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    gold["suffix_tree"] = mod.suffix_tree_of(gold)
    return gold


def render():
    return json.dumps(with_suffix_tree(build()), indent=2) + "\n"


def main():
    text = render()
    if "--check" in sys.argv[1:]:
        with open(OUT) as f:
            same = (f.read() == text)
        print("paper_example.json:", "reproduced byte for byte" if same else "DIFFERS from what the figures give")
        sys.exit(0 if same else 1)
    with open(OUT, "w") as f:
        f.write(text)
    print("wrote", OUT)


if __name__ == "__main__":
    main()

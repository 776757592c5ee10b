#!/usr/bin/env python3
"""Writes tests/golden/text_example.json: the second worked example the reference holds, Figure 1 of the paper
(paper/paper.tex:147-151, drawn in paper/gcsa2_text_indexes.ipe) -- the sorted suffixes of the text GCATCATA$ with
their BWT, SA, LCP and LF columns.  The columns are read mechanically from the figure's text labels: every label is an
<text> element whose position (`pos` + the translation in `matrix`) gives its column (x) and its row (y; the rows
are the sorted suffixes, 12 units apart).  Nothing is computed here except two consistency checks on the transcription
(the suffixes are sorted; SA names them).

A GCSA of a text is its FM-index ("degenerates to an FM-index with one value per node"), so the
figure is a known-answer vector for LF, locate, the LCP array and everything derived from it -- in particular a
reference-held LCP array, which the paper's GCSA figure (paper_example.json) does not carry.

Runs only where /root/reference exists (the build container); the JSON travels.

    python tests/golden/make_text_example.py
"""
import importlib.util
import json
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
FIGURE = "/root/reference/paper/gcsa2_text_indexes.ipe"
OUT = os.path.join(HERE, "text_example.json")

TEXT = re.compile(r'<text[^>]*matrix="1 0 0 1 (-?[\d.]+) (-?[\d.]+)"[^>]*pos="(-?[\d.]+) (-?[\d.]+)"[^>]*>(.*?)</text>')


def clean(label):
    """'$\\dnaseq{ATA}\\$$' -> 'ATA$', '$\\baseA$' -> 'A', '$\\$$' -> '$', '7' -> '7', '$\\SA$' -> 'SA'."""
    if len(label) >= 2 and label[0] == "$" and label[-1] == "$":     # math mode delimiters
        label = label[1:-1]
    label = re.sub(r"\\dnaseq\{([ACGT]*)\}", r"\1", label)
    label = re.sub(r"\\base([ACGT])", r"\1", label)
    label = label.replace("\\$", "$").replace("\\textbf{Suffixes}", "Suffixes")
    return label.lstrip("\\")


def main():
    with open(FIGURE) as f:
        raw = f.read()
    labels = [(float(px) + float(tx), float(py) + float(ty), clean(body)) for tx, ty, px, py, body in TEXT.findall(raw)]
    headers = {name: x for x, y, name in labels if name in ("SA", "BWT", "LCP", "LF")}
    assert set(headers) == {"SA", "BWT", "LCP", "LF"}, headers
    header_y = {y for x, y, name in labels if name in headers}
    assert len(header_y) == 1
    # the suffix column drawn next to SA / LCP (left-aligned copy; the figure repeats it right-aligned beside LF)
    suffix_x = min(x for x, y, name in labels if name.endswith("$") and x > headers["BWT"])
    rows = sorted(((y, name) for x, y, name in labels if x == suffix_x and name.endswith("$")), reverse=True)
    ys = [y for y, name in rows]
    suffixes = [name for y, name in rows]
    assert len(suffixes) == 9 and all(abs((ys[i] - ys[i + 1]) - 12) < 1e-6 for i in range(8)), rows

    suffix_columns = {x for x, y, name in labels if name in suffixes and len(name) > 1}     # the two copies of the suffix list
    assert len(suffix_columns) == 2 and suffix_x in suffix_columns

    def column(name):
        """Values of the column under header `name`: labels on a suffix row, in no suffix column, exactly under the header."""
        cells = {}
        for x, y, label in labels:
            if y in header_y or x in suffix_columns or label == "Suffixes":
                continue
            if y not in ys:
                continue
            if x == headers[name]:
                assert ys.index(y) not in cells, (name, y)
                cells[ys.index(y)] = label
        assert sorted(cells) == list(range(9)), (name, cells)
        return [cells[i] for i in range(9)]

    bwt = column("BWT")
    sa = [int(v) for v in column("SA")]
    lcp = [int(v) for v in column("LCP")]
    lf = [int(v) for v in column("LF")]
    text = suffixes[sa.index(0)]
    # transcription checks: SA names the suffixes, and they are listed in lexicographic order ($ smallest)
    assert all(text[sa[i]:] == suffixes[i] for i in range(9))
    assert suffixes == sorted(suffixes, key=lambda s: ["$ACGT".index(c) for c in s])
    out = {
        "source": "paper/paper.tex:147-151 (Figure 1), labels of paper/gcsa2_text_indexes.ipe; transcribed by tests/golden/make_text_example.py",
        "text": text, "suffixes": suffixes, "BWT": bwt, "SA": sa, "LCP": lcp, "LF": lf,
    }
    # The GCSA of the text's path graph has one more path node: the source marker '#' the graph starts with sorts after
    # every base (comp order $ACGTN#, support.cpp:69-92), shares no prefix with its neighbour, precedes the text
    # (it is the predecessor of the suffix with SA = 0) and is preceded by the sink (the edge (t, s), paper.tex:237).
    # Everything below follows from the figure's columns by definitions; nothing is computed by the oracle or the builder.
    keys = suffixes + ["#" + text]
    n = len(keys)
    gcsa_lcp = lcp + [0]
    gcsa_lf = [(n - 1 if sa[i] == 0 else lf[i]) for i in range(9)] + [0]
    gcsa_bwt = [("#" if sa[i] == 0 else bwt[i]) for i in range(9)] + ["$"]
    body = text[:-1]
    present = sorted({body[i:j] for i in range(len(body)) for j in range(i + 1, len(body) + 1)} | {text[i:] for i in range(len(text))})
    find = []
    for x in present:
        rows = [i for i in range(n) if keys[i].startswith(x)]
        assert rows == list(range(rows[0], rows[-1] + 1))
        find.append({"pattern": x, "range": [rows[0], rows[-1]], "positions": sorted(sa[i] for i in rows if i < 9)})
    absent = [x for x in ("AA", "CC", "GG", "TT", "AC", "AG", "CG", "CT", "GA", "GT", "TG", "CATG", "GCATCATAA", "TAT", "N", "ANA")
              if x not in text]
    spec = importlib.util.spec_from_file_location("make_paper_lcp", os.path.join(HERE, "make_paper_lcp.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    ranges = [(i, i) for i in range(n)] + [tuple(q["range"]) for q in find] + [(0, n - 1), (0, 3), (1, 5), (4, 8), (7, 9)]
    out["gcsa"] = {
        "_source": "the figure's columns plus the source-marker row; find ranges, positions and the suffix tree by definitions",
        "path_nodes": n, "keys": keys, "BWT": gcsa_bwt, "LF": gcsa_lf, "find": find, "absent": absent,
        "suffix_tree": mod.derive_suffix_tree(keys, gcsa_lcp, ranges),
    }
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print({k: out[k] for k in ("text", "BWT", "SA", "LCP", "LF")}, len(find), "find cases")


if __name__ == "__main__":
    main()

"""result.xlsx without openpyxl: a minimal OOXML (SpreadsheetML) workbook writer on zipfile.

The reference writes `results/result.xlsx` with pandas + openpyxl (run_YACHT.py:231-254: one sheet per min_coverage,
named `min_coverage{c}`, plus `raw_result` with --keep_raw) and its own workflow test reads the file back with
`pd.read_excel` (tests/test_workflow.py:52-66).  openpyxl is an optional dependency; the file format is not: a workbook
is a zip of a handful of XML parts.  This writer emits exactly what that read-back needs -- a header row, one row per
record; numbers as numeric cells (repr: doubles round-trip), booleans as boolean cells (`in_sample_est` reads back as
True / False), everything else as inline strings, missing values as empty cells -- and nothing else (no styles beyond
the mandatory default, no shared-string table, no column widths).  `read_xlsx` is the matching minimal reader
(zipfile + xml.etree) used by the tests and by callers that have no openpyxl either.
"""
from __future__ import annotations

import math
import re
import zipfile
from typing import Dict, List, Sequence, Tuple
from xml.etree import ElementTree as ET
from xml.sax.saxutils import escape

import numpy as np
import pandas as pd

_NS = "http://schemas.openxmlformats.org/spreadsheetml/2006/main"
_NS_R = "http://schemas.openxmlformats.org/officeDocument/2006/relationships"
_NS_PKG = "http://schemas.openxmlformats.org/package/2006/relationships"
_BAD_SHEET_CHARS = re.compile(r"[\[\]:*?/\\]")
# characters XML 1.0 cannot carry (control characters but tab / LF / CR)
_BAD_XML = re.compile("[\x00-\x08\x0b\x0c\x0e-\x1f]")
# anything that keeps a string from going into a cell as it is: markup characters, control characters, other white space at an end
_NOT_PLAIN = re.compile("[&<>\x00-\x1f]|^\\s|\\s$")


def _col(i: int) -> str:
    """0-based column index -> A, B, ..., Z, AA, ..."""
    s = ""
    i += 1
    while i:
        i, r = divmod(i - 1, 26)
        s = chr(65 + r) + s
    return s


def sheet_title(name: str, taken: Sequence[str] = ()) -> str:
    """Excel's rules: at most 31 characters, none of []:*?/\\, unique per workbook (openpyxl enforces the same)."""
    t = _BAD_SHEET_CHARS.sub("_", str(name))[:31] or "Sheet"
    base, k = t, 1
    while t in taken:
        suffix = f"_{k}"
        t = base[: 31 - len(suffix)] + suffix
        k += 1
    return t


def _cell(ref: str, v) -> str:
    if v is None:
        return ""
    if isinstance(v, (bool, np.bool_)):
        return f'<c r="{ref}" t="b"><v>{int(bool(v))}</v></c>'
    if v is None or v is getattr(_pd_na(), "NA", None) or v is getattr(_pd_na(), "NaT", None):
        return ""  # a missing value of a nullable column: an empty cell, as NaN is
    if isinstance(v, (int, np.integer)):
        return f'<c r="{ref}"><v>{int(v)}</v></c>'
    if isinstance(v, (float, np.floating)):
        f = float(v)
        if math.isnan(f):
            return ""  # pandas writes NaN as an empty cell too
        if math.isinf(f):
            return f'<c r="{ref}" t="inlineStr"><is><t>{"inf" if f > 0 else "-inf"}</t></is></c>'
        return f'<c r="{ref}"><v>{f!r}</v></c>'
    s = _BAD_XML.sub("", str(v))
    keep = ' xml:space="preserve"' if s != s.strip() else ""
    return f'<c r="{ref}" t="inlineStr"><is><t{keep}>{escape(s)}</t></is></c>'


def _pd_na():
    """pandas, when it is importable (the tables written here are DataFrames, so it is): its NA / NaT singletons."""
    try:
        import pandas as pd
        return pd
    except Exception:  # noqa: BLE001
        return None


def _column_cells(letters: str, values: list, dtype) -> List[str]:
    """The cells of one column, row 2 on (one string per row; "" = no cell).  Columns of one numpy dtype -- what a result
    table is made of -- are formatted without a type test per cell (a 20 000-row sheet: 0.3 s of `_cell` calls before)."""
    n = len(values)
    # the fast paths are for plain numpy dtypes only: pandas' nullable Int64 / UInt64 / boolean report kind "i" / "u" / "b" too,
    # but their .tolist() holds pd.NA ("<v><NA></v>" is not XML) -- those go cell by cell, NA as an empty cell
    kind = dtype.kind if isinstance(dtype, np.dtype) else "O"
    if kind == "b":
        return [f'<c r="{letters}{r}" t="b"><v>{1 if v else 0}</v></c>' for r, v in zip(range(2, n + 2), values)]
    if kind in "iu":
        return [f'<c r="{letters}{r}"><v>{v}</v></c>' for r, v in zip(range(2, n + 2), values)]
    if kind == "f":
        out = []
        for r, v in zip(range(2, n + 2), values):
            if v != v:  # NaN: pandas writes an empty cell too
                out.append("")
            elif v in (math.inf, -math.inf):
                out.append(f'<c r="{letters}{r}" t="inlineStr"><is><t>{"inf" if v > 0 else "-inf"}</t></is></c>')
            else:
                out.append(f'<c r="{letters}{r}"><v>{v!r}</v></c>')
        return out
    out = []
    for r, v in zip(range(2, n + 2), values):
        if type(v) is str and not _NOT_PLAIN.search(v) and v[:1] != " " and v[-1:] != " ":
            out.append(f'<c r="{letters}{r}" t="inlineStr"><is><t>{v}</t></is></c>')  # (nothing to escape, strip or drop)
        else:
            out.append(_cell(f"{letters}{r}", v))
    return out


def _sheet_xml(df: pd.DataFrame) -> str:
    out: List[str] = ['<?xml version="1.0" encoding="UTF-8" standalone="yes"?>\n',
                      f'<worksheet xmlns="{_NS}"><sheetData>']
    out.append('<row r="1">' + "".join(_cell(f"{_col(c)}1", str(name)) for c, name in enumerate(df.columns)) + "</row>")
    # (python scalars, column by column: .tolist() turns numpy scalars into int / float / bool)
    cols = [_column_cells(_col(c), df.iloc[:, c].tolist(), df.dtypes.iloc[c]) for c in range(df.shape[1])]
    for r, cells in enumerate(zip(*cols) if cols else ()):
        out.append(f'<row r="{r + 2}">' + "".join(cells) + "</row>")
    if not cols:
        out.extend(f'<row r="{r + 2}"></row>' for r in range(len(df)))
    out.append("</sheetData></worksheet>")
    return "".join(out)


def write_xlsx(path: str, tables: Sequence[Tuple[str, pd.DataFrame]]) -> List[str]:
    """One sheet per (name, DataFrame), in order; returns the sheet titles as written."""
    titles: List[str] = []
    for name, _ in tables:
        titles.append(sheet_title(name, titles))
    if not titles:  # a workbook needs one sheet (openpyxl raises here; an empty sheet keeps the file readable)
        tables, titles = [("Sheet1", pd.DataFrame())], ["Sheet1"]
    n = len(titles)
    content_types = ('<?xml version="1.0" encoding="UTF-8" standalone="yes"?>\n'
                     '<Types xmlns="http://schemas.openxmlformats.org/package/2006/content-types">'
                     '<Default Extension="rels" ContentType="application/vnd.openxmlformats-package.relationships+xml"/>'
                     '<Default Extension="xml" ContentType="application/xml"/>'
                     '<Override PartName="/xl/workbook.xml" ContentType="application/vnd.openxmlformats-officedocument.spreadsheetml.sheet.main+xml"/>'
                     '<Override PartName="/xl/styles.xml" ContentType="application/vnd.openxmlformats-officedocument.spreadsheetml.styles+xml"/>'
                     + "".join(f'<Override PartName="/xl/worksheets/sheet{i + 1}.xml" ContentType="application/vnd.openxmlformats-officedocument.spreadsheetml.worksheet+xml"/>'
                               for i in range(n)) + "</Types>")
    root_rels = ('<?xml version="1.0" encoding="UTF-8" standalone="yes"?>\n'
                 f'<Relationships xmlns="{_NS_PKG}">'
                 f'<Relationship Id="rId1" Type="{_NS_R}/officeDocument" Target="xl/workbook.xml"/></Relationships>')
    workbook = ('<?xml version="1.0" encoding="UTF-8" standalone="yes"?>\n'
                f'<workbook xmlns="{_NS}" xmlns:r="{_NS_R}"><sheets>'
                + "".join(f'<sheet name="{escape(t, {chr(34): "&quot;"})}" sheetId="{i + 1}" r:id="rId{i + 1}"/>' for i, t in enumerate(titles))
                + "</sheets></workbook>")
    wb_rels = ('<?xml version="1.0" encoding="UTF-8" standalone="yes"?>\n'
               f'<Relationships xmlns="{_NS_PKG}">'
               + "".join(f'<Relationship Id="rId{i + 1}" Type="{_NS_R}/worksheet" Target="worksheets/sheet{i + 1}.xml"/>' for i in range(n))
               + f'<Relationship Id="rId{n + 1}" Type="{_NS_R}/styles" Target="styles.xml"/></Relationships>')
    styles = ('<?xml version="1.0" encoding="UTF-8" standalone="yes"?>\n'
              f'<styleSheet xmlns="{_NS}"><fonts count="1"><font><sz val="11"/><name val="Calibri"/></font></fonts>'
              '<fills count="2"><fill><patternFill patternType="none"/></fill><fill><patternFill patternType="gray125"/></fill></fills>'
              '<borders count="1"><border><left/><right/><top/><bottom/><diagonal/></border></borders>'
              '<cellStyleXfs count="1"><xf numFmtId="0" fontId="0" fillId="0" borderId="0"/></cellStyleXfs>'
              '<cellXfs count="1"><xf numFmtId="0" fontId="0" fillId="0" borderId="0" xfId="0"/></cellXfs>'
              '<cellStyles count="1"><cellStyle name="Normal" xfId="0" builtinId="0"/></cellStyles></styleSheet>')
    # (deflate level 1: the sheets are 10 MB of XML per 20 000 rows and level 6 was two thirds of the call; the file is ~15 % larger)
    with zipfile.ZipFile(path, "w", compression=zipfile.ZIP_DEFLATED, compresslevel=1) as z:
        z.writestr("[Content_Types].xml", content_types)
        z.writestr("_rels/.rels", root_rels)
        z.writestr("xl/workbook.xml", workbook)
        z.writestr("xl/_rels/workbook.xml.rels", wb_rels)
        z.writestr("xl/styles.xml", styles)
        for i, (_, df) in enumerate(tables):
            z.writestr(f"xl/worksheets/sheet{i + 1}.xml", _sheet_xml(df))
    return titles


def _col_index(ref: str) -> int:
    n = 0
    for ch in ref:
        if ch.isalpha():
            n = n * 26 + (ord(ch.upper()) - 64)
    return n - 1


def read_xlsx(path: str) -> Dict[str, pd.DataFrame]:
    """Every sheet of a workbook as a DataFrame (first row = header): inline and shared strings, numbers (int when the
    text is one), booleans; empty cells come back as NaN.  Enough for workbooks written by write_xlsx, pandas or Excel."""
    ns = {"m": _NS}
    with zipfile.ZipFile(path) as z:
        wb = ET.fromstring(z.read("xl/workbook.xml"))
        rels = ET.fromstring(z.read("xl/_rels/workbook.xml.rels"))
        target = {r.get("Id"): r.get("Target") for r in rels}
        shared: List[str] = []
        if "xl/sharedStrings.xml" in z.namelist():
            for si in ET.fromstring(z.read("xl/sharedStrings.xml")).findall("m:si", ns):
                shared.append("".join(t.text or "" for t in si.iter(f"{{{_NS}}}t")))
        out: Dict[str, pd.DataFrame] = {}
        for sh in wb.find("m:sheets", ns):
            t = target[sh.get(f"{{{_NS_R}}}id")]
            part = t.lstrip("/") if t.startswith("/") else "xl/" + t
            rows: List[List] = []
            for row in ET.fromstring(z.read(part)).find("m:sheetData", ns):
                vals: List = []
                for c in row:
                    j = _col_index(c.get("r")) if c.get("r") else len(vals)
                    while len(vals) < j:
                        vals.append(np.nan)
                    typ, v = c.get("t"), c.find("m:v", ns)
                    if typ == "inlineStr":
                        vals.append("".join(x.text or "" for x in c.iter(f"{{{_NS}}}t")))
                    elif v is None or v.text is None:
                        vals.append(np.nan)
                    elif typ == "s":
                        vals.append(shared[int(v.text)])
                    elif typ == "b":
                        vals.append(v.text.strip() == "1")
                    elif typ in ("str", "e"):
                        vals.append(v.text)
                    else:
                        txt = v.text
                        vals.append(int(txt) if re.fullmatch(r"-?\d+", txt) else float(txt))
                rows.append(vals)
            if not rows:
                out[sh.get("name")] = pd.DataFrame()
                continue
            width = max(len(r) for r in rows)
            rows = [r + [np.nan] * (width - len(r)) for r in rows]
            out[sh.get("name")] = pd.DataFrame(rows[1:], columns=rows[0])
    return out

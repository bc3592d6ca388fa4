"""result.xlsx without openpyxl (yacht_amd/xlsx.py): what the reference's own workflow test does with the file --
`pd.read_excel(...)`, then look up organism_name / in_sample_est / num_matches / acceptance_threshold_with_coverage
(tests/test_workflow.py:52-66 there) -- here with a reader made of zipfile + xml.etree, since openpyxl is optional."""
import os
import zipfile
from xml.etree import ElementTree as ET

import numpy as np
import pandas as pd
import pytest

from yacht_amd import xlsx


def _tables():
    df = pd.DataFrame({
        "organism_name": ["CP032507.1 Ectothiorhodospiraceae bacterium BW-2 chromosome, complete genome", "a <b> & \"c\"", "  padded  ", "ünïcode"],
        "in_sample_est": [True, False, True, False],
        "num_matches": np.array([2, 0, 17, 3], dtype=np.uint32),
        "p_vals": [1.0, 0.1 + 0.2, 5e-324, float("nan")],
        "acceptance_threshold_with_coverage": [0.0, 706.0, 1e300, -1.5],
        "min_coverage": [0.001] * 4,
        "maybe": [None, "x", None, "y"],
    })
    return [("min_coverage0.001", df), ("raw_result", df.iloc[:2].reset_index(drop=True)), ("empty", df.iloc[:0])]


def test_roundtrip_values_types_and_sheet_names(tmp_path):
    path = str(tmp_path / "result.xlsx")
    tables = _tables()
    titles = xlsx.write_xlsx(path, tables)
    assert titles == ["min_coverage0.001", "raw_result", "empty"]  # the reference's sheet naming (run_YACHT.py:250-253)
    back = xlsx.read_xlsx(path)
    assert list(back) == titles
    for name, df in tables:
        got = back[name]
        assert list(got.columns) == list(df.columns)
        assert len(got) == len(df)
        for c in df.columns:
            for a, b in zip(df[c].tolist(), got[c].tolist() if len(got) else []):
                if a is None or (isinstance(a, float) and np.isnan(a)):
                    assert isinstance(b, float) and np.isnan(b)
                elif isinstance(a, float):
                    assert float(b) == a  # exact: repr round-trips doubles
                else:
                    assert b == a and type(b) in (type(a), int, str, bool)
    got = back["min_coverage0.001"]
    # the three look-ups of the reference's workflow test
    row = got[got["organism_name"] == "CP032507.1 Ectothiorhodospiraceae bacterium BW-2 chromosome, complete genome"]
    assert str(row["in_sample_est"].values[0]) == "True"
    assert row["num_matches"].values[0] == 2
    assert row["acceptance_threshold_with_coverage"].values[0] == 0


def test_package_parts_are_wellformed_ooxml(tmp_path):
    path = str(tmp_path / "r.xlsx")
    xlsx.write_xlsx(path, _tables())
    with zipfile.ZipFile(path) as z:
        names = set(z.namelist())
        assert {"[Content_Types].xml", "_rels/.rels", "xl/workbook.xml", "xl/_rels/workbook.xml.rels", "xl/styles.xml",
                "xl/worksheets/sheet1.xml", "xl/worksheets/sheet2.xml", "xl/worksheets/sheet3.xml"} <= names
        for n in names:
            ET.fromstring(z.read(n))  # every part parses
        ct = z.read("[Content_Types].xml").decode()
        assert ct.count("worksheet+xml") == 3
        sheet = ET.fromstring(z.read("xl/worksheets/sheet1.xml"))
        ns = {"m": "http://schemas.openxmlformats.org/spreadsheetml/2006/main"}
        cells = sheet.findall(".//m:c", ns)
        assert any(c.get("t") == "b" for c in cells) and any(c.get("t") == "inlineStr" for c in cells)
        refs = [c.get("r") for c in sheet.findall(".//m:row", ns)[1]]
        assert refs[0] == "A2" and refs[1] == "B2"


def test_sheet_titles_follow_excels_rules():
    assert xlsx.sheet_title("min_coverage0.001") == "min_coverage0.001"
    assert xlsx.sheet_title("a/b:c*d?e[f]g\\h") == "a_b_c_d_e_f_g_h"
    long = "x" * 40
    assert len(xlsx.sheet_title(long)) == 31
    assert xlsx.sheet_title("dup", ["dup"]) == "dup_1"
    assert [xlsx._col(i) for i in (0, 25, 26, 27, 701, 702)] == ["A", "Z", "AA", "AB", "ZZ", "AAA"]


def test_pandas_reads_it_when_openpyxl_is_there(tmp_path):
    pytest.importorskip("openpyxl")
    path = str(tmp_path / "r.xlsx")
    tables = _tables()
    xlsx.write_xlsx(path, tables)
    df = pd.read_excel(path)  # first sheet, as the reference's test reads it
    assert list(df.columns) == list(tables[0][1].columns) and len(df) == 4
    assert str(df["in_sample_est"].values[0]) == "True" and df["num_matches"].values[0] == 2


def test_nullable_extension_dtypes_with_missing_values(tmp_path):
    """pandas' nullable Int64 / UInt64 / boolean / string columns report numpy-like kinds while their values
    hold pd.NA: such a column must not take the per-dtype fast path (it wrote `<v><NA></v>` -- a workbook nobody can open).
    NA = an empty cell, every part stays well-formed XML, the other values round-trip."""
    df = pd.DataFrame({
        "name": pd.array(["a", None, "c"], dtype="string"),
        "count": pd.array([1, pd.NA, 3], dtype="Int64"),
        "big": pd.array([2 ** 40 + 5, 7, pd.NA], dtype="UInt64"),
        "flag": pd.array([True, pd.NA, False], dtype="boolean"),
        "plain": np.array([10, 20, 30], dtype=np.int64),
    })
    path = str(tmp_path / "nullable.xlsx")
    xlsx.write_xlsx(path, [("s", df)])
    with zipfile.ZipFile(path) as z:
        for n in z.namelist():
            if n.endswith(".xml") or n.endswith(".rels"):
                ET.fromstring(z.read(n))  # well-formed
        assert b"NA" not in z.read("xl/worksheets/sheet1.xml")
    got = xlsx.read_xlsx(path)["s"]
    assert got["count"].tolist()[0] == 1 and np.isnan(got["count"].tolist()[1]) and got["count"].tolist()[2] == 3
    assert int(got["big"].tolist()[0]) == 2 ** 40 + 5 and np.isnan(got["big"].tolist()[2])
    assert got["flag"].tolist()[0] is True or got["flag"].tolist()[0] == True  # noqa: E712
    assert got["plain"].tolist() == [10, 20, 30]

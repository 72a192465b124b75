"""The only values of this path that the reference itself holds and that can be pinned here: the
rBRIEF sampling pattern and the compile-time constants.  They are re-read from the reference's
source text (present in the build container only; skipped elsewhere) and compared with the
constants compiled into the library and the oracle."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")), reason="reference tree not present")


def _ref(path):
    return open(os.path.join(REF, path), encoding="utf-8", errors="ignore").read()


def _ours(path):
    return open(os.path.join(ROOT, path), encoding="utf-8").read()


def _const(text, name):
    m = re.search(r"\b%s\s*=\s*(-?\d+)\s*;" % re.escape(name), text)
    assert m, name
    return int(m.group(1))


def test_sampling_pattern_equals_the_reference_table():
    src = _ref("src/ORBextractor.cc")
    start = src.index("bit_pattern_31_[256*4]")  # src/ORBextractor.cc:209-467
    body = re.sub(r"/\*.*?\*/", "", src[src.index("{", start) + 1: src.index("};", start)], flags=re.S)
    ref = [int(t) for t in re.findall(r"-?\d+", body)]
    assert len(ref) == 1024
    inc = _ours("orb_slam2_amd/csrc/orb_pattern.inc")
    ours = [int(t) for t in re.findall(r"-?\d+", re.sub(r"//.*", "", inc))]
    assert ours == ref
    # the oracle reads the same file
    assert "orb_pattern.inc" in _ours("oracle/orb_oracle.c")


def test_matcher_and_extractor_constants_equal_the_reference():
    m = _ref("src/ORBmatcher.cc")  # :41-43
    th_high, th_low, histo = _const(m, "ORBmatcher::TH_HIGH"), _const(m, "ORBmatcher::TH_LOW"), _const(m, "ORBmatcher::HISTO_LENGTH")
    e = _ref("src/ORBextractor.cc")  # :81-83
    patch, half, edge = _const(e, "PATCH_SIZE"), _const(e, "HALF_PATCH_SIZE"), _const(e, "EDGE_THRESHOLD")
    f = _ref("include/Frame.h")  # :47,52
    rows = int(re.search(r"#define\s+FRAME_GRID_ROWS\s+(\d+)", f).group(1))
    cols = int(re.search(r"#define\s+FRAME_GRID_COLS\s+(\d+)", f).group(1))
    assert (th_high, th_low, histo, patch, half, edge, rows, cols) == (100, 50, 30, 31, 15, 19, 48, 64)

    hip = _ours("orb_slam2_amd/csrc/orb_match.hip")
    mm = re.search(r"constexpr int TH_HIGH = (\d+), TH_LOW = (\d+), HISTO_LENGTH = (\d+);", hip)
    assert tuple(int(v) for v in mm.groups()) == (th_high, th_low, histo)
    mm = re.search(r"constexpr int GRID_COLS = (\d+), GRID_ROWS = (\d+);", hip)
    assert tuple(int(v) for v in mm.groups()) == (cols, rows)
    common = _ours("orb_slam2_amd/csrc/orb_common.h")
    assert _const(common, "kEdge") == edge and _const(common, "kBorder") == edge - 3
    # the orientation disc and the keypoint size use the patch size
    host = _ours("orb_slam2_amd/csrc/orb_extract_host.hip")
    assert "const int R = %d;" % half in host and "(int)(%d * h->sf[l])" % patch in host

    orc = _ours("oracle/orb_oracle.c")
    for name, val in (("TH_HIGH", th_high), ("TH_LOW", th_low), ("HISTO_LENGTH", histo), ("GRID_COLS", cols),
                      ("GRID_ROWS", rows), ("EDGE_THRESHOLD", edge), ("PATCH_SIZE", patch), ("HALF_PATCH_SIZE", half)):
        mm = re.search(r"#define\s+%s\s+(\d+)" % name, orc) or re.search(r"\b%s\s*=\s*(\d+)" % name, orc)
        assert mm and int(mm.group(1)) == val, name


def test_python_mirror_constants():
    from orb_slam2_amd.matcher import ORBmatcher
    m = _ref("src/ORBmatcher.cc")
    assert ORBmatcher.TH_HIGH == _const(m, "ORBmatcher::TH_HIGH")
    assert ORBmatcher.TH_LOW == _const(m, "ORBmatcher::TH_LOW")
    assert ORBmatcher.HISTO_LENGTH == _const(m, "ORBmatcher::HISTO_LENGTH")

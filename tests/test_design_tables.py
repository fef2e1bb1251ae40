"""DESIGN.md section 4 quotes no hand-typed measurement: its tables are the output of tools/make_design_tables.py over the round's JSON files
(results/<tag>/kernels.json, profiles/<tag>_*.json).  Round 2's DESIGN quoted a FIR time and a BCH rate its own results did not carry."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_design_tables_are_the_generators_output():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_design_tables as M
    blk, _ = M.design_block()
    assert blk is not None, "DESIGN.md lost its GENERATED markers"
    tag = re.search(r"make_design_tables\.py (r\d+)`", blk).group(1)
    assert blk == M.render(tag), "DESIGN.md's generated block is stale or hand-edited: python tools/make_design_tables.py %s --write" % tag
    assert "Headline (bench.py" in blk and "fir_mfma_kernel<2>" in blk and "ldpc_cu1_kernel<27, 3>" in blk and "ldpc_wg8_kernel<27, 5, 0>" in blk

"""Writes tests/golden/cie1931_2deg_5nm.json: the CIE 1931 2-degree observer at 5 nm steps (380..780 nm) as the reference tabulates it
(src/color_conversions.rs CIE_OBSERVERS) -- DATA the second restatement of the white-balance helpers needs (tests/test_oracle_second_restatement.py).
Run in the build container, where /root/reference exists:  python tests/golden/make_cie_fixture.py"""
import json, os, re
src = open("/root/reference/src/color_conversions.rs").read()
rows = re.findall(r"\(\s*(\d{3})\s*,\s*\[\s*([0-9.eE+-]+)\s*,\s*([0-9.eE+-]+)\s*,\s*([0-9.eE+-]+)\s*\]\s*\)", src)
assert len(rows) == 81 and rows[0][0] == "380" and rows[-1][0] == "780", len(rows)
out = [[int(w), x, y, z] for w, x, y, z in rows]      # the decimal literals kept as strings: float(s) is the f64 the Rust literal denotes
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "cie1931_2deg_5nm.json"), "w"))
print(len(out), out[0], out[-1])

"""The number-carrying blocks of DESIGN.md and profiles/README.md are generated from the tracked evidence files (tools/evidence_readme.py):
a block that no longer matches its sources fails here, so documentation cannot quote a figure that a profiles/ file contradicts."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_generated_blocks_are_current():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "evidence_readme.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_design_md_stays_short():
    assert os.path.getsize(os.path.join(ROOT, "DESIGN.md")) < 40 * 1024        # the current state only; history lives under profiles/ (round 6 added the context and versioning sections)

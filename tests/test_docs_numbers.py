"""The measured numbers in DESIGN.md (section 9) and README.md (headline block) are generated from the records under
profiles/r06/ by tools/gen_numbers.py; this test fails when a block no longer matches its records -- nothing measured is
typed by hand, nothing goes stale unnoticed.  DESIGN.md also has to stay a description, not a log: under 40 KB."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_documentation_numbers_match_their_records():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_numbers.py"), "--check"], capture_output=True,
                         text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr


def test_design_md_is_a_description_not_a_log():
    assert os.path.getsize(os.path.join(ROOT, "DESIGN.md")) <= 40 * 1024
    text = open(os.path.join(ROOT, "DESIGN.md")).read()
    for section in ("## 1. The path and its boundary", "## 2. The oracle", "## 3. Data layout in HBM", "## 4. Kernels",
                    "## 6. Multi-GPU", "## 7. Measurement", "<!-- numbers:begin -->"):
        assert section in text, section


def test_bench_main_stays_small():
    """bench.py is a driver over bench_legs/: its main() stays under 300 lines."""
    import re

    text = open(os.path.join(ROOT, "bench.py")).read()
    body = re.search(r"\ndef main\(\):\n(.*?)\nif __name__", text, re.S).group(1)
    assert body.count("\n") < 300
    for leg in ("timed", "repeat", "single", "two_stage", "host", "content", "configs", "match", "cpu", "models", "launch"):
        assert os.path.exists(os.path.join(ROOT, "bench_legs", leg + ".py")), leg

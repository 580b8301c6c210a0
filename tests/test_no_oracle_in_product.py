"""The oracle is test infrastructure: nothing under minkowskiengine_amd/ may import, load or execute
anything under oracle/ (nor /root/reference)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_product_never_touches_the_oracle():
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "minkowskiengine_amd")):
        for f in files:
            if not f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                continue
            text = open(os.path.join(dirpath, f)).read()
            if re.search(r"(from|import)\s+oracle|oracle[/.]|me_oracle|_ref/_C|sys\.path.*reference", text):
                bad.append(os.path.join(dirpath, f))
    assert not bad, f"product files reference the oracle: {bad}"

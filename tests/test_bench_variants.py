"""The A/B experiments under bench_tools/variants/ are edits against the product sources (bench_tools/ab_variants.py
applies them to a scratch copy): every edit's old text must still occur exactly once, or the experiment a profile cites
can no longer be rebuilt."""
import glob
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "swift-homomorphic-encryption_amd", "csrc")


def test_every_variant_still_applies():
    specs = sorted(glob.glob(os.path.join(ROOT, "bench_tools", "variants", "*.py")))
    assert specs
    for path in specs:
        spec = importlib.util.spec_from_file_location("variant", path)
        module = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(module)
        assert module.DESCRIPTION and module.EDITS, path
        for file, old, new in module.EDITS:
            text = open(os.path.join(CSRC, file)).read()
            assert text.count(old) == 1, (os.path.basename(path), file, old[:60])
            assert old != new

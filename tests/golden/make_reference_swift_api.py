"""Writes tests/golden/reference_swift_api.json: for every function / initialiser / enum-case NAME that the Swift package
under swift/Sources/HeAmd (and its tests under swift/Tests/HeAmdTests) calls, every argument-label list under which the REFERENCE declares that name (anywhere under
its Sources/), plus the reference's property and type names the package mentions.  tests/test_swift_reference_api.py
resolves the package's calls against this list where the reference checkout is absent (the GPU box) and re-derives it
where it is present.

    python tests/golden/make_reference_swift_api.py
"""
import glob
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from swift_source import call_signatures, declared_signatures  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(HERE))
REFERENCE = "/root/reference/Sources"


def split_signature(signature):
    name, labels = signature[:-1].split("(", 1)
    return name, [l for l in labels.split(":") if l]


def reference_declarations():
    by_name, bare = {}, set()
    for path in glob.glob(os.path.join(REFERENCE, "**", "*.swift"), recursive=True):
        for signature in declared_signatures(open(path).read()):
            if "(" in signature:
                name, labels = split_signature(signature)
                by_name.setdefault(name, set()).add(tuple(labels))
            else:
                bare.add(signature)
    return by_name, bare


def package_call_names():
    names = set()
    for path in sorted(glob.glob(os.path.join(ROOT, "swift", "Sources", "HeAmd", "*.swift")) +
                       glob.glob(os.path.join(ROOT, "swift", "Tests", "HeAmdTests", "*.swift"))):
        for name, signature, is_init in call_signatures(open(path).read()):
            names.add("init" if is_init else name)
    return names


if __name__ == "__main__":
    by_name, bare = reference_declarations()
    names = package_call_names()
    text = "\n".join(open(p).read() for p in glob.glob(os.path.join(ROOT, "swift", "Sources", "HeAmd", "*.swift")) +
                     glob.glob(os.path.join(ROOT, "swift", "Tests", "HeAmdTests", "*.swift")))
    mentioned = sorted(b for b in bare if re.search(r"\b%s\b" % re.escape(b), text))
    out = {"source": "declarations under /root/reference/Sources (the checkout of this build)",
           "label_lists": {n: sorted(list(l) for l in by_name[n]) for n in sorted(names) if n in by_name},
           "names": mentioned}
    path = os.path.join(HERE, "reference_swift_api.json")
    json.dump(out, open(path, "w"), indent=0)
    print(len(out["label_lists"]), "called names resolve to reference declarations,", len(mentioned), "bare names ->", path)

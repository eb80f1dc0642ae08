"""Every call the Swift package makes resolves to a declaration: the reference's (label for label, defaulted parameters
aside), the package's own, the C header's, or a short list of standard-library members.  There is no Swift toolchain in
this image, so this is the nearest thing to a type-check of swift/Sources/HeAmd against the reference's API: a misspelt
argument label, a reference function that does not exist, or a call to a helper the package never defined fails here.

The reference's declarations come from /root/reference/Sources when that checkout exists (this container) and from
tests/golden/reference_swift_api.json otherwise (the GPU box); the two are held equal when both are present.
"""
import glob
import json
import os

import pytest

from swift_source import (call_signatures, declared_signatures, header_functions, is_ordered_subset,
                          memberwise_initialisers, signature_parts)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PACKAGE = sorted(glob.glob(os.path.join(ROOT, "swift", "Sources", "HeAmd", "*.swift")))
GOLDEN = os.path.join(ROOT, "tests", "golden", "reference_swift_api.json")
REFERENCE = "/root/reference/Sources"

# Standard library / Foundation / concurrency members the package calls (name -> label lists).
STANDARD = {
    "assumingMemoryBound": [["to"]], "bindMemory": [["to", "capacity"]], "compactMap": [["_"]], "map": [["_"]],
    "enumerated": [[]], "fromOpaque": [["_"]], "toOpaque": [[]], "passRetained": [["_"]], "takeRetainedValue": [[]],
    "lock": [[]], "unlock": [[]], "max": [["_", "_"]], "min": [["_", "_"], ["by"]], "zip": [["_", "_"]],
    "reduce": [["_", "_"]], "resume": [[], ["throwing"], ["returning"]], "withExtendedLifetime": [["_", "_"]],
    "withUnsafeBufferPointer": [["_"]], "withUnsafeMutableBufferPointer": [["_"]], "withUnsafeBytes": [["_"]],
    "withCheckedThrowingContinuation": [["_"]], "allSatisfy": [["_"]], "contains": [["_"], ["where"]],
    "append": [["_"], ["contentsOf"]], "reserveCapacity": [["_"]], "first": [["where"]], "firstIndex": [["of"], ["where"]],
    "flatMap": [["_"]], "filter": [["_"]], "forEach": [["_"]], "sorted": [[], ["by"]], "joined": [[], ["separator"]],
    "removeAll": [[], ["keepingCapacity"]], "removeValue": [["forKey"]], "index": [["forKey"]], "hash": [["into"]],
    "combine": [["_"]], "stride": [["from", "to", "by"], ["from", "through", "by"]], "dropFirst": [[], ["_"]],
    "prefix": [["_"]], "suffix": [["_"]], "indices": [[]], "reversed": [[]], "isMultiple": [["of"]], "swapAt": [["_", "_"]],
    "update": [["repeating", "count"], ["from", "count"]], "initialize": [["repeating", "count"], ["from", "count"]],
    "load": [["fromByteOffset", "as"], ["as"]], "advanced": [["by"]], "deallocate": [[]], "allocate": [["capacity"]],
    "release": [[]], "retain": [[]], "description": [[]], "unsafeBitCast": [["_", "to"]], "type": [["of"]],
    "truncatingIfNeeded": [["_"]],
}
# Initialisers of standard types: any label list is accepted for these type names.
STANDARD_TYPES = {"Array", "String", "Int", "UInt8", "UInt32", "UInt64", "Int32", "Int64", "UInt", "Bool", "Set", "Dictionary",
                  "UnsafePointer", "UnsafeMutablePointer", "UnsafeRawPointer", "UnsafeMutableRawPointer", "Unmanaged",
                  "UnsafeBufferPointer", "UnsafeMutableBufferPointer", "NSLock", "ObjectIdentifier", "Optional", "Double",
                  "CheckedContinuation", "Data", "Range", "ClosedRange", "Task"}


def _collect(paths):
    by_name, bare = {}, set()
    for path in paths:
        for signature in declared_signatures(open(path).read()):
            if "(" in signature:
                name, labels = signature_parts(signature)
                by_name.setdefault(name, []).append(labels)
            else:
                bare.add(signature)
    return by_name, bare


def _package_calls():
    for path in PACKAGE:
        text = open(path).read()
        for name, signature, is_init in call_signatures(text):
            yield os.path.basename(path), name, signature_parts(signature)[1], is_init


def _reference_from_golden():
    golden = json.load(open(GOLDEN))
    return golden["label_lists"], set(golden["names"])


def _reference_from_checkout():
    return _collect(glob.glob(os.path.join(REFERENCE, "**", "*.swift"), recursive=True))


def _resolve(reference_by_name, reference_bare):
    own_by_name, own_bare = _collect(PACKAGE)
    own_structs = {}
    for path in PACKAGE:
        own_structs.update(memberwise_initialisers(open(path).read()))
    c_functions = set(header_functions(open(os.path.join(ROOT, "swift", "Sources", "CHeAmd", "include", "he_amd.h")).read()))
    unresolved, resolved_in_reference = [], set()
    for path, name, labels, is_init in _package_calls():
        key = "init" if is_init or name == "init" else name
        if name.startswith("he_") or name.startswith("hip"):
            if name in c_functions or name == "he_status":
                continue
            unresolved.append((path, name, labels, "not in he_amd.h"))
            continue
        if is_init and name in STANDARD_TYPES:
            continue
        if any(is_ordered_subset(labels, declared) for declared in reference_by_name.get(key, [])):
            resolved_in_reference.add(key)
            continue
        if any(is_ordered_subset(labels, declared) for declared in own_by_name.get(key, [])):
            continue
        if is_init and name in own_structs and labels == own_structs[name]:
            continue
        if any(is_ordered_subset(labels, declared) for declared in STANDARD.get(name, [])):
            continue
        if not is_init and (name in own_bare or name in reference_bare) and "_" * len(labels) == "".join(labels):
            continue  # a closure-typed property or local called with unlabelled arguments: read(), translate(...)
        unresolved.append((path, name, labels, "no declaration with these labels"))
    return unresolved, resolved_in_reference


def test_every_call_resolves_against_the_committed_reference_declarations():
    reference_by_name, reference_bare = _reference_from_golden()
    unresolved, resolved = _resolve(reference_by_name, reference_bare)
    assert not unresolved, "\n".join("%s: %s(%s) -- %s" % (p, n, ":".join(l), why) for p, n, l, why in unresolved)
    # the check is not vacuous: the scheme and PIR surface really is reached through the reference's names
    for name in ("init", "validateEquality", "encrypt", "generateEvaluationKey", "zeroCiphertextCoeff", "invalidBatchSize"):
        assert name in resolved, name
    assert len(resolved) >= 25, sorted(resolved)


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="the reference checkout is only in the build container")
def test_committed_reference_declarations_match_the_checkout():
    golden_by_name, golden_bare = _reference_from_golden()
    checkout_by_name, checkout_bare = _reference_from_checkout()
    for name, label_lists in golden_by_name.items():
        assert sorted(label_lists) == sorted(list(l) for l in {tuple(x) for x in checkout_by_name[name]}), name
    assert golden_bare <= checkout_bare
    unresolved, _ = _resolve(checkout_by_name, checkout_bare)
    assert not unresolved, unresolved


def test_a_wrong_label_is_caught():
    """The reader really distinguishes labels: a call with a label the reference does not declare is unresolved."""
    reference_by_name, _ = _reference_from_golden()
    assert any(is_ordered_subset(["config", "using"], d) for d in reference_by_name["generateEvaluationKey"])
    assert not any(is_ordered_subset(["config", "with"], d) for d in reference_by_name["generateEvaluationKey"])
    assert not any(is_ordered_subset(["using", "config"], d) for d in reference_by_name["generateEvaluationKey"])
    assert not any(is_ordered_subset(["databaseCount", "queryCount"], d) for d in reference_by_name["invalidBatchSize"])


# ---- the reference's generic suites over the drop-in types (swift/Tests/HeAmdTests/ReferenceSuites.swift) ----------------
TEST_SOURCES = sorted(glob.glob(os.path.join(ROOT, "swift", "Tests", "HeAmdTests", "*.swift")))
TEST_STANDARD = {"expect": [["_"], ["throws"], ["_", "_"]], "require": [["_"]], "shuffle": [[]], "random": [["in"]]}


def test_every_call_of_the_swift_tests_resolves():
    """The calls swift/Tests/HeAmdTests makes -- above all the reference's generic suites instantiated with GpuBfv /
    GpuPirUtil -- resolve label for label to declarations of the reference (its _TestUtilities product included), of the
    package, or of the tests themselves."""
    reference_by_name, reference_bare = _reference_from_golden()
    own_by_name, own_bare = _collect(PACKAGE + TEST_SOURCES)
    unresolved = []
    for path in TEST_SOURCES:
        for name, signature, is_init in call_signatures(open(path).read()):
            labels = signature_parts(signature)[1]
            key = "init" if is_init or name == "init" else name
            if is_init and name in STANDARD_TYPES:
                continue
            candidates = (reference_by_name.get(key, []) + own_by_name.get(key, []) + STANDARD.get(name, []) +
                          TEST_STANDARD.get(name, []))
            if any(is_ordered_subset(labels, declared) for declared in candidates):
                continue
            if not is_init and (name in own_bare or name in reference_bare) and "_" * len(labels) == "".join(labels):
                continue
            unresolved.append((os.path.basename(path), name, labels))
    assert not unresolved, unresolved


def test_reference_suites_cover_what_the_reference_runs_for_bfv():
    """ReferenceSuites.swift calls every HeAPITestHelpers.* the reference's runBfvTests calls for Bfv<T>
    (Tests/HomomorphicEncryptionTests/HeAPITests.swift:176-221) -- with GpuBfv.self -- plus the index-PIR suite with the
    GpuPirUtil server and the public indexPir(scheme:) entry point."""
    import re

    text = open(os.path.join(ROOT, "swift", "Tests", "HeAmdTests", "ReferenceSuites.swift")).read()
    ours = set(re.findall(r"HeAPITestHelpers\.(\w+)\(", text))
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_bfv_suite_calls.json")))
    assert set(golden["helpers"]) <= ours, sorted(set(golden["helpers"]) - ours)
    for helper in golden["helpers"]:
        if helper != "schemeEvaluationKeyTest":  # (context:) only
            assert re.search(r"HeAPITestHelpers\.%s\(context: context, scheme: GpuBfv\.self\)" % helper, text), helper
    assert "indexPirTest(\n            server: MulPirServer<GpuPirUtil<GpuBfv>>.self" in text
    assert "IndexPirTests.indexPir(scheme: GpuBfv.self)" in text
    assert "@testable import _TestUtilities" in text
    package = open(os.path.join(ROOT, "swift", "Package.swift")).read()
    assert '.product(name: "_TestUtilities", package: "swift-homomorphic-encryption")' in package
    reference_tests = "/root/reference/Tests/HomomorphicEncryptionTests/HeAPITests.swift"
    if os.path.exists(reference_tests):
        body = open(reference_tests).read()
        body = body[body.index("private func runBfvTests"):body.index("func bfvUInt32")]
        assert sorted(set(re.findall(r"HeAPITestHelpers\.(\w+)\(", body))) == sorted(golden["helpers"])

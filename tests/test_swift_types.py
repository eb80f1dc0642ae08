"""The Swift drop-ins against the reference's protocols at the level of TYPES (no Swift toolchain in the image, so nothing
here compiles Swift): every requirement of `HeScheme` (Sources/HomomorphicEncryption/HeScheme.swift:190-1090) that
`GpuBfv` has to supply itself, and every requirement of `PirUtilProtocol`
(Sources/PrivateInformationRetrieval/IndexPir/PirUtil.swift:22-147), must have a witness whose parameter labels, ownership
modifiers (`inout`, `consuming`), parameter and return types (typealiases expanded on both sides), `async` / `throws` and
generic / `where` clauses are the requirement's.  The requirements travel as tests/golden/swift_protocol_signatures.json
(re-derived from the reference checkout whenever it is present, like the other goldens); the checker itself is held by
mutations of real signatures that must each be reported."""
import copy
import glob
import json
import os

import swift_types as st

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = "/root/reference/Sources"
GOLDEN = os.path.join(ROOT, "tests", "golden", "swift_protocol_signatures.json")


def _read(*paths):
    return "\n".join(open(p).read() for p in paths)


def _derive_from_reference():
    he = _read(os.path.join(REFERENCE, "HomomorphicEncryption", "HeScheme.swift"))
    he_extensions = _read(os.path.join(REFERENCE, "HomomorphicEncryption", "HeScheme.swift"),
                          os.path.join(REFERENCE, "HomomorphicEncryption", "HeSchemeAsync.swift"))
    pir = _read(os.path.join(REFERENCE, "PrivateInformationRetrieval", "IndexPir", "PirUtil.swift"))
    return {
        "source": "HeScheme.swift:190-1090 + extensions (HeScheme.swift, HeSchemeAsync.swift); PirUtil.swift:22-147 + extension "
                  "(reference checkout of this build)",
        "HeScheme": {
            "requirements": st.static_function_signatures(he, "HeScheme", kinds=("protocol",)),
            "aliases": st.typealiases(he, "HeScheme", kinds=("protocol",)),
            "defaults": st.static_function_signatures(he_extensions, "HeScheme", kinds=("extension",)),
        },
        "PirUtilProtocol": {
            "requirements": st.static_function_signatures(pir, "PirUtilProtocol", kinds=("protocol",)),
            "aliases": st.typealiases(pir, "PirUtilProtocol", kinds=("protocol",)),
            "defaults": st.static_function_signatures(pir, "PirUtilProtocol", kinds=("extension",)),
        },
    }


def _listify(value):  # tuples -> lists, as JSON round-trips them
    return json.loads(json.dumps(value))


def _protocols():
    if os.path.isdir(REFERENCE):
        derived = _listify(_derive_from_reference())
        stored = json.load(open(GOLDEN)) if os.path.exists(GOLDEN) else None
        if stored != derived:
            with open(GOLDEN, "w") as f:
                json.dump(derived, f, indent=1, sort_keys=True)
                f.write("\n")
    protocols = json.load(open(GOLDEN))
    for protocol in ("HeScheme", "PirUtilProtocol"):
        for group in ("requirements", "defaults"):
            for signature in protocols[protocol][group]:
                signature["parameters"] = [tuple(p) for p in signature["parameters"]]
    return protocols


def _gpu_bfv():
    text = _read(*sorted(glob.glob(os.path.join(ROOT, "swift", "Sources", "HeAmd", "GpuBfv*.swift"))))
    return st.static_function_signatures(text, "GpuBfv"), st.typealiases(text, "GpuBfv")


def _gpu_pir_util():
    text = _read(os.path.join(ROOT, "swift", "Sources", "HeAmd", "GpuPirUtil.swift"))
    witnesses = st.static_function_signatures(text, "GpuPirUtil")
    # `enum GpuPirUtil<Scheme: HeScheme>`: Scheme is the generic parameter on both sides; Scalar == Scheme.Scalar
    return witnesses, {"Scalar": "Scheme.Scalar", "CanonicalCiphertext": "Scheme.CanonicalCiphertext"}


def _required_he_scheme_keys():
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", "noop_scheme_members.json")))
    return set(golden["noop_scheme_static_members"]) | set(golden["accelerated_protocol_requirements"])


def test_golden_holds_the_protocols():
    protocols = _protocols()
    assert len(protocols["HeScheme"]["requirements"]) >= 90 and len(protocols["PirUtilProtocol"]["requirements"]) == 7
    keys = {st.key_of(r) for r in protocols["HeScheme"]["requirements"]}
    assert {"mulAssign(_:_:)", "relinearizeAsync(_:using:)", "applyGalois(ciphertext:element:using:)",
            "innerProduct(ciphertexts:plaintexts:)", "forwardNtt(_:)"} <= keys
    relinearize = [r for r in protocols["HeScheme"]["requirements"] if r["name"] == "relinearizeAsync"][0]
    assert relinearize["is_async"] and relinearize["throws"] and relinearize["parameters"][0][1] == "inout"


def test_gpu_bfv_witnesses_have_the_requirements_types():
    protocol = _protocols()["HeScheme"]
    witnesses, aliases = _gpu_bfv()
    problems = st.conformance_mismatches(protocol["requirements"], witnesses, protocol["aliases"], aliases, "GpuBfv",
                                         required_keys=_required_he_scheme_keys(), defaults=protocol["defaults"])
    assert not problems, "\n".join(problems)


def test_gpu_pir_util_witnesses_have_the_requirements_types():
    protocol = _protocols()["PirUtilProtocol"]
    witnesses, aliases = _gpu_pir_util()
    problems = st.conformance_mismatches(protocol["requirements"], witnesses, protocol["aliases"], aliases, "GpuPirUtil",
                                         defaults=protocol["defaults"])
    assert not problems, "\n".join(problems)


def _mutate(witnesses, name, change):
    """A copy of `witnesses` with `change` applied to every overload called `name` (at least one must exist)."""
    mutated = copy.deepcopy(witnesses)
    hits = [w for w in mutated if w["name"] == name]
    assert hits, name
    for w in hits:
        change(w)
    return mutated


def test_the_checker_reports_wrong_signatures():
    """What the round-4 verdict asked the check to fail on -- a wrong `inout`, a missing `async throws`, a wrong generic
    constraint -- and a wrong parameter / return type, each injected into the real witness lists."""
    he, pir = _protocols()["HeScheme"], _protocols()["PirUtilProtocol"]
    witnesses, aliases = _gpu_bfv()
    required = _required_he_scheme_keys()

    def he_problems(mutated):
        return st.conformance_mismatches(he["requirements"], mutated, he["aliases"], aliases, "GpuBfv", required_keys=required,
                                         defaults=he["defaults"])

    def drop_inout(w):
        w["parameters"] = [(label, "", t) for label, own, t in w["parameters"]]

    def drop_async(w):
        w["is_async"] = False

    def drop_throws(w):
        w["throws"] = False

    def wrong_return(w):
        w["returns"] = "CoeffCiphertext"

    def wrong_parameter(w):
        w["parameters"] = [(label, own, "EvalCiphertext") for label, own, t in w["parameters"]]

    assert any("relinearize(" in p for p in he_problems(_mutate(witnesses, "relinearize", drop_inout)))
    assert any("mulAssignAsync" in p for p in he_problems(_mutate(witnesses, "mulAssignAsync", drop_async)))
    assert any("modSwitchDownAsync" in p for p in he_problems(_mutate(witnesses, "modSwitchDownAsync", drop_throws)))
    assert any("forwardNtt(" in p for p in he_problems(_mutate(witnesses, "forwardNtt", wrong_return)))
    assert any("relinearize(" in p for p in he_problems(_mutate(witnesses, "relinearize", wrong_parameter)))
    assert any("missing witness" in p and "applyGalois" in p
               for p in he_problems([w for w in witnesses if w["name"] != "applyGalois"]))

    pir_witnesses, pir_aliases = _gpu_pir_util()

    def pir_problems(mutated):
        return st.conformance_mismatches(pir["requirements"], mutated, pir["aliases"], pir_aliases, "GpuPirUtil", defaults=pir["defaults"])

    def wrong_constraint(w):
        w["generics"] = [g.replace("Sendable&", "") for g in w["generics"]]

    def drop_where(w):
        w["where"] = []

    def not_consuming(w):
        w["parameters"] = [(label, "" if own == "consuming" else own, t) for label, own, t in w["parameters"]]

    assert any("computeResponseForOneChunk" in p for p in pir_problems(_mutate(pir_witnesses, "computeResponseForOneChunk", wrong_constraint)))
    assert any("computeResponseForOneChunk" in p for p in pir_problems(_mutate(pir_witnesses, "computeResponseForOneChunk", drop_where)))
    assert any("expand(" in p for p in pir_problems(_mutate(pir_witnesses, "expand", not_consuming)))


def test_the_reader_parses_declarations_as_swift_means_them():
    text = """
    public enum Example<S: HeScheme>: P {
        public static func f<A: Sendable & Collection<Int>, B>(
            _ x: inout [S.Scalar], using key: consuming EvaluationKey<S>,
            callback: @escaping (Int) -> Void = { _ in }) async throws -> [Ciphertext<S, Coeff>]
            where A.Index == Int, B: Equatable
        { fatalError() }
        static func g(of value: borrowing Foo) -> Bool { true }
        static func h() {}
    }
    """
    f, g, h = st.static_function_signatures(text, "Example")
    assert f["name"] == "f" and f["generics"] == ["A:Sendable&Collection<Int>", "B"]
    assert f["parameters"] == [("_", "inout", "[S.Scalar]"), ("using", "consuming", "EvaluationKey<S>"), ("callback", "", "(Int)->Void")]
    assert f["is_async"] and f["throws"] and f["returns"] == "[Ciphertext<S,Coeff>]" and f["where"] == ["A.Index==Int", "B:Equatable"]
    assert g["parameters"] == [("of", "borrowing", "Foo")] and not g["is_async"] and not g["throws"] and g["returns"] == "Bool"
    assert h["parameters"] == [] and h["returns"] == "Void"
    aliases = {"CoeffCiphertext": "Ciphertext<Self,Coeff>", "SecretKey": "HomomorphicEncryption.SecretKey<Self>", "Scalar": "UInt64"}
    assert st.canonical_type("[CoeffCiphertext]", aliases, "GpuBfv") == "[Ciphertext<$,Coeff>]"
    assert st.canonical_type("SecretKey<GpuBfv>", aliases, "GpuBfv") == st.canonical_type("SecretKey", aliases, "GpuBfv") == "SecretKey<$>"
    assert st.canonical_type("someCollection<Scalar.SignedScalar>", aliases, "GpuBfv") == "someCollection<Int64>"

"""Type-level reading of Swift SOURCE TEXT (no Swift toolchain exists in the build image): the full signature of every
`static func` of a type -- generic clause, each parameter's label, ownership modifier and type, `async` / `throws`, return
type, `where` clause -- normalized so that a protocol requirement written against `Self` and a conformer's witness written
against the concrete type compare equal exactly when Swift would accept one for the other (up to the typealiases both
sides declare).  tests/test_swift_types.py holds `GpuBfv` to `HeScheme` and `GpuPirUtil` to `PirUtilProtocol` with it."""
import re

from swift_source import balanced, split_top_level, strip_comments

OWNERSHIP = ("inout", "consuming", "borrowing", "__owned", "__shared", "sending")
# associated types of standard-library conformances this reader cannot look up (ScalarType.SignedScalar,
# Sources/ModularArithmetic/Scalar.swift: UInt32 -> Int32, UInt64 -> Int64)
KNOWN_MEMBER_TYPES = {"UInt64.SignedScalar": "Int64", "UInt32.SignedScalar": "Int32"}


def _balanced_angle(text, start):
    """text[start] == '<' -> index just past the matching '>' ('->' arrows inside are not brackets)."""
    depth, i = 0, start
    while i < len(text):
        if text.startswith("->", i):
            i += 2
            continue
        if text[i] == "<":
            depth += 1
        elif text[i] == ">":
            depth -= 1
            if depth == 0:
                return i + 1
        i += 1
    raise ValueError("unbalanced <>")


def type_bodies(text, type_name, kinds=("enum", "protocol", "struct", "class", "extension")):
    """Bodies of every `enum|protocol|struct|class|extension type_name ... { }` in comment-stripped text."""
    pattern = r"\b(?:%s)\s+%s\b[^{]*\{" % ("|".join(kinds), re.escape(type_name))
    for head in re.finditer(pattern, text):
        yield text[head.end():balanced(text, head.end() - 1, "{", "}") - 1]


def _declaration_end(body, start):
    """End of the declaration that starts at `start` (just past the parameter list): the body's opening brace for a
    definition, or -- a protocol requirement -- the next member."""
    depth, i = 0, start
    while i < len(body):
        if body.startswith("->", i):
            i += 2
            continue
        c = body[i]
        if c in "(<[":
            depth += 1
        elif c in ")>]":
            depth -= 1
        elif c == "{" and depth == 0:
            return i
        elif c == "\n" and depth == 0:
            rest = body[i:].lstrip()
            if not rest or re.match(r"(static|public|internal|private|fileprivate|associatedtype|typealias|func|var|let|init|@|\}|//|mutating|nonisolated)\b|[@}]", rest):
                return i
        i += 1
    return len(body)


def _squash(text):
    return re.sub(r"\s+", "", text)


def parse_parameter(text):
    """'label name: modifiers Type = default' -> (label, ownership, type) with whitespace removed from the type."""
    head, _, rest = text.partition(":")
    names = head.split()
    label = names[0] if names else "_"
    depth, cut = 0, len(rest)
    for i, c in enumerate(rest):  # a default value is not part of the signature
        if rest.startswith("->", i):
            continue
        if c in "(<[":
            depth += 1
        elif c in ")>]":
            depth -= 1
        elif c == "=" and depth == 0 and rest[i:i + 2] != "==":
            cut = i
            break
    words = rest[:cut].split()
    ownership = ""
    while words and (words[0] in OWNERSHIP or words[0].startswith("@")):
        if words[0] in OWNERSHIP:
            ownership = words[0]
        words = words[1:]
    return label, ownership, _squash(" ".join(words))


def static_function_signatures(text, type_name, kinds=("enum", "protocol", "struct", "class", "extension")):
    """[{name, generics, parameters: [(label, ownership, type)], is_async, throws, returns, where}] of every `static func`
    declared in the bodies of `type_name` (its extensions included)."""
    text = strip_comments(text)
    found = []
    for body in type_bodies(text, type_name, kinds):
        for match in re.finditer(r"\bstatic\s+func\s+([A-Za-z_][A-Za-z0-9_]*)\s*", body):
            i = match.end()
            generics = ""
            if i < len(body) and body[i] == "<":
                end = _balanced_angle(body, i)
                generics = body[i + 1:end - 1]
                i = end
            while i < len(body) and body[i].isspace():
                i += 1
            if i >= len(body) or body[i] != "(":
                continue
            end = balanced(body, i)
            parameters = [parse_parameter(p) for p in split_top_level(body[i + 1:end - 1].replace("->", "\x00\x00"))]
            parameters = [(label, own, t.replace("\x00\x00", "->")) for label, own, t in parameters]
            tail = body[end:_declaration_end(body, end)]
            effects, _, after_arrow = tail.partition("->")
            where = ""
            if "where" in re.split(r"\W+", after_arrow if after_arrow else effects):
                source = after_arrow if after_arrow else effects
                at = re.search(r"\bwhere\b", source).start()
                where = source[at + len("where"):]
                if after_arrow:
                    after_arrow = source[:at]
                else:
                    effects = source[:at]
            generic_list = sorted(_squash(g) for g in split_top_level(generics) if g.strip())
            found.append({
                "name": match.group(1),
                "generics": generic_list,
                "parameters": parameters,
                "is_async": bool(re.search(r"\basync\b", effects)),
                "throws": bool(re.search(r"\b(re)?throws\b", effects)),
                "returns": _squash(after_arrow) or "Void",
                "where": sorted(_squash(w) for w in split_top_level(where) if w.strip()),
            })
    return found


def typealiases(text, type_name, kinds=("enum", "protocol", "struct", "class", "extension")):
    """{alias: target} declared in the bodies of `type_name`."""
    text = strip_comments(text)
    out = {}
    for body in type_bodies(text, type_name, kinds):
        for match in re.finditer(r"\btypealias\s+([A-Za-z_][A-Za-z0-9_]*)\s*=\s*([^\n]+)", body):
            out[match.group(1)] = _squash(match.group(2))
    return out


def canonical_type(type_text, aliases, self_name):
    """A type spelled in canonical form: module prefixes dropped, `Self` / `Self.X` / the conformer's own name folded to
    `$`, typealiases expanded (bare identifiers only: `SecretKey<...>` is the generic type, `SecretKey` the alias)."""
    text = type_text
    for _ in range(8):
        before = text
        text = re.sub(r"\bHomomorphicEncryption\.", "", text)
        text = re.sub(r"\bSelf\.", "", text)
        text = re.sub(r"\b(?:Self|%s)\b(?!\.)" % re.escape(self_name), "$", text)

        def expand(match):
            word = match.group(0)
            end = match.end()
            if end < len(text) and text[end] == "<":  # the generic type of that name, not the alias
                return word
            start = match.start()
            if start > 0 and text[start - 1] == ".":  # a member of another type
                return word
            return aliases.get(word, word)

        text = re.sub(r"[A-Za-z_][A-Za-z0-9_]*", expand, text)
        for spelled, meant in KNOWN_MEMBER_TYPES.items():
            text = text.replace(spelled, meant)
        if text == before:
            break
    return text


def canonical_signature(signature, aliases, self_name):
    """Hashable canonical form of a signature (see canonical_type)."""
    canon = lambda t: canonical_type(t, aliases, self_name)  # noqa: E731
    return (
        signature["name"],
        tuple(signature["generics"] and [canon(g) for g in signature["generics"]]),
        tuple((label, own, canon(t)) for label, own, t in signature["parameters"]),
        signature["is_async"],
        signature["throws"],
        canon(signature["returns"]),
        tuple(canon(w) for w in signature["where"]),
    )


def key_of(signature):
    """name(label:label:) -- what overloads are told apart by before their types are looked at."""
    return "%s(%s)" % (signature["name"], "".join(label + ":" for label, _, _ in signature["parameters"]))


def describe(canonical):
    name, generics, parameters, is_async, throws, returns, where = canonical
    text = name + ("<%s>" % ", ".join(generics) if generics else "")
    text += "(" + ", ".join("%s: %s%s" % (label, own + " " if own else "", t) for label, own, t in parameters) + ")"
    text += (" async" if is_async else "") + (" throws" if throws else "") + " -> " + returns
    return text + (" where " + ", ".join(where) if where else "")


def conformance_mismatches(requirements, witnesses, requirement_aliases, witness_aliases, self_name, required_keys=None,
                           defaults=()):
    """Every requirement (optionally only those whose key is in required_keys) must have a witness with the same canonical
    signature.  Returns a list of human-readable mismatches (empty: conforms).

    Requirement types are expanded with the protocol's aliases AND the conformer's bindings of its associated types (the
    conformer's take precedence: `Scalar` is UInt64 for a conformer that says so)."""
    bindings = dict(requirement_aliases)
    bindings.update(witness_aliases)
    have = {}
    for witness in witnesses:
        have.setdefault(key_of(witness), set()).add(canonical_signature(witness, bindings, self_name))
    # `defaults`: the static functions of the protocol's extensions -- a requirement they implement needs no witness
    defaulted = {canonical_signature(d, bindings, self_name) for d in defaults}
    generic_defaults = [d for d in defaults if d["generics"]]

    def specializes(default, want):
        """A generic default (`addAssign<L: PolyFormat, R: PolyFormat>(_: inout Ciphertext<Self, L>, _: Ciphertext<Self, R>)`)
        implements a concrete requirement when its types match with the generic parameters read as wildcards."""
        name, _, parameters, is_async, throws, returns, _ = want
        have_name, _, have_parameters, have_async, have_throws, have_returns, _ = canonical_signature(default, bindings, self_name)
        if (name, is_async, throws, len(parameters)) != (have_name, have_async, have_throws, len(have_parameters)):
            return False
        names = [g.split(":")[0] for g in default["generics"]]

        def matches(pattern_type, concrete):
            pattern = re.escape(pattern_type)
            for generic in names:
                pattern = re.sub(r"(?<![A-Za-z0-9_.])%s(?![A-Za-z0-9_])" % re.escape(generic), r"[A-Za-z_][A-Za-z0-9_]*", pattern)
            return re.fullmatch(pattern, concrete) is not None

        return matches(have_returns, returns) and all(
            (label, own) == (have_label, have_own) and matches(have_type, t)
            for (label, own, t), (have_label, have_own, have_type) in zip(parameters, have_parameters))

    problems = []
    for requirement in requirements:
        key = key_of(requirement)
        if required_keys is not None and key not in required_keys and key not in have:
            continue  # a requirement with a default implementation that the conformer does not re-declare
        want = canonical_signature(requirement, bindings, self_name)
        candidates = have.get(key)
        if candidates and want in candidates:
            continue
        if want in defaulted or any(specializes(d, want) for d in generic_defaults):
            continue
        if not candidates:
            problems.append("missing witness for " + describe(want))
        elif want not in candidates:
            problems.append("no witness of %s matches the requirement\n    required: %s\n    declared: %s" % (
                key, describe(want), "\n              ".join(sorted(describe(c) for c in candidates))))
    # A declaration that shares name and labels with requirements but matches none of them is not a witness: where the
    # protocol has a default (every `...Async` twin does) the package would still compile -- and silently run the default.
    wanted = {}
    for requirement in requirements:
        wanted.setdefault(key_of(requirement), set()).add(canonical_signature(requirement, bindings, self_name))
    for key, candidates in sorted(have.items()):
        for candidate in sorted(candidates - wanted.get(key, candidates)):
            problems.append("%s is declared with the labels of a requirement but the types of none\n    declared: %s\n    required: %s" % (
                key, describe(candidate), "\n              ".join(sorted(describe(w) for w in wanted[key]))))
    return problems

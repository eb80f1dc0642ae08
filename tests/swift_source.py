"""A small reader of Swift SOURCE TEXT (no Swift toolchain exists in the build image): enough structure to hold the
Swift package under swift/ to the C header and to the reference's protocol by name, label and argument count."""
import re


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", "", text)


def balanced(text, start, open_char="(", close_char=")"):
    """text[start] == open_char -> index just past the matching close_char."""
    depth = 0
    for i in range(start, len(text)):
        c = text[i]
        if c == open_char:
            depth += 1
        elif c == close_char:
            depth -= 1
            if depth == 0:
                return i + 1
    raise ValueError("unbalanced")


def split_top_level(text, sep=","):
    parts, depth, current = [], 0, []
    for c in text:
        if c in "([{<":
            depth += 1
        elif c in ")]}>":
            depth -= 1
        if c == sep and depth == 0:
            parts.append("".join(current))
            current = []
        else:
            current.append(c)
    rest = "".join(current).strip()
    if rest:
        parts.append(rest)
    return [p.strip() for p in parts if p.strip()]


def c_calls(text, prefix="he_"):
    """[(name, argument count)] of every call `he_xxx(...)` in Swift source text."""
    text = strip_comments(text)
    calls = []
    for match in re.finditer(r"\b(%s[a-z0-9_]+)\s*\(" % prefix, text):
        end = balanced(text, match.end() - 1)
        args = text[match.end():end - 1]
        # `->` inside closures would unbalance the <> counting of split_top_level: closures do not occur in C calls here
        calls.append((match.group(1), len(split_top_level(args.replace("->", "  ")))))
    return calls


def header_functions(header_text):
    """{name: parameter count} of the C prototypes in include/he_amd.h."""
    text = re.sub(r"/\*.*?\*/", " ", header_text, flags=re.S)
    functions = {}
    for match in re.finditer(r"\b(he_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.S):
        params = match.group(2).strip()
        functions[match.group(1)] = 0 if params in ("", "void") else len(split_top_level(params))
    return functions


def static_members(text, type_name, kind="enum"):
    """Signature keys of the static members declared in the body of `enum|protocol type_name` (extensions included
    when kind == 'enum+extensions')."""
    text = strip_comments(text)
    members = set()
    pattern = r"\b(?:public\s+)?(?:enum|protocol|struct|extension)\s+%s\b[^{]*\{" % re.escape(type_name)
    for head in re.finditer(pattern, text):
        body = text[head.end():balanced(text, head.end() - 1, "{", "}") - 1]
        for match in re.finditer(r"\bstatic\s+func\s+([A-Za-z_][A-Za-z0-9_]*)\s*(<[^(]*?>)?\s*\(", body):
            # generic clauses may span lines and contain parentheses only in rare cases; find the parameter list
            open_index = body.index("(", match.start(1))
            if match.group(2):
                open_index = body.index("(", match.end(2))
            end = balanced(body, open_index)
            labels = []
            for parameter in split_top_level(body[open_index + 1:end - 1].replace("->", "  ")):
                names = parameter.split(":")[0].split()
                labels.append(names[0] if names else "_")
            members.add("%s(%s)" % (match.group(1), "".join(label + ":" for label in labels)))
        for match in re.finditer(r"\bstatic\s+var\s+([A-Za-z_][A-Za-z0-9_]*)", body):
            members.add("var " + match.group(1))
    return members


SWIFT_KEYWORDS = {"if", "guard", "while", "for", "switch", "return", "throw", "try", "await", "catch", "in", "where", "case",
                  "let", "var", "func", "init", "else", "do", "defer", "precondition", "preconditionFailure", "fatalError"}


def strip_strings(text):
    """String literals (interpolations included) -> "": their contents are not code for the purposes of this reader."""
    out, i, n = [], 0, len(text)
    while i < n:
        c = text[i]
        if c == '"':
            j = i + 1
            depth = 0
            while j < n:
                if text[j] == "\\" and j + 1 < n and text[j + 1] == "(":
                    depth += 1
                    j += 2
                    continue
                if depth and text[j] == "(":
                    depth += 1
                elif depth and text[j] == ")":
                    depth -= 1
                elif not depth and text[j] == "\\":
                    j += 2
                    continue
                elif not depth and text[j] == '"':
                    break
                j += 1
            out.append('""')
            i = j + 1
        else:
            out.append(c)
            i += 1
    return "".join(out)


def _labels(parameter_text, declaration):
    labels = []
    for part in split_top_level(parameter_text.replace("->", "  ")):
        if declaration:
            names = part.split(":")[0].split()
            labels.append(names[0] if names else "_")
        else:
            match = re.match(r"^([A-Za-z_][A-Za-z0-9_]*)\s*:(?!:)", part)
            labels.append(match.group(1) if match else "_")
    return "".join(label + ":" for label in labels)


def declared_signatures(text):
    """{"name(label:...)"} for every func / init / enum case with payload declared in Swift source text, plus the bare
    names of properties, cases and types ("name")."""
    text = strip_strings(strip_comments(text))
    found = set()
    for match in re.finditer(r"\b(func\s+([A-Za-z_][A-Za-z0-9_]*|[-+*/=<>!&|^~%]+)|init[?!]?|case\s+([A-Za-z_][A-Za-z0-9_]*))\s*(<[^(){}]*?>)?\s*\(",
                             text):
        open_index = match.end() - 1
        try:
            end = balanced(text, open_index)
        except ValueError:
            continue
        head = match.group(1)
        name = "init" if head.startswith("init") else (match.group(2) or match.group(3))
        found.add("%s(%s)" % (name, _labels(text[open_index + 1:end - 1], declaration=True)))
    for match in re.finditer(r"\b(?:var|let|case|class|struct|enum|protocol|typealias|associatedtype|actor)\s+([A-Za-z_][A-Za-z0-9_]*)", text):
        found.add(match.group(1))
    for match in re.finditer(r"\bcase\s+([A-Za-z_][A-Za-z0-9_, ]*)\n", text):
        for name in match.group(1).split(","):
            found.add(name.strip())
    return found


def call_signatures(text):
    """[(name, "name(label:...)")] of every call with at least one labelled argument or through a member access in Swift
    source text: `x.name(...)`, `name(...)`, `Type(...)` / `Type<...>(...)` (reported as init)."""
    text = strip_strings(strip_comments(text))
    calls = []
    for match in re.finditer(r"(\.)?\b([A-Za-z_][A-Za-z0-9_]*)\s*(<[^(){};=]*?>)?\(", text):
        name = match.group(2)
        if name in SWIFT_KEYWORDS:
            continue
        before = text[max(0, match.start() - 12):match.start()]
        if re.search(r"\b(func|init|case|class|struct|enum)\s+$", before) or re.search(r"\bfunc\s+$", text[max(0, match.start() - 6):match.start()]):
            continue  # a declaration, not a call
        open_index = match.end() - 1
        try:
            end = balanced(text, open_index)
        except ValueError:
            continue
        labels = _labels(text[open_index + 1:end - 1], declaration=False)
        if match.start() and text[match.start() - 1] == "@":
            continue  # an attribute (@escaping (...) -> ..., @inline(__always))
        is_init = name.lstrip("_")[:1].isupper()  # Type(...), Module.Type(...), _Type(...); enum cases are lower case
        calls.append((name, "%s(%s)" % ("init" if is_init else name, labels), is_init))
    return calls


def signature_parts(signature):
    """"name(a:b:)" -> ("name", ["a", "b"])."""
    name, labels = signature[:-1].split("(", 1)
    return name, [label for label in labels.split(":") if label]


def is_ordered_subset(call_labels, declared_labels):
    """A call may leave out defaulted parameters and a trailing closure: its labels appear, in order, among the declared."""
    position = 0
    for label in call_labels:
        while position < len(declared_labels) and declared_labels[position] != label:
            position += 1
        if position == len(declared_labels):
            return False
        position += 1
    return True


def memberwise_initialisers(text):
    """{"Name": [stored property names in order]} for every struct in Swift source text that declares no init."""
    text = strip_strings(strip_comments(text))
    found = {}
    for match in re.finditer(r"\bstruct\s+([A-Za-z_][A-Za-z0-9_]*)[^{]*\{", text):
        end = balanced(text, match.end() - 1, "{", "}")
        body = text[match.end():end - 1]
        if re.search(r"\binit\s*[(<]", body):
            continue
        depth, fields = 0, []
        for line in body.split("\n"):
            if depth == 0:
                field = re.match(r"\s*(?:public\s+|private\s+|fileprivate\s+)?(?:let|var)\s+([A-Za-z_][A-Za-z0-9_]*)\s*:[^{]*$", line)
                if field:
                    fields.append(field.group(1))
            depth += line.count("{") - line.count("}")
        found[match.group(1)] = fields
    return found

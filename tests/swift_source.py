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

"""Minimal protobuf-text parser for Caffe prototxt files (the subset the MNC graphs use).

parse(text) -> Message: a dict-like object where every field maps to a LIST of values (scalars or nested Messages),
in file order.  No caffe.proto is needed; field semantics are applied by mnc_amd.engine."""
import re

_TOKEN = re.compile(r"""\s*(?:(\#[^\n]*)|([{}:])|("(?:[^"\\]|\\.)*"|'(?:[^'\\]|\\.)*')|([^\s{}:#"']+))""")


class Message(dict):
    def get1(self, key, default=None):
        v = self.get(key)
        return v[0] if v else default

    def all(self, key):
        return self.get(key, [])


def _tokens(text):
    pos = 0
    while pos < len(text):
        m = _TOKEN.match(text, pos)
        if not m:
            if text[pos:].strip() == "":
                return
            raise ValueError("prototxt: cannot tokenize at %r" % text[pos:pos + 30])
        pos = m.end()
        if m.group(1):
            continue
        if m.group(2):
            yield ("p", m.group(2))
        elif m.group(3):
            s = m.group(3)
            yield ("s", bytes(s[1:-1], "utf-8").decode("unicode_escape"))
        else:
            yield ("w", m.group(4))


def _scalar(kind, tok):
    if kind == "s":
        return tok
    if tok in ("true", "false"):
        return tok == "true"
    try:
        return int(tok)
    except ValueError:
        pass
    try:
        return float(tok)
    except ValueError:
        return tok          # enum identifier, e.g. MAX


def parse(text):
    toks = list(_tokens(text))
    pos = 0

    def message(depth):
        nonlocal pos
        msg = Message()
        while pos < len(toks):
            kind, tok = toks[pos]
            if kind == "p" and tok == "}":
                if depth == 0:
                    raise ValueError("prototxt: unbalanced '}'")
                pos += 1
                return msg
            if kind != "w":
                raise ValueError("prototxt: expected a field name, got %r" % (tok,))
            name = tok
            pos += 1
            kind, tok = toks[pos]
            if kind == "p" and tok == ":":
                pos += 1
                kind, tok = toks[pos]
            if kind == "p" and tok == "{":
                pos += 1
                msg.setdefault(name, []).append(message(depth + 1))
            else:
                pos += 1
                msg.setdefault(name, []).append(_scalar(kind, tok))
        if depth:
            raise ValueError("prototxt: missing '}'")
        return msg

    return message(0)


def parse_file(path):
    with open(path) as f:
        return parse(f.read())

#!/usr/bin/env python3
"""Extract the literal case tables of the reference's own tests into JSON fixtures.

Reads (as text, in THIS container only; /root/reference does not travel):
  pkg/sql/colexec/hashjoiner_test.go      getHJTestCases()        (:50-925)
  pkg/sql/colexec/mergejoiner_test.go     getMJTestCases()        (run through the hash joiner,
                                                                   hashjoiner_test.go:1022)
  pkg/sql/colexec/aggregators_test.go     aggregatorsTestCases    (:186-785)
  pkg/sql/colexec/hash_aggregator_test.go hashAggregatorTestCases (:34-397)
  pkg/sql/colexec/distinct_test.go        distinctTestCases       (:45-330)

Only DATA is extracted: column types, input tuples, specs and expected tuples.
`coldata.BatchSize()` is evaluated at its default of 1024 (pkg/col/coldata/batch.go:79).
Values the GPU path does not model (apd decimals, intervals, TimeTZ datums)
are kept as {"unsupported": "<go expr>"} and the case is marked
"supported": false; see SURVEY.md 8(a) "Type coverage decision".

Usage: python tests/golden/extract_reference_cases.py [/root/reference]
"""
import json
import os
import re
import sys

BATCH_SIZE = 1024


class Tok:
    def __init__(self, kind, val, pos):
        self.kind, self.val, self.pos = kind, val, pos

    def __repr__(self):
        return f"{self.kind}:{self.val!r}"


TOKEN_RE = re.compile(r"""
    (?P<ws>\s+|//[^\n]*|/\*.*?\*/)
  | (?P<str>"(?:[^"\\]|\\.)*"|`[^`]*`)
  | (?P<num>\d+\.\d*(?:[eE][-+]?\d+)?|\.\d+|\d+(?:[eE][-+]?\d+)?)
  | (?P<id>[A-Za-z_][A-Za-z_0-9]*(?:\.[A-Za-z_][A-Za-z_0-9]*)*)
  | (?P<p>[{}\[\](),:*+\-&/])
""", re.X | re.S)


def tokenize(text, start):
    toks, pos = [], start
    while pos < len(text):
        m = TOKEN_RE.match(text, pos)
        if not m:
            raise SyntaxError(f"bad char {text[pos]!r} at {pos}")
        pos = m.end()
        if m.lastgroup == "ws":
            continue
        toks.append(Tok(m.lastgroup, m.group(m.lastgroup), m.start()))
        yield toks[-1]


class Unsupported:
    def __init__(self, text):
        self.text = text


class Parser:
    def __init__(self, text, start):
        self.text = text
        self.toks = []
        self.gen = tokenize(text, start)
        self.i = 0

    def peek(self, k=0):
        while len(self.toks) <= self.i + k:
            self.toks.append(next(self.gen))
        return self.toks[self.i + k]

    def next(self):
        t = self.peek()
        self.i += 1
        return t

    def accept(self, val):
        if self.peek().val == val and self.peek().kind == "p":
            self.i += 1
            return True
        return False

    def expect(self, val):
        t = self.next()
        if t.val != val:
            raise SyntaxError(f"expected {val!r} got {t} near {self.text[t.pos-40:t.pos+40]!r}")

    # type prefix: ([])* (*)? ident?   followed by '{'
    def try_type_prefix(self):
        save = self.i
        n = 0
        while self.peek().val == "[" and self.peek(1).val == "]":
            self.i += 2
            n += 1
        if self.peek().val == "*":
            self.i += 1
            n += 1
        if self.peek().kind == "id":
            self.i += 1
            n += 1
        if self.peek().val == "{" and self.peek().kind == "p":
            return True
        self.i = save
        return False

    def parse_value(self):
        if self.peek().val == "{" and self.peek().kind == "p":
            return self.parse_composite()
        if self.peek().val == "&":
            self.i += 1
        if self.try_type_prefix():
            return self.parse_composite()
        return self.parse_expr()

    def parse_composite(self):
        self.expect("{")
        items, d = [], None
        while not (self.peek().val == "}" and self.peek().kind == "p"):
            if self.peek().kind == "id" and self.peek(1).val == ":" and self.peek(1).kind == "p":
                key = self.next().val
                self.next()
                if d is None:
                    d = {}
                d[key] = self.parse_value()
            else:
                items.append(self.parse_value())
            if not self.accept(","):
                break
        self.expect("}")
        return d if d is not None else items

    def parse_expr(self):
        v = self.parse_term()
        while self.peek().kind == "p" and self.peek().val in "*+-/":
            op = self.next().val
            r = self.parse_term()
            if isinstance(v, Unsupported) or isinstance(r, Unsupported):
                v = Unsupported("expr")
            elif op == "*":
                v = v * r
            elif op == "+":
                v = v + r
            elif op == "-":
                v = v - r
            else:
                v = v / r
        return v

    def parse_args_text(self):
        """consume a balanced (...) and return its text"""
        start = self.peek().pos
        depth = 0
        while True:
            t = self.next()
            if t.kind == "p" and t.val in "([{":
                depth += 1
            elif t.kind == "p" and t.val in ")]}":
                depth -= 1
                if depth == 0:
                    return self.text[start:t.pos + 1]

    def parse_term(self):
        t = self.next()
        if t.kind == "num":
            return float(t.val) if any(c in t.val for c in ".eE") else int(t.val)
        if t.kind == "str":
            return json.loads(t.val) if t.val[0] == '"' else t.val[1:-1]
        if t.kind == "p" and t.val == "-":
            v = self.parse_term()
            return -v
        if t.kind == "id":
            name = t.val
            nxt = self.peek()
            if nxt.kind == "p" and nxt.val == "(":
                if name in ("int", "int16", "int32", "int64", "float64", "uint32"):
                    self.next()
                    v = self.parse_expr()
                    self.expect(")")
                    return v
                if name == "coldata.BatchSize":
                    self.parse_args_text()
                    return BATCH_SIZE
                args = self.parse_args_text()
                return Unsupported(name + args)
            if nxt.kind == "p" and nxt.val == "[":
                args = self.parse_args_text()
                return Unsupported(name + args)
            if name == "nil":
                return None
            if name == "true":
                return True
            if name == "false":
                return False
            if name == "tree.NoColumnIdx":
                return -1
            if name.startswith("types."):
                return {"types.Int": "int", "types.Int2": "int2", "types.Int4": "int4", "types.Float": "float",
                        "types.Bool": "bool", "types.Bytes": "bytes", "types.String": "string",
                        "types.Decimal": "decimal", "types.Interval": "interval", "types.TimeTZ": "timetz",
                        "types.TwoIntCols": ["int", "int"], "types.OneIntCol": ["int"],
                        "types.ThreeIntCols": ["int", "int", "int"]}.get(name, name)
            if name.startswith(("descpb.", "execinfrapb.")):
                return name.split(".", 1)[1]
            return Unsupported(name)
        raise SyntaxError(f"unexpected token {t} near {self.text[t.pos-40:t.pos+40]!r}")


def to_json(v, flags):
    if isinstance(v, Unsupported):
        flags["unsupported"] = True
        return {"unsupported": v.text}
    if isinstance(v, list):
        return [to_json(x, flags) for x in v]
    if isinstance(v, dict):
        return {k: to_json(x, flags) for k, x in v.items()}
    return v


def parse_table(path, anchor_re):
    text = open(path).read()
    m = re.search(anchor_re, text)
    if not m:
        raise RuntimeError(f"anchor {anchor_re} not found in {path}")
    brace = text.index("{", m.end() - 1)
    p = Parser(text, brace)
    cases = p.parse_composite()
    return text, cases, p


JOIN_TYPES = {"InnerJoin": "INNER", "LeftOuterJoin": "LEFT_OUTER", "RightOuterJoin": "RIGHT_OUTER",
              "FullOuterJoin": "FULL_OUTER", "LeftSemiJoin": "LEFT_SEMI", "LeftAntiJoin": "LEFT_ANTI",
              "IntersectAllJoin": "INTERSECT_ALL", "ExceptAllJoin": "EXCEPT_ALL",
              "RightSemiJoin": "RIGHT_SEMI", "RightAntiJoin": "RIGHT_ANTI"}
GPU_TYPES = {"int", "int2", "int4", "float", "bool", "bytes", "string"}


def line_of(text, pos):
    return text.count("\n", 0, pos) + 1


def join_cases(path, anchor, relpath):
    text, cases, _p = parse_table(path, anchor)
    out = []
    for c in cases:
        flags = {}
        jc = {
            "description": c.get("description", ""),
            "join_type": JOIN_TYPES[c.get("joinType", "InnerJoin")],
            "left_types": c["leftTypes"], "right_types": c["rightTypes"],
            "left": c.get("leftTuples", []), "right": c.get("rightTuples", []),
            "left_eq": c.get("leftEqCols", []), "right_eq": c.get("rightEqCols", []),
            "left_out": c.get("leftOutCols", []), "right_out": c.get("rightOutCols", []),
            "left_eq_cols_are_key": bool(c.get("leftEqColsAreKey", False)),
            "right_eq_cols_are_key": bool(c.get("rightEqColsAreKey", False)),
            "expected": c.get("expected", []),
        }
        if "onExpr" in c:
            jc["on_expr"] = c["onExpr"]["Expr"]
        jc = to_json(jc, flags)
        types_ok = all(t in GPU_TYPES for t in jc["left_types"] + jc["right_types"])
        jc["supported"] = bool(types_ok and not flags.get("unsupported"))
        jc["source"] = relpath
        out.append(jc)
    return out


AGG_FUNCS = {"AnyNotNull": "ANY_NOT_NULL", "Avg": "AVG", "BoolAnd": "BOOL_AND", "BoolOr": "BOOL_OR",
             "ConcatAgg": "CONCAT_AGG", "Count": "COUNT", "Max": "MAX", "Min": "MIN", "Sum": "SUM",
             "SumInt": "SUM_INT", "CountRows": "COUNT_ROWS"}


def agg_cases(path, anchor, relpath):
    text, cases, _p = parse_table(path, anchor)
    out = []
    for c in cases:
        flags = {}
        typs = c["typs"]
        ac = {
            "name": c["name"], "types": typs, "input": c["input"],
            "group_cols": c.get("groupCols", []), "agg_cols": c["aggCols"],
            "agg_fns": [AGG_FUNCS.get(f, f) for f in c["aggFns"]],
            "expected": c["expected"],
            "unordered_input": bool(c.get("unorderedInput", False)),
            "ordered_cols": c.get("orderedCols", []),
            "conv_to_decimal": bool(c.get("convToDecimal", False)),
        }
        if "aggDistinct" in c:
            ac["agg_distinct"] = c["aggDistinct"]
        if "aggFilter" in c:
            ac["agg_filter"] = c["aggFilter"]
        ac = to_json(ac, flags)
        ac["has_unsupported_values"] = bool(flags.get("unsupported"))
        ac["source"] = relpath
        out.append(ac)
    return out


def distinct_cases(path, anchor, relpath):
    text, cases, _p = parse_table(path, anchor)
    out = []
    for c in cases:
        flags = {}
        dc = {
            "distinct_cols": c["distinctCols"], "types": c["typs"], "tuples": c["tuples"], "expected": c.get("expected", []),
            "ordered_on_distinct_cols": bool(c.get("isOrderedOnDistinctCols", False)),
            "nulls_are_distinct": bool(c.get("nullsAreDistinct", False)),
            "error_on_dup": c.get("errorOnDup", ""), "no_error": bool(c.get("noError", False)),
        }
        dc = to_json(dc, flags)
        dc["supported"] = bool(all(t in GPU_TYPES for t in dc["types"]) and not flags.get("unsupported"))
        dc["source"] = relpath
        out.append(dc)
    return out


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    here = os.path.dirname(os.path.abspath(__file__))
    colexec = os.path.join(ref, "pkg/sql/colexec")
    hj = join_cases(os.path.join(colexec, "hashjoiner_test.go"), r"hjTestCases := \[\]\*joinTestCase\{",
                    "pkg/sql/colexec/hashjoiner_test.go:50-925")
    mj = join_cases(os.path.join(colexec, "mergejoiner_test.go"), r"mjTestCases := \[\]\*joinTestCase\{",
                    "pkg/sql/colexec/mergejoiner_test.go (getMJTestCases)")
    ag = agg_cases(os.path.join(colexec, "aggregators_test.go"), r"var aggregatorsTestCases = \[\]aggregatorTestCase\{",
                   "pkg/sql/colexec/aggregators_test.go:186-785")
    ha = agg_cases(os.path.join(colexec, "hash_aggregator_test.go"),
                   r"var hashAggregatorTestCases = \[\]aggregatorTestCase\{",
                   "pkg/sql/colexec/hash_aggregator_test.go:34-397")
    di = distinct_cases(os.path.join(colexec, "distinct_test.go"), r"var distinctTestCases = \[\]distinctTestCase\{",
                        "pkg/sql/colexec/distinct_test.go:45-330")
    for name, data in [("hashjoiner_cases.json", hj), ("mergejoiner_cases.json", mj),
                       ("aggregator_cases.json", ag), ("hash_aggregator_cases.json", ha), ("distinct_cases.json", di)]:
        with open(os.path.join(here, name), "w") as f:
            json.dump({"batch_size_assumed": BATCH_SIZE, "cases": data}, f, indent=1)
        print(name, len(data), "cases")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Transcribe known-answer vectors from the reference's Catch2 unit tests.

Run in the dev container only (needs /root/reference); the JSON it writes under
tests/golden/ is committed so the GPU box never needs the reference tree.

    python tests/golden/extract_reference_tests.py

Sources (all under /root/reference/src/unittest/):
    aligner.cpp              gssw local alignment, full-length bonus
    pinned_alignment.cpp     gssw pinned alignment (+ multi)
    xdrop_aligner.cpp        dozeu X-drop alignment
    banded_global_aligner.cpp  BandedGlobalAligner

The tests are literal tiny graphs built in code followed by REQUIRE lines on the
resulting Alignment; this script evaluates exactly those literal statements (it
never executes reference code).  Each leaf SECTION (or section-less TEST_CASE)
becomes one case per aligner call:

    {"source": "src/unittest/x.cpp:LINE", "name": ..., "nodes": [[id, seq]...],
     "edges": [[from, to]...], "read": ..., "quality": [...]|null,
     "scores": [match, mismatch, gap_open, gap_extend, bonus], "qual_adj": bool,
     "call": "align_pinned", "args": {...}, "aln": "aln",
     "expect": [[kind, ...]...], "unparsed": [raw REQUIRE text...]}

Expectation kinds:
    ["score", v] ["mapping_size", v] ["node_id", m, v] ["offset", m, v]
    ["is_reverse", m, bool] ["edit_size", m, v] ["from_length", m, e, v]
    ["to_length", m, e, v] ["sequence", m, e, str] ["mapping_from_length", m, v]
    ["mapping_to_length", m, v] ["rank", m, v] ["path_from_length", v]
    ["score_minus", other_aln, delta]   (score == score of the sibling case whose
                                         "aln" is other_aln, same "source", + delta)
Mapping index may be negative (-1 = last, from `path.mapping_size() - 1`).
"""
import json
import os
import re
import sys

REF = "/root/reference/src/unittest"
OUT = os.path.dirname(os.path.abspath(__file__))


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", lambda m: "\n" * m.group(0).count("\n"), text, flags=re.S)
    return re.sub(r"//[^\n]*", "", text)


def find_block(text, open_idx):
    """text[open_idx] == '{' -> index of matching '}'."""
    depth = 0
    i = open_idx
    in_str = False
    while i < len(text):
        ch = text[i]
        if in_str:
            if ch == "\\":
                i += 1
            elif ch == '"':
                in_str = False
        else:
            if ch == '"':
                in_str = True
            elif ch == "{":
                depth += 1
            elif ch == "}":
                depth -= 1
                if depth == 0:
                    return i
        i += 1
    raise ValueError("unbalanced")


SEC_RE = re.compile(r"\b(TEST_CASE|SECTION)\s*\(\s*\"((?:[^\"\\]|\\.)*)\"")


def parse_sections(text, base_line=1):
    """Return list of (kind, title, line, body_text, body_start_line) for top-level
    TEST_CASE/SECTION blocks in text."""
    out = []
    pos = 0
    while True:
        m = SEC_RE.search(text, pos)
        if not m:
            break
        brace = text.index("{", m.end())
        end = find_block(text, brace)
        line = base_line + text.count("\n", 0, m.start())
        out.append((m.group(1), m.group(2), line, text[brace + 1:end], base_line + text.count("\n", 0, brace + 1), m.start(), end + 1))
        pos = end + 1
    return out


def leaf_programs(text, base_line=1, prefix="", title_path=()):
    """Yield (title_path, line, program_text) for every leaf section: ancestors'
    non-section code + the leaf body (Catch2 re-runs the enclosing code per leaf)."""
    secs = parse_sections(text, base_line)
    # code of this level with child section blocks blanked out
    own = text
    for s in reversed(secs):
        own = own[:s[5]] + own[s[6]:]
    if not secs:
        yield title_path, base_line, prefix + "\n" + text
        return
    for kind, title, line, body, body_line, _, _ in secs:
        # code before the section runs before; code after runs after (rare). Put own code first.
        child_prefix = prefix + "\n" + own
        sub = parse_sections(body, body_line)
        if sub:
            yield from leaf_programs(body, body_line, child_prefix, title_path + (title,))
        else:
            yield title_path + (title,), line, child_prefix + "\n" + body


def split_statements(prog):
    # crude statement splitter: split on ';' outside strings/parens
    stmts = []
    cur = []
    depth = 0
    in_str = False
    for i, ch in enumerate(prog):
        if in_str:
            cur.append(ch)
            if ch == '"' and prog[i - 1] != "\\":
                in_str = False
            continue
        if ch == '"':
            in_str = True
            cur.append(ch)
        elif ch in "([":
            depth += 1
            cur.append(ch)
        elif ch in ")]":
            depth -= 1
            cur.append(ch)
        elif ch == ";" and depth == 0:
            stmts.append("".join(cur).strip())
            cur = []
        elif ch in "{}" and depth == 0:
            s = "".join(cur).strip()
            if s:
                stmts.append(s)
            cur = []
        else:
            cur.append(ch)
    s = "".join(cur).strip()
    if s:
        stmts.append(s)
    return [re.sub(r"\s+", " ", s) for s in stmts if s]


class Env:
    def __init__(self):
        self.nodes = []          # [id, seq]
        self.node_var = {}       # var -> id
        self.edges = []
        self.strings = {}        # var -> str
        self.ints = {}           # var -> int
        self.bools = {}
        self.aln_seq = {}        # aln var -> read
        self.aln_qual = {}       # aln var -> list[int]
        self.sources = {}        # aligner_source var -> [m,x,go,ge,b]
        self.aligners = {}       # aligner var -> (source var, qual_adj)
        self.path_alias = {}     # path var -> aln var
        self.ptr_alias = {}      # Node* alias var -> node var
        self.calls = []          # dict
        self.expects = {}        # aln var -> list
        self.unparsed = []
        self.mems = {}           # mem vector var -> list of dict
        self.alternatives = {}   # 'first' -> [expectation...]: what one of the returned alignments must look like (is_first_opt ... found_first_opt)
        self.alt_scores = []     # REQUIRE(alt_aln.score() == N) inside the loop over the returned alignments


DEFAULT_SCORES = [1, 4, 6, 1, 5]


def eval_int(expr, env):
    """Evaluate small integer expressions used on the right of REQUIREs."""
    e = expr.strip()
    e = re.sub(r"\(\s*(?:size_t|int|int32_t|int64_t|id_t|vg::id_t)\s*\)", "", e)
    def repl_seq_len(m):
        v = m.group(1)
        v = env.ptr_alias.get(v, v)
        if v in env.node_var:
            return str(len(dict(env.nodes)[env.node_var[v]]))
        raise KeyError(v)
    e = re.sub(r"(\w+)->sequence\(\)\.(?:length|size)\(\)", repl_seq_len, e)
    e = re.sub(r"graph\.get_length\((\w+)\)", repl_seq_len, e)
    def repl_id(m):
        v = m.group(1)
        v = env.ptr_alias.get(v, v)
        return str(env.node_var[v])
    e = re.sub(r"(\w+)->id\(\)", repl_id, e)
    e = re.sub(r"graph\.get_id\((\w+)\)", repl_id, e)
    def repl_strlen(m):
        v = m.group(1)
        if v in env.strings:
            return str(len(env.strings[v]))
        raise KeyError(v)
    e = re.sub(r"(\w+)\.(?:size|length)\(\)", repl_strlen, e)
    def repl_alnlen(m):
        return str(len(env.aln_seq[m.group(1)]))
    e = re.sub(r"(\w+)\.sequence\(\)\.(?:size|length)\(\)", repl_alnlen, e)
    for k, v in env.ints.items():
        e = re.sub(r"\b%s\b" % re.escape(k), str(v), e)
    if not re.fullmatch(r"[\d\s+\-*/()]+", e):
        raise ValueError("cannot evaluate: " + expr)
    return int(eval(e, {"__builtins__": {}}))


def eval_str(expr, env):
    e = expr.strip()
    m = re.fullmatch(r"(?:std::)?(?:string\()?\"([^\"]*)\"\)?", e)
    if m:
        return m.group(1)
    m = re.fullmatch(r"(\w+)\.sequence\(\)", e)
    if m and m.group(1) in env.aln_seq:
        return env.aln_seq[m.group(1)]
    if e in env.strings:
        return env.strings[e]
    m = re.fullmatch(r"(\w+)\.substr\((.+),(.+)\)", e)
    if m and m.group(1) in env.strings:
        a = eval_int(m.group(2), env); b = eval_int(m.group(3), env)
        return env.strings[m.group(1)][a:a + b]
    raise ValueError("cannot evaluate string: " + expr)


MAP_IDX = r"(path\.mapping_size\(\) - \d+|\w+\.path\(\)\.mapping_size\(\) - \d+|\d+)"


def map_index(txt):
    txt = txt.strip()
    m = re.search(r"mapping_size\(\) - (\d+)$", txt)
    if m:
        return -int(m.group(1))
    return int(txt)


def parse_require(cond, env):
    """Return (aln_var, expectation) or None."""
    cond = cond.strip()
    # normalise path accessors:  X.path().mapping(  /  path.mapping(
    def aln_of(prefix):
        prefix = prefix.strip()
        m = re.fullmatch(r"(\w+)\.path\(\)", prefix)
        if m:
            return m.group(1)
        if prefix in env.path_alias:
            return env.path_alias[prefix]
        return None

    m = re.fullmatch(r"(\w+)\.score\(\) == (\w+)\.score\(\)(?: \+ (.+))?", cond)
    if m and m.group(1) in env.aln_seq and m.group(2) in env.aln_seq:
        return m.group(1), ["score_minus", m.group(2), eval_int(m.group(3), env) if m.group(3) else 0]
    m = re.fullmatch(r"(\w+)\.score\(\) == (.+)", cond)
    if m and m.group(1) in env.aln_seq:
        return m.group(1), ["score", eval_int(m.group(2), env)]
    m = re.fullmatch(r"(.+?)\.mapping_size\(\) == (.+)", cond)
    if m and aln_of(m.group(1)):
        return aln_of(m.group(1)), ["mapping_size", eval_int(m.group(2), env)]
    m = re.fullmatch(r"(.+?)\.mapping\(" + MAP_IDX + r"\)\.position\(\)\.node_id\(\) == (.+)", cond)
    if m and aln_of(m.group(1)):
        return aln_of(m.group(1)), ["node_id", map_index(m.group(2)), eval_int(m.group(3), env)]
    m = re.fullmatch(r"(.+?)\.mapping\(" + MAP_IDX + r"\)\.position\(\)\.offset\(\) == (.+)", cond)
    if m and aln_of(m.group(1)):
        return aln_of(m.group(1)), ["offset", map_index(m.group(2)), eval_int(m.group(3), env)]
    m = re.fullmatch(r"(.+?)\.mapping\(" + MAP_IDX + r"\)\.position\(\)\.is_reverse\(\) == (true|false)", cond)
    if m and aln_of(m.group(1)):
        return aln_of(m.group(1)), ["is_reverse", map_index(m.group(2)), m.group(3) == "true"]
    m = re.fullmatch(r"(.+?)\.mapping\(" + MAP_IDX + r"\)\.rank\(\) == (.+)", cond)
    if m and aln_of(m.group(1)):
        return aln_of(m.group(1)), ["rank", map_index(m.group(2)), eval_int(m.group(3), env)]
    m = re.fullmatch(r"(.+?)\.mapping\(" + MAP_IDX + r"\)\.edit_size\(\) == (.+)", cond)
    if m and aln_of(m.group(1)):
        return aln_of(m.group(1)), ["edit_size", map_index(m.group(2)), eval_int(m.group(3), env)]
    m = re.fullmatch(r"(.+?)\.mapping\(" + MAP_IDX + r"\)\.edit\((\d+)\)\.(from_length|to_length)\(\) == (.+)", cond)
    if m and aln_of(m.group(1)):
        return aln_of(m.group(1)), [m.group(4), map_index(m.group(2)), int(m.group(3)), eval_int(m.group(5), env)]
    m = re.fullmatch(r"(.+?)\.mapping\(" + MAP_IDX + r"\)\.edit\((\d+)\)\.sequence\(\)\.empty\(\)", cond)
    if m and aln_of(m.group(1)):
        return aln_of(m.group(1)), ["sequence", map_index(m.group(2)), int(m.group(3)), ""]
    m = re.fullmatch(r"(.+?)\.mapping\(" + MAP_IDX + r"\)\.edit\((\d+)\)\.sequence\(\) == (.+)", cond)
    if m and aln_of(m.group(1)):
        return aln_of(m.group(1)), ["sequence", map_index(m.group(2)), int(m.group(3)), eval_str(m.group(4), env)]
    m = re.fullmatch(r"mapping_(from|to)_length\((.+?)\.mapping\(" + MAP_IDX + r"\)\) == (.+)", cond)
    if m and aln_of(m.group(2)):
        return aln_of(m.group(2)), ["mapping_%s_length" % m.group(1), map_index(m.group(3)), eval_int(m.group(4), env)]
    m = re.fullmatch(r"path_(from|to)_length\((.+?)\) == (.+)", cond)
    if m and aln_of(m.group(2)):
        return aln_of(m.group(2)), ["path_%s_length" % m.group(1), eval_int(m.group(3), env)]
    return None


def resolve_bool_ifs(prog):
    """Keep only the taken branch of `if (flag) {...} else {...}` for literal bool flags."""
    bools = {m.group(1): m.group(2) == "true" for m in re.finditer(r"\bbool (\w+) = (true|false)\s*;", prog)}
    out = prog
    while True:
        m = None
        for cand in re.finditer(r"\bif\s*\(\s*(!?)(\w+)\s*\)\s*\{", out):
            if cand.group(2) in bools:
                m = cand
                break
        if not m:
            return out
        val = bools[m.group(2)] != (m.group(1) == "!")
        b0 = out.index("{", m.start())
        e0 = find_block(out, b0)
        then_body = out[b0 + 1:e0]
        rest = out[e0 + 1:]
        m2 = re.match(r"\s*else\s*\{", rest)
        else_body = ""
        tail_start = e0 + 1
        if m2:
            b1 = e0 + 1 + m2.end() - 1
            e1 = find_block(out, b1)
            else_body = out[b1 + 1:e1]
            tail_start = e1 + 1
        out = out[:m.start()] + (then_body if val else else_body) + out[tail_start:]


def run_program(prog):
    env = Env()
    prog = resolve_bool_ifs(prog)
    for st in split_statements(prog):
        try:
            handle_statement(st, env)
        except (KeyError, ValueError) as ex:
            if st.startswith("REQUIRE"):
                env.unparsed.append(st)
    return env


def handle_statement(st, env):
    m = re.fullmatch(r"(?:Node\*|handle_t|auto) (\w+) = graph\.create_(?:node|handle)\(\s*(?:string\()?\"([^\"]*)\"\)?\s*\)", st)
    if m:
        nid = len(env.nodes) + 1
        env.nodes.append([nid, m.group(2)])
        env.node_var[m.group(1)] = nid
        return
    m = re.fullmatch(r"graph\.create_edge\((\w+), (\w+)\)", st)
    if m:
        env.edges.append([env.node_var[m.group(1)], env.node_var[m.group(2)]])
        return
    m = re.fullmatch(r"(?:Node\*|handle_t|auto) (\w+) = graph\.create_handle\(\s*\"([^\"]*)\", (\d+)\)", st)
    if m:      # explicit node id
        nid = int(m.group(3))
        env.nodes.append([nid, m.group(2)])
        env.node_var[m.group(1)] = nid
        return
    m = re.fullmatch(r"string graph_json = R\"\((.*)\)\"", st, re.S)
    if m:      # vg::io::json2graph input: nodes with ids, edges with optional from_start / to_end
        import json as _json
        gj = _json.loads(m.group(1))
        for n in gj.get("node", []):
            env.nodes.append([int(n["id"]), n.get("sequence", "")])
        for e in gj.get("edge", []):
            fs, te = bool(e.get("from_start")), bool(e.get("to_end"))
            if fs != te:
                raise ValueError("reversing edge")
            a, b = int(e["from"]), int(e["to"])
            env.edges.append([b, a] if fs else [a, b])
        return
    m = re.fullmatch(r"(\w+)\.set_quality\(string_quality_short_to_char\((\w+)\)\)", st)
    if m and m.group(1) in env.aln_seq:      # adds 33 to every byte (src/alignment.cpp string_quality_short_to_char)
        env.aln_qual[m.group(1)] = ("raw", [ord(c) + 33 for c in env.strings[m.group(2)]])
        return
    m = re.fullmatch(r"graph\.create_edge\((\w+), (\w+), true, true\)", st)
    if m:      # from the end of a reversed to the start of b reversed == b forward -> a forward
        env.edges.append([env.node_var[m.group(2)], env.node_var[m.group(1)]])
        return
    m = re.fullmatch(r"(\w+) = (\"[^\"]*\"|string\(.+\))", st)
    if m and m.group(1) in env.strings:       # re-assignment between two alignments of one SECTION
        env.strings[m.group(1)] = eval_str(m.group(2), env)
        return
    m = re.fullmatch(r"(?:const )?(?:std::)?string (\w+) = (.+)", st)
    if m:
        try:
            env.strings[m.group(1)] = eval_str(m.group(2), env)
        except ValueError:
            pass
        return
    m = re.fullmatch(r"(?:const )?(?:std::)?string (\w+)\(\"([^\"]*)\"\)", st)
    if m:
        env.strings[m.group(1)] = m.group(2)
        return
    m = re.fullmatch(r"(?:const )?(?:int8_t|int|int32_t|int64_t|uint16_t|size_t|uint64_t) (\w+) = (.+)", st)
    if m:
        try:
            env.ints[m.group(1)] = eval_int(m.group(2), env)
        except (ValueError, KeyError):
            pass
        return
    m = re.fullmatch(r"bool (\w+) = (true|false)", st)
    if m:
        env.bools[m.group(1)] = (m.group(2) == "true")
        return
    m = re.fullmatch(r"Node\* (\w+) = (\w+)", st)
    if m and m.group(2) in env.node_var:
        env.ptr_alias[m.group(1)] = m.group(2)
        return
    m = re.fullmatch(r"TestAligner (\w+)", st)
    if m:
        env.sources[m.group(1)] = list(DEFAULT_SCORES)
        return
    m = re.fullmatch(r"(\w+)\.set_alignment_scores\((.+)\)", st)
    if m and m.group(1) in env.sources:
        env.sources[m.group(1)] = [eval_int(x, env) for x in m.group(2).split(",")]
        return
    m = re.fullmatch(r"const (Aligner|QualAdjAligner)& (\w+) = \*(\w+)\.get_(regular|qual_adj)_aligner\(\)", st)
    if m:
        env.aligners[m.group(2)] = (m.group(3), m.group(4) == "qual_adj")
        return
    m = re.fullmatch(r"Alignment (.+)", st)
    if m:
        for v in m.group(1).split(","):
            env.aln_seq.setdefault(v.strip(), None)
        return
    m = re.fullmatch(r"(\w+)\.set_sequence\((.+)\)", st)
    if m and m.group(1) in env.aln_seq:
        env.aln_seq[m.group(1)] = eval_str(m.group(2), env)
        return
    m = re.fullmatch(r"(\w+)\.set_quality\((.+)\)", st)
    if m and m.group(1) in env.aln_seq:
        q = eval_str(m.group(2), env)
        env.aln_qual[m.group(1)] = ("ascii", q)
        return
    m = re.fullmatch(r"alignment_quality_char_to_short\((\w+)\)", st)
    if m and m.group(1) in env.aln_qual:
        kind, q = env.aln_qual[m.group(1)]
        env.aln_qual[m.group(1)] = ("raw", [ord(c) - 33 for c in q])
        return
    m = re.fullmatch(r"const Path& (\w+) = (\w+)\.path\(\)", st)
    if m:
        env.path_alias[m.group(1)] = m.group(2)
        return
    # MEM construction in xdrop tests
    m = re.fullmatch(r"vector<MaximalExactMatch> (\w+)", st)
    if m:
        env.mems[m.group(1)] = []
        return
    m = re.fullmatch(r"(\w+)\.emplace_back\(\)", st)
    if m and m.group(1) in env.mems:
        env.mems[m.group(1)].append({"begin": None, "end": None, "nodes": []})
        return
    m = re.fullmatch(r"(\w+)\.back\(\)\.(begin|end) = (\w+)\.sequence\(\)\.begin\(\)(?: \+ (.+))?", st)
    if m and m.group(1) in env.mems:
        env.mems[m.group(1)][-1][m.group(2)] = eval_int(m.group(4), env) if m.group(4) else 0
        return
    m = re.fullmatch(r"(\w+)\.back\(\)\.(begin|end) = (\w+)\.sequence\(\)\.end\(\)(?: - (.+))?", st)
    if m and m.group(1) in env.mems:
        L = len(env.aln_seq[m.group(3)])
        env.mems[m.group(1)][-1][m.group(2)] = L - (eval_int(m.group(4), env) if m.group(4) else 0)
        return
    m = re.fullmatch(r"(\w+)\.back\(\)\.nodes\.push_back\(gcsa::Node::encode\((.+), (.+), (true|false)\)\)", st)
    if m and m.group(1) in env.mems:
        env.mems[m.group(1)][-1]["nodes"].append([eval_int(m.group(2), env), eval_int(m.group(3), env), m.group(4) == "true"])
        return
    # aligner calls
    m = re.fullmatch(r"(\w+)(?:\.|->)(align\w*)\((.*)\)", st)
    if m and m.group(1) in env.aligners:
        args = [a.strip() for a in m.group(3).split(",")]
        aln = args[0]
        qual = env.aln_qual.get(aln)
        call = {"aligner": m.group(1), "call": m.group(2), "aln": aln, "raw_args": args[1:],
                "read": env.aln_seq.get(aln), "quality": qual, "args": [resolve_arg(a, env) for a in args[1:]],
                "expect": [], "nodes": [list(n) for n in env.nodes], "edges": [list(e) for e in env.edges]}
        env.calls.append(call)
        env.expects[aln] = call["expect"]      # REQUIREs that follow describe this call until the next one on the same object
        return
    m = re.fullmatch(r"is_(\w+)_opt = is_\1_opt && \((.+)\)", st)
    if m:      # one condition of "some returned alignment is this one" (src/unittest/banded_global_aligner.cpp:1811-1862)
        saved = dict(env.path_alias); env.path_alias["path"] = "__alt__"; env.aln_seq.setdefault("__alt__", "")
        try:
            r = parse_require(m.group(2), env)
        finally:
            env.path_alias = saved
        if r is not None:
            env.alternatives.setdefault(m.group(1), []).append(r[1])
        return
    m = re.fullmatch(r"REQUIRE\(alt_aln\.score\(\) == (.+)\)", st)
    if m:
        env.alt_scores.append(eval_int(m.group(1), env))
        return
    m = re.fullmatch(r"REQUIRE\((.+)\)", st)
    if m:
        r = parse_require(m.group(1), env)
        if r is None:
            env.unparsed.append(st)
        else:
            env.expects.setdefault(r[0], []).append(r[1])
        return


def resolve_arg(a, env):
    if a in env.bools:
        return env.bools[a]
    if a in ("true", "false"):
        return a == "true"
    if a in env.mems:
        return {"mems": env.mems[a]}
    try:
        return eval_int(a, env)
    except (ValueError, KeyError):
        return a


def cases_from_file(fname):
    raw = open(os.path.join(REF, fname)).read()
    text = strip_comments(raw)
    cases = []
    for title_path, line, prog in leaf_programs(text):
        env = run_program(prog)
        for call in env.calls:
            aln = call["aln"]
            if call["read"] is None:
                continue
            src, qual_adj = env.aligners[call["aligner"]]
            qual = call["quality"]
            case = {
                "source": "src/unittest/%s:%d" % (fname, line),
                "name": " / ".join(title_path),
                "nodes": call["nodes"],
                "edges": call["edges"],
                "read": call["read"],
                "quality": (qual[1] if qual and qual[0] == "raw" else ([ord(c) for c in qual[1]] if qual else None)),
                "scores": env.sources.get(src, DEFAULT_SCORES),
                "qual_adj": qual_adj,
                "call": call["call"],
                "args": call["args"],
                "aln": aln,
                "expect": call["expect"],
                "alternatives": env.alternatives if call["call"].endswith("_multi") else {},
                "alt_scores": env.alt_scores if call["call"].endswith("_multi") else [],
                "unparsed": env.unparsed,
            }
            cases.append(case)
    return cases


def main():
    if not os.path.isdir(REF):
        sys.exit("reference tree not present; golden JSON is already committed")
    total = 0
    for fname, out in [("aligner.cpp", "ref_aligner.json"),
                       ("pinned_alignment.cpp", "ref_pinned_alignment.json"),
                       ("xdrop_aligner.cpp", "ref_xdrop_aligner.json"),
                       ("banded_global_aligner.cpp", "ref_banded_global_aligner.json")]:
        cases = cases_from_file(fname)
        nexp = sum(len(c["expect"]) for c in cases)
        nun = sum(len(c["unparsed"]) for c in cases)
        with open(os.path.join(OUT, out), "w") as f:
            json.dump(cases, f, indent=1)
        print("%-28s %3d cases, %4d expectations, %3d REQUIREs not transcribed" % (fname, len(cases), nexp, nun))
        total += len(cases)
    print("total", total)


if __name__ == "__main__":
    main()

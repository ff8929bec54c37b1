"""hostwave (TEST INFRASTRUCTURE): rewrites the two things of the kernel sources that no macro can reach, line for line (line numbers
stay: they are the emulator's rendezvous sites and what its diagnostics print):

  * gfx950 inline assembly -> the same single operation as a C++ expression (hw::v_pk<>, hw::update_dpp, ...); an instruction this
    file does not know is an error, never a guess;
  * `__shared__` declarations -> references into the emulated workgroup's LDS arena (the dynamic part ends at a guard page).

Everything else of the sources -- every index, shift, table, cross-lane pattern, the float arithmetic -- is compiled as it stands.
"""
import re
import sys


def _split_top(s, sep):
    """split s at `sep` characters that are outside parentheses and string literals"""
    out, depth, cur, in_str, i = [], 0, [], False, 0
    while i < len(s):
        c = s[i]
        if in_str:
            cur.append(c)
            if c == "\\":
                cur.append(s[i + 1]); i += 1
            elif c == '"':
                in_str = False
        elif c == '"':
            in_str = True; cur.append(c)
        elif c in "([":
            depth += 1; cur.append(c)
        elif c in ")]":
            depth -= 1; cur.append(c)
        elif c == sep and depth == 0:
            out.append("".join(cur)); cur = []
        else:
            cur.append(c)
        i += 1
    out.append("".join(cur))
    return out


def _operands(part):
    """'"=v"(a), "v"(b >> 2)' -> [("=v", "a"), ("v", "b >> 2")]"""
    res = []
    for piece in _split_top(part, ","):
        piece = piece.strip()
        if not piece:
            continue
        m = re.match(r'"([^"]*)"\s*\((.*)\)\s*$', piece, re.S)
        assert m, "asm operand %r" % piece
        res.append((m.group(1), m.group(2).strip()))
    return res


def _bits(mods, name, default):
    m = re.search(name + r":\[(\d),(\d)\]", mods)
    return (int(m.group(1)) | int(m.group(2)) << 1) if m else default


DPP_CTRL = {"quad_perm:[1,0,3,2]": 0xB1, "quad_perm:[2,3,0,1]": 0x4E, "row_ror:8": 0x128}


def asm_to_cpp(template, outs, ins, where):
    """one asm statement -> C++ statement(s)"""
    t = " ".join(template.replace("\\n", " ").replace("\\t", " ").split())
    if t == "" or t.startswith(";"):
        return "((void)0);"                                            # scheduling fences and profile markers
    mnem = t.split()[0]
    ops = [o for o in outs]
    o = lambda k: outs[k][1]
    i = lambda k: ins[k][1]
    if mnem == "v_mad_i32_i24":
        return "%s = hw::v_mad_i32_i24(%s, %s, %s);" % (o(0), i(0), i(1), i(2))
    if mnem in ("v_pk_mul_f32", "v_pk_add_f32"):
        args = re.match(r"\S+ %0, %(\d), %(\d)(.*)$", t)
        assert args, where
        a, b, mods = ins[int(args.group(1)) - 1][1], ins[int(args.group(2)) - 1][1], args.group(3)
        return "%s = hw::v_pk<'%s'>(%s, %s, %d, %d, %d, %d);" % (o(0), "*" if "mul" in mnem else "+", a, b, _bits(mods, "op_sel", 0), _bits(mods, "op_sel_hi", 3),
                                                                _bits(mods, "neg_lo", 0), _bits(mods, "neg_hi", 0))
    if mnem == "v_mul_f32_dpp":
        m = re.match(r"v_mul_f32_dpp %0, %1, (-?)%2 (quad_perm:\[[\d,]+\]|row_ror:\d+) row_mask:0xf bank_mask:0xf bound_ctrl:0$", t)
        assert m and m.group(2) in DPP_CTRL, where + ": " + t
        return "%s = __int_as_float((int)hw::update_dpp(0, (uint32_t)__float_as_int(%s), 0x%X, 0xF, 0xF, true, __LINE__)) * (%s(%s));" % (
            o(0), i(0), DPP_CTRL[m.group(2)], m.group(1), i(1))
    if mnem == "v_cvt_f32_i32_sdwa":
        m = re.search(r"sext\(%1\).*src0_sel:(BYTE|WORD)_(\d)", t)
        assert m, where
        if m.group(1) == "BYTE":
            return "%s = (float)(int8_t)((%s) >> %d);" % (o(0), i(0), 8 * int(m.group(2)))
        return "%s = (float)(int16_t)((%s) >> %d);" % (o(0), i(0), 16 * int(m.group(2)))
    if mnem == "v_bcnt_u32_b32":
        return "%s = (uint32_t)__builtin_popcount(%s) + (%s);" % (o(0), i(0), i(1))
    if mnem == "ds_add_u32":
        assert "s_waitcnt" in t and not outs, where
        return "hw::ds_add_u32(%s, %s);" % (i(0), i(1))
    raise SystemExit("%s: inline assembly not modelled: %s" % (where, t))


ASM_RE = re.compile(r"\basm\s*(?:volatile\s*)?\(")
SHARED_DYN = re.compile(r"extern\s+__shared__\s+(?:__attribute__\(\(aligned\(\d+\)\)\)\s+)?([A-Za-z_][\w:]*)\s+(\w+)\s*\[\s*\]\s*;")
SHARED_STATIC = re.compile(r"(?<!extern\s)__shared__\s+(?:__attribute__\(\(aligned\((\d+)\)\)\)\s+)?([A-Za-z_][\w:]*\s*\**)\s*(\w+)\s*((?:\[[^\]]+\])+)\s*;")


def translate_line(line, where):
    # LDS declarations
    line = SHARED_DYN.sub(lambda m: "%s* const %s = (%s*)hw::dyn_lds();" % (m.group(1), m.group(2), m.group(1)), line)

    def static(m):
        align, ty, name, dims = m.group(1) or "4", m.group(2).strip(), m.group(3), m.group(4)
        return "using %s_lds_t = %s %s; %s_lds_t& %s = *(%s_lds_t*)hw::static_lds(sizeof(%s_lds_t), %s, __LINE__ * 8 + %d);" % (
            name, ty, dims, name, name, name, name, align, static.count)
    static.count = 0
    while True:
        new = SHARED_STATIC.sub(static, line, count=1)
        if new == line:
            break
        static.count += 1
        line = new
    assert "__shared__" not in line[:_comment_start(line)], "%s: __shared__ form not handled: %s" % (where, line.strip())
    # inline assembly (one statement per occurrence, all on this line)
    code_end = _comment_start(line)
    out, pos = [], 0
    while True:
        m = ASM_RE.search(line, pos)
        if not m or m.start() >= code_end:
            out.append(line[pos:])
            break
        depth, j = 1, m.end()
        in_str = False
        while depth:
            c = line[j]
            if in_str:
                if c == "\\":
                    j += 1
                elif c == '"':
                    in_str = False
            elif c == '"':
                in_str = True
            elif c == "(":
                depth += 1
            elif c == ")":
                depth -= 1
            j += 1
        body = line[m.end():j - 1]
        k = j
        while k < len(line) and line[k] in " \t":
            k += 1
        has_semi = k < len(line) and line[k] == ";"
        parts = _split_top(body, ":")
        tm = re.match(r'\s*((?:"(?:[^"\\]|\\.)*"\s*)+)$', parts[0], re.S)
        if not tm and line.lstrip().startswith("#define"):             # a marker macro (stringised argument): nothing to execute
            cpp = "((void)0);"
        else:
            assert tm, "%s: asm template %r" % (where, parts[0])
            template = "".join(re.findall(r'"((?:[^"\\]|\\.)*)"', tm.group(1)))
            outs = _operands(parts[1]) if len(parts) > 1 else []
            ins = _operands(parts[2]) if len(parts) > 2 else []
            cpp = asm_to_cpp(template, outs, ins, where)
        if not has_semi:
            cpp = cpp.rstrip(";")
        out.append(line[pos:m.start()] + cpp)
        pos = k + 1 if has_semi else j
    return "".join(out)


def _comment_start(line):
    in_str = False
    i = 0
    while i < len(line) - 1:
        c = line[i]
        if in_str:
            if c == "\\":
                i += 1
            elif c == '"':
                in_str = False
        elif c == '"':
            in_str = True
        elif c == "/" and line[i + 1] == "/":
            return i
        i += 1
    return len(line)


def translate(text, name):
    lines = text.split("\n")
    return "\n".join(translate_line(l, "%s:%d" % (name, n + 1)) for n, l in enumerate(lines))


if __name__ == "__main__":
    src, dst = sys.argv[1:3]
    with open(src) as f:
        t = translate(f.read(), src)
    with open(dst, "w") as f:
        f.write(t)

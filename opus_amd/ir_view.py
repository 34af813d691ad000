"""Backend hand-off view: what a host backend reads from an ``IrGenerator``, served from the flat
arrays of the C ABI (``opusfe_copy_array``) alone.

The reference's backend (backend/mod.rs:6-10 ``Backend::translate_ir(&mut self, ir: &IrGenerator)``; llvm.rs:355-420) walks
``ir.function``, ``ir.variables[i].typ``, ``ir.bbs[b].instructions`` and calls ``ir.get_lvalue_type(v)``.  ``IrView`` gives the
same accessors over the GPU-produced arrays and re-renders ``impl Display for IrGenerator`` (generate_ir.rs:1201-1283) from
them — the parity tests compare that text with the oracle's, which proves the arrays carry everything the text does.
Nothing here computes the front end: the arrays come from the HIP path and this module only reads them.
"""
from decimal import Decimal

import numpy as np

SEM_CONTINUATION = 0x100
(A_SOURCE, A_TOKENS, A_VAL64, A_STRINGS, A_NODES, A_DEFINITIONS, A_SIGNATURES, A_GLOBALS, A_TYPES, A_NODE_TYPES, A_FUNCTIONS,
 A_VARIABLES, A_BB_OFFSETS, A_INSTRS, A_CALL_ARGS, A_SEM_ERRORS, A_PARENTS) = range(17)

u8, u16, u32, u64 = np.uint8, np.uint16, np.uint32, np.uint64
DTYPES = {
    A_SOURCE: u8, A_VAL64: u64, A_STRINGS: u8, A_DEFINITIONS: u32, A_NODE_TYPES: u32, A_BB_OFFSETS: u32, A_CALL_ARGS: u32, A_PARENTS: u32,
    A_TOKENS: np.dtype([("kind", u8), ("flags", u8), ("aux", u16), ("low", u32), ("len", u32), ("payload", u32)]),
    A_NODES: np.dtype([("kind", u8), ("sub", u8), ("nchild", u16), ("first_child", u32), ("next", u32), ("payload", u32),
                       ("low", u32), ("high", u32)]),
    A_SIGNATURES: np.dtype([(n, u32) for n in ("def_node", "name_tok", "nargs", "ret_type", "first_arg", "flags")]),
    A_GLOBALS: np.dtype([(n, u32) for n in ("def_node", "type_id", "is_const", "name_tok")]),
    A_TYPES: np.dtype([("kind", u8), ("ptr", u8), ("src", u8), ("_pad", u8), ("sub", u32), ("ref", u32), ("alias", u32)]),
    A_FUNCTIONS: np.dtype([(n, u32) for n in ("def_node", "func", "n_vars", "n_instrs", "n_bbs", "n_types", "n_call_args",
                                               "n_errors", "var_base", "instr_base", "bb_base", "type_base", "call_arg_base",
                                               "error_base", "status", "_pad")]),
    A_VARIABLES: np.dtype([("type_id", u32), ("is_temporary", u32)]),
    A_INSTRS: np.dtype([("opcode", u16), ("sub", u16), ("dst", u32), ("a", u32), ("b", u32), ("c", u32), ("extra", u32)]),
    A_SEM_ERRORS: np.dtype([("kind", u32), ("low", u32), ("len", u32), ("name", u32), ("nargs", u32), ("t", u32, (4,))]),
}
NONE = 0xFFFFFFFF

# compile.rs:45-75 Type variant order; TypePrinter names (compile.rs:200-274)
TYPE_NAMES = ["int8", "int16", "int32", "int64", "nat8", "nat16", "nat32", "nat64", "float32", "float64", "offset", "size",
              "<integer>", "<unknown>", "bool", "null", None, "string", None, None, None, "<error>"]
TY_POINTER, TY_RECORD, TY_FUNCTION, TY_ARRAY = 16, 18, 19, 20
POINTER_PREFIX = ["@", "@mut ", "{} ", "mut {} "]
# generate_ir.rs:8-56 Instruction variant order
(I_INTEGER, I_FLOAT, I_NULL, I_BOOL, I_STRING, I_SIZEOF, I_ALIGNOF, I_CALL, I_ALLOCATE, I_LOAD, I_STORE, I_FIELD, I_BUILTINFIELD,
 I_INDEXPOINTER, I_INDEXARRAY, I_ADD, I_SUBTRACT, I_MULTIPLY, I_DIVIDE, I_MODULO, I_EQUALS, I_LESSTHAN, I_GREATERTHAN,
 I_LESSTHANEQUALS, I_GREATERTHANEQUALS, I_FUNCTION, I_GLOBALVARIABLE, I_GLOBALCONSTANT, I_NOT, I_NEGATE, I_CAST, I_RETURN, I_JUMP,
 I_BRANCH, I_UNREACHABLE, I_PHI, I_BREAKPLACEHOLDER, I_ERROR) = range(38)
BINARY = ["add", "subtract", "multiply", "divide", "modulo", "equals", "lessthan", "greaterthan", "lessthanequals",
          "greaterthanequals"]
CASTS = ["None", "Pointer", "PointerType", "Integer", "PointerToInteger", "IntegerToPointer", "PointerToArray", "Error"]
T_EQUALSEQUALS, T_IDENTIFIER = 8, 26
TF_INLINE = 1


def rust_f64(d):
    """Rust `{}` of an f64: shortest digits that round-trip, never scientific."""
    if d != d:
        return "NaN"
    if d in (float("inf"), float("-inf")):
        return "inf" if d > 0 else "-inf"
    s = format(Decimal(repr(d)), "f")
    if "." in s:
        s = s.rstrip("0").rstrip(".")
    return s


def rust_debug_str(b):
    """<str as Debug> of String::from_utf8_lossy(b) (generate_ir.rs:1216)."""
    out = ['"']
    for ch in b.decode("utf-8", errors="replace"):
        c = ord(ch)
        if ch == "\0":
            out.append("\\0")
        elif ch == "\t":
            out.append("\\t")
        elif ch == "\r":
            out.append("\\r")
        elif ch == "\n":
            out.append("\\n")
        elif ch == "\\":
            out.append("\\\\")
        elif ch == '"':
            out.append('\\"')
        elif 0x20 <= c < 0x7F or c >= 0xA0:
            out.append(ch)
        else:
            out.append("\\u{%x}" % c)
    out.append('"')
    return "".join(out)


class FunctionView:
    """One ``IrGenerator`` (generate_ir.rs:58-98): `function`, `variables`, `bbs`."""

    def __init__(self, view, index):
        self.view, self.index = view, index
        self.rec = view.functions[index]

    @property
    def is_extern(self):
        return int(self.rec["n_bbs"]) == 0

    @property
    def variables(self):
        b = int(self.rec["var_base"])
        return self.view.variables[b:b + int(self.rec["n_vars"])]

    @property
    def bbs(self):
        """list of instruction slices, one per basic block (generate_ir.rs:70 `bbs`)"""
        v, b0 = self.view, int(self.rec["bb_base"])
        off = v.bb_offsets
        return [v.instrs[int(off[b0 + k]):int(off[b0 + k + 1])] for k in range(int(self.rec["n_bbs"]))]

    def call_args(self, instr):
        c = int(instr["c"])
        return self.view.call_args[c:c + int(instr["sub"])]

    def get_lvalue_type(self, var):
        """generate_ir.rs get_lvalue_type: the pointee of the variable's pointer type (type id)."""
        t = self.view.types[int(self.variables[var]["type_id"])]
        if int(t["kind"]) != TY_POINTER:
            raise ValueError("variable is not an lvalue")
        return int(t["sub"])

    def text(self):
        v = self.view
        out = [v.function_text(int(self.rec["func"])), "\n"]
        vs = self.variables
        for i in range(len(vs)):
            out.append(f"VAR %{i}: {v.type_text(int(vs[i]['type_id']))}\n")
        for b, instrs in enumerate(self.bbs):
            if b:
                out.append("\n")
            out.append(f"@{b}")
            for ins in instrs:
                out.append("\n  ")
                out.append(self.instr_text(ins))
        out.append("\n")
        return "".join(out)

    def instr_text(self, ins):
        v = self.view
        op, sub, dst, a, b, c, extra = (int(ins[k]) for k in ("opcode", "sub", "dst", "a", "b", "c", "extra"))
        imm = c | (extra << 32)
        if op == I_INTEGER:
            return f"%{dst} = {imm - (1 << 64) if imm >> 63 else imm}"
        if op == I_FLOAT:
            return f"%{dst} = {rust_f64(float(np.array([imm], dtype=u64).view(np.float64)[0]))}"
        if op == I_NULL:
            return f"%{dst} = null"
        if op == I_BOOL:
            return f"%{dst} = {'true' if sub else 'false'}"
        if op == I_STRING:
            return f"%{dst} = {rust_debug_str(v.token_string(c))}"
        if op in (I_SIZEOF, I_ALIGNOF):
            return f"%{dst} = {'sizeof' if op == I_SIZEOF else 'alignof'} {v.type_text(c)}"
        if op == I_CALL:
            return f"%{dst} = call %{a}" + "".join(f", {int(x)}" for x in self.call_args(ins))
        if op == I_ALLOCATE:
            return f"%{dst} = alloc"
        if op == I_LOAD:
            return f"%{dst} = load %{a}"
        if op == I_STORE:
            return f"in %{a} store %{b}"
        if op == I_FIELD:
            return f"%{dst} = fieldptr %{a}, {v.token_name(c)}"
        if op == I_BUILTINFIELD:
            return f"%{dst} = builtinfieldptr %{a}, {c}"
        if op == I_INDEXPOINTER:
            return f"%{dst} = indexptr %{a}, %{b}"
        if op == I_INDEXARRAY:
            return f"%{dst} = indexarray %{a}, %{b}"
        if I_ADD <= op <= I_GREATERTHANEQUALS:
            return f"%{dst} = {BINARY[op - I_ADD]} %{a}, %{b}"
        if op == I_FUNCTION:
            return f"%{dst} = global ${v.function_text(c)}"
        if op == I_GLOBALVARIABLE:
            return f"%{dst} = globalvar %{c}"
        if op == I_GLOBALCONSTANT:
            return f"%{dst} = globalconst %{c}"
        if op == I_NOT:
            return f"%{dst} = not %{a}"
        if op == I_NEGATE:
            return f"%{dst} = negate %{a}"
        if op == I_CAST:
            vs = self.variables
            return (f"%{dst} = cast {CASTS[sub]} [{v.type_text(int(vs[a]['type_id']))} to "
                    f"{v.type_text(int(vs[dst]['type_id']))}] %{a}")
        if op == I_RETURN:
            return f"return %{a}"
        if op == I_JUMP:
            return f"jump @{c}"
        if op == I_BRANCH:
            return f"branch %{a}, @{b}, @{c}"
        if op == I_UNREACHABLE:
            return "unreachable"
        if op == I_PHI:
            return f"%{dst} = phi @{c} => %{a}, @{extra} => %{b}"
        if op == I_BREAKPLACEHOLDER:
            return "<break placeholder>"
        if op == I_ERROR:
            return f"%{dst} = error"
        return "?"


class IrView:
    """All arrays of a lowered unit, copied to the host once."""

    def __init__(self, unit):
        get = unit.array
        self.source = get(A_SOURCE).tobytes()
        self.tokens, self.val64, self.strings = get(A_TOKENS), get(A_VAL64), get(A_STRINGS).tobytes()
        self.nodes, self.definitions = get(A_NODES), get(A_DEFINITIONS)
        self.signatures, self.globals, self.types, self.node_types = get(A_SIGNATURES), get(A_GLOBALS), get(A_TYPES), get(A_NODE_TYPES)
        self.functions, self.variables, self.bb_offsets = get(A_FUNCTIONS), get(A_VARIABLES), get(A_BB_OFFSETS)
        self.instrs, self.call_args = get(A_INSTRS), get(A_CALL_ARGS)
        # OPUSFE_A_SEM_ERRORS: the diagnostics in report order; a function identifier with more than four argument types
        # (compile.rs:115-122) is followed by continuation records (kind 0x100: eight type ids behind `kind`)
        raw = get(A_SEM_ERRORS)
        keep = raw["kind"] != SEM_CONTINUATION
        self.sem_errors = raw[keep]
        self.sem_error_arg_types = []          # per diagnostic: every argument type id of its function identifier
        words = raw.view(np.uint32).reshape(len(raw), 9) if len(raw) else np.zeros((0, 9), np.uint32)
        for k in range(len(raw)):
            if keep[k]:
                n = int(raw[k]["nargs"])
                self.sem_error_arg_types.append([int(x) for x in raw[k]["t"][:min(n, 4)]])
            elif self.sem_error_arg_types:
                self.sem_error_arg_types[-1] += [int(x) for x in words[k][1:9] if int(x) != 0xFFFFFFFF]

    def __len__(self):
        return len(self.functions)

    def function(self, i):
        return FunctionView(self, i)

    # ---- names and literals ---------------------------------------------------------------------------------------
    def token_name(self, ti):
        """identifier text, or parse.rs:224-243 Operator::symbol for the token after `operator`"""
        t = self.tokens[ti]
        if int(t["kind"]) == T_EQUALSEQUALS:
            return "="
        low = int(t["low"])
        s = self.source[low:low + int(t["len"])].decode("utf-8", errors="replace")
        return s if int(t["kind"]) == T_IDENTIFIER else s.strip()

    def token_u64(self, ti):
        t = self.tokens[ti]
        return int(t["payload"]) if int(t["flags"]) & TF_INLINE else int(self.val64[int(t["payload"])])

    def token_string(self, ti):
        p = int(self.tokens[ti]["payload"])
        n = int.from_bytes(self.strings[p:p + 4], "little")
        return self.strings[p + 4:p + 4 + n]

    def children(self, node):
        c = int(self.nodes[node]["first_child"])
        for _ in range(int(self.nodes[node]["nchild"])):
            yield c
            c = int(self.nodes[c]["next"])

    # ---- TypePrinter / FunctionPrinter (compile.rs:200-298) ---------------------------------------------------------
    def type_text(self, t):
        prefix = []          # pointer / array chains are walked in a loop (a depth-1000 `@@@...` type is a test case)
        while True:
            r = self.types[t]
            kind = int(r["kind"])
            if int(r["alias"]) != NONE or kind not in (TY_POINTER, TY_ARRAY):
                break
            prefix.append(POINTER_PREFIX[int(r["ptr"])] if kind == TY_POINTER else "{%d} " % self.token_u64(int(r["ref"])))
            t = int(r["sub"])
        return "".join(prefix) + self.type_text_head(t)

    def type_text_head(self, t):
        r = self.types[t]
        if int(r["alias"]) != NONE:
            return self.token_name(int(r["alias"]))
        kind = int(r["kind"])
        if TYPE_NAMES[kind] is not None:
            return TYPE_NAMES[kind]
        if kind == TY_POINTER:
            return POINTER_PREFIX[int(r["ptr"])] + self.type_text(int(r["sub"]))
        if kind == TY_RECORD:
            fields = "".join(f" {self.token_name(int(self.nodes[c]['payload']))}: "
                             f"{self.type_text(int(self.node_types[int(self.nodes[c]['first_child'])]))}"
                             for c in self.children(int(r["ref"])))
            return "record {" + fields + " }"
        if kind == TY_FUNCTION:
            if int(r["src"]) == 2:
                args = self.signature_arg_types(int(r["ref"]))
            else:
                kids = list(self.children(int(r["ref"])))
                args = [int(self.node_types[c]) for c in kids[:-1]]
            return "[" + " ".join(self.type_text(a) for a in args) + "] " + self.type_text(int(r["sub"]))
        if kind == TY_ARRAY:
            return "{%d} %s" % (self.token_u64(int(r["ref"])), self.type_text(int(r["sub"])))
        return "?"

    def signature_arg_types(self, f):
        s = self.signatures[f]
        out, c = [], int(s["first_arg"])
        for _ in range(int(s["nargs"])):
            out.append(int(self.node_types[int(self.nodes[c]["first_child"])]))
            c = int(self.nodes[c]["next"])
        return out

    def function_text(self, f):
        s = self.signatures[f]
        args = " ".join(":" + self.type_text(a) for a in self.signature_arg_types(f))
        return f"{self.token_name(int(s['name_tok']))}[{args}]: {self.type_text(int(s['ret_type']))}"

    def text(self):
        """the text `opus -Z` prints for the unit (frontend.rs:165-169), rendered from the arrays only"""
        return "".join(self.function(i).text() for i in range(len(self)) if not self.function(i).is_extern)

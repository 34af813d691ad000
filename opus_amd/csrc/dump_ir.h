// Host-side IR text: exactly `impl Display for IrGenerator` (generate_ir.rs:1201-1283) per function, each followed
// by '\n' (the eprintln! of frontend.rs:168), with TypePrinter / FunctionPrinter (compile.rs:200-298).
#pragma once
#include "dump_ast.h"
#include "lower.h"

namespace opusfe {

struct IrText {
    const opusfe_unit* u;
    const IrResult* R;
    HostView v;
    const opusfe_node& n(uint32_t i) const { return u->h_nodes[i]; }
    // TypePrinter (compile.rs:200-274)
    void type(std::string& out, uint32_t t) const {
        const TypeRec& r = R->types[t];
        if (r.alias != OPUSFE_NONE) { out += name_of_token(v, r.alias); return; }
        switch (r.kind) {
            case TY_Integer8: out += "int8"; return; case TY_Integer16: out += "int16"; return; case TY_Integer32: out += "int32"; return;
            case TY_Integer64: out += "int64"; return; case TY_Natural8: out += "nat8"; return; case TY_Natural16: out += "nat16"; return;
            case TY_Natural32: out += "nat32"; return; case TY_Natural64: out += "nat64"; return; case TY_Size: out += "size"; return;
            case TY_Offset: out += "offset"; return; case TY_Float32: out += "float32"; return; case TY_Float64: out += "float64"; return;
            case TY_GenericInteger: out += "<integer>"; return; case TY_Generic: out += "<unknown>"; return; case TY_Bool: out += "bool"; return;
            case TY_Null: out += "null"; return; case TY_String: out += "string"; return; case TY_Error: out += "<error>"; return;
            case TY_Pointer:
                out += r.ptr == PT_Reference ? "@" : r.ptr == PT_Mutable ? "@mut " : r.ptr == PT_Array ? "{} " : "mut {} ";
                type(out, r.sub);
                return;
            case TY_Record: {
                out += "record {";
                uint32_t c = n(r.ref).first_child;
                for (uint32_t k = 0; k < n(r.ref).nchild; k++) {
                    out += ' '; out += name_of_token(v, n(c).payload); out += ": "; type(out, R->tid_of_node[n(c).first_child]);
                    c = n(c).next;
                }
                out += " }";
                return;
            }
            case TY_Function: {
                out += '[';
                if (r.src == TS_FUNC) {
                    const FuncRec& f = R->funcrecs[r.ref];
                    uint32_t c = f.first_arg;
                    for (uint32_t k = 0; k < f.nargs; k++) { if (k) out += ' '; type(out, R->tid_of_node[n(c).first_child]); c = n(c).next; }
                } else {
                    uint32_t c = n(r.ref).first_child;
                    for (uint32_t k = 0; k + 1 < n(r.ref).nchild; k++) { if (k) out += ' '; type(out, R->tid_of_node[c]); c = n(c).next; }
                }
                out += "] "; type(out, r.sub);
                return;
            }
            case TY_Array: {
                out += '{'; out += std::to_string(v.tok_u64(v.tok(r.ref))); out += "} "; type(out, r.sub);
                return;
            }
            default: out += "?"; return;
        }
    }
    // FunctionPrinter (compile.rs:276-298)
    void function(std::string& out, uint32_t f) const {
        const FuncRec& fr = R->funcrecs[f];
        out += name_of_token(v, fr.name_tok); out += '[';
        uint32_t c = fr.first_arg;
        for (uint32_t k = 0; k < fr.nargs; k++) { if (k) out += ' '; out += ':'; type(out, R->tid_of_node[n(c).first_child]); c = n(c).next; }
        out += "]: "; type(out, fr.ret);
    }
    static void var(std::string& out, uint32_t x) { out += '%'; out += std::to_string(x); }
    uint32_t glob_off = 0, f0 = 0, f1 = 0xFFFFFFFFu;      // (DumpView: one file of a packed corpus)
    void text(std::string& out) const {
        static const char* const OPS[] = {"add", "subtract", "multiply", "divide", "modulo", "equals", "lessthan", "greaterthan", "lessthanequals",
                                          "greaterthanequals"};
        static const char* const CASTS[] = {"None", "Pointer", "PointerType", "Integer", "PointerToInteger", "IntegerToPointer", "PointerToArray", "Error"};
        for (size_t fi = f0; fi < R->funcs.size() && fi < f1; fi++) {
            const FuncIr& f = R->funcs[fi];
            if (f.n_bbs == 0) continue;   // extern
            function(out, f.func); out += '\n';
            for (uint32_t i = 0; i < f.n_vars; i++) { out += "VAR %"; out += std::to_string(i); out += ": "; type(out, R->vars[f.var_base + i].type_id); out += '\n'; }
            for (uint32_t b = 0; b < f.n_bbs; b++) {
                if (b) out += '\n';
                out += '@'; out += std::to_string(b);
                for (uint32_t k = R->bb_offsets[f.bb_base + b]; k < R->bb_offsets[f.bb_base + b + 1]; k++) {
                    const opusfe_instr& in = R->instrs[k];
                    out += "\n  ";
                    switch (in.opcode) {
                        case I_Integer: var(out, in.dst); out += " = "; out += std::to_string((int64_t)((uint64_t)in.c | ((uint64_t)in.extra << 32))); break;
                        case I_Float: { uint64_t bits = (uint64_t)in.c | ((uint64_t)in.extra << 32); double d; memcpy(&d, &bits, 8); var(out, in.dst); out += " = "; out += fmt_f64_display(d); break; }
                        case I_Null: var(out, in.dst); out += " = null"; break;
                        case I_Bool: var(out, in.dst); out += in.sub ? " = true" : " = false"; break;
                        case I_String: { uint32_t len; const uint8_t* p = v.tok_str(v.tok(in.c), &len); var(out, in.dst); out += " = "; fmt_debug_str(out, p, len); break; }
                        case I_Sizeof: var(out, in.dst); out += " = sizeof "; type(out, in.c); break;
                        case I_Alignof: var(out, in.dst); out += " = alignof "; type(out, in.c); break;
                        case I_Call:
                            var(out, in.dst); out += " = call "; var(out, in.a);
                            for (uint32_t a = 0; a < in.sub; a++) { out += ", "; out += std::to_string(R->cargs[in.c + a]); }
                            break;
                        case I_Allocate: var(out, in.dst); out += " = alloc"; break;
                        case I_Load: var(out, in.dst); out += " = load "; var(out, in.a); break;
                        case I_Store: out += "in "; var(out, in.a); out += " store "; var(out, in.b); break;
                        case I_Field: var(out, in.dst); out += " = fieldptr "; var(out, in.a); out += ", "; out += name_of_token(v, in.c); break;
                        case I_BuiltinField: var(out, in.dst); out += " = builtinfieldptr "; var(out, in.a); out += ", "; out += std::to_string(in.c); break;
                        case I_IndexPointer: var(out, in.dst); out += " = indexptr "; var(out, in.a); out += ", "; var(out, in.b); break;
                        case I_IndexArray: var(out, in.dst); out += " = indexarray "; var(out, in.a); out += ", "; var(out, in.b); break;
                        case I_Add: case I_Subtract: case I_Multiply: case I_Divide: case I_Modulo: case I_Equals: case I_LessThan: case I_GreaterThan:
                        case I_LessThanEquals: case I_GreaterThanEquals:
                            var(out, in.dst); out += " = "; out += OPS[in.opcode - I_Add]; out += ' '; var(out, in.a); out += ", "; var(out, in.b); break;
                        case I_Function: var(out, in.dst); out += " = global $"; function(out, in.c); break;
                        case I_GlobalVariable: var(out, in.dst); out += " = globalvar %"; out += std::to_string(in.c - glob_off); break;
                        case I_GlobalConstant: var(out, in.dst); out += " = globalconst %"; out += std::to_string(in.c - glob_off); break;
                        case I_Not: var(out, in.dst); out += " = not "; var(out, in.a); break;
                        case I_Negate: var(out, in.dst); out += " = negate "; var(out, in.a); break;
                        case I_Cast:
                            var(out, in.dst); out += " = cast "; out += CASTS[in.sub]; out += " ["; type(out, R->vars[f.var_base + in.a].type_id); out += " to ";
                            type(out, R->vars[f.var_base + in.dst].type_id); out += "] "; var(out, in.a);
                            break;
                        case I_Return: out += "return "; var(out, in.a); break;
                        case I_Jump: out += "jump @"; out += std::to_string(in.c); break;
                        case I_Branch: out += "branch "; var(out, in.a); out += ", @"; out += std::to_string(in.b); out += ", @"; out += std::to_string(in.c); break;
                        case I_Unreachable: out += "unreachable"; break;
                        case I_Phi:
                            var(out, in.dst); out += " = phi @"; out += std::to_string(in.c); out += " => "; var(out, in.a); out += ", @"; out += std::to_string(in.extra);
                            out += " => "; var(out, in.b);
                            break;
                        case I_BreakPlaceholder: out += "<break placeholder>"; break;
                        case I_Error: var(out, in.dst); out += " = error"; break;
                        default: out += "?"; break;
                    }
                }
            }
            out += '\n';
        }
    }
};

inline std::string dump_ir(const opusfe_unit* u, const DumpView& w = DumpView{}) {
    const IrResult* R = (const IrResult*)u->ir;
    std::string out;
    if (!R || u->status != OPUSFE_OK) return out;
    IrText t{u, R, HostView{u, w.pos_off}};
    t.glob_off = w.glob_off; t.f0 = w.f0; t.f1 = w.f1;
    t.text(out);
    return out;
}

// "<Kind> <low> <len>\n" per semantic error raised during IR generation, functions in definition order
inline std::string dump_semantic_errors(const opusfe_unit* u, const DumpView* w = nullptr) {
    static const char* const NAMES[] = {"ParseError", "UndefinedVariable", "ShadowedName", "UndefinedFunction", "NoOverloadForFunction", "AmbiguousOverload",
        "AmbiguousFunctionReference", "UnexpectedType", "InvalidType", "InvalidOperandType", "InvalidOperandTypes", "FunctionMightNotReturn", "BreakOutsideLoop",
        "ContinueOutsideLoop", "InvalidLvalue", "ImmutableLvalue", "ArrayPointerToNonArray", "InvalidCast", "UndefinedType", "FieldAccessOnNonRecord",
        "FieldDoesNotExist", "CallOfNonFunction", "ReferenceToBuiltinFunction", "AttemptToCallOrIndexConstant", "InvalidNumberOfArgs", "InvalidConstantExpression"};
    const IrResult* R = (const IrResult*)u->ir;
    std::string out;
    if (!R) return out;
    if (w) {      // one file of a packed corpus: the diagnostics of its functions, positions relative to the file
        for (size_t fi = w->f0; fi < R->funcs.size() && fi < w->f1; fi++) {
            const FuncIr& f = R->funcs[fi];
            for (uint32_t k = 0; k < f.n_errs && (size_t)f.err_base + k < R->errs.size(); k++) {
                const SemErr& e = R->errs[f.err_base + k];
                if (e.kind == SE_Continuation) continue;
                out += NAMES[e.kind]; out += ' '; out += std::to_string(e.low - w->pos_off); out += ' '; out += std::to_string(e.len); out += '\n';
            }
        }
        return out;
    }
    for (const SemErr& e : R->errs) { if (e.kind == SE_Continuation) continue; out += NAMES[e.kind]; out += ' '; out += std::to_string(e.low); out += ' '; out += std::to_string(e.len); out += '\n'; }
    return out;
}

}  // namespace opusfe

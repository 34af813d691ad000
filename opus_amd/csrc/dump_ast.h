// Host-side canonical AST / error text (formats: SURVEY.md §8c) rendered from the flat node array.
//   per definition:  "<Rust {:?} of parse::Definition> @ <low>..<high>\n" followed by one line per
//   node that owns a span-table entry in the reference, in pre-order: "  <class> <Kind> <low> <high>\n".
#pragma once
#include "dump.h"

namespace opusfe {

#ifndef OPUSFE_EMU
inline void ensure_host_ast(opusfe_unit* u) {
    if (u->have_h_ast) return;
    opusfe_ctx* ctx = u->ctx;
    u->h_nodes.resize(u->n_nodes); u->h_def_nodes.resize(u->n_defs);
    if (u->n_nodes && u->d_nodes) HIP_TRY(hipMemcpyAsync(u->h_nodes.data(), u->d_nodes, u->n_nodes * sizeof(opusfe_node), hipMemcpyDeviceToHost, ctx->stream));
    if (u->n_defs && u->d_def_nodes) HIP_TRY(hipMemcpyAsync(u->h_def_nodes.data(), u->d_def_nodes, u->n_defs * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    u->have_h_ast = true;
}
#endif

// parse.rs:224-243 Operator::symbol for the token that follows `operator`; identifier text otherwise
inline std::string name_of_token(const HostView& v, uint32_t ti) {
    const opusfe_token& t = v.tok(ti);
    switch (t.kind) {
        case T_Identifier: return std::string((const char*)v.src() + t.low, t.len);
        case T_Plus: return "+"; case T_Minus: return "-"; case T_Asterisk: return "*"; case T_Slash: return "/"; case T_Percent: return "%";
        case T_EqualsEquals: return "="; case T_LessThan: return "<"; case T_GreaterThan: return ">"; case T_LessThanEquals: return "<=";
        case T_GreaterThanEquals: return ">="; case T_Is: return "is"; case T_Not: return "not"; case T_For: return "for";
        case T_Continue: return "continue";
        default: return "?";
    }
}

struct AstText {
    const opusfe_unit* u;
    HostView v;
    std::string dbg, spans;
    static const char* kind_name(uint8_t k) {
        static const char* const N[] = {"Function", "Extern", "Type", "Variable", "Constant", "Import", "Library",
            "Expression", "VariableDefinition", "Assignment", "Return", "If", "While", "For", "Break", "Continue",
            "Integer", "Float", "Null", "Bool", "String", "Variable", "VariableDefinition", "Proc", "CallOrIndex", "NamedCall",
            "Record", "Operator", "Reference", "MutableReference", "ArrayReference", "MutableArrayReference", "Dereference",
            "Field", "Cast", "Negate", "Sizeof", "Alignof", "Parentheses",
            "Name", "Null", "Reference", "MutableReference", "ArrayReference", "MutableArrayReference", "Proc", "Record", "Array",
            "Block", "FieldInit", "NamedType"};
        return N[k];
    }
    const opusfe_node& n(uint32_t i) const { return u->h_nodes[i]; }
    void span_line(char cls, uint32_t i) {
        spans += "  "; spans += cls; spans += ' '; spans += kind_name(n(i).kind); spans += ' ';
        spans += std::to_string(n(i).low - v.pos_off); spans += ' '; spans += std::to_string(n(i).high - v.pos_off); spans += '\n';
    }
    void dbg_name(uint32_t tok) { std::string s = name_of_token(v, tok); fmt_debug_str(dbg, (const uint8_t*)s.data(), s.size()); }
    void type(uint32_t i) {
        span_line('T', i);
        const opusfe_node& t = n(i);
        switch (t.kind) {
            case Y_Name: dbg += "Name("; dbg_name(t.payload); dbg += ')'; break;
            case Y_Null: dbg += "Null"; break;
            case Y_Reference: case Y_MutableReference: case Y_ArrayReference: case Y_MutableArrayReference:
                dbg += kind_name(t.kind); dbg += '('; type(t.first_child); dbg += ')'; break;
            case Y_Proc: {
                dbg += "Proc([";
                uint32_t c = t.first_child;
                for (uint32_t k = 0; k + 1 < t.nchild; k++) { if (k) dbg += ", "; type(c); c = n(c).next; }
                dbg += "], "; type(c); dbg += ')';
                break;
            }
            case Y_Record: {
                dbg += "Record([";
                uint32_t c = t.first_child;
                for (uint32_t k = 0; k < t.nchild; k++) { if (k) dbg += ", "; dbg += '('; dbg_name(n(c).payload); dbg += ", "; type(n(c).first_child); dbg += ')'; c = n(c).next; }
                dbg += "])";
                break;
            }
            case Y_Array: dbg += "Array("; dbg += std::to_string(v.tok_u64(v.tok(t.payload))); dbg += ", "; type(t.first_child); dbg += ')'; break;
            default: dbg += "?"; break;
        }
    }
    void expr_list(uint32_t c, uint32_t count) {
        dbg += '[';
        for (uint32_t k = 0; k < count; k++) { if (k) dbg += ", "; expr(c); c = n(c).next; }
        dbg += ']';
    }
    void expr(uint32_t i) {
        span_line('E', i);
        const opusfe_node& e = n(i);
        static const char* const OPN[] = {"Plus", "Minus", "Times", "Divide", "Modulo", "Equals", "LessThan", "GreaterThan",
                                          "LessThanEquals", "GreaterThanEquals", "Is", "And", "Or", "Not"};
        switch (e.kind) {
            case E_Integer: { const opusfe_token& t = v.tok(e.payload); dbg += "Integer("; dbg += std::to_string(v.tok_u64(t)); dbg += ", "; fmt_int_meta(dbg, t.aux); dbg += ')'; break; }
            case E_Float: {
                const opusfe_token& t = v.tok(e.payload);
                dbg += "Float("; dbg += fmt_f64_debug(v.tok_f64(t)); dbg += ", ";
                if (t.aux) { dbg += "Some("; dbg += std::to_string(t.aux); dbg += ')'; } else dbg += "None";
                dbg += ')';
                break;
            }
            case E_Null: dbg += "Null"; break;
            case E_Bool: dbg += e.sub ? "Bool(true)" : "Bool(false)"; break;
            case E_String: { uint32_t len; const uint8_t* p = v.tok_str(v.tok(e.payload), &len); dbg += "String("; fmt_bytes_list(dbg, p, len); dbg += ')'; break; }
            case E_Variable: dbg += "Variable("; dbg_name(e.payload); dbg += ')'; break;
            case E_VariableDefinition: dbg += "VariableDefinition("; dbg_name(e.payload); dbg += ", "; type(e.first_child); dbg += ')'; break;
            case E_Proc: {
                dbg += "Proc("; dbg_name(e.payload); dbg += ", [";
                uint32_t c = e.first_child;
                for (uint32_t k = 0; k < e.nchild; k++) { if (k) dbg += ", "; type(c); c = n(c).next; }
                dbg += "])";
                break;
            }
            case E_CallOrIndex: dbg += "CallOrIndex("; expr(e.first_child); dbg += ", "; expr_list(n(e.first_child).next, e.nchild - 1); dbg += ')'; break;
            case E_NamedCall: dbg += "NamedCall("; dbg_name(e.payload); dbg += ", "; expr_list(e.first_child, e.nchild); dbg += ')'; break;
            case E_Record: {
                dbg += "Record("; type(e.first_child); dbg += ", [";
                uint32_t c = n(e.first_child).next;
                for (uint32_t k = 0; k + 1 < e.nchild; k++) { if (k) dbg += ", "; dbg += '('; dbg_name(n(c).payload); dbg += ", "; expr(n(c).first_child); dbg += ')'; c = n(c).next; }
                dbg += "])";
                break;
            }
            case E_Operator: dbg += "Operator("; dbg += OPN[e.sub]; dbg += ", "; expr(e.first_child); dbg += ", "; expr(n(e.first_child).next); dbg += ')'; break;
            case E_Reference: case E_MutableReference: case E_ArrayReference: case E_MutableArrayReference: case E_Dereference:
            case E_Negate: case E_Parentheses:
                dbg += kind_name(e.kind); dbg += '('; expr(e.first_child); dbg += ')'; break;
            case E_Field: dbg += "Field("; dbg_name(e.payload); dbg += ", "; expr(e.first_child); dbg += ')'; break;
            case E_Cast: dbg += "Cast("; expr(e.first_child); dbg += ", "; type(n(e.first_child).next); dbg += ')'; break;
            case E_Sizeof: case E_Alignof: dbg += kind_name(e.kind); dbg += '('; type(e.first_child); dbg += ')'; break;
            default: dbg += "?"; break;
        }
    }
    void block(uint32_t b) {   // b may be OPUSFE_NONE: absent else block == empty Vec
        dbg += '[';
        if (b != OPUSFE_NONE) {
            uint32_t c = n(b).first_child;
            for (uint32_t k = 0; k < n(b).nchild; k++) { if (k) dbg += ", "; stmt(c); c = n(c).next; }
        }
        dbg += ']';
    }
    void stmt(uint32_t i) {
        span_line('S', i);
        const opusfe_node& s = n(i);
        switch (s.kind) {
            case S_Expression: dbg += "Expression("; expr(s.first_child); dbg += ')'; break;
            case S_VariableDefinition: dbg += "VariableDefinition("; dbg_name(s.payload); dbg += ", "; expr(s.first_child); dbg += ')'; break;
            case S_Assignment: dbg += "Assignment("; expr(s.first_child); dbg += ", "; expr(n(s.first_child).next); dbg += ')'; break;
            case S_Return: dbg += "Return("; expr(s.first_child); dbg += ')'; break;
            case S_If: case S_While: {
                uint32_t cond = s.first_child, then_b = n(cond).next, else_b = n(then_b).next;
                dbg += kind_name(s.kind); dbg += '('; expr(cond); dbg += ", "; block(then_b); dbg += ", "; block(else_b); dbg += ')';
                break;
            }
            case S_For: {
                uint32_t c = s.first_child;
                dbg += "For("; dbg_name(s.payload); dbg += ", ";
                if (s.sub) { dbg += "Some("; type(c); dbg += ')'; c = n(c).next; } else dbg += "None";
                uint32_t then_b = n(c).next, else_b = n(then_b).next;
                dbg += ", "; expr(c); dbg += ", "; block(then_b); dbg += ", "; block(else_b); dbg += ')';
                break;
            }
            case S_Break: dbg += "Break"; break;
            case S_Continue: dbg += "Continue"; break;
            default: dbg += "?"; break;
        }
    }
    // parse.rs:330-349 (Q16): `name[` args `)`; the return type is not printed but owns a span
    uint32_t sig(const opusfe_node& d) {
        dbg += name_of_token(v, d.payload); dbg += '[';
        uint32_t c = d.first_child;
        bool any = false;
        while (n(c).kind == X_NamedType) {
            if (any) dbg += ' ';
            dbg += name_of_token(v, n(c).payload); dbg += ": "; type(n(c).first_child);
            any = true; c = n(c).next;
        }
        dbg += ')';
        std::string save = std::move(dbg); dbg.clear(); type(c); dbg = std::move(save);
        return n(c).next;   // body block (Function) or NONE
    }
    void def(uint32_t i) {
        const opusfe_node& d = n(i);
        switch (d.kind) {
            case D_Function: { dbg += "Function("; uint32_t b = sig(d); dbg += ", "; block(b); dbg += ')'; break; }
            case D_Extern: dbg += "Extern("; sig(d); dbg += ')'; break;
            case D_Type: dbg += "Type("; dbg_name(d.payload); dbg += ", "; type(d.first_child); dbg += ')'; break;
            case D_Variable:
                dbg += "Variable("; dbg_name(d.payload); dbg += ", "; type(d.first_child); dbg += ", ";
                if (d.nchild > 1) { dbg += "Some("; expr(n(d.first_child).next); dbg += ')'; } else dbg += "None";
                dbg += ')';
                break;
            case D_Constant: dbg += "Constant("; dbg_name(d.payload); dbg += ", "; type(d.first_child); dbg += ", "; expr(n(d.first_child).next); dbg += ')'; break;
            case D_Import: case D_Library: {
                uint32_t len; const uint8_t* p = v.tok_str(v.tok(d.payload), &len);
                dbg += d.kind == D_Import ? "Import(" : "Library("; fmt_bytes_list(dbg, p, len); dbg += ')';
                break;
            }
            default: dbg += "?"; break;
        }
        dbg += " @ "; dbg += std::to_string(d.low - v.pos_off); dbg += ".."; dbg += std::to_string(d.high - v.pos_off); dbg += '\n';
    }
};

inline std::string dump_ast(const opusfe_unit* u, const DumpView& w = DumpView{}) {
    std::string out;
    AstText a{u, HostView{u, w.pos_off}, {}, {}};
    for (size_t k = w.d0; k < u->h_def_nodes.size() && k < w.d1; k++) {
        const uint32_t d = u->h_def_nodes[k];
        a.dbg.clear(); a.spans.clear();
        a.def(d);
        out += a.dbg;
        out += a.spans;
    }
    return out;
}

// "Ok\n" | "Unsupported\n" | "<Kind> <low> <len> [payload...] line <n>\n"
inline std::string dump_error_line(const opusfe_unit* u) {
    if (u->status == OPUSFE_UNSUPPORTED) return "Unsupported\n";
    if (!u->has_error) return "Ok\n";
    HostView v{u};
    std::string out;
    const opusfe_error& e = u->error;
    const opusfe_token* found = e.tok_index < u->h_tokens.size() ? &u->h_tokens[e.tok_index] : nullptr;
    fmt_error_body(out, e.kind, e.low, e.len, e.payload, v, found, (int)e.expected);
    out += " line "; out += std::to_string(e.line); out += '\n';
    return out;
}

}  // namespace opusfe

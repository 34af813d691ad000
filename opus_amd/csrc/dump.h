// Host-side canonical text renderers of the flat result arrays (formats: SURVEY.md §8c).  These
// serialise GPU results for tests / the CLI; they are outside the timed region.
#pragma once
#include <charconv>
#include <string>
#include "unit.h"

namespace opusfe {

static const char* const TOKEN_NAMES[] = {"LeftParenthesis", "RightParenthesis", "LeftSquareBracket", "RightSquareBracket",
    "LeftBrace", "RightBrace", "Colon", "Equals", "EqualsEquals", "LessThan", "GreaterThan", "LessThanEquals",
    "GreaterThanEquals", "Plus", "Minus", "Asterisk", "Slash", "Percent", "At", "Tilde", "Dot", "Comma", "QuestionMark",
    "Integer", "Float", "String", "Identifier", "Indent", "Dedent", "Var", "Const", "Return", "If", "Else", "While",
    "For", "Break", "Continue", "Extern", "Ref", "Mut", "Refs", "Muts", "Null", "False", "True", "Type", "Record", "Is",
    "In", "And", "Or", "Not", "Sizeof", "Alignof", "Import", "Library", "Proc", "Operator", "EndOfFile"};
static const char* const PERR_NAMES[] = {"UnexpectedCharacter", "UnexpectedEndOfFile", "InvalidEscapeSequence", "UnexpectedToken",
    "ExpectedFoundToken", "ExpectedIdentifier", "InvalidNumericSize"};

inline std::string fmt_f64_display(double v) {   // Rust `{}`: shortest round trip, never scientific
    char buf[512];
    auto r = std::to_chars(buf, buf + sizeof buf, v, std::chars_format::fixed);
    return std::string(buf, r.ptr);
}
inline std::string fmt_f64_debug(double v) {
    std::string s = fmt_f64_display(v);
    if (s.find('.') == std::string::npos) s += ".0";
    return s;
}

// String::from_utf8_lossy + <str as Debug>
inline void fmt_debug_str(std::string& out, const uint8_t* p, size_t n) {
    out += '"';
    auto put = [&](uint32_t c, const uint8_t* raw, size_t len) {
        switch (c) {
            case 0: out += "\\0"; return; case 9: out += "\\t"; return; case 13: out += "\\r"; return;
            case 10: out += "\\n"; return; case '\\': out += "\\\\"; return; case '"': out += "\\\""; return;
            default: break;
        }
        if ((c >= 0x20 && c < 0x7f) || c >= 0xa0) out.append((const char*)raw, len);
        else { char b[16]; snprintf(b, sizeof b, "\\u{%x}", c); out += b; }
    };
    static const uint8_t REPL[3] = {0xEF, 0xBF, 0xBD};
    size_t i = 0;
    while (i < n) {
        uint8_t b0 = p[i];
        if (b0 < 0x80) { put(b0, p + i, 1); i++; continue; }
        size_t need; uint8_t lo = 0x80, hi = 0xBF;
        if (b0 >= 0xC2 && b0 <= 0xDF) need = 1;
        else if (b0 == 0xE0) { need = 2; lo = 0xA0; } else if (b0 >= 0xE1 && b0 <= 0xEC) need = 2;
        else if (b0 == 0xED) { need = 2; hi = 0x9F; } else if (b0 == 0xEE || b0 == 0xEF) need = 2;
        else if (b0 == 0xF0) { need = 3; lo = 0x90; } else if (b0 >= 0xF1 && b0 <= 0xF3) need = 3;
        else if (b0 == 0xF4) { need = 3; hi = 0x8F; }
        else { put(0xFFFD, REPL, 3); i++; continue; }
        size_t k = 1; bool ok = true;
        for (; k <= need; k++) {
            if (i + k >= n) { ok = false; break; }
            uint8_t b = p[i + k], l = k == 1 ? lo : 0x80, h = k == 1 ? hi : 0xBF;
            if (b < l || b > h) { ok = false; break; }
        }
        if (!ok) { put(0xFFFD, REPL, 3); i += k; continue; }
        uint32_t c = need == 1 ? (b0 & 0x1F) : need == 2 ? (b0 & 0x0F) : (b0 & 0x07);
        for (size_t j = 1; j <= need; j++) c = (c << 6) | (p[i + j] & 0x3F);
        put(c, p + i, need + 1);
        i += need + 1;
    }
    out += '"';
}

// Which part of a unit a dump shows: everything, or (group mode) the slice of one file of a packed corpus with positions and global
// indices relative to that file, so that the text equals the dump of the file compiled alone.
struct DumpView { uint32_t pos_off = 0, glob_off = 0; uint64_t t0 = 0, t1 = ~0ull; uint32_t d0 = 0, d1 = 0xFFFFFFFFu, f0 = 0, f1 = 0xFFFFFFFFu; };

struct HostView {
    const opusfe_unit* u;
    uint32_t pos_off = 0;
    const uint8_t* src() const { return u->h_src.data(); }
    const opusfe_token& tok(uint32_t i) const { return u->h_tokens[i]; }
    uint64_t tok_u64(const opusfe_token& t) const { return (t.flags & OPUSFE_TF_INLINE) ? t.payload : u->h_val64[t.payload]; }
    double tok_f64(const opusfe_token& t) const { uint64_t b = u->h_val64[t.payload]; double d; memcpy(&d, &b, 8); return d; }
    const uint8_t* tok_str(const opusfe_token& t, uint32_t* n) const { const uint8_t* p = u->h_strpool.data() + t.payload; memcpy(n, p, 4); return p + 4; }
};

inline void fmt_bytes_list(std::string& out, const uint8_t* p, uint32_t n) {
    out += '[';
    for (uint32_t i = 0; i < n; i++) { if (i) out += ", "; out += std::to_string(p[i]); }
    out += ']';
}

inline void fmt_int_meta(std::string& out, uint16_t aux) {
    if (aux & 0x8000) { out += "Some(("; out += (aux & 0x100) ? "true" : "false"; out += ", "; out += std::to_string(aux & 0xFF); out += "))"; }
    else out += "None";
}

inline void fmt_token_debug(std::string& out, const HostView& v, const opusfe_token& t);

inline void fmt_error_body(std::string& out, uint32_t kind, uint64_t low, uint32_t len, uint64_t payload, const HostView& v, const opusfe_token* found, int expected_kind) {
    out += PERR_NAMES[kind]; out += ' '; out += std::to_string(low); out += ' '; out += std::to_string(len);
    switch (kind) {
        case PE_UnexpectedCharacter: case PE_InvalidNumericSize: out += ' '; out += std::to_string(payload); break;
        case PE_UnexpectedToken: case PE_ExpectedIdentifier: out += ' '; fmt_token_debug(out, v, *found); break;
        case PE_ExpectedFoundToken: out += ' '; out += TOKEN_NAMES[expected_kind]; out += ' '; fmt_token_debug(out, v, *found); break;
        default: break;
    }
}

inline void fmt_token_debug(std::string& out, const HostView& v, const opusfe_token& t) {
    switch (t.kind) {
        case T_Integer: out += "Integer("; out += std::to_string(v.tok_u64(t)); out += ", "; fmt_int_meta(out, t.aux); out += ')'; break;
        case T_Float:
            out += "Float("; out += fmt_f64_debug(v.tok_f64(t)); out += ", ";
            if (t.aux) { out += "Some("; out += std::to_string(t.aux); out += ')'; } else out += "None";
            out += ')'; break;
        case T_String: { uint32_t n; const uint8_t* p = v.tok_str(t, &n); out += "String("; fmt_bytes_list(out, p, n); out += ')'; break; }
        case T_Identifier: out += "Identifier("; fmt_debug_str(out, v.src() + t.low, t.len); out += ')'; break;
        case T_Error: {
            uint64_t sp = v.u->h_val64[t.payload], pl = v.u->h_val64[t.payload + 1];
            out += "Err("; fmt_error_body(out, t.aux, (uint32_t)sp - v.pos_off, (uint32_t)(sp >> 32), pl, v, nullptr, 0); out += ')';
            break;
        }
        default: out += TOKEN_NAMES[t.kind];
    }
}

inline std::string dump_tokens(const opusfe_unit* u, uint64_t limit, const DumpView& w = DumpView{}) {
    HostView v{u, w.pos_off};
    std::string out;
    uint64_t n = u->h_tokens.size() < limit ? u->h_tokens.size() : limit;
    if (w.t1 < n) n = w.t1;
    for (uint64_t i = w.t0; i < n; i++) {
        const opusfe_token& t = u->h_tokens[i];
        uint64_t end = (uint64_t)t.low + t.len;
        if (t.kind == T_Error && (t.flags & OPUSFE_TF_RESYNC)) { uint64_t sp = u->h_val64[t.payload]; end = (uint32_t)sp + (sp >> 32); }   // the reference resumes right behind the error span; `len` runs to where the context automaton returns to code
        out += std::to_string(t.low - w.pos_off); out += ' '; out += std::to_string(end - w.pos_off); out += ' ';
        fmt_token_debug(out, v, t);
        out += '\n';
    }
    return out;
}

}  // namespace opusfe

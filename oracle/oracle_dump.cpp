// ORACLE — TEST INFRASTRUCTURE ONLY (see opus_oracle.hpp).
// Canonical text dumps (SURVEY.md §8c "Canonical dump formats") + the C API used by tests / bench.
//   IR     : exactly `impl Display for IrGenerator` (generate_ir.rs:1201-1283), one function after the
//            other, each followed by "\n" (the eprintln! of frontend.rs:168).
//   Tokens : one per line  "<low> <high> <Debug>"  with Rust #[derive(Debug)] text of parse::Token.
//   AST    : per definition "<Debug of Definition> @ <low>..<high>" followed by one line per node
//            "  <class> <Kind> <low> <high>" in pre-order (class D/S/E/T).
#include "opus_oracle.hpp"
#include "oracle_internal.hpp"
#include <charconv>
#include <cstring>
#include <chrono>

namespace oracle {

static const char* TOK_NAMES[] = {"LeftParenthesis", "RightParenthesis", "LeftSquareBracket", "RightSquareBracket",
    "LeftBrace", "RightBrace", "Colon", "Equals", "EqualsEquals", "LessThan", "GreaterThan", "LessThanEquals",
    "GreaterThanEquals", "Plus", "Minus", "Asterisk", "Slash", "Percent", "At", "Tilde", "Dot", "Comma", "QuestionMark",
    "Integer", "Float", "String", "Identifier", "Indent", "Dedent", "Var", "Const", "Return", "If", "Else", "While",
    "For", "Break", "Continue", "Extern", "Ref", "Mut", "Refs", "Muts", "Null", "False", "True", "Type", "Record", "Is",
    "In", "And", "Or", "Not", "Sizeof", "Alignof", "Import", "Library", "Proc", "Operator", "EndOfFile"};
static const char* OP_NAMES[] = {"Plus", "Minus", "Times", "Divide", "Modulo", "Equals", "LessThan", "GreaterThan",
    "LessThanEquals", "GreaterThanEquals", "Is", "And", "Or", "Not"};
static const char* PERR_NAMES[] = {"UnexpectedCharacter", "UnexpectedEndOfFile", "InvalidEscapeSequence", "UnexpectedToken",
    "ExpectedFoundToken", "ExpectedIdentifier", "InvalidNumericSize"};
static const char* SERR_NAMES[] = {"ParseError", "UndefinedVariable", "ShadowedName", "UndefinedFunction",
    "NoOverloadForFunction", "AmbiguousOverload", "AmbiguousFunctionReference", "UnexpectedType", "InvalidType",
    "InvalidOperandType", "InvalidOperandTypes", "FunctionMightNotReturn", "BreakOutsideLoop", "ContinueOutsideLoop",
    "InvalidLvalue", "ImmutableLvalue", "ArrayPointerToNonArray", "InvalidCast", "UndefinedType", "FieldAccessOnNonRecord",
    "FieldDoesNotExist", "CallOfNonFunction", "ReferenceToBuiltinFunction", "AttemptToCallOrIndexConstant",
    "InvalidNumberOfArgs", "InvalidConstantExpression"};
static const char* CAST_NAMES[] = {"None", "Pointer", "PointerType", "Integer", "PointerToInteger", "IntegerToPointer",
    "PointerToArray", "Error"};

// Rust `{}` for f64: shortest round-trip digits, never scientific.  std::to_chars(fixed) without a
// precision gives exactly that (SURVEY §7 hard part 5).
static std::string f64_display(double v) {
    char buf[512];
    auto r = std::to_chars(buf, buf + sizeof buf, v, std::chars_format::fixed);
    return std::string(buf, r.ptr);
}
// `{:?}` for f64 as used by this project's dumps: Display plus ".0" when integral.
static std::string f64_debug(double v) {
    std::string s = f64_display(v);
    if (s.find('.') == std::string::npos) s += ".0";
    return s;
}

// String::from_utf8_lossy (maximal-subpart replacement) followed by str's Debug escaping.
static std::string rust_debug_bytes_as_str(const uint8_t* p, size_t n) {
    std::string out = "\"";
    size_t i = 0;
    auto emit = [&](uint32_t c, const uint8_t* raw, size_t len) {
        switch (c) {
            case 0: out += "\\0"; return;
            case '\t': out += "\\t"; return;
            case '\r': out += "\\r"; return;
            case '\n': out += "\\n"; return;
            case '\\': out += "\\\\"; return;
            case '"': out += "\\\""; return;
            default: break;
        }
        bool printable = (c >= 0x20 && c < 0x7f) || c >= 0xa0;
        if (printable) out.append((const char*)raw, len);
        else { char b[16]; snprintf(b, sizeof b, "\\u{%x}", c); out += b; }
    };
    static const uint8_t REPL[3] = {0xEF, 0xBF, 0xBD};
    while (i < n) {
        uint8_t b0 = p[i];
        if (b0 < 0x80) { emit(b0, p + i, 1); i++; continue; }
        size_t need = 0; uint8_t lo = 0x80, hi = 0xBF;
        if (b0 >= 0xC2 && b0 <= 0xDF) need = 1;
        else if (b0 == 0xE0) { need = 2; lo = 0xA0; }
        else if (b0 >= 0xE1 && b0 <= 0xEC) need = 2;
        else if (b0 == 0xED) { need = 2; hi = 0x9F; }
        else if (b0 >= 0xEE && b0 <= 0xEF) need = 2;
        else if (b0 == 0xF0) { need = 3; lo = 0x90; }
        else if (b0 >= 0xF1 && b0 <= 0xF3) need = 3;
        else if (b0 == 0xF4) { need = 3; hi = 0x8F; }
        else { emit(0xFFFD, REPL, 3); i++; continue; }
        size_t k = 1; bool ok = true;
        for (; k <= need; k++) {
            if (i + k >= n) { ok = false; break; }
            uint8_t b = p[i + k];
            uint8_t l = (k == 1) ? lo : 0x80, h = (k == 1) ? hi : 0xBF;
            if (b < l || b > h) { ok = false; break; }
        }
        if (!ok) { emit(0xFFFD, REPL, 3); i += k; continue; }
        uint32_t c = (need == 1) ? (b0 & 0x1F) : (need == 2) ? (b0 & 0x0F) : (b0 & 0x07);
        for (size_t j = 1; j <= need; j++) c = (c << 6) | (p[i + j] & 0x3F);
        emit(c, p + i, need + 1);
        i += need + 1;
    }
    out += "\"";
    return out;
}
// String::from_utf8_lossy alone (parse.rs:113-116 prints a String token through it): valid sequences verbatim, U+FFFD per maximal invalid subpart
static std::string utf8_lossy(const uint8_t* p, size_t n) {
    std::string out;
    size_t i = 0;
    while (i < n) {
        const uint8_t b0 = p[i];
        if (b0 < 0x80) { out += (char)b0; i++; continue; }
        size_t need = 0; uint8_t lo = 0x80, hi = 0xBF;
        if (b0 >= 0xC2 && b0 <= 0xDF) need = 1;
        else if (b0 == 0xE0) { need = 2; lo = 0xA0; }
        else if ((b0 >= 0xE1 && b0 <= 0xEC) || b0 == 0xEE || b0 == 0xEF) need = 2;
        else if (b0 == 0xED) { need = 2; hi = 0x9F; }
        else if (b0 == 0xF0) { need = 3; lo = 0x90; }
        else if (b0 >= 0xF1 && b0 <= 0xF3) need = 3;
        else if (b0 == 0xF4) { need = 3; hi = 0x8F; }
        else { out += "\xEF\xBF\xBD"; i++; continue; }
        size_t k = 1; bool ok = true;
        for (; k <= need; k++) {
            if (i + k >= n) { ok = false; break; }
            const uint8_t b = p[i + k], l = k == 1 ? lo : 0x80, h = k == 1 ? hi : 0xBF;
            if (b < l || b > h) { ok = false; break; }
        }
        if (!ok) { out += "\xEF\xBF\xBD"; i += k; continue; }
        out.append((const char*)p + i, need + 1);
        i += need + 1;
    }
    return out;
}
static size_t utf8_len(uint8_t b0) { return b0 < 0x80 ? 1 : b0 >= 0xF0 ? 4 : b0 >= 0xE0 ? 3 : b0 >= 0xC0 ? 2 : 1; }
static std::string utf8_encode(uint32_t c) {      // char Display
    std::string s;
    if (c < 0x80) s += (char)c;
    else if (c < 0x800) { s += (char)(0xC0 | (c >> 6)); s += (char)(0x80 | (c & 0x3F)); }
    else if (c < 0x10000) { s += (char)(0xE0 | (c >> 12)); s += (char)(0x80 | ((c >> 6) & 0x3F)); s += (char)(0x80 | (c & 0x3F)); }
    else { s += (char)(0xF0 | (c >> 18)); s += (char)(0x80 | ((c >> 12) & 0x3F)); s += (char)(0x80 | ((c >> 6) & 0x3F)); s += (char)(0x80 | (c & 0x3F)); }
    return s;
}
static std::string rust_debug_str(std::string_view s) { return rust_debug_bytes_as_str((const uint8_t*)s.data(), s.size()); }
static std::string bytes_debug(const std::vector<uint8_t>& b) {
    std::string s = "[";
    for (size_t i = 0; i < b.size(); i++) { if (i) s += ", "; s += std::to_string(b[i]); }
    return s + "]";
}

static std::string token_debug(const Token& t) {
    switch (t.kind) {
        case Tok::Integer: {
            std::string s = "Integer(" + std::to_string(t.ival) + ", ";
            if (t.has_meta) s += std::string("Some((") + (t.is_signed ? "true" : "false") + ", " + std::to_string(t.size) + "))";
            else s += "None";
            return s + ")";
        }
        case Tok::Float: {
            std::string s = "Float(" + f64_debug(t.fval) + ", ";
            if (t.has_meta) s += "Some(" + std::to_string(t.size) + ")"; else s += "None";
            return s + ")";
        }
        case Tok::String: return "String(" + bytes_debug(t.bytes) + ")";
        case Tok::Identifier: return "Identifier(" + rust_debug_str(t.ident) + ")";
        default: return TOK_NAMES[(int)t.kind];
    }
}

static std::string error_line(const ParseError& e) {
    std::string s = std::string(PERR_NAMES[(int)e.kind]) + " " + std::to_string(e.span.low) + " " + std::to_string(e.span.len);
    switch (e.kind) {
        case PErr::UnexpectedCharacter: s += " " + std::to_string(e.ch); break;
        case PErr::InvalidNumericSize: s += " " + std::to_string(e.size); break;
        case PErr::UnexpectedToken: case PErr::ExpectedIdentifier: s += " " + token_debug(e.found); break;
        case PErr::ExpectedFoundToken: s += " " + token_debug(e.expected) + " " + token_debug(e.found); break;
        default: break;
    }
    return s;
}

std::string dump_tokens(const Unit& u) {
    std::string out;
    for (auto& e : u.token_log) {
        out += std::to_string(e.low) + " " + std::to_string(e.high) + " ";
        if (e.is_error) out += "Err(" + error_line(e.error) + ")"; else out += token_debug(e.token);
        out += "\n";
    }
    return out;
}

std::string dump_error(const Unit& u) {
    if (u.unsupported) return "Unsupported\n";
    if (!u.has_parse_error) return "Ok\n";
    // line number rule: frontend.rs:213-225
    size_t line = 1;
    for (size_t i = 0; i < u.parse_error.span.low && i < u.source.size(); i++) if (u.source[i] == '\n') line++;
    return error_line(u.parse_error) + " line " + std::to_string(line) + "\n";
}

std::string dump_semantic_errors(const Unit& u) {
    std::string out;
    for (auto& e : u.compiler->errors) out += std::string(SERR_NAMES[(int)e.kind]) + " " + std::to_string(e.span.low) + " " + std::to_string(e.span.len) + "\n";
    return out;
}

// ---- error rendering: what the reference writes to stderr ----
// parse.rs:69-158 impl Display for Token
static std::string token_display(const Token& t) {
    static const char* TEXT[] = {"(", ")", "[", "]", "{", "}", ":", "=", "==", "<", ">", "<=", ">=", "+", "-", "*", "/", "%", "@", "\xC2\xAF" /* U+00AF */, ".", ",", "?",
        nullptr, nullptr, nullptr, nullptr, "<indent>", "<dedent>", "var", "const", "return", "if", "else", "while", "for", "break", "continue", "extern",
        "ref", "mut", "refs", "muts", "null", "false", "true", "type", "record", "is", "in", "and", "or", "not", "sizeof", "alignof", "import", "library",
        "proc", "operator", "<end of file>"};
    switch (t.kind) {
        case Tok::Integer:
            if (!t.has_meta) return std::to_string(t.ival);                                                               // :93-96
            if (t.size == 0) return std::to_string(t.ival) + (t.is_signed ? "o" : "s");                                 // :98-100
            return std::to_string(t.ival) + (t.is_signed ? "i" : "n") + std::to_string(t.size);                          // :101
        case Tok::Float: return f64_display(t.fval) + (t.has_meta ? "f" + std::to_string(t.size) : std::string());       // :105-112
        case Tok::String: return utf8_lossy(t.bytes.data(), t.bytes.size());                                            // :113-116
        case Tok::Identifier: return std::string(t.ident);
        default: return TEXT[(int)t.kind];
    }
}
// the line behind "Error: " for a parse::Error (compile.rs:788-815)
static std::string parse_error_message(const ParseError& e) {
    switch (e.kind) {
        case PErr::UnexpectedCharacter: return "unexpected character: " + utf8_encode(e.ch);
        case PErr::UnexpectedEndOfFile: return "unexpected end of file";
        case PErr::InvalidEscapeSequence: return "invalid escape sequence";
        case PErr::UnexpectedToken: return "unexpected token: " + token_display(e.found);
        case PErr::ExpectedFoundToken: return "unexpected token: expected " + token_display(e.expected) + ", found " + token_display(e.found);
        case PErr::ExpectedIdentifier: return "unexpected token: expected name, found " + token_display(e.found);
        default: return "invalid numeric size: " + std::to_string(e.size);
    }
}
static std::string pad4(size_t n) { std::string s = std::to_string(n); while (s.size() < 4) s = " " + s; return s; }      // {:>4}
// frontend.rs:201-256 print_span, for the one source the span lies in (:204-208 finds it by pointer distance)
static std::string print_span(const std::string& filename, const std::string& source, Span span) {
    const char* RED = "\x1b[31;1m"; const char* YEL = "\x1b[33m"; const char* RST = "\x1b[0m";
    const size_t low = span.low, high = span.low + span.len;
    std::string out = std::string(YEL) + filename + ":" + RST + "\n";                                                      // :209
    size_t line_low = 0, line_number = 1, position = 0;
    while (position < low) {                                                                                              // :216-225 (per character; '\n' is one byte)
        const bool nl = source[position] == '\n';
        position += utf8_len((uint8_t)source[position]);
        if (nl) { line_number++; line_low = position; }
    }
    size_t line_high = high;
    while (line_high < source.size() && source[line_high] != '\n') line_high += utf8_len((uint8_t)source[line_high]);      // :226-232
    if (line_high > source.size()) line_high = source.size();
    out += std::string(YEL) + pad4(line_number) + ":" + RST + " ";                                                        // :234
    for (size_t i = line_low; i < line_high;) {                                                                           // :235-253
        const size_t L = utf8_len((uint8_t)source[i]);
        if (i == low) out += RED; else if (i == high) out += RST;
        const char c = source[i];
        if (c == '\t') out += "    ";
        else if (c == '\n') {
            line_number++;
            out += std::string("\n") + YEL + pad4(line_number) + ":" + RST + " ";
            if (i >= low && i < high) out += RED;
        } else out.append(source, i, L);
        i += L;
    }
    out += std::string(RST) + "\n";                                                                                       // :254
    return out;
}
std::string dump_rendered_errors(const Unit& u, const std::string& filename) {
    const std::string head = "\x1b[31;1mError:\x1b[0m ";                                                                 // compile.rs:786
    if (u.unsupported) return "Unsupported\n";
    if (u.has_parse_error) return head + parse_error_message(u.parse_error) + "\n" + print_span(filename, u.source, u.parse_error.span);
    std::string out;
    for (auto& e : u.compiler->errors) out += head + e.message + "\n" + print_span(filename, u.source, e.span);
    return out;
}

// ---- AST ----
static const char* TK_NAMES[] = {"Name", "Null", "Reference", "MutableReference", "ArrayReference", "MutableArrayReference", "Proc", "Record", "Array"};
static const char* EK_NAMES[] = {"Integer", "Float", "Null", "Bool", "String", "Variable", "VariableDefinition", "Proc", "CallOrIndex",
    "NamedCall", "Record", "Operator", "Reference", "MutableReference", "ArrayReference", "MutableArrayReference", "Dereference",
    "Field", "Cast", "Negate", "Sizeof", "Alignof", "Parentheses"};
static const char* SK_NAMES[] = {"Expression", "VariableDefinition", "Assignment", "Return", "If", "While", "For", "Break", "Continue"};
[[maybe_unused]] static const char* DK_NAMES[] = {"Function", "Extern", "Type", "Variable", "Constant", "Import", "Library"};

struct AstDumper {
    const Compiler& c;
    std::string dbg, spans;
    void span_line(char cls, const char* kind, Span s) {
        spans += "  "; spans += cls; spans += " "; spans += kind; spans += " " + std::to_string(s.low) + " " + std::to_string(s.high()) + "\n";
    }
    void type(const PType& t) {
        span_line('T', TK_NAMES[(int)t.kind], c.type_span(&t));
        switch (t.kind) {
            case TK::Name: dbg += "Name(" + rust_debug_str(t.name) + ")"; break;
            case TK::Null: dbg += "Null"; break;
            case TK::Reference: case TK::MutableReference: case TK::ArrayReference: case TK::MutableArrayReference:
                dbg += TK_NAMES[(int)t.kind]; dbg += "("; type(*t.sub); dbg += ")"; break;
            case TK::Proc:
                dbg += "Proc([";
                for (size_t i = 0; i < t.args.size(); i++) { if (i) dbg += ", "; type(*t.args[i]); }
                dbg += "], "; type(*t.sub); dbg += ")"; break;
            case TK::Record:
                dbg += "Record([";
                for (size_t i = 0; i < t.fields.size(); i++) { if (i) dbg += ", "; dbg += "(" + rust_debug_str(t.fields[i].first) + ", "; type(*t.fields[i].second); dbg += ")"; }
                dbg += "])"; break;
            case TK::Array: dbg += "Array(" + std::to_string(t.size) + ", "; type(*t.sub); dbg += ")"; break;
        }
    }
    void expr(const Expression& e) {
        span_line('E', EK_NAMES[(int)e.kind], c.expression_span(&e));
        switch (e.kind) {
            case EK::Integer:
                dbg += "Integer(" + std::to_string(e.ival) + ", ";
                if (e.has_meta) dbg += std::string("Some((") + (e.is_signed ? "true" : "false") + ", " + std::to_string(e.size) + "))"; else dbg += "None";
                dbg += ")"; break;
            case EK::Float:
                dbg += "Float(" + f64_debug(e.fval) + ", ";
                if (e.has_meta) dbg += "Some(" + std::to_string(e.size) + ")"; else dbg += "None";
                dbg += ")"; break;
            case EK::Null: dbg += "Null"; break;
            case EK::Bool: dbg += e.bval ? "Bool(true)" : "Bool(false)"; break;
            case EK::String: dbg += "String(" + bytes_debug(e.bytes) + ")"; break;
            case EK::Variable: dbg += "Variable(" + rust_debug_str(e.name) + ")"; break;
            case EK::VariableDefinition: dbg += "VariableDefinition(" + rust_debug_str(e.name) + ", "; type(*e.type); dbg += ")"; break;
            case EK::Proc:
                dbg += "Proc(" + rust_debug_str(e.name) + ", [";
                for (size_t i = 0; i < e.types.size(); i++) { if (i) dbg += ", "; type(*e.types[i]); }
                dbg += "])"; break;
            case EK::CallOrIndex:
                dbg += "CallOrIndex("; expr(*e.a); dbg += ", [";
                for (size_t i = 0; i < e.args.size(); i++) { if (i) dbg += ", "; expr(*e.args[i]); }
                dbg += "])"; break;
            case EK::NamedCall:
                dbg += "NamedCall(" + rust_debug_str(e.name) + ", [";
                for (size_t i = 0; i < e.args.size(); i++) { if (i) dbg += ", "; expr(*e.args[i]); }
                dbg += "])"; break;
            case EK::Record:
                dbg += "Record("; type(*e.type); dbg += ", [";
                for (size_t i = 0; i < e.fields.size(); i++) { if (i) dbg += ", "; dbg += "(" + rust_debug_str(e.fields[i].first) + ", "; expr(*e.fields[i].second); dbg += ")"; }
                dbg += "])"; break;
            case EK::Operator:
                dbg += std::string("Operator(") + OP_NAMES[(int)e.op] + ", "; expr(*e.a); dbg += ", "; expr(*e.b); dbg += ")"; break;
            case EK::Reference: case EK::MutableReference: case EK::ArrayReference: case EK::MutableArrayReference:
            case EK::Dereference: case EK::Negate: case EK::Parentheses:
                dbg += EK_NAMES[(int)e.kind]; dbg += "("; expr(*e.a); dbg += ")"; break;
            case EK::Field: dbg += "Field(" + rust_debug_str(e.name) + ", "; expr(*e.a); dbg += ")"; break;
            case EK::Cast: dbg += "Cast("; expr(*e.a); dbg += ", "; type(*e.type); dbg += ")"; break;
            case EK::Sizeof: case EK::Alignof: dbg += EK_NAMES[(int)e.kind]; dbg += "("; type(*e.type); dbg += ")"; break;
        }
    }
    void block(const Block& b) {
        dbg += "[";
        for (size_t i = 0; i < b.size(); i++) { if (i) dbg += ", "; stmt(*b[i]); }
        dbg += "]";
    }
    void stmt(const Statement& s) {
        span_line('S', SK_NAMES[(int)s.kind], c.statement_span(&s));
        switch (s.kind) {
            case SK::Expression: dbg += "Expression("; expr(*s.a); dbg += ")"; break;
            case SK::VariableDefinition: dbg += "VariableDefinition(" + rust_debug_str(s.name) + ", "; expr(*s.a); dbg += ")"; break;
            case SK::Assignment: dbg += "Assignment("; expr(*s.a); dbg += ", "; expr(*s.b); dbg += ")"; break;
            case SK::Return: dbg += "Return("; expr(*s.a); dbg += ")"; break;
            case SK::If: case SK::While:
                dbg += SK_NAMES[(int)s.kind]; dbg += "("; expr(*s.a); dbg += ", "; block(s.then_block); dbg += ", "; block(s.else_block); dbg += ")"; break;
            case SK::For:
                dbg += "For(" + rust_debug_str(s.name) + ", ";
                if (s.type) { dbg += "Some("; type(*s.type); dbg += ")"; } else dbg += "None";
                dbg += ", "; expr(*s.a); dbg += ", "; block(s.then_block); dbg += ", "; block(s.else_block); dbg += ")"; break;
            case SK::Break: dbg += "Break"; break;
            case SK::Continue: dbg += "Continue"; break;
        }
    }
    // parse.rs:330-349 (Q16: `name[` ... `)`)
    void sig(const FunctionSignature& s) {
        dbg += std::string(s.name) + "[";
        bool any = false;
        for (auto& a : s.arguments) { if (any) dbg += " "; dbg += std::string(a.first) + ": "; type(*a.second); any = true; }
        dbg += ")";
        // the return type is not printed by the reference's Debug impl, but it has a span
        std::string save = std::move(dbg); dbg.clear(); type(*s.return_type); dbg = std::move(save);
    }
    void def(const Definition& d) {
        Span ds = c.definition_spans.at(&d);
        switch (d.kind) {
            case DK::Function: dbg += "Function("; sig(d.sig); dbg += ", "; block(d.body); dbg += ")"; break;
            case DK::Extern: dbg += "Extern("; sig(d.sig); dbg += ")"; break;
            case DK::Type: dbg += "Type(" + rust_debug_str(d.name) + ", "; type(*d.type); dbg += ")"; break;
            case DK::Variable:
                dbg += "Variable(" + rust_debug_str(d.name) + ", "; type(*d.type); dbg += ", ";
                if (d.value) { dbg += "Some("; expr(*d.value); dbg += ")"; } else dbg += "None";
                dbg += ")"; break;
            case DK::Constant: dbg += "Constant(" + rust_debug_str(d.name) + ", "; type(*d.type); dbg += ", "; expr(*d.value); dbg += ")"; break;
            case DK::Import: dbg += "Import(" + bytes_debug(d.path) + ")"; break;
            case DK::Library: dbg += "Library(" + bytes_debug(d.path) + ")"; break;
        }
        dbg += " @ " + std::to_string(ds.low) + ".." + std::to_string(ds.high()) + "\n";
    }
};

std::string dump_ast(const Unit& u) {
    std::string out;
    for (auto& d : u.definitions) {
        AstDumper a{*u.compiler, {}, {}};
        a.def(*d);
        out += a.dbg;
        out += a.spans;
    }
    return out;
}

// generate_ir.rs:1201-1283
std::string dump_ir(const Unit& u) {
    std::string f;
    const Compiler& c = *u.compiler;
    auto V = [](VariableId v) { return "%" + std::to_string(v); };
    for (auto& fn : u.ir) {
        f += c.function_to_string(fn.function) + "\n";
        for (size_t i = 0; i < fn.variables.size(); i++) f += "VAR %" + std::to_string(i) + ": " + c.type_to_string(fn.variables[i].typ) + "\n";
        bool first = true;
        for (size_t bi = 0; bi < fn.bbs.size(); bi++) {
            if (!first) f += "\n";
            first = false;
            f += "@" + std::to_string(bi);
            for (auto& in : fn.bbs[bi]) {
                f += "\n  ";
                auto op3 = [&](const char* n) { f += V(in.a) + " = " + n + " " + V(in.b) + ", " + V(in.c); };
                switch (in.kind) {
                    case IK::Integer: f += V(in.a) + " = " + std::to_string(in.ival); break;
                    case IK::Float: f += V(in.a) + " = " + f64_display(in.fval); break;
                    case IK::Null: f += V(in.a) + " = null"; break;
                    case IK::Bool: f += V(in.a) + " = " + (in.bval ? "true" : "false"); break;
                    case IK::String: f += V(in.a) + " = " + rust_debug_bytes_as_str(in.bytes.data(), in.bytes.size()); break;
                    case IK::Sizeof: f += V(in.a) + " = sizeof " + c.type_to_string(in.typ); break;
                    case IK::Alignof: f += V(in.a) + " = alignof " + c.type_to_string(in.typ); break;
                    case IK::Call:
                        // in.b == !0 (builtin) cannot reach a Call instruction
                        f += V(in.a) + " = call " + V(in.b);
                        for (auto a : in.args) f += ", " + std::to_string(a);
                        break;
                    case IK::Allocate: f += V(in.a) + " = alloc"; break;
                    case IK::Load: f += V(in.a) + " = load " + V(in.b); break;
                    case IK::Store: f += "in " + V(in.a) + " store " + V(in.b); break;
                    case IK::Field: f += V(in.a) + " = fieldptr " + V(in.b) + ", " + std::string(in.field); break;
                    case IK::BuiltinField: f += V(in.a) + " = builtinfieldptr " + V(in.b) + ", " + std::to_string(in.i1); break;
                    case IK::IndexPointer: f += V(in.a) + " = indexptr " + V(in.b) + ", " + V(in.c); break;
                    case IK::IndexArray: f += V(in.a) + " = indexarray " + V(in.b) + ", " + V(in.c); break;
                    case IK::Add: op3("add"); break;
                    case IK::Subtract: op3("subtract"); break;
                    case IK::Multiply: op3("multiply"); break;
                    case IK::Divide: op3("divide"); break;
                    case IK::Modulo: op3("modulo"); break;
                    case IK::Equals: op3("equals"); break;
                    case IK::LessThan: op3("lessthan"); break;
                    case IK::GreaterThan: op3("greaterthan"); break;
                    case IK::LessThanEquals: op3("lessthanequals"); break;
                    case IK::GreaterThanEquals: op3("greaterthanequals"); break;
                    case IK::Function: f += V(in.a) + " = global $" + c.function_to_string(in.func); break;
                    case IK::GlobalVariable: f += V(in.a) + " = globalvar %" + std::to_string(in.i1); break;
                    case IK::GlobalConstant: f += V(in.a) + " = globalconst %" + std::to_string(in.i1); break;
                    case IK::Not: f += V(in.a) + " = not " + V(in.b); break;
                    case IK::Negate: f += V(in.a) + " = negate " + V(in.b); break;
                    case IK::Cast:
                        f += V(in.a) + " = cast " + CAST_NAMES[(int)in.cast] + " [" + c.type_to_string(fn.variables[in.b].typ) + " to "
                             + c.type_to_string(fn.variables[in.a].typ) + "] " + V(in.b);
                        break;
                    case IK::Return: f += "return " + V(in.a); break;
                    case IK::Jump: f += "jump @" + std::to_string(in.i1); break;
                    case IK::Branch: f += "branch " + V(in.a) + ", @" + std::to_string(in.i1) + ", @" + std::to_string(in.i2); break;
                    case IK::Unreachable: f += "unreachable"; break;
                    case IK::Phi: f += V(in.a) + " = phi @" + std::to_string(in.i1) + " => " + V(in.b) + ", @" + std::to_string(in.i2) + " => " + V(in.c); break;
                    case IK::BreakPlaceholder: f += "<break placeholder>"; break;
                    case IK::Error: f += V(in.a) + " = error"; break;
                }
            }
        }
        f += "\n";
    }
    return f;
}

}  // namespace oracle

// ------------------------------------------------------------------------------------------------
// C API (ctypes-friendly).  All returned strings are malloc'ed; free with oracle_free.
extern "C" {

struct oracle_counts { uint64_t tokens, nodes, instrs, vars, definitions, functions; int32_t status; };

static char* dup_string(const std::string& s) {
    char* p = (char*)malloc(s.size() + 1);
    memcpy(p, s.data(), s.size());
    p[s.size()] = 0;
    return p;
}

// what: 0 tokens, 1 ast, 2 ir, 3 error, 4 semantic errors, 5 lexer alone, 6 the stderr text of print_error + print_span (file name "input.opus")
char* oracle_dump(const char* src, uint64_t len, int what, uint64_t* out_len) {
    if (what == 5) {   // lexer alone
        oracle::Unit u;
        u.source.assign(src, len);
        u.compiler = std::make_unique<oracle::Compiler>();
        oracle::lex_only(u);
        std::string s = u.unsupported ? std::string("Unsupported\n") : oracle::dump_tokens(u);
        if (out_len) *out_len = s.size();
        return dup_string(s);
    }
    auto u = oracle::run(src, len, what == 2 || what == 4 || what == 6);
    std::string s;
    switch (what) {
        case 0: s = oracle::dump_tokens(*u); break;
        case 1: s = oracle::dump_ast(*u); break;
        case 2: s = oracle::dump_ir(*u); break;
        case 3: s = oracle::dump_error(*u); break;
        case 4: s = oracle::dump_semantic_errors(*u); break;
        case 6: s = oracle::dump_rendered_errors(*u, "input.opus"); break;
    }
    if (out_len) *out_len = s.size();
    return dup_string(s);
}

void oracle_free(char* p) { free(p); }

// Timed CPU baseline entry: source in memory -> IR structures built, dumps disabled.
// Returns seconds.
double oracle_time(const char* src, uint64_t len, int lower, oracle_counts* counts) {
    auto t0 = std::chrono::steady_clock::now();
    auto u = oracle::run(src, len, lower != 0, false);
    auto t1 = std::chrono::steady_clock::now();
    if (counts) {
        counts->tokens = u->n_tokens; counts->nodes = u->n_nodes; counts->instrs = u->n_instrs; counts->vars = u->n_vars;
        counts->definitions = u->definitions.size(); counts->functions = u->ir.size();
        counts->status = u->unsupported ? 2 : u->has_parse_error ? 1 : (u->compiler->has_errors ? 3 : 0);
    }
    return std::chrono::duration<double>(t1 - t0).count();
}

}  // extern "C"

#ifdef ORACLE_MAIN
#include <fstream>
#include <iostream>
#include <sstream>
// CLI mirroring `opus -Z file.opus` for the hot path only:  oracle_cli [--tokens|--ast|--ir|--error|--sem] file
int main(int argc, char** argv) {
    int what = 2; const char* path = nullptr;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        if (a == "--tokens") what = 0; else if (a == "--ast") what = 1; else if (a == "--ir" || a == "-Z") what = 2;
        else if (a == "--lex") what = 5; else if (a == "--error") what = 3; else if (a == "--sem") what = 4; else path = argv[i];
    }
    if (!path) { std::cerr << "usage: oracle_cli [--tokens|--ast|--ir|--error|--sem] file\n"; return 2; }
    std::ifstream f(path, std::ios::binary);
    if (!f) { std::cerr << "error: could not open file " << path << "\n"; return 1; }
    std::stringstream ss; ss << f.rdbuf();
    std::string src = ss.str();
    uint64_t n;
    char* s = oracle_dump(src.data(), src.size(), what, &n);
    fwrite(s, 1, n, stdout);
    oracle_free(s);
    return 0;
}
#endif

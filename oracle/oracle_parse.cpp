// ORACLE — TEST INFRASTRUCTURE ONLY (see opus_oracle.hpp).  Restates reference src/parse.rs.
#include "opus_oracle.hpp"
#include "oracle_internal.hpp"
#include <cstdlib>
#include <cstring>

namespace oracle {

// #[derive(PartialEq)] on Token (parse.rs:5).  f64 compares with IEEE ==.
bool Token::operator==(const Token& o) const {
    if (kind != o.kind) return false;
    switch (kind) {
        case Tok::Integer: return ival == o.ival && has_meta == o.has_meta && (!has_meta || (is_signed == o.is_signed && size == o.size));
        case Tok::Float: return fval == o.fval && has_meta == o.has_meta && (!has_meta || size == o.size);
        case Tok::String: return bytes == o.bytes;
        case Tok::Identifier: return ident == o.ident;
        default: return true;
    }
}

static Token simple(Tok k) { Token t; t.kind = k; return t; }

// parse.rs:179-196
static std::optional<Op> binary_from_token(const Token& t) {
    switch (t.kind) {
        case Tok::Plus: return Op::Plus;
        case Tok::Minus: return Op::Minus;
        case Tok::Asterisk: return Op::Times;
        case Tok::Slash: return Op::Divide;
        case Tok::Percent: return Op::Modulo;
        case Tok::EqualsEquals: return Op::Equals;
        case Tok::LessThan: return Op::LessThan;
        case Tok::GreaterThan: return Op::GreaterThan;
        case Tok::LessThanEquals: return Op::LessThanEquals;
        case Tok::GreaterThanEquals: return Op::GreaterThanEquals;
        case Tok::Is: return Op::Is;
        case Tok::And: return Op::And;
        case Tok::Or: return Op::Or;
        default: return std::nullopt;
    }
}
// parse.rs:198-203
static std::optional<Op> unary_from_token(const Token& t) {
    if (t.kind == Tok::Not) return Op::Not;
    return std::nullopt;
}
// parse.rs:205-222
int op_precedence(Op o) {
    switch (o) {
        case Op::Or: return 0;
        case Op::And: return 1;
        case Op::Not: return 2;
        case Op::Equals: case Op::LessThan: case Op::GreaterThan: case Op::LessThanEquals:
        case Op::GreaterThanEquals: case Op::Is: return 20;
        case Op::Plus: case Op::Minus: return 30;
        case Op::Times: case Op::Divide: case Op::Modulo: return 31;
    }
    return 0;
}
// parse.rs:224-243
const char* op_symbol(Op o) {
    switch (o) {
        case Op::Plus: return "+"; case Op::Minus: return "-"; case Op::Times: return "*";
        case Op::Divide: return "/"; case Op::Modulo: return "%"; case Op::Equals: return "=";
        case Op::LessThan: return "<"; case Op::GreaterThan: return ">";
        case Op::LessThanEquals: return "<="; case Op::GreaterThanEquals: return ">=";
        case Op::Is: return "is"; case Op::And: return "and"; case Op::Or: return "or";
        case Op::Not: return "not";
    }
    return "";
}

// ------------------------------------------------------------------------------------------------
// Parser  (parse.rs:364-391)
struct Parser {
    Compiler& compiler;
    std::string_view source;
    uint32_t indent = 0;
    int32_t indents_to_output = 0;
    size_t real_position = 0;
    bool is_end_of_file = false;
    uint32_t ignore_dents = 0;
    std::optional<Token> peeked_token;
    size_t position = 0;
    size_t token_low = 0;
    std::vector<TokenLogEntry>* log = nullptr;
    uint64_t n_nodes = 0;

    Parser(Compiler& c, std::string_view s) : compiler(c), source(s) {}

    static constexpr uint32_t NONE = 0xFFFFFFFFu;  // Option<char>::None

    // Decode one scalar at `pos`; returns NONE at end.  Input is assumed to be valid UTF-8
    // (frontend.rs:85 read_to_string guarantees it in the reference).
    uint32_t decode(size_t pos, size_t* len) const {
        if (pos >= source.size()) { *len = 0; return NONE; }
        unsigned char b0 = (unsigned char)source[pos];
        if (b0 < 0x80) { *len = 1; return b0; }
        size_t n = (b0 >= 0xF0) ? 4 : (b0 >= 0xE0) ? 3 : (b0 >= 0xC0) ? 2 : 1;
        if (pos + n > source.size()) n = source.size() - pos;
        uint32_t c = (n == 4) ? (b0 & 0x07) : (n == 3) ? (b0 & 0x0F) : (n == 2) ? (b0 & 0x1F) : b0;
        for (size_t i = 1; i < n; i++) c = (c << 6) | ((unsigned char)source[pos + i] & 0x3F);
        *len = n;
        return c;
    }
    // parse.rs:415-421
    uint32_t advance() {
        size_t n; uint32_t c = decode(real_position, &n);
        if (c != NONE) { real_position += n; position = real_position; }
        return c;
    }
    // parse.rs:423-425
    uint32_t peek() const { size_t n; return decode(real_position, &n); }

    // char::is_whitespace = the Unicode White_Space property: 25 code points, unchanged since Unicode 6.3 (2013), so the
    // same in every Rust release that can build the reference.  (is_alphanumeric is a different matter: its tables grow
    // with every Unicode version and are not pinned, see the identifier loop below.)
    static bool unicode_ws(uint32_t c) {
        return c == ' ' || (c >= 9 && c <= 13) || c == 0x85 || c == 0xA0 || c == 0x1680 || (c >= 0x2000 && c <= 0x200A) || c == 0x2028 || c == 0x2029
            || c == 0x202F || c == 0x205F || c == 0x3000;
    }
    bool is_ws(uint32_t c) const { return c != NONE && unicode_ws(c); }

    // parse.rs:427-479.  The tail recursion at :458 is written as a loop; a recursive call starts
    // with new_indent = 0, measure_indents = check_indent = (real_position == 0) = false.
    int32_t skip_whitespace() {
        uint32_t new_indent = 0;
        bool measure_indents = real_position == 0;
        bool check_indent = measure_indents;
        for (;;) {
            for (;;) {
                uint32_t c = peek();
                if (!is_ws(c)) break;
                if (ignore_dents == 0) {
                    if (c == '\n') { new_indent = 0; measure_indents = true; check_indent = true; }
                    else if (measure_indents) {
                        if (c == '\t') new_indent += 1; else measure_indents = false;
                    }
                }
                advance();
            }
            if (peek() == ';') {
                for (;;) { uint32_t c = peek(); if (c == NONE || c == '\n') break; advance(); }
                // return self.skip_whitespace()
                new_indent = 0;
                measure_indents = real_position == 0;
                check_indent = measure_indents;
                continue;
            }
            break;
        }
        if (real_position >= source.size() && !is_end_of_file) {
            is_end_of_file = true;
            return -(int32_t)indent;
        }
        int32_t ret;
        if (check_indent) { ret = (int32_t)new_indent - (int32_t)indent; indent = new_indent; }
        else ret = 0;
        // parse.rs:474-476 is dead code (all whitespace was consumed above).
        return ret;
    }

    [[noreturn]] void fail(ParseError e) { throw e; }
    Span sp(size_t lo, size_t hi) const { return Span{lo, hi - lo}; }
    // parse.rs:730-733
    Span char_at(size_t pos) const { size_t n; decode(pos, &n); return Span{pos, n}; }

    uint64_t n_tokens = 0;
    void log_token(size_t low, const Token& t) {
        n_tokens++;
        if (log) { TokenLogEntry e; e.low = low; e.high = real_position; e.is_error = false; e.token = t; log->push_back(std::move(e)); }
    }

    Token parse_token();       // parse.rs:481-721 (wrapper logs lex errors too)
    Token parse_token_inner();

    // parse.rs:735-741
    Token peek_token() {
        size_t old_position = position;
        Token token = parse_token();
        peeked_token = token;
        position = old_position;
        return token;
    }
    // parse.rs:411-413 — an Err from peek_token is just "not EOF" (Q7)
    bool is_at_end_of_file() {
        try { return peek_token() == simple(Tok::EndOfFile); } catch (ParseError&) { return false; }
    }
    // parse.rs:743-751
    void expect(Tok expected) {
        Token found = parse_token();
        if (found.kind != expected) {
            ParseError e; e.kind = PErr::ExpectedFoundToken; e.span = sp(token_low, position);
            e.expected = simple(expected); e.found = found; fail(e);
        }
    }
    // parse.rs:753-761
    bool maybe_consume(Tok expected) {
        Token token = peek_token();
        if (token.kind == expected) { try { parse_token(); } catch (ParseError&) {} return true; }
        return false;
    }
    // parse.rs:763-771
    uint64_t parse_integer() {
        Token t = parse_token();
        if (t.kind == Tok::Integer) return t.ival;
        ParseError e; e.kind = PErr::UnexpectedToken; e.span = sp(token_low, position); e.found = t; fail(e);
    }
    // parse.rs:723-728
    std::vector<uint8_t> parse_string() {
        Token t = parse_token();
        if (t.kind == Tok::String) return t.bytes;
        ParseError e; e.kind = PErr::UnexpectedToken; e.span = sp(token_low, position); e.found = t; fail(e);
    }
    // parse.rs:773-801
    std::string_view parse_identifier() {
        Token t = parse_token();
        if (t.kind == Tok::Identifier) return t.ident;
        if (t.kind == Tok::Operator) {
            Token token = parse_token();
            Span span = sp(token_low, position);
            auto b = binary_from_token(token);
            if (b && token.kind != Tok::And && token.kind != Tok::Or) return std::string_view(op_symbol(*b));
            auto u = unary_from_token(token);
            if (u) return std::string_view(op_symbol(*u));
            if (token.kind == Tok::For || token.kind == Tok::Continue) return source.substr(span.low, span.len);
            ParseError e; e.kind = PErr::UnexpectedToken; e.span = span; e.found = token; fail(e);
        }
        ParseError e; e.kind = PErr::ExpectedIdentifier; e.span = sp(token_low, position); e.found = t; fail(e);
    }

    ExprBox parse_expression();
    void correct_precedence(Expression* e);
    template <class T, class F> std::vector<T> parse_comma_separated(Tok close, F parse_item);
    TypeBox parse_type();
    ExprBox parse_atom();
    StmtBox parse_statement();
    Block parse_block();
    DefBox parse_definition();
    FunctionSignature parse_function_signature();

    ExprBox new_expr(EK k) { n_nodes++; auto e = std::make_unique<Expression>(); e->kind = k; return e; }
    TypeBox new_type(TK k) { n_nodes++; auto t = std::make_unique<PType>(); t->kind = k; return t; }
};

Token Parser::parse_token() {
    size_t before = real_position;
    bool fresh = !peeked_token.has_value();
    try {
        Token t = parse_token_inner();
        if (fresh) log_token(token_low, t);
        return t;
    } catch (ParseError& e) {
        n_tokens++;
        if (log) { TokenLogEntry le; le.low = token_low; le.high = real_position; le.is_error = true; le.error = e; log->push_back(std::move(le)); }
        (void)before;
        throw;
    }
}

// parse.rs:481-721
Token Parser::parse_token_inner() {
    if (peeked_token.has_value()) {
        Token t = std::move(*peeked_token);
        peeked_token.reset();
        position = real_position;
        return t;
    }
    if (ignore_dents > 0) skip_whitespace();
    else if (indents_to_output == 0) indents_to_output = skip_whitespace();

    token_low = position;

    if (indents_to_output > 0) { indents_to_output -= 1; return simple(Tok::Indent); }
    else if (indents_to_output < 0) { indents_to_output += 1; return simple(Tok::Dedent); }

    size_t old_position = position;
    uint32_t c = advance();
    if (c == NONE) return simple(Tok::EndOfFile);
    switch (c) {
        case '(': return simple(Tok::LeftParenthesis);
        case ')': return simple(Tok::RightParenthesis);
        case '[': return simple(Tok::LeftSquareBracket);
        case ']': return simple(Tok::RightSquareBracket);
        case '{': return simple(Tok::LeftBrace);
        case '}': return simple(Tok::RightBrace);
        case ':': return simple(Tok::Colon);
        case '=': if (peek() == '=') { advance(); return simple(Tok::EqualsEquals); } return simple(Tok::Equals);
        case '<': if (peek() == '=') { advance(); return simple(Tok::LessThanEquals); } return simple(Tok::LessThan);
        case '>': if (peek() == '=') { advance(); return simple(Tok::GreaterThanEquals); } return simple(Tok::GreaterThan);
        case '+': return simple(Tok::Plus);
        case '-': return simple(Tok::Minus);
        case '*': return simple(Tok::Asterisk);
        case '/': return simple(Tok::Slash);
        case '%': return simple(Tok::Percent);
        case '@': return simple(Tok::At);
        case '~': return simple(Tok::Tilde);
        case '.': return simple(Tok::Dot);
        case ',': return simple(Tok::Comma);
        case '?': return simple(Tok::QuestionMark);
        default: break;
    }
    if (c >= '0' && c <= '9') {
        uint64_t i = c - '0';
        for (;;) { uint32_t d = peek(); if (d == NONE || d < '0' || d > '9') break; advance(); i *= 10; i += d - '0'; }
        if (peek() == '.') {
            advance();
            for (;;) { uint32_t d = peek(); if (d == NONE || d < '0' || d > '9') break; advance(); }
            // parse.rs:562 — str::parse::<f64> is correctly rounded; so is glibc strtod.
            std::string text(source.substr(old_position, real_position - old_position));
            double f = strtod(text.c_str(), nullptr);
            Token t; t.kind = Tok::Float; t.fval = f;
            if (peek() == 'f') {
                advance();
                uint32_t a = advance();      // two *characters*, whatever they are (parse.rs:565-566); the size arithmetic below
                uint32_t b = advance();      // is on their code points (:572,577), so non-ASCII needs no Unicode table
                if (a == '3' && b == '2') { t.has_meta = true; t.size = 32; }
                else if (a == '6' && b == '4') { t.has_meta = true; t.size = 64; }
                else if (a != NONE && b == NONE) {
                    ParseError e; e.kind = PErr::InvalidNumericSize; e.size = (uint32_t)a - (uint32_t)'0';
                    e.span = sp(old_position, real_position); fail(e);
                } else if (a != NONE && b != NONE) {
                    ParseError e; e.kind = PErr::InvalidNumericSize;
                    e.size = ((uint32_t)a - (uint32_t)'0') * 10u + ((uint32_t)b - (uint32_t)'0');  // wrapping u32 (release)
                    e.span = sp(old_position, real_position); fail(e);
                } else {
                    // parse.rs:579 rewinds to the token start: the reference re-lexes the same token
                    // forever (Q4).  There is no output to be bit-exact with.
                    throw Unsupported{"float literal followed by 'f' at end of file (reference hangs)", old_position};
                }
            }
            return t;
        } else {
            Token t; t.kind = Tok::Integer; t.ival = i;
            bool is_signed = true;
            uint32_t p = peek();
            if (p == 's') { is_signed = false; t.has_meta = true; t.size = 0; }
            else if (p == 'o') { is_signed = true; t.has_meta = true; t.size = 0; }
            else if (p == 'i' || p == 'n') {
                is_signed = p == 'i';
                advance();
                size_t old_position2 = real_position;
                uint32_t a = advance();
                uint32_t b = peek();
                t.has_meta = true;
                if (a == '8') t.size = 8;
                else if (a == '1' && b == '6') { advance(); t.size = 16; }
                else if (a == '3' && b == '2') { advance(); t.size = 32; }
                else if (a == '6' && b == '4') { advance(); t.size = 64; }
                else if (a != NONE && a >= '0' && a <= '9' && b != NONE && b >= '0' && b <= '9') {
                    advance();
                    ParseError e; e.kind = PErr::InvalidNumericSize; e.size = (a - '0') * 10 + b - '0';
                    e.span = sp(old_position2, real_position); fail(e);
                } else if (a != NONE && a >= '0' && a <= '9') {
                    ParseError e; e.kind = PErr::InvalidNumericSize; e.size = a - '0';
                    e.span = sp(old_position2, real_position); fail(e);
                } else { real_position = old_position2; position = old_position2; t.size = 32; }
            }
            t.is_signed = is_signed;
            return t;
        }
    }
    if ((c >= 'a' && c <= 'z') || c == '_' || (c >= 'A' && c <= 'Z')) {
        size_t byte_len = 1;
        for (;;) {
            uint32_t d = peek();
            if (d == NONE) break;
            if (d >= 0x80) {
                if (unicode_ws(d)) break;        // White_Space is never alphanumeric
                throw Unsupported{"non-ASCII character behind an identifier (char::is_alphanumeric tables unpinned)", real_position};
            }
            bool alnum = (d >= '0' && d <= '9') || (d >= 'a' && d <= 'z') || (d >= 'A' && d <= 'Z');
            if (alnum || d == '_' || d == '\\') { byte_len += 1; advance(); } else break;
        }
        std::string_view id = source.substr(old_position, byte_len);
        static const std::pair<const char*, Tok> kw[] = {
            {"var", Tok::Var}, {"const", Tok::Const}, {"return", Tok::Return}, {"if", Tok::If},
            {"else", Tok::Else}, {"while", Tok::While}, {"for", Tok::For}, {"break", Tok::Break},
            {"continue", Tok::Continue}, {"extern", Tok::Extern}, {"ref", Tok::Ref}, {"mut", Tok::Mut},
            {"refs", Tok::Refs}, {"muts", Tok::Muts}, {"null", Tok::Null}, {"false", Tok::False},
            {"true", Tok::True}, {"type", Tok::Type}, {"record", Tok::Record}, {"is", Tok::Is},
            {"in", Tok::In}, {"and", Tok::And}, {"or", Tok::Or}, {"not", Tok::Not},
            {"sizeof", Tok::Sizeof}, {"alignof", Tok::Alignof}, {"import", Tok::Import},
            {"library", Tok::Library}, {"proc", Tok::Proc}, {"operator", Tok::Operator}};
        for (auto& k : kw) if (id == k.first) return simple(k.second);
        Token t; t.kind = Tok::Identifier; t.ident = id; return t;
    }
    if (c == '"') {
        std::vector<uint8_t> chars;
        for (;;) {
            size_t before = real_position;
            uint32_t ch = advance();
            if (ch == '"') break;
            if (ch == '\\') {
                size_t escape_position = position;
                uint32_t esc = advance();
                switch (esc) {
                    case '\\': chars.push_back('\\'); break;
                    case '"': chars.push_back('"'); break;
                    case 'n': chars.push_back('\n'); break;
                    case 'r': chars.push_back('\r'); break;
                    case 't': chars.push_back('\t'); break;
                    case '0': chars.push_back('\0'); break;
                    case 'x': {
                        uint32_t hi = advance();
                        uint32_t lo = advance();
                        auto dig = [](uint32_t x) -> int {
                            if (x >= '0' && x <= '9') return (int)(x - '0');
                            if (x >= 'a' && x <= 'f') return (int)(x - 'a' + 10);
                            if (x >= 'A' && x <= 'F') return (int)(x - 'A' + 10);
                            return -1;
                        };
                        int h = (hi == NONE) ? -1 : dig(hi), l = (lo == NONE) ? -1 : dig(lo);
                        if (h >= 0 && l >= 0) chars.push_back((uint8_t)(h * 16 + l));
                        else { ParseError e; e.kind = PErr::InvalidEscapeSequence; e.span = sp(escape_position, position); fail(e); }
                        break;
                    }
                    case NONE: { ParseError e; e.kind = PErr::UnexpectedEndOfFile; e.span = sp(position, source.size()); fail(e); }
                    default: { ParseError e; e.kind = PErr::InvalidEscapeSequence; e.span = char_at(escape_position); fail(e); }
                }
            } else if (ch == NONE) {
                ParseError e; e.kind = PErr::UnexpectedEndOfFile; e.span = sp(position, source.size()); fail(e);
            } else {
                // parse.rs:704-711 encode_utf8 == the original bytes for valid UTF-8 input
                for (size_t i = before; i < real_position; i++) chars.push_back((uint8_t)source[i]);
            }
        }
        Token t; t.kind = Tok::String; t.bytes = std::move(chars); return t;
    }
    // parse.rs:718: any other character, ASCII or not, is an UnexpectedCharacter error spanning that character
    ParseError e; e.kind = PErr::UnexpectedCharacter; e.span = char_at(token_low); e.ch = c; fail(e);
}

// parse.rs:815-835
ExprBox Parser::parse_expression() {
    ExprBox left = parse_atom();
    Span left_span = compiler.expression_span(left.get());
    size_t low = left_span.low;
    Token operator_token = peek_token();
    if (auto op = binary_from_token(operator_token)) {
        parse_token();
        ExprBox right = parse_expression();
        ExprBox e = new_expr(EK::Operator);
        e->op = *op; e->a = std::move(left); e->b = std::move(right);
        correct_precedence(e.get());
        compiler.expression_spans[e.get()] = sp(low, position);
        return e;
    }
    return left;
}

// parse.rs:837-869
void Parser::correct_precedence(Expression* expression) {
    if (expression->kind != EK::Operator) return;
    Expression* right = expression->b.get();
    if (right->kind != EK::Operator) return;
    if (op_precedence(expression->op) >= op_precedence(right->op)) {
        Span left_span = compiler.expression_span(expression->a.get());
        size_t low = left_span.low;
        Span right_left_span = compiler.expression_span(right->a.get());
        size_t high = right_left_span.high();
        std::swap(expression->op, right->op);
        std::swap(expression->a, right->a);   // swap(left, right_left)
        std::swap(expression->a, right->b);   // swap(left, right_right)
        std::swap(expression->a, expression->b);  // swap(left, right)
        correct_precedence(expression->a.get());
        compiler.expression_spans[expression->a.get()] = sp(low, high);
    }
}

// parse.rs:871-897
template <class T, class F>
std::vector<T> Parser::parse_comma_separated(Tok close, F parse_item) {
    ignore_dents += 1;
    Token token = peek_token();
    if (token.kind == close) { try { parse_token(); } catch (ParseError&) {} ignore_dents -= 1; return {}; }
    std::vector<T> items;
    for (;;) {
        items.push_back(parse_item());
        bool saw_comma = maybe_consume(Tok::Comma);
        Token token2 = peek_token();
        if (token2.kind == close) { try { parse_token(); } catch (ParseError&) {} ignore_dents -= 1; return items; }
        else if (!saw_comma) {
            ParseError e; e.kind = PErr::UnexpectedToken; e.span = sp(token_low, real_position); e.found = token2; fail(e);
        }
    }
}

// parse.rs:899-969
TypeBox Parser::parse_type() {
    Token token = parse_token();
    size_t low = token_low;
    TypeBox typ;
    switch (token.kind) {
        case Tok::Identifier: typ = new_type(TK::Name); typ->name = token.ident; break;
        case Tok::Null: typ = new_type(TK::Null); break;
        case Tok::At: {
            Token next_token = peek_token();
            if (next_token.kind == Tok::Mut) {
                try { parse_token(); } catch (ParseError&) {}
                TypeBox sub = parse_type();
                typ = new_type(TK::MutableReference); typ->sub = std::move(sub);
            } else {
                TypeBox sub = parse_type();
                typ = new_type(TK::Reference); typ->sub = std::move(sub);
            }
            break;
        }
        case Tok::LeftParenthesis: {
            auto args = parse_comma_separated<TypeBox>(Tok::RightParenthesis, [&] { return parse_type(); });
            TypeBox ret = parse_type();
            typ = new_type(TK::Proc); typ->args = std::move(args); typ->sub = std::move(ret);
            break;
        }
        case Tok::Record: {
            expect(Tok::LeftBrace);
            std::vector<std::pair<std::string_view, TypeBox>> fields;
            ignore_dents += 1;
            for (;;) {
                Token t = peek_token();
                if (t.kind == Tok::RightBrace) { try { parse_token(); } catch (ParseError&) {} break; }
                std::string_view field_name = parse_identifier();
                TypeBox field_type = parse_type();
                fields.emplace_back(field_name, std::move(field_type));
            }
            ignore_dents -= 1;
            typ = new_type(TK::Record); typ->fields = std::move(fields);
            break;
        }
        case Tok::LeftBrace: {
            if (maybe_consume(Tok::RightBrace)) {
                bool is_mut = maybe_consume(Tok::Mut);
                TypeBox sub = parse_type();
                typ = new_type(is_mut ? TK::MutableArrayReference : TK::ArrayReference); typ->sub = std::move(sub);
            } else {
                uint64_t value = parse_integer();
                expect(Tok::RightBrace);
                TypeBox sub = parse_type();
                typ = new_type(TK::Array); typ->size = value; typ->sub = std::move(sub);
            }
            break;
        }
        default: {
            ParseError e; e.kind = PErr::UnexpectedToken; e.span = sp(low, position); e.found = token; fail(e);
        }
    }
    compiler.type_spans[typ.get()] = sp(low, position);
    return typ;
}

// parse.rs:971-1124
ExprBox Parser::parse_atom() {
    Token token = parse_token();
    size_t low = token_low;
    ExprBox expression;
    switch (token.kind) {
        case Tok::Integer: expression = new_expr(EK::Integer); expression->ival = token.ival; expression->has_meta = token.has_meta; expression->is_signed = token.is_signed; expression->size = token.size; break;
        case Tok::Float: expression = new_expr(EK::Float); expression->fval = token.fval; expression->has_meta = token.has_meta; expression->size = token.size; break;
        case Tok::String: expression = new_expr(EK::String); expression->bytes = token.bytes; break;
        case Tok::Identifier: expression = new_expr(EK::Variable); expression->name = token.ident; break;
        case Tok::LeftParenthesis: {
            ignore_dents += 1;
            ExprBox inner = parse_expression();
            expect(Tok::RightParenthesis);
            ignore_dents -= 1;
            expression = new_expr(EK::Parentheses); expression->a = std::move(inner);
            break;
        }
        case Tok::Minus: { ExprBox s = parse_atom(); expression = new_expr(EK::Negate); expression->a = std::move(s); break; }
        case Tok::Ref: { ExprBox s = parse_atom(); expression = new_expr(EK::Reference); expression->a = std::move(s); break; }
        case Tok::Mut: { ExprBox s = parse_atom(); expression = new_expr(EK::MutableReference); expression->a = std::move(s); break; }
        case Tok::Refs: { ExprBox s = parse_atom(); expression = new_expr(EK::ArrayReference); expression->a = std::move(s); break; }
        case Tok::Muts: { ExprBox s = parse_atom(); expression = new_expr(EK::MutableArrayReference); expression->a = std::move(s); break; }
        case Tok::Var: {
            std::string_view name = parse_identifier();
            TypeBox t = parse_type();
            expression = new_expr(EK::VariableDefinition); expression->name = name; expression->type = std::move(t);
            break;
        }
        case Tok::Null: expression = new_expr(EK::Null); break;
        case Tok::False: expression = new_expr(EK::Bool); expression->bval = false; break;
        case Tok::True: expression = new_expr(EK::Bool); expression->bval = true; break;
        case Tok::Not: {
            ExprBox sub = parse_expression();
            ExprBox null_box = new_expr(EK::Null);
            compiler.expression_spans[null_box.get()] = sp(low, low);
            ExprBox e = new_expr(EK::Operator);
            e->op = Op::Not; e->a = std::move(null_box); e->b = std::move(sub);
            correct_precedence(e.get());
            compiler.expression_spans[e.get()] = sp(low, position);
            return e;
        }
        case Tok::Sizeof: { TypeBox t = parse_type(); expression = new_expr(EK::Sizeof); expression->type = std::move(t); break; }
        case Tok::Alignof: { TypeBox t = parse_type(); expression = new_expr(EK::Alignof); expression->type = std::move(t); break; }
        case Tok::Record: {
            std::string_view name = parse_identifier();
            TypeBox type_box = new_type(TK::Name); type_box->name = name;
            compiler.type_spans[type_box.get()] = sp(low, position);
            expect(Tok::LeftBrace);
            ignore_dents += 1;
            std::vector<std::pair<std::string_view, ExprBox>> fields;
            for (;;) {
                Token t = parse_token();
                if (t.kind == Tok::Identifier) {
                    expect(Tok::Equals);
                    ExprBox value = parse_expression();
                    fields.emplace_back(t.ident, std::move(value));
                } else if (t.kind == Tok::RightBrace) break;
                else { ParseError e; e.kind = PErr::UnexpectedToken; e.span = sp(token_low, position); e.found = t; fail(e); }
            }
            ignore_dents -= 1;
            expression = new_expr(EK::Record); expression->type = std::move(type_box); expression->fields = std::move(fields);
            break;
        }
        case Tok::Proc: {
            std::string_view name = parse_identifier();
            expect(Tok::LeftParenthesis);
            auto types = parse_comma_separated<TypeBox>(Tok::RightParenthesis, [&] { return parse_type(); });
            expression = new_expr(EK::Proc); expression->name = name; expression->types = std::move(types);
            break;
        }
        default: {
            ParseError e; e.kind = PErr::UnexpectedToken; e.span = sp(low, position); e.found = token; fail(e);
        }
    }
    ExprBox expression_box = std::move(expression);
    compiler.expression_spans[expression_box.get()] = sp(low, position);

    for (;;) {
        Token t = peek_token();
        if (t.kind == Tok::At) {
            try { parse_token(); } catch (ParseError&) {}
            ExprBox e = new_expr(EK::Dereference); e->a = std::move(expression_box);
            expression_box = std::move(e);
            compiler.expression_spans[expression_box.get()] = sp(low, position);
        } else if (t.kind == Tok::Dot) {
            try { parse_token(); } catch (ParseError&) {}
            std::string_view field_name = parse_identifier();
            ExprBox e = new_expr(EK::Field); e->name = field_name; e->a = std::move(expression_box);
            expression_box = std::move(e);
            compiler.expression_spans[expression_box.get()] = sp(low, position);
        } else if (t.kind == Tok::LeftParenthesis) {
            try { parse_token(); } catch (ParseError&) {}
            auto args = parse_comma_separated<ExprBox>(Tok::RightParenthesis, [&] { return parse_expression(); });
            if (expression_box->kind == EK::Variable) {
                // The reference drops the Variable box here; its span-table entry stays behind.
                ExprBox e = new_expr(EK::NamedCall); e->name = expression_box->name; e->args = std::move(args);
                n_nodes--;  // the Variable node is replaced, not kept
                expression_box = std::move(e);
            } else {
                ExprBox e = new_expr(EK::CallOrIndex); e->a = std::move(expression_box); e->args = std::move(args);
                expression_box = std::move(e);
            }
            compiler.expression_spans[expression_box.get()] = sp(low, position);
        } else if (t.kind == Tok::Colon) {
            try { parse_token(); } catch (ParseError&) {}
            TypeBox typ = parse_type();
            ExprBox e = new_expr(EK::Cast); e->a = std::move(expression_box); e->type = std::move(typ);
            expression_box = std::move(e);
            compiler.expression_spans[expression_box.get()] = sp(low, position);
        } else break;
    }
    return expression_box;
}

// parse.rs:1126-1235
StmtBox Parser::parse_statement() {
    Token token = peek_token();
    size_t low = token_low;
    auto statement = std::make_unique<Statement>();
    n_nodes++;
    auto consume = [&] { try { parse_token(); } catch (ParseError&) {} };
    switch (token.kind) {
        case Tok::Var: {
            consume();
            std::string_view variable_name = parse_identifier();
            if (peek_token().kind == Tok::Equals) {
                consume();
                ExprBox value = parse_expression();
                statement->kind = SK::VariableDefinition; statement->name = variable_name; statement->a = std::move(value);
            } else {
                TypeBox typ = parse_type();
                Span left_span = sp(low, position);
                if (peek_token().kind == Tok::Equals) {
                    consume();
                    ExprBox value = parse_expression();
                    ExprBox left = new_expr(EK::VariableDefinition); left->name = variable_name; left->type = std::move(typ);
                    compiler.expression_spans[left.get()] = left_span;
                    statement->kind = SK::Assignment; statement->a = std::move(left); statement->b = std::move(value);
                } else {
                    ExprBox e = new_expr(EK::VariableDefinition); e->name = variable_name; e->type = std::move(typ);
                    compiler.expression_spans[e.get()] = left_span;
                    statement->kind = SK::Expression; statement->a = std::move(e);
                }
            }
            break;
        }
        case Tok::Return: {
            consume();
            statement->kind = SK::Return; statement->a = parse_expression();
            break;
        }
        case Tok::If: case Tok::While: {
            consume();
            ExprBox condition = parse_expression();
            Block then = parse_block();
            Block els;
            // parse.rs:1172 — `if let Ok(Token::Else) = self.peek_token()` discards Err (Q6)
            bool is_else = false;
            try { is_else = peek_token().kind == Tok::Else; } catch (ParseError&) { is_else = false; }
            if (is_else) { consume(); els = parse_block(); }
            statement->kind = token.kind == Tok::If ? SK::If : SK::While;
            statement->a = std::move(condition); statement->then_block = std::move(then); statement->else_block = std::move(els);
            break;
        }
        case Tok::For: {
            consume();
            std::string_view name = parse_identifier();
            TypeBox type_option;
            if (peek_token().kind != Tok::In) type_option = parse_type();
            expect(Tok::In);
            ExprBox iteree = parse_expression();
            Block block = parse_block();
            Block els;
            bool is_else = false;
            try { is_else = peek_token().kind == Tok::Else; } catch (ParseError&) { is_else = false; }
            if (is_else) { consume(); els = parse_block(); }
            statement->kind = SK::For; statement->name = name; statement->type = std::move(type_option);
            statement->a = std::move(iteree); statement->then_block = std::move(block); statement->else_block = std::move(els);
            break;
        }
        case Tok::Break: consume(); statement->kind = SK::Break; break;
        case Tok::Continue: consume(); statement->kind = SK::Continue; break;
        default: {
            ExprBox expression = parse_expression();
            if (peek_token().kind == Tok::Equals) {
                consume();
                ExprBox right = parse_expression();
                statement->kind = SK::Assignment; statement->a = std::move(expression); statement->b = std::move(right);
            } else { statement->kind = SK::Expression; statement->a = std::move(expression); }
        }
    }
    compiler.statement_spans[statement.get()] = sp(low, position);
    return statement;
}

// parse.rs:1238-1260
Block Parser::parse_block() {
    Block statements;
    if (peek_token().kind == Tok::Indent) { try { parse_token(); } catch (ParseError&) {} }
    else { statements.push_back(parse_statement()); return statements; }
    for (;;) {
        if (peek_token().kind == Tok::Dedent) break;
        statements.push_back(parse_statement());
    }
    try { parse_token(); } catch (ParseError&) {}
    return statements;
}

// parse.rs:1262-1337
DefBox Parser::parse_definition() {
    Token token = peek_token();
    size_t low = token_low;
    Span span;
    auto d = std::make_unique<Definition>();
    n_nodes++;
    auto consume = [&] { try { parse_token(); } catch (ParseError&) {} };
    switch (token.kind) {
        case Tok::Extern: consume(); d->sig = parse_function_signature(); span = sp(low, position); d->kind = DK::Extern; break;
        case Tok::Var: {
            consume();
            d->name = parse_identifier();
            d->type = parse_type();
            if (peek_token().kind == Tok::Equals) { consume(); d->value = parse_expression(); }
            span = sp(low, position); d->kind = DK::Variable; break;
        }
        case Tok::Const: {
            consume();
            d->name = parse_identifier();
            d->type = parse_type();
            expect(Tok::Equals);
            d->value = parse_expression();
            span = sp(low, position); d->kind = DK::Constant; break;
        }
        case Tok::Type: {
            consume();
            d->name = parse_identifier();
            expect(Tok::Equals);
            d->type = parse_type();
            span = sp(low, position); d->kind = DK::Type; break;
        }
        case Tok::Import: consume(); d->path = parse_string(); span = sp(low, position); d->kind = DK::Import; break;
        case Tok::Library: consume(); d->path = parse_string(); span = sp(low, position); d->kind = DK::Library; break;
        default: {
            d->sig = parse_function_signature();
            span = sp(low, position);
            d->body = parse_block();
            d->kind = DK::Function;
        }
    }
    compiler.definition_spans[d.get()] = span;
    return d;
}

// parse.rs:1339-1355
FunctionSignature Parser::parse_function_signature() {
    FunctionSignature s;
    s.name = parse_identifier();
    expect(Tok::LeftParenthesis);
    s.arguments = parse_comma_separated<std::pair<std::string_view, TypeBox>>(Tok::RightParenthesis, [&] {
        std::string_view name2 = parse_identifier();
        TypeBox typ = parse_type();
        return std::make_pair(name2, std::move(typ));
    });
    s.return_type = parse_type();
    return s;
}

// frontend.rs:120-138
void get_definitions(Unit& u) {
    Parser parser(*u.compiler, std::string_view(u.source));
    parser.log = u.keep_token_log ? &u.token_log : nullptr;
    try {
        while (!parser.is_at_end_of_file()) {
            u.definitions.push_back(parser.parse_definition());
        }
    } catch (ParseError& e) {
        u.has_parse_error = true;
        u.parse_error = e;
    } catch (Unsupported& e) {
        u.unsupported = true; u.unsupported_why = e.why; u.unsupported_at = e.at;
    }
    u.n_nodes = parser.n_nodes;
    u.n_tokens = parser.n_tokens;
}

// KAT-L mode: the lexer alone, parse_token called repeatedly with
// ignore_dents == 0; a lexical error is logged and lexing resumes where the failing call stopped
// (the Q6/Q7 resume rule, parse.rs:737).
void lex_only(Unit& u) {
    Parser parser(*u.compiler, std::string_view(u.source));
    parser.log = &u.token_log;
    try {
        for (size_t guard = 0; guard < 2 * u.source.size() + 16; guard++) {
            try { if (parser.parse_token().kind == Tok::EndOfFile) break; } catch (ParseError&) {}
        }
    } catch (Unsupported& e) {
        u.unsupported = true; u.unsupported_why = e.why; u.unsupported_at = e.at;
    }
}

}  // namespace oracle

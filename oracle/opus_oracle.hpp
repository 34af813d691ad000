// ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED.
//
// CPU restatement (C++17, deliberately sequential) of the hot path of the reference compiler
// ftxqxd/opus:  lex -> parse -> semantic tables -> IR generation.
//
//   * follows  src/parse.rs (all), src/compile.rs:37-103,172-198,322-775,919-1151,
//     src/generate_ir.rs (all), src/frontend.rs:120-187 (driver order).
//   * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call this code, and
//     only as the checker / the timed CPU baseline.  The product (opus_amd/) never links it.
//   * "Parity unpinned": the reference ships no tests, fixtures or golden vectors and cannot be
//     built here (no rustc / cargo / LLVM 7).  The restatement is pinned only by the hand-derived
//     known-answer tests of SURVEY.md §8c / Appendix D (tests/golden/kat_*.txt).
//
// Each function cites the reference file:line it restates.
#pragma once
#include <cstdint>
#include <cstddef>
#include <deque>
#include <memory>
#include <optional>
#include <string>
#include <string_view>
#include <unordered_map>
#include <vector>

namespace oracle {

// ------------------------------------------------------------------------------------------------
// Spans: the reference uses `&'source str` slices; we keep (low, len) byte offsets into the source.
struct Span {
    size_t low = 0;
    size_t len = 0;
    size_t high() const { return low + len; }
};

// parse.rs:5-67 — same variant order.
enum class Tok : uint8_t {
    LeftParenthesis, RightParenthesis, LeftSquareBracket, RightSquareBracket, LeftBrace, RightBrace,
    Colon, Equals, EqualsEquals, LessThan, GreaterThan, LessThanEquals, GreaterThanEquals,
    Plus, Minus, Asterisk, Slash, Percent, At, Tilde, Dot, Comma, QuestionMark,
    Integer, Float, String, Identifier, Indent, Dedent,
    Var, Const, Return, If, Else, While, For, Break, Continue, Extern, Ref, Mut, Refs, Muts,
    Null, False, True, Type, Record, Is, In, And, Or, Not, Sizeof, Alignof, Import, Library, Proc,
    Operator, EndOfFile,
    Count
};

struct Token {
    Tok kind = Tok::EndOfFile;
    uint64_t ival = 0;           // Integer value (wrapping u64, parse.rs:550-555)
    bool has_meta = false;       // Integer: Option<(bool,u8)> / Float: Option<u8>
    bool is_signed = false;
    uint8_t size = 0;
    double fval = 0.0;
    std::vector<uint8_t> bytes;  // String payload
    std::string_view ident;      // Identifier payload (slice of the source)
    bool operator==(const Token& o) const;  // #[derive(PartialEq)]
    bool operator!=(const Token& o) const { return !(*this == o); }
};

// parse.rs:160-176
enum class Op : uint8_t { Plus, Minus, Times, Divide, Modulo, Equals, LessThan, GreaterThan,
                          LessThanEquals, GreaterThanEquals, Is, And, Or, Not };

// parse.rs:352-360
enum class PErr : uint8_t { UnexpectedCharacter, UnexpectedEndOfFile, InvalidEscapeSequence,
                            UnexpectedToken, ExpectedFoundToken, ExpectedIdentifier,
                            InvalidNumericSize };
struct ParseError {
    PErr kind;
    Span span;
    uint32_t ch = 0;     // UnexpectedCharacter
    uint32_t size = 0;   // InvalidNumericSize
    Token found;         // UnexpectedToken / ExpectedFoundToken / ExpectedIdentifier
    Token expected;      // ExpectedFoundToken
};

// Thrown for inputs on which the behaviour of the reference is either a hang / panic or depends on
// an unpinned Unicode table (SURVEY §7 hard part 7).  The product reports OPUSFE_UNSUPPORTED on
// exactly the same inputs.
struct Unsupported { std::string why; size_t at; };

// ------------------------------------------------------------------------------------------------
// AST  (parse.rs:253-328).  Nodes are individually heap allocated like the reference's Boxes so
// that the CPU baseline has the same allocation-pattern class.
struct PType;
struct Expression;
struct Statement;
using TypeBox = std::unique_ptr<PType>;
using ExprBox = std::unique_ptr<Expression>;
using StmtBox = std::unique_ptr<Statement>;
using Block = std::vector<StmtBox>;

enum class TK : uint8_t { Name, Null, Reference, MutableReference, ArrayReference,
                          MutableArrayReference, Proc, Record, Array };
struct PType {
    TK kind;
    std::string_view name;                                  // Name
    TypeBox sub;                                            // pointers / Array / Proc return
    std::vector<TypeBox> args;                              // Proc
    std::vector<std::pair<std::string_view, TypeBox>> fields;  // Record
    uint64_t size = 0;                                      // Array
};

enum class EK : uint8_t { Integer, Float, Null, Bool, String, Variable, VariableDefinition, Proc,
                          CallOrIndex, NamedCall, Record, Operator, Reference, MutableReference,
                          ArrayReference, MutableArrayReference, Dereference, Field, Cast, Negate,
                          Sizeof, Alignof, Parentheses };
struct Expression {
    EK kind;
    uint64_t ival = 0; bool has_meta = false; bool is_signed = false; uint8_t size = 0;
    double fval = 0;
    bool bval = false;
    std::vector<uint8_t> bytes;
    std::string_view name;
    Op op = Op::Plus;
    ExprBox a, b;                 // children (left/right, or the single sub-expression in `a`)
    TypeBox type;
    std::vector<TypeBox> types;   // Proc
    std::vector<ExprBox> args;    // CallOrIndex / NamedCall
    std::vector<std::pair<std::string_view, ExprBox>> fields;  // Record
};

enum class SK : uint8_t { Expression, VariableDefinition, Assignment, Return, If, While, For,
                          Break, Continue };
struct Statement {
    SK kind;
    std::string_view name;
    ExprBox a, b;
    TypeBox type;        // For: Option<Type>
    Block then_block, else_block;
};

struct FunctionSignature {
    std::string_view name;
    std::vector<std::pair<std::string_view, TypeBox>> arguments;
    TypeBox return_type;
};

enum class DK : uint8_t { Function, Extern, Type, Variable, Constant, Import, Library };
struct Definition {
    DK kind;
    FunctionSignature sig;
    Block body;
    std::string_view name;
    TypeBox type;
    ExprBox value;          // Variable: optional
    std::vector<uint8_t> path;
};
using DefBox = std::unique_ptr<Definition>;

// ------------------------------------------------------------------------------------------------
// compile.rs
enum class TyK : uint8_t { Integer8, Integer16, Integer32, Integer64, Natural8, Natural16, Natural32,
                           Natural64, Float32, Float64, Offset, Size, GenericInteger, Generic, Bool,
                           Null, Pointer, String, Record, Function, Array, Error };
enum class PointerType : uint8_t { Reference, Mutable, Array, ArrayMutable };
enum class CastType : uint8_t { None, Pointer, PointerType, Integer, PointerToInteger,
                                IntegerToPointer, PointerToArray, Error };
struct Type;
using TypeId = Type*;
struct Type {
    TyK kind = TyK::Error;
    PointerType ptr = PointerType::Reference;
    TypeId sub = nullptr;                                   // Pointer / Array element
    std::vector<std::pair<std::string, TypeId>> fields;     // Record
    std::vector<TypeId> args; TypeId ret = nullptr;         // Function
    uint64_t size = 0;                                      // Array
    bool is_integral() const;
};
struct Function {
    std::string name;
    std::vector<TypeId> arguments;
    TypeId return_type;
    bool is_extern, is_builtin;
};
using FunctionId = Function*;

enum class PrimitiveType { Integer8 = 0, Integer16, Integer32, Integer64, Natural8, Natural16,
                           Natural32, Natural64, Bool, Null, Error, GenericInteger, Generic, String,
                           Size, Offset, Float32, Float64, NumberOfPrimitives };

// compile.rs:124-152 (order kept)
enum class SErr : uint8_t { ParseError, UndefinedVariable, ShadowedName, UndefinedFunction,
    NoOverloadForFunction, AmbiguousOverload, AmbiguousFunctionReference, UnexpectedType, InvalidType,
    InvalidOperandType, InvalidOperandTypes, FunctionMightNotReturn, BreakOutsideLoop,
    ContinueOutsideLoop, InvalidLvalue, ImmutableLvalue, ArrayPointerToNonArray, InvalidCast,
    UndefinedType, FieldAccessOnNonRecord, FieldDoesNotExist, CallOfNonFunction,
    ReferenceToBuiltinFunction, AttemptToCallOrIndexConstant, InvalidNumberOfArgs,
    InvalidConstantExpression };
// `message`: the line Compiler::print_error writes behind "Error: " (compile.rs:816-915), rendered WHEN the error is reported, as the
// reference's report_error does (compile.rs:777-780 prints at once: TypePrinter sees the types as they are at that moment)
struct SemError { SErr kind; Span span; std::string message; };

enum class ValK : uint8_t { None, Integer, Error };
struct Value { ValK kind = ValK::None; int64_t i = 0; };

struct Compiler {
    const char* src_base = nullptr;   // for span arithmetic on string_views
    size_t src_len = 0;
    std::unordered_map<std::string_view, std::vector<FunctionId>> name_resolution_map;
    std::unordered_map<const FunctionSignature*, FunctionId> signature_resolution_map;
    std::deque<Function> function_arena;
    std::unordered_map<std::string_view, size_t> global_resolution_map;
    std::vector<std::pair<TypeId, Value>> globals;
    std::vector<bool> global_is_constant;
    mutable bool has_errors = false;
    mutable std::vector<SemError> errors;
    std::unordered_map<const Expression*, Span> expression_spans;
    std::unordered_map<const PType*, Span> type_spans;
    std::unordered_map<const Statement*, Span> statement_spans;
    std::unordered_map<const Definition*, Span> definition_spans;
    TypeId primitive_types[(int)PrimitiveType::NumberOfPrimitives];
    mutable std::deque<Type> type_arena;
    std::unordered_map<std::string_view, TypeId> type_resolution_map;
    std::unordered_map<const Type*, std::string_view> reverse_type_resolution_map;

    Compiler();
    Span sv_span(std::string_view s) const { return Span{(size_t)(s.data() - src_base), s.size()}; }
    FunctionId new_function_id(Function f) { function_arena.push_back(std::move(f)); return &function_arena.back(); }
    TypeId new_type_id(Type t) const { type_arena.push_back(std::move(t)); return &type_arena.back(); }
    TypeId type_primitive(PrimitiveType p) const { return primitive_types[(int)p]; }
    TypeId type_error() const { return type_primitive(PrimitiveType::Error); }
    TypeId type_bool() const { return type_primitive(PrimitiveType::Bool); }
    TypeId type_null() const { return type_primitive(PrimitiveType::Null); }
    TypeId type_string() const { return type_primitive(PrimitiveType::String); }
    TypeId type_pointer(PointerType p, TypeId o) const { Type t; t.kind = TyK::Pointer; t.ptr = p; t.sub = o; return new_type_id(t); }
    TypeId type_ref(TypeId o) const { return type_pointer(PointerType::Reference, o); }
    TypeId type_mut(TypeId o) const { return type_pointer(PointerType::Mutable, o); }
    TypeId type_refs(TypeId o) const { return type_pointer(PointerType::Array, o); }
    TypeId type_muts(TypeId o) const { return type_pointer(PointerType::ArrayMutable, o); }
    TypeId type_array(uint64_t n, TypeId o) const { Type t; t.kind = TyK::Array; t.size = n; t.sub = o; return new_type_id(t); }
    TypeId type_function(std::vector<TypeId> a, TypeId r) const { Type t; t.kind = TyK::Function; t.args = std::move(a); t.ret = r; return new_type_id(t); }
    TypeId type_deref(TypeId t) const;
    bool is_error_type(TypeId t) const { return t->kind == TyK::Error; }
    std::optional<CastType> can_autocast(TypeId from, TypeId to) const;
    bool can_autocast_pointer_type(PointerType from, PointerType to) const;
    std::optional<CastType> can_cast(TypeId from, TypeId to) const;
    bool types_match(TypeId a, TypeId b) const;
    bool types_match_allow_generic(TypeId a, TypeId b) const;
    bool types_match_default_generic(TypeId a, TypeId b) const;
    // compile.rs:777-780 report_error; one overload per payload shape of compile.rs:124-152
    void report(SErr kind, Span span, std::string message) const { errors.push_back({kind, span, std::move(message)}); has_errors = true; }
    std::string span_text(Span s) const { return std::string(src_base + s.low, s.len); }
    void report_error(SErr kind, Span span) const;                         // the kinds whose message prints at most the span text
    void report_error(SErr kind, Span span, TypeId a) const;               // InvalidType :844, InvalidOperandType :848, FieldAccessOnNonRecord :888, CallOfNonFunction :896
    void report_error(SErr kind, Span span, TypeId a, TypeId b) const;     // UnexpectedType {expected, found} :840, InvalidCast {from, to} :880
    void report_operand_types(Span span, Op op, TypeId left, TypeId right) const;                                 // :852
    void report_shadowed(Span span, std::string_view name) const;                                                // :820
    void report_function(SErr kind, Span span, std::string_view name, const std::vector<TypeId>& args) const;   // :824-835, FunctionIdentifierPrinter :300-320
    void report_field_missing(Span span, TypeId typ, std::string_view field) const;                              // :892
    TypeId resolve_type(const PType& t) const;
    void preload_definition(const Definition& d);
    Value make_value(const Expression& e);
    void finalize_definition(const Definition& d);
    FunctionId lookup_function(Span span, std::string_view name, const std::vector<TypeId>& argument_types) const;
    Span statement_span(const Statement* s) const { return statement_spans.at(s); }
    Span expression_span(const Expression* e) const { return expression_spans.at(e); }
    Span type_span(const PType* t) const { return type_spans.at(t); }
    std::string type_to_string(TypeId t) const;       // TypePrinter  compile.rs:200-274
    std::string function_to_string(FunctionId f) const;  // FunctionPrinter compile.rs:276-298
};

// ------------------------------------------------------------------------------------------------
// generate_ir.rs
enum class IK : uint8_t { Integer, Float, Null, Bool, String, Sizeof, Alignof, Call, Allocate, Load,
    Store, Field, BuiltinField, IndexPointer, IndexArray, Add, Subtract, Multiply, Divide, Modulo,
    Equals, LessThan, GreaterThan, LessThanEquals, GreaterThanEquals, Function, GlobalVariable,
    GlobalConstant, Not, Negate, Cast, Return, Jump, Branch, Unreachable, Phi, BreakPlaceholder,
    Error };
using VariableId = size_t;
struct Instruction {
    IK kind;
    VariableId a = 0, b = 0, c = 0;      // operands in declaration order
    size_t i1 = 0, i2 = 0;               // bb indices / global id / field index
    int64_t ival = 0; double fval = 0; bool bval = false;
    std::vector<uint8_t> bytes;
    std::vector<VariableId> args;
    TypeId typ = nullptr;
    FunctionId func = nullptr;
    std::string_view field;
    CastType cast = CastType::None;
};
struct Variable { TypeId typ; bool is_temporary; };

// One parsed + lowered translation unit (what frontend.rs:23-187 builds for one file).
struct TokenLogEntry { size_t low, high; bool is_error; Token token; ParseError error; };

struct Unit {
    std::string source;   // owned copy
    std::unique_ptr<Compiler> compiler;
    std::vector<DefBox> definitions;
    std::vector<TokenLogEntry> token_log;     // every fresh lex in reference order
    bool keep_token_log = true;               // false for timed runs (the log is dump infrastructure, not the path)
    bool has_parse_error = false;
    ParseError parse_error;
    bool unsupported = false;
    std::string unsupported_why;
    size_t unsupported_at = 0;
    // IR
    struct FunctionIr {
        FunctionId function;
        std::vector<Variable> variables;
        std::vector<std::vector<Instruction>> bbs;
    };
    std::vector<FunctionIr> ir;
    bool ir_generated = false;   // frontend.rs:145 — false when finalize reported errors
    // counts for the roofline formula
    uint64_t n_tokens = 0, n_nodes = 0, n_instrs = 0, n_vars = 0;
};

// frontend.rs:120-138 (get_definitions) + 140-187 (compile_source, IR part)
std::unique_ptr<Unit> run(const char* src, size_t len, bool lower, bool keep_token_log = true);

std::string dump_tokens(const Unit& u);
std::string dump_ast(const Unit& u);
std::string dump_ir(const Unit& u);
std::string dump_error(const Unit& u);
std::string dump_semantic_errors(const Unit& u);
// what the reference writes to stderr for this unit's errors: compile.rs:782-917 print_error + frontend.rs:201-256 print_span
// (the parse error, else every semantic error in report order); `filename` is what compiler.filenames[i] would print
std::string dump_rendered_errors(const Unit& u, const std::string& filename);

}  // namespace oracle

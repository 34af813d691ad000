// ORACLE — TEST INFRASTRUCTURE ONLY (see opus_oracle.hpp).  Restates reference src/compile.rs.
#include "opus_oracle.hpp"
#include "oracle_internal.hpp"

namespace oracle {

// compile.rs:172-180
bool Type::is_integral() const {
    switch (kind) {
        case TyK::Natural8: case TyK::Natural16: case TyK::Natural32: case TyK::Natural64: case TyK::Size:
        case TyK::Integer8: case TyK::Integer16: case TyK::Integer32: case TyK::Integer64: case TyK::Offset:
            return true;
        default: return false;
    }
}

// compile.rs:354-503
Compiler::Compiler() {
    static const TyK prim_kinds[] = {TyK::Integer8, TyK::Integer16, TyK::Integer32, TyK::Integer64,
        TyK::Natural8, TyK::Natural16, TyK::Natural32, TyK::Natural64, TyK::Bool, TyK::Null, TyK::Error,
        TyK::GenericInteger, TyK::Generic, TyK::String, TyK::Size, TyK::Offset, TyK::Float32, TyK::Float64};
    for (int i = 0; i < (int)PrimitiveType::NumberOfPrimitives; i++) {
        Type t; t.kind = prim_kinds[i];
        primitive_types[i] = new_type_id(t);
    }
    static const PrimitiveType arith[] = {PrimitiveType::Integer8, PrimitiveType::Integer16, PrimitiveType::Integer32,
        PrimitiveType::Integer64, PrimitiveType::Natural8, PrimitiveType::Natural16, PrimitiveType::Natural32,
        PrimitiveType::Natural64, PrimitiveType::Size, PrimitiveType::Offset, PrimitiveType::Float32, PrimitiveType::Float64};
    for (Op op : {Op::Plus, Op::Minus, Op::Times, Op::Divide, Op::Modulo}) {
        std::vector<FunctionId> overloads;
        const char* name = op_symbol(op);
        for (PrimitiveType p : arith) {
            TypeId typ = type_primitive(p);
            overloads.push_back(new_function_id(Function{name, {typ, typ}, typ, false, true}));
        }
        name_resolution_map[std::string_view(name)] = overloads;
    }
    static const PrimitiveType cmp[] = {PrimitiveType::Integer8, PrimitiveType::Integer16, PrimitiveType::Integer32,
        PrimitiveType::Integer64, PrimitiveType::Natural8, PrimitiveType::Natural16, PrimitiveType::Natural32,
        PrimitiveType::Natural64, PrimitiveType::Size, PrimitiveType::Offset, PrimitiveType::Bool, PrimitiveType::Null,
        PrimitiveType::Float32, PrimitiveType::Float64};
    for (Op op : {Op::Equals, Op::LessThan, Op::GreaterThan, Op::LessThanEquals, Op::GreaterThanEquals}) {
        std::vector<FunctionId> overloads;
        const char* name = op_symbol(op);
        for (PrimitiveType p : cmp) {
            TypeId typ = type_primitive(p);
            overloads.push_back(new_function_id(Function{name, {typ, typ}, type_bool(), false, true}));
        }
        name_resolution_map[std::string_view(name)] = overloads;
    }
    {
        const char* name = op_symbol(Op::Not);
        TypeId typ = type_primitive(PrimitiveType::Bool);
        name_resolution_map[std::string_view(name)] = {new_function_id(Function{name, {typ}, type_bool(), false, true})};
    }
}

// compile.rs:591-597
TypeId Compiler::type_deref(TypeId t) const {
    if (t->kind == TyK::Pointer) return t->sub;
    if (t->kind == TyK::Error) return type_error();
    throw Unsupported{"type_deref on non-pointer (reference: unreachable!())", 0};
}

// compile.rs:624-670
std::optional<CastType> Compiler::can_autocast(TypeId from, TypeId to) const {
    if (types_match(from, to)) return CastType::None;
    const Type* type1 = from; const Type* type2 = to;
    if (type1->kind == TyK::Error || type2->kind == TyK::Error) return CastType::Error;
    if (type1->kind == TyK::GenericInteger && type2->is_integral()) return CastType::Error;
    if (type1->kind == TyK::Generic) return CastType::Error;
    if (type1->kind == TyK::Pointer && type2->kind == TyK::Pointer) {
        PointerType p1 = type1->ptr, p2 = type2->ptr;
        TypeId typ1 = type1->sub, typ2 = type2->sub;
        if (can_autocast_pointer_type(p1, p2) && types_match(typ1, typ2)) return CastType::PointerType;
        TypeId nat8 = type_primitive(PrimitiveType::Natural8);
        if (((p1 == PointerType::Mutable && p2 == PointerType::ArrayMutable)
             || (p1 == PointerType::Mutable && p2 == PointerType::Array)
             || (p1 == PointerType::Reference && p2 == PointerType::Array))
            && types_match(typ2, nat8)) return CastType::PointerType;
        if (((p1 == PointerType::ArrayMutable && p2 == PointerType::Mutable)
             || (p1 == PointerType::ArrayMutable && p2 == PointerType::Reference)
             || (p1 == PointerType::Array && p2 == PointerType::Reference))
            && types_match(typ1, nat8)) return CastType::PointerType;
        if (typ1->kind == TyK::Array && types_match(typ1->sub, typ2)) {
            if ((p1 == PointerType::Reference && p2 == PointerType::Array)
                || (p1 == PointerType::Mutable && p2 == PointerType::ArrayMutable)
                || (p1 == PointerType::Mutable && p2 == PointerType::Array)) return CastType::PointerToArray;
            return std::nullopt;
        }
        return std::nullopt;
    }
    return std::nullopt;
}

// compile.rs:672-680
bool Compiler::can_autocast_pointer_type(PointerType from, PointerType to) const {
    if (from == to) return true;
    if (from == PointerType::Mutable && to == PointerType::Reference) return true;
    if (from == PointerType::ArrayMutable && to == PointerType::Array) return true;
    return false;
}

// compile.rs:682-696
std::optional<CastType> Compiler::can_cast(TypeId from, TypeId to) const {
    if (auto c = can_autocast(from, to)) return c;
    const Type* type1 = from; const Type* type2 = to;
    if (type1->is_integral() && type2->is_integral()) return CastType::Integer;
    if (type1->kind == TyK::Pointer && type2->kind == TyK::Pointer && type1->ptr == type2->ptr) return CastType::Pointer;
    if ((type1->kind == TyK::Pointer || type1->kind == TyK::Function) && type2->is_integral()) return CastType::PointerToInteger;
    if ((type2->kind == TyK::Pointer || type2->kind == TyK::Function) && type1->is_integral()) return CastType::IntegerToPointer;
    return std::nullopt;
}

// compile.rs:698-751
bool Compiler::types_match(TypeId a, TypeId b) const {
    if (a->kind != b->kind) return false;
    switch (a->kind) {
        case TyK::Pointer: return a->ptr == b->ptr && types_match(a->sub, b->sub);
        case TyK::Record: {
            if (a->fields.size() != b->fields.size()) return false;
            for (size_t i = 0; i < a->fields.size(); i++)
                if (a->fields[i].first != b->fields[i].first || !types_match(a->fields[i].second, b->fields[i].second)) return false;
            return true;
        }
        case TyK::Function: {
            if (a->args.size() != b->args.size()) return false;
            if (!types_match(a->ret, b->ret)) return false;
            for (size_t i = 0; i < a->args.size(); i++) if (!types_match(a->args[i], b->args[i])) return false;
            return true;
        }
        case TyK::Array: return a->size == b->size && types_match(a->sub, b->sub);
        default: return true;   // all payload-free kinds, including (Error, Error)
    }
}

// compile.rs:753-763
bool Compiler::types_match_allow_generic(TypeId a, TypeId b) const {
    if ((a->kind == TyK::GenericInteger && b->is_integral()) || (b->kind == TyK::GenericInteger && a->is_integral())) return true;
    if (a->kind == TyK::Generic || b->kind == TyK::Generic) return true;
    return types_match(a, b);
}

// compile.rs:765-775
bool Compiler::types_match_default_generic(TypeId a, TypeId b) const {
    if ((a->kind == TyK::GenericInteger && b->kind == TyK::Integer32) || (a->kind == TyK::Integer32 && b->kind == TyK::GenericInteger)) return true;
    if (a->kind == TyK::Generic || b->kind == TyK::Generic) return false;
    return types_match(a, b);
}

// compile.rs:919-966
TypeId Compiler::resolve_type(const PType& typ) const {
    Span span = type_span(&typ);
    switch (typ.kind) {
        case TK::Name: {
            static const std::pair<const char*, PrimitiveType> names[] = {
                {"int8", PrimitiveType::Integer8}, {"int16", PrimitiveType::Integer16}, {"int32", PrimitiveType::Integer32},
                {"int64", PrimitiveType::Integer64}, {"nat8", PrimitiveType::Natural8}, {"nat16", PrimitiveType::Natural16},
                {"nat32", PrimitiveType::Natural32}, {"nat64", PrimitiveType::Natural64}, {"size", PrimitiveType::Size},
                {"offset", PrimitiveType::Offset}, {"float32", PrimitiveType::Float32}, {"float64", PrimitiveType::Float64},
                {"bool", PrimitiveType::Bool}, {"string", PrimitiveType::String}};
            for (auto& n : names) if (typ.name == n.first) return type_primitive(n.second);
            auto it = type_resolution_map.find(typ.name);
            if (it != type_resolution_map.end()) return it->second;
            report_error(SErr::UndefinedType, span);
            return type_error();
        }
        case TK::Null: return type_null();
        case TK::Reference: return type_ref(resolve_type(*typ.sub));
        case TK::MutableReference: return type_mut(resolve_type(*typ.sub));
        case TK::ArrayReference: return type_refs(resolve_type(*typ.sub));
        case TK::MutableArrayReference: return type_muts(resolve_type(*typ.sub));
        case TK::Proc: {
            std::vector<TypeId> args;
            for (auto& a : typ.args) args.push_back(resolve_type(*a));
            TypeId ret = resolve_type(*typ.sub);
            return type_function(std::move(args), ret);
        }
        case TK::Record: {
            Type t; t.kind = TyK::Record;
            for (auto& f : typ.fields) t.fields.emplace_back(std::string(f.first), resolve_type(*f.second));
            return new_type_id(std::move(t));
        }
        case TK::Array: return type_array(typ.size, resolve_type(*typ.sub));
    }
    return type_error();
}

// compile.rs:968-981
void Compiler::preload_definition(const Definition& d) {
    if (d.kind == DK::Type) {
        Type t; t.kind = TyK::Error;
        TypeId id = new_type_id(t);
        type_resolution_map[d.name] = id;                 // insert() overwrites
        reverse_type_resolution_map[id] = d.name;
    }
}

// compile.rs:983-1000
Value Compiler::make_value(const Expression& e) {
    Span span = expression_span(&e);
    if (e.kind == EK::Integer) return Value{ValK::Integer, (int64_t)e.ival};
    if (e.kind == EK::Negate && e.a->kind == EK::Integer) return Value{ValK::Integer, (int64_t)(0 - e.a->ival)};
    report_error(SErr::InvalidConstantExpression, span);
    return Value{ValK::Error, 0};
}

// compile.rs:154-170
static bool type_is_valid(const Value& v, const Type* t) {
    if (v.kind == ValK::Error) return true;
    if (t->is_integral()) return v.kind == ValK::Integer;
    return false;
}

// compile.rs:1002-1065
void Compiler::finalize_definition(const Definition& d) {
    switch (d.kind) {
        case DK::Function: case DK::Extern: {
            std::vector<TypeId> arguments;
            for (auto& a : d.sig.arguments) arguments.push_back(resolve_type(*a.second));
            TypeId return_type = resolve_type(*d.sig.return_type);
            FunctionId id = new_function_id(Function{std::string(d.sig.name), arguments, return_type, d.kind == DK::Extern, false});
            name_resolution_map[d.sig.name].push_back(id);
            signature_resolution_map[&d.sig] = id;
            break;
        }
        case DK::Variable: {
            TypeId typ = resolve_type(*d.type);
            Value value;
            if (d.value) {
                Span span = expression_span(d.value.get());
                value = make_value(*d.value);
                if (!type_is_valid(value, typ)) { report_error(SErr::InvalidType, span, typ); value = Value{ValK::Error, 0}; }
            }
            size_t id = globals.size();
            globals.emplace_back(typ, value); global_is_constant.push_back(false);
            global_resolution_map[d.name] = id;
            break;
        }
        case DK::Constant: {
            TypeId typ = resolve_type(*d.type);
            Span span = expression_span(d.value.get());
            Value value = make_value(*d.value);
            if (!type_is_valid(value, typ)) { report_error(SErr::InvalidType, span, typ); value = Value{ValK::Error, 0}; }
            size_t id = globals.size();
            globals.emplace_back(typ, value); global_is_constant.push_back(true);
            global_resolution_map[d.name] = id;
            break;
        }
        case DK::Type: {
            TypeId type_id = resolve_type(*d.type);
            Type copy = *type_id;
            *type_resolution_map.at(d.name) = copy;       // Q14: copies the *current* contents
            break;
        }
        case DK::Import: case DK::Library: break;
    }
}

// ---- report_error + the message lines of print_error (compile.rs:777-780, 816-915) ----
void Compiler::report_error(SErr kind, Span span) const {
    std::string m;
    switch (kind) {
        case SErr::UndefinedVariable: m = "undefined variable: " + span_text(span); break;                       // :816
        case SErr::AmbiguousFunctionReference: m = "reference to overloaded function is ambiguous"; break;      // :836
        case SErr::FunctionMightNotReturn: m = "function might not return"; break;                              // :856
        case SErr::BreakOutsideLoop: m = "break outside loop"; break;                                           // :860
        case SErr::ContinueOutsideLoop: m = "continue outside loop"; break;                                     // :864
        case SErr::InvalidLvalue: m = "invalid lvalue"; break;                                                  // :868
        case SErr::ImmutableLvalue: m = "lvalue is immutable"; break;                                           // :872
        case SErr::ArrayPointerToNonArray: m = "attempt to take array reference to non-array lvalue"; break;    // :876
        case SErr::UndefinedType: m = "undefined type: " + span_text(span); break;                              // :884
        case SErr::ReferenceToBuiltinFunction: m = "refernce to built-in function"; break;                      // :900 (sic)
        case SErr::AttemptToCallOrIndexConstant: m = "attempt to call or index constant"; break;                // :904
        case SErr::InvalidNumberOfArgs: m = "invalid number of indices"; break;                                 // :908
        case SErr::InvalidConstantExpression: m = "invalid constant expression"; break;                         // :912
        default: m = "<payload missing>"; break;      // a typed kind reported without its payload: a bug of the restatement, visible in every comparison
    }
    report(kind, span, std::move(m));
}
void Compiler::report_error(SErr kind, Span span, TypeId a) const {
    std::string m;
    switch (kind) {
        case SErr::InvalidType: m = "invalid type: " + type_to_string(a); break;                                  // :844
        case SErr::InvalidOperandType: m = "invalid operand type: " + type_to_string(a); break;                   // :848
        case SErr::FieldAccessOnNonRecord: m = "attempt to set or access field of non-record: " + type_to_string(a); break;   // :888
        case SErr::CallOfNonFunction: m = "cannot call value of type " + type_to_string(a); break;                // :896
        default: m = "<payload mismatch>"; break;
    }
    report(kind, span, std::move(m));
}
void Compiler::report_error(SErr kind, Span span, TypeId a, TypeId b) const {
    std::string m;
    switch (kind) {
        case SErr::UnexpectedType: m = "invalid type: expected " + type_to_string(a) + ", found " + type_to_string(b); break;   // :840
        case SErr::InvalidCast: m = "invalid cast: " + type_to_string(a) + " to " + type_to_string(b); break;                   // :880
        default: m = "<payload mismatch>"; break;
    }
    report(kind, span, std::move(m));
}
void Compiler::report_operand_types(Span span, Op op, TypeId left, TypeId right) const {                        // :852
    report(SErr::InvalidOperandTypes, span, "invalid operand types: " + type_to_string(left) + " " + op_symbol(op) + " " + type_to_string(right));
}
void Compiler::report_shadowed(Span span, std::string_view name) const {                                       // :820
    report(SErr::ShadowedName, span, "shadowed variable: " + std::string(name));
}
void Compiler::report_function(SErr kind, Span span, std::string_view name, const std::vector<TypeId>& args) const {
    // FunctionIdentifierPrinter compile.rs:300-320: name[ :T :T ... ]
    std::string id = std::string(name) + "[";
    for (size_t i = 0; i < args.size(); i++) { if (i) id += " "; id += ":" + type_to_string(args[i]); }
    id += "]";
    const char* head = kind == SErr::UndefinedFunction ? "undefined function: "                                   // :824
                     : kind == SErr::NoOverloadForFunction ? "no matching overload for function: "               // :828
                     : "call to overloaded function is ambiguous: ";                                             // :832
    report(kind, span, head + id);
}
void Compiler::report_field_missing(Span span, TypeId typ, std::string_view field) const {                     // :892
    report(SErr::FieldDoesNotExist, span, "type " + type_to_string(typ) + " has no field named '" + std::string(field) + "'");
}

// compile.rs:1083-1139
FunctionId Compiler::lookup_function(Span span, std::string_view name, const std::vector<TypeId>& argument_types) const {
    auto it = name_resolution_map.find(name);
    if (it == name_resolution_map.end()) { report_function(SErr::UndefinedFunction, span, name, argument_types); return nullptr; }
    std::vector<FunctionId> candidates;
    for (FunctionId f : it->second) {
        if (f->arguments.size() != argument_types.size()) continue;
        bool all = true;
        for (size_t i = 0; i < argument_types.size(); i++)
            if (!can_autocast(argument_types[i], f->arguments[i]).has_value()) { all = false; break; }
        if (all) candidates.push_back(f);
    }
    if (candidates.empty()) { report_function(SErr::NoOverloadForFunction, span, name, argument_types); return nullptr; }
    FunctionId candidate1 = candidates[0];
    std::vector<bool> ignorable(argument_types.size(), true);
    for (FunctionId c : candidates)
        for (size_t i = 0; i < ignorable.size(); i++)
            ignorable[i] = ignorable[i] && types_match_allow_generic(c->arguments[i], candidate1->arguments[i]);
    for (FunctionId f : candidates) {
        bool all = true;
        for (size_t i = 0; i < argument_types.size(); i++) {
            if (ignorable[i]) continue;
            if (!types_match_default_generic(f->arguments[i], argument_types[i])) { all = false; break; }
        }
        if (all) return f;
    }
    report_function(SErr::AmbiguousOverload, span, name, argument_types);
    return nullptr;
}

// compile.rs:200-274
std::string Compiler::type_to_string(TypeId t) const {
    auto rit = reverse_type_resolution_map.find(t);
    if (rit != reverse_type_resolution_map.end()) return std::string(rit->second);
    switch (t->kind) {
        case TyK::Integer8: return "int8"; case TyK::Integer16: return "int16"; case TyK::Integer32: return "int32";
        case TyK::Integer64: return "int64"; case TyK::Natural8: return "nat8"; case TyK::Natural16: return "nat16";
        case TyK::Natural32: return "nat32"; case TyK::Natural64: return "nat64"; case TyK::Size: return "size";
        case TyK::Offset: return "offset"; case TyK::Float32: return "float32"; case TyK::Float64: return "float64";
        case TyK::GenericInteger: return "<integer>"; case TyK::Generic: return "<unknown>"; case TyK::Bool: return "bool";
        case TyK::Null: return "null"; case TyK::String: return "string"; case TyK::Error: return "<error>";
        case TyK::Pointer:
            switch (t->ptr) {
                case PointerType::Reference: return "@" + type_to_string(t->sub);
                case PointerType::Mutable: return "@mut " + type_to_string(t->sub);
                case PointerType::Array: return "{} " + type_to_string(t->sub);
                case PointerType::ArrayMutable: return "mut {} " + type_to_string(t->sub);
            }
            return "";
        case TyK::Record: {
            std::string s = "record {";
            for (auto& f : t->fields) s += " " + f.first + ": " + type_to_string(f.second);
            s += " }";
            return s;
        }
        case TyK::Function: {
            std::string s = "[";
            bool any = false;
            for (TypeId a : t->args) { if (any) s += " "; any = true; s += type_to_string(a); }
            s += "] " + type_to_string(t->ret);
            return s;
        }
        case TyK::Array: return "{" + std::to_string(t->size) + "} " + type_to_string(t->sub);
    }
    return "";
}

// compile.rs:276-298
std::string Compiler::function_to_string(FunctionId f) const {
    std::string s = f->name + "[";
    bool any = false;
    for (TypeId a : f->arguments) { if (any) s += " "; s += ":" + type_to_string(a); any = true; }
    s += "]: " + type_to_string(f->return_type);
    return s;
}

}  // namespace oracle

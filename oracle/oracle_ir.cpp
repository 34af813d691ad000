// ORACLE — TEST INFRASTRUCTURE ONLY (see opus_oracle.hpp).  Restates reference src/generate_ir.rs.
#include "opus_oracle.hpp"
#include "oracle_internal.hpp"
#include <algorithm>

namespace oracle {

namespace {

constexpr VariableId NOVAR = ~(VariableId)0;

// generate_ir.rs:75-97
struct IrGenerator {
    const Compiler& compiler;
    std::unordered_map<std::string_view, VariableId> locals;
    std::vector<Variable> variables;
    std::vector<std::vector<Instruction>> bbs;
    const FunctionSignature& signature;
    Function* function;
    size_t bb_index = 0;
    Span span;
    std::vector<std::pair<size_t, size_t>> break_instructions_to_insert;
    std::optional<size_t> innermost_loop_begin;
    std::vector<std::string_view> locals_stack;

    IrGenerator(const Compiler& c, const FunctionSignature& sig, Span sp)
        : compiler(c), signature(sig), function(c.signature_resolution_map.at(&sig)), span(sp) { bbs.emplace_back(); }

    // :118-121
    VariableId new_variable(TypeId typ, bool is_temporary) { variables.push_back({typ, is_temporary}); return variables.size() - 1; }
    // :123-125
    std::vector<Instruction>& instructions() { return bbs[bb_index]; }
    void push(Instruction i) { instructions().push_back(std::move(i)); }
    static Instruction ins(IK k, VariableId a = 0, VariableId b = 0, VariableId c = 0) { Instruction i; i.kind = k; i.a = a; i.b = b; i.c = c; return i; }
    // :127-131
    VariableId generate_error() { VariableId v = new_variable(compiler.type_error(), true); push(ins(IK::Error, v)); return v; }
    bool contains_local(std::string_view n) const { return std::find(locals_stack.begin(), locals_stack.end(), n) != locals_stack.end(); }

    // :133-174
    void generate_ir_from_function(const Block& block) {
        std::vector<TypeId> types;
        for (auto& a : signature.arguments) { TypeId t = compiler.resolve_type(*a.second); types.push_back(t); new_variable(t, true); }
        for (size_t i = 0; i < signature.arguments.size(); i++) {
            TypeId typ = compiler.type_mut(types[i]);
            VariableId v = new_variable(typ, false);
            push(ins(IK::Allocate, v));
            push(ins(IK::Store, v, i));
            locals[signature.arguments[i].first] = v;
        }
        bool returns_null = function->return_type->kind == TyK::Null;
        bool diverges = generate_ir_from_block(block);
        if (diverges || returns_null) {
            auto& last_bb = bbs[bb_index];
            bool term = !last_bb.empty() && is_terminator(last_bb.back());
            if (!term) {
                if (returns_null) {
                    VariableId v = new_variable(compiler.type_null(), true);
                    push(ins(IK::Null, v));
                    push(ins(IK::Return, v));
                } else last_bb.push_back(ins(IK::Unreachable));
            }
        } else compiler.report_error(SErr::FunctionMightNotReturn, span);
    }
    // :59-64
    static bool is_terminator(const Instruction& i) { return i.kind == IK::Return || i.kind == IK::Jump || i.kind == IK::Branch || i.kind == IK::Unreachable; }

    size_t begin_scope() { return locals_stack.size(); }   // :176-178
    void end_scope(size_t n) { while (locals_stack.size() > n) { auto name = locals_stack.back(); locals_stack.pop_back(); locals.erase(name); } }  // :180-185

    // :189-203
    bool generate_ir_from_block(const Block& block) {
        bool diverges = false;
        size_t scope = begin_scope();
        for (auto& s : block) if (generate_ir_from_statement(*s)) diverges = true;
        end_scope(scope);
        return diverges;
    }

    // :207-231
    VariableId autocast_variable_to_type(VariableId variable, TypeId expected_type, Span sp) {
        TypeId found_type = variables[variable].typ;
        if (compiler.types_match(found_type, expected_type)) return variable;
        if (auto cast_type = compiler.can_autocast(found_type, expected_type)) {
            if (found_type->kind == TyK::GenericInteger || found_type->kind == TyK::Generic) {
                variables[variable].typ = expected_type;
                return variable;
            }
            VariableId r = new_variable(expected_type, true);
            Instruction i = ins(IK::Cast, r, variable); i.cast = *cast_type; push(i);
            return r;
        }
        compiler.report_error(SErr::UnexpectedType, sp, expected_type, found_type);
        return generate_error();
    }

    // :235-242
    TypeId get_lvalue_type(VariableId v) const {
        TypeId t = variables[v].typ;
        if (t->kind == TyK::Pointer) return t->sub;
        if (t->kind == TyK::Error) return compiler.type_error();
        throw Unsupported{"lvalue of non-pointer type (reference: unreachable!())", span.low};
    }

    // :246-493
    bool generate_ir_from_statement(const Statement& st) {
        Span sspan = compiler.statement_span(&st);
        switch (st.kind) {
            case SK::Expression: generate_ir_from_expression(*st.a, nullptr); break;
            case SK::VariableDefinition: {
                VariableId v = generate_ir_from_expression(*st.a, nullptr);
                TypeId typ = variables[v].typ;
                if (contains_local(st.name)) compiler.report_shadowed(sspan, st.name);
                else {
                    VariableId nv = new_variable(compiler.type_mut(typ), false);
                    push(ins(IK::Allocate, nv));
                    push(ins(IK::Store, nv, v));
                    locals_stack.push_back(st.name);
                    locals[st.name] = nv;
                }
                break;
            }
            case SK::Assignment: {
                VariableId left = generate_pointer_from_expression(*st.a, PointerType::Mutable);
                TypeId left_type = get_lvalue_type(left);
                VariableId right = generate_ir_from_expression(*st.b, left_type);
                Span sp = compiler.expression_span(st.b.get());
                VariableId fin = autocast_variable_to_type(right, left_type, sp);
                push(ins(IK::Store, left, fin));
                break;
            }
            case SK::Return: {
                VariableId rv = generate_ir_from_expression(*st.a, function->return_type);
                Span sp = compiler.expression_span(st.a.get());
                rv = autocast_variable_to_type(rv, function->return_type, sp);
                push(ins(IK::Return, rv));
                return true;
            }
            case SK::If: {
                VariableId cond = generate_ir_from_expression(*st.a, compiler.type_bool());
                Span csp = compiler.expression_span(st.a.get());
                cond = autocast_variable_to_type(cond, compiler.type_bool(), csp);
                size_t branch = bb_index;
                bool diverges = true;
                size_t then_branch = bbs.size(); bbs.emplace_back(); bb_index = then_branch;
                diverges &= generate_ir_from_block(st.then_block);
                size_t then_branch_end = bb_index;
                size_t else_branch = bbs.size(); bbs.emplace_back(); bb_index = else_branch;
                diverges &= generate_ir_from_block(st.else_block);
                size_t else_branch_end = bb_index;
                size_t merge = bbs.size(); bbs.emplace_back(); bb_index = merge;
                { Instruction i = ins(IK::Branch, cond); i.i1 = then_branch; i.i2 = else_branch; bbs[branch].push_back(i); }
                { Instruction i = ins(IK::Jump); i.i1 = merge; bbs[then_branch_end].push_back(i); }
                { Instruction i = ins(IK::Jump); i.i1 = merge; bbs[else_branch_end].push_back(i); }
                return diverges;
            }
            case SK::While: {
                size_t loop_start = bbs.size();
                { Instruction i = ins(IK::Jump); i.i1 = loop_start; push(i); }
                bb_index = loop_start; bbs.emplace_back();
                bool is_infinite = st.a->kind == EK::Bool && st.a->bval;
                VariableId cond = generate_ir_from_expression(*st.a, compiler.type_bool());
                Span csp = compiler.expression_span(st.a.get());
                cond = autocast_variable_to_type(cond, compiler.type_bool(), csp);
                return finish_loop(st, loop_start, cond, is_infinite, false, 0);
            }
            case SK::For: {
                VariableId iteree = generate_ir_from_expression(*st.a, nullptr);
                std::vector<VariableId> argv{iteree};
                Span esp = compiler.expression_span(st.a.get());
                VariableId for_fn, ret_var;
                if (auto x = get_function_call_variables(esp, "for", argv, {esp})) { for_fn = std::get<0>(*x); ret_var = std::get<1>(*x); }
                else { for_fn = generate_error(); ret_var = generate_error(); }
                TypeId iterator_type = compiler.type_mut(variables[ret_var].typ);
                VariableId iterator_variable = new_variable(iterator_type, true);
                { Instruction i = ins(IK::Call, ret_var, for_fn); i.args = argv; push(i); }
                push(ins(IK::Allocate, iterator_variable));
                push(ins(IK::Store, iterator_variable, ret_var));
                size_t loop_start = bbs.size();
                { Instruction i = ins(IK::Jump); i.i1 = loop_start; push(i); }
                bb_index = loop_start; bbs.emplace_back();
                size_t scope = begin_scope();
                TypeId pointer_type = st.type ? compiler.type_mut(compiler.resolve_type(*st.type)) : compiler.type_primitive(PrimitiveType::Generic);
                VariableId iteration_variable;
                if (contains_local(st.name)) { compiler.report_shadowed(sspan, st.name); iteration_variable = generate_error(); }
                else {
                    iteration_variable = new_variable(pointer_type, false);
                    push(ins(IK::Allocate, iteration_variable));
                    locals_stack.push_back(st.name);
                    locals[st.name] = iteration_variable;
                }
                std::vector<VariableId> argv2{iterator_variable, iteration_variable};
                VariableId cont_fn, cond;
                if (auto x = get_function_call_variables(esp, "continue", argv2, {esp, esp})) { cont_fn = std::get<0>(*x); cond = std::get<1>(*x); }
                else { cont_fn = generate_error(); cond = generate_error(); }
                { Instruction i = ins(IK::Call, cond, cont_fn); i.args = argv2; push(i); }
                cond = autocast_variable_to_type(cond, compiler.type_bool(), esp);
                return finish_loop(st, loop_start, cond, false, true, scope);
            }
            case SK::Break:
                if (innermost_loop_begin) {
                    size_t len = instructions().size();
                    break_instructions_to_insert.emplace_back(bb_index, len);
                    push(ins(IK::BreakPlaceholder));
                    bb_index = bbs.size(); bbs.emplace_back();
                } else compiler.report_error(SErr::BreakOutsideLoop, sspan);
                break;
            case SK::Continue:
                if (innermost_loop_begin) {
                    Instruction i = ins(IK::Jump); i.i1 = *innermost_loop_begin; push(i);
                    bb_index = bbs.size(); bbs.emplace_back();
                } else compiler.report_error(SErr::ContinueOutsideLoop, sspan);
                break;
        }
        return false;
    }

    // Shared tail of While (:327-361) and For (:432-468).
    bool finish_loop(const Statement& st, size_t loop_start, VariableId cond, bool is_infinite, bool is_for, size_t scope) {
        size_t branch = bb_index;
        std::vector<std::pair<size_t, size_t>> old_breaks; old_breaks.swap(break_instructions_to_insert);
        auto old_loop = innermost_loop_begin;
        innermost_loop_begin = loop_start;
        size_t then_branch = bbs.size(); bbs.emplace_back(); bb_index = then_branch;
        generate_ir_from_block(st.then_block);
        size_t then_branch_end = bb_index;
        std::vector<std::pair<size_t, size_t>> new_breaks; new_breaks.swap(break_instructions_to_insert);
        break_instructions_to_insert.swap(old_breaks);
        innermost_loop_begin = old_loop;
        if (is_for) end_scope(scope);
        size_t else_branch = bbs.size(); bbs.emplace_back(); bb_index = else_branch;
        bool diverges = generate_ir_from_block(st.else_block);
        if (!is_for) diverges = diverges | (is_infinite && new_breaks.empty());
        size_t else_branch_end = bb_index;
        size_t merge = bbs.size(); bbs.emplace_back(); bb_index = merge;
        { Instruction i = ins(IK::Branch, cond); i.i1 = then_branch; i.i2 = else_branch; bbs[branch].push_back(i); }
        { Instruction i = ins(IK::Jump); i.i1 = loop_start; bbs[then_branch_end].push_back(i); }
        { Instruction i = ins(IK::Jump); i.i1 = merge; bbs[else_branch_end].push_back(i); }
        for (auto& b : new_breaks) { Instruction i = ins(IK::Jump); i.i1 = merge; bbs[b.first][b.second] = i; }
        return diverges;
    }

    // :495-503
    VariableId dereference_pointer(VariableId p) {
        TypeId typ = get_lvalue_type(p);
        VariableId v = new_variable(typ, true);
        push(ins(IK::Load, v, p));
        return v;
    }

    // :510-806
    VariableId generate_ir_from_expression(const Expression& e, TypeId expected_type) {
        Span espan = compiler.expression_span(&e);
        switch (e.kind) {
            case EK::Integer: {
                TypeId typ;
                if (e.has_meta) {
                    PrimitiveType p;
                    if (!e.is_signed) p = e.size == 8 ? PrimitiveType::Natural8 : e.size == 16 ? PrimitiveType::Natural16 : e.size == 32 ? PrimitiveType::Natural32 : e.size == 64 ? PrimitiveType::Natural64 : PrimitiveType::Size;
                    else p = e.size == 8 ? PrimitiveType::Integer8 : e.size == 16 ? PrimitiveType::Integer16 : e.size == 32 ? PrimitiveType::Integer32 : e.size == 64 ? PrimitiveType::Integer64 : PrimitiveType::Offset;
                    typ = compiler.type_primitive(p);
                } else if (expected_type && (expected_type->is_integral() || expected_type->kind == TyK::GenericInteger)) typ = expected_type;
                else typ = compiler.type_primitive(PrimitiveType::Integer32);
                VariableId v = new_variable(typ, true);
                Instruction i = ins(IK::Integer, v); i.ival = (int64_t)e.ival; push(i);
                return v;
            }
            case EK::Float: {
                TypeId typ = compiler.type_primitive(e.has_meta && e.size == 32 ? PrimitiveType::Float32 : PrimitiveType::Float64);
                VariableId v = new_variable(typ, true);
                Instruction i = ins(IK::Float, v); i.fval = e.fval; push(i);
                return v;
            }
            case EK::Null: { VariableId v = new_variable(compiler.type_null(), true); push(ins(IK::Null, v)); return v; }
            case EK::Bool: { VariableId v = new_variable(compiler.type_bool(), true); Instruction i = ins(IK::Bool, v); i.bval = e.bval; push(i); return v; }
            case EK::String: { VariableId v = new_variable(compiler.type_string(), true); Instruction i = ins(IK::String, v); i.bytes = e.bytes; push(i); return v; }
            case EK::Variable: case EK::Dereference: case EK::VariableDefinition: case EK::Field:
            case EK::NamedCall: case EK::CallOrIndex: case EK::Record: {
                auto [p, is_lvalue] = try_generate_pointer_from_expression(e, PointerType::Reference);
                return is_lvalue ? dereference_pointer(p) : p;
            }
            case EK::Proc: {
                std::vector<TypeId> types;
                for (auto& t : e.types) types.push_back(compiler.resolve_type(*t));
                if (FunctionId f = compiler.lookup_function(espan, e.name, types)) {
                    if (f->is_builtin) { compiler.report_error(SErr::ReferenceToBuiltinFunction, espan); return generate_error(); }
                    TypeId typ = compiler.type_function(f->arguments, f->return_type);
                    VariableId v = new_variable(typ, true);
                    Instruction i = ins(IK::Function, v); i.func = f; push(i);
                    return v;
                }
                return generate_error();
            }
            case EK::Operator: {
                Op op = e.op;
                if (op != Op::Not) {
                    IK kind;
                    switch (op) {
                        case Op::Plus: kind = IK::Add; break; case Op::Minus: kind = IK::Subtract; break;
                        case Op::Times: kind = IK::Multiply; break; case Op::Divide: kind = IK::Divide; break;
                        case Op::Modulo: kind = IK::Modulo; break; case Op::Equals: kind = IK::Equals; break;
                        case Op::LessThan: kind = IK::LessThan; break; case Op::GreaterThan: kind = IK::GreaterThan; break;
                        case Op::LessThanEquals: kind = IK::LessThanEquals; break; case Op::GreaterThanEquals: kind = IK::GreaterThanEquals; break;
                        case Op::Is: {
                            VariableId l = generate_ir_from_expression(*e.a, nullptr);
                            VariableId r = generate_ir_from_expression(*e.b, nullptr);
                            TypeId lt = variables[l].typ, rt = variables[r].typ;
                            if (lt->kind == TyK::Pointer && rt->kind == TyK::Pointer && lt->ptr == rt->ptr && compiler.types_match(lt->sub, rt->sub)) {
                                VariableId v = new_variable(compiler.type_bool(), true);
                                push(ins(IK::Equals, v, l, r));
                                return v;
                            }
                            compiler.report_operand_types(espan, op, lt, rt);
                            return generate_error();
                        }
                        case Op::And: case Op::Or: {
                            VariableId cond = generate_ir_from_expression(*e.a, compiler.type_bool());
                            Span csp = compiler.expression_span(e.a.get());
                            cond = autocast_variable_to_type(cond, compiler.type_bool(), csp);
                            size_t branch = bb_index;
                            size_t then_branch = bbs.size(); bbs.emplace_back();
                            bb_index = then_branch;
                            VariableId then_var = generate_ir_from_expression(*e.b, compiler.type_bool());
                            TypeId then_type = variables[then_var].typ;
                            if (!compiler.types_match(then_type, compiler.type_bool())) {
                                compiler.report_error(SErr::InvalidOperandType, compiler.expression_span(e.b.get()), then_type);
                                return generate_error();
                            }
                            size_t then_end = bb_index;
                            size_t merge = bbs.size(); bbs.emplace_back();
                            { Instruction i = ins(IK::Jump); i.i1 = merge; push(i); }
                            { Instruction i = ins(IK::Branch, cond);
                              if (op == Op::And) { i.i1 = then_branch; i.i2 = merge; } else { i.i1 = merge; i.i2 = then_branch; }
                              bbs[branch].push_back(i); }
                            bb_index = merge;
                            VariableId out = new_variable(compiler.type_bool(), true);
                            { Instruction i = ins(IK::Phi, out, cond, then_var); i.i1 = branch; i.i2 = then_end; push(i); }
                            return out;
                        }
                        default: throw Unsupported{"unreachable", 0};
                    }
                    const char* name = op_symbol(op);
                    TypeId gi = compiler.type_primitive(PrimitiveType::GenericInteger);
                    VariableId l = generate_ir_from_expression(*e.a, gi);
                    VariableId r = generate_ir_from_expression(*e.b, gi);
                    Span lsp = compiler.expression_span(e.a.get()), rsp = compiler.expression_span(e.b.get());
                    std::vector<VariableId> argv{l, r};
                    if (auto x = get_function_call_variables(espan, name, argv, {lsp, rsp})) {
                        auto [fv, out, is_builtin] = *x;
                        if (is_builtin) push(ins(kind, out, argv[0], argv[1]));
                        else { Instruction i = ins(IK::Call, out, fv); i.args = argv; push(i); }
                        return out;
                    }
                    return generate_error();
                } else {
                    TypeId gi = compiler.type_primitive(PrimitiveType::GenericInteger);
                    VariableId r = generate_ir_from_expression(*e.b, gi);
                    Span rsp = compiler.expression_span(e.b.get());
                    std::vector<VariableId> argv{r};
                    if (auto x = get_function_call_variables(espan, op_symbol(op), argv, {rsp})) {
                        auto [fv, out, is_builtin] = *x;
                        if (is_builtin) push(ins(IK::Not, out, argv[0]));
                        else { Instruction i = ins(IK::Call, out, fv); i.args = argv; push(i); }
                        return out;
                    }
                    return generate_error();
                }
            }
            case EK::Cast: {
                VariableId v = generate_ir_from_expression(*e.a, nullptr);
                TypeId old_type = variables[v].typ;
                TypeId typ = compiler.resolve_type(*e.type);
                if (auto ct = compiler.can_cast(old_type, typ)) {
                    VariableId nv = new_variable(typ, true);
                    Instruction i = ins(IK::Cast, nv, v); i.cast = *ct; push(i);
                    return nv;
                }
                compiler.report_error(SErr::InvalidCast, espan, old_type, typ);
                return generate_error();
            }
            case EK::Reference: return generate_pointer_from_expression(*e.a, PointerType::Reference);
            case EK::MutableReference: return generate_pointer_from_expression(*e.a, PointerType::Mutable);
            case EK::ArrayReference: return generate_pointer_from_expression(*e.a, PointerType::Array);
            case EK::MutableArrayReference: return generate_pointer_from_expression(*e.a, PointerType::ArrayMutable);
            case EK::Negate: {
                VariableId sub = generate_ir_from_expression(*e.a, expected_type);
                TypeId subtype = variables[sub].typ;
                switch (subtype->kind) {
                    case TyK::Integer8: case TyK::Integer16: case TyK::Integer32: case TyK::Integer64: case TyK::Offset: {
                        VariableId out = new_variable(subtype, true);
                        push(ins(IK::Negate, out, sub));
                        return out;
                    }
                    case TyK::GenericInteger: {
                        auto& is = instructions();
                        if (!is.empty() && is.back().kind == IK::Integer) {
                            is.back().ival = (int64_t)(0 - (uint64_t)is.back().ival);
                            return is.back().a;
                        }
                        throw Unsupported{"negate of <integer> not preceded by an Integer instruction (reference: unreachable!())", espan.low};
                    }
                    case TyK::Error: return generate_error();
                    default: compiler.report_error(SErr::InvalidOperandType, espan, subtype); return generate_error();
                }
            }
            case EK::Sizeof: case EK::Alignof: {
                VariableId v = new_variable(compiler.type_primitive(PrimitiveType::Size), true);
                TypeId typ = compiler.resolve_type(*e.type);
                Instruction i = ins(e.kind == EK::Sizeof ? IK::Sizeof : IK::Alignof, v); i.typ = typ; push(i);
                return v;
            }
            case EK::Parentheses: return generate_ir_from_expression(*e.a, expected_type);
        }
        throw Unsupported{"unreachable", 0};
    }

    // :808-855
    VariableId try_cast_pointer(VariableId v, PointerType pt, Span sp) {
        TypeId typ = variables[v].typ;
        if (typ->kind == TyK::Pointer) {
            PointerType pt2 = typ->ptr;
            if (pt2 == pt) return v;
            auto is = [&](PointerType a, PointerType b) { return pt2 == a && pt == b; };
            if (is(PointerType::Mutable, PointerType::Reference) || is(PointerType::ArrayMutable, PointerType::Array)
                || is(PointerType::Array, PointerType::Reference) || is(PointerType::ArrayMutable, PointerType::Mutable)
                || is(PointerType::ArrayMutable, PointerType::Reference)) {
                TypeId nt = compiler.type_pointer(pt, typ->sub);
                VariableId nv = new_variable(nt, true);
                Instruction i = ins(IK::Cast, nv, v); i.cast = CastType::PointerType; push(i);
                return nv;
            }
            if (is(PointerType::Reference, PointerType::Mutable) || is(PointerType::Array, PointerType::ArrayMutable)) {
                compiler.report_error(SErr::ImmutableLvalue, sp); return generate_error();
            }
            compiler.report_error(SErr::ArrayPointerToNonArray, sp); return generate_error();
        }
        if (typ->kind == TyK::Error) return generate_error();
        compiler.report_error(SErr::InvalidOperandType, sp, typ);
        return generate_error();
    }

    // :857-893
    std::optional<std::pair<VariableId, bool>> lookup_name(std::string_view name, Span sp, PointerType pt) {
        bool is_lvalue = true;
        VariableId var;
        auto lit = locals.find(name);
        if (lit != locals.end()) var = try_cast_pointer(lit->second, pt, sp);
        else {
            auto git = compiler.global_resolution_map.find(name);
            if (git != compiler.global_resolution_map.end()) {
                size_t gid = git->second;
                if (compiler.global_is_constant[gid]) return std::nullopt;
                TypeId typ = compiler.type_mut(compiler.globals[gid].first);
                VariableId v = new_variable(typ, true);
                Instruction i = ins(IK::GlobalVariable, v); i.i1 = gid; push(i);
                var = try_cast_pointer(v, pt, sp);
            } else {
                auto fit = compiler.name_resolution_map.find(name);
                if (fit != compiler.name_resolution_map.end()) {
                    auto& ids = fit->second;
                    if (ids.size() == 0) { compiler.report_error(SErr::UndefinedVariable, compiler.sv_span(name)); var = generate_error(); }
                    else if (ids.size() > 1) { compiler.report_error(SErr::AmbiguousFunctionReference, compiler.sv_span(name)); var = generate_error(); }
                    else {
                        is_lvalue = false;
                        FunctionId f = ids[0];
                        TypeId typ = compiler.type_function(f->arguments, f->return_type);
                        VariableId v = new_variable(typ, true);
                        Instruction i = ins(IK::Function, v); i.func = f; push(i);
                        var = v;
                    }
                } else { compiler.report_error(SErr::UndefinedVariable, compiler.sv_span(name)); var = generate_error(); }
            }
        }
        return std::make_pair(var, is_lvalue);
    }

    // :895-967
    std::pair<VariableId, bool> generate_ir_for_index(VariableId indexee, bool indexee_is_lvalue, Span sp, Span indexee_span,
                                                      const std::vector<ExprBox>& arguments, PointerType pt) {
        TypeId indexee_type = indexee_is_lvalue ? compiler.type_deref(variables[indexee].typ) : variables[indexee].typ;
        bool is_lvalue = true;
        TypeId type_size = compiler.type_primitive(PrimitiveType::Size);
        VariableId var;
        switch (indexee_type->kind) {
            case TyK::Pointer: {
                PointerType pt2 = indexee_type->ptr; TypeId subtype = indexee_type->sub;
                if (indexee_is_lvalue) indexee = dereference_pointer(indexee);
                std::vector<VariableId> argv;
                for (auto& a : arguments) argv.push_back(generate_ir_from_expression(*a, type_size));
                if (argv.size() != 1) compiler.report_error(SErr::InvalidNumberOfArgs, sp);
                if (argv.empty()) throw Unsupported{"index with zero arguments (reference panics on pop().unwrap())", sp.low};
                VariableId index = argv.back();
                TypeId result_type;
                if (pt2 == PointerType::Array) result_type = compiler.type_refs(subtype);
                else if (pt2 == PointerType::ArrayMutable) result_type = compiler.type_muts(subtype);
                else { compiler.report_error(SErr::InvalidOperandType, indexee_span, indexee_type); return {generate_error(), true}; }
                VariableId rv = new_variable(result_type, true);
                push(ins(IK::IndexPointer, rv, indexee, index));
                var = try_cast_pointer(rv, pt, indexee_span);
                break;
            }
            case TyK::Array: {
                TypeId subtype = indexee_type->sub;
                if (!indexee_is_lvalue) { compiler.report_error(SErr::InvalidLvalue, sp); return {generate_error(), true}; }
                std::vector<VariableId> argv;
                for (auto& a : arguments) argv.push_back(generate_ir_from_expression(*a, type_size));
                if (argv.size() != 1) compiler.report_error(SErr::InvalidNumberOfArgs, sp);
                if (argv.empty()) throw Unsupported{"index with zero arguments (reference panics on pop().unwrap())", sp.low};
                VariableId index = argv.back();
                TypeId it = variables[indexee].typ;
                if (it->kind != TyK::Pointer) throw Unsupported{"unreachable", sp.low};
                TypeId result_type = compiler.type_pointer(it->ptr, subtype);
                VariableId rv = new_variable(result_type, true);
                push(ins(IK::IndexArray, rv, indexee, index));
                var = try_cast_pointer(rv, pt, indexee_span);
                break;
            }
            case TyK::Function: {
                TypeId return_type = indexee_type->ret;
                if (indexee_is_lvalue) indexee = dereference_pointer(indexee);
                is_lvalue = false;
                std::vector<VariableId> argv;
                for (auto& a : arguments) argv.push_back(generate_ir_from_expression(*a, nullptr));
                VariableId out = new_variable(return_type, true);
                Instruction i = ins(IK::Call, out, indexee); i.args = argv; push(i);
                var = out;
                break;
            }
            default:
                compiler.report_error(SErr::InvalidOperandType, indexee_span, indexee_type);
                var = generate_error();
        }
        return {var, is_lvalue};
    }

    // :970-979
    VariableId generate_pointer_from_expression(const Expression& e, PointerType pt) {
        auto [var, lvalue] = try_generate_pointer_from_expression(e, pt);
        if (!lvalue) { compiler.report_error(SErr::InvalidLvalue, compiler.expression_span(&e)); return generate_error(); }
        return var;
    }

    // :985-1088
    std::pair<VariableId, bool> try_generate_pointer_from_expression(const Expression& e, PointerType pt) {
        Span sp = compiler.expression_span(&e);
        bool is_lvalue = true;
        VariableId var;
        switch (e.kind) {
            case EK::Variable: {
                if (auto r = lookup_name(e.name, sp, pt)) { is_lvalue = r->second; var = r->first; }
                else {
                    auto git = compiler.global_resolution_map.find(e.name);
                    if (git != compiler.global_resolution_map.end()) {
                        size_t gid = git->second;
                        if (compiler.global_is_constant[gid]) {
                            VariableId v = new_variable(compiler.globals[gid].first, true);
                            Instruction i = ins(IK::GlobalConstant, v); i.i1 = gid; push(i);
                            return {v, false};
                        }
                        compiler.report_error(SErr::UndefinedVariable, sp); var = generate_error();
                    } else { compiler.report_error(SErr::UndefinedVariable, sp); var = generate_error(); }
                }
                break;
            }
            case EK::VariableDefinition: {
                if (contains_local(e.name)) { compiler.report_shadowed(sp, e.name); var = generate_error(); }
                else {
                    TypeId typ = compiler.resolve_type(*e.type);
                    VariableId v = new_variable(compiler.type_mut(typ), false);
                    push(ins(IK::Allocate, v));
                    locals_stack.push_back(e.name);
                    locals[e.name] = v;
                    var = try_cast_pointer(v, pt, sp);
                }
                break;
            }
            case EK::Dereference: {
                VariableId v = generate_ir_from_expression(*e.a, nullptr);
                var = try_cast_pointer(v, pt, compiler.expression_span(e.a.get()));
                break;
            }
            case EK::Field: {
                VariableId rec = generate_pointer_from_expression(*e.a, pt);
                var = generate_ir_for_field_access(rec, e.name, pt, sp);
                break;
            }
            case EK::NamedCall: {
                if (compiler.name_resolution_map.count(e.name)) {
                    is_lvalue = false;
                    std::vector<VariableId> argv;
                    TypeId gi = compiler.type_primitive(PrimitiveType::GenericInteger);
                    for (auto& a : e.args) argv.push_back(generate_ir_from_expression(*a, gi));
                    std::vector<Span> spans;
                    for (auto& a : e.args) spans.push_back(compiler.expression_span(a.get()));
                    if (auto x = get_function_call_variables(sp, e.name, argv, spans)) {
                        auto [fv, out, b] = *x; (void)b;
                        Instruction i = ins(IK::Call, out, fv); i.args = argv; push(i);
                        var = out;
                    } else var = generate_error();
                } else {
                    if (auto r = lookup_name(e.name, sp, pt)) {
                        Span name_span = compiler.sv_span(e.name);
                        auto [v2, l2] = generate_ir_for_index(r->first, r->second, sp, name_span, e.args, pt);
                        is_lvalue = l2; var = v2;
                    } else { compiler.report_error(SErr::AttemptToCallOrIndexConstant, sp); var = generate_error(); }
                }
                break;
            }
            case EK::CallOrIndex: {
                PointerType pt2 = (pt == PointerType::Array || pt == PointerType::Reference) ? PointerType::Reference : PointerType::Mutable;
                auto [iv, il] = try_generate_pointer_from_expression(*e.a, pt2);
                Span isp = compiler.expression_span(e.a.get());
                auto [v2, l2] = generate_ir_for_index(iv, il, sp, isp, e.args, pt);
                is_lvalue = l2; var = v2;
                break;
            }
            case EK::Record: {
                TypeId typ = compiler.type_mut(compiler.resolve_type(*e.type));
                VariableId out = new_variable(typ, true);
                push(ins(IK::Allocate, out));
                for (auto& f : e.fields) {
                    TypeId ft = get_field_type(out, f.first);
                    VariableId val = generate_ir_from_expression(*f.second, ft);
                    Span vsp = compiler.expression_span(f.second.get());
                    VariableId fv = generate_ir_for_field_access(out, f.first, PointerType::Mutable, vsp);
                    push(ins(IK::Store, fv, val));
                }
                var = out;
                break;
            }
            case EK::Parentheses: return try_generate_pointer_from_expression(*e.a, pt);
            default:
                is_lvalue = false;
                var = generate_ir_from_expression(e, nullptr);
        }
        return {var, is_lvalue};
    }

    // :1090-1101
    TypeId get_field_type(VariableId rec, std::string_view field_name) {
        TypeId typ = get_lvalue_type(rec);
        if (typ->kind == TyK::Record) for (auto& f : typ->fields) if (f.first == field_name) return f.second;
        return nullptr;
    }

    // :1103-1149
    VariableId generate_ir_for_field_access(VariableId rec, std::string_view field_name, PointerType pt, Span sp) {
        if (pt == PointerType::Array || pt == PointerType::ArrayMutable) { compiler.report_error(SErr::ArrayPointerToNonArray, sp); return generate_error(); }
        TypeId typ = get_lvalue_type(rec);
        if (typ->kind == TyK::Record) {
            TypeId ft = nullptr;
            for (auto& f : typ->fields) if (f.first == field_name) { ft = f.second; break; }
            if (ft) {
                VariableId rv = new_variable(compiler.type_pointer(pt, ft), true);
                Instruction i = ins(IK::Field, rv, rec); i.field = field_name; push(i);
                return rv;
            }
            compiler.report_field_missing(sp, typ, field_name); return generate_error();
        }
        if (typ->kind == TyK::String) {
            size_t idx; TypeId ft;
            if (field_name == "data") { idx = 0; ft = compiler.type_refs(compiler.type_primitive(PrimitiveType::Natural8)); }
            else if (field_name == "length") { idx = 1; ft = compiler.type_primitive(PrimitiveType::Size); }
            else { compiler.report_field_missing(sp, typ, field_name); return generate_error(); }
            VariableId rv = new_variable(compiler.type_pointer(pt, ft), true);
            Instruction i = ins(IK::BuiltinField, rv, rec); i.i1 = idx; push(i);
            return rv;
        }
        compiler.report_error(SErr::FieldAccessOnNonRecord, sp, typ);
        return generate_error();
    }

    // :1151-1179
    struct Lookup { VariableId fv; const std::vector<TypeId>* args; TypeId ret; bool builtin; };
    std::optional<Lookup> lookup_function(Span sp, std::string_view name, const std::vector<TypeId>& argument_types) {
        auto lit = locals.find(name);
        if (lit != locals.end()) {
            VariableId variable = lit->second;
            TypeId typ = get_lvalue_type(variable);
            VariableId nv = new_variable(typ, true);
            push(ins(IK::Load, nv, variable));
            if (typ->kind == TyK::Function) return Lookup{nv, &typ->args, typ->ret, false};
            compiler.report_error(SErr::CallOfNonFunction, sp, typ);
            return std::nullopt;
        }
        if (FunctionId f = compiler.lookup_function(sp, name, argument_types)) {
            TypeId typ = compiler.type_function(f->arguments, f->return_type);
            if (f->is_builtin) return Lookup{NOVAR, &f->arguments, f->return_type, true};
            VariableId v = new_variable(typ, true);
            Instruction i = ins(IK::Function, v); i.func = f; push(i);
            return Lookup{v, &f->arguments, f->return_type, false};
        }
        return std::nullopt;
    }

    // :1181-1198
    std::optional<std::tuple<VariableId, VariableId, bool>> get_function_call_variables(Span sp, std::string_view name, std::vector<VariableId>& argv, std::vector<Span> arg_spans) {
        std::vector<TypeId> argument_types;
        for (VariableId v : argv) argument_types.push_back(variables[v].typ);
        for (TypeId t : argument_types) if (compiler.is_error_type(t)) return std::nullopt;
        auto l = lookup_function(sp, name, argument_types);
        if (!l) return std::nullopt;
        if (l->args->size() != argv.size()) {
            // debug_assert_eq! at :1189; in release zip() truncates.  Only reachable through a local
            // function-typed variable called with the wrong arity.
        }
        size_t n = std::min(l->args->size(), argv.size());
        for (size_t i = 0; i < n; i++) argv[i] = autocast_variable_to_type(argv[i], (*l->args)[i], arg_spans[i]);
        VariableId out = new_variable(l->ret, true);
        return std::make_tuple(l->fv, out, l->builtin);
    }
};

}  // namespace

// frontend.rs:140-187 (finalize, then one IrGenerator per Definition::Function, in order)
void generate_ir(Unit& u) {
    Compiler& c = *u.compiler;
    for (auto& d : u.definitions) c.finalize_definition(*d);
    if (c.has_errors) { u.ir_generated = false; return; }   // frontend.rs:145,150
    u.ir_generated = true;
    for (auto& d : u.definitions) {
        if (d->kind != DK::Function) continue;
        IrGenerator g(c, d->sig, c.definition_spans.at(d.get()));
        g.generate_ir_from_function(d->body);
        Unit::FunctionIr f;
        f.function = g.function;
        u.n_vars += g.variables.size();
        for (auto& bb : g.bbs) u.n_instrs += bb.size();
        f.variables = std::move(g.variables);
        f.bbs = std::move(g.bbs);
        u.ir.push_back(std::move(f));
    }
}

std::unique_ptr<Unit> run(const char* src, size_t len, bool lower, bool keep_token_log) {
    auto u = std::make_unique<Unit>();
    u->keep_token_log = keep_token_log;
    u->source.assign(src, len);
    u->compiler = std::make_unique<Compiler>();
    u->compiler->src_base = u->source.data();
    u->compiler->src_len = len;
    get_definitions(*u);
    if (u->has_parse_error || u->unsupported) return u;
    for (auto& d : u->definitions) u->compiler->preload_definition(*d);   // frontend.rs:98
    if (lower) {
        try { generate_ir(*u); }
        catch (Unsupported& e) { u->unsupported = true; u->unsupported_why = e.why; u->unsupported_at = e.at; }
    }
    return u;
}

}  // namespace oracle

// Deterministic synthetic Opus-source generator G(seed, bytes)   + single-fault
// syntax-error injector (C5).  Test/bench tooling: it produces INPUTS only; it is not part of the
// product path and not part of the oracle.
//
// Programs are ASCII, TAB-indented and well-typed with respect to the reference's rules
// (compile.rs:624-775 type algebra, :1083-1139 overload resolution, generate_ir.rs literal typing
// :514-549), so that IR generation reports no semantic error unless a fault is injected.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

struct Rng {
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed) {}
    uint64_t next() {  // SplitMix64
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    uint32_t below(uint32_t n) { return (uint32_t)(next() % n); }
    bool chance(uint32_t pct) { return below(100) < pct; }
    uint32_t range(uint32_t lo, uint32_t hi) { return lo + below(hi - lo + 1); }
};

enum Ty { I32, I64, N8, BOOL, F64, SIZE, NTY };
const char* TY_NAME[] = {"int32", "int64", "nat8", "bool", "float64", "size"};
const char* TY_SUFFIX[] = {"i32", "i64", "n8", "", "", ""};

struct Local { std::string name; Ty ty; };
struct RecType { std::string name; std::vector<std::pair<std::string, Ty>> fields; };
struct RecLocal { std::string name; int rec; };
struct ArrLocal { std::string name; Ty ty; uint32_t n; };
struct Func { std::string name; std::vector<Ty> params; int ret; /* -1 = null */ };
struct Global { std::string name; Ty ty; };

const char* WORDS[] = {"x", "i", "n", "tmp", "acc", "val", "idx", "count", "sum", "buf", "len", "pos", "res", "flag",
                       "k", "j", "lo", "hi", "mid", "cur", "next", "prev", "total", "delta", "step", "limit", "base",
                       "off", "mask", "bits", "rate", "scale", "w", "h", "a", "b", "c", "d", "e", "t", "u", "v"};
const int NWORDS = sizeof(WORDS) / sizeof(WORDS[0]);
const char* FWORDS[] = {"calc", "advance", "fold", "scan", "mix", "eval", "pack", "hash", "norm", "clip", "blend", "walk",
                        "probe", "merge", "split", "apply", "emit", "lower", "visit", "tally"};
const int NFWORDS = sizeof(FWORDS) / sizeof(FWORDS[0]);

struct Gen {
    Rng rng;
    std::string out;
    std::string tag;        // block tag making names globally unique
    uint32_t uid = 0;       // per-block unique counter for definitions
    uint32_t lid = 0;       // per-function unique counter for locals
    std::vector<Func> funcs;
    std::vector<Global> gvars, gconsts;
    std::vector<RecType> recs;
    // function-local state
    std::vector<std::vector<Local>> scopes;
    std::vector<std::vector<RecLocal>> rscopes;
    std::vector<std::vector<ArrLocal>> ascopes;
    int cur_ret = -1;
    bool have_prelude;

    Gen(uint64_t seed, std::string tag_, bool prelude) : rng(seed), tag(std::move(tag_)), have_prelude(prelude) {}

    // Zipf-ish word pick: low indices far more likely
    const char* word() { uint32_t r = rng.below(NWORDS * NWORDS); uint32_t k = 0; while ((k + 1) * (k + 1) <= r) k++; return WORDS[NWORDS - 1 - (k >= (uint32_t)NWORDS ? NWORDS - 1 : k)]; }
    std::string local_name() { return std::string(word()) + "_" + std::to_string(lid++); }
    std::string def_name(const char* w) { return std::string(w) + "_" + tag + std::to_string(uid++); }

    void tabs(int d) { out.append((size_t)d, '\t'); }
    void maybe_trailing_comment() { if (rng.chance(3)) { out += " ; "; out += word(); out += " note"; } }
    void eol() { maybe_trailing_comment(); out += '\n'; }
    void noise(int d) {
        if (rng.chance(5)) { tabs(d); out += "; "; out += word(); out += ' '; out += word(); out += " comment line\n"; }
        if (rng.chance(5)) out += '\n';
    }

    // ---------------- literals ----------------
    void int_literal(Ty t, bool need_suffix) {
        uint32_t v = (uint32_t)(rng.next() & 0x7FFFFFFF);
        if (t == N8) v &= 0xFF;
        if (rng.chance(60)) v %= 1000;
        out += std::to_string(v);
        if (t == SIZE) return;   // Q1: a bare `s`/`o` suffix is never consumed by the reference lexer -> never emitted
        if (need_suffix || rng.chance(10)) out += TY_SUFFIX[t];
    }
    void float_literal() {
        out += std::to_string(rng.below(100000)); out += '.'; out += std::to_string(rng.below(10000));
        if (rng.chance(2)) out += "f64";
    }
    void string_literal() {
        out += '"';
        uint32_t n = rng.below(41);
        for (uint32_t i = 0; i < n; i++) {
            if (rng.chance(10)) {
                switch (rng.below(7)) {
                    case 0: out += "\\\\"; break; case 1: out += "\\\""; break; case 2: out += "\\n"; break;
                    case 3: out += "\\r"; break; case 4: out += "\\t"; break; case 5: out += "\\0"; break;
                    default: { static const char* H = "0123456789abcdefABCDEF"; out += "\\x"; out += H[rng.below(8)]; out += H[rng.below(22)]; }
                }
            } else {
                char c = (char)rng.range(0x20, 0x7e);
                if (c == '"' || c == '\\') c = '_';
                out += c;
            }
        }
        out += '"';
    }

    // ---------------- lookup of typed names ----------------
    void collect(Ty t, std::vector<const std::string*>& v) {
        for (auto& s : scopes) for (auto& l : s) if (l.ty == t) v.push_back(&l.name);
    }
    bool typed_var(Ty t) {   // emit a variable (local / global var / global const) of type t
        std::vector<const std::string*> v; collect(t, v);
        size_t nl = v.size();
        for (auto& g : gvars) if (g.ty == t) v.push_back(&g.name);
        for (auto& g : gconsts) if (g.ty == t) v.push_back(&g.name);
        if (v.empty()) return false;
        // prefer locals
        size_t k = (nl && rng.chance(80)) ? rng.below((uint32_t)nl) : rng.below((uint32_t)v.size());
        out += *v[k];
        return true;
    }
    bool mutable_var(Ty t, std::string& name) {
        std::vector<const std::string*> v; collect(t, v);
        size_t nl = v.size();
        for (auto& g : gvars) if (g.ty == t) v.push_back(&g.name);
        if (v.empty()) return false;
        size_t k = (nl && rng.chance(85)) ? rng.below((uint32_t)nl) : rng.below((uint32_t)v.size());
        name = *v[k];
        return true;
    }
    int pick_func_returning(int ret) {
        if (funcs.empty()) return -1;
        for (int tries = 0; tries < 6; tries++) {
            // earlier functions, biased to recent ones
            uint32_t n = (uint32_t)funcs.size();
            uint32_t k = rng.chance(70) ? n - 1 - rng.below(n < 16 ? n : 16) : rng.below(n);
            if (ret == -2 || funcs[k].ret == ret) return (int)k;
        }
        return -1;
    }

    void call(int f, int depth) {
        out += funcs[f].name; out += '(';
        for (size_t i = 0; i < funcs[f].params.size(); i++) {
            if (i) out += ", ";
            expr(funcs[f].params[i], depth > 1 ? 1 : depth, /*direct=*/true);
        }
        out += ')';
    }

    // An atom of integer-ish/float type t (binds tighter than any binary operator).
    void atom(Ty t, int depth) {
        uint32_t r = rng.below(100);
        if (depth > 0 && r < 15) { int f = pick_func_returning((int)t); if (f >= 0) { call(f, depth - 1); return; } }
        if (depth > 0 && r < 35) { out += '('; expr(t, depth - 1, false); out += ')'; return; }
        if (r < 42 && (t == I32 || t == I64)) {   // unary minus (signed only: generate_ir.rs:749-789)
            out += '-';
            if (rng.chance(50) && typed_var(t)) return;
            int_literal(t, t != I32);
            return;
        }
        if (r < 47 && t != BOOL && t != F64 && depth > 0) {   // cast from another integer type
            Ty o = (Ty)rng.below(3);
            if (o != t && t != SIZE) { std::vector<const std::string*> v; collect(o, v); if (!v.empty()) { out += *v[rng.below((uint32_t)v.size())]; out += " : "; out += TY_NAME[t]; return; } }
        }
        if (r < 52 && field_read(t)) return;
        if (r < 55 && array_read(t)) return;
        if (r < 80 && typed_var(t)) return;
        if (t == F64) float_literal();
        else if (t == BOOL) out += rng.chance(50) ? "true" : "false";
        else int_literal(t, t != I32);
    }

    bool field_read(Ty t) {
        std::vector<std::pair<const std::string*, const std::string*>> v;
        for (auto& s : rscopes) for (auto& rl : s) for (auto& f : recs[rl.rec].fields) if (f.second == t) v.push_back({&rl.name, &f.first});
        if (v.empty()) return false;
        auto& p = v[rng.below((uint32_t)v.size())];
        out += *p.first; out += '.'; out += *p.second;
        return true;
    }
    bool array_read(Ty t) {
        std::vector<const ArrLocal*> v;
        for (auto& s : ascopes) for (auto& a : s) if (a.ty == t) v.push_back(&a);
        if (v.empty()) return false;
        auto* a = v[rng.below((uint32_t)v.size())];
        out += a->name; out += '('; out += std::to_string(rng.below(a->n)); out += ')';
        return true;
    }

    // expression of type t.  `direct`: the expression sits where the reference passes an expected
    // type that a lone unsuffixed literal adopts (call argument, typed var init, assignment, return).
    void expr(Ty t, int depth, bool direct) {
        if (t == BOOL) { bool_expr(depth); return; }
        if (direct && rng.chance(depth > 0 ? 15 : 40) && t != F64) { int_literal(t, false); return; }
        if (depth <= 0 || rng.chance(30)) { atom(t, depth); return; }
        int nops = (int)rng.range(1, 3);
        atom(t, depth - 1);
        for (int i = 0; i < nops; i++) {
            static const char* OPS[] = {" + ", " - ", " * ", " / ", " % "};
            out += OPS[rng.below(t == F64 ? 4 : 5)];
            atom(t, depth - 1);
        }
    }
    void bool_atom(int depth) {
        uint32_t r = rng.below(100);
        if (r < 45 || depth <= 0) {
            if (r < 12 && typed_var(BOOL)) return;
            if (r < 18) { out += rng.chance(50) ? "true" : "false"; return; }
            Ty t = (Ty)rng.below(3);
            if (rng.chance(8)) t = F64;
            static const char* CMP[] = {" == ", " < ", " > ", " <= ", " >= "};
            expr(t, depth > 1 ? 1 : 0, false); out += CMP[rng.below(5)]; expr(t, depth > 1 ? 1 : 0, false);
            return;
        }
        if (r < 55) { int f = pick_func_returning((int)BOOL); if (f >= 0) { call(f, depth - 1); return; } }
        if (r < 70) { out += '('; bool_expr(depth - 1); out += ')'; return; }
        if (r < 80) { out += "not "; if (!typed_var(BOOL)) { out += '('; bool_expr(depth - 1); out += ')'; } return; }
        Ty t = (Ty)rng.below(3);
        static const char* CMP[] = {" == ", " < ", " > ", " <= ", " >= "};
        atom(t, depth - 1); out += CMP[rng.below(5)]; atom(t, depth - 1);
    }
    void bool_expr(int depth) {
        bool_atom(depth);
        int n = depth > 0 ? (int)rng.below(3) : 0;
        for (int i = 0; i < n; i++) { out += rng.chance(50) ? " and " : " or "; bool_atom(depth - 1); }
    }

    // ---------------- statements ----------------
    void push_scope() { scopes.emplace_back(); rscopes.emplace_back(); ascopes.emplace_back(); }
    void pop_scope() { scopes.pop_back(); rscopes.pop_back(); ascopes.pop_back(); }

    void block(int d, int nstmts, bool in_loop, int nest) {
        push_scope();
        for (int i = 0; i < nstmts; i++) statement(d, in_loop, nest);
        pop_scope();
    }

    void statement(int d, bool in_loop, int nest) {
        noise(d);
        uint32_t r = rng.below(100);
        int depth = (int)rng.below(5);
        if (depth > 3) depth = (int)rng.range(3, 6);
        if (r < 25) {   // var-def
            Ty t = (Ty)rng.below(5);
            if (rng.chance(50)) t = I32;
            std::string name = local_name();
            tabs(d); out += "var "; out += name;
            uint32_t form = rng.below(100);
            if (t == I32 && form < 60) { out += " = "; expr(t, depth, false); }
            else if (t == F64 && form < 60) { out += " = "; expr(t, depth, false); }
            else if (t == BOOL && form < 60) { out += " = "; bool_expr(depth); }
            else if (form < 90) { out += ' '; out += TY_NAME[t]; out += " = "; expr(t, depth, true); }
            else { out += ' '; out += TY_NAME[t]; }
            eol();
            scopes.back().push_back({name, t});
        } else if (r < 50) {   // assignment
            Ty t = (Ty)rng.below(5);
            std::string name;
            if (rng.chance(12) && field_write(d, depth)) return;
            if (rng.chance(6) && array_write(d, depth)) return;
            if (!mutable_var(t, name)) { t = I32; if (!mutable_var(t, name)) { simple_def(d, depth); return; } }
            tabs(d); out += name; out += " = "; expr(t, depth, true); eol();
        } else if (r < 65) {   // if / else
            tabs(d); out += "if "; bool_expr(depth > 3 ? 3 : depth); eol();
            if (nest < 3) block(d + 1, (int)rng.range(1, 4), in_loop, nest + 1); else { push_scope(); leaf_statement(d + 1, in_loop); pop_scope(); }
            if (rng.chance(40)) {
                tabs(d); out += "else"; eol();
                if (nest < 3) block(d + 1, (int)rng.range(1, 3), in_loop, nest + 1); else { push_scope(); leaf_statement(d + 1, in_loop); pop_scope(); }
            }
        } else if (r < 73) {   // while
            tabs(d); out += "while "; bool_expr(depth > 2 ? 2 : depth); eol();
            if (nest < 3) block(d + 1, (int)rng.range(1, 4), true, nest + 1); else { push_scope(); leaf_statement(d + 1, true); pop_scope(); }
            if (rng.chance(10)) { tabs(d); out += "else"; eol(); push_scope(); leaf_statement(d + 1, in_loop); pop_scope(); }
        } else if (r < 83) {   // return
            tabs(d); out += "return ";
            if (cur_ret < 0) out += "null"; else expr((Ty)cur_ret, depth, true);
            eol();
        } else if (r < 95) {   // call statement
            int f = pick_func_returning(-2);
            if (f < 0) { simple_def(d, depth); return; }
            tabs(d); call(f, depth > 2 ? 2 : depth); eol();
        } else {   // break / continue / for / record / string / misc
            if (in_loop && rng.chance(60)) { tabs(d); out += rng.chance(50) ? "break" : "continue"; eol(); return; }
            misc_statement(d, in_loop, nest, depth);
        }
    }
    void leaf_statement(int d, bool in_loop) {
        int depth = (int)rng.below(3);
        std::string name;
        if (in_loop && rng.chance(20)) { tabs(d); out += rng.chance(50) ? "break" : "continue"; eol(); return; }
        Ty t = (Ty)rng.below(3);
        if (mutable_var(t, name)) { tabs(d); out += name; out += " = "; expr(t, depth, true); eol(); }
        else simple_def(d, depth);
    }
    void simple_def(int d, int depth) {
        std::string name = local_name();
        tabs(d); out += "var "; out += name; out += " = "; expr(I32, depth, false); eol();
        scopes.back().push_back({name, I32});
    }
    bool field_write(int d, int depth) {
        std::vector<std::pair<const RecLocal*, int>> v;
        for (auto& s : rscopes) for (auto& rl : s) for (size_t k = 0; k < recs[rl.rec].fields.size(); k++) v.push_back({&rl, (int)k});
        if (v.empty()) return false;
        auto& p = v[rng.below((uint32_t)v.size())];
        auto& f = recs[p.first->rec].fields[p.second];
        tabs(d); out += p.first->name; out += '.'; out += f.first; out += " = "; expr(f.second, depth, true); eol();
        return true;
    }
    bool array_write(int d, int depth) {
        std::vector<const ArrLocal*> v;
        for (auto& s : ascopes) for (auto& a : s) v.push_back(&a);
        if (v.empty()) return false;
        auto* a = v[rng.below((uint32_t)v.size())];
        tabs(d); out += a->name; out += '('; out += std::to_string(rng.below(a->n)); out += ") = "; expr(a->ty, depth, true); eol();
        return true;
    }
    void misc_statement(int d, bool /*in_loop*/, int nest, int depth) {
        uint32_t r = rng.below(100);
        if (r < 30 && !recs.empty()) {   // record literal
            int k = (int)rng.below((uint32_t)recs.size());
            std::string name = local_name();
            tabs(d); out += "var "; out += name; out += " = record "; out += recs[k].name; out += " {";
            bool multi = rng.chance(30);
            for (auto& f : recs[k].fields) {
                if (multi) { out += '\n'; tabs(d + 2); } else out += ' ';
                out += f.first; out += " = "; expr(f.second, depth > 2 ? 2 : depth, true);
            }
            if (multi) { out += '\n'; tabs(d); } else out += ' ';
            out += '}'; eol();
            rscopes.back().push_back({name, k});
        } else if (r < 45) {   // string
            std::string name = local_name();
            tabs(d); out += "var "; out += name; out += " = "; string_literal(); eol();
            if (rng.chance(50)) {
                std::string n2 = local_name();
                tabs(d); out += "var "; out += n2; out += " = "; out += name; out += ".length"; eol();
                scopes.back().push_back({n2, SIZE});
            }
        } else if (r < 60) {   // fixed array
            Ty t = (Ty)rng.below(3);
            uint32_t n = rng.range(2, 16);
            std::string name = local_name();
            tabs(d); out += "var "; out += name; out += " {"; out += std::to_string(n); out += "} "; out += TY_NAME[t]; eol();
            ascopes.back().push_back({name, t, n});
        } else if (r < 72 && have_prelude && nest < 3) {   // for loop over the prelude Range
            std::string name = local_name();
            tabs(d); out += "for "; out += name;
            if (rng.chance(50)) out += " int32";
            out += " in record Range { lo = "; expr(I32, 1, true); out += " hi = "; expr(I32, 1, true); out += " }"; eol();
            push_scope();
            scopes.back().push_back({name, I32});
            for (int i = 0, n = (int)rng.range(1, 3); i < n; i++) statement(d + 1, true, nest + 1);
            pop_scope();
        } else if (r < 82) {   // pointers
            std::string target;
            Ty t = (Ty)rng.below(3);
            std::vector<const std::string*> v; collect(t, v);
            if (v.empty()) { simple_def(d, depth); return; }
            target = *v[rng.below((uint32_t)v.size())];
            std::string p = local_name();
            tabs(d); out += "var "; out += p; out += " = mut "; out += target; eol();
            tabs(d); out += p; out += "@ = "; expr(t, depth > 2 ? 2 : depth, true); eol();
            if (rng.chance(50)) { std::string q = local_name(); tabs(d); out += "var "; out += q; out += " = "; out += p; out += "@ + "; int_literal(t, t != I32); eol(); scopes.back().push_back({q, t}); }
        } else if (r < 90) {   // sizeof / alignof
            std::string name = local_name();
            tabs(d); out += "var "; out += name; out += rng.chance(50) ? " = sizeof " : " = alignof ";
            static const char* TS[] = {"int32", "@int64", "{} nat8", "{4} int32", "@mut bool", "(int32, int64) bool", "{} mut float64"};
            out += TS[rng.below(7)]; eol();
            scopes.back().push_back({name, SIZE});
        } else simple_def(d, depth);
    }

    // ---------------- definitions ----------------
    void function_def() {
        Func f;
        f.name = def_name(FWORDS[rng.below(NFWORDS)]);
        int np = (int)rng.range(1, 4);
        static const Ty PT[] = {I32, I64, N8, BOOL};
        for (int i = 0; i < np; i++) f.params.push_back(PT[rng.below(4)]);
        uint32_t rr = rng.below(100);
        f.ret = rr < 15 ? -1 : rr < 60 ? I32 : rr < 75 ? I64 : rr < 85 ? N8 : rr < 95 ? BOOL : F64;
        lid = 0; cur_ret = f.ret;
        out += f.name; out += '(';
        push_scope();
        for (int i = 0; i < np; i++) {
            if (i) out += ", ";
            std::string pn = local_name();
            out += pn; out += ' '; out += TY_NAME[f.params[i]];
            scopes.back().push_back({pn, f.params[i]});
        }
        out += ") "; out += f.ret < 0 ? "null" : TY_NAME[f.ret]; out += '\n';
        int n = (int)rng.range(3, 30);
        if (rng.chance(50)) n = (int)rng.range(3, 10);
        push_scope();
        for (int i = 0; i < n; i++) statement(1, false, 0);
        if (decline_pct && rng.chance((uint32_t)decline_pct)) decliner();      // (never consulted without the flag: the plain generator's output is unchanged)
        if (f.ret >= 0) { tabs(1); out += "return "; expr((Ty)f.ret, (int)rng.below(4), true); out += '\n'; }
        pop_scope();
        pop_scope();
        funcs.push_back(std::move(f));
    }
    // flags bit 1: one statement that the node-parallel lowering declines (the function then goes through the sequential machine):
    // a shadowed name, an undefined function (a diagnostic), a call through a local of function type, a parenthesised callee
    // (CallOrIndex), a `proc` expression — what hand-written code has and the plain generator never emits
    int decline_pct = 0;
    void call_args(const Func& f) {
        static const char* const LIT[] = {"1", "1i64", "1n8", "true"};
        out += '(';
        for (size_t i = 0; i < f.params.size(); i++) { if (i) out += ", "; out += LIT[(int)f.params[i] < 4 ? (int)f.params[i] : 0]; }
        out += ")\n";
    }
    void decliner() {
        uint32_t k = rng.below(5);
        const Func* callee = funcs.empty() ? nullptr : &funcs[rng.below((uint32_t)funcs.size())];
        if (callee) for (Ty t : callee->params) if ((int)t >= 4) { callee = nullptr; break; }
        if (k >= 2 && !callee) k = 1;
        tabs(1);
        switch (k) {
            case 0: out += "var "; out += scopes[scopes.size() - 2][0].name; out += " = 0\n"; break;
            case 1: out += "nosuch_q(1)\n"; break;
            case 2: out += "var cb_q = "; out += callee->name; out += '\n'; tabs(1); out += "cb_q"; call_args(*callee); break;
            case 3: out += '('; out += callee->name; out += ')'; call_args(*callee); break;
            default: {
                out += "var pr_q = proc "; out += callee->name; out += '(';
                for (size_t i = 0; i < callee->params.size(); i++) { if (i) out += ", "; out += TY_NAME[callee->params[i]]; }
                out += ")\n";
                break;
            }
        }
    }
    void global_def() {
        uint32_t r = rng.below(100);
        static const Ty GT[] = {I32, I64, N8};
        Ty t = GT[rng.below(3)];
        if (r < 45) {
            std::string name = def_name("g");
            out += "var "; out += name; out += ' '; out += TY_NAME[t];
            if (rng.chance(70)) { out += " = "; if (t != N8 && rng.chance(30)) out += '-'; int_literal(t, false); }
            out += '\n';
            gvars.push_back({name, t});
        } else if (r < 75) {
            std::string name = def_name("K");
            out += "const "; out += name; out += ' '; out += TY_NAME[t]; out += " = ";
            if (t != N8 && rng.chance(30)) out += '-';
            int_literal(t, false); out += '\n';
            gconsts.push_back({name, t});
        } else if (r < 90) {
            static const char* TS[] = {"@int32", "{} nat8", "{8} int64", "(int32) bool", "@mut {4} nat8", "record { a int32 b @int64 }"};
            out += "type "; out += def_name("T"); out += " = "; out += TS[rng.below(6)]; out += '\n';
        } else {
            Func f; f.name = def_name("ext");
            int np = (int)rng.range(0, 3);
            static const Ty PT[] = {I32, I64, N8, BOOL};
            out += "extern "; out += f.name; out += '(';
            for (int i = 0; i < np; i++) { if (i) out += ", "; Ty p = PT[rng.below(4)]; f.params.push_back(p); out += 'p'; out += std::to_string(i); out += ' '; out += TY_NAME[p]; }
            f.ret = rng.chance(20) ? -1 : (int)PT[rng.below(4)];
            out += ") "; out += f.ret < 0 ? "null" : TY_NAME[f.ret]; out += '\n';
            funcs.push_back(std::move(f));
        }
    }
    void record_def() {
        RecType r; r.name = def_name("Rec");
        int nf = (int)rng.range(1, 5);
        out += "type "; out += r.name; out += " = record {";
        bool multi = rng.chance(30);
        static const Ty FT[] = {I32, I64, N8, BOOL, F64};
        for (int i = 0; i < nf; i++) {
            Ty t = FT[rng.below(5)];
            std::string fn = std::string(word()) + std::to_string(i);
            if (multi) out += "\n\t"; else out += ' ';
            out += fn; out += ' '; out += TY_NAME[t];
            r.fields.push_back({fn, t});
        }
        out += multi ? "\n}\n" : " }\n";
        recs.push_back(std::move(r));
    }
    void definition() {
        if (rng.chance(5)) { out += "; "; out += word(); out += " section\n"; }
        if (rng.chance(5)) out += '\n';
        uint32_t r = rng.below(100);
        if (r < 70) function_def(); else if (r < 90) global_def(); else record_def();
    }

    void prelude() {
        out +=
            "; prelude: iteration protocol used by `for` (generate_ir.rs:363-469)\n"
            "type Range = record { lo int32 hi int32 }\n"
            "type RangeIter = record { cur int32 hi int32 }\n"
            "operator for(r Range) RangeIter\n"
            "\treturn record RangeIter { cur = r.lo hi = r.hi }\n"
            "operator continue(it @mut RangeIter, x @mut int32) bool\n"
            "\tif it@.cur < it@.hi\n"
            "\t\tx@ = it@.cur\n"
            "\t\tit@.cur = it@.cur + 1\n"
            "\t\treturn true\n"
            "\treturn false\n";
    }

    // C3 deep regions: 1000-deep if pyramid, 1000-deep parentheses, 1000-deep `@` type.
    void deep_regions(int depth) {
        std::string n1 = def_name("deepif");
        out += n1; out += "(a int32) int32\n";
        for (int k = 1; k <= depth; k++) { tabs(k); out += "if a < "; out += std::to_string(k); out += '\n'; }
        tabs(depth + 1); out += "return 0\n";
        out += "\treturn a\n";
        std::string n2 = def_name("deeppar");
        out += n2; out += "(a int32) int32\n\treturn ";
        out.append((size_t)depth, '('); out += "a + 1"; out.append((size_t)depth, ')'); out += '\n';
        std::string n3 = def_name("deepty");
        out += n3; out += "(p "; out.append((size_t)depth, '@'); out += "int32) null\n\tvar q = p\n";
    }
};

void pad_to(std::string& s, size_t target) {
    // fill with comment lines so the file is exactly `target` bytes
    while (s.size() < target) {
        size_t room = target - s.size();
        if (room == 1) { s += '\n'; break; }
        size_t line = room < 80 ? room : 80;
        s += ';';
        s.append(line - 2, '.');
        s += '\n';
    }
}

std::string generate_block(uint64_t seed, uint32_t block, size_t target, bool prelude_here, bool deep, int deep_depth, int decline_pct = 0) {
    Gen g(seed * 0x9E3779B97F4A7C15ull + 0x1234567ull * (block + 1), block == 0 ? std::string("") : "b" + std::to_string(block) + "_", true);
    g.decline_pct = decline_pct;
    g.out.reserve(target + 4096);
    if (prelude_here) g.prelude();
    if (deep) g.deep_regions(deep_depth);
    for (;;) {
        size_t before = g.out.size();
        g.definition();
        if (g.out.size() > target) { g.out.resize(before); break; }
    }
    pad_to(g.out, target);
    return g.out;
}

}  // namespace

extern "C" {

// flags: bit0 = inject the C3 deep regions (depth 1000) into block 0; bit1 = 5 % of the functions carry one statement that the
// node-parallel lowering declines (Gen::decliner): the "fallback cliff" workload.
// Generates exactly `target` bytes into out (cap >= target).  Blocks of `block_bytes` are generated
// independently (seed, block index) so that the result does not depend on the thread count.
uint64_t opusgen_generate(uint64_t seed, uint64_t target, uint32_t flags, uint64_t block_bytes, char* out, uint64_t cap, int threads) {
    if (cap < target) return 0;
    if (block_bytes == 0) block_bytes = 64ull << 20;
    uint64_t nblocks = (target + block_bytes - 1) / block_bytes;
    if (nblocks == 0) return 0;
    if (threads < 1) threads = 1;
    std::vector<std::thread> pool;
    std::vector<int> ok(nblocks, 1);
    auto work = [&](int tid) {
        for (uint64_t b = (uint64_t)tid; b < nblocks; b += (uint64_t)threads) {
            uint64_t lo = b * block_bytes, hi = lo + block_bytes < target ? lo + block_bytes : target;
            std::string s = generate_block(seed, (uint32_t)b, (size_t)(hi - lo), b == 0, b == 0 && (flags & 1), 1000, (flags & 2) ? 5 : 0);
            if (s.size() != hi - lo) { ok[b] = 0; continue; }
            memcpy(out + lo, s.data(), s.size());
        }
    };
    if (threads == 1) work(0);
    else { for (int t = 0; t < threads; t++) pool.emplace_back(work, t); for (auto& t : pool) t.join(); }
    for (auto v : ok) if (!v) return 0;
    return target;
}

// Blocks [first_block, first_block + n_blocks) of the source opusgen_generate(seed, ...) would produce with the same
// block size: lets every rank of a multi-GPU run materialise only ITS part of one huge logical source.
uint64_t opusgen_generate_range(uint64_t seed, uint64_t first_block, uint64_t n_blocks, uint64_t block_bytes, uint32_t flags, char* out, uint64_t cap, int threads) {
    if (block_bytes == 0) block_bytes = 64ull << 20;
    if (cap < n_blocks * block_bytes) return 0;
    if (threads < 1) threads = 1;
    std::vector<std::thread> pool;
    std::vector<int> ok(n_blocks, 1);
    auto work = [&](int tid) {
        for (uint64_t i = (uint64_t)tid; i < n_blocks; i += (uint64_t)threads) {
            uint64_t b = first_block + i;
            std::string s = generate_block(seed, (uint32_t)b, (size_t)block_bytes, b == 0, b == 0 && (flags & 1), 1000, (flags & 2) ? 5 : 0);
            if (s.size() != block_bytes) { ok[i] = 0; continue; }
            memcpy(out + i * block_bytes, s.data(), s.size());
        }
    };
    if (threads == 1) work(0);
    else { for (int t = 0; t < threads; t++) pool.emplace_back(work, t); for (auto& t : pool) t.join(); }
    for (auto v : ok) if (!v) return 0;
    return n_blocks * block_bytes;
}

// Generates at least `min_lines` lines cut at a definition boundary (C1).  Returns the length.
uint64_t opusgen_generate_lines(uint64_t seed, uint64_t min_lines, char* out, uint64_t cap) {
    Gen g(seed * 0x9E3779B97F4A7C15ull + 0x1234567ull, "", true);
    g.prelude();
    uint64_t lines = 0; size_t counted = 0;
    while (lines < min_lines) {
        g.definition();
        for (; counted < g.out.size(); counted++) if (g.out[counted] == '\n') lines++;
    }
    if (g.out.size() > cap) return 0;
    memcpy(out, g.out.data(), g.out.size());
    return g.out.size();
}

// Single-fault injector (C5): rewrites `src` (len bytes) in place into out (cap >= len + 16).
// kind 0..9 = {illegal char '#', unterminated string, bad escape \q, bad \xZ, missing ')', extra ')',
//              missing comma, InvalidNumericSize (5i7), double indent, keyword in identifier position}
// Returns the new length, or 0 when no suitable site exists.
uint64_t opusgen_inject(const char* src, uint64_t len, uint64_t seed, uint32_t kind, char* out, uint64_t cap) {
    Rng rng(seed ^ 0xFA17ull);
    std::string s(src, len);
    auto sites = [&](auto pred) { std::vector<size_t> v; for (size_t i = 0; i < s.size(); i++) if (pred(i)) v.push_back(i); return v; };
    auto in_code_line = [&](size_t i) {   // crude: not inside a comment or string on this line
        size_t b = i; while (b > 0 && s[b - 1] != '\n') b--;
        bool str = false;
        for (size_t k = b; k < i; k++) { if (s[k] == '\\' && str) { k++; continue; } if (s[k] == '"') str = !str; else if (s[k] == ';' && !str) return false; }
        return !str;
    };
    std::vector<size_t> v;
    std::string ins;
    switch (kind) {
        case 0: v = sites([&](size_t i) { return s[i] == ' ' && in_code_line(i); }); if (v.empty()) return 0; s.insert(v[rng.below((uint32_t)v.size())], " #"); break;
        case 1: v = sites([&](size_t i) { return s[i] == '=' && i + 1 < s.size() && s[i + 1] == ' ' && in_code_line(i); }); if (v.empty()) return 0; s.insert(v[rng.below((uint32_t)v.size())] + 2, "\"unterminated "); break;
        case 2: v = sites([&](size_t i) { return s[i] == '=' && i + 1 < s.size() && s[i + 1] == ' ' && i > 0 && s[i - 1] == ' ' && in_code_line(i); }); if (v.empty()) return 0; s.insert(v[rng.below((uint32_t)v.size())] + 2, "\"a\\qb\" + "); break;
        case 3: v = sites([&](size_t i) { return s[i] == '=' && i + 1 < s.size() && s[i + 1] == ' ' && i > 0 && s[i - 1] == ' ' && in_code_line(i); }); if (v.empty()) return 0; s.insert(v[rng.below((uint32_t)v.size())] + 2, "\"a\\xZ1\" + "); break;
        case 4: v = sites([&](size_t i) { return s[i] == ')' && in_code_line(i); }); if (v.empty()) return 0; s.erase(v[rng.below((uint32_t)v.size())], 1); break;
        case 5: v = sites([&](size_t i) { return s[i] == ')' && in_code_line(i); }); if (v.empty()) return 0; s.insert(v[rng.below((uint32_t)v.size())], ")"); break;
        case 6: v = sites([&](size_t i) { return s[i] == ',' && in_code_line(i); }); if (v.empty()) return 0; s.erase(v[rng.below((uint32_t)v.size())], 1); break;
        case 7: v = sites([&](size_t i) { return s[i] == '=' && i + 1 < s.size() && s[i + 1] == ' ' && i > 0 && s[i - 1] == ' ' && in_code_line(i); }); if (v.empty()) return 0; s.insert(v[rng.below((uint32_t)v.size())] + 2, "5i7 + "); break;
        case 8: v = sites([&](size_t i) { return s[i] == '\t' && i > 0 && s[i - 1] == '\n'; }); if (v.empty()) return 0; s.insert(v[rng.below((uint32_t)v.size())], "\t\t"); break;
        case 9: v = sites([&](size_t i) { return i + 4 < s.size() && s[i] == 'v' && s[i + 1] == 'a' && s[i + 2] == 'r' && s[i + 3] == ' ' && i > 0 && s[i - 1] == '\t'; }); if (v.empty()) return 0; s.insert(v[rng.below((uint32_t)v.size())] + 4, "while "); break;
        default: return 0;
    }
    if (s.size() > cap) return 0;
    memcpy(out, s.data(), s.size());
    return s.size();
}

}  // extern "C"

#ifdef OPUSGEN_MAIN
#include <cstdlib>
// opus_gen <seed> <bytes> [flags] > out.opus      |  opus_gen --lines <seed> <lines>
int main(int argc, char** argv) {
    if (argc >= 4 && !strcmp(argv[1], "--lines")) {
        std::vector<char> buf(64u << 20);
        uint64_t n = opusgen_generate_lines(strtoull(argv[2], 0, 10), strtoull(argv[3], 0, 10), buf.data(), buf.size());
        fwrite(buf.data(), 1, n, stdout);
        return 0;
    }
    if (argc < 3) { fprintf(stderr, "usage: opus_gen <seed> <bytes> [flags]\n"); return 2; }
    uint64_t seed = strtoull(argv[1], 0, 10), bytes = strtoull(argv[2], 0, 10);
    uint32_t flags = argc > 3 ? (uint32_t)atoi(argv[3]) : 0;
    std::vector<char> buf(bytes);
    uint64_t n = opusgen_generate(seed, bytes, flags, 0, buf.data(), buf.size(), 8);
    fwrite(buf.data(), 1, n, stdout);
    return n == bytes ? 0 : 1;
}
#endif

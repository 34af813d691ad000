// Data-parallel parser for gfx950 (the hot path; parse.h's SegParser is the exact fallback for what this path declines).
//
// parse.rs:723-1355 is an LL(1) recursive descent.  Here it is two kernels over the token array:
//
//   k_pp_roles : one LANE per top-level segment runs a table-driven pushdown automaton over the token kinds.  The table is
//                derived from the grammar of parse.rs: a state is a position in a grammar rule, the stack
//                holds one entry per open bracket / indented block (the state to return to).  It accepts exactly the segments
//                that parse without error and without the few constructs listed under "declined" below, and writes one
//                8-byte record per token: the token's ROLE in the grammar (what node it creates), the matching bracket, the
//                number of list items of a bracket group and the token's first node number inside the segment.  Node and
//                definition totals per segment fall out of the same pass, so the dense node array is laid out by ONE scan over
//                segments before any node is written (no sparse region, no compaction).
//   k_pp_build : one THREAD per token writes the final node records of the nodes its token creates, at their final place.
//                Children, siblings and spans are found by short walks over the items of the token's own bracket level
//                (groups are skipped through the matching-bracket links); expression trees are the left-associative
//                precedence trees of parse.rs:815-869 (parse_expression + correct_precedence), built by the
//                previous-smaller-value rule: the parent of an operator is the nearer of the first operator to its left with
//                lower and the first to its right with lower-or-equal precedence (parse.rs:205-222), `not` being a binary
//                operator with a virtual Null left operand (parse.rs:1027-1038, Q9); spans are hulls of leaf spans, which is
//                what the rotation's re-spanning (parse.rs:852-865) amounts to.
//
// Declined (the segment goes to SegParser, which is exact for everything): any syntax or lexical error, a block that is not
// indented (parse.rs:1245-1248 single-statement blocks), a prefix operator directly followed by `not` (the one case where the
// flat precedence view differs, parse.rs:993-1012 + 1029), segments beyond 65535 tokens / nodes / list items.
#pragma once
#include <hip/hip_runtime.h>
#include <array>
#include <vector>
#include "unit.h"

namespace opusfe {
namespace pp {

// ---- roles: what a token is in the grammar -------------------------------------------------------------------------
enum Role : uint8_t {
    R_NONE = 0,
    R_NAME, R_OPNAME1, R_OPNAME2,                    // parse_identifier parse.rs:773-801 in a slot that creates no node of its own
    // definitions parse.rs:1262-1337
    R_D_EXTERN, R_D_VAR, R_D_CONST, R_D_TYPE, R_D_IMPORT, R_D_FN /* first token of a function's name */,
    R_D_STRING, R_D_EQ /* `=` of a var / const definition */, R_D_TEQ /* `=` of a type definition */,
    R_SIG_LP, R_SIG_RP_FN, R_SIG_RP_EXT, R_SIG_COMMA, R_SIG_ARG /* first token of an argument name */,
    // types parse.rs:899-969
    R_TY_NAME, R_TY_NULL, R_TY_AT, R_TY_MUT, R_TY_LP, R_TY_RP, R_TY_COMMA, R_TY_RECORD, R_TY_REC_LB, R_TY_REC_RB, R_TY_FIELD,
    R_TY_LB, R_TY_ARR_INT, R_TY_ARR_RB,
    // expressions parse.rs:971-1124, 815-835
    R_E_LEAF, R_E_LP, R_E_RP, R_E_PREFIX, R_E_VARDEF, R_E_NOT, R_E_SIZEOF, R_E_RECORD, R_E_REC_LB, R_E_REC_RB, R_E_FIELD, R_E_FEQ,
    R_E_PROC, R_E_PROC_LP, R_BINOP, R_POST_AT, R_POST_DOT, R_CALL_LP, R_CALL_RP, R_CALL_COMMA, R_POST_COLON,
    // statements parse.rs:1126-1260
    R_S_VAR_DEF /* var n = e */, R_S_VAR_T /* var n T [= e] */, R_S_RETURN, R_S_IFWHILE, R_S_FOR, R_S_IN, R_S_BREAKCONT, R_S_ELSE, R_S_EQ,
    R_INDENT, R_DEDENT, R_EOF,
    R_COUNT,
    R_HEAD = 0x80                                    // flag: the token also starts an expression statement (parse.rs:1216-1228)
};

// token classes: the alphabet of the automaton (token kind, refined by one token of look-back / look-ahead where the
// reference's own LL decision needs it)
enum Cls : uint8_t {
    C_LP, C_RP, C_LB, C_LBE /* `{` directly followed by `}` */, C_RB, C_COLON, C_EQ, C_MINUS, C_BINOP, C_AT, C_DOT, C_COMMA,
    C_INT, C_LIT, C_STRING, C_IDENT, C_INDENT, C_DEDENT,
    C_VAR, C_VAR_EQ /* `var <name> =` parse.rs:1134-1139 */, C_CONST, C_RETURN, C_IFWHILE, C_ELSE, C_FOR, C_BREAKCONT, C_EXTERN,
    C_PREFIXKW, C_MUT, C_NULL, C_BOOL, C_TYPE, C_RECORD, C_IN, C_NOT, C_SIZEOF, C_IMPORT, C_PROC, C_OPERATOR,
    C_OPNAME2 /* token after `operator` that parse_identifier accepts */, C_EOF, C_BAD,
    C_END /* pseudo class: end of the segment; an entry without A_ERR means "accepting" */,
    NCLS
};

enum Act : uint8_t { A_PUSH = 1, A_POP = 2, A_ITEM = 4, A_ERR = 8, A_SEQ_END = 16 /* the token ends the expression that is open at its level */ };
struct Entry { uint8_t next, role, act, ret; };
static_assert(sizeof(Entry) == 4, "one dword per transition");

// per token record written by k_pp_roles (token indices are relative to the segment start)
struct PRec {
    uint8_t role;       // Role | R_HEAD
    uint8_t kind;       // the token's kind (copy: k_pp_build then reads one record per token it looks at)
    uint16_t cnt;       // closing bracket / Dedent: number of list items of the group; first token of a definition: its ordinal in the segment
    uint16_t link;      // bracket tokens: the matching bracket
    uint16_t nid;       // number of the first node this token creates, relative to the segment's first node
    // binary operator / `not`: a = previous operator of the same expression, or PX_END | first token of the expression;
    //                          b = next operator, or PX_END | the token that ends the expression; c = token of the root node of the atom
    //                          to its left (binary operators); d = the same to its right
    // first token of an expression (not a `not`): a = token of the expression's root node, b = the token that ends the expression
    // `not` that starts an expression: c = token of the expression's root node (a, b, d as for operators)
    uint16_t a, b, c, d;
};
static_assert(sizeof(PRec) == 16, "one 16-byte store per token");
enum { PX_END = 0x8000 };

// device table blob: [n_states * NCLS entries][64 kind -> class bytes][128 role -> info bytes: bits 0-1 nodes the token creates, bits 2-4 item
// class of the role inside an expression (IC_*), bit 5: the token behind it starts an expression]
enum { IC_TERM, IC_OP, IC_PREFIX, IC_PRIMARY, IC_POSTFIX };
struct TableView { const Entry* t; const uint8_t* kind_class; const uint8_t* role_nodes; uint32_t start; };

// parse.rs:179-203 / 205-222
__host__ __device__ __forceinline__ int prec_of_kind(uint8_t k) {
    switch (k) {
        case T_Or: return 0; case T_And: return 1; case T_Not: return 2;
        case T_EqualsEquals: case T_LessThan: case T_GreaterThan: case T_LessThanEquals: case T_GreaterThanEquals: case T_Is: return 20;
        case T_Plus: case T_Minus: return 30;
        default: return 31;    // * / %
    }
}
__host__ __device__ __forceinline__ uint8_t op_of_kind(uint8_t k) {
    switch (k) {
        case T_Plus: return OP_Plus; case T_Minus: return OP_Minus; case T_Asterisk: return OP_Times; case T_Slash: return OP_Divide;
        case T_Percent: return OP_Modulo; case T_EqualsEquals: return OP_Equals; case T_LessThan: return OP_LessThan;
        case T_GreaterThan: return OP_GreaterThan; case T_LessThanEquals: return OP_LessThanEquals;
        case T_GreaterThanEquals: return OP_GreaterThanEquals; case T_Is: return OP_Is; case T_And: return OP_And; case T_Or: return OP_Or;
        default: return OP_Not;
    }
}
// tokens that parse_identifier accepts after `operator` (parse.rs:776-794)
__host__ __device__ __forceinline__ bool opname_ok(uint8_t k) {
    const uint64_t m = (1ull << T_Plus) | (1ull << T_Minus) | (1ull << T_Asterisk) | (1ull << T_Slash) | (1ull << T_Percent) | (1ull << T_EqualsEquals) |
                       (1ull << T_LessThan) | (1ull << T_GreaterThan) | (1ull << T_LessThanEquals) | (1ull << T_GreaterThanEquals) | (1ull << T_Is) |
                       (1ull << T_Not) | (1ull << T_For) | (1ull << T_Continue);
    return k < 64 && ((m >> k) & 1);
}

// ---- the table, derived from the grammar (host) ----------------------------------------------------------------------
struct Table {
    std::vector<uint32_t> blob;     // the device blob (see TableView)
    uint32_t n_states = 0, start = 0;
    bool ok = true;                 // false: two rules claimed the same (state, class) — a bug in the derivation below
};

struct Builder {
    std::vector<std::array<Entry, NCLS>> rows;
    bool ok = true;
    bool soft = false;              // soft: a rule that is already taken stays (the postfix / binary reading of `(` and `-` behind a complete
                                    // atom wins over "a new statement starts here", parse.rs:1087-1121 before 1216)
    int st() { std::array<Entry, NCLS> r; r.fill(Entry{0, 0, A_ERR, 0}); rows.push_back(r); return (int)rows.size() - 1; }
    void on(int s, int c, int next, uint8_t role, uint8_t act = 0, int ret = 0) {
        Entry& e = rows[(size_t)s][(size_t)c];
        if (!(e.act & A_ERR)) { if (!soft) ok = false; return; }
        e = Entry{(uint8_t)next, role, act, (uint8_t)ret};
    }
    void accept(int s) { on(s, C_END, s, R_NONE); }
    // a name (parse_identifier): IDENT, or `operator` + one of the tokens opname_ok() lists
    void name_slot(int n, int next, uint8_t role_first, uint8_t act = 0) {
        const int n2 = st();
        on(n, C_IDENT, next, role_first, act);
        on(n, C_OPERATOR, n2, role_first == R_NAME ? (uint8_t)R_OPNAME1 : role_first, act);
        on(n2, C_OPNAME2, next, R_OPNAME2);
    }
};

enum { X_STMT, X_RHS, X_COND, X_PAREN, X_ARG, X_FIELD, X_DEF, NX };
enum { K_SIGARG_F, K_SIGARG_E, K_SIGRET_F, K_SIGRET_E, K_TL, K_RT, K_DEFVAR, K_DEFCONST, K_DEFTYPE, K_STMTVAR, K_FOR, K_ATOM0, NK = K_ATOM0 + NX };

inline Table build_table() {
    Builder b;
    int A[NX], AP[NX], C[NX], VN[NX], RN[NX], RB[NX], PN[NX], PB[NX], FN[NX];
    int T[NK], TA[NK], TR[NK], after[NK];
    const int TOP = b.st();
    for (int x = 0; x < NX; x++) { A[x] = b.st(); AP[x] = b.st(); C[x] = b.st(); VN[x] = b.st(); RN[x] = b.st(); RB[x] = b.st(); PN[x] = b.st(); PB[x] = b.st(); FN[x] = b.st(); }
    for (int k = 0; k < NK; k++) { T[k] = b.st(); TA[k] = b.st(); TR[k] = b.st(); }
    const int TOP_END = b.st(), EXT_N = b.st(), EXT_LP = b.st(), DV_N = b.st(), DC_N = b.st(), DT_N = b.st(), DT_EQ = b.st(), IM_S = b.st(),
              FN_N2 = b.st(), FN_LP = b.st(), DV1 = b.st(), DC1 = b.st(), FN_BODY = b.st();
    int SIG0[2] = {b.st(), b.st()}, SIGN2[2] = {b.st(), b.st()}, SIGSEP[2] = {b.st(), b.st()};
    const int TL0 = b.st(), TLSEP = b.st(), RT0 = b.st(), RTN2 = b.st(), AE0 = b.st(), AN0 = b.st(), AN1 = b.st(), CALL0 = b.st(), RL0 = b.st(), RL_EQ = b.st();
    const int B0 = b.st(), SVD_N = b.st(), SVD_EQ = b.st(), SVT_N = b.st(), SVC = b.st(), FOR_N = b.st(), FOR1 = b.st(), FOR_IN = b.st(), AFTER_BLK = b.st(), ELSE_B = b.st();

    after[K_SIGARG_F] = SIGSEP[0]; after[K_SIGARG_E] = SIGSEP[1]; after[K_SIGRET_F] = FN_BODY; after[K_SIGRET_E] = TOP;
    after[K_TL] = TLSEP; after[K_RT] = RT0; after[K_DEFVAR] = DV1; after[K_DEFCONST] = DC1; after[K_DEFTYPE] = TOP;
    after[K_STMTVAR] = SVC; after[K_FOR] = FOR_IN;
    for (int x = 0; x < NX; x++) after[K_ATOM0 + x] = C[x];

    // parse_type parse.rs:899-969: the first token of a type in state s, continuation k
    auto type_start = [&](int s, int k, uint8_t act) {
        b.on(s, C_IDENT, after[k], R_TY_NAME, act);
        b.on(s, C_NULL, after[k], R_TY_NULL, act);
        b.on(s, C_AT, TA[k], R_TY_AT, act);
        b.on(s, C_LP, TL0, R_TY_LP, act | A_PUSH, T[k]);             // (types) return-type
        b.on(s, C_RECORD, TR[k], R_TY_RECORD, act);
        b.on(s, C_LBE, AE0, R_TY_LB, act | A_PUSH, TA[k]);           // {} [mut] T
        b.on(s, C_LB, AN0, R_TY_LB, act | A_PUSH, T[k]);             // {N} T
    };
    for (int k = 0; k < NK; k++) {
        type_start(T[k], k, 0);
        type_start(TA[k], k, 0);
        b.on(TA[k], C_MUT, T[k], R_TY_MUT);
        b.on(TR[k], C_LB, RT0, R_TY_REC_LB, A_PUSH, after[k]);
        b.on(TR[k], C_LBE, RT0, R_TY_REC_LB, A_PUSH, after[k]);
    }
    // parse_atom parse.rs:971-1086: the first token of an atom in state s, expression context x
    auto atom_start = [&](int s, int x, uint8_t act, bool head, bool with_var, bool with_not) {
        const uint8_t f = head ? (uint8_t)R_HEAD : (uint8_t)0;
        const int leaf[] = {C_INT, C_LIT, C_STRING, C_IDENT, C_BOOL, C_NULL};
        for (int c : leaf) b.on(s, c, C[x], R_E_LEAF | f, act);
        b.on(s, C_LP, A[X_PAREN], R_E_LP | f, act | A_PUSH, C[x]);
        const int pre[] = {C_MINUS, C_PREFIXKW, C_MUT};
        for (int c : pre) b.on(s, c, AP[x], R_E_PREFIX | f, act);
        if (with_var) { b.on(s, C_VAR, VN[x], R_E_VARDEF | f, act); b.on(s, C_VAR_EQ, VN[x], R_E_VARDEF | f, act); }
        if (with_not) b.on(s, C_NOT, A[x], R_E_NOT | f, act);
        b.on(s, C_SIZEOF, T[K_ATOM0 + x], R_E_SIZEOF | f, act);
        b.on(s, C_RECORD, RN[x], R_E_RECORD | f, act);
        b.on(s, C_PROC, PN[x], R_E_PROC | f, act);
    };
    // parse_statement parse.rs:1126-1235 + the end of a block parse.rs:1251-1257
    auto stmt_start = [&](int s, bool behind_atom) {
        b.soft = behind_atom;
        b.on(s, C_VAR_EQ, SVD_N, R_S_VAR_DEF, A_ITEM);
        b.on(s, C_VAR, SVT_N, R_S_VAR_T, A_ITEM);
        b.on(s, C_RETURN, A[X_RHS], R_S_RETURN, A_ITEM);
        b.on(s, C_IFWHILE, A[X_COND], R_S_IFWHILE, A_ITEM);
        b.on(s, C_FOR, FOR_N, R_S_FOR, A_ITEM);
        b.on(s, C_BREAKCONT, B0, R_S_BREAKCONT, A_ITEM);
        atom_start(s, X_STMT, A_ITEM, true, false, true);
        b.on(s, C_DEDENT, 0, R_DEDENT, A_POP);
        b.soft = false;
    };
    // parse_definition parse.rs:1262-1337
    auto top_start = [&](int s) {
        b.on(s, C_EXTERN, EXT_N, R_D_EXTERN, A_ITEM);
        b.on(s, C_VAR, DV_N, R_D_VAR, A_ITEM); b.on(s, C_VAR_EQ, DV_N, R_D_VAR, A_ITEM);
        b.on(s, C_CONST, DC_N, R_D_CONST, A_ITEM);
        b.on(s, C_TYPE, DT_N, R_D_TYPE, A_ITEM);
        b.on(s, C_IMPORT, IM_S, R_D_IMPORT, A_ITEM);
        b.on(s, C_IDENT, FN_LP, R_D_FN, A_ITEM);
        b.on(s, C_OPERATOR, FN_N2, R_D_FN, A_ITEM);
        b.on(s, C_EOF, TOP_END, R_EOF);
        b.accept(s);
    };

    // ---- top level ----
    top_start(TOP); top_start(DV1); top_start(C[X_DEF]);
    b.accept(TOP_END);
    b.name_slot(EXT_N, EXT_LP, R_NAME);
    b.on(EXT_LP, C_LP, SIG0[1], R_SIG_LP, A_PUSH, T[K_SIGRET_E]);
    b.name_slot(DV_N, T[K_DEFVAR], R_NAME);
    b.on(DV1, C_EQ, A[X_DEF], R_D_EQ);
    b.name_slot(DC_N, T[K_DEFCONST], R_NAME);
    b.on(DC1, C_EQ, A[X_DEF], R_D_EQ);
    b.name_slot(DT_N, DT_EQ, R_NAME);
    b.on(DT_EQ, C_EQ, T[K_DEFTYPE], R_D_TEQ);
    b.on(IM_S, C_STRING, TOP, R_D_STRING);
    b.on(FN_N2, C_OPNAME2, FN_LP, R_OPNAME2);
    b.on(FN_LP, C_LP, SIG0[0], R_SIG_LP, A_PUSH, T[K_SIGRET_F]);
    b.on(FN_BODY, C_INDENT, B0, R_INDENT, A_PUSH, TOP);
    // ---- signature arguments parse.rs:1339-1355 (parse_comma_separated parse.rs:871-897) ----
    for (int v = 0; v < 2; v++) {
        const uint8_t rp = v == 0 ? R_SIG_RP_FN : R_SIG_RP_EXT;
        const int k = v == 0 ? K_SIGARG_F : K_SIGARG_E;
        b.on(SIG0[v], C_RP, 0, rp, A_POP);
        b.on(SIG0[v], C_IDENT, T[k], R_SIG_ARG, A_ITEM);
        b.on(SIG0[v], C_OPERATOR, SIGN2[v], R_SIG_ARG, A_ITEM);
        b.on(SIGN2[v], C_OPNAME2, T[k], R_OPNAME2);
        b.on(SIGSEP[v], C_COMMA, SIG0[v], R_SIG_COMMA);
        b.on(SIGSEP[v], C_RP, 0, rp, A_POP);
    }
    // ---- type lists, record types, array braces ----
    b.on(TL0, C_RP, 0, R_TY_RP, A_POP);
    type_start(TL0, K_TL, A_ITEM);
    b.on(TLSEP, C_COMMA, TL0, R_TY_COMMA);
    b.on(TLSEP, C_RP, 0, R_TY_RP, A_POP);
    b.on(RT0, C_RB, 0, R_TY_REC_RB, A_POP);
    b.on(RT0, C_IDENT, T[K_RT], R_TY_FIELD, A_ITEM);
    b.on(RT0, C_OPERATOR, RTN2, R_TY_FIELD, A_ITEM);
    b.on(RTN2, C_OPNAME2, T[K_RT], R_OPNAME2);
    b.on(AE0, C_RB, 0, R_TY_ARR_RB, A_POP);
    b.on(AN0, C_INT, AN1, R_TY_ARR_INT);
    b.on(AN1, C_RB, 0, R_TY_ARR_RB, A_POP);
    // ---- expressions ----
    for (int x = 0; x < NX; x++) {
        atom_start(A[x], x, 0, false, true, true);
        atom_start(AP[x], x, 0, false, true, false);        // prefix operator then `not`: declined
        b.name_slot(VN[x], T[K_ATOM0 + x], R_NAME);
        b.name_slot(RN[x], RB[x], R_NAME);
        b.on(RB[x], C_LB, RL0, R_E_REC_LB, A_PUSH, C[x]);
        b.on(RB[x], C_LBE, RL0, R_E_REC_LB, A_PUSH, C[x]);
        b.name_slot(PN[x], PB[x], R_NAME);
        b.on(PB[x], C_LP, TL0, R_E_PROC_LP, A_PUSH, C[x]);
        // postfix loop parse.rs:1087-1121 and the binary operator of parse_expression parse.rs:821-824
        b.on(C[x], C_AT, C[x], R_POST_AT);
        b.on(C[x], C_DOT, FN[x], R_POST_DOT);
        b.name_slot(FN[x], C[x], R_NAME);
        b.on(C[x], C_LP, CALL0, R_CALL_LP, A_PUSH, C[x]);
        b.on(C[x], C_COLON, T[K_ATOM0 + x], R_POST_COLON);
        b.on(C[x], C_BINOP, A[x], R_BINOP);
        b.on(C[x], C_MINUS, A[x], R_BINOP);
    }
    b.on(C[X_PAREN], C_RP, 0, R_E_RP, A_POP);
    b.on(CALL0, C_RP, 0, R_CALL_RP, A_POP);
    atom_start(CALL0, X_ARG, A_ITEM, false, true, true);
    b.on(C[X_ARG], C_COMMA, CALL0, R_CALL_COMMA);
    b.on(C[X_ARG], C_RP, 0, R_CALL_RP, A_POP);
    b.on(RL0, C_RB, 0, R_E_REC_RB, A_POP);
    b.on(RL0, C_IDENT, RL_EQ, R_E_FIELD, A_ITEM);
    b.on(RL_EQ, C_EQ, A[X_FIELD], R_E_FEQ);
    b.on(C[X_FIELD], C_IDENT, RL_EQ, R_E_FIELD, A_ITEM);
    b.on(C[X_FIELD], C_RB, 0, R_E_REC_RB, A_POP);
    // ---- blocks ----
    stmt_start(B0, false); stmt_start(SVC, false); stmt_start(AFTER_BLK, false);
    b.on(C[X_STMT], C_EQ, A[X_RHS], R_S_EQ);
    stmt_start(C[X_STMT], true); stmt_start(C[X_RHS], true);
    b.name_slot(SVD_N, SVD_EQ, R_NAME);
    b.on(SVD_EQ, C_EQ, A[X_RHS], R_S_EQ);
    b.name_slot(SVT_N, T[K_STMTVAR], R_NAME);
    b.on(SVC, C_EQ, A[X_RHS], R_S_EQ);
    b.name_slot(FOR_N, FOR1, R_NAME);
    b.on(FOR1, C_IN, A[X_COND], R_S_IN);
    type_start(FOR1, K_FOR, 0);
    b.on(FOR_IN, C_IN, A[X_COND], R_S_IN);
    b.on(C[X_COND], C_INDENT, B0, R_INDENT, A_PUSH, AFTER_BLK);
    b.on(AFTER_BLK, C_ELSE, ELSE_B, R_S_ELSE);
    b.on(ELSE_B, C_INDENT, B0, R_INDENT, A_PUSH, B0);

    for (int x = 0; x < NX; x++)
        for (int c = 0; c < NCLS; c++) {
            Entry& e = b.rows[(size_t)C[x]][(size_t)c];
            if ((e.act & A_ERR) || c == C_AT || c == C_DOT || c == C_LP || c == C_COLON || c == C_BINOP || c == C_MINUS) continue;
            e.act |= A_SEQ_END;
        }
    Table t;
    t.ok = b.ok && b.rows.size() <= 255;
    t.n_states = (uint32_t)b.rows.size();
    t.start = (uint32_t)TOP;
    t.blob.assign((size_t)t.n_states * NCLS + 16 + 32, 0);
    for (size_t s = 0; s < b.rows.size(); s++)
        for (size_t c = 0; c < NCLS; c++) { const Entry e = b.rows[s][c]; uint32_t w; memcpy(&w, &e, 4); t.blob[s * NCLS + c] = w; }
    uint8_t* kc = (uint8_t*)(t.blob.data() + (size_t)t.n_states * NCLS);
    uint8_t* rn = kc + 64;
    for (int k = 0; k < 64; k++) kc[k] = C_BAD;
    kc[T_LeftParenthesis] = C_LP; kc[T_RightParenthesis] = C_RP; kc[T_LeftBrace] = C_LB; kc[T_RightBrace] = C_RB; kc[T_Colon] = C_COLON; kc[T_Equals] = C_EQ;
    kc[T_Minus] = C_MINUS; kc[T_At] = C_AT; kc[T_Dot] = C_DOT; kc[T_Comma] = C_COMMA; kc[T_Integer] = C_INT; kc[T_Float] = C_LIT; kc[T_String] = C_STRING;
    kc[T_Identifier] = C_IDENT; kc[T_Indent] = C_INDENT; kc[T_Dedent] = C_DEDENT; kc[T_Var] = C_VAR; kc[T_Const] = C_CONST; kc[T_Return] = C_RETURN;
    kc[T_If] = C_IFWHILE; kc[T_While] = C_IFWHILE; kc[T_Else] = C_ELSE; kc[T_For] = C_FOR; kc[T_Break] = C_BREAKCONT; kc[T_Continue] = C_BREAKCONT;
    kc[T_Extern] = C_EXTERN; kc[T_Ref] = C_PREFIXKW; kc[T_Refs] = C_PREFIXKW; kc[T_Muts] = C_PREFIXKW; kc[T_Mut] = C_MUT; kc[T_Null] = C_NULL;
    kc[T_False] = C_BOOL; kc[T_True] = C_BOOL; kc[T_Type] = C_TYPE; kc[T_Record] = C_RECORD; kc[T_In] = C_IN; kc[T_Not] = C_NOT;
    kc[T_Sizeof] = C_SIZEOF; kc[T_Alignof] = C_SIZEOF; kc[T_Import] = C_IMPORT; kc[T_Library] = C_IMPORT; kc[T_Proc] = C_PROC; kc[T_Operator] = C_OPERATOR;
    kc[T_EndOfFile] = C_EOF;
    const uint8_t binops[] = {T_EqualsEquals, T_LessThan, T_GreaterThan, T_LessThanEquals, T_GreaterThanEquals, T_Plus, T_Asterisk, T_Slash, T_Percent, T_Is, T_And, T_Or};
    for (uint8_t k : binops) kc[k] = C_BINOP;
    const uint8_t one[] = {R_D_EXTERN, R_D_VAR, R_D_CONST, R_D_TYPE, R_D_IMPORT, R_D_FN, R_SIG_ARG, R_TY_NAME, R_TY_NULL, R_TY_AT, R_TY_LP, R_TY_RECORD, R_TY_FIELD,
                           R_TY_LB, R_E_LEAF, R_E_LP, R_E_PREFIX, R_E_VARDEF, R_E_SIZEOF, R_E_FIELD, R_E_PROC, R_BINOP, R_POST_AT, R_POST_DOT, R_CALL_LP,
                           R_POST_COLON, R_S_VAR_DEF, R_S_RETURN, R_S_IFWHILE, R_S_FOR, R_S_BREAKCONT, R_INDENT};
    for (uint8_t r : one) rn[r] = 1;
    rn[R_E_NOT] = 2; rn[R_E_RECORD] = 2; rn[R_S_VAR_T] = 2;
    rn[R_BINOP] |= IC_OP << 2; rn[R_E_NOT] |= IC_OP << 2; rn[R_E_PREFIX] |= IC_PREFIX << 2;
    const uint8_t prim[] = {R_E_LEAF, R_E_LP, R_E_VARDEF, R_E_SIZEOF, R_E_RECORD, R_E_PROC};
    for (uint8_t r : prim) rn[r] |= IC_PRIMARY << 2;
    const uint8_t post[] = {R_POST_AT, R_POST_DOT, R_CALL_LP, R_POST_COLON};
    for (uint8_t r : post) rn[r] |= IC_POSTFIX << 2;
    const uint8_t intro[] = {R_S_RETURN, R_S_IFWHILE, R_S_IN, R_S_EQ, R_D_EQ, R_E_LP, R_CALL_LP, R_CALL_COMMA, R_E_FEQ};
    for (uint8_t r : intro) rn[r] |= 1 << 5;
    return t;
}
inline const Table& table() { static const Table t = build_table(); return t; }

__host__ __device__ __forceinline__ TableView table_view(const uint32_t* blob, uint32_t n_states, uint32_t start) {
    TableView v;
    v.t = (const Entry*)blob;
    v.kind_class = (const uint8_t*)(blob + (size_t)n_states * NCLS);
    v.role_nodes = v.kind_class + 64;
    v.start = start;
    return v;
}

// ---- k_pp_roles -------------------------------------------------------------------------------------------------------
// per-segment result words (parse.h seg_info): see SI_* there.  A declined segment takes its place in the fallback list and
// reserves its sparse node / frame regions for SegParser with two atomics.
//
// Besides the role of every token the lane keeps, per bracket level, what a sequential parser knows for free and a parallel one
// would have to search for: the expression that is open at this level (its first token, its operators in order, the rightmost
// operator of lowest precedence = its root, parse.rs:837-869) and the atom that is being read (first prefix operator, newest
// postfix node).  They go into the token records as links between operators, so that k_pp_build's previous-smaller-value search
// is a walk over operators only.
struct StackEnt { uint16_t opener, count, seq_head, last_op, best_op, first_prefix, top; uint8_t ret, best_prec; };
static_assert(sizeof(StackEnt) == 16, "stack entry");

enum { PW_DECLINED = 0, PW_SPARSE = 2 /* u64: words 2,3 */, PW_FRAMES = 4 /* u64: words 4,5 */, PW_WORDS = 8 };
enum { PP_MAX_STATES = 224 };       // LDS copy of the table: PP_MAX_STATES * NCLS dwords + the two byte tables
enum { PP_MAX_SEG_TOKENS = 0x7FFF, PP_NONE16 = 0xFFFF };

struct Rec16 { uint64_t w, x; };    // PRec as two 8-byte words

struct RolesArgs {
    const opusfe_token* tok; uint32_t n_tokens;
    const uint8_t* kinds;             // kinds[i] = tok[i].kind, one byte per token (k_tok_kinds): the automaton reads nothing else of a token
    const uint32_t* seg_start; uint32_t n_segs;
    const uint32_t* order;            // largest segments first (scan.h), may be null
    const uint32_t* table; uint32_t n_states, start;
    Rec16* prec;                      // [n_tokens] PRec
    StackEnt* stack;                  // [n_tokens / 2 + n_segs + 1]
    uint32_t* seg_info;               // [n_segs * SI_WORDS]
    uint32_t* work;                   // PW_* counters
    uint32_t* decl_list;              // [n_segs] segments for SegParser
};

// seg_info words used here (parse.h declares the enum; repeated as constants so that this header stands alone)
enum { PSI_NODE_COUNT = 1, PSI_DEF_COUNT = 2, PSI_ERR_KIND = 4, PSI_SPARSE_BASE = 11, PSI_FRAME_BASE = 12, PSI_PATH = 13, PSI_WORDS = 14 };

__host__ __device__ __forceinline__ uint64_t prec_pack(uint32_t role, uint32_t kind, uint32_t cnt, uint32_t link, uint32_t nid) {
    return (uint64_t)(role & 0xFF) | ((uint64_t)(kind & 0xFF) << 8) | ((uint64_t)(cnt & 0xFFFF) << 16) | ((uint64_t)(link & 0xFFFF) << 32) | ((uint64_t)(nid & 0xFFFF) << 48);
}
// 16-bit field f (0..7) of the record of token t
enum { PF_LINK = 2, PF_A = 4, PF_B = 5, PF_C = 6, PF_D = 7 };

struct RolesLds { MACHINE_STATE uint32_t tbl[PP_MAX_STATES * NCLS + 16 + 32]; };

// the kind of every token as a byte array: the automaton walks its segment's kinds only (1 byte of every 16-byte token record)
__global__ void k_tok_kinds(const opusfe_token* __restrict__ tok, uint32_t n_tokens, uint8_t* __restrict__ kinds) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_tokens) kinds[i] = tok[i].kind;
}

__global__ __launch_bounds__(256) void k_pp_roles(RolesArgs a) {
#ifdef OPUSFE_EMU
    const TableView tv = table_view(a.table, a.n_states, a.start);
#else
    {   // the table is read once per token by every lane: LDS
        const uint32_t words = a.n_states * NCLS + 16 + 32;
        for (uint32_t j = threadIdx.x; j < words; j += blockDim.x) RolesLds::tbl[j] = a.table[j];
        __syncthreads();
    }
    const TableView tv = table_view((const uint32_t*)RolesLds::tbl, a.n_states, a.start);
#endif
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= a.n_segs) return;
    const uint32_t s = a.order ? a.order[slot] : slot;
    const uint32_t tb = a.seg_start[s], te = s + 1 < a.n_segs ? a.seg_start[s + 1] : a.n_tokens;
    StackEnt* const stack = a.stack + ((uint64_t)tb / 2 + s);
    const uint32_t scap = (te - tb) / 2 + 1;
    // The lane's records leave in WHOLE 128-byte lines: the eight records of an aligned group of tokens are kept in LDS (where the
    // two-byte patches of the operator chain mostly land too) and stored back to back when the automaton moves on to the next group.
    // One 16-byte store per token per lane was 64 partly written lines per wave instruction, each of them evicted and merged
    // again several times: 12 GB written per GB of source for 3.5 GB of records.
    __shared__ Rec16 ring[8][256];
    const uint32_t lane = threadIdx.x;
    uint32_t cur_grp = tb >> 3, put_end = tb;      // absolute token indices: group in the ring, end of the records written to it
    auto flush_ring = [&]() {
        const uint32_t lo = cur_grp * 8 > tb ? cur_grp * 8 : tb;
        for (uint32_t ai = lo; ai < put_end; ai++) gst(&a.prec[ai], ring[ai & 7][lane]);
    };
    auto prec_put = [&](uint32_t rel, const Rec16& rc) {
        const uint32_t ai = tb + rel;
        if ((ai >> 3) != cur_grp) { flush_ring(); cur_grp = ai >> 3; }
        ring[ai & 7][lane] = rc;
        put_end = ai + 1;
    };
    auto prec_patch = [&](uint32_t t, uint32_t f, uint32_t v) {      // field f (16-bit words) of the record of token t (relative), written earlier
        const uint32_t ai = tb + t;
        if ((ai >> 3) == cur_grp && ai < put_end) ((uint16_t*)&ring[ai & 7][lane])[f] = (uint16_t)v;
        else gst((uint16_t*)(a.prec + ai) + f, (uint16_t)v);
    };
    uint32_t state = tv.start, depth = 0, count = 0, ncount = 0;
    uint32_t prev_kd = T_EndOfFile, prev_role = R_NONE;
    // the open expression and the atom being read at the current level
    uint32_t seq_head = PP_NONE16, last_op = PP_NONE16, best_op = PP_NONE16, best_prec = 127, first_prefix = PP_NONE16, top = PP_NONE16;
    bool ok = te - tb <= (uint32_t)PP_MAX_SEG_TOKENS;
    // token kinds are fetched eight tokens ahead (the loads of a group are independent; the automaton steps are not)
    enum { AHEAD = 8 };
    uint64_t cur = 0, nxt = 0;
    auto fetch = [&](uint32_t from) {
        uint64_t w = 0;
#pragma unroll
        for (uint32_t q = 0; q < AHEAD; q++) { const uint32_t j = from + q; const uint32_t k = j < a.n_tokens ? (uint32_t)gld(&a.kinds[j]) : (uint32_t)T_EndOfFile; w |= (uint64_t)k << (8 * q); }
        return w;
    };
    // the expression open at this level ends before token `at` (relative)
    auto close_sequence = [&](uint32_t at) {
        if (seq_head == PP_NONE16) return;
        const uint32_t atom_root = first_prefix != PP_NONE16 ? first_prefix : top;
        if (last_op != PP_NONE16) { prec_patch(last_op, PF_B, at | PX_END); prec_patch(last_op, PF_D, atom_root); }
        const uint32_t root = best_op != PP_NONE16 ? best_op : atom_root;
        if (best_prec & 0x80) prec_patch(seq_head, PF_C, root);       // the expression starts with `not`
        else { prec_patch(seq_head, PF_A, root); prec_patch(seq_head, PF_B, at); }
        seq_head = PP_NONE16;
    };
    if (ok && tb < te) { cur = fetch(tb); nxt = fetch(tb + AHEAD); }
    for (uint32_t i = tb; ok && i < te; i++) {
        const uint32_t q = (i - tb) % AHEAD, rel = i - tb;
        if (q == 0 && i != tb) { cur = nxt; nxt = fetch(i + AHEAD); }
        const uint32_t kd = (uint32_t)(cur >> (8 * q)) & 0xFF;
        const uint32_t next_kd = q + 1 < AHEAD ? (uint32_t)(cur >> (8 * (q + 1))) & 0xFF : (uint32_t)nxt & 0xFF;
        uint32_t cls = kd < 64 ? tv.kind_class[kd] : (uint32_t)C_BAD;
        if (prev_kd == T_Operator) cls = opname_ok((uint8_t)kd) ? C_OPNAME2 : C_BAD;
        else if (kd == T_LeftBrace && next_kd == T_RightBrace) cls = C_LBE;
        else if (kd == T_Var) {            // parse.rs:1134: `=` behind the name decides between a definition and a typed declaration
            const uint32_t j = i + (next_kd == T_Operator ? 3u : 2u);
            if (j < a.n_tokens && gld(&a.kinds[j]) == T_Equals) cls = C_VAR_EQ;
        }
        const Entry e = tv.t[state * NCLS + cls];
        if (e.act & A_ERR) { ok = false; break; }
        const uint32_t role = e.role & 0x7F, info = tv.role_nodes[role], ic = (info >> 2) & 7;
        const bool named_call = role == R_CALL_LP && prev_kd == T_Identifier && (prev_role & 0x7F) == R_E_LEAF;      // the Variable node becomes the call (parse.rs:1103-1106)
        const uint32_t nn = named_call ? 0u : (info & 3) + (e.role >> 7);
        uint32_t r_cnt = 0, r_link = 0, r_a = 0, r_b = 0, r_c = 0, r_d = 0;
        if (e.act & A_SEQ_END) close_sequence(rel);
        // ---- expression bookkeeping ----
        if (ic != IC_TERM && ic != IC_POSTFIX && ((e.role & R_HEAD) || ((tv.role_nodes[prev_role & 0x7F] >> 5) & 1))) {     // first token of an expression
            seq_head = rel; last_op = PP_NONE16; best_op = PP_NONE16; best_prec = role == R_E_NOT ? (127u | 0x80u) : 127u; first_prefix = PP_NONE16; top = PP_NONE16;
        }
        if (ic == IC_OP) {
            const uint32_t atom_root = first_prefix != PP_NONE16 ? first_prefix : top;
            const uint32_t p = (uint32_t)prec_of_kind((uint8_t)kd);
            r_a = last_op != PP_NONE16 ? last_op : (seq_head | PX_END);
            r_c = atom_root;
            if (last_op != PP_NONE16) { prec_patch(last_op, PF_B, rel); prec_patch(last_op, PF_D, atom_root); }
            if (p <= (best_prec & 0x7F)) { best_op = rel; best_prec = (best_prec & 0x80) | p; }
            last_op = rel; first_prefix = PP_NONE16; top = PP_NONE16;
        } else if (ic == IC_PREFIX) { if (first_prefix == PP_NONE16) first_prefix = rel; }
        else if (ic == IC_PRIMARY || (ic == IC_POSTFIX && !named_call)) top = rel;
        // ---- brackets and list items ----
        if (e.act & A_ITEM) { r_cnt = count; count++; }
        if (e.act & A_PUSH) {
            if (depth >= scap) { ok = false; break; }
            StackEnt se; se.opener = (uint16_t)rel; se.count = (uint16_t)count; se.ret = e.ret;
            se.seq_head = (uint16_t)seq_head; se.last_op = (uint16_t)last_op; se.best_op = (uint16_t)best_op; se.best_prec = (uint8_t)best_prec;
            se.first_prefix = (uint16_t)first_prefix; se.top = (uint16_t)top;
            gst(&stack[depth++], se);
            count = 0; state = e.next; seq_head = PP_NONE16;
        } else if (e.act & A_POP) {
            if (depth == 0) { ok = false; break; }
            const StackEnt se = gld(&stack[--depth]);
            r_link = se.opener; r_cnt = count;
            prec_patch(se.opener, PF_LINK, rel);
            count = se.count; state = se.ret;
            seq_head = se.seq_head; last_op = se.last_op; best_op = se.best_op; best_prec = se.best_prec; first_prefix = se.first_prefix; top = se.top;
        } else state = e.next;
        Rec16 rc; rc.w = prec_pack(e.role, kd, r_cnt, r_link, ncount);
        rc.x = (uint64_t)(r_a & 0xFFFF) | ((uint64_t)(r_b & 0xFFFF) << 16) | ((uint64_t)(r_c & 0xFFFF) << 32) | ((uint64_t)(r_d & 0xFFFF) << 48);
        prec_put(rel, rc);
        ncount += nn;
        if (ncount > 65535u || count > 65535u) { ok = false; break; }
        prev_kd = kd; prev_role = e.role;
    }
    if (ok && (depth != 0 || (tv.t[state * NCLS + C_END].act & A_ERR))) ok = false;
    if (ok) { close_sequence(te - tb); flush_ring(); }        // `var x T = e` / `const` at the end of the segment
    uint32_t* si = a.seg_info + (uint64_t)s * PSI_WORDS;
    if (ok) {
        si[PSI_NODE_COUNT] = ncount; si[PSI_DEF_COUNT] = count; si[PSI_ERR_KIND] = OPUSFE_NONE; si[PSI_PATH] = 0;
    } else {
        // SegParser takes the segment: its place in the list and its sparse node / frame regions
        const uint32_t idx = atomicAdd(&a.work[PW_DECLINED], 1u);
        a.decl_list[idx] = s;
        const unsigned long long ncap = 2ull * (te - tb) + 8ull, fcap = 2ull * (te - tb) + 16ull;
        const unsigned long long nb = atomicAdd((unsigned long long*)(a.work + PW_SPARSE), ncap);
        const unsigned long long fb = atomicAdd((unsigned long long*)(a.work + PW_FRAMES), fcap);
        si[PSI_NODE_COUNT] = 0; si[PSI_DEF_COUNT] = 0; si[PSI_ERR_KIND] = OPUSFE_NONE; si[PSI_PATH] = 1;
        si[PSI_SPARSE_BASE] = (uint32_t)(nb < 0xFFFFFFFFull ? nb : 0xFFFFFFFFull); si[PSI_FRAME_BASE] = (uint32_t)(fb < 0xFFFFFFFFull ? fb : 0xFFFFFFFFull);
    }
}

// every segment to SegParser (opusfe_set_parser_mode(ctx, 1): the sequential machine alone, for A/B tests of the two parsers)
__global__ void k_pp_all_sequential(const uint32_t* __restrict__ seg_start, uint32_t n_segs, uint32_t n_tokens, uint32_t* __restrict__ seg_info,
                                    uint32_t* __restrict__ work, uint32_t* __restrict__ decl_list) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_segs) return;
    const uint32_t tb = seg_start[s], te = s + 1 < n_segs ? seg_start[s + 1] : n_tokens;
    uint32_t* si = seg_info + (uint64_t)s * PSI_WORDS;
    const uint32_t idx = atomicAdd(&work[PW_DECLINED], 1u);
    decl_list[idx] = s;
    const unsigned long long nb = atomicAdd((unsigned long long*)(work + PW_SPARSE), 2ull * (te - tb) + 8ull);
    const unsigned long long fb = atomicAdd((unsigned long long*)(work + PW_FRAMES), 2ull * (te - tb) + 16ull);
    si[PSI_NODE_COUNT] = 0; si[PSI_DEF_COUNT] = 0; si[PSI_ERR_KIND] = OPUSFE_NONE; si[PSI_PATH] = 1;
    si[PSI_SPARSE_BASE] = (uint32_t)(nb < 0xFFFFFFFFull ? nb : 0xFFFFFFFFull); si[PSI_FRAME_BASE] = (uint32_t)(fb < 0xFFFFFFFFull ? fb : 0xFFFFFFFFull);
}

// ---- k_pp_build -------------------------------------------------------------------------------------------------------
struct NodeBase { uint32_t nodes, defs; };      // exclusive scan over segments (parse.h Cnt2)

enum { PP_WIN = 1024, PP_WIN_BACK = 256, PP_BLOCK = 256 };      // LDS window of token records around the block's 256 tokens

// the records of the tokens [lo, lo + PP_WIN) of this block: a look at a neighbour rarely leaves them
struct BuildLds { MACHINE_STATE uint64_t w[PP_WIN]; MACHINE_STATE uint64_t x[PP_WIN]; };

// fields of a token record
struct Rec {
    uint64_t w, x;
    __device__ __forceinline__ uint32_t role() const { return (uint32_t)w & 0x7F; }
    __device__ __forceinline__ bool head() const { return ((uint32_t)w & R_HEAD) != 0; }
    __device__ __forceinline__ uint32_t kind() const { return ((uint32_t)w >> 8) & 0xFF; }
    __device__ __forceinline__ uint32_t cnt() const { return (uint32_t)w >> 16; }
    __device__ __forceinline__ uint32_t link() const { return (uint32_t)(w >> 32) & 0xFFFF; }
    __device__ __forceinline__ uint32_t nid() const { return (uint32_t)(w >> 48); }
    __device__ __forceinline__ uint32_t hflag() const { return ((uint32_t)w >> 7) & 1; }
    __device__ __forceinline__ uint32_t a() const { return (uint32_t)x & 0xFFFF; }
    __device__ __forceinline__ uint32_t b() const { return ((uint32_t)x >> 16) & 0xFFFF; }
    __device__ __forceinline__ uint32_t c() const { return (uint32_t)(x >> 32) & 0xFFFF; }
    __device__ __forceinline__ uint32_t d() const { return (uint32_t)(x >> 48); }
    __device__ __forceinline__ int prec() const { return prec_of_kind((uint8_t)kind()); }
};
__device__ __forceinline__ int iclass_of(uint32_t role) {
    switch (role) {
        case R_BINOP: case R_E_NOT: return IC_OP;
        case R_E_PREFIX: return IC_PREFIX;
        case R_E_LEAF: case R_E_LP: case R_E_VARDEF: case R_E_SIZEOF: case R_E_RECORD: case R_E_PROC: return IC_PRIMARY;
        case R_POST_AT: case R_POST_DOT: case R_CALL_LP: case R_POST_COLON: return IC_POSTFIX;
        default: return IC_TERM;
    }
}

// One segment seen from one thread.
struct Walker {
    const opusfe_token* tok; const Rec16* pr;
    uint32_t tb, te, nb, wlo;                 // the segment's token range (absolute), its first node, first token of the LDS window

    __device__ __forceinline__ bool in(uint32_t j) const { return j - tb < te - tb; }
    // record of token j; outside the segment: an end-of-file record (every walk stops there)
    __device__ __forceinline__ Rec rec(uint32_t j) const {
        Rec r;
        if (!in(j)) { r.w = prec_pack(R_EOF, T_EndOfFile, 0, 0, 0); r.x = 0; return r; }
#ifndef OPUSFE_EMU
        const uint32_t o = j - wlo;
        if (o < (uint32_t)PP_WIN) { r.w = BuildLds::w[o]; r.x = BuildLds::x[o]; return r; }
#endif
        const Rec16 g = gld(pr + j);
        r.w = g.w; r.x = g.x;
        return r;
    }
    // the first word alone (role, kind, cnt, link, nid)
    __device__ __forceinline__ Rec rec1(uint32_t j) const {
        Rec r; r.x = 0;
        if (!in(j)) { r.w = prec_pack(R_EOF, T_EndOfFile, 0, 0, 0); return r; }
#ifndef OPUSFE_EMU
        const uint32_t o = j - wlo;
        if (o < (uint32_t)PP_WIN) { r.w = BuildLds::w[o]; return r; }
#endif
        r.w = gld(&pr[j].w);
        return r;
    }
    __device__ __forceinline__ uint32_t role(uint32_t j) const { return rec1(j).role(); }
    __device__ __forceinline__ bool head(uint32_t j) const { return rec1(j).head(); }
    __device__ __forceinline__ uint32_t kind(uint32_t j) const { return rec1(j).kind(); }
    __device__ __forceinline__ uint32_t link(uint32_t j) const { return in(j) ? tb + rec1(j).link() : j; }
    __device__ __forceinline__ uint32_t cnt(uint32_t j) const { return rec1(j).cnt(); }
    __device__ __forceinline__ uint32_t nid(uint32_t j) const { return in(j) ? nb + rec1(j).nid() : OPUSFE_NONE; }                               // first node of token j
    __device__ __forceinline__ uint32_t enode(uint32_t j) const { const Rec r = rec1(j); return in(j) ? nb + r.nid() + r.hflag() : OPUSFE_NONE; }   // first expression node of token j
    __device__ __forceinline__ uint32_t lo(uint32_t j) const { return in(j) ? gld(&tok[j].low) : 0u; }
    __device__ __forceinline__ uint32_t hi(uint32_t j) const { return in(j) ? gld(&tok[j].low) + gld(&tok[j].len) : 0u; }
    __device__ __forceinline__ uint32_t nl(uint32_t j) const { return kind(j) == T_Operator ? 2u : 1u; }     // tokens of the name that starts at j
    __device__ __forceinline__ uint32_t bound() const { return te - tb + 2; }                                // no walk takes more steps than the segment has tokens
    // Variable(name)(args) -> NamedCall (parse.rs:1103-1106): the `(` directly behind an identifier leaf
    __device__ __forceinline__ bool named_call(uint32_t lp) const { if (lp <= tb) return false; const Rec r = rec1(lp - 1); return r.role() == R_E_LEAF && r.kind() == T_Identifier; }

    // last token of the type that starts at t (parse.rs:899-969: types are prefix coded)
    __device__ __forceinline__ uint32_t type_end(uint32_t t) const {
        for (uint32_t g = bound(); g; g--) {
            const Rec r = rec1(t);
            switch (r.role()) {
                case R_TY_AT: t += 1; if (role(t) == R_TY_MUT) t++; break;
                case R_TY_LP: t = tb + r.link() + 1; break;
                case R_TY_RECORD: return link(t + 1);
                case R_TY_LB: t = tb + r.link() + 1; if (role(t) == R_TY_MUT) t++; break;
                default: return t;       // R_TY_NAME, R_TY_NULL
            }
        }
        return t;
    }
    // first token of the outermost type whose last token is t
    __device__ __forceinline__ uint32_t type_start(uint32_t t) const {
        uint32_t s = role(t) == R_TY_REC_RB ? link(t) - 1 : t;
        for (uint32_t g = bound(); g && s > tb; g--) {
            const uint32_t u = s - 1;
            const Rec r = rec1(u);
            if (r.role() == R_TY_AT || r.role() == R_TY_MUT) s = u;
            else if (r.role() == R_TY_ARR_RB) s = tb + r.link();
            else if (r.role() == R_TY_RP && role(tb + r.link()) == R_TY_LP) s = tb + r.link();
            else break;
        }
        return s;
    }
    // index behind the atom item (primary / postfix) that starts at k (r = its record)
    __device__ __forceinline__ uint32_t next_item(uint32_t k, Rec r) const {
        switch (r.role()) {
            case R_E_LP: case R_CALL_LP: return tb + r.link() + 1;
            case R_E_VARDEF: return type_end(k + 1 + nl(k + 1)) + 1;
            case R_E_SIZEOF: case R_POST_COLON: return type_end(k + 1) + 1;
            case R_E_RECORD: case R_E_PROC: return link(k + 1 + nl(k + 1)) + 1;
            case R_POST_DOT: return k + 1 + nl(k + 1);
            default: return k + 1;       // leaf, prefix, `@`
        }
    }
    // start of the atom item that ends right before j
    __device__ __forceinline__ uint32_t prev_item(uint32_t j) const {
        const uint32_t t = j - 1;
        const Rec r = rec1(t);
        switch (r.role()) {
            case R_E_RP: case R_CALL_RP: return tb + r.link();
            case R_E_REC_RB: case R_TY_RP: { const uint32_t lb = tb + r.link(); return lb - (role(lb - 1) == R_OPNAME2 ? 2u : 1u) - 1u; }     // record name {..} / proc name(types)
            case R_NAME: return t - 1;                           // `.` name
            case R_OPNAME2: return t - 2;                        // `.` operator X
            case R_TY_NAME: case R_TY_NULL: case R_TY_REC_RB: {  // an item that ends with a type: cast, sizeof / alignof, var name T
                const uint32_t p = type_start(t) - 1;
                const uint32_t rp = role(p);
                if (rp == R_NAME) return p - 1;
                if (rp == R_OPNAME2) return p - 2;
                return p;
            }
            default: return t;           // leaf, `@`
        }
    }
    __device__ __forceinline__ bool is_seq_head(uint32_t j) const {
        if (j <= tb || head(j)) return true;
        switch (role(j - 1)) {
            case R_S_RETURN: case R_S_IFWHILE: case R_S_IN: case R_S_EQ: case R_D_EQ: case R_E_LP: case R_CALL_LP: case R_CALL_COMMA: case R_E_FEQ: return true;
            default: return false;
        }
    }
    // node of the atom-chain element (primary or postfix item) that starts at k
    __device__ __forceinline__ uint32_t item_node(uint32_t k) const {
        const Rec r = rec1(k);
        if (r.role() == R_CALL_LP && named_call(k)) return enode(k - 1);
        return nb + r.nid() + r.hflag();
    }
    __device__ __forceinline__ uint32_t stmt_next(uint32_t term) const { const Rec r = rec1(term); return (!in(term) || r.role() == R_DEDENT) ? OPUSFE_NONE : nb + r.nid(); }
    __device__ __forceinline__ uint32_t def_next(uint32_t term) const { const Rec r = rec1(term); return (!in(term) || r.role() == R_EOF) ? OPUSFE_NONE : nb + r.nid(); }
    // first token of the primary of the atom whose postfix item starts at i
    __device__ __forceinline__ uint32_t primary_start(uint32_t i) const {
        uint32_t k = prev_item(i);
        for (uint32_t g = bound(); g && iclass_of(role(k)) == IC_POSTFIX; g--) k = prev_item(k);
        return k;
    }
    // the expression that starts at h: its root node and the token that ends it (both recorded by k_pp_roles)
    __device__ __forceinline__ void sequence(uint32_t h, uint32_t* root, uint32_t* term) const {
        Rec r = rec(h);
        if (r.role() == R_E_NOT) {      // the root is in field c; the end is behind the last operator
            *root = enode(tb + r.c());
            uint32_t t = h;
            for (uint32_t g = bound(); g && !(r.b() & PX_END); g--) { t = tb + r.b(); r = rec(t); }
            *term = tb + (r.b() & (PX_END - 1));
        } else { *root = enode(tb + r.a()); *term = tb + r.b(); }
    }
    __device__ __forceinline__ uint32_t sequence_root(uint32_t h) const { uint32_t root, term; sequence(h, &root, &term); return root; }
    // node of the operand directly to the right of the operator at o (ro = its record): the virtual Null of a `not` that follows, else the atom k_pp_roles recorded
    __device__ __forceinline__ uint32_t right_atom(uint32_t o, Rec ro) const { return role(o + 1) == R_E_NOT ? enode(o + 1) + 1 : enode(tb + ro.d()); }
    // Root of the operand to the right of the operator at o, whose precedence is px: walks the operators behind it up to the first
    // of precedence <= px (*r; else *term = end of the expression); the rightmost operator of lowest precedence passed is the
    // root (parse.rs:837-869 leaves exactly that one on top), else the single atom there.
    __device__ __forceinline__ uint32_t right_operand(uint32_t o, Rec ro, int px, uint32_t* r, uint32_t* term) const {
        uint32_t best = OPUSFE_NONE; int bestp = 1000;
        *r = OPUSFE_NONE; *term = OPUSFE_NONE;
        Rec rk = ro;
        for (uint32_t g = bound(); g; g--) {
            const uint32_t b = rk.b();
            if (b & PX_END) { *term = tb + (b & (PX_END - 1)); break; }
            const uint32_t k = tb + b;
            rk = rec(k);
            const int p = rk.prec();
            if (p <= px) { *r = k; break; }
            if (p <= bestp) { best = k; bestp = p; }
        }
        return best != OPUSFE_NONE ? enode(best) : right_atom(o, ro);
    }
    // next sibling of the root of the expression [hd, term)
    __device__ __forceinline__ uint32_t context_next(uint32_t hd, uint32_t term) const {
        if (head(hd)) return role(term) == R_S_EQ ? sequence_root(term + 1) : OPUSFE_NONE;                 // Assignment(left, right)
        switch (hd > tb ? role(hd - 1) : (uint32_t)R_EOF) {
            case R_S_IFWHILE: case R_S_IN: return nid(term);                                               // condition -> block
            case R_CALL_LP: case R_CALL_COMMA: return (role(term) == R_CALL_COMMA && role(term + 1) != R_CALL_RP) ? sequence_root(term + 1) : OPUSFE_NONE;
            default: return OPUSFE_NONE;
        }
    }
    // next sibling of the root of an atom that occupies [a, after)
    __device__ __forceinline__ uint32_t atom_next(uint32_t a, uint32_t after) const {
        const bool has_l = !is_seq_head(a);                  // then the token before the atom is an operator
        const Rec rr = rec(after);
        const bool has_r = !rr.head() && rr.role() == R_BINOP;
        if (!has_l && !has_r) return context_next(a, after);
        if (!has_r || (has_l && rec1(a - 1).prec() >= rr.prec())) return OPUSFE_NONE;                       // right operand of the operator before it
        uint32_t r2, t2;
        return right_operand(after, rr, rr.prec(), &r2, &t2);                                               // left operand of the operator behind it: the sibling is that one's right operand
    }
};

__device__ __forceinline__ void emit_node(opusfe_node* nodes, uint32_t id, uint32_t kind_, uint32_t sub, uint32_t nchild, uint32_t first_child, uint32_t next, uint32_t payload, uint32_t low, uint32_t high) {
    opusfe_node n; n.kind = (uint8_t)kind_; n.sub = (uint8_t)sub; n.nchild = (uint16_t)nchild; n.first_child = first_child; n.next = next; n.payload = payload; n.low = low; n.high = high;
    gst(&nodes[id], n);
}

// next sibling of the outermost type [ts, te_]
__device__ __forceinline__ uint32_t type_next(const Walker& W, uint32_t ts, uint32_t te_) {
    if (ts <= W.tb) return OPUSFE_NONE;
    const uint32_t p = ts - 1;
    const uint32_t rp_ = W.role(p);
    switch (rp_) {
        case R_SIG_RP_FN: return W.nid(te_ + 1);                                                          // return type -> body
        case R_TY_LP: case R_TY_COMMA: case R_E_PROC_LP: {
            if (W.role(te_ + 1) == R_TY_COMMA && W.role(te_ + 2) != R_TY_RP) return W.nid(te_ + 2);
            const uint32_t rp = W.role(te_ + 1) == R_TY_RP ? te_ + 1 : te_ + 2;
            return W.role(W.link(rp)) == R_TY_LP ? W.nid(rp + 1) : OPUSFE_NONE;                             // argument types, then the return type of a proc type
        }
        case R_NAME: case R_OPNAME2: {
            const uint32_t h = rp_ == R_OPNAME2 ? p - 1 : p;
            const uint32_t rh = W.role(h);
            if (rh != R_NAME && rh != R_OPNAME1) return OPUSFE_NONE;                                       // argument / field name
            switch (h > W.tb ? W.role(h - 1) : (uint32_t)R_EOF) {
                case R_D_VAR: return W.role(te_ + 1) == R_D_EQ ? W.sequence_root(te_ + 2) : OPUSFE_NONE;
                case R_D_CONST: case R_S_FOR: return W.sequence_root(te_ + 2);
                default: return OPUSFE_NONE;
            }
        }
        default: return OPUSFE_NONE;
    }
}

// role class of a token for k_pp_build's in-block ordering (7: the role creates no node)
__device__ __forceinline__ uint32_t build_class(uint32_t role) {
    switch (role) {
        case R_E_LEAF: return 0;
        case R_BINOP: case R_E_NOT: return 1;
        case R_POST_AT: case R_POST_DOT: case R_CALL_LP: case R_POST_COLON: case R_E_LP: case R_E_PREFIX: return 2;
        case R_S_RETURN: case R_S_VAR_DEF: case R_S_VAR_T: case R_S_IFWHILE: case R_S_FOR: case R_S_BREAKCONT: case R_INDENT: return 3;
        case R_TY_NAME: case R_TY_NULL: case R_TY_AT: case R_TY_LB: case R_TY_LP: case R_TY_RECORD: case R_TY_FIELD: case R_SIG_ARG: return 4;
        case R_D_EXTERN: case R_D_FN: case R_D_VAR: case R_D_CONST: case R_D_TYPE: case R_D_IMPORT: return 5;
        case R_E_VARDEF: case R_E_SIZEOF: case R_E_RECORD: case R_E_FIELD: case R_E_PROC: return 6;
        default: return 7;
    }
}

// One thread per token: the node records of the nodes its token creates.
__global__ __launch_bounds__(PP_BLOCK) void k_pp_build(const opusfe_token* __restrict__ tok, uint32_t n_tokens, const uint32_t* __restrict__ seg_start, uint32_t n_segs,
                                                        const Rec16* __restrict__ pr, const uint32_t* __restrict__ seg_info, const NodeBase* __restrict__ base,
                                                        opusfe_node* __restrict__ nodes, uint32_t* __restrict__ def_nodes) {
    const uint32_t b0 = blockIdx.x * PP_BLOCK;
    const uint32_t wlo = b0 >= (uint32_t)PP_WIN_BACK ? b0 - PP_WIN_BACK : 0u;
#ifndef OPUSFE_EMU
    for (uint32_t j = threadIdx.x; j < (uint32_t)PP_WIN; j += PP_BLOCK) {
        Rec16 g; g.w = 0; g.x = 0;
        if (wlo + j < n_tokens) g = gld(pr + wlo + j);
        BuildLds::w[j] = g.w; BuildLds::x[j] = g.x;
    }
    __syncthreads();
#endif
    uint32_t i = b0 + threadIdx.x;
    uint32_t s = 0;
#ifndef OPUSFE_EMU
    {   // Tokens of one role class to neighbouring lanes: the switch below is one body per role and a wave pays for every body any of
        // its lanes takes; in token order a wave holds a dozen roles.  A counting sort of the block's 256 tokens by role class (LDS
        // atomics) puts leaves next to leaves and operators next to operators; tokens that create no node drop out together.
        __shared__ uint32_t hist[8];
        __shared__ uint16_t order[PP_BLOCK];
        uint32_t cls = 7;
        if (i < n_tokens) {
            Rec r; r.w = BuildLds::w[i - wlo]; r.x = 0;
            cls = build_class(r.role());
            if (cls == 7 && r.head()) cls = 3;
        }
        if (threadIdx.x < 8) hist[threadIdx.x] = 0;
        __syncthreads();
        const uint32_t rank = atomicAdd(&hist[cls], 1u);
        __syncthreads();
        if (threadIdx.x == 0) { uint32_t run = 0; for (int c = 0; c < 8; c++) { const uint32_t v = hist[c]; hist[c] = run; run += v; } }
        __syncthreads();
        order[hist[cls] + rank] = (uint16_t)threadIdx.x;
        __syncthreads();
        i = b0 + order[threadIdx.x];
        if (threadIdx.x >= hist[7]) i = 0xFFFFFFFFu;      // no node to write (leaves behind the barriers below)
    }
    {   // the segment of every token of the block: ONE search of the segment table (for the block's first token), then the block's
        // window of segment starts in LDS (a segment has at least one token, so the block touches at most PP_BLOCK segments)
        __shared__ uint32_t seg_first;
        __shared__ uint32_t segw[PP_BLOCK];
        if (threadIdx.x == 0) { uint32_t a = 0, b = n_segs; while (b - a > 1) { const uint32_t m = (a + b) >> 1; if (seg_start[m] <= b0) a = m; else b = m; } seg_first = a; }
        __syncthreads();
        const uint32_t s0 = seg_first;
        segw[threadIdx.x] = s0 + threadIdx.x < n_segs ? seg_start[s0 + threadIdx.x] : 0xFFFFFFFFu;
        __syncthreads();
        uint32_t a = 0, b = PP_BLOCK;
        while (b - a > 1) { const uint32_t m = (a + b) >> 1; if (segw[m] <= i) a = m; else b = m; }
        s = s0 + a;
    }
    if (i >= n_tokens) return;      // (also the tokens that create no node)
#else
    if (i >= n_tokens) return;
    { uint32_t a = 0, b = n_segs; while (b - a > 1) { const uint32_t m = (a + b) >> 1; if (seg_start[m] <= i) a = m; else b = m; } s = a; }
#endif
    if (seg_info[(uint64_t)s * PSI_WORDS + PSI_PATH] != 0) return;          // SegParser's segment
    Walker W; W.tok = tok; W.pr = pr; W.wlo = wlo;
    W.tb = seg_start[s]; W.te = s + 1 < n_segs ? seg_start[s + 1] : n_tokens; W.nb = base[s].nodes;
    const Rec me = W.rec(i);
    const uint32_t role = me.role(), kd = me.kind();
    const uint32_t NONE = OPUSFE_NONE, low = tok[i].low;
    const uint32_t id = W.nb + me.nid() + me.hflag();

    if (me.head()) {     // Statement::Expression / Assignment  parse.rs:1216-1228
        uint32_t root, term;
        W.sequence(i, &root, &term);
        if (W.role(term) == R_S_EQ) { uint32_t r2, t2; W.sequence(term + 1, &r2, &t2); emit_node(nodes, W.nb + me.nid(), S_Assignment, 0, 2, root, W.stmt_next(t2), 0, low, W.hi(t2 - 1)); }
        else emit_node(nodes, W.nb + me.nid(), S_Expression, 0, 1, root, W.stmt_next(term), 0, low, W.hi(term - 1));
    }

    uint32_t kind_ = 0xFF, sub = 0, nchild = 0, fc = NONE, next = NONE, payload = 0, nlow = low, high = 0;     // kind_ 0xFF: no node
    bool chain = false; uint32_t ch_after = 0, ch_ps = 0;      // an element of an atom chain: its sibling is found below
    switch (role) {
        // ---- definitions ----
        case R_D_EXTERN: case R_D_FN: {
            const uint32_t lp = role == R_D_EXTERN ? i + 1 + W.nl(i + 1) : i + W.nl(i), rp = W.link(lp), n = W.cnt(rp), te_ = W.type_end(rp + 1);
            const uint32_t term = role == R_D_FN ? W.link(te_ + 1) + 1 : te_ + 1;
            kind_ = role == R_D_FN ? D_Function : D_Extern; nchild = n + (role == R_D_FN ? 2u : 1u); fc = n ? W.nid(lp + 1) : W.nid(rp + 1); next = W.def_next(term); payload = lp - 1; high = W.hi(te_);
            def_nodes[base[s].defs + me.cnt()] = id;
            break;
        }
        case R_D_VAR: case R_D_CONST: {
            const uint32_t ts = i + 1 + W.nl(i + 1), te_ = W.type_end(ts);
            kind_ = role == R_D_VAR ? D_Variable : D_Constant; fc = W.nid(ts); payload = ts - 1;
            if (W.role(te_ + 1) == R_D_EQ) { uint32_t root, term; W.sequence(te_ + 2, &root, &term); nchild = 2; next = W.def_next(term); high = W.hi(term - 1); }
            else { nchild = 1; next = W.def_next(te_ + 1); high = W.hi(te_); }
            def_nodes[base[s].defs + me.cnt()] = id;
            break;
        }
        case R_D_TYPE: {
            const uint32_t eq = i + 1 + W.nl(i + 1), te_ = W.type_end(eq + 1);
            kind_ = D_Type; nchild = 1; fc = W.nid(eq + 1); next = W.def_next(te_ + 1); payload = eq - 1; high = W.hi(te_);
            def_nodes[base[s].defs + me.cnt()] = id;
            break;
        }
        case R_D_IMPORT:
            kind_ = kd == T_Import ? D_Import : D_Library; next = W.def_next(i + 2); payload = i + 1; high = W.hi(i + 1);
            def_nodes[base[s].defs + me.cnt()] = id;
            break;
        case R_SIG_ARG: {
            const uint32_t ts = i + W.nl(i), te_ = W.type_end(ts);
            const uint32_t r1 = W.role(te_ + 1), r2 = W.role(te_ + 2);
            if (r1 == R_SIG_COMMA && r2 != R_SIG_RP_FN && r2 != R_SIG_RP_EXT) next = W.nid(te_ + 2);
            else next = W.nid((r1 == R_SIG_COMMA ? te_ + 2 : te_ + 1) + 1);                                  // the return type
            kind_ = X_NamedType; nchild = 1; fc = W.nid(ts); payload = ts - 1; high = W.hi(te_);
            break;
        }
        // ---- types ----
        case R_TY_NAME: case R_TY_NULL: case R_TY_AT: case R_TY_LB: case R_TY_LP: case R_TY_RECORD: {
            uint32_t te_ = i;
            kind_ = Y_Reference; nchild = 1;
            if (role == R_TY_NAME) { kind_ = Y_Name; nchild = 0; payload = i; }
            else if (role == R_TY_NULL) { kind_ = Y_Null; nchild = 0; }
            else if (role == R_TY_AT) { const uint32_t m = W.role(i + 1) == R_TY_MUT ? 1u : 0u; kind_ = m ? Y_MutableReference : Y_Reference; fc = W.nid(i + 1 + m); te_ = W.type_end(i + 1 + m); }
            else if (role == R_TY_LB) {
                const uint32_t rb = W.link(i);
                uint32_t sub_;
                if (rb == i + 1) { const uint32_t m = W.role(rb + 1) == R_TY_MUT ? 1u : 0u; kind_ = m ? Y_MutableArrayReference : Y_ArrayReference; sub_ = rb + 1 + m; }
                else { kind_ = Y_Array; payload = i + 1; sub_ = rb + 1; }
                fc = W.nid(sub_); te_ = W.type_end(sub_);
            } else if (role == R_TY_LP) { const uint32_t rp = W.link(i), n = W.cnt(rp); kind_ = Y_Proc; nchild = n + 1; fc = n ? W.nid(i + 1) : W.nid(rp + 1); te_ = W.type_end(rp + 1); }
            else { const uint32_t rb = W.link(i + 1), n = W.cnt(rb); kind_ = Y_Record; nchild = n; fc = n ? W.nid(i + 2) : NONE; te_ = rb; }
            high = W.hi(te_);
            // only the outermost type of a chain has a sibling; an inner one (behind `@`, `mut`, `{..}`, a proc type's `)`) is the only child of its parent
            const uint32_t pr_ = i > W.tb ? W.role(i - 1) : (uint32_t)R_EOF;
            const bool inner = pr_ == R_TY_AT || pr_ == R_TY_MUT || pr_ == R_TY_ARR_RB || (pr_ == R_TY_RP && W.role(W.link(i - 1)) == R_TY_LP);
            if (!inner) next = type_next(W, i, te_);
            break;
        }
        case R_TY_FIELD: {
            const uint32_t ts = i + W.nl(i), te_ = W.type_end(ts);
            kind_ = X_NamedType; nchild = 1; fc = W.nid(ts); next = W.role(te_ + 1) == R_TY_FIELD ? W.nid(te_ + 1) : NONE; payload = ts - 1; high = W.hi(te_);
            break;
        }
        // ---- expressions ----
        case R_E_LEAF: {
            if (kd == T_Identifier && W.role(i + 1) == R_CALL_LP) {      // NamedCall(name, args)
                const uint32_t rp = W.link(i + 1), n = W.cnt(rp);
                kind_ = E_NamedCall; nchild = n; payload = i; high = W.hi(rp); fc = n ? W.sequence_root(i + 2) : NONE;
                chain = true; ch_after = rp + 1; ch_ps = i;
            } else {
                kind_ = kd == T_Integer ? E_Integer : kd == T_Float ? E_Float : kd == T_String ? E_String : kd == T_Identifier ? E_Variable : kd == T_Null ? E_Null : E_Bool;
                sub = kd == T_True ? 1 : 0; payload = (kd == T_Null || kd == T_False || kd == T_True) ? 0u : i; high = low + tok[i].len;
                chain = true; ch_after = i + 1; ch_ps = i;
            }
            break;
        }
        case R_E_LP: { const uint32_t rp = W.link(i); kind_ = E_Parentheses; nchild = 1; high = W.hi(rp); fc = W.sequence_root(i + 1); chain = true; ch_after = rp + 1; ch_ps = i; break; }
        case R_E_VARDEF: {
            const uint32_t ts = i + 1 + W.nl(i + 1), te_ = W.type_end(ts);
            kind_ = E_VariableDefinition; nchild = 1; fc = W.nid(ts); payload = ts - 1; high = W.hi(te_); chain = true; ch_after = te_ + 1; ch_ps = i;
            break;
        }
        case R_E_SIZEOF: { const uint32_t te_ = W.type_end(i + 1); kind_ = kd == T_Sizeof ? E_Sizeof : E_Alignof; nchild = 1; fc = W.nid(i + 1); high = W.hi(te_); chain = true; ch_after = te_ + 1; ch_ps = i; break; }
        case R_E_RECORD: {
            const uint32_t lb = i + 1 + W.nl(i + 1), rb = W.link(lb), n = W.cnt(rb);
            kind_ = E_Record; nchild = n + 1; fc = id + 1; high = W.hi(rb); chain = true; ch_after = rb + 1; ch_ps = i;
            emit_node(nodes, id + 1, Y_Name, 0, 0, NONE, n ? W.nid(lb + 1) : NONE, lb - 1, low, W.hi(lb - 1));      // parse.rs:1049-1050: the name's span starts at `record`
            break;
        }
        case R_E_FIELD: {
            uint32_t root, term;
            W.sequence(i + 2, &root, &term);
            kind_ = X_FieldInit; nchild = 1; payload = i; fc = root; next = W.role(term) == R_E_FIELD ? W.nid(term) : NONE; high = W.hi(term - 1);
            break;
        }
        case R_E_PROC: {
            const uint32_t lp = i + 1 + W.nl(i + 1), rp = W.link(lp), n = W.cnt(rp);
            kind_ = E_Proc; nchild = n; fc = n ? W.nid(lp + 1) : NONE; payload = lp - 1; high = W.hi(rp); chain = true; ch_after = rp + 1; ch_ps = i;
            break;
        }
        case R_E_PREFIX: {
            uint32_t q = i + 1;
            for (uint32_t g = W.bound(); g && W.role(q) == R_E_PREFIX; g--) q++;
            Rec rq = W.rec1(q);
            uint32_t top = W.nb + rq.nid() + rq.hflag(), k = W.next_item(q, rq);
            for (uint32_t g = W.bound(); g; g--) {
                const Rec rk = W.rec1(k);
                if (rk.head() || iclass_of(rk.role()) != IC_POSTFIX) break;
                if (!(rk.role() == R_CALL_LP && W.named_call(k))) top = W.nb + rk.nid();
                k = W.next_item(k, rk);
            }
            kind_ = kd == T_Minus ? E_Negate : kd == T_Ref ? E_Reference : kd == T_Mut ? E_MutableReference : kd == T_Refs ? E_ArrayReference : E_MutableArrayReference;
            nchild = 1; fc = W.role(i + 1) == R_E_PREFIX ? W.enode(i + 1) : top; high = W.hi(k - 1);
            const bool inner = i > W.tb && !me.head() && W.role(i - 1) == R_E_PREFIX;
            if (!inner) next = W.atom_next(i, k);
            break;
        }
        case R_E_NOT: case R_BINOP: {   // Operator(op, left, right); `not`: Operator(Not, Null, e)  parse.rs:1027-1038
            const int px = me.prec();
            // to the left: the first operator of lower precedence bounds the left operand; the operator of lowest precedence passed on the way is its root
            uint32_t l = NONE, best = NONE, hd = i, ladj = NONE; int bestp = 1000;
            Rec rk = me;
            for (uint32_t g = W.bound(); g; g--) {
                const uint32_t a = rk.a();
                if (a & PX_END) { hd = W.tb + (a & (PX_END - 1)); break; }
                const uint32_t k = W.tb + a;
                rk = W.rec(k);
                if (ladj == NONE) ladj = k;
                const int p = rk.prec();
                if (p < px) { l = k; break; }
                if (p < bestp) { best = k; bestp = p; }
            }
            uint32_t r, term;
            const uint32_t rchild = W.right_operand(i, me, px, &r, &term);
            kind_ = E_Operator; sub = op_of_kind((uint8_t)kd); nchild = 2;
            fc = best != NONE ? W.enode(best) : (role == R_E_NOT ? id + 1 : W.enode(W.tb + me.c()));
            nlow = l != NONE ? W.lo(l + 1) : W.lo(hd);
            high = r != NONE ? (W.role(r) == R_E_NOT ? W.lo(r) : W.hi(r - 1)) : W.hi(term - 1);      // before a `not` stands that operator's Null, empty at the keyword (parse.rs:1032)
            if (l == NONE && r == NONE) next = W.context_next(hd, term);
            else if (r == NONE || (l != NONE && W.rec1(l).prec() >= W.rec1(r).prec())) next = NONE;     // right operand of l
            else { uint32_t r2, t2; const Rec rr = W.rec(r); next = W.right_operand(r, rr, rr.prec(), &r2, &t2); }   // left operand of r: the sibling is r's right operand
            if (role == R_E_NOT) {
                const bool right_of_left = ladj != NONE && W.rec1(ladj).prec() >= 2;       // `a == not b`: the Null is the right operand of `==` (Q9)
                emit_node(nodes, id + 1, E_Null, 0, 0, NONE, right_of_left ? NONE : rchild, 0, low, low);
            }
            break;
        }
        case R_POST_AT: case R_POST_DOT: case R_CALL_LP: case R_POST_COLON: {
            if (role == R_CALL_LP && W.named_call(i)) break;
            const uint32_t ps = W.primary_start(i);
            fc = W.item_node(W.prev_item(i)); nlow = W.lo(ps); chain = true; ch_ps = ps; nchild = 1;
            if (role == R_POST_AT) { kind_ = E_Dereference; high = W.hi(i); ch_after = i + 1; }
            else if (role == R_POST_DOT) { const uint32_t last = i + W.nl(i + 1); kind_ = E_Field; payload = last; high = W.hi(last); ch_after = last + 1; }
            else if (role == R_CALL_LP) { const uint32_t rp = W.link(i); kind_ = E_CallOrIndex; nchild = W.cnt(rp) + 1; high = W.hi(rp); ch_after = rp + 1; }
            else { const uint32_t te_ = W.type_end(i + 1); kind_ = E_Cast; nchild = 2; high = W.hi(te_); ch_after = te_ + 1; }
            break;
        }
        // ---- statements ----
        case R_S_RETURN: case R_S_VAR_DEF: {
            const uint32_t e0 = role == R_S_RETURN ? i + 1 : i + 2 + W.nl(i + 1);
            uint32_t root, term;
            W.sequence(e0, &root, &term);
            kind_ = role == R_S_RETURN ? S_Return : S_VariableDefinition; nchild = 1; payload = role == R_S_RETURN ? 0u : e0 - 2; fc = root; next = W.stmt_next(term); high = W.hi(term - 1);
            break;
        }
        case R_S_VAR_T: {   // parse.rs:1140-1159
            const uint32_t ts = i + 1 + W.nl(i + 1), te_ = W.type_end(ts);
            fc = id + 1;
            if (W.role(te_ + 1) == R_S_EQ) {
                uint32_t root, term;
                W.sequence(te_ + 2, &root, &term);
                kind_ = S_Assignment; nchild = 2; next = W.stmt_next(term); high = W.hi(term - 1);
                emit_node(nodes, id + 1, E_VariableDefinition, 0, 1, W.nid(ts), root, ts - 1, low, W.hi(te_));
            } else {
                kind_ = S_Expression; nchild = 1; next = W.stmt_next(te_ + 1); high = W.hi(te_);
                emit_node(nodes, id + 1, E_VariableDefinition, 0, 1, W.nid(ts), NONE, ts - 1, low, high);
            }
            break;
        }
        case R_S_IFWHILE: case R_S_FOR: {
            uint32_t e0 = i + 1;
            if (role == R_S_FOR) {
                const uint32_t p = i + 1 + W.nl(i + 1);
                payload = p - 1;
                if (W.role(p) != R_S_IN) { sub = 1; fc = W.nid(p); e0 = W.type_end(p) + 2; } else e0 = p + 1;
            }
            uint32_t root, body;
            W.sequence(e0, &root, &body);
            const uint32_t d = W.link(body);
            const bool has_else = W.role(d + 1) == R_S_ELSE;
            const uint32_t term = has_else ? W.link(d + 2) + 1 : d + 1;
            kind_ = role == R_S_FOR ? S_For : kd == T_If ? S_If : S_While;
            nchild = 2u + sub + (has_else ? 1u : 0u); if (!sub) fc = root; next = W.stmt_next(term); high = W.hi(term - 1);
            break;
        }
        case R_S_BREAKCONT: kind_ = kd == T_Break ? S_Break : S_Continue; next = W.stmt_next(i + 1); high = low + tok[i].len; break;
        case R_INDENT: {
            const uint32_t d = W.link(i), n = W.cnt(d);
            const bool then_with_else = i > W.tb && W.role(i - 1) != R_S_ELSE && W.role(d + 1) == R_S_ELSE;
            kind_ = X_Block; nchild = n; fc = n ? W.nid(i + 1) : NONE; next = then_with_else ? W.nid(d + 2) : NONE; high = W.hi(d);
            break;
        }
        default: break;
    }
    // the sibling of an atom-chain element that ends before ch_after (ch_ps = first token of the atom's primary)
    if (chain) {
        const Rec ra = W.rec1(ch_after);
        if (!ra.head() && iclass_of(ra.role()) == IC_POSTFIX) {
            if (ra.role() == R_POST_COLON) next = W.nid(ch_after + 1);                                                   // Cast(expression, type)
            else if (ra.role() == R_CALL_LP && W.cnt(W.tb + ra.link())) next = W.sequence_root(ch_after + 1);              // CallOrIndex(callee, args)
        } else if (ch_ps > W.tb && !W.head(ch_ps) && W.role(ch_ps - 1) == R_E_PREFIX) { /* operand of a prefix operator */ }
        else next = W.atom_next(ch_ps, ch_after);
    }
    if (kind_ != 0xFF) emit_node(nodes, id, kind_, sub, nchild, fc, next, payload, nlow, high);
}

}  // namespace pp
}  // namespace opusfe

// GPU parser for gfx950.  Restates parse.rs:723-1355 (recursive descent, LL(1) on the peeked token)
// as an explicit-stack pushdown machine so that nesting depth (1000-deep blocks / parentheses /
// pointer types) costs frames (the top of the stack in LDS, the rest in global memory) instead of call-stack recursion.
//
// Unit of parallelism: one top-level SEGMENT (lex.h k_final: the run of tokens between two returns
// of the indentation to column 0) per WAVE (wave-uniform control flow, machine state in LDS, tokens by scalar loads,
// the newest nodes in an LDS ring).  Segments contain whole definitions, so waves never
// communicate; the first syntax error of the file is the error of the lowest-numbered failing
// segment (frontend.rs:130-133 stops at the first Err).
//
//   k_parse_segments : tokens -> sparse node records (segment s owns [2*tok_begin(s) + 8 s, ...)),
//                      per-segment info (node count, definition count, first error)
//   scans            : node / definition bases
//   k_ast_compact    : sparse -> dense node array (indices remapped), definition table
#pragma once
#include <hip/hip_runtime.h>
#include "unit.h"
#include "scan.h"
#include "parse_par.h"

namespace opusfe {

// per-segment info words
enum { SI_NODE_BASE = 0, SI_NODE_COUNT, SI_DEF_COUNT, SI_FIRST_DEF, SI_ERR_KIND, SI_ERR_TOK, SI_ERR_LOW, SI_ERR_LEN, SI_ERR_EXPECTED,
       SI_END_TOK, SI_MAX_DEPTH, SI_SPARSE_BASE /* SegParser's node region */, SI_FRAME_BASE, SI_PATH /* 0: data-parallel parser (parse_par.h), 1: SegParser */,
       SI_WORDS = 14 };
static_assert(SI_NODE_COUNT == pp::PSI_NODE_COUNT && SI_DEF_COUNT == pp::PSI_DEF_COUNT && SI_ERR_KIND == pp::PSI_ERR_KIND && SI_SPARSE_BASE == pp::PSI_SPARSE_BASE &&
              SI_FRAME_BASE == pp::PSI_FRAME_BASE && SI_PATH == pp::PSI_PATH && SI_WORDS == pp::PSI_WORDS, "parse_par.h shares these words");

#ifndef OPUSFE_PARSE_LDS_FRAMES
#define OPUSFE_PARSE_LDS_FRAMES 48
#endif
enum { PARSE_LDS_FRAMES = OPUSFE_PARSE_LDS_FRAMES };
#ifndef OPUSFE_PARSE_RING
#define OPUSFE_PARSE_RING 128
#endif
enum { PARSE_RING = OPUSFE_PARSE_RING };
struct PFrame { uint8_t st; uint8_t aux; uint16_t cnt; uint32_t a, b, c; };
static_assert(sizeof(PFrame) == 16, "frame is one 16-byte store");

// sparse regions of a segment that SegParser parses (reserved by pp::k_pp_roles / k_pp_all_sequential with the same formulas)
__host__ __device__ __forceinline__ uint64_t seg_node_cap(uint32_t tok_begin, uint32_t tok_end) { return 2ull * (tok_end - tok_begin) + 8ull; }
__host__ __device__ __forceinline__ uint64_t seg_frame_cap(uint32_t tok_begin, uint32_t tok_end) { return 2ull * (tok_end - tok_begin) + 16ull; }

// parse.rs:179-196 binary operator of a token kind, OP_None otherwise
__host__ __device__ __forceinline__ uint8_t binop_of(uint8_t k) {
    switch (k) {
        case T_Plus: return OP_Plus; case T_Minus: return OP_Minus; case T_Asterisk: return OP_Times; case T_Slash: return OP_Divide;
        case T_Percent: return OP_Modulo; case T_EqualsEquals: return OP_Equals; case T_LessThan: return OP_LessThan;
        case T_GreaterThan: return OP_GreaterThan; case T_LessThanEquals: return OP_LessThanEquals;
        case T_GreaterThanEquals: return OP_GreaterThanEquals; case T_Is: return OP_Is; case T_And: return OP_And; case T_Or: return OP_Or;
        default: return OP_None;
    }
}
// parse.rs:205-222
__host__ __device__ __forceinline__ int op_prec(uint8_t op) {
    switch (op) {
        case OP_Or: return 0; case OP_And: return 1; case OP_Not: return 2;
        case OP_Equals: case OP_LessThan: case OP_GreaterThan: case OP_LessThanEquals: case OP_GreaterThanEquals: case OP_Is: return 20;
        case OP_Plus: case OP_Minus: return 30;
        default: return 31;
    }
}

// parser states (resume points).  *_0 are function entries.
enum PState : uint8_t {
    PS_ROOT, PS_ROOT_ITEM,
    PS_DEF_0, PS_DEF_VAR_1, PS_DEF_CONST_1, PS_DEF_FIN, PS_SIG_ARGS, PS_SIG_ARG_ITEM, PS_SIG_RET, PS_DEF_FN_BODY,
    PS_TY_0, PS_TY_WRAP, PS_TY_LIST, PS_TY_LIST_ITEM, PS_TY_REC_LOOP, PS_TY_REC_ITEM,
    PS_AT_0, PS_AT_PAREN, PS_AT_PREFIX, PS_AT_NOT, PS_AT_REC_LOOP, PS_AT_REC_ITEM, PS_AT_POST, PS_AT_POST_LOOP, PS_AT_CALL_LIST,
    PS_AT_CALL_ITEM, PS_AT_CAST,
    PS_EX_0, PS_EX_1, PS_EX_2,
    PS_ST_0, PS_ST_FIN1, PS_ST_VAR_T, PS_ST_IF_1, PS_ST_IF_2, PS_ST_FOR_T, PS_ST_EXPR_1,
    PS_BL_0, PS_BL_SINGLE, PS_BL_LOOP, PS_BL_ITEM
};

#ifdef OPUSFE_PARSE_PROF
// optional cycle accounting per parser state (build with -DOPUSFE_PARSE_PROF): [0..63] cycles, [64..127] steps
__device__ unsigned long long g_parse_prof[128];
#endif
struct SegParser {
    // registers of the machine: static LDS variables, one machine per workgroup (defs.h MACHINE_STATE)
#ifdef OPUSFE_EMU
    MACHINE_STATE GTable<opusfe_token> tok;
#else
    MACHINE_STATE STable<opusfe_token> tok;     // tokens are read-only here: scalar loads
#endif
    MACHINE_STATE opusfe_node* nodes;     // sparse array
    MACHINE_STATE PFrame* stack;          // frames deeper than PARSE_LDS_FRAMES (global memory)
    MACHINE_STATE PFrame lframes[PARSE_LDS_FRAMES];   // the top of the frame stack lives in LDS
    MACHINE_STATE uint32_t nbase, ncap, fcap;     // (the host refuses units whose sparse regions do not fit 32-bit indices)
    MACHINE_STATE uint32_t i;          // next token to consume
    MACHINE_STATE bool peeked;         // token i has been lexed by the reference at this point (peek_token)
    MACHINE_STATE uint32_t position;   // parse.rs `position`: end of the last consumed token
    MACHINE_STATE uint32_t nn;         // nodes created
    MACHINE_STATE uint32_t sp, max_sp;
    MACHINE_STATE PFrame cur;
    MACHINE_STATE uint32_t ret;
    // error
    MACHINE_STATE uint32_t err_kind, err_tok, err_low, err_len, err_expected;
    MACHINE_STATE uint32_t dev_status;
    MACHINE_STATE uint32_t unsup_pos;   // where the reference could no longer be followed (swallowed RESYNC error), else NONE

    MACHINE_STATE uint32_t n_tokens;

    // Every access is range checked: on malformed input (the part of a file after its first syntax error is
    // still parsed by other lanes) a wrong index becomes an internal-error status, never a GPU memory fault.
    // The newest PARSE_RING nodes of the segment live in an LDS ring (write-back: a node goes to the sparse global array
    // when its slot is reused, the rest when the segment is done).  A node is linked, spanned and rotated
    // (correct_precedence) shortly after it is created: those reads and read-modify-writes used to be dependent global
    // round trips on addresses the wave had just stored to.
    MACHINE_STATE opusfe_node ring[PARSE_RING];
#ifdef OPUSFE_PARSE_PROF
    MACHINE_STATE uint32_t pprof[128];     // per state: cycles / 16, steps
#endif
    __device__ __forceinline__ opusfe_node& N(uint32_t id) {
        uint32_t off = id - nbase;
        if (off >= ncap) { dev_status = DS_INTERNAL_CAPACITY; return nodes[nbase + ncap - 1]; }   // last slot is a dummy
        if (off < nn && nn - off <= PARSE_RING) return ring[off % PARSE_RING];
        return nodes[nbase + off];
    }
    __device__ void flush_ring() {     // all lanes: the nodes still in the ring -> global
        const uint32_t first = nn > PARSE_RING ? nn - PARSE_RING : 0u;
        for (uint32_t j = first + threadIdx.x % UNIT_LANES; j < nn; j += UNIT_LANES) gst(&nodes[nbase + j], (opusfe_node)ring[j % PARSE_RING]);
    }
    __device__ __forceinline__ uint32_t tk(uint32_t k) const { return k < n_tokens ? k : n_tokens - 1; }   // clamps to EndOfFile
    __device__ __forceinline__ uint8_t kind_at(uint32_t k) const { return tok[tk(k)].kind; }
    __device__ __forceinline__ uint32_t low_at(uint32_t k) const { return tok[tk(k)].low; }
    __device__ __forceinline__ uint32_t end_at(uint32_t k) const { k = tk(k); return tok[k].low + tok[k].len; }
    // parse.rs token_low / real_position as functions of (i, peeked)
    __device__ __forceinline__ uint32_t token_low() const { return peeked ? low_at(i) : (i > 0 ? low_at(i - 1) : 0u); }
    __device__ __forceinline__ uint32_t real_position() const { return peeked ? end_at(i) : (i > 0 ? end_at(i - 1) : 0u); }
    __device__ __forceinline__ void consume() { position = end_at(i); i++; peeked = false; }

    __device__ __forceinline__ bool fail(uint32_t kind, uint32_t low, uint32_t high, uint32_t found, uint32_t expected = 0) {
        err_kind = kind; err_low = low; err_len = high - low; err_tok = found; err_expected = expected;
        return false;
    }
    // a lexical error token reached through `?`: the lexer's error is the parse error
    __device__ __forceinline__ bool fail_lex(uint32_t k) { err_kind = 0x100u | tok[tk(k)].aux; err_tok = k; err_low = 0; err_len = 0; err_expected = 0; return false; }

    __device__ __forceinline__ uint32_t new_node(uint8_t kind, uint8_t sub, uint32_t payload, uint32_t low) {
        if (nn + 1 >= ncap) { dev_status = DS_INTERNAL_CAPACITY; return (uint32_t)(nbase + ncap - 1); }
        const uint32_t off = nn++;
        if (off >= PARSE_RING) gst(&nodes[nbase + off - PARSE_RING], (opusfe_node)ring[off % PARSE_RING]);   // the slot's previous node leaves the ring
        opusfe_node n; n.kind = kind; n.sub = sub; n.nchild = 0; n.first_child = OPUSFE_NONE; n.next = OPUSFE_NONE; n.payload = payload; n.low = low; n.high = low;
        ring[off % PARSE_RING] = n;
        return nbase + off;
    }
    // append `c` to the child list of cur.a (frame keeps last child in cur.c, count in cur.cnt)
    __device__ __forceinline__ void append(uint32_t c) {
        if (cur.cnt == 0) N(cur.a).first_child = c; else N(cur.c).next = c;
        N(c).next = OPUSFE_NONE;
        cur.c = c; cur.cnt++;
    }
    __device__ __forceinline__ void finish(uint32_t n) { N(n).high = position; N(n).nchild = cur.cnt; }
    __device__ __forceinline__ void wrap_single(uint32_t parent, uint32_t child) {   // parent with exactly one child
        N(parent).first_child = child; N(parent).nchild = 1; N(child).next = OPUSFE_NONE; N(parent).high = position;
    }

    // parse.rs:773-801; returns the token index that carries the name, or fails
    __device__ __forceinline__ bool parse_identifier(uint32_t* name) {
        uint8_t k = kind_at(i);
        if (k == T_Error) return fail_lex(i);
        consume();
        if (k == T_Identifier) { *name = i - 1; return true; }
        if (k == T_Operator) {
            uint8_t k2 = kind_at(i);
            if (k2 == T_Error) return fail_lex(i);
            consume();
            uint8_t b = binop_of(k2);
            if ((b != OP_None && k2 != T_And && k2 != T_Or) || k2 == T_Not || k2 == T_For || k2 == T_Continue) { *name = i - 1; return true; }
            return fail(PE_UnexpectedToken, low_at(i - 1), position, i - 1);
        }
        return fail(PE_ExpectedIdentifier, low_at(i - 1), position, i - 1);
    }
    // parse.rs:743-751
    __device__ __forceinline__ bool expect(uint8_t want) {
        uint8_t k = kind_at(i);
        if (k == T_Error) return fail_lex(i);
        consume();
        if (k != want) return fail(PE_ExpectedFoundToken, low_at(i - 1), position, i - 1, want);
        return true;
    }
    // peek_token()? : kind of the next token, lexical errors propagate
    __device__ __forceinline__ bool peek(uint8_t* k) {
        peeked = true;
        *k = kind_at(i);
        if (*k == T_Error) return fail_lex(i);
        return true;
    }

    // parse.rs:837-869 (recursion on the new left child unrolled into a loop)
    __device__ void correct_precedence(uint32_t e) {
        e = uni(e);
        for (;;) {   // (no node is created here, so the references stay valid; each N() costs a range check and a ring lookup)
            opusfe_node& ne = N(e);
            if (ne.kind != E_Operator) return;
            const uint32_t L = ne.first_child;
            opusfe_node& nL = N(L);
            const uint32_t r = nL.next;
            opusfe_node& nr = N(r);
            if (nr.kind != E_Operator) return;
            const uint8_t oe = ne.sub, orr = nr.sub;
            if (op_prec(oe) < op_prec(orr)) return;
            const uint32_t RL = nr.first_child;
            opusfe_node& nRL = N(RL);
            const uint32_t RR = nRL.next;
            const uint32_t low = nL.low, high = nRL.high;
            ne.sub = orr; nr.sub = oe;
            nr.first_child = L; nL.next = RL; nRL.next = OPUSFE_NONE;
            ne.first_child = r; nr.next = RR; N(RR).next = OPUSFE_NONE;
            nr.low = low; nr.high = high;
            e = r;
        }
    }

    __device__ __forceinline__ bool call(uint8_t entry, uint8_t resume) {
        cur.st = resume;
        if (sp + 1 >= fcap) { dev_status = DS_INTERNAL_CAPACITY; err_kind = 0x200u | DS_INTERNAL_CAPACITY; err_tok = tk(i); return false; }
        if (sp < PARSE_LDS_FRAMES) lframes[sp] = cur; else gst(stack + sp, (PFrame)cur);
        sp++;
        if (sp > max_sp) max_sp = sp;
        cur.st = entry; cur.aux = 0; cur.cnt = 0; cur.a = cur.b = cur.c = 0;
        return true;
    }
    __device__ __forceinline__ void retn(uint32_t value) {
        ret = value;
        if (sp == 0) { dev_status = DS_INTERNAL_SEGMENT; cur.st = PS_ROOT; return; }
        --sp;
        cur = sp < PARSE_LDS_FRAMES ? (PFrame)lframes[sp] : gld(stack + sp);
    }

    // Runs until the segment's tokens are used up or the first error.  Returns false on error.
    __device__ bool run(uint32_t seg_end, uint32_t* first_def, uint32_t* def_count) {
        seg_end = uni(seg_end);
#define CALL(entry, resume) do { if (!call(entry, resume)) return false; goto next; } while (0)
#define TRY(x) do { if (!(x)) return false; } while (0)
        uint32_t last_def = OPUSFE_NONE;
        *first_def = OPUSFE_NONE; *def_count = 0;
        cur.st = PS_ROOT; cur.aux = 0; cur.cnt = 0; cur.a = cur.b = cur.c = 0;
#ifdef OPUSFE_PARSE_PROF
        uint64_t step_t0 = __builtin_readcyclecounter(); uint32_t step_state = 63;
#endif
        for (;;) {
            uint8_t k;
#ifdef OPUSFE_PARSE_PROF
            { const uint64_t t = __builtin_readcyclecounter(); pprof[step_state] += (uint32_t)((t - step_t0) >> 4); pprof[64 + step_state]++;   // LDS, flushed once per segment
              step_t0 = __builtin_readcyclecounter(); step_state = cur.st < 63 ? cur.st : 63; }
#endif
            if (dev_status >= DS_INTERNAL_CAPACITY) { err_kind = 0x200u | dev_status; err_tok = tk(i); return false; }
            switch (cur.st) {
            // ---- frontend.rs:125-135: while !parser.is_at_end_of_file() { parse_definition()? } ----
            case PS_ROOT: {
                if (i >= seg_end) return true;
                peeked = true;
                k = kind_at(i);
                if (k == T_EndOfFile) { i = seg_end; return true; }   // EndOfFile is never consumed
                if (k == T_Error) {   // Q7: is_at_end_of_file drops the Err; lexing resumes after it
                    if (tok[tk(i)].flags & OPUSFE_TF_RESYNC) { dev_status = DS_UNSUPPORTED_RESYNC; if (low_at(i) < unsup_pos) unsup_pos = low_at(i); }
                    consume();
                    if (i >= seg_end) return true;   // the definition that follows is the next segment's
                }
                CALL(PS_DEF_0, PS_ROOT_ITEM);
            }
            case PS_ROOT_ITEM: {
                if (last_def == OPUSFE_NONE) *first_def = ret; else N(last_def).next = ret;
                last_def = ret; (*def_count)++;
                cur.st = PS_ROOT;
                goto next;
            }
            // ---- parse_definition  parse.rs:1262-1337 ----
            case PS_DEF_0: {
                TRY(peek(&k));
                uint32_t low = low_at(i);
                uint32_t name;
                switch (k) {
                    case T_Extern: consume(); cur.a = new_node(D_Extern, 0, 0, low); goto signature;
                    case T_Var:
                        consume(); TRY(parse_identifier(&name));
                        cur.a = new_node(D_Variable, 0, name, low);
                        CALL(PS_TY_0, PS_DEF_VAR_1);
                    case T_Const:
                        consume(); TRY(parse_identifier(&name));
                        cur.a = new_node(D_Constant, 0, name, low);
                        CALL(PS_TY_0, PS_DEF_CONST_1);
                    case T_Type:
                        consume(); TRY(parse_identifier(&name));
                        TRY(expect(T_Equals));
                        cur.a = new_node(D_Type, 0, name, low);
                        CALL(PS_TY_0, PS_DEF_FIN);
                    case T_Import: case T_Library: {
                        consume();
                        uint8_t k2 = kind_at(i);                       // parse_string  parse.rs:723-728
                        if (k2 == T_Error) return fail_lex(i);
                        consume();
                        if (k2 != T_String) return fail(PE_UnexpectedToken, low_at(i - 1), position, i - 1);
                        uint32_t d = new_node(k == T_Import ? D_Import : D_Library, 0, i - 1, low);
                        N(d).high = position;
                        retn(d);
                        goto next;
                    }
                    default: cur.a = new_node(D_Function, 0, 0, low); goto signature;
                }
            signature:   // parse_function_signature  parse.rs:1339-1355
                TRY(parse_identifier(&name));
                N(cur.a).payload = name;
                TRY(expect(T_LeftParenthesis));
                cur.st = PS_SIG_ARGS; cur.aux = 0;   // aux: 0 = first item, 1 = after an item
                goto next;
            }
            case PS_SIG_ARGS: {   // parse_comma_separated  parse.rs:871-897 with item = (identifier, type)
                bool saw_comma = true;
                if (cur.aux) {
                    TRY(peek(&k));
                    saw_comma = k == T_Comma;
                    if (saw_comma) consume();
                }
                TRY(peek(&k));
                if (k == T_RightParenthesis) { consume(); CALL(PS_TY_0, PS_SIG_RET); }
                if (!saw_comma) return fail(PE_UnexpectedToken, low_at(i), end_at(i), i);
                uint32_t name;
                uint32_t low = low_at(i);
                TRY(parse_identifier(&name));
                uint32_t nt = new_node(X_NamedType, 0, name, low);
                append(nt);
                cur.b = nt;
                CALL(PS_TY_0, PS_SIG_ARG_ITEM);
            }
            case PS_SIG_ARG_ITEM: wrap_single(cur.b, ret); cur.st = PS_SIG_ARGS; cur.aux = 1; goto next;
            case PS_SIG_RET: {
                append(ret);
                finish(cur.a);                         // definition span = the signature (parse.rs:1325)
                if (N(cur.a).kind == D_Extern) { retn(cur.a); goto next; }
                CALL(PS_BL_0, PS_DEF_FN_BODY);
            }
            case PS_DEF_FN_BODY: append(ret); N(cur.a).nchild = cur.cnt; retn(cur.a); goto next;
            case PS_DEF_VAR_1:
                append(ret);
                TRY(peek(&k));
                if (k == T_Equals) { consume(); do { if (!call(PS_EX_0, PS_DEF_FIN)) return false; goto l_ex_0; } while (0); }
                finish(cur.a); retn(cur.a); goto next;
            case PS_DEF_CONST_1:
                append(ret);
                TRY(expect(T_Equals));
                do { if (!call(PS_EX_0, PS_DEF_FIN)) return false; goto l_ex_0; } while (0);
            case PS_DEF_FIN: append(ret); finish(cur.a); retn(cur.a); goto next;

            // ---- parse_type  parse.rs:899-969 ----
            case PS_TY_0: {
                k = kind_at(i);
                if (k == T_Error) return fail_lex(i);
                consume();
                uint32_t low = low_at(i - 1);
                switch (k) {
                    case T_Identifier: { uint32_t t = new_node(Y_Name, 0, i - 1, low); N(t).high = position; retn(t); goto next; }
                    case T_Null: { uint32_t t = new_node(Y_Null, 0, 0, low); N(t).high = position; retn(t); goto next; }
                    case T_At: {
                        TRY(peek(&k));
                        if (k == T_Mut) { consume(); cur.a = new_node(Y_MutableReference, 0, 0, low); }
                        else cur.a = new_node(Y_Reference, 0, 0, low);
                        CALL(PS_TY_0, PS_TY_WRAP);
                    }
                    case T_LeftParenthesis: cur.a = new_node(Y_Proc, 0, 0, low); cur.st = PS_TY_LIST; cur.aux = 0; cur.b = 0; goto next;
                    case T_Record:
                        TRY(expect(T_LeftBrace));
                        cur.a = new_node(Y_Record, 0, 0, low);
                        cur.st = PS_TY_REC_LOOP;
                        goto next;
                    case T_LeftBrace: {
                        TRY(peek(&k));
                        if (k == T_RightBrace) {
                            consume();
                            TRY(peek(&k));
                            bool is_mut = k == T_Mut;
                            if (is_mut) consume();
                            cur.a = new_node(is_mut ? Y_MutableArrayReference : Y_ArrayReference, 0, 0, low);
                        } else {
                            consume();                                   // parse_integer  parse.rs:763-771 (token already peeked)
                            if (k != T_Integer) return fail(PE_UnexpectedToken, low_at(i - 1), position, i - 1);
                            uint32_t itok = i - 1;
                            TRY(expect(T_RightBrace));
                            cur.a = new_node(Y_Array, 0, itok, low);
                        }
                        CALL(PS_TY_0, PS_TY_WRAP);
                    }
                    default: return fail(PE_UnexpectedToken, low, position, i - 1);
                }
            }
            case PS_TY_WRAP: append(ret); finish(cur.a); retn(cur.a); goto next;
            // comma separated list of types closed by ')' — shared by Type::Proc (cur.b == 0: a return type follows)
            // and Expression::Proc (cur.b == 1: continue with the postfix loop)
            case PS_TY_LIST: {
                bool saw_comma = true;
                if (cur.aux) {
                    TRY(peek(&k));
                    saw_comma = k == T_Comma;
                    if (saw_comma) consume();
                }
                TRY(peek(&k));
                if (k == T_RightParenthesis) {
                    consume();
                    if (cur.b == 1) { goto l_at_post; }
                    CALL(PS_TY_0, PS_TY_WRAP);
                }
                if (!saw_comma) return fail(PE_UnexpectedToken, low_at(i), end_at(i), i);
                CALL(PS_TY_0, PS_TY_LIST_ITEM);
            }
            case PS_TY_LIST_ITEM: append(ret); cur.st = PS_TY_LIST; cur.aux = 1; goto next;
            case PS_TY_REC_LOOP: {
                TRY(peek(&k));
                if (k == T_RightBrace) { consume(); finish(cur.a); retn(cur.a); goto next; }
                uint32_t name, low = low_at(i);
                TRY(parse_identifier(&name));
                uint32_t nt = new_node(X_NamedType, 0, name, low);
                append(nt);
                cur.b = nt;
                CALL(PS_TY_0, PS_TY_REC_ITEM);
            }
            case PS_TY_REC_ITEM: wrap_single(cur.b, ret); cur.st = PS_TY_REC_LOOP; goto next;

            // ---- parse_atom  parse.rs:971-1124 ----
            case PS_AT_0: l_at_0: {
                k = kind_at(i);
                if (k == T_Error) return fail_lex(i);
                consume();
                uint32_t low = low_at(i - 1);
                uint32_t name;
                switch (k) {
                    case T_Integer: cur.a = new_node(E_Integer, 0, i - 1, low); goto l_at_post;
                    case T_Float: cur.a = new_node(E_Float, 0, i - 1, low); goto l_at_post;
                    case T_String: cur.a = new_node(E_String, 0, i - 1, low); goto l_at_post;
                    case T_Identifier: cur.a = new_node(E_Variable, 0, i - 1, low); goto l_at_post;
                    case T_LeftParenthesis: cur.a = new_node(E_Parentheses, 0, 0, low); do { if (!call(PS_EX_0, PS_AT_PAREN)) return false; goto l_ex_0; } while (0);
                    case T_Minus: cur.a = new_node(E_Negate, 0, 0, low); do { if (!call(PS_AT_0, PS_AT_PREFIX)) return false; goto l_at_0; } while (0);
                    case T_Ref: cur.a = new_node(E_Reference, 0, 0, low); do { if (!call(PS_AT_0, PS_AT_PREFIX)) return false; goto l_at_0; } while (0);
                    case T_Mut: cur.a = new_node(E_MutableReference, 0, 0, low); do { if (!call(PS_AT_0, PS_AT_PREFIX)) return false; goto l_at_0; } while (0);
                    case T_Refs: cur.a = new_node(E_ArrayReference, 0, 0, low); do { if (!call(PS_AT_0, PS_AT_PREFIX)) return false; goto l_at_0; } while (0);
                    case T_Muts: cur.a = new_node(E_MutableArrayReference, 0, 0, low); do { if (!call(PS_AT_0, PS_AT_PREFIX)) return false; goto l_at_0; } while (0);
                    case T_Var:
                        TRY(parse_identifier(&name));
                        cur.a = new_node(E_VariableDefinition, 0, name, low);
                        CALL(PS_TY_0, PS_AT_PREFIX);
                    case T_Null: cur.a = new_node(E_Null, 0, 0, low); goto l_at_post;
                    case T_False: cur.a = new_node(E_Bool, 0, 0, low); goto l_at_post;
                    case T_True: cur.a = new_node(E_Bool, 1, 0, low); goto l_at_post;
                    case T_Not: {
                        cur.a = new_node(E_Operator, OP_Not, 0, low);
                        uint32_t nul = new_node(E_Null, 0, 0, low);      // virtual Null, empty span at `not` (parse.rs:1032)
                        append(nul);
                        do { if (!call(PS_EX_0, PS_AT_NOT)) return false; goto l_ex_0; } while (0);
                    }
                    case T_Sizeof: cur.a = new_node(E_Sizeof, 0, 0, low); CALL(PS_TY_0, PS_AT_PREFIX);
                    case T_Alignof: cur.a = new_node(E_Alignof, 0, 0, low); CALL(PS_TY_0, PS_AT_PREFIX);
                    case T_Record: {
                        TRY(parse_identifier(&name));
                        cur.a = new_node(E_Record, 0, 0, low);
                        uint32_t tn = new_node(Y_Name, 0, name, low);
                        N(tn).high = position;
                        append(tn);
                        TRY(expect(T_LeftBrace));
                        cur.st = PS_AT_REC_LOOP;
                        goto next;
                    }
                    case T_Proc:
                        TRY(parse_identifier(&name));
                        TRY(expect(T_LeftParenthesis));
                        cur.a = new_node(E_Proc, 0, name, low);
                        cur.st = PS_TY_LIST; cur.aux = 0; cur.b = 1;
                        goto next;
                    default: return fail(PE_UnexpectedToken, low, position, i - 1);
                }
            }
            case PS_AT_PAREN: append(ret); TRY(expect(T_RightParenthesis)); goto l_at_post;
            case PS_AT_PREFIX: append(ret); goto l_at_post;
            case PS_AT_NOT:   // parse.rs:1027-1038: returns immediately, no postfix loop
                append(ret);
                N(cur.a).nchild = 2;
                correct_precedence(cur.a);
                N(cur.a).high = position;
                retn(cur.a);
                goto next;
            case PS_AT_REC_LOOP: {
                k = kind_at(i);
                if (k == T_Error) return fail_lex(i);
                consume();
                if (k == T_Identifier) {
                    uint32_t fi = new_node(X_FieldInit, 0, i - 1, low_at(i - 1));
                    append(fi);
                    cur.b = fi;
                    TRY(expect(T_Equals));
                    do { if (!call(PS_EX_0, PS_AT_REC_ITEM)) return false; goto l_ex_0; } while (0);
                }
                if (k == T_RightBrace) { goto l_at_post; }
                return fail(PE_UnexpectedToken, low_at(i - 1), position, i - 1);
            }
            case PS_AT_REC_ITEM: wrap_single(cur.b, ret); cur.st = PS_AT_REC_LOOP; goto next;
            // (states that always continue in one known state jump there directly instead of going through the dispatch)
            case PS_AT_POST: l_at_post: finish(cur.a);
            case PS_AT_POST_LOOP: l_at_post_loop: {   // parse.rs:1087-1121
                TRY(peek(&k));
                uint32_t low = N(cur.a).low;
                if (k == T_At) {
                    consume();
                    uint32_t w = new_node(E_Dereference, 0, 0, low);
                    wrap_single(w, cur.a);
                    cur.a = w;
                    goto l_at_post_loop;
                }
                if (k == T_Dot) {
                    consume();
                    uint32_t name;
                    TRY(parse_identifier(&name));
                    uint32_t w = new_node(E_Field, 0, name, low);
                    wrap_single(w, cur.a);
                    cur.a = w;
                    goto l_at_post_loop;
                }
                if (k == T_LeftParenthesis) {
                    consume();
                    if (N(cur.a).kind == E_Variable) {           // parse.rs:1103-1106: Variable(name)(args) -> NamedCall
                        N(cur.a).kind = E_NamedCall;
                        cur.cnt = 0; cur.c = 0;
                    } else {
                        uint32_t w = new_node(E_CallOrIndex, 0, 0, low);
                        uint32_t callee = cur.a;
                        cur.a = w; cur.cnt = 0; cur.c = 0;
                        append(callee);
                    }
                    cur.st = PS_AT_CALL_LIST; cur.aux = 0;
                    goto next;
                }
                if (k == T_Colon) {
                    consume();
                    uint32_t w = new_node(E_Cast, 0, 0, low);
                    uint32_t inner = cur.a;
                    cur.a = w; cur.cnt = 0; cur.c = 0;
                    append(inner);
                    CALL(PS_TY_0, PS_AT_CAST);
                }
                retn(cur.a);
                goto next;
            }
            case PS_AT_CALL_LIST: {
                bool saw_comma = true;
                if (cur.aux) {
                    TRY(peek(&k));
                    saw_comma = k == T_Comma;
                    if (saw_comma) consume();
                }
                TRY(peek(&k));
                if (k == T_RightParenthesis) { consume(); finish(cur.a); goto l_at_post_loop; }
                if (!saw_comma) return fail(PE_UnexpectedToken, low_at(i), end_at(i), i);
                do { if (!call(PS_EX_0, PS_AT_CALL_ITEM)) return false; goto l_ex_0; } while (0);
            }
            case PS_AT_CALL_ITEM: append(ret); cur.st = PS_AT_CALL_LIST; cur.aux = 1; goto next;
            case PS_AT_CAST: append(ret); finish(cur.a); goto l_at_post_loop;

            // ---- parse_expression  parse.rs:815-835 ----
            case PS_EX_0: l_ex_0: {
                // An atom that is one literal / name token with no postfix operator behind it (the usual case) is built here:
                // it is exactly what PS_AT_0 -> PS_AT_POST -> PS_AT_POST_LOOP would do in a frame of their own.
                k = kind_at(i);
                if (k == T_Integer || k == T_Float || k == T_String || k == T_Identifier || k == T_Null || k == T_False || k == T_True) {
                    const uint8_t k2 = kind_at(i + 1);
                    if (k2 != T_At && k2 != T_Dot && k2 != T_LeftParenthesis && k2 != T_Colon && k2 != T_Error) {
                        consume();
                        const uint32_t low = low_at(i - 1);
                        const uint8_t nk = k == T_Integer ? E_Integer : k == T_Float ? E_Float : k == T_String ? E_String : k == T_Identifier ? E_Variable : k == T_Null ? E_Null : E_Bool;
                        const uint32_t a = new_node(nk, k == T_True ? 1 : 0, (k == T_Null || k == T_False || k == T_True) ? 0u : i - 1, low);
                        N(a).high = position;       // finish(): no children
                        peeked = true;              // the postfix loop has looked at the next token
                        ret = a;
                        cur.st = PS_EX_1;
                        goto next;
                    }
                }
                if (!call(PS_AT_0, PS_EX_1)) return false;
                goto l_at_0;
            }
            case PS_EX_1: {
                TRY(peek(&k));
                uint8_t op = binop_of(k);
                if (op == OP_None) { retn(ret); goto next; }
                consume();
                cur.b = ret; cur.aux = op;
                do { if (!call(PS_EX_0, PS_EX_2)) return false; goto l_ex_0; } while (0);
            }
            case PS_EX_2: {
                uint32_t left = cur.b, right = ret;
                const uint32_t e = new_node(E_Operator, cur.aux, 0, N(left).low);
                opusfe_node& ne = N(e);                // (taken after new_node: creating a node may reuse a ring slot)
                ne.first_child = left; N(left).next = right; N(right).next = OPUSFE_NONE; ne.nchild = 2;
                correct_precedence(e);
                ne.high = position;
                retn(e);
                goto next;
            }

            // ---- parse_statement  parse.rs:1126-1235 ----
            case PS_ST_0: {
                TRY(peek(&k));
                uint32_t low = low_at(i);
                uint32_t name;
                switch (k) {
                    case T_Var: {
                        consume();
                        TRY(parse_identifier(&name));
                        uint8_t k2;
                        TRY(peek(&k2));
                        if (k2 == T_Equals) {
                            consume();
                            cur.a = new_node(S_VariableDefinition, 0, name, low);
                            do { if (!call(PS_EX_0, PS_ST_FIN1)) return false; goto l_ex_0; } while (0);
                        }
                        cur.a = new_node(S_Expression, 0, name, low);   // kind / payload fixed in PS_ST_VAR_T
                        CALL(PS_TY_0, PS_ST_VAR_T);
                    }
                    case T_Return: consume(); cur.a = new_node(S_Return, 0, 0, low); do { if (!call(PS_EX_0, PS_ST_FIN1)) return false; goto l_ex_0; } while (0);
                    case T_If: consume(); cur.a = new_node(S_If, 0, 0, low); do { if (!call(PS_EX_0, PS_ST_IF_1)) return false; goto l_ex_0; } while (0);
                    case T_While: consume(); cur.a = new_node(S_While, 0, 0, low); do { if (!call(PS_EX_0, PS_ST_IF_1)) return false; goto l_ex_0; } while (0);
                    case T_For: {
                        consume();
                        TRY(parse_identifier(&name));
                        cur.a = new_node(S_For, 0, name, low);
                        uint8_t k2;
                        TRY(peek(&k2));
                        if (k2 != T_In) { N(cur.a).sub = 1; CALL(PS_TY_0, PS_ST_FOR_T); }
                        TRY(expect(T_In));
                        do { if (!call(PS_EX_0, PS_ST_IF_1)) return false; goto l_ex_0; } while (0);
                    }
                    case T_Break: consume(); cur.a = new_node(S_Break, 0, 0, low); N(cur.a).high = position; retn(cur.a); goto next;
                    case T_Continue: consume(); cur.a = new_node(S_Continue, 0, 0, low); N(cur.a).high = position; retn(cur.a); goto next;
                    default: cur.a = new_node(S_Expression, 0, 0, low); do { if (!call(PS_EX_0, PS_ST_EXPR_1)) return false; goto l_ex_0; } while (0);
                }
            }
            case PS_ST_FIN1: append(ret); finish(cur.a); retn(cur.a); goto next;
            case PS_ST_VAR_T: {   // `var name T [= e]`  parse.rs:1140-1159
                uint32_t low = N(cur.a).low, name = N(cur.a).payload;
                uint32_t vd = new_node(E_VariableDefinition, 0, name, low);
                wrap_single(vd, ret);                                 // span (low, position) taken before the `=` probe
                N(cur.a).payload = 0;
                append(vd);
                TRY(peek(&k));
                if (k == T_Equals) { consume(); N(cur.a).kind = S_Assignment; do { if (!call(PS_EX_0, PS_ST_FIN1)) return false; goto l_ex_0; } while (0); }
                finish(cur.a); retn(cur.a); goto next;
            }
            case PS_ST_FOR_T: append(ret); TRY(expect(T_In)); do { if (!call(PS_EX_0, PS_ST_IF_1)) return false; goto l_ex_0; } while (0);
            case PS_ST_IF_1: append(ret); CALL(PS_BL_0, PS_ST_IF_2);
            case PS_ST_IF_2: {
                append(ret);
                // `if let Ok(Token::Else) = self.peek_token()` (parse.rs:1172,1199): an Err is discarded (Q6)
                peeked = true;
                k = kind_at(i);
                if (k == T_Error) {
                    if (tok[tk(i)].flags & OPUSFE_TF_RESYNC) { dev_status = DS_UNSUPPORTED_RESYNC; if (low_at(i) < unsup_pos) unsup_pos = low_at(i); }
                    consume();
                } else if (k == T_Else) { consume(); CALL(PS_BL_0, PS_ST_FIN1); }
                finish(cur.a); retn(cur.a); goto next;
            }
            case PS_ST_EXPR_1:
                append(ret);
                TRY(peek(&k));
                if (k == T_Equals) { consume(); N(cur.a).kind = S_Assignment; do { if (!call(PS_EX_0, PS_ST_FIN1)) return false; goto l_ex_0; } while (0); }
                finish(cur.a); retn(cur.a); goto next;

            // ---- parse_block  parse.rs:1238-1260 ----
            case PS_BL_0: {
                TRY(peek(&k));
                cur.a = new_node(X_Block, 0, 0, low_at(i));
                if (k == T_Indent) { consume(); cur.st = PS_BL_LOOP; goto next; }
                CALL(PS_ST_0, PS_BL_SINGLE);
            }
            case PS_BL_SINGLE: append(ret); finish(cur.a); retn(cur.a); goto next;
            case PS_BL_LOOP:
                TRY(peek(&k));
                if (k == T_Dedent) { consume(); finish(cur.a); retn(cur.a); goto next; }
                CALL(PS_ST_0, PS_BL_ITEM);
            case PS_BL_ITEM: append(ret); cur.st = PS_BL_LOOP; goto next;
            default: dev_status = DS_INTERNAL_SEGMENT; err_kind = 0x200u | DS_INTERNAL_SEGMENT; err_tok = tk(i); return false;
            }
        next:;
        }
#undef CALL
#undef TRY
    }
};

__global__ MACHINE_ATTR(OPUSFE_PARSE_WAVES) __launch_bounds__(64) void k_parse_segments(const opusfe_token* __restrict__ tok, uint32_t n_tokens, const uint32_t* __restrict__ seg_start,
                                                        uint32_t n_segs, opusfe_node* __restrict__ nodes, PFrame* __restrict__ frames,
                                                        uint32_t* __restrict__ seg_info, uint32_t* __restrict__ status, uint32_t* __restrict__ first_err_seg,
                                                        const uint32_t* __restrict__ list, uint32_t n_list) {
    const uint32_t slot = blockIdx.x * (blockDim.x / UNIT_LANES) + threadIdx.x / UNIT_LANES;
    const uint32_t lane = threadIdx.x % UNIT_LANES;
    if (slot >= n_list) return;
    const uint32_t s = list[slot];     // the segments the data-parallel parser declined
    if (s >= n_segs) return;
    uint32_t tb = seg_start[s], te = (s + 1 < n_segs) ? seg_start[s + 1] : n_tokens;
    SegParser P;    // no fields of its own: the state is static LDS (see lower.h)
    P.tok.p = tok; P.nodes = nodes; P.n_tokens = n_tokens;
    P.nbase = seg_info[(uint64_t)s * SI_WORDS + SI_SPARSE_BASE]; P.ncap = (uint32_t)seg_node_cap(tb, te);
    P.stack = frames + seg_info[(uint64_t)s * SI_WORDS + SI_FRAME_BASE]; P.fcap = (uint32_t)seg_frame_cap(tb, te);
    P.i = tb; P.peeked = false; P.position = tb > 0 ? P.end_at(tb - 1) : 0u;
    P.nn = 0; P.sp = 0; P.max_sp = 0; P.ret = OPUSFE_NONE;
    P.err_kind = OPUSFE_NONE; P.err_tok = 0; P.err_low = 0; P.err_len = 0; P.err_expected = 0; P.dev_status = DS_OK; P.unsup_pos = OPUSFE_NONE;
    { opusfe_node d{}; d.kind = X_Block; d.first_child = OPUSFE_NONE; d.next = OPUSFE_NONE; nodes[P.nbase + P.ncap - 1] = d; }
#ifdef OPUSFE_PARSE_PROF
    for (uint32_t j = lane; j < 128; j += UNIT_LANES) P.pprof[j] = 0;
#endif
    uint32_t first_def, def_count;
    bool ok = P.run(te, &first_def, &def_count);
#ifdef OPUSFE_PARSE_PROF
    for (uint32_t j = lane; j < 128; j += UNIT_LANES) if (P.pprof[j]) atomicAdd(&g_parse_prof[j], (unsigned long long)P.pprof[j]);
#endif
    if (ok && P.i != te) {   // a definition straddled a segment boundary: the segmentation rule did not hold
        ok = false; P.err_kind = 0x200u | DS_INTERNAL_SEGMENT; P.err_tok = P.tk(P.i);
    }
    P.flush_ring();
    if (lane != 0) return;
    uint32_t* si = seg_info + (uint64_t)s * SI_WORDS;
    si[SI_NODE_BASE] = 0; si[SI_NODE_COUNT] = P.nn; si[SI_DEF_COUNT] = def_count; si[SI_FIRST_DEF] = first_def;
    si[SI_ERR_KIND] = ok ? OPUSFE_NONE : P.err_kind; si[SI_ERR_TOK] = P.err_tok; si[SI_ERR_LOW] = P.err_low; si[SI_ERR_LEN] = P.err_len;
    si[SI_ERR_EXPECTED] = P.err_expected; si[SI_END_TOK] = P.i; si[SI_MAX_DEPTH] = P.max_sp;
    if (!ok) atomicMin(first_err_seg, s);
    // internal conditions travel as the segment's error: they only matter if no syntax error precedes them
    if (P.dev_status != DS_OK && P.dev_status < DS_INTERNAL_CAPACITY) atomicMax(status, P.dev_status);
    if (P.unsup_pos != OPUSFE_NONE) atomicMin(status + 1, P.unsup_pos);
}

// Group mode (a corpus of files packed into one unit): a failing segment makes its FILE a syntax-error file — the lowest failing
// segment of a file holds the file's first error — and contributes no nodes or definitions; the other files go on.
__global__ void k_group_seg_errors(uint32_t* __restrict__ seg_info, uint32_t n_segs, const uint32_t* __restrict__ seg_start, const opusfe_token* __restrict__ tok, uint32_t n_tokens,
                                   const uint32_t* __restrict__ file_start, uint32_t n_files, uint32_t* __restrict__ err_seg, uint32_t* __restrict__ flags) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_segs) return;
    uint32_t* si = seg_info + (uint64_t)s * SI_WORDS;
    if (si[SI_ERR_KIND] == OPUSFE_NONE) return;
    const uint32_t t0 = seg_start[s];
    if (t0 >= n_tokens) return;
    const uint32_t f = pos_file(file_start, n_files, tok[t0].low);
    atomicMin(&err_seg[f], s);
    atomicOr(&flags[f], (si[SI_ERR_KIND] & 0x200u) ? (uint32_t)FF_INTERNAL : (uint32_t)FF_SYNTAX);
    si[SI_NODE_COUNT] = 0; si[SI_DEF_COUNT] = 0;
}
// counts -> (node_count, def_count) pairs for the scan
struct Cnt2 { uint32_t nodes, defs; };
struct OpAddCnt2 { __host__ __device__ Cnt2 operator()(Cnt2 a, Cnt2 b) const { return Cnt2{a.nodes + b.nodes, a.defs + b.defs}; } };

__global__ void k_ast_counts(const uint32_t* __restrict__ seg_info, uint32_t n_segs, Cnt2* __restrict__ out) {
    uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_segs) return;
    out[s] = Cnt2{seg_info[(uint64_t)s * SI_WORDS + SI_NODE_COUNT], seg_info[(uint64_t)s * SI_WORDS + SI_DEF_COUNT]};
}

// One wave per SegParser segment (grid-stride): copies the segment's sparse nodes to the dense array, rebasing indices;
// lane 0 then walks the definition chain.
__global__ __launch_bounds__(256) void k_ast_compact(const opusfe_node* __restrict__ sparse, const uint32_t* __restrict__ list, uint32_t n_list,
                                                      uint32_t* __restrict__ seg_info, const Cnt2* __restrict__ base, opusfe_node* __restrict__ dense,
                                                      uint32_t* __restrict__ def_nodes) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t q = wave; q < n_list; q += nwaves) {
        const uint32_t s = list[q];
        uint32_t* si = seg_info + (uint64_t)s * SI_WORDS;
        const uint32_t count = si[SI_NODE_COUNT];
        const uint64_t sb = si[SI_SPARSE_BASE];
        const uint32_t db = base[s].nodes;
        const uint32_t delta = db - (uint32_t)sb;     // mod 2^32 arithmetic: new = old + delta
        for (uint32_t j = lane; j < count; j += 64) {
            opusfe_node n = sparse[sb + j];
            if (n.first_child != OPUSFE_NONE) n.first_child += delta;
            if (n.next != OPUSFE_NONE) n.next += delta;
            dense[db + j] = n;
        }
        if (lane == 0) {
            si[SI_NODE_BASE] = db;
            uint32_t d = si[SI_FIRST_DEF], o = base[s].defs;
            const uint32_t nd = si[SI_DEF_COUNT];
            for (uint32_t j = 0; j < nd && d != OPUSFE_NONE; j++) { def_nodes[o + j] = d + delta; d = sparse[d].next; }
        }
    }
}

// number of '\n' in src[0, n) -> *out (frontend.rs:213-225 line-number rule); only run on the error path
__global__ void k_count_newlines(const uint8_t* __restrict__ src, uint32_t n, uint32_t* __restrict__ out) {
    uint32_t c = 0;
    for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (uint64_t)gridDim.x * blockDim.x) c += src[p] == '\n';
    for (int d = 32; d > 0; d >>= 1) c += __shfl_down(c, d, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}

// ---- definition headers (multi-GPU IR generation: what another rank must know about this shard's definitions) ----
// Function -> "extern " + signature text; Extern / Type / Variable / Constant -> their own text; each followed by '\n'.
__device__ __forceinline__ uint32_t hdr_len(const opusfe_node& d) {
    if (d.kind == D_Import || d.kind == D_Library) return 0;
    return (d.high - d.low) + 1u + (d.kind == D_Function ? 7u : 0u);
}
__global__ void k_hdr_len(const opusfe_node* __restrict__ nodes, const uint32_t* __restrict__ def_nodes, uint32_t n_defs, uint32_t* __restrict__ len) {
    uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d < n_defs) len[d] = hdr_len(nodes[def_nodes[d]]);
}
__global__ __launch_bounds__(256) void k_hdr_copy(const uint8_t* __restrict__ src, const opusfe_node* __restrict__ nodes, const uint32_t* __restrict__ def_nodes,
                                                   uint32_t n_defs, const uint32_t* __restrict__ off, uint8_t* __restrict__ out) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t d = wave; d < n_defs; d += nwaves) {
        const opusfe_node n = nodes[def_nodes[d]];
        const uint32_t total = hdr_len(n);
        if (total == 0) continue;
        uint8_t* o = out + off[d];
        uint32_t pre = 0;
        if (n.kind == D_Function) { const char E[8] = "extern "; if (lane < 7) o[lane] = (uint8_t)E[lane]; pre = 7; }
        const uint32_t body = n.high - n.low;
        for (uint32_t j = lane; j < body; j += 64) o[pre + j] = src[n.low + j];
        if (lane == 0) o[pre + body] = '\n';
    }
}

#ifndef OPUSFE_EMU
// ---- host driver ------------------------------------------------------------------------------
enum { WP_FIRST_ERR = 16, WP_LINES = 17, WP_TOTALS = 18 /* Cnt2 */, WP_PAR = 24 /* pp::PW_* counters, 8 words */ };

// the automaton's table, uploaded once per context
inline void ensure_pp_table(opusfe_ctx* ctx) {
    if (ctx->d_pp_table) return;
    const pp::Table& t = pp::table();
    if (!t.ok) throw HipError{hipErrorUnknown, "parse_par.h: the grammar table derivation is inconsistent"};
    HIP_TRY(hipMalloc((void**)&ctx->d_pp_table, t.blob.size() * 4));
    HIP_TRY(hipMemcpy(ctx->d_pp_table, t.blob.data(), t.blob.size() * 4, hipMemcpyHostToDevice));
    ctx->pp_states = t.n_states; ctx->pp_start = t.start;
}

inline void run_parser(opusfe_ctx* ctx, opusfe_unit* u) {
    hipStream_t st = ctx->stream;
    TempList tmp(ctx);
    const uint32_t n_segs = (uint32_t)u->n_segs, n_tok = (uint32_t)u->n_tokens;
    u->parsed = true;
    u->n_nodes = u->n_defs = 0; u->n_seg_sequential = 0;
    if (n_segs == 0) { u->status = OPUSFE_OK; return; }
    ensure_pp_table(ctx);
    u->d_seg_info = dev_alloc_t<uint32_t>(ctx, (uint64_t)n_segs * SI_WORDS);
    Cnt2* cnt = tmp.get<Cnt2>(n_segs);
    Cnt2* scan_tmp = tmp.get<Cnt2>((n_segs + SCAN_TILE - 1) / SCAN_TILE + 1);
    uint32_t init[4] = {0xFFFFFFFFu, 0, 0, 0};
    HIP_TRY(hipMemcpyAsync(ctx->d_status + WP_FIRST_ERR, init, sizeof init, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(ctx->d_status + WP_PAR, 0, pp::PW_WORDS * sizeof(uint32_t), st));
    uint32_t* decl_list = tmp.get<uint32_t>(n_segs);
    pp::Rec16* prec = nullptr;       // pp::PRec per token
    if (ctx->parser_mode == 1) {
        LAUNCH(ctx, "pp_all_sequential", pp::k_pp_all_sequential, grid_for(n_segs, 256), 256, u->d_seg_start, n_segs, n_tok, u->d_seg_info, ctx->d_status + WP_PAR, decl_list);
    } else {
        // pass 1: one lane per segment runs the grammar automaton (largest segments first: lanes of a wave then have similar lengths)
        uint32_t* order = tmp.get<uint32_t>(n_segs);
        uint32_t* order_hist = tmp.get<uint32_t>(ORDER_CLASSES);
        HIP_TRY(hipMemsetAsync(order_hist, 0, ORDER_CLASSES * sizeof(uint32_t), st));
        LAUNCH(ctx, "order_hist", k_order_hist, grid_for(n_segs, 256), 256, u->d_seg_start, n_segs, n_tok, order_hist);
        LAUNCH(ctx, "order_scan", k_order_scan, 1, 64, order_hist);
        LAUNCH(ctx, "order_scatter", k_order_scatter, grid_for(n_segs, 256), 256, u->d_seg_start, n_segs, n_tok, order_hist, order);
        prec = tmp.get<pp::Rec16>((uint64_t)n_tok + 1);
        pp::RolesArgs A{};
        uint8_t* kinds = tmp.get<uint8_t>((uint64_t)n_tok + 16);
        LAUNCH(ctx, "tok_kinds", pp::k_tok_kinds, grid_for(n_tok, 256), 256, (const opusfe_token*)u->d_tokens, n_tok, kinds);
        A.tok = u->d_tokens; A.n_tokens = n_tok; A.kinds = kinds; A.seg_start = u->d_seg_start; A.n_segs = n_segs; A.order = order;
        A.table = ctx->d_pp_table; A.n_states = ctx->pp_states; A.start = ctx->pp_start;
        A.prec = prec; A.stack = tmp.get<pp::StackEnt>((uint64_t)n_tok / 2 + n_segs + 2);
        A.seg_info = u->d_seg_info; A.work = ctx->d_status + WP_PAR; A.decl_list = decl_list;
        LAUNCH(ctx, "pp_roles", pp::k_pp_roles, grid_for(n_segs, 256), 256, A);
    }
    auto count_and_scan = [&] {
        LAUNCH(ctx, "ast_counts", k_ast_counts, grid_for(n_segs, 256), 256, u->d_seg_info, n_segs, cnt);
        prof_begin(ctx, "ast_scan");
        device_scan<true>(cnt, cnt, n_segs, OpAddCnt2(), Cnt2{0, 0}, scan_tmp, (Cnt2*)(ctx->d_status + WP_TOTALS), st);
        prof_end(ctx);
        HIP_TRY(hipMemcpyAsync(ctx->h_status, ctx->d_status, 64 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        prof_collect(ctx);
    };
    count_and_scan();
    // pass 2 (only when something was declined: syntax errors, the few constructs parse_par.h leaves out): SegParser on those segments
    const uint32_t n_decl = ctx->h_status[WP_PAR + pp::PW_DECLINED];
    opusfe_node* sparse = nullptr;
    u->n_seg_sequential = n_decl;
    if (n_decl) {
        uint64_t sparse_cap, frame_cap;
        memcpy(&sparse_cap, ctx->h_status + WP_PAR + pp::PW_SPARSE, 8); memcpy(&frame_cap, ctx->h_status + WP_PAR + pp::PW_FRAMES, 8);
        if (sparse_cap >= 0xFFFFFFF0ull || frame_cap >= 0xFFFFFFF0ull) { u->status = OPUSFE_ERR_ARG; ctx->msg = "unit has too many tokens for 32-bit node indices: shard it"; return; }
        sparse = tmp.get<opusfe_node>(sparse_cap + 8);
        PFrame* frames = tmp.get<PFrame>(frame_cap + 16);
        LAUNCH(ctx, "parse_segments", k_parse_segments, grid_for((uint64_t)n_decl * UNIT_LANES, 64), 64, u->d_tokens, n_tok, u->d_seg_start, n_segs, sparse, frames,
               u->d_seg_info, ctx->d_status, ctx->d_status + WP_FIRST_ERR, (const uint32_t*)decl_list, n_decl);
#ifdef OPUSFE_PARSE_PROF
        {
            unsigned long long h[128];
            HIP_TRY(hipStreamSynchronize(st));
            HIP_TRY(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_parse_prof), sizeof h));
            unsigned long long z[128] = {0};
            HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_parse_prof), z, sizeof z));
            double tot = 0; for (int i = 0; i < 64; i++) tot += (double)h[i];
            fprintf(stderr, "parser cycle accounting per state (index into enum PState): share, steps, cycles per step\n");
            for (int i = 0; i < 64; i++) if (h[64 + i]) fprintf(stderr, "  state %2d  %5.1f%%  %12llu steps  %8.0f cycles/step\n", i, 100.0 * h[i] / tot, h[64 + i], 16.0 * (double)h[i] / (double)h[64 + i]);
        }
#endif
        if (u->group)
            LAUNCH(ctx, "group_seg_errors", k_group_seg_errors, grid_for(n_segs, 256), 256, u->d_seg_info, n_segs, (const uint32_t*)u->d_seg_start, (const opusfe_token*)u->d_tokens, n_tok,
                   (const uint32_t*)u->group->d_start, u->group->n_files, u->group->d_err_seg, u->group->d_flags);
        count_and_scan();
    }
    uint32_t dev_status = ctx->h_status[0];
    uint32_t first_err = ctx->h_status[WP_FIRST_ERR];
    if (u->group) {
        // syntax errors are per file (k_group_seg_errors); a dropped resync token concerns a file that is flagged for the single-file
        // path anyway (lex.h FF_RESYNC); anything else the packed unit cannot attribute to a file fails the whole group
        first_err = 0xFFFFFFFFu;
        if (dev_status == DS_UNSUPPORTED_RESYNC) dev_status = DS_OK;
    }
    if (first_err != 0xFFFFFFFFu || dev_status != DS_OK) u->n_nodes = u->n_defs = 0;   // no AST: the counts must not promise arrays that do not exist
    else { u->n_nodes = ctx->h_status[WP_TOTALS]; u->n_defs = ctx->h_status[WP_TOTALS + 1]; }
    if (first_err != 0xFFFFFFFFu) {
        // first syntax error of the file = error of the lowest failing segment
        uint32_t si[SI_WORDS];
        HIP_TRY(hipMemcpy(si, u->d_seg_info + (uint64_t)first_err * SI_WORDS, sizeof si, hipMemcpyDeviceToHost));
        opusfe_error e{};
        opusfe_token t{};
        HIP_TRY(hipMemcpy(&t, u->d_tokens + si[SI_ERR_TOK], sizeof t, hipMemcpyDeviceToHost));
        e.tok_index = si[SI_ERR_TOK];
        if (si[SI_ERR_KIND] & 0x200u) {
            u->status = OPUSFE_ERR_INTERNAL; u->dev_status = si[SI_ERR_KIND] & 0xFFu;
            ctx->msg = "parser: internal capacity / segment consistency check failed in segment " + std::to_string(first_err) + " near byte " + std::to_string(t.low);
            return;
        }
        if (si[SI_ERR_KIND] & 0x100u) {
            uint64_t v[2];
            HIP_TRY(hipMemcpy(v, u->d_val64 + t.payload, sizeof v, hipMemcpyDeviceToHost));
            e.kind = si[SI_ERR_KIND] & 0xFFu; e.low = (uint32_t)v[0]; e.len = (uint32_t)(v[0] >> 32); e.payload = v[1]; e.tok_kind = T_Error;
        } else {
            e.kind = si[SI_ERR_KIND]; e.low = si[SI_ERR_LOW]; e.len = si[SI_ERR_LEN]; e.expected = si[SI_ERR_EXPECTED]; e.tok_kind = t.kind;
        }
        // the reference lexes one token ahead of the parser: it meets an unsupported byte before this
        // error only if that byte lies before the end of the token the error was raised on
        const uint32_t unsup_pos = ctx->h_status[1];
        if (unsup_pos < t.low + t.len || (t.kind == T_Error && unsup_pos <= t.low)) { u->status = OPUSFE_UNSUPPORTED; u->dev_status = dev_status; return; }
        if (e.low > 0) {
            LAUNCH(ctx, "parse_error_line", k_count_newlines, 1024, 256, u->d_src, (uint32_t)(e.low < u->len ? e.low : u->len), ctx->d_status + WP_LINES);
            HIP_TRY(hipMemcpyAsync(ctx->h_status + WP_LINES, ctx->d_status + WP_LINES, 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            prof_collect(ctx);
        }
        e.line = 1 + (e.low > 0 ? ctx->h_status[WP_LINES] : 0);
        u->error = e; u->has_error = true;
        u->err_seg = first_err;
        u->status = OPUSFE_SYNTAX;
        u->dev_status = DS_OK;
        return;
    }
    if (dev_status != DS_OK) { u->status = OPUSFE_UNSUPPORTED; u->dev_status = dev_status; return; }
    u->dev_status = DS_OK;
    u->d_nodes = dev_alloc_t<opusfe_node>(ctx, u->n_nodes);
    u->d_def_nodes = dev_alloc_t<uint32_t>(ctx, u->n_defs);
    // pass 3: every node is written once, at its final place
    if (n_decl < n_segs)
        LAUNCH(ctx, "pp_build", pp::k_pp_build, grid_for(n_tok, 256), 256, u->d_tokens, n_tok, u->d_seg_start, n_segs, (const pp::Rec16*)prec, (const uint32_t*)u->d_seg_info,
               (const pp::NodeBase*)cnt, u->d_nodes, u->d_def_nodes);
    if (n_decl) {
        unsigned grid = (unsigned)((n_decl + 3) / 4);
        if (grid > 8192) grid = 8192;
        LAUNCH(ctx, "ast_compact", k_ast_compact, grid, 256, sparse, (const uint32_t*)decl_list, n_decl, u->d_seg_info, cnt, u->d_nodes, u->d_def_nodes);
    }
    HIP_TRY(hipStreamSynchronize(st));
    prof_collect(ctx);
    u->status = OPUSFE_OK;
}

#endif  // OPUSFE_EMU

// OPUSFE_A_PARENTS: every node writes its index into the slot of each of its children (the chain is bounded by the node count, like
// every walk over links that came from memory)
__global__ void k_ast_parents(const opusfe_node* __restrict__ nodes, uint32_t n_nodes, uint32_t* __restrict__ parents) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_nodes) return;
    uint32_t c = nodes[p].first_child, guard = 0;
    while (c != OPUSFE_NONE && c < n_nodes && guard++ < n_nodes) { parents[c] = p; c = nodes[c].next; }
}

}  // namespace opusfe

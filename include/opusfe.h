/* opusfe — MI355X-native front end (lex -> parse -> IR) for the Opus language of ftxqxd/opus.
 *
 * C ABI of the drop-in boundary.  Everything here is plain C: pointers and sizes,
 * no C++/torch types.  The library (libopusfe.so) is built by hipcc for gfx950; every compute entry
 * point runs hand-written HIP kernels and FAILS (OPUSFE_ERR_HIP) when no GPU is present — there is
 * no CPU fallback.
 *
 * Reference interfaces replaced (file:line relative to the reference repository):
 *   opusfe_parse*      <- frontend.rs:120-138  get_definitions(compiler, source)
 *                         = Parser::from_source (parse.rs:396) + is_at_end_of_file (parse.rs:411)
 *                           + parse_definition (parse.rs:1262) + the four span tables
 *                           (compile.rs:20-23)
 *   opusfe_lower       <- frontend.rs:140-187  compile_source: finalize_definition for every
 *                         definition (compile.rs:1002) then IrGenerator::from_function
 *                         (generate_ir.rs:100) per Definition::Function
 *   opusfe_dump(IR)    <- impl Display for IrGenerator (generate_ir.rs:1201-1283), the text that
 *                         `opus -Z` prints (frontend.rs:167-169)
 *   opusfe_error       <- parse::Error (parse.rs:352-360) as data instead of
 *                         report_error + process::exit(1) (frontend.rs:130-133)
 */
#ifndef OPUSFE_H
#define OPUSFE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct opusfe_ctx opusfe_ctx;
typedef struct opusfe_unit opusfe_unit;

/* return codes */
enum {
    OPUSFE_OK = 0,
    OPUSFE_SYNTAX = 1,        /* first syntax error available through opusfe_get_error */
    OPUSFE_UNSUPPORTED = 2,   /* input on which the reference hangs/panics or needs unpinned Unicode tables */
    OPUSFE_ERR_HIP = 3,       /* HIP failure, including "no GPU" */
    OPUSFE_ERR_OOM = 4,
    OPUSFE_ERR_ARG = 5,
    OPUSFE_ERR_INTERNAL = 6,  /* internal capacity / consistency check tripped (never silently wrong) */
    OPUSFE_SEMANTIC = 7       /* opusfe_lower: finalize reported semantic errors, no IR (frontend.rs:145) */
};

/* ---- flat result records (device layout == host layout) ---- */

/* parse.rs:5-67 Token, same variant order; 60 = lexical error token */
typedef struct opusfe_token {
    uint8_t kind;
    uint8_t flags;      /* OPUSFE_TF_* */
    uint16_t aux;       /* Integer: bit15 has-suffix, bit8 signed, bits0-7 size; Float: size (0,32,64); Error: parse::Error kind */
    uint32_t low;       /* byte offset of the token start (parse.rs token_low) */
    uint32_t len;       /* bytes up to where lexing resumes (real_position - token_low) */
    uint32_t payload;   /* Integer: value if OPUSFE_TF_INLINE else index into val64 pool; Float: val64 index (f64 bits);
                           String: byte offset into the string pool ([u32 n][n bytes] padded to 4);
                           Identifier: FNV-1a hash of the name; Error: val64 index of {span, payload} */
} opusfe_token;
#define OPUSFE_TF_INLINE 1u   /* Integer value stored inline */
#define OPUSFE_TF_RESYNC 2u   /* Error token behind which the reference resumes lexing inside what the context automaton sees as a
                                 string body; opusfe_parse re-lexes with that position when the parser drops the token
                                 (DESIGN.md §5 resync passes); opusfe_lex alone leaves the flag for the caller */
#define OPUSFE_TOK_ERROR 60

typedef struct opusfe_node {
    uint8_t kind;          /* OPUSFE_N_* */
    uint8_t sub;           /* Operator: parse::Operator; Bool: value; For: has-type */
    uint16_t nchild;
    uint32_t first_child;  /* node index or 0xFFFFFFFF */
    uint32_t next;         /* next sibling or 0xFFFFFFFF */
    uint32_t payload;      /* token index (names, literals) */
    uint32_t low, high;    /* byte span [low, high) as recorded by the reference's span tables */
} opusfe_node;

typedef struct opusfe_instr {
    uint16_t opcode;       /* generate_ir.rs:8-56 Instruction, same variant order */
    uint16_t sub;          /* Cast: CastType; Bool: value; Call: number of arguments */
    uint32_t dst, a, b;    /* variable ids in declaration order; Jump/Branch/Phi: bb indices where the reference has them */
    uint32_t c, extra;     /* Integer/Float: 64-bit immediate in (c, extra); Call: c = offset into the call-arg pool;
                              String/Field: token index; Sizeof/Alignof: type id; Function: function id; globals: global id */
} opusfe_instr;

typedef struct opusfe_variable { uint32_t type_id; uint32_t is_temporary; } opusfe_variable;

typedef struct opusfe_error {
    uint32_t kind;         /* parse.rs:352-360 variant index */
    uint32_t tok_kind;     /* found token kind (UnexpectedToken / ExpectedFoundToken / ExpectedIdentifier) */
    uint64_t low;          /* span start */
    uint32_t len;          /* span length */
    uint32_t expected;     /* expected token kind (ExpectedFoundToken) */
    uint64_t payload;      /* UnexpectedCharacter: char; InvalidNumericSize: size */
    uint64_t line;         /* 1 + number of '\n' before low (frontend.rs:213-225) */
    uint32_t tok_index;    /* index of the found token in the token stream */
    uint32_t _pad;
} opusfe_error;

typedef struct opusfe_counts {
    uint64_t source_bytes, tokens, nodes, definitions, functions, instrs, vars, bbs, types;
} opusfe_counts;

/* ---- context ---- */
int opusfe_create(opusfe_ctx** out, int device);          /* one ctx per (thread, device); calls on one ctx are serialised */
void opusfe_destroy(opusfe_ctx* ctx);
const char* opusfe_last_message(opusfe_ctx* ctx);          /* human-readable detail of the last non-OK return */
void* opusfe_stream(opusfe_ctx* ctx);                      /* the hipStream_t all kernels of this ctx are launched on */

/* ---- seam A: source -> tokens + AST (frontend.rs:120 get_definitions) ---- */
/* PRECONDITION of every entry point that takes source bytes: `src` is valid UTF-8, as the reference's &str is
   (fs::read_to_string, frontend.rs:85, rejects anything else before the parser runs); the library does not validate it.
   host buffer: copied to HBM first (PCIe-inclusive path) */
int opusfe_parse(opusfe_ctx* ctx, const uint8_t* src, uint64_t len, opusfe_unit** out);
/* source already resident in HBM on ctx's device (the timed path of bench.py) */
int opusfe_parse_device(opusfe_ctx* ctx, const uint8_t* d_src, uint64_t len, opusfe_unit** out);
/* only the lexer (token stream); `force_dents` != 0 lexes as if ignore_dents were always 0 (KAT-L mode) */
int opusfe_lex_device(opusfe_ctx* ctx, const uint8_t* d_src, uint64_t len, int force_dents, opusfe_unit** out);
int opusfe_lex(opusfe_ctx* ctx, const uint8_t* src, uint64_t len, int force_dents, opusfe_unit** out);

/* corpus of independent files (one reference run per file, frontend.rs:23-118 without imports): parses (and lowers when
   `lower` != 0) every file; units[i] / statuses[i] receive the unit and its return code; the call itself only fails on
   device / argument errors.  Several files are in flight at a time on worker contexts owned by `ctx` (host threads,
   OPUSFE_BATCH_WORKERS, default 8): release every unit with opusfe_free_unit before opusfe_destroy(ctx). */
int opusfe_parse_batch(opusfe_ctx* ctx, uint32_t n, const uint8_t* const* srcs, const uint64_t* lens, int lower,
                       opusfe_unit** units, int* statuses);

/* The same corpus as ONE packed pass (BASELINE config 5, thousands of small files).  Replaces, per file, the
   same reference run as opusfe_parse_batch (frontend.rs:23-118 without imports: read_to_string + Compiler::parse_source +
   generate_ir per definition) — every file stays its own universe of names — but the files travel through the kernels together:
   they are packed into one buffer (file f at a 64-byte aligned start, '\n' bytes in between) and every stage is one launch for the
   whole group.  A file the packed pass cannot answer for exactly (empty, first line indented, unsupported input, a swallowed
   lexical error that needs a resync pass, finalize-time semantic errors) is run alone
   through the single-source pipeline by the same call; it then has a unit of its own (opusfe_group_file_unit).
   The call fails only on device / argument errors; per-file outcomes are read with opusfe_group_file_status:
   OPUSFE_OK, OPUSFE_SYNTAX (opusfe_group_file_error), OPUSFE_SEMANTIC, OPUSFE_UNSUPPORTED — the codes the file gets alone.
   In the packed unit the payload of a token of file f is shifted by f * 0x9E3779B1 where it names something (an Identifier's hash;
   keyword and punctuation tokens, whose payload is otherwise 0): that is how one name table scopes every name to its file. */
typedef struct opusfe_group opusfe_group;
int opusfe_parse_group(opusfe_ctx* ctx, uint32_t n_files, const uint8_t* const* srcs, const uint64_t* lens, int lower, opusfe_group** out);
/* already packed and resident in HBM: starts[0] == 0, every start a multiple of 64, ascending, file f = [starts[f], starts[f] + lens[f]),
   at least one byte between two files, every byte between files == '\n'; the buffer must stay valid until opusfe_free_group */
int opusfe_parse_group_device(opusfe_ctx* ctx, const uint8_t* d_packed, uint64_t packed_len, uint32_t n_files, const uint64_t* starts,
                              const uint64_t* lens, int lower, opusfe_group** out);
int opusfe_group_file_status(const opusfe_group* group, uint32_t file);
/* first syntax error of a file whose status is OPUSFE_SYNTAX; positions relative to the file */
int opusfe_group_file_error(const opusfe_group* group, uint32_t file, opusfe_error* out);
/* the packed unit: every array of opusfe_array, positions / token / node indices relative to the packed buffer; owned by the group */
opusfe_unit* opusfe_group_unit(opusfe_group* group);
/* the unit of a file that took the single-file path, else NULL; owned by the group */
opusfe_unit* opusfe_group_file_unit(opusfe_group* group, uint32_t file);
/* where a file's results lie in the packed unit: ranges[0..1] tokens, [2..3] definitions, [4..5] functions (OPUSFE_A_SIGNATURES and
   OPUSFE_A_FUNCTIONS), [6..7] globals, [8] byte offset of the file, [9] why the file left the packed pass (bit 0 empty, 1 first line
   indented, 4 unsupported input, 5 swallowed lexical error, 6 internal capacity, 8 finalize-time error;
   bit 9: syntax error, a result of the packed pass); the index ranges are zero-length for a file with a unit of its own;
   definitions / functions / globals only for a file whose status is OPUSFE_OK in a lowered group */
int opusfe_group_file_ranges(const opusfe_group* group, uint32_t file, uint64_t ranges[10]);
/* how many files were answered by the packed pass / took the single-file path / have a syntax error */
int opusfe_group_stats(const opusfe_group* group, uint64_t* n_packed, uint64_t* n_single, uint64_t* n_syntax);
/* opusfe_dump of ONE file of the group: byte-identical to the dump of the file compiled alone */
int opusfe_group_dump(opusfe_group* group, uint32_t file, int what, char** out, uint64_t* out_len);
void opusfe_free_group(opusfe_group* group);

/* ---- one huge source sharded over several GPUs (no reference counterpart: the reference is single-threaded; this is the
   byte-range analogue of its sequential lexer state, parse.rs:364-391, carried across shards) ----
   A shard starts right after a '\n'.  The host chains three tiny exchanges (all_gather over RCCL): context function ->
   start state, token-kind tails -> bracket depth, then picks as its cut the first top-level segment start. */
typedef struct opusfe_shard opusfe_shard;
int opusfe_shard_ctx(opusfe_ctx* ctx, const uint8_t* d_src, uint64_t len, uint32_t* fn);
/* opusfe_shard_ctx for callers that cannot allocate (code under HIP graph capture, tight serving loops, harnesses that
   forbid hipMalloc).  NO COUNTERPART IN THE REFERENCE.  It allocates nothing: the tile aggregates live in a static device
   array of 256 KiB that every GPU the library runs on holds until the process ends, so `len` is limited to 1 GiB
   (OPUSFE_ERR_ARG beyond that: split the range or use opusfe_shard_ctx); the composed context function is written to the
   caller's device word *d_fn by work queued on ctx's stream (opusfe_stream), with no host synchronisation.  All calls on one
   GPU share that GPU's array: calls that overlap in time on one GPU (two contexts / streams) race — serialise them. */
int opusfe_shard_ctx_static(opusfe_ctx* ctx, const uint8_t* d_src, uint64_t len, uint32_t* d_fn);
int opusfe_shard_lex(opusfe_ctx* ctx, const uint8_t* d_src, uint64_t len, uint32_t start_state, opusfe_shard** out,
                     uint64_t* n_tokens, uint64_t* tail_kinds);
int opusfe_shard_net_depth(opusfe_shard* shard, uint64_t prev_kinds, int64_t* net);
int opusfe_shard_boundary(opusfe_shard* shard, int64_t depth_in, uint64_t* pos);
/* the whole fixed-size carry of a shard in one call (context function, token count, tail / head token kinds, net bracket depth,
   first top-level segment start, length, look-back flag: uint64_t carry[8]), computed speculatively for the common entry
   (context N, depth 0): what ONE all-gather of 64 bytes per rank exchanges.  Every rank can verify every rank's
   speculation from the gathered carries alone; only when it fails is the dependent chain above needed.  `*out` as for
   opusfe_shard_lex.  No counterpart in the reference. */
int opusfe_shard_carry(opusfe_ctx* ctx, const uint8_t* d_src, uint64_t len, uint64_t* carry, opusfe_shard** out);
void opusfe_shard_free(opusfe_shard* shard);
/* own (+) hdr: a new parsed unit whose definition order is hdr's definitions that start before byte `split_pos` of hdr's
   source, then own's, then the rest of hdr's; arrays concatenated and re-based on the GPU (no second parse of `own`).
   `hdr` is the parsed text of the other shards' headers (those of the ranks before this one first).  No counterpart in
   the reference (one file, one thread); for lowering a shard against the whole program's declarations. */
int opusfe_unit_join(opusfe_ctx* ctx, opusfe_unit* own, opusfe_unit* hdr, uint64_t split_pos, opusfe_unit** out);
/* A parsed unit from its six flat arrays (device pointers on ctx's device; the library copies them): the inverse of
   opusfe_array for OPUSFE_A_SOURCE, _TOKENS, _VAL64, _STRINGS, _NODES and _DEFINITIONS.  For moving parsed units between
   GPUs: the shards exchange their parsed definition headers and join them (opusfe_unit_join) instead of every rank parsing
   every other rank's header text.  No counterpart in the reference (one file, one thread).  The arrays must come from
   opusfe_array of a unit that parsed with OPUSFE_OK; indices are range-checked by the consumers as always. */
int opusfe_unit_from_arrays(opusfe_ctx* ctx, const uint8_t* d_src, uint64_t src_len, const opusfe_token* d_tokens, uint64_t n_tokens,
                            const uint64_t* d_val64, uint64_t n_val64, const uint8_t* d_strings, uint64_t n_strings,
                            const opusfe_node* d_nodes, uint64_t n_nodes, const uint32_t* d_defs, uint64_t n_defs, opusfe_unit** out);
/* device text (owned by the unit) that other shards prepend / append to see this unit's definitions: functions as
   `extern <signature>`, externs / types / globals verbatim, one per line, in file order */
int opusfe_unit_headers(opusfe_unit* unit, const uint8_t** d_out, uint64_t* len);

/* ---- seam B: AST -> IR (frontend.rs:140 compile_source, IR part) ---- */
int opusfe_lower(opusfe_ctx* ctx, opusfe_unit* unit);

/* number of semantic errors (compile.rs:124-152) reported by finalize / IR generation of the unit; IR generated while
   errors were reported contains Error instructions exactly like the reference's (compile.rs:777-780).  The diagnostics
   themselves, in the reference's report order, are the array OPUSFE_A_SEM_ERRORS (also after OPUSFE_SEMANTIC, where they
   are the finalize-time errors of compile.rs:1002-1065 and no IR exists) and opusfe_dump(unit, 4). */
int opusfe_semantic_error_count(const opusfe_unit* unit, uint64_t* n);

/* how the functions of a lowered unit were lowered: by the node-parallel kernels (lower_par.h) / by the sequential machine (the
   functions the parallel path declines: semantic errors, shadowing, structural comparisons of record / function types, 2048 nodes
   and more).  No counterpart in the reference; both produce the reference's IR. */
int opusfe_lowering_stats(const opusfe_unit* unit, uint64_t* n_parallel, uint64_t* n_sequential);
/* how the top-level segments of a parsed unit were parsed: by the data-parallel parser (parse_par.h: grammar automaton + one thread per
   token) / by the sequential machine SegParser (the segments the former declines: every segment with a syntax or lexical error, blocks
   that are not indented, a prefix operator directly followed by `not`, segments beyond 65535 tokens).  Both produce the reference's AST. */
int opusfe_parsing_stats(const opusfe_unit* unit, uint64_t* n_parallel, uint64_t* n_sequential);
/* mode 0 (default): data-parallel parser with SegParser as fallback; mode 1: SegParser alone (diagnostic: A/B tests of the two parsers).
   No counterpart in the reference. */
int opusfe_set_parser_mode(opusfe_ctx* ctx, int mode);
/* device memory this context holds (result buffers of live units + the caching allocator's free blocks): the high-water mark
   of the HBM footprint of the calls made so far */
int opusfe_memory_held(opusfe_ctx* ctx, uint64_t* bytes);
/* what the held bytes are: stats[0] live results (units not yet freed), [1] cached blocks (freed results kept for the next unit of
   the same size), [2] the temporaries' arena (one per context: the largest stage's need, not the sum), [3] the high-water mark of
   [0] + [1] + [2] + transient overflow allocations since opusfe_create */
int opusfe_memory_stats(opusfe_ctx* ctx, uint64_t stats[4]);

/* ---- results ---- */
/* number of resync passes opusfe_parse took for this unit (0 for well-formed input).  A byte-range shard of a
   larger source that needed one combines with its neighbours only if the lexer is back in code context at depth 0 at its end
   (opusfe_unit_end_state); else opus_amd/shard.py reports OPUSFE_UNSUPPORTED. */
int opusfe_unit_resync_passes(const opusfe_unit* unit, uint32_t* n);
/* the lexer state behind the unit's last byte as the reference's lexer has it (parse.rs:364-391) after every resync pass: context
   (0 code, 1-4 inside a string, 5 comment) and the bracket counter `ignore_dents`.  A byte-range shard that needed a resync pass
   composes with its right neighbour exactly when this is (0 or 5, 0): the reference then lexes the neighbour's bytes as the shard
   carries assumed (opus_amd/shard.py). */
int opusfe_unit_end_state(const opusfe_unit* unit, uint32_t* ctx_state, int32_t* depth);
int opusfe_unit_status(const opusfe_unit* unit);           /* OPUSFE_OK / _SYNTAX / _UNSUPPORTED ... of the unit */
int opusfe_get_counts(const opusfe_unit* unit, opusfe_counts* out);
int opusfe_get_error(const opusfe_unit* unit, opusfe_error* out);
/* copies device result arrays to caller-provided host buffers (cap in elements); returns element count via *n */
int opusfe_copy_tokens(opusfe_unit* unit, opusfe_token* dst, uint64_t cap, uint64_t* n);
/* ---- flat result arrays: what a host backend reads instead of IrGenerator's fields
   (generate_ir.rs:58-98 `variables`, `bbs`, `function`; llvm.rs:355-420 walks exactly these) ----
   Ids: variables and basic blocks are numbered per function from 0 (dst/a/b and Jump/Branch/Phi targets are those local
   numbers; add var_base / bb_base of the function for the global arrays); type ids index OPUSFE_A_TYPES directly.
   LAYOUT: OPUSFE_A_VARIABLES / _BB_OFFSETS / _INSTRS / _CALL_ARGS and the function-local part of OPUSFE_A_TYPES are DENSE.  The
   functions that the flat lowering takes (lower_par.h; all of a well-typed generated source but the ones of 2048 nodes and
   more) lie in DEFINITION ORDER in one range of every array — their places are an exclusive scan of their sizes — and that
   part is the same from run to run.  The functions of the sequential machine (oversized ones, and whatever the flat path
   declines: any function with a diagnostic) take their places with an atomic add each, before or behind that range, in the
   order in which they finish: their *_base fields (and the ids of their local types) can differ from run to run.  What is
   reached through a function's bases is always stable: the function's own records, its local numbering, and
   OPUSFE_A_FUNCTIONS itself (definition order).  Compare per function (opus_amd/ir_view.py does) unless
   opusfe_lowering_stats says that no function took the sequential machine. */
typedef struct opusfe_function_ir {      /* one per Definition::Function / extern, in definition order; n_bbs == 0: extern */
    uint32_t def_node;                   /* root node of the definition */
    uint32_t func;                       /* index into OPUSFE_A_SIGNATURES */
    uint32_t n_vars, n_instrs, n_bbs, n_types, n_call_args, n_errors;
    uint32_t var_base, instr_base, bb_base, type_base, call_arg_base, error_base;
    uint32_t status, _pad;
} opusfe_function_ir;
typedef struct opusfe_type {             /* compile.rs:45-75 Type, structural */
    uint8_t kind;                        /* variant index in compile.rs:45-75 order (Integer8 .. Error) */
    uint8_t ptr;                         /* Pointer: 0 @, 1 @mut, 2 {}, 3 mut {} (compile.rs:38-43 PointerType) */
    uint8_t src;                         /* Function: 1 = `ref` is a proc-type node, 2 = `ref` is a signature index */
    uint8_t _pad;
    uint32_t sub;                        /* Pointer / Array: element type id; Function: return type id */
    uint32_t ref;                        /* Record: record-type node (children = fields); Function: see src; Array: length token */
    uint32_t alias;                      /* name token of the `type` definition this id stands for, else 0xFFFFFFFF */
} opusfe_type;
typedef struct opusfe_signature {        /* compile.rs:300-352 Function / FunctionSignature */
    uint32_t def_node, name_tok, nargs, ret_type, first_arg /* node; siblings follow */, flags /* bit0 extern */;
} opusfe_signature;
typedef struct opusfe_global { uint32_t def_node, type_id, is_const, name_tok; } opusfe_global;
/* kind and span, plus what the message of compile.rs:816-915 prints besides the span: `name` = token index of the variable /
   field / function name (0x80000000 | token kind for a builtin operator symbol, 0xFFFFFFFF = none); `nargs` = number of argument
   types of a function identifier (compile.rs:115-122), the first four of them in t[]; for the typed kinds t[0], t[1] are the one
   or two types (expected, found / from, to / left, right); type ids index OPUSFE_A_TYPES.  A function identifier with more than four
   argument types is followed by CONTINUATION records (kind == OPUSFE_SEM_CONTINUATION): the eight words behind `kind` are the next
   eight argument type ids (0xFFFFFFFF = unused).  They are not diagnostics: opusfe_semantic_error_count does not count them. */
#define OPUSFE_SEM_CONTINUATION 0x100u
typedef struct opusfe_semantic_error { uint32_t kind /* compile.rs:124-152 variant index */, low, len, name, nargs, t[4]; } opusfe_semantic_error;

enum {
    OPUSFE_A_SOURCE = 0,      /* uint8_t   source bytes */
    OPUSFE_A_TOKENS = 1,      /* opusfe_token */
    OPUSFE_A_VAL64 = 2,       /* uint64_t  integer / float-bit / error payload pool */
    OPUSFE_A_STRINGS = 3,     /* uint8_t   string pool: [u32 n][n bytes] padded to 4 */
    OPUSFE_A_NODES = 4,       /* opusfe_node, dense, definitions in file order, each definition contiguous, nodes in the parser's creation order (operands before the operator / postfix node that takes them; NOT pre-order) */
    OPUSFE_A_DEFINITIONS = 5, /* uint32_t  root node of each definition */
    OPUSFE_A_SIGNATURES = 6,  /* opusfe_signature, one per function / extern (after opusfe_lower) */
    OPUSFE_A_GLOBALS = 7,     /* opusfe_global */
    OPUSFE_A_TYPES = 8,       /* opusfe_type (also after OPUSFE_SEMANTIC: the global records, which the finalize-time diagnostics name) */
    OPUSFE_A_NODE_TYPES = 9,  /* uint32_t  per node: resolved type id of a type node (compile.rs:919 resolve_type), else 0xFFFFFFFF */
    OPUSFE_A_FUNCTIONS = 10,  /* opusfe_function_ir */
    OPUSFE_A_VARIABLES = 11,  /* opusfe_variable  (generate_ir.rs:69 variables) */
    OPUSFE_A_BB_OFFSETS = 12, /* uint32_t  bbs + 1 entries: instructions of global bb g are [off[g], off[g+1]) */
    OPUSFE_A_INSTRS = 13,     /* opusfe_instr */
    OPUSFE_A_CALL_ARGS = 14,  /* uint32_t  variable ids of Call arguments (instr.c = offset, instr.sub = count) */
    OPUSFE_A_SEM_ERRORS = 15, /* opusfe_semantic_error, in report order */
    OPUSFE_A_PARENTS = 16,    /* uint32_t  per node: index of its parent node, 0xFFFFFFFF for the root of a definition.  Built by a node-parallel
                                 kernel on first request (device memory of the unit, freed with it); not an input of opusfe_unit_from_arrays.
                                 The reference has no such field: a Rust Box tree is walked from the root (parse.rs:253-328); flat consumers
                                 (tree-parallel passes like lower_par.h) need the upward link. */
    OPUSFE_A_COUNT = 17
};
/* device pointer (owned by the unit, valid until opusfe_free_unit / the next opusfe_lower), element count and size */
int opusfe_array(opusfe_unit* unit, int which, const void** d_ptr, uint64_t* count, uint32_t* elem_size);
/* device -> host copy of one array; dst == NULL only queries *n; cap in elements */
int opusfe_copy_array(opusfe_unit* unit, int which, void* dst, uint64_t cap, uint64_t* n);

/* canonical text dumps (formats: SURVEY.md §8c).  what: 0 tokens, 1 AST, 2 IR, 3 first-error line, 4 semantic errors of IR generation.
   *out is malloc'ed by the library; release with opusfe_free_text. */
int opusfe_dump(opusfe_unit* unit, int what, char** out, uint64_t* out_len);
void opusfe_free_text(char* p);
void opusfe_free_unit(opusfe_unit* unit);

/* ---- measurement hooks (bench.py): HIP-event timing of each kernel of the last call on ctx ---- */
int opusfe_set_profiling(opusfe_ctx* ctx, int enabled);
int opusfe_kernel_count(opusfe_ctx* ctx);
int opusfe_kernel_info(opusfe_ctx* ctx, int index, const char** name, double* ms, uint64_t* launches);

#ifdef __cplusplus
}
#endif
#endif

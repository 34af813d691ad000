// One translation unit's results (device-resident; host mirrors are filled lazily for dumps).
#pragma once
#include "runtime.h"

// A corpus of independent files packed into one unit (opusfe_parse_group, BASELINE config 5): file f is the bytes
// [start[f], start[f] + len[f]) of the unit's source (starts are multiples of 64, '\n' in between).  Device tables, owned by the group.
struct GroupInfo {
    uint32_t n_files = 0;
    uint32_t* d_start = nullptr; uint32_t* d_len = nullptr;
    uint32_t* d_flags = nullptr;        // lex.h FF_*: why a file needs the single-file path / has a syntax error
    uint32_t* d_fstatus = nullptr;      // [2 n_files] the lexer's unsupported-input record per file {code, first position}
    uint32_t* d_tok_begin = nullptr;    // first token of the file in the final token stream
    uint32_t* d_err_seg = nullptr;      // lowest failing segment of the file, NONE = none
    uint32_t* d_first_def = nullptr;    // first definition / function / global of the file (filled by the parser / the lowering)
    uint32_t* d_first_func = nullptr; uint32_t* d_first_global = nullptr;
};

struct opusfe_unit {
    opusfe_ctx* ctx = nullptr;
    GroupInfo* group = nullptr;                                // set while the unit is the packed source of a group
    int status = OPUSFE_OK;
    uint32_t dev_status = 0;
    // source
    const uint8_t* d_src = nullptr; bool own_src = false; uint64_t len = 0;
    std::vector<uint8_t> h_src; bool have_h_src = false;
    // lexer
    opusfe_token* d_tokens = nullptr; uint64_t n_tokens = 0;
    uint64_t* d_val64 = nullptr; uint64_t n_val64 = 0;
    uint8_t* d_strpool = nullptr; uint64_t n_strpool = 0;
    uint32_t* d_seg_start = nullptr; uint64_t n_segs = 0;
    // parser
    opusfe_node* d_nodes = nullptr; uint64_t node_cap = 0;     // sparse: segment s owns [node_base(s), +count)
    uint32_t* d_seg_info = nullptr;                            // per segment: node_base, node_count, first_def, def_count, err..., 8 words
    uint64_t n_nodes = 0, n_defs = 0, n_funcs = 0;
    uint64_t n_seg_sequential = 0;                             // segments parsed by SegParser (the rest by parse_par.h)
    uint32_t* d_parents = nullptr;                             // OPUSFE_A_PARENTS, built on first request
    uint32_t* d_def_nodes = nullptr;                           // definition index -> root node
    uint8_t* d_headers = nullptr; uint64_t n_headers = 0;      // definition headers (opusfe_unit_headers)
    bool parsed = false;
    bool has_error = false; opusfe_error error{}; uint32_t err_seg = 0;
    uint32_t unsup_pos = 0xFFFFFFFFu;                          // first byte the reference cannot be followed on (lexer)
    std::vector<uint32_t> resync;                              // sorted resync points found by earlier passes (lex.h Resync)
    uint32_t end_ctx_state = 0; int32_t end_depth = 0;         // lexer context and bracket depth behind the last byte (opusfe_unit_end_state)
    // IR
    bool lowered = false;
    void* ir = nullptr;                                        // opusfe::IrResult (lower.h)
    uint64_t n_instrs = 0, n_vars = 0, n_bbs = 0, n_types = 0;
    // host mirrors
    std::vector<opusfe_token> h_tokens; std::vector<uint64_t> h_val64; std::vector<uint8_t> h_strpool;
    std::vector<opusfe_node> h_nodes; std::vector<uint32_t> h_def_nodes;
    bool have_h_lex = false, have_h_ast = false;
};

// opusfe — joining two parsed units into one (multi-GPU IR generation).
//
// A rank that owns a byte range of a huge source has parsed it (unit A) and receives the other ranks' definition headers
// as text, parsed as unit H.  Lowering needs ONE unit whose definitions are in source order: the headers of the ranks
// before it, its own definitions, the headers of the ranks after it.  Parsing that concatenated text again would parse the
// rank's own gigabyte a second time; instead the arrays of A and H are concatenated and re-based:
//   source, tokens, literal pools : A then H                      (their order does not matter: nodes index into them)
//   nodes, definition table       : H[definitions before] , A , H[definitions after]   (node order IS definition order:
//                                   sem.h compares node ids to decide "defined before")
// The reference has no counterpart (it compiles one file on one thread).
#pragma once
#include "defs.h"

namespace opusfe {

struct JoinSplit { uint32_t k, n1; };   // definitions / nodes of H that go before A

// first definition of H that starts at or after byte `split_pos` of H's source (one thread)
__global__ void k_join_split(const opusfe_node* __restrict__ h_nodes, const uint32_t* __restrict__ h_defs, uint32_t h_ndefs, uint32_t h_nnodes,
                             uint32_t split_pos, JoinSplit* __restrict__ out) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    uint32_t lo = 0, hi = h_ndefs;
    while (lo < hi) { uint32_t mid = lo + (hi - lo) / 2; if (h_nodes[h_defs[mid]].low < split_pos) lo = mid + 1; else hi = mid; }
    out->k = lo;
    out->n1 = lo < h_ndefs ? h_defs[lo] : h_nnodes;
}

// H's tokens behind A's: positions and pool indices move (lex.h k_lex_emit: Error / Float / non-inline Integer tokens carry
// an index into val64, String tokens an offset into the string pool, every other payload is a value)
__global__ void k_join_tokens(const opusfe_token* __restrict__ h_tok, uint32_t n, opusfe_token* __restrict__ out,
                              uint32_t src_off, uint32_t val_off, uint32_t str_off) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    opusfe_token t = h_tok[i];
    t.low += src_off;
    if (t.kind == T_String) t.payload += str_off;
    else if (t.kind == T_Error || t.kind == T_Float || (t.kind == T_Integer && !(t.flags & OPUSFE_TF_INLINE))) t.payload += val_off;
    out[i] = t;
}

// node kinds whose payload is a token index (parse.h: every new_node / payload assignment)
__host__ __device__ __forceinline__ bool node_payload_is_token(uint8_t k) {
    switch (k) {
        case D_Function: case D_Extern: case D_Constant: case D_Type: case D_Variable: case D_Import: case D_Library:
        case E_Field: case E_Float: case E_Integer: case E_String: case E_Variable: case E_NamedCall: case E_Proc: case E_VariableDefinition:
        case S_For: case S_VariableDefinition: case X_FieldInit: case X_NamedType: case Y_Array: case Y_Name:
            return true;
        default: return false;
    }
}

__global__ void k_join_nodes(const opusfe_node* __restrict__ a_nodes, uint32_t a_n, const opusfe_node* __restrict__ h_nodes, uint32_t h_n,
                             const JoinSplit* __restrict__ split, uint32_t tok_off, uint32_t src_off, opusfe_node* __restrict__ out) {
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= a_n + h_n) return;
    const uint32_t n1 = split->n1;
    opusfe_node nd;
    if (j >= n1 && j < n1 + a_n) {                  // a node of A: only its links move
        nd = a_nodes[j - n1];
        if (nd.first_child != OPUSFE_NONE) nd.first_child += n1;
        if (nd.next != OPUSFE_NONE) nd.next += n1;
    } else {                                        // a node of H
        nd = h_nodes[j < n1 ? j : j - a_n];
        if (nd.first_child != OPUSFE_NONE && nd.first_child >= n1) nd.first_child += a_n;
        if (nd.next != OPUSFE_NONE && nd.next >= n1) nd.next += a_n;
        if (node_payload_is_token(nd.kind)) nd.payload += tok_off;
        nd.low += src_off; nd.high += src_off;
    }
    out[j] = nd;
}

__global__ void k_join_defs(const uint32_t* __restrict__ a_defs, uint32_t a_nd, uint32_t a_nnodes, const uint32_t* __restrict__ h_defs, uint32_t h_nd,
                            const JoinSplit* __restrict__ split, uint32_t* __restrict__ out) {
    uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= a_nd + h_nd) return;
    const uint32_t k = split->k, n1 = split->n1;
    if (d < k) out[d] = h_defs[d];
    else if (d < k + a_nd) out[d] = a_defs[d - k] + n1;
    else out[d] = h_defs[d - a_nd] + a_nnodes;
}

}  // namespace opusfe

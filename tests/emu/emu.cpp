// TEST INFRASTRUCTURE: host emulation of the opusfe device code (see fake/hip/hip_runtime.h).
// Runs the REAL kernel bodies of opus_amd/csrc (lex_one / lex_chunk, the token-level kernels, the
// segment parser, the lowering kernels) lane by lane on the CPU under AddressSanitizer so that
// indexing bugs are found here and not as GPU memory faults.  The sequencing below mirrors
// run_lexer / run_parser / run_lower of the product with every device-wide scan replaced by a
// sequential loop.  It is never part of the product and is not a CPU fallback.
#define OPUSFE_EMU 1
#include <hip/hip_runtime.h>
#include <string>
#include <vector>
#include "defs.h"
#include "runtime.h"
#include "unit.h"
#include "scan.h"
#include "lex.h"
#include "parse.h"
#include "dump.h"
#include "dump_ast.h"
#include "lower.h"
#include "dump_ir.h"

using namespace opusfe;

namespace {

inline unsigned grid_for(uint64_t n, unsigned block) { return (unsigned)((n + block - 1) / block); }

struct EmuUnit {
    opusfe_unit u;
    std::vector<uint8_t> src;
    std::vector<uint32_t> seg_start;
    std::vector<uint32_t> seg_info;
    std::vector<uint32_t> resync;     // api.hip body_parse: resync points found by earlier passes
    uint32_t status[64] = {0};
};

void emu_lex(EmuUnit& E, int force_dents) {
    const uint8_t* src = E.src.data();
    const uint32_t n = (uint32_t)E.u.len;
    uint32_t* status = E.status;
    memset(status, 0, sizeof E.status);
    status[1] = 0xFFFFFFFFu;
    // count pass / emit pass exactly like k_lex_count / k_lex_emit, with exact-size heap blocks so that
    // AddressSanitizer sees any write outside a token's reserved pool slice
    std::vector<uint8_t> chunk_state;
    Cnt3 tot{0, 0, 0, 0};
    const Resync rs{E.resync.data(), (uint32_t)E.resync.size()};
    {
        uint32_t state = CS_N;
        for (uint64_t cb = 0; cb < n; cb += LEX_CHUNK) {
            uint32_t cbeg = (uint32_t)cb, cend = (uint32_t)((cb + LEX_CHUNK < n) ? cb + LEX_CHUNK : n);
            chunk_state.push_back((uint8_t)state);
            lex_chunk(src, n, cbeg, cend, state, status, rs, [&](uint32_t pos) {
                LexTok t;
                lex_one(src, n, pos, t, nullptr, status, rs);
                tot.tok++; tot.val += t.nval; tot.str += t.strbytes;
                return t.end;
            });
            for (uint32_t p = cbeg; p < cend; p++) { if (rs_has(rs, p)) state = CS_N; state = ctx_apply(ctx_of_byte(src[p]), state); }   // k_ctx_chunk
        }
    }
    opusfe_token* raw_p = (opusfe_token*)malloc(sizeof(opusfe_token) * (tot.tok ? tot.tok : 1));
    uint64_t* val_p = (uint64_t*)malloc(8 * (tot.val ? tot.val : 1));
    uint8_t* pool_p = (uint8_t*)malloc(tot.str ? tot.str : 1);
    {
        uint32_t ti = 0, vi = 0, si = 0, c = 0;
        for (uint64_t cb = 0; cb < n; cb += LEX_CHUNK, c++) {
            uint32_t cbeg = (uint32_t)cb, cend = (uint32_t)((cb + LEX_CHUNK < n) ? cb + LEX_CHUNK : n);
            lex_chunk(src, n, cbeg, cend, chunk_state[c], status, rs, [&](uint32_t pos) {
                LexTok t;
                lex_one(src, n, pos, t, pool_p + si, status, rs);
                opusfe_token o;
                o.kind = t.kind; o.flags = t.flags; o.aux = t.aux; o.low = t.low; o.len = t.end - t.low;
                if (t.kind == T_Error) { o.low = pos; o.payload = vi; }
                else if (t.kind == T_String) o.payload = si;
                else if (t.nval) o.payload = vi;
                else o.payload = t.payload;
                if (t.nval >= 1) val_p[vi] = t.v0;
                if (t.nval >= 2) val_p[vi + 1] = t.v1;
                vi += t.nval; si += t.strbytes;
                raw_p[ti++] = o;
                return t.end;
            });
        }
        if (ti != tot.tok || vi != tot.val || si != tot.str) { fprintf(stderr, "emu: count/emit mismatch\n"); abort(); }
    }
    std::vector<opusfe_token> raw(raw_p, raw_p + tot.tok);
    E.u.h_val64.assign(val_p, val_p + tot.val);
    E.u.h_strpool.assign(pool_p, pool_p + tot.str);
    free(raw_p); free(val_p); free(pool_p);
    const uint32_t T = (uint32_t)raw.size();
    const uint64_t G = (uint64_t)T + 1;
    std::vector<int32_t> depth(G);
    std::vector<uint32_t> gapval(G), prev(G), cnt(G), out_off(G), seg_flag(G), seg_excl(G);
    if (T) {
        emu_launch(k_tok_delta, grid_for(T, 256), 256, raw.data(), T, depth.data(), force_dents, (uint64_t)TK_NO_PREV, FileTok{});
        for (uint32_t k = 1; k < T; k++) depth[k] += depth[k - 1];
    }
    emu_launch(k_gap, grid_for(G, 256), 256, src, n, raw.data(), T, depth.data(), gapval.data(), (int32_t)0, FileTok{});
    { uint32_t run = 0; for (uint64_t k = 0; k < G; k++) { prev[k] = run; run = OpLast()(run, gapval[k]); } }
    emu_launch(k_dent_count, grid_for(G, 256), 256, gapval.data(), prev.data(), T, cnt.data(), FileTok{}, (const int32_t*)depth.data());
    uint32_t total = 0;
    for (uint64_t k = 0; k < G; k++) { out_off[k] = total; total += cnt[k]; }
    E.u.h_tokens.assign(total, opusfe_token{});
    emu_launch(k_final, grid_for(G, 256), 256, raw.data(), T, n, gapval.data(), prev.data(), out_off.data(), E.u.h_tokens.data(), seg_flag.data(), FileTok{}, (const int32_t*)depth.data(), (uint32_t*)nullptr);
    uint32_t nseg = 0;
    for (uint32_t k = 0; k < T; k++) { seg_excl[k] = nseg; nseg += seg_flag[k]; }
    E.seg_start.assign((size_t)T + 1, 0);
    if (T) emu_launch(k_seg_starts, grid_for(T, 256), 256, seg_flag.data(), seg_excl.data(), out_off.data(), cnt.data(), T, E.seg_start.data());
    E.u.n_tokens = total; E.u.n_segs = nseg; E.u.n_val64 = E.u.h_val64.size(); E.u.n_strpool = E.u.h_strpool.size();
    E.u.have_h_lex = true; E.u.have_h_src = true;
    E.u.dev_status = status[0]; E.u.unsup_pos = status[1];
    E.u.status = status[0] == DS_OK ? OPUSFE_OK : status[0] >= DS_INTERNAL_CAPACITY ? OPUSFE_ERR_INTERNAL : OPUSFE_UNSUPPORTED;
}

template <class T> T* exact(size_t n) { return (T*)calloc(n ? n : 1, sizeof(T)); }   // exact-size heap blocks: ASan sees overruns

void emu_parse(EmuUnit& E) {
    opusfe_unit& u = E.u;
    const uint32_t n_segs = (uint32_t)u.n_segs, n_tok = (uint32_t)u.n_tokens;
    u.parsed = true; u.n_nodes = u.n_defs = 0;
    if (n_segs == 0) { if (u.status == OPUSFE_OK || E.status[0] == DS_OK) u.status = E.status[0] == DS_OK ? OPUSFE_OK : OPUSFE_UNSUPPORTED; return; }
    // product sequencing of parse.h run_parser: the data-parallel parser (parse_par.h) first, SegParser on what it declines.
    // EMU_PARSER_SEQ=1 sends every segment to SegParser (opusfe_set_parser_mode(ctx, 1)).
    E.seg_info.assign((size_t)n_segs * SI_WORDS, 0);
    uint32_t first_err = 0xFFFFFFFFu;
    uint32_t work[pp::PW_WORDS + 2] = {0};
    std::vector<uint32_t> decl_list(n_segs);
    pp::Rec16* prec = exact<pp::Rec16>((size_t)n_tok + 1);
    pp::StackEnt* pstack = exact<pp::StackEnt>((size_t)n_tok / 2 + n_segs + 2);
    const pp::Table& PT = pp::table();
    if (!PT.ok) { fprintf(stderr, "emu: the grammar table derivation is inconsistent\n"); abort(); }
    if (getenv("EMU_PARSER_SEQ")) emu_launch(pp::k_pp_all_sequential, grid_for(n_segs, 256), 256, (const uint32_t*)E.seg_start.data(), n_segs, n_tok, E.seg_info.data(), work, decl_list.data());
    else {
        pp::RolesArgs A{};
        std::vector<uint8_t> kinds(n_tok);
        emu_launch(pp::k_tok_kinds, grid_for(n_tok, 256), 256, (const opusfe_token*)u.h_tokens.data(), n_tok, kinds.data());
        A.kinds = kinds.data();
        A.tok = u.h_tokens.data(); A.n_tokens = n_tok; A.seg_start = E.seg_start.data(); A.n_segs = n_segs; A.order = nullptr;
        A.table = PT.blob.data(); A.n_states = PT.n_states; A.start = PT.start; A.prec = prec; A.stack = pstack;
        A.seg_info = E.seg_info.data(); A.work = work; A.decl_list = decl_list.data();
        emu_launch(pp::k_pp_roles, grid_for(n_segs, 256), 256, A);
    }
    const uint32_t n_decl = work[pp::PW_DECLINED];
    u.n_seg_sequential = n_decl;
    if (getenv("EMU_REQUIRE_PP") && n_decl) { fprintf(stderr, "emu: %u of %u segments declined by the data-parallel parser\n", n_decl, n_segs); free(prec); free(pstack); u.status = OPUSFE_ERR_INTERNAL; return; }
    uint64_t sparse_cap = 0, frame_cap = 0;
    memcpy(&sparse_cap, work + pp::PW_SPARSE, 8); memcpy(&frame_cap, work + pp::PW_FRAMES, 8);
    // exact-size scratch so that AddressSanitizer sees every overrun
    std::vector<opusfe_node> sparse(sparse_cap + 8);
    std::vector<PFrame> frames(frame_cap + 16);
    if (n_decl)
        emu_launch(k_parse_segments, grid_for(n_decl, 64), 64, u.h_tokens.data(), n_tok, E.seg_start.data(), n_segs, sparse.data(), frames.data(),
                   E.seg_info.data(), E.status, &first_err, (const uint32_t*)decl_list.data(), n_decl);
    if (getenv("EMU_VERBOSE")) for (uint32_t s = 0; s < n_segs; s++) { uint32_t k = E.seg_info[(size_t)s * SI_WORDS + SI_ERR_KIND]; if (k != OPUSFE_NONE && (k & 0x200u)) fprintf(stderr, "segment %u of %u: internal condition %u (first error segment %u)\n", s, n_segs, k & 0xFF, first_err); }
    if (getenv("EMU_VERBOSE")) fprintf(stderr, "parser: %u of %u segments by SegParser\n", n_decl, n_segs);
    std::vector<Cnt2> cnt(n_segs);
    emu_launch(k_ast_counts, grid_for(n_segs, 256), 256, E.seg_info.data(), n_segs, cnt.data());
    Cnt2 run{0, 0};
    for (uint32_t s = 0; s < n_segs; s++) { Cnt2 c = cnt[s]; cnt[s] = run; run = OpAddCnt2()(run, c); }
    u.n_nodes = run.nodes; u.n_defs = run.defs;
    struct FreeOnExit { void* a; void* b; ~FreeOnExit() { free(a); free(b); } } free_on_exit{prec, pstack};
    const uint32_t dev_status = E.status[0];
    if (first_err != 0xFFFFFFFFu) {
        const uint32_t* si = E.seg_info.data() + (size_t)first_err * SI_WORDS;
        const opusfe_token& t = u.h_tokens[si[SI_ERR_TOK]];
        opusfe_error e{};
        e.tok_index = si[SI_ERR_TOK];
        if (si[SI_ERR_KIND] & 0x200u) { u.status = OPUSFE_ERR_INTERNAL; u.dev_status = si[SI_ERR_KIND] & 0xFFu; return; }
        if (si[SI_ERR_KIND] & 0x100u) {
            uint64_t v0 = u.h_val64[t.payload], v1 = u.h_val64[t.payload + 1];
            e.kind = si[SI_ERR_KIND] & 0xFFu; e.low = (uint32_t)v0; e.len = (uint32_t)(v0 >> 32); e.payload = v1; e.tok_kind = T_Error;
        } else { e.kind = si[SI_ERR_KIND]; e.low = si[SI_ERR_LOW]; e.len = si[SI_ERR_LEN]; e.expected = si[SI_ERR_EXPECTED]; e.tok_kind = t.kind; }
        const uint32_t unsup_pos = E.status[1];
        if (unsup_pos < t.low + t.len || (t.kind == T_Error && unsup_pos <= t.low)) { u.status = OPUSFE_UNSUPPORTED; u.dev_status = dev_status; return; }
        uint32_t lines = 0;
        for (uint64_t p = 0; p < e.low && p < u.len; p++) lines += E.src[p] == '\n';   // k_count_newlines is a shuffle reduction
        e.line = 1 + lines;
        u.error = e; u.has_error = true; u.err_seg = first_err; u.status = OPUSFE_SYNTAX; u.dev_status = DS_OK;
        return;
    }
    if (dev_status != DS_OK) { u.status = OPUSFE_UNSUPPORTED; u.dev_status = dev_status; return; }
    {   // exact-size result blocks: a node written outside the counted total is an ASan error
        opusfe_node* dn = exact<opusfe_node>(u.n_nodes); uint32_t* dd = exact<uint32_t>(u.n_defs);
        memset(dn, 0xEE, (size_t)u.n_nodes * sizeof(opusfe_node));
        if (n_decl < n_segs)
            emu_launch(pp::k_pp_build, grid_for(n_tok, 256), 256, (const opusfe_token*)u.h_tokens.data(), n_tok, (const uint32_t*)E.seg_start.data(), n_segs, (const pp::Rec16*)prec,
                       (const uint32_t*)E.seg_info.data(), (const pp::NodeBase*)cnt.data(), dn, dd);
        if (n_decl) emu_launch(k_ast_compact, 2, 256, (const opusfe_node*)sparse.data(), (const uint32_t*)decl_list.data(), n_decl, E.seg_info.data(), (const Cnt2*)cnt.data(), dn, dd);
        u.h_nodes.assign(dn, dn + u.n_nodes); u.h_def_nodes.assign(dd, dd + u.n_defs);
        free(dn); free(dd);
    }
    u.have_h_ast = true;
    u.status = OPUSFE_OK;
}



void emu_lower(EmuUnit& E) {
    opusfe_unit& u = E.u;
    IrResult* R = new IrResult();
    u.ir = R; u.lowered = true;
    const uint32_t n_defs = (uint32_t)u.n_defs, n_nodes = (uint32_t)u.n_nodes;
    R->have_host = true;
    if (n_defs == 0) return;
    const opusfe_node* nodes = u.h_nodes.data();
    const opusfe_token* tok = u.h_tokens.data();
    const uint32_t* def_nodes = u.h_def_nodes.data();
    std::vector<DefCnt> def_excl(n_defs);
    emu_launch(k_sem_classify, grid_for(n_defs, 256), 256, nodes, def_nodes, n_defs, def_excl.data());
    DefCnt dt{0, 0, 0, 0};
    for (uint32_t d = 0; d < n_defs; d++) { DefCnt c = def_excl[d]; def_excl[d] = dt; dt = OpAddDefCnt()(dt, c); }
    std::vector<uint32_t> slot_of_node(n_nodes);
    emu_launch(k_sem_type_flags, grid_for(n_nodes, 256), 256, nodes, n_nodes, slot_of_node.data());
    uint32_t gt = 0;
    for (uint32_t i = 0; i < n_nodes; i++) { uint32_t f = slot_of_node[i]; slot_of_node[i] = gt; gt += f; }
    const uint32_t n_funcs = dt.funcs, n_globals = dt.globals, n_named = dt.named, n_gt0 = TP_COUNT + gt, pp_base = n_gt0, n_gtypes = n_gt0 + par::PP_COUNT;
    uint32_t n_slots = 16; while (n_slots < 2ull * n_named + 16) n_slots <<= 1;
    NameSlot* slots = exact<NameSlot>(n_slots);
    NameEntry* entries = exact<NameEntry>(n_named + 1);
    std::vector<uint32_t> def_slot(n_defs), slot_off(n_slots);
    emu_launch(k_sem_slots_init, grid_for(n_slots, 256), 256, slots, n_slots);
    emu_launch(k_sem_table_insert, grid_for(n_defs, 256), 256, E.src.data(), tok, nodes, def_nodes, n_defs, slots, n_slots - 1, def_slot.data());
    emu_launch(k_sem_slot_counts, grid_for(n_slots, 256), 256, slots, n_slots, slot_off.data());
    { uint32_t run = 0; for (uint32_t i = 0; i < n_slots; i++) { uint32_t c = slot_off[i]; slot_off[i] = run; run += c; } }
    emu_launch(k_sem_table_fill, grid_for(n_defs, 256), 256, nodes, def_nodes, n_defs, def_excl.data(), def_slot.data(), slot_off.data(), slots, entries);
    emu_launch(k_sem_table_sort, grid_for(n_slots, 256), 256, slots, n_slots, slot_off.data(), entries);
    TypeRec* gtypes = exact<TypeRec>(n_gtypes);
    R->tid_of_node.assign(n_nodes, 0);
    std::vector<uint8_t> undefined_flag(n_nodes);
    std::vector<uint32_t> ph_def(n_gt0), ph_prev(n_gt0);
    R->funcrecs.assign(n_funcs, FuncRec{}); R->globals.assign(n_globals, GlobalRec{});
    emu_launch(k_sem_resolve_types, grid_for(std::max<uint32_t>(n_nodes, TP_COUNT), 256), 256, E.src.data(), tok, nodes, n_nodes, slot_of_node.data(), slots, n_slots - 1,
               entries, R->tid_of_node.data(), gtypes, undefined_flag.data(), ph_def.data(), ph_prev.data());
    emu_launch(k_sem_pp_types, 1, 128, gtypes + pp_base);
    uint32_t finerr = 0;
    emu_launch(k_sem_finalize, grid_for(n_defs, 256), 256, nodes, def_nodes, n_defs, def_excl.data(), R->tid_of_node.data(), undefined_flag.data(), ph_def.data(), ph_prev.data(), n_nodes,
               gtypes, R->funcrecs.data(), R->globals.data(), &finerr);
    R->n_funcs = n_funcs; R->n_globals = n_globals; R->n_gtypes = n_gtypes;
    if (finerr) {
        std::vector<uint32_t> fin_cnt(n_defs);
        emu_launch(k_sem_fin_count, grid_for(n_defs, 256), 256, nodes, def_nodes, n_defs, (const uint32_t*)R->tid_of_node.data(), (const uint8_t*)undefined_flag.data(),
                   (const uint32_t*)ph_def.data(), (const uint32_t*)ph_prev.data(), n_nodes, (const TypeRec*)gtypes, fin_cnt.data());
        { uint32_t run = 0; for (uint32_t d = 0; d < n_defs; d++) { uint32_t c = fin_cnt[d]; fin_cnt[d] = run; run += c; } }
        SemErr* errs = exact<SemErr>(finerr);   // exact-size block: an emit beyond the counted total is an ASan error
        emu_launch(k_sem_fin_emit, grid_for(n_defs, 256), 256, nodes, def_nodes, n_defs, (const uint32_t*)R->tid_of_node.data(), (const uint8_t*)undefined_flag.data(),
                   (const uint32_t*)ph_def.data(), (const uint32_t*)ph_prev.data(), n_nodes, (const TypeRec*)gtypes, (const uint32_t*)fin_cnt.data(), errs);
        R->errs.assign(errs, errs + finerr); R->n_errs = finerr;
        free(errs);
        R->n_sem_errors = finerr; u.status = OPUSFE_SEMANTIC; free(slots); free(entries); free(gtypes); return;
    }
    if (n_funcs == 0) { free(slots); free(entries); free(gtypes); return; }
    // node-parallel path (lower_par.h), one emulated lane per function; EMU_PAR_NMAX / EMU_LOWER_SEQ steer what falls back,
    // EMU_PAR_TINYCAP makes the first attempt overflow so that the exact-size second attempt is exercised
    SemView S{E.src.data(), tok, u.h_val64.data(), nodes, R->tid_of_node.data(), gtypes, slots, n_slots - 1, entries, R->funcrecs.data(), R->globals.data(), n_gtypes};
    R->funcs.assign(n_funcs, FuncIr{});
    uint32_t lstatus[2] = {0, 0};
    uint64_t user_syms = 0;
    emu_launch(k_sem_user_syms, 1, 1, S, &user_syms);
    const bool seq_only = getenv("EMU_LOWER_SEQ") != nullptr;
    uint64_t cap[5];
    for (int k = 0; k < 5; k++) cap[k] = getenv("EMU_PAR_TINYCAP") ? 3 : 8ull * n_nodes + 64;
    std::vector<uint32_t> fb_flag(n_funcs);
    std::vector<FbIdx> fbx(n_funcs);
    bool cleanup_only = false;
    for (int attempt = 0; attempt < 2; attempt++) {
        lstatus[0] = lstatus[1] = 0;
        uint32_t n_bodies = 0;
        unsigned long long counters[par::CTR_COUNT + 2] = {0};
        opusfe_variable* dv = exact<opusfe_variable>(cap[0]); opusfe_instr* di = exact<opusfe_instr>(cap[1]);
        uint32_t* dbo = exact<uint32_t>(cap[2] + 1); uint32_t* dc = exact<uint32_t>(cap[4]);
        TypeRec* dty = exact<TypeRec>(n_gtypes + cap[3]); SemErr* de = nullptr;
        memcpy(dty, gtypes, (size_t)n_gtypes * sizeof(TypeRec));
        std::vector<par::FnInfo> fninfo(n_funcs); std::vector<uint32_t> fbw(n_funcs);
        emu_launch(par::k_par_init, grid_for(n_defs, 256), 256, nodes, def_nodes, n_defs, (const DefCnt*)def_excl.data(), R->funcs.data(), &n_bodies, (const uint8_t*)nullptr, fninfo.data(), fbw.data());
        par::ParArgs PA{};
        PA.def_nodes = def_nodes; PA.n_defs = n_defs; PA.n_nodes = n_nodes; PA.def_excl = def_excl.data(); PA.undefined_flag = undefined_flag.data(); PA.user_syms = &user_syms;
        PA.pp_base = pp_base; PA.out = R->funcs.data(); PA.status = lstatus;
        PA.counters = counters; PA.overflow = (uint32_t*)(counters + par::CTR_COUNT); PA.err_counter = (uint32_t*)(counters + par::CTR_COUNT + 1);
        PA.cap_vars = cap[0]; PA.cap_instrs = cap[1]; PA.cap_bbs = cap[2]; PA.cap_types = cap[3]; PA.cap_cargs = cap[4]; PA.cap_errs = ~0ull;     // (the dense diagnostics block below has the sparse capacity)
        PA.vars = dv; PA.instrs = di; PA.bb_offsets = dbo; PA.cargs = dc; PA.types_out = dty;
        PA.list = nullptr; PA.n_list = n_defs; PA.nmax = getenv("EMU_PAR_NMAX") ? (uint32_t)atoi(getenv("EMU_PAR_NMAX")) : 2048u; PA.maxdef = getenv("EMU_PAR_MAXDEF") ? (uint32_t)atoi(getenv("EMU_PAR_MAXDEF")) : 256u;
        S.types = gtypes;
        if (!seq_only) {
            // the flat passes of lower_par.h in the product's order (run_lower), every list and per-node array an exact-size heap block
            par::FlatArgs X{};
            uint32_t* ctl = exact<uint32_t>(par::CTL_WORDS);
            X.ctl = ctl; X.fninfo = fninfo.data(); X.fb = fbw.data();
            X.rec = exact<par::NodeRec>(n_nodes); X.cnt = exact<uint64_t>(2ull * n_nodes); X.tail = exact<uint8_t>(2ull * n_nodes); X.cg = exact<uint16_t>(n_nodes);
            X.key1 = exact<uint16_t>(n_nodes);
            memset(X.key1, 0xFF, 2ull * n_nodes); memset(X.tail, 0xFF, 2ull * n_nodes);      // (the bins of the schedule: NONE16 = none)
            memset((void*)X.rec, 0xFF, sizeof(par::NodeRec) * (size_t)n_nodes); memset(X.cg, 0xA5, 2ull * n_nodes);
            memset(X.cnt, 0, 16ull * n_nodes);      // as run_lower clears them
            const size_t lds = par::lds_bytes(PA.nmax, PA.maxdef, 1);
            par::g_emu_lds = (uint32_t*)malloc(lds);
            memset(par::g_emu_lds, 0xA5, lds);
            emu_launch(par::k_flat_pre, n_defs, 1, S, PA, X);
            free(par::g_emu_lds); par::g_emu_lds = nullptr;
            // the schedule: counting sort of the (class, round) bins, then one launch per bin in round order (run_lower's plan)
            emu_launch(par::k_sched_pass<false>, 1, 1, X, n_nodes, n_nodes, 0u);
            emu_launch(par::k_sched_scan, 1, 1, ctl);
            const uint32_t* off = ctl + par::CTL_BINS;
            const uint32_t n_sched = off[par::SCHED_BINS], n_list = n_sched;
            X.sched = exact<uint32_t>(n_sched); X.order = exact<uint32_t>(n_sched);
            emu_launch(par::k_sched_pass<true>, 1, 1, X, n_nodes, n_nodes, n_sched);
            memset(X.tail, 0, 2ull * n_nodes);
            const uint32_t any_below = getenv("EMU_FLAT_TAIL") ? (uint32_t)atoi(getenv("EMU_FLAT_TAIL")) : 64u;
            auto round_nodes = [&](uint32_t r) { uint64_t n = 0; for (uint32_t c = 0; c < par::KC_ACTIVE; c++) n += off[c * par::SCHED_R + r + 1] - off[c * par::SCHED_R + r]; return n; };
            uint32_t rounds = 0, tail_from = 0;
            const uint32_t tail_below = getenv("EMU_FLAT_TAILKERNEL") ? (uint32_t)atoi(getenv("EMU_FLAT_TAILKERNEL")) : 16u;
            while (rounds < par::FLAT_MAX_ROUNDS && round_nodes(rounds)) { if (round_nodes(rounds) > tail_below) tail_from = rounds + 1; rounds++; }
            for (uint32_t r = 0; r < rounds; r++) {
                const uint64_t n_round = round_nodes(r);
                if (r >= tail_from && rounds - r >= 2) { emu_launch(par::k_flat_count_tail, 1, 1, S, PA, X, r, rounds); break; }
                if (n_round <= any_below) { emu_launch(par::k_flat_count_round, 1, 1, S, PA, X, r); continue; }
#define EMU_COUNT(C) emu_launch(par::k_flat_count<C>, 1, 1, S, PA, X, off[C * par::SCHED_R + r], off[C * par::SCHED_R + r + 1] - off[C * par::SCHED_R + r])
                EMU_COUNT(0); EMU_COUNT(1); EMU_COUNT(2); EMU_COUNT(3); EMU_COUNT(4); EMU_COUNT(5); EMU_COUNT(6); EMU_COUNT(7);
            }
#define EMU_CLASSES(K, ...) do { emu_launch(par::K<0>, 1, 1, __VA_ARGS__); emu_launch(par::K<1>, 1, 1, __VA_ARGS__); emu_launch(par::K<2>, 1, 1, __VA_ARGS__); emu_launch(par::K<3>, 1, 1, __VA_ARGS__); \
                                     emu_launch(par::K<4>, 1, 1, __VA_ARGS__); emu_launch(par::K<5>, 1, 1, __VA_ARGS__); emu_launch(par::K<6>, 1, 1, __VA_ARGS__); emu_launch(par::K<7>, 1, 1, __VA_ARGS__); } while (0)
            std::vector<IrCnt> fn_tot(n_funcs + 1, IrCnt{0, 0, 0, 0, 0, 0});
            emu_launch(par::k_flat_scan, grid_for(n_defs, 256), 256, S, PA, X, fn_tot.data());
            { IrCnt run{0, 0, 0, 0, 0, 0}; for (uint32_t k = 0; k < n_funcs; k++) { const IrCnt c = fn_tot[k]; fn_tot[k] = run; run = OpAddIrCnt()(run, c); } fn_tot[n_funcs] = run; }
            fn_tot.push_back(IrCnt{0, 0, 0, 0, 0, 0});
            emu_launch(par::k_flat_reserve, 1, 1, PA, (const IrCnt*)(fn_tot.data() + n_funcs), fn_tot.data() + n_funcs + 1);
            emu_launch(par::k_flat_bases, grid_for(n_funcs, 256), 256, PA, X, (const IrCnt*)fn_tot.data(), (const IrCnt*)(fn_tot.data() + n_funcs + 1), n_funcs);
            emu_launch(par::k_flat_rv, 1, 1, PA, X);
            EMU_CLASSES(k_flat_emit, S, PA, X);
            if (ctl[par::CTL_LISTOVF]) lstatus[0] = DS_INTERNAL_CAPACITY;
            if (getenv("EMU_VERBOSE")) fprintf(stderr, "flat lowering: %u rounds, %u listed nodes of %u\n", rounds, n_list, n_nodes);
            free(ctl); free(X.rec); free(X.cnt); free(X.tail); free(X.cg); free(X.key1); free(X.sched); free(X.order);
        }
        emu_launch(k_fb_mark, grid_for(n_defs, 256), 256, (const FuncIr*)R->funcs.data(), n_funcs, def_nodes, n_defs, n_nodes, (const DefCnt*)def_excl.data(), nodes, fb_flag.data(), fbx.data());
        FbIdx fbt{0, 0};
        for (uint32_t i = 0; i < n_funcs; i++) { FbIdx c = fbx[i]; fbx[i] = fbt; fbt = OpAddFbIdx()(fbt, c); }
        const uint32_t fb_nodes = fbt.nodes, fb_funcs = fbt.funcs;
        R->n_fallback = fb_funcs;
        if (getenv("EMU_VERBOSE")) { fprintf(stderr, "functions lowered by the sequential machine: %u of %u\n", fb_funcs, n_funcs); for (auto& fr : R->funcs) if (fr.status == par::DS_FALLBACK) fprintf(stderr, "  function %u (node %u): declined at lower_par.h:%u\n", fr.func, fr.def_node, fr.pad); }
        if (getenv("EMU_REQUIRE_PAR") && fb_funcs) { fprintf(stderr, "emu: %u functions fell back\n", fb_funcs); u.status = OPUSFE_ERR_INTERNAL; cleanup_only = true; }
        LowerBuffers B{};
        B.type_sparse_base = n_gtypes;
        TypeRec* types = nullptr;
        if (fb_funcs && !cleanup_only) {
            const uint64_t type_cap = (uint64_t)n_gtypes + lower_base(CAP_TYPES, fb_nodes, fb_funcs) + CAP_CONST;
            types = exact<TypeRec>(type_cap);
            memcpy(types, gtypes, (size_t)n_gtypes * sizeof(TypeRec));
            S.types = types;
            B.vars = exact<opusfe_variable>(lower_base(CAP_VARS, fb_nodes, fb_funcs) + CAP_CONST);
            const uint64_t icap = lower_base(CAP_INSTRS, fb_nodes, fb_funcs) + CAP_CONST;
            B.instrs = exact<opusfe_instr>(icap); B.ibb = exact<uint32_t>(icap); B.irank = exact<uint32_t>(icap);
            const uint64_t bcap = lower_base(CAP_BBS, fb_nodes, fb_funcs) + CAP_CONST;
            B.bbcnt = exact<uint32_t>(bcap); B.bblast = exact<uint32_t>(bcap);
            B.cargs = exact<uint32_t>(lower_base(CAP_CARGS, fb_nodes, fb_funcs) + CAP_CONST);
            const uint64_t lcap = lower_base(CAP_LOCALS, fb_nodes, fb_funcs) + CAP_CONST;
            B.lname = exact<uint32_t>(lcap); B.lvar = exact<uint32_t>(lcap); B.lstack = exact<uint32_t>(lcap);
            B.frames = exact<LFrame>(lower_base(CAP_FRAMES, fb_nodes, fb_funcs) + CAP_CONST);
            const uint64_t ecap = lower_base(CAP_ERRS, fb_nodes, fb_funcs) + CAP_CONST;
            B.errs = exact<SemErr>(ecap);
            de = exact<SemErr>(ecap);
            emu_launch(k_lower_functions, grid_for(n_defs, 64), 64, S, B, (const uint8_t*)undefined_flag.data(), def_nodes, n_defs, n_nodes, (uint32_t)u.n_tokens, (const DefCnt*)def_excl.data(),
                       R->funcs.data(), lstatus, (uint64_t*)nullptr, (const uint64_t*)&user_syms, (const uint32_t*)nullptr, (const uint32_t*)fb_flag.data(), (const FbIdx*)fbx.data());
            emu_launch(k_fb_bases, grid_for(n_funcs, 64), 64, R->funcs.data(), n_funcs, B, PA, (const uint32_t*)fb_flag.data(), (const FbIdx*)fbx.data());
            IrOut O{dv, di, dc, dty, de, dbo, n_gtypes};
            emu_launch(k_ir_compact, 2, 256, (const FuncIr*)R->funcs.data(), n_funcs, B, (const TypeRec*)types, O, (const uint32_t*)fb_flag.data(), (const FbIdx*)fbx.data());
        }
        const uint64_t tot[5] = {counters[par::CTR_VARS], counters[par::CTR_INSTR_BB] & ((1ull << par::CTR_BB_SHIFT) - 1), counters[par::CTR_INSTR_BB] >> par::CTR_BB_SHIFT, counters[par::CTR_TYPES], counters[par::CTR_CARGS]};
        const bool overflow = (uint32_t)counters[par::CTR_COUNT] != 0;
        if (lstatus[0] >= DS_INTERNAL_CAPACITY) { u.status = OPUSFE_ERR_INTERNAL; cleanup_only = true; }
        else if (lstatus[0] != DS_OK) { u.status = OPUSFE_UNSUPPORTED; u.dev_status = lstatus[0]; cleanup_only = true; }
        if (!cleanup_only && overflow && attempt == 1) { u.status = OPUSFE_ERR_INTERNAL; cleanup_only = true; }
        const bool again = !cleanup_only && overflow;
        if (!cleanup_only && !overflow) {
            R->n_sem_errors = lstatus[1];
            R->n_vars = tot[0]; R->n_instrs = tot[1]; R->n_bbs = tot[2]; R->n_types = (uint64_t)n_gtypes + tot[3]; R->n_cargs = tot[4]; R->n_errs = (uint32_t)counters[par::CTR_COUNT + 1];
            dbo[R->n_bbs] = (uint32_t)R->n_instrs;
            R->vars.assign(dv, dv + R->n_vars); R->instrs.assign(di, di + R->n_instrs); R->bb_offsets.assign(dbo, dbo + R->n_bbs + 1);
            R->cargs.assign(dc, dc + R->n_cargs); R->types.assign(dty, dty + R->n_types); R->errs.assign(de, de + R->n_errs);
            u.n_funcs = n_funcs; u.n_instrs = R->n_instrs; u.n_vars = R->n_vars;
        }
        free(dv); free(di); free(dbo); free(dc); free(dty); free(de); free(types);
        free(B.vars); free(B.instrs); free(B.ibb); free(B.irank); free(B.bbcnt); free(B.bblast); free(B.cargs); free(B.lname); free(B.lvar); free(B.lstack);
        free(B.frames); free(B.errs);
        if (!again) break;
        for (int k = 0; k < 5; k++) cap[k] = tot[k] + 16;
    }
    free(slots); free(entries); free(gtypes);
}

}  // namespace

extern "C" {

// what: 0 tokens, 1 AST, 2 IR, 3 error line, 4 semantic errors, 5 lexer alone (force_dents).  *status receives the OPUSFE_* code.
char* emu_dump(const uint8_t* src, uint64_t len, int what, uint64_t* out_len, int* status) {
    EmuUnit E;
    E.src.assign(src, src + len);
    E.src.resize(len + 16, 0);          // the product pads device sources the same way
    E.u.len = len;
    E.u.h_src.assign(src, src + len);
    for (;;) {   // api.hip body_parse: lex and parse again with every resync point a pass finds
        emu_lex(E, what == 5);
        if (what != 5 && (E.u.status == OPUSFE_OK || E.u.status == OPUSFE_UNSUPPORTED)) emu_parse(E);
        if (what == 5 || E.resync.size() >= 16 || E.u.status != OPUSFE_UNSUPPORTED || E.u.dev_status != DS_UNSUPPORTED_RESYNC) break;
        uint32_t r = OPUSFE_NONE;
        for (const opusfe_token& t : E.u.h_tokens)
            if (t.low == E.status[1] && t.kind == T_Error && (t.flags & OPUSFE_TF_RESYNC)) { uint64_t sp = E.u.h_val64[t.payload]; r = (uint32_t)sp + (uint32_t)(sp >> 32); break; }
        if (r == OPUSFE_NONE || r > E.u.len || (!E.resync.empty() && r <= E.resync.back())) break;
        E.resync.push_back(r);
        E.u.has_error = false; E.u.parsed = false; E.u.have_h_ast = false;
    }
    if ((what == 2 || what == 4) && E.u.status == OPUSFE_OK) emu_lower(E);
    std::string s;
    switch (what) {
        case 0: case 5: s = dump_tokens(&E.u, E.u.has_error ? (uint64_t)E.u.error.tok_index + 1 : ~0ull); break;
        case 1: s = E.u.status == OPUSFE_OK ? dump_ast(&E.u) : std::string(); break;
        case 2: s = dump_ir(&E.u); break;
        case 4: s = dump_semantic_errors(&E.u); break;
        case 3: s = E.u.status == OPUSFE_ERR_INTERNAL ? std::string("Internal\n") : dump_error_line(&E.u); break;
        default: break;
    }
    if (status) *status = E.u.status;
    delete (IrResult*)E.u.ir;
    if (out_len) *out_len = s.size();
    char* p = (char*)malloc(s.size() + 1);
    memcpy(p, s.data(), s.size()); p[s.size()] = 0;
    return p;
}
void emu_free(char* p) { free(p); }

}

#ifdef EMU_MAIN
#include <fstream>
#include <sstream>
// emu_cli <what> <file>   -> dump on stdout, "status N" on stderr
int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: emu_cli <what> <file>...\n"); return 2; }
    int what = atoi(argv[1]);
    for (int a = 2; a < argc; a++) {
        std::ifstream f(argv[a], std::ios::binary);
        std::stringstream ss; ss << f.rdbuf();
        std::string src = ss.str();
        uint64_t n; int status;
        char* s = emu_dump((const uint8_t*)src.data(), src.size(), what, &n, &status);
        printf("== %s status %d bytes %llu\n", argv[a], status, (unsigned long long)n);
        fwrite(s, 1, n, stdout);
        emu_free(s);
    }
#ifdef OPUSFE_EMU_COUNT
    for (auto& kv : emu_counts()) fprintf(stderr, "%12llu %s\n", (unsigned long long)kv.second, kv.first.c_str());
#endif
    return 0;
}
#endif
